// Shared device helpers for the SMIN hot-path kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define SMIN_LAUNCH_CHECK()                                  \
    do {                                                     \
        hipError_t e__ = hipGetLastError();                  \
        if (e__ != hipSuccess) return (int)e__;              \
    } while (0)

#define SMIN_REQUIRE(cond)                                   \
    do {                                                     \
        if (!(cond)) return -1000 - __LINE__;                \
    } while (0)

// benchmark launch timing (smin_prof_enable / smin_prof_read): HIP events on the launch stream around a scope
namespace smin {
extern volatile int g_prof_on;
void prof_record(hipStream_t st, int tag, bool begin);
struct ProfScope {
    hipStream_t st; int tag; bool on;
    ProfScope(hipStream_t s, int t) : st(s), tag(t), on(g_prof_on != 0) { if (on) prof_record(st, tag, true); }
    ~ProfScope() { if (on) prof_record(st, tag, false); }
};
}  // namespace smin

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// chunking of each sample's cell range over workgroups.
// coarse (attention kernels: every workgroup first stages its sample's K/V tiles in LDS): <= 64 chunks, >= 64 cells each
static inline void chunking(int L, int* cells_per_chunk, int* max_chunks)
{
    const int max_cells = L * L;
    int cpc = cdiv(max_cells, 64);
    if (cpc < 64) cpc = 64;
    *cells_per_chunk = cpc;
    *max_chunks = cdiv(max_cells, cpc);
}
// fine (streaming element-wise kernels with a per-sample reduction): <= 512 chunks, >= 32 cells each
static inline void chunking_fine(int L, int* cells_per_chunk, int* max_chunks)
{
    const int max_cells = L * L;
    int mc = cdiv(max_cells, 32);
    if (mc > 512) mc = 512;
    if (mc < 1) mc = 1;
    const int cpc = cdiv(max_cells, mc);
    *cells_per_chunk = cpc;
    *max_chunks = cdiv(max_cells, cpc);
}

// One packed cell of the L x L proposal map: (sample, start snippet, end snippet, mask).
struct Cell { int b, i, j, m; };

__device__ __forceinline__ Cell load_cell(const int* __restrict__ cells, int n) {
    const int4 v = *reinterpret_cast<const int4*>(cells + 4 * (size_t)n);
    Cell c; c.b = v.x; c.i = v.y; c.j = v.z; c.m = v.w; return c;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

// ---- wave64 all-reduce through DPP (no LDS traffic) -------------------------------------
// quad_perm [1,0,3,2]=0xB1, [2,3,0,1]=0x4E, row_half_mirror=0x141, row_mirror=0x140,
// row_bcast15=0x142 (rows 1,3), row_bcast31=0x143 (rows 2,3); total lands in lane 63.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_f<0xB1, 0xF>(v);
    v += dpp_f<0x4E, 0xF>(v);
    v += dpp_f<0x141, 0xF>(v);
    v += dpp_f<0x140, 0xF>(v);
    v += dpp_f<0x142, 0xA>(v);
    v += dpp_f<0x143, 0xC>(v);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
// reduction inside each 32-lane half (every lane of the half gets its half's result)
__device__ __forceinline__ float half_sum(float v) {
    v += dpp_f<0xB1, 0xF>(v);
    v += dpp_f<0x4E, 0xF>(v);
    v += dpp_f<0x141, 0xF>(v);
    v += dpp_f<0x140, 0xF>(v);          // every lane: its 16-lane row total
    v += __shfl_xor(v, 16);             // other row of the same half
    return v;
}
__device__ __forceinline__ float half_max(float v) {
    v = fmaxf(v, __shfl_xor(v, 1));
    v = fmaxf(v, __shfl_xor(v, 2));
    v = fmaxf(v, __shfl_xor(v, 4));
    v = fmaxf(v, __shfl_xor(v, 8));
    v = fmaxf(v, __shfl_xor(v, 16));
    return v;
}

__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4mul(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float4 f4scale(float4 a, float s) { return make_float4(a.x * s, a.y * s, a.z * s, a.w * s); }
__device__ __forceinline__ float4 f4fma(float4 a, float s, float4 c) { return make_float4(fmaf(a.x, s, c.x), fmaf(a.y, s, c.y), fmaf(a.z, s, c.z), fmaf(a.w, s, c.w)); }
__device__ __forceinline__ float4 ldg4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void stg4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
