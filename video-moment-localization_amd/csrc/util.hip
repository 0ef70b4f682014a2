// Library identity, workspace sizing, slab reduction, dense<->packed layout helpers, stand-alone GEMM.
#include "gemm.h"
#include "smin_hip.h"
#include <stdlib.h>
#include <mutex>
#include <vector>

namespace smin {

int g_gemm_mode = 0;

// see launch_gemm_tn (gemm.h).  The exact-fp32 weight-gradient kernel is built for three workgroups per CU (32 KB of LDS, 168
// registers: 3 x 168 of a SIMD's 512).  In the train step it runs on the low-priority stream beside the critical chain, whose
// kernels then find no registers free until a 130 us tile retires (moment_dfb 93 -> 340 us, proposal_map_bwd_events2 485 -> 790 us
// beside it).  24 KB of unused dynamic LDS per workgroup hold it to TWO per CU: measured 18.40 -> 18.25 ms/step on one box
// (one per CU: 19.5).  SMIN_DW_EXTRA_LDS overrides (bytes; 0 = three per CU).
int tn_extra_lds()
{
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("SMIN_DW_EXTRA_LDS");
        v = e ? atoi(e) : 24576;
        if (v < 0) v = 0;
        if (v > 96 * 1024) v = 96 * 1024;
    }
    return v;
}

// ---- launch timing (smin_prof_*): autograd runs backward on its own thread, hence the lock
volatile int g_prof_on = 0;
struct ProfRec { int tag; hipEvent_t e0, e1; bool closed; };
static std::mutex g_prof_mu;
static std::vector<ProfRec> g_prof;

void prof_record(hipStream_t st, int tag, bool begin)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (begin) {
        ProfRec r{tag, nullptr, nullptr, false};
        if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return;
        (void)hipEventRecord(r.e0, st);
        g_prof.push_back(r);
    } else {
        for (size_t k = g_prof.size(); k-- > 0;)
            if (g_prof[k].tag == tag && !g_prof[k].closed) { (void)hipEventRecord(g_prof[k].e1, st); g_prof[k].closed = true; break; }
    }
}

__global__ void reduce_slabs_kernel(const float* __restrict__ slab, float* __restrict__ out, int n, int P)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    // eight independent partial sums keep eight loads in flight (a single running sum is one HBM round trip per slab);
    // the association is fixed, so the result is still bitwise reproducible.
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int z = 0;
    for (; z + 8 <= P; z += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) s[u] += slab[(size_t)(z + u) * n + idx];
    }
    for (; z < P; ++z) s[0] += slab[(size_t)z * n + idx];
    out[idx] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
}

// the weight and bias slabs of one contraction in one launch (same fixed association as reduce_slabs_kernel)
// (outB2: optional second copy of the bias result -- two parameters that receive the same gradient each get a tensor of their own)
__global__ void reduce_slabs2_kernel(const float* __restrict__ slabA, float* __restrict__ outA, int nA,
                                     const float* __restrict__ slabB, float* __restrict__ outB, int nB, int P, float* __restrict__ outB2)
{
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const float* slab = slabA; float* out = outA; int n = nA;
    float* out2 = nullptr;
    if (idx >= nA) { idx -= nA; slab = slabB; out = outB; n = nB; out2 = outB2; }
    if (idx >= n) return;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int z = 0;
    for (; z + 8 <= P; z += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) s[u] += slab[(size_t)(z + u) * n + idx];
    }
    for (; z < P; ++z) s[0] += slab[(size_t)z * n + idx];
    const float v = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    out[idx] = v;
    if (out2) out2[idx] = v;
}

// The same sums for MANY slabs of few outputs (P >= 64: the score head's per-chunk partials, 1 575 x 513; a dl x dl weight gradient
// cut into 768 row splits): a thread per output walked P rows on a handful of workgroups (71 us for 3 MB on the critical path at the
// start of the backward pass).  Here a workgroup owns 32 outputs and eight threads share each: thread (column, phase) sums rows
// phase, phase + 8, ... with four sums in flight, the phases meet in LDS in phase order.  Fixed association: reproducible.
__global__ __launch_bounds__(256)
void reduce_slabs_wide_kernel(const float* __restrict__ slabA, float* __restrict__ outA, int nA,
                              const float* __restrict__ slabB, float* __restrict__ outB, int nB, int P, float* __restrict__ outB2)
{
    __shared__ float part[8][32];
    const int col = threadIdx.x & 31, ph = threadIdx.x >> 5;
    int idx = blockIdx.x * 32 + col;
    const int blocksA = (nA + 31) / 32;
    const float* slab = slabA; float* out = outA; int n = nA; float* out2 = nullptr;
    if ((int)blockIdx.x >= blocksA) { idx = ((int)blockIdx.x - blocksA) * 32 + col; slab = slabB; out = outB; n = nB; out2 = outB2; }
    const bool ok = idx < n;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    if (ok) {
        int z = ph;
        for (; z + 24 < P; z += 32) {
#pragma unroll
            for (int u = 0; u < 4; ++u) s[u] += slab[(size_t)(z + 8 * u) * n + idx];
        }
        for (; z < P; z += 8) s[0] += slab[(size_t)z * n + idx];
    }
    part[ph][col] = (s[0] + s[1]) + (s[2] + s[3]);
    __syncthreads();
    if (ph == 0 && ok) {
        float v = part[0][col];
#pragma unroll
        for (int q = 1; q < 8; ++q) v += part[q][col];
        out[idx] = v;
        if (out2) out2[idx] = v;
    }
}

int launch_reduce_slabs2(hipStream_t st, const float* slabA, float* outA, int nA, const float* slabB, float* outB, int nB, int P, float* outB2)
{
    if (nA <= 0 || nB <= 0 || !outB) return launch_reduce_slabs(st, slabA, outA, nA, P);
    if (P >= 64) {
        hipLaunchKernelGGL(reduce_slabs_wide_kernel, dim3(cdiv(nA, 32) + cdiv(nB, 32)), dim3(256), 0, st, slabA, outA, nA, slabB, outB, nB, P, outB2);
        SMIN_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(reduce_slabs2_kernel, dim3(cdiv(nA + nB, 256)), dim3(256), 0, st, slabA, outA, nA, slabB, outB, nB, P, outB2);
    SMIN_LAUNCH_CHECK();
    return 0;
}

int launch_reduce_slabs(hipStream_t st, const float* slab, float* out, int n, int P)
{
    if (n <= 0) return 0;
    if (P >= 64) {
        hipLaunchKernelGGL(reduce_slabs_wide_kernel, dim3(cdiv(n, 32)), dim3(256), 0, st, slab, out, n, (const float*)nullptr, (float*)nullptr, 0, P, (float*)nullptr);
        SMIN_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, slab, out, n, P);
    SMIN_LAUNCH_CHECK();
    return 0;
}

__global__ void pack_cells_kernel(const float* __restrict__ dense, const int* __restrict__ cells, int N, int L, int W4, float* __restrict__ packed)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)N * W4) return;
    const size_t n = idx / W4; const int k = (int)(idx % W4);
    const Cell c = load_cell(cells, (int)n);
    const size_t src = (((size_t)c.b * L + c.i) * L + c.j) * W4 + k;
    stg4(packed + idx * 4, ldg4(dense + src * 4));
}

__global__ void unpack_cells_kernel(const float* __restrict__ packed, const int* __restrict__ cells, int N, int L, int W4, float* __restrict__ dense)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)N * W4) return;
    const size_t n = idx / W4; const int k = (int)(idx % W4);
    const Cell c = load_cell(cells, (int)n);
    const size_t dst = (((size_t)c.b * L + c.i) * L + c.j) * W4 + k;
    stg4(dense + dst * 4, ldg4(packed + idx * 4));
}

struct EpStore {
    float* out;
    __device__ __forceinline__ void chunk(const float* Ws, int row0, int col0, int ncols, int M, int N, int lane) const {
        chunk_rows_f4(Ws, row0, col0, ncols, M, N, lane, [&](int row, int col, float4 v) { stg4(out + (size_t)row * N + col, v); });
    }
};

}  // namespace smin

using namespace smin;

extern "C" int smin_abi_version(void) { return SMIN_HIP_ABI_VERSION; }
extern "C" int smin_set_gemm_mode(int mode)
{
    if (mode < 0 || mode > 3) return -1;
    g_gemm_mode = mode;
    return 0;
}
extern "C" int smin_get_gemm_mode(void) { return g_gemm_mode; }
extern "C" const char* smin_target_arch(void) { return "gfx950"; }

extern "C" int smin_prof_enable(int on)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (on) {
        for (auto& r : g_prof) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
        g_prof.clear();
    }
    g_prof_on = on ? 1 : 0;
    return 0;
}

extern "C" int smin_prof_read(int32_t* tags, float* ms, int cap)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    int n = 0;
    for (auto& r : g_prof) {
        if (n >= cap) break;
        if (!r.closed) continue;
        if (hipEventSynchronize(r.e1) != hipSuccess) return -1;
        float t = 0.f;
        if (hipEventElapsedTime(&t, r.e0, r.e1) != hipSuccess) return -2;
        tags[n] = r.tag; ms[n] = t; ++n;
    }
    return n;
}

extern "C" size_t smin_workspace_bytes(int N, int B, int C, int D, int dl, int Nq)
{
    (void)Nq;
    const size_t M = (size_t)N * C;
    // content unit bwd: 2 x [M][dl] + TN slabs + attention slabs + gate partials
    const size_t sp_c = (size_t)(M > 0 ? tn_splits((int)M, D, dl) + tn_splits((int)M, dl, D) : 2);
    size_t content = 3 * M * dl + 2 * M * 32 + sp_c * ((size_t)D * dl + D + dl) + (size_t)B * 64 * ((size_t)64 * dl + dl + 32) + (size_t)B * 512 * D;
    // moment unit bwd: dX1 [N][D] + TN slabs + bias slabs
    size_t moment = (size_t)N * D + (size_t)(N > 0 ? tn_splits(N, D, 2 * D) : 1) * ((size_t)D * 2 * D + D);
    // boundary / score: per-row partials
    size_t other = (size_t)N + 8 * (size_t)B * 64 * D + (size_t)N * 4 + (size_t)(N / 32 + 1) * (D + 4);
    // boundary unit bwd (L <= 64 assumed here; the host adds 2*B*L*L + 3*B*L*D for longer maps)
    other += (size_t)B * 64 * 64 * 2 + (size_t)B * 64 * D * 3 + (size_t)B * 64 * 64 + (size_t)B * 64 * D + 2 * (size_t)64 * ((size_t)D * D + D);
    size_t fl = content > moment ? content : moment;
    if (other > fl) fl = other;
    return (fl + 1024) * sizeof(float);
}

extern "C" int smin_pack_cells(void* stream, const float* dense, const int32_t* cells, int N, int L, int W, float* packed)
{
    SMIN_REQUIRE(W % 4 == 0);
    if (N == 0) return 0;
    const size_t tot = (size_t)N * (W / 4);
    hipLaunchKernelGGL(pack_cells_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dense, cells, N, L, W / 4, packed);
    SMIN_LAUNCH_CHECK();
    return 0;
}

extern "C" int smin_unpack_cells(void* stream, const float* packed, const int32_t* cells, int N, int L, int W, float* dense)
{
    SMIN_REQUIRE(W % 4 == 0);
    if (N == 0) return 0;
    const size_t tot = (size_t)N * (W / 4);
    hipLaunchKernelGGL(unpack_cells_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, packed, cells, N, L, W / 4, dense);
    SMIN_LAUNCH_CHECK();
    return 0;
}

// ---- small host helpers of the fused step (torch_binding.cpp): many tiny matrices / vectors in one launch each
namespace smin {
struct TransposeBatch { const float* src[SMIN_BATCH_MAX]; float* dst[SMIN_BATCH_MAX]; int rows[SMIN_BATCH_MAX], cols[SMIN_BATCH_MAX], tile0[SMIN_BATCH_MAX + 1]; int n; };

// dst[c][r] = src[r][c] for every matrix of the batch; one 32x32 tile per workgroup (256 threads), tiles of all matrices in one grid
__global__ __launch_bounds__(256)
void transpose_batch_kernel(TransposeBatch tb)
{
    __shared__ float t[32][33];
    int m = 0;
    while (m + 1 < tb.n && (int)blockIdx.x >= tb.tile0[m + 1]) ++m;
    const int rows = tb.rows[m], cols = tb.cols[m], tc = (cols + 31) / 32;
    const int tile = blockIdx.x - tb.tile0[m], r0 = (tile / tc) * 32, c0 = (tile % tc) * 32;
    const int x = threadIdx.x & 31, y = threadIdx.x >> 5;
    const float* __restrict__ src = tb.src[m];
    float* __restrict__ dst = tb.dst[m];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = r0 + y + 8 * k, c = c0 + x;
        t[y + 8 * k][x] = (r < rows && c < cols) ? src[(size_t)r * cols + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = c0 + y + 8 * k, r = r0 + x;
        if (r < rows && c < cols) dst[(size_t)c * rows + r] = t[x][y + 8 * k];
    }
}

struct SumList { const float* p[SMIN_BATCH_MAX]; int n; };
// out[i] = sum_k p[k][i], in list order (deterministic); float4 columns, scalar tail
template <int NFIX>
__global__ __launch_bounds__(256)
void sum_lists_kernel(SumList sl, size_t numel, float* __restrict__ out)
{
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= numel) return;
    const int n = NFIX ? NFIX : sl.n;
    if (i + 4 <= numel) {
        float4 s = ldg4(sl.p[0] + i);
        if (NFIX) {
            float4 v[NFIX ? NFIX : 1];
#pragma unroll
            for (int k = 1; k < NFIX; ++k) v[k] = ldg4(sl.p[k] + i);
#pragma unroll
            for (int k = 1; k < NFIX; ++k) s = f4add(s, v[k]);
        } else {
            for (int k = 1; k < n; ++k) s = f4add(s, ldg4(sl.p[k] + i));
        }
        stg4(out + i, s);
    } else {
        for (size_t j = i; j < numel; ++j) {
            float s = sl.p[0][j];
            for (int k = 1; k < n; ++k) s += sl.p[k][j];
            out[j] = s;
        }
    }
}
}  // namespace smin

extern "C" int smin_transpose_batch(void* stream, const float* const* src, float* const* dst, const int32_t* rows, const int32_t* cols, int n)
{
    SMIN_REQUIRE(n >= 0 && n <= SMIN_BATCH_MAX);
    if (n == 0) return 0;
    TransposeBatch tb;
    int tiles = 0;
    for (int m = 0; m < n; ++m) {
        SMIN_REQUIRE(rows[m] > 0 && cols[m] > 0);
        tb.src[m] = src[m]; tb.dst[m] = dst[m]; tb.rows[m] = rows[m]; tb.cols[m] = cols[m]; tb.tile0[m] = tiles;
        tiles += cdiv(rows[m], 32) * cdiv(cols[m], 32);
    }
    tb.tile0[n] = tiles; tb.n = n;
    hipLaunchKernelGGL(transpose_batch_kernel, dim3(tiles), dim3(256), 0, (hipStream_t)stream, tb);
    SMIN_LAUNCH_CHECK();
    return 0;
}

extern "C" int smin_sum_lists(void* stream, const float* const* srcs, int n, size_t numel, float* out)
{
    SMIN_REQUIRE(n >= 1 && n <= SMIN_BATCH_MAX);
    if (numel == 0) return 0;
    SumList sl;
    for (int k = 0; k < n; ++k) sl.p[k] = srcs[k];
    sl.n = n;
    for (int k = 0; k < n; ++k) SMIN_REQUIRE(((uintptr_t)srcs[k] & 15) == 0);
    SMIN_REQUIRE(((uintptr_t)out & 15) == 0);
    const dim3 grid((unsigned)((numel + 1023) / 1024));
    if (n == 2) hipLaunchKernelGGL(sum_lists_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, sl, numel, out);
    else if (n == 4) hipLaunchKernelGGL(sum_lists_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, sl, numel, out);
    else hipLaunchKernelGGL(sum_lists_kernel<0>, grid, dim3(256), 0, (hipStream_t)stream, sl, numel, out);
    SMIN_LAUNCH_CHECK();
    return 0;
}

namespace smin {
// partial[blockIdx.x][:] = sum of rows [blockIdx.x * rpb, +rpb) of x [R][4*W4]; 256 threads = W4 float4 columns x (256 / W4) row groups,
// four rows in flight per thread, the groups' sums meet in LDS in group order (fixed order: deterministic)
__global__ __launch_bounds__(256)
void col_sum_kernel(const float* __restrict__ x, int R, int W4, int rpb, float* __restrict__ partial)
{
    __shared__ float4 sh[256];
    const int groups = 256 / W4, g = threadIdx.x / W4, c = threadIdx.x - g * W4;
    const int r0 = blockIdx.x * rpb, r1 = min(R, r0 + rpb);
    float4 s = f4zero();
    if (g < groups) {
        const float* p = x + (size_t)c * 4;
        const size_t W = (size_t)W4 * 4;
        int i = r0 + g;
        for (; i + 3 * groups < r1; i += 4 * groups) {
            const float4 a0 = ldg4(p + (size_t)i * W), a1 = ldg4(p + (size_t)(i + groups) * W), a2 = ldg4(p + (size_t)(i + 2 * groups) * W),
                         a3 = ldg4(p + (size_t)(i + 3 * groups) * W);
            s = f4add(f4add(f4add(f4add(s, a0), a1), a2), a3);
        }
        for (; i < r1; i += groups) s = f4add(s, ldg4(p + (size_t)i * W));
    }
    sh[threadIdx.x] = s;
    __syncthreads();
    if (g == 0) {
        for (int q = 1; q < groups; ++q) s = f4add(s, sh[q * W4 + c]);
        stg4(partial + ((size_t)blockIdx.x * W4 + c) * 4, s);
    }
}
static int col_sum_blocks(int R) { return R <= 0 ? 1 : cdiv(R, max(256, cdiv(R, 1024))); }
}  // namespace smin

extern "C" size_t smin_col_sum_workspace_bytes(int R, int W) { return sizeof(float) * (size_t)smin::col_sum_blocks(R) * (size_t)(W > 0 ? W : 0) + 64; }

extern "C" int smin_col_sum(void* stream, const float* x, int R, int W, float* out, void* ws, size_t ws_bytes)
{
    using namespace smin;
    SMIN_REQUIRE(W % 4 == 0 && W >= 4 && W <= 1024 && R >= 0);
    hipStream_t st = (hipStream_t)stream;
    if (R == 0) return (int)hipMemsetAsync(out, 0, sizeof(float) * (size_t)W, st);
    SMIN_REQUIRE(ws_bytes >= smin_col_sum_workspace_bytes(R, W));
    const int nb = col_sum_blocks(R), rpb = cdiv(R, nb);
    float* partial = reinterpret_cast<float*>(ws);
    hipLaunchKernelGGL(col_sum_kernel, dim3(nb), dim3(256), 0, st, x, R, W / 4, rpb, nb == 1 ? out : partial);
    SMIN_LAUNCH_CHECK();
    if (nb > 1) {
        hipLaunchKernelGGL(col_sum_kernel, dim3(1), dim3(256), 0, st, partial, nb, W / 4, nb, out);
        SMIN_LAUNCH_CHECK();
    }
    return 0;
}

namespace smin {
struct EpAccumTest {                // out += acc (same visit as the boundary unit's accumulate epilogue)
    float* out;
    __device__ __forceinline__ void chunk(const float* Ws, int row0, int col0, int ncols, int M, int N, int lane) const {
        chunk_rows_f4(Ws, row0, col0, ncols, M, N, lane, [&](int row, int col, float4 v) {
            float* o = out + (size_t)row * N + col;
            stg4(o, f4add(v, ldg4(o)));
        });
    }
};
}  // namespace smin
/* C[M][N] += A[M][K] * B[N][K]^T  (test entry: the accumulate epilogue of the engine) */
extern "C" int smin_gemm_nt_acc(void* stream, const float* A, const float* Bm, float* Cm, int M, int N, int K)
{
    SMIN_REQUIRE(K % 4 == 0 && N % 4 == 0);
    return launch_gemm_nt((hipStream_t)stream, PlainMat{A, K}, PlainMat{Bm, K}, EpAccumTest{Cm}, M, N, K);
}

extern "C" int smin_gemm_nt(void* stream, const float* A, const float* Bm, float* Cm, int M, int N, int K)
{
    SMIN_REQUIRE(K % 4 == 0 && N % 4 == 0);
    return launch_gemm_nt((hipStream_t)stream, PlainMat{A, K}, PlainMat{Bm, K}, EpStore{Cm}, M, N, K);
}
