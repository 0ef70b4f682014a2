// BoundaryUnit, map-sized part (reference models.py:190-194): the gated row reduction of the moment map
//   fbm[b,i,:] = sum_j A_b[b,i,j] * sigmoid(f_m[b,i,j,:] * f_s[b,:]) * f_m[b,i,j,:]
// HBM-bound (reads f_m once).  The L x L boundary self-attention that produces A_b (models.py:164-188) is
// O(B L^2 D) and runs on the host side as library GEMMs; this file owns everything that touches the map.
#include "common.h"
#include "smin_hip.h"

namespace smin {

__global__ __launch_bounds__(128)
void boundary_reduce_fwd_kernel(const float* __restrict__ Ab, const float* __restrict__ fm, const float* __restrict__ fs,
                                const int* __restrict__ cells, const int* __restrict__ row_ptr, int L, int D, float* __restrict__ fbm)
{
    const int i = blockIdx.x, b = blockIdx.y;
    const int r0 = row_ptr[b * L + i], r1 = row_ptr[b * L + i + 1];
    const float* arow = Ab + ((size_t)b * L + i) * L;
    for (int d = threadIdx.x * 4; d < D; d += 512) {
        const float4 s4 = ldg4(fs + (size_t)b * D + d);
        float4 acc = f4zero();
        for (int n = r0; n < r1; ++n) {
            const float a = arow[cells[4 * (size_t)n + 2]];
            const float4 x = ldg4(fm + (size_t)n * D + d);
            acc.x = fmaf(a, x.x / (1.0f + expf(-x.x * s4.x)), acc.x);
            acc.y = fmaf(a, x.y / (1.0f + expf(-x.y * s4.y)), acc.y);
            acc.z = fmaf(a, x.z / (1.0f + expf(-x.z * s4.z)), acc.z);
            acc.w = fmaf(a, x.w / (1.0f + expf(-x.w * s4.w)), acc.w);
        }
        stg4(fbm + ((size_t)b * L + i) * D + d, acc);
    }
}

// one workgroup (4 waves) per row (b, i); each wave owns cells r0 + wave, r0 + wave + 4, ...
//   h = g * fm, g = sigmoid(fm * fs)
//   dAb[b,i,j] = <dfbm[b,i,:], h[n,:]>                 dfm[n,:] = Ab * dfbm * (g + fm*g*(1-g)*fs)
//   dfs[b,:]  += Ab * dfbm * fm^2 * g*(1-g)            (per-row partial, reduced over rows afterwards)
__global__ __launch_bounds__(256)
void boundary_reduce_bwd_kernel(const float* __restrict__ dfbm, const float* __restrict__ Ab, const float* __restrict__ fm,
                                const float* __restrict__ fs, const int* __restrict__ cells, const int* __restrict__ row_ptr,
                                int L, int D, float* __restrict__ dAb, float* __restrict__ dfm, float* __restrict__ partial)
{
    __shared__ __attribute__((aligned(16))) float red[4 * 1024];
    const int i = blockIdx.x, b = blockIdx.y;
    const int r0 = row_ptr[b * L + i], r1 = row_ptr[b * L + i + 1];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float* arow = Ab + ((size_t)b * L + i) * L;
    float* darow = dAb + ((size_t)b * L + i) * L;
    float4 acc[4], dy[4], s4[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int d = lane * 4 + 256 * k;
        acc[k] = f4zero();
        dy[k] = d < D ? ldg4(dfbm + ((size_t)b * L + i) * D + d) : f4zero();
        s4[k] = d < D ? ldg4(fs + (size_t)b * D + d) : f4zero();
    }
    for (int n = r0 + wave; n < r1; n += 4) {
        const int j = cells[4 * (size_t)n + 2];
        const float a = arow[j];
        float dot = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int d = lane * 4 + 256 * k;
            if (d < D) {
                const float4 x = ldg4(fm + (size_t)n * D + d);
                float4 o;
#define BR1(F)                                                                       \
                {                                                                    \
                    const float g = 1.0f / (1.0f + expf(-x.F * s4[k].F));            \
                    const float gg = g * (1.0f - g);                                 \
                    const float u = a * dy[k].F;                                     \
                    dot = fmaf(dy[k].F, g * x.F, dot);                               \
                    o.F = u * (g + x.F * gg * s4[k].F);                              \
                    acc[k].F = fmaf(u, x.F * x.F * gg, acc[k].F);                    \
                }
                BR1(x) BR1(y) BR1(z) BR1(w)
#undef BR1
                stg4(dfm + (size_t)n * D + d, o);
            }
        }
        dot = wave_sum(dot);
        if (lane == 0) darow[j] = dot;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int d = lane * 4 + 256 * k;
        if (d < D) stg4(red + wave * 1024 + d, acc[k]);
    }
    __syncthreads();
    for (int d = threadIdx.x; d < D; d += 256)
        partial[((size_t)b * L + i) * D + d] = (red[d] + red[1024 + d]) + (red[2048 + d] + red[3072 + d]);
}

__global__ void rows_reduce_kernel(const float* __restrict__ partial, int L, int D, float* __restrict__ out)
{
    const int b = blockIdx.y;
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= D) return;
    float s = 0.f;
    for (int i = 0; i < L; ++i) s += partial[((size_t)b * L + i) * D + d];
    out[(size_t)b * D + d] = s;
}

}  // namespace smin

using namespace smin;

extern "C" int smin_boundary_reduce_fwd(void* stream, const float* Ab, const float* fm, const float* fs, const int32_t* cells,
                                        const int32_t* row_ptr, int N, int B, int L, int D, float* fbm)
{
    (void)N;
    SMIN_REQUIRE(D % 4 == 0);
    hipLaunchKernelGGL(boundary_reduce_fwd_kernel, dim3(L, B), dim3(128), 0, (hipStream_t)stream, Ab, fm, fs, cells, row_ptr, L, D, fbm);
    SMIN_LAUNCH_CHECK();
    return 0;
}

extern "C" int smin_boundary_reduce_bwd(void* stream, const float* dfbm, const float* Ab, const float* fm, const float* fs,
                                        const int32_t* cells, const int32_t* row_ptr, int N, int B, int L, int D,
                                        float* dAb, float* dfm, float* dfs, void* ws, size_t ws_bytes)
{
    (void)N;
    hipStream_t st = (hipStream_t)stream;
    SMIN_REQUIRE(D % 4 == 0 && D <= 1024);
    SMIN_REQUIRE(ws_bytes >= sizeof(float) * (size_t)B * L * D);
    float* partial = reinterpret_cast<float*>(ws);
    hipError_t e = hipMemsetAsync(dAb, 0, sizeof(float) * (size_t)B * L * L, st);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(boundary_reduce_bwd_kernel, dim3(L, B), dim3(256), 0, st, dfbm, Ab, fm, fs, cells, row_ptr, L, D, dAb, dfm, partial);
    SMIN_LAUNCH_CHECK();
    hipLaunchKernelGGL(rows_reduce_kernel, dim3(cdiv(D, 256), B), dim3(256), 0, st, partial, L, D, dfs);
    SMIN_LAUNCH_CHECK();
    return 0;
}
