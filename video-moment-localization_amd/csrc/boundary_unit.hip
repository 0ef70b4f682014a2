// BoundaryUnit, map-sized part (reference models.py:190-194): the gated row reduction of the moment map
//   fbm[b,i,:] = sum_j A_b[b,i,j] * hbar[(b,i,j),:],   hbar = sigmoid(f_m * f_s) * f_m  (gate.hip)
// HBM-bound (reads hbar once).  The L x L boundary self-attention that produces A_b (models.py:164-188) is
// O(B L^2 D) and runs on the host side as library GEMMs; this file owns everything that touches the map.
#include "common.h"
#include "smin_hip.h"

namespace smin {

// fbm[b,i,:] = sum over the cells (i, j) of row i of  A_b[b,i,j] * hbar[n,:]
__global__ __launch_bounds__(128)
void boundary_reduce_fwd_kernel(const float* __restrict__ Ab, const float* __restrict__ hbar,
                                const int* __restrict__ cells, const int* __restrict__ row_ptr, int L, int D, float* __restrict__ fbm)
{
    const int i = blockIdx.x, b = blockIdx.y;
    const int r0 = row_ptr[b * L + i], r1 = row_ptr[b * L + i + 1];
    const float* arow = Ab + ((size_t)b * L + i) * L;
    for (int d = threadIdx.x * 4; d < D; d += 512) {
        float4 acc = f4zero();
        for (int n = r0; n < r1; ++n)
            acc = f4fma(ldg4(hbar + (size_t)n * D + d), arow[cells[4 * (size_t)n + 2]], acc);
        stg4(fbm + ((size_t)b * L + i) * D + d, acc);
    }
}

// one workgroup (4 waves) per row (b, i); each wave owns cells r0 + wave, r0 + wave + 4, ...
//   dAb[b,i,j] = <dfbm[b,i,:], hbar[n,:]>          dhbar[n,:] = Ab[b,i,j] * dfbm[b,i,:]
__global__ __launch_bounds__(256)
void boundary_reduce_bwd_kernel(const float* __restrict__ dfbm, const float* __restrict__ Ab, const float* __restrict__ hbar,
                                const int* __restrict__ cells, const int* __restrict__ row_ptr,
                                int L, int D, float* __restrict__ dAb, float* __restrict__ dhbar)
{
    const int i = blockIdx.x, b = blockIdx.y;
    const int r0 = row_ptr[b * L + i], r1 = row_ptr[b * L + i + 1];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float* arow = Ab + ((size_t)b * L + i) * L;
    float* darow = dAb + ((size_t)b * L + i) * L;
    for (int n = r0 + wave; n < r1; n += 4) {
        const int j = cells[4 * (size_t)n + 2];
        const float a = arow[j];
        float dot = 0.f;
        for (int d = lane * 4; d < D; d += 256) {
            const float4 dy = ldg4(dfbm + ((size_t)b * L + i) * D + d);
            const float4 x = ldg4(hbar + (size_t)n * D + d);
            dot = fmaf(dy.x, x.x, dot); dot = fmaf(dy.y, x.y, dot); dot = fmaf(dy.z, x.z, dot); dot = fmaf(dy.w, x.w, dot);
            stg4(dhbar + (size_t)n * D + d, f4scale(dy, a));
        }
        dot = wave_sum(dot);
        if (lane == 0) darow[j] = dot;
    }
}

}  // namespace smin

using namespace smin;

extern "C" int smin_boundary_reduce_fwd(void* stream, const float* Ab, const float* hbar, const int32_t* cells,
                                        const int32_t* row_ptr, int N, int B, int L, int D, float* fbm)
{
    (void)N;
    SMIN_REQUIRE(D % 4 == 0);
    hipLaunchKernelGGL(boundary_reduce_fwd_kernel, dim3(L, B), dim3(128), 0, (hipStream_t)stream, Ab, hbar, cells, row_ptr, L, D, fbm);
    SMIN_LAUNCH_CHECK();
    return 0;
}

extern "C" int smin_boundary_reduce_bwd(void* stream, const float* dfbm, const float* Ab, const float* hbar,
                                        const int32_t* cells, const int32_t* row_ptr, int N, int B, int L, int D,
                                        float* dAb, float* dhbar)
{
    (void)N;
    hipStream_t st = (hipStream_t)stream;
    SMIN_REQUIRE(D % 4 == 0);
    hipError_t e = hipMemsetAsync(dAb, 0, sizeof(float) * (size_t)B * L * L, st);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(boundary_reduce_bwd_kernel, dim3(L, B), dim3(256), 0, st, dfbm, Ab, hbar, cells, row_ptr, L, D, dAb, dhbar);
    SMIN_LAUNCH_CHECK();
    return 0;
}
