// R@n, IoU=m metric of the reference (utils.py:10-31) on the device:
//   score = pm * sqrt(ps[i]) * sqrt(pe[j]) * moment_mask ; top-5 moments per sample ; hit if any of the top-n has IoU > m.
// One workgroup per sample: five rounds of block arg-max (ties -> lowest flat index, like a stable top-k would
// not guarantee either), then the 2 x 4 hit flags; a second pass sums the samples.  One host read per call.
#include "common.h"
#include "smin_hip.h"

namespace smin {

__global__ __launch_bounds__(256)
void ious_kernel(const float* __restrict__ pm, const float* __restrict__ ps, const float* __restrict__ pe, const uint8_t* __restrict__ mm,
                 const float* __restrict__ sm, int L, float* __restrict__ hits /* [B][8] */)
{
    extern __shared__ float sc[];                       // [L*L] scores
    __shared__ float rv[4];
    __shared__ int ri[4];
    __shared__ float top[5];
    const int b = blockIdx.x, t = threadIdx.x, n = L * L;
    for (int k = t; k < n; k += 256) {
        const int i = k / L, j = k % L;
        const size_t o = (size_t)b * n + k;
        sc[k] = pm[o] * sqrtf(ps[(size_t)b * L + i]) * sqrtf(pe[(size_t)b * L + j]) * (mm[o] ? 1.f : 0.f);
    }
    __syncthreads();
    for (int r = 0; r < 5; ++r) {
        float bv = -INFINITY; int bi = 0x7fffffff;
        for (int k = t; k < n; k += 256) { const float v = sc[k]; if (v > bv || (v == bv && k < bi)) { bv = v; bi = k; } }
        for (int o = 32; o >= 1; o >>= 1) {
            const float ov = __shfl_xor(bv, o); const int oi = __shfl_xor(bi, o);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if ((t & 63) == 0) { rv[t >> 6] = bv; ri[t >> 6] = bi; }
        __syncthreads();
        if (t == 0) {
            for (int w = 1; w < 4; ++w) if (rv[w] > bv || (rv[w] == bv && ri[w] < bi)) { bv = rv[w]; bi = ri[w]; }
            top[r] = (bi < n) ? sm[(size_t)b * n + bi] : 0.f;
            if (bi < n) sc[bi] = -INFINITY;
        }
        __syncthreads();
    }
    if (t < 8) {
        const int nn = t < 4 ? 1 : 5;
        const float thr = (t & 3) == 0 ? 0.1f : (t & 3) == 1 ? 0.3f : (t & 3) == 2 ? 0.5f : 0.7f;
        bool hit = false;
        for (int r = 0; r < nn; ++r) hit = hit || top[r] > thr;
        hits[(size_t)b * 8 + t] = hit ? 1.f : 0.f;
    }
}

__global__ void ious_sum_kernel(const float* __restrict__ hits, int B, float* __restrict__ out)
{
    const int t = threadIdx.x;
    if (t >= 8) return;
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += hits[(size_t)b * 8 + t];
    out[t] = s;
}

}  // namespace smin

using namespace smin;

extern "C" int smin_compute_ious(void* stream, const float* pm, const float* ps, const float* pe, const uint8_t* mm, const float* sm,
                                 int B, int L, float* counts /* [8]: R@1 x {0.1,0.3,0.5,0.7}, R@5 x {...} */, float* ws /* [B][8] */)
{
    hipStream_t st = (hipStream_t)stream;
    SMIN_REQUIRE(L * L >= 5 && (size_t)L * L * sizeof(float) <= 150 * 1024);
    const size_t lds = sizeof(float) * (size_t)L * L;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(ious_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(ious_kernel, dim3(B), dim3(256), lds, st, pm, ps, pe, mm, sm, L, ws);
    SMIN_LAUNCH_CHECK();
    hipLaunchKernelGGL(ious_sum_kernel, dim3(1), dim3(64), 0, st, ws, B, counts);
    SMIN_LAUNCH_CHECK();
    return 0;
}
