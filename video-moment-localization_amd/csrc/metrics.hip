// R@n, IoU=m metric of the reference (utils.py:10-31) on the device:
//   score = pm * sqrt(ps[i]) * sqrt(pe[j]) * moment_mask ; top-5 moments per sample ; hit if any of the top-n has IoU > m.
// One workgroup per sample: per-thread running top-5 over the streamed scores, then five rounds of block arg-max over the
// candidates (ties -> lowest flat index; torch.topk leaves tie order unspecified, so ties are "parity unpinned"), then the
// 2 x 4 hit flags; a second pass sums the samples.  One host read per call.  No limit on L.
#include "common.h"
#include "smin_hip.h"

namespace smin {

// candidate order of the metric: higher score first, ties -> lower flat index
__device__ __forceinline__ bool better(float v, int k, float bv, int bi) { return v > bv || (v == bv && k < bi); }

__global__ __launch_bounds__(256)
void ious_kernel(const float* __restrict__ pm, const float* __restrict__ ps, const float* __restrict__ pe, const uint8_t* __restrict__ mm,
                 const float* __restrict__ sm, int L, float* __restrict__ hits /* [B][8] */)
{
    // Any L: each thread streams its share of the L*L scores keeping its own five best in registers (no score buffer, so
    // the 512 x 512 long-video map costs the same LDS as a 16 x 16 one); the 256 x 5 candidates then go through five rounds
    // of workgroup arg-max.
    __shared__ float cv[256 * 5];
    __shared__ int ci[256 * 5];
    __shared__ float rv[4];
    __shared__ int ri[4], rs[4];
    __shared__ float top[5];
    const int b = blockIdx.x, t = threadIdx.x, n = L * L;
    float v5[5]; int i5[5];
#pragma unroll
    for (int r = 0; r < 5; ++r) { v5[r] = -INFINITY; i5[r] = 0x7fffffff; }
    const float* psb = ps + (size_t)b * L;
    const float* peb = pe + (size_t)b * L;
    for (int k = t; k < n; k += 256) {
        const int i = k / L, j = k - i * L;
        const size_t o = (size_t)b * n + k;
        float v = pm[o] * sqrtf(psb[i]) * sqrtf(peb[j]) * (mm[o] ? 1.f : 0.f);
        int kk = k;
        if (better(v, kk, v5[4], i5[4])) {
#pragma unroll
            for (int r = 0; r < 5; ++r) {               // insertion into the sorted five
                if (better(v, kk, v5[r], i5[r])) { const float tv = v5[r]; const int ti = i5[r]; v5[r] = v; i5[r] = kk; v = tv; kk = ti; }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 5; ++r) { cv[t * 5 + r] = v5[r]; ci[t * 5 + r] = i5[r]; }
    __syncthreads();
    for (int r = 0; r < 5; ++r) {
        float bv = -INFINITY; int bi = 0x7fffffff, bs = 0;
        for (int s = t; s < 256 * 5; s += 256) { if (better(cv[s], ci[s], bv, bi)) { bv = cv[s]; bi = ci[s]; bs = s; } }
        for (int o = 32; o >= 1; o >>= 1) {
            const float ov = __shfl_xor(bv, o); const int oi = __shfl_xor(bi, o); const int os = __shfl_xor(bs, o);
            if (better(ov, oi, bv, bi)) { bv = ov; bi = oi; bs = os; }
        }
        if ((t & 63) == 0) { rv[t >> 6] = bv; ri[t >> 6] = bi; rs[t >> 6] = bs; }
        __syncthreads();
        if (t == 0) {
            for (int w = 1; w < 4; ++w) if (better(rv[w], ri[w], bv, bi)) { bv = rv[w]; bi = ri[w]; bs = rs[w]; }
            top[r] = (bi < n) ? sm[(size_t)b * n + bi] : 0.f;
            if (bi < n) { cv[bs] = -INFINITY; ci[bs] = 0x7fffffff; }
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
    SMIN_REQUIRE(B >= 1 && L >= 1 && (size_t)L * L >= 5 && (size_t)L * L < 0x7fffffff);
    hipLaunchKernelGGL(ious_kernel, dim3(B), dim3(256), 0, st, pm, ps, pe, mm, sm, L, ws);
    SMIN_LAUNCH_CHECK();
    hipLaunchKernelGGL(ious_sum_kernel, dim3(1), dim3(64), 0, st, ws, B, counts);
    SMIN_LAUNCH_CHECK();
    return 0;
}
