// Restated training loss of the reference's train loop (main.py:89-116) as two device kernels:
//   L = L_m + L_s + L_e + 0.5 L_a ;  each term = mean over samples of ( sum_masked elem / sum mask )
//   scaled BCE elem (main.py:92-95): BCE(p, y) * s*y + BCE(1-p, 1-y) * (1-s)*(1-y), logs clamped at -100 like torch
//   L_a: plain BCE.  (The reference passes reduction=None, which raises; reduction='none' is the evident intent.)
// One workgroup per sample reduces its map and boundary rows in a fixed order; a second tiny pass sums the samples,
// so the scalar is bitwise reproducible.  Backward uses torch's BCELoss gradient w*(p-y)/max(p(1-p), 1e-12).
#include "common.h"
#include "smin_hip.h"

namespace smin {

__device__ __forceinline__ float bce_elem(float p, float y) {
    const float lp = fmaxf(logf(p), -100.f), l1p = fmaxf(logf(1.f - p), -100.f);
    return -(y * lp + (1.f - y) * l1p);
}
__device__ __forceinline__ float scaled_elem(float p, float y, float s) {
    return (s * y) * bce_elem(p, y) + ((1.f - s) * (1.f - y)) * bce_elem(1.f - p, 1.f - y);
}
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// part[b] = { sum_m, cnt_m, sum_s, sum_e, sum_a, cnt_l }
__global__ __launch_bounds__(256)
void loss_fwd_kernel(const float* __restrict__ pm, const uint8_t* __restrict__ ym, const float* __restrict__ sm, const uint8_t* __restrict__ mm,
                     const float* __restrict__ ps, const uint8_t* __restrict__ ys, const float* __restrict__ ss,
                     const float* __restrict__ pe, const uint8_t* __restrict__ ye, const float* __restrict__ se,
                     const float* __restrict__ pa, const uint8_t* __restrict__ ya, const uint8_t* __restrict__ lm,
                     int L, float* __restrict__ part)
{
    __shared__ float red[4];
    const int b = blockIdx.x, t = threadIdx.x;
    float sumM = 0.f, cntM = 0.f;
    for (int k = t; k < L * L; k += 256) {
        const size_t o = (size_t)b * L * L + k;
        const float m = mm[o] ? 1.f : 0.f;
        sumM += scaled_elem(pm[o], ym[o] ? 1.f : 0.f, sm[o]) * m;
        cntM += m;
    }
    float sS = 0.f, sE = 0.f, sA = 0.f, cL = 0.f;
    for (int k = t; k < L; k += 256) {
        const size_t o = (size_t)b * L + k;
        const float m = lm[o] ? 1.f : 0.f;
        sS += scaled_elem(ps[o], ys[o] ? 1.f : 0.f, ss[o]) * m;
        sE += scaled_elem(pe[o], ye[o] ? 1.f : 0.f, se[o]) * m;
        sA += bce_elem(pa[o], ya[o] ? 1.f : 0.f) * m;
        cL += m;
    }
    sumM = block_sum(sumM, red); cntM = block_sum(cntM, red);
    sS = block_sum(sS, red); sE = block_sum(sE, red); sA = block_sum(sA, red); cL = block_sum(cL, red);
    if (t == 0) {
        float* p = part + (size_t)b * 6;
        p[0] = sumM; p[1] = cntM; p[2] = sS; p[3] = sE; p[4] = sA; p[5] = cL;
    }
}

__global__ void loss_final_kernel(const float* __restrict__ part, int B, float* __restrict__ loss)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float lmv = 0.f, ls = 0.f, le = 0.f, la = 0.f;
    for (int b = 0; b < B; ++b) {
        const float* p = part + (size_t)b * 6;
        lmv += p[0] / p[1]; ls += p[2] / p[5]; le += p[3] / p[5]; la += p[4] / p[5];
    }
    const float inv = 1.0f / (float)B;
    loss[0] = lmv * inv + ls * inv + le * inv + 0.5f * (la * inv);
}

__device__ __forceinline__ float bce_grad(float p, float y, float w) {
    return w * (p - y) / fmaxf(p * (1.f - p), 1e-12f);
}

__global__ __launch_bounds__(256)
void loss_bwd_kernel(const float* __restrict__ dloss, const float* __restrict__ part,
                     const float* __restrict__ pm, const uint8_t* __restrict__ ym, const float* __restrict__ sm, const uint8_t* __restrict__ mm,
                     const float* __restrict__ ps, const uint8_t* __restrict__ ys, const float* __restrict__ ss,
                     const float* __restrict__ pe, const uint8_t* __restrict__ ye, const float* __restrict__ se,
                     const float* __restrict__ pa, const uint8_t* __restrict__ ya, const uint8_t* __restrict__ lm,
                     int B, int L, float* __restrict__ dpm, float* __restrict__ dps, float* __restrict__ dpe, float* __restrict__ dpa)
{
    const int b = blockIdx.x, t = threadIdx.x;
    const float g = dloss[0] / (float)B;
    const float gm = g / part[(size_t)b * 6 + 1], gl = g / part[(size_t)b * 6 + 5];
    for (int k = t; k < L * L; k += 256) {
        const size_t o = (size_t)b * L * L + k;
        const float y = ym[o] ? 1.f : 0.f, s = sm[o];
        dpm[o] = mm[o] ? gm * bce_grad(pm[o], y, s * y + (1.f - s) * (1.f - y)) : 0.f;
    }
    for (int k = t; k < L; k += 256) {
        const size_t o = (size_t)b * L + k;
        const bool m = lm[o] != 0;
        const float y1 = ys[o] ? 1.f : 0.f, s1 = ss[o], y2 = ye[o] ? 1.f : 0.f, s2 = se[o], y3 = ya[o] ? 1.f : 0.f;
        dps[o] = m ? gl * bce_grad(ps[o], y1, s1 * y1 + (1.f - s1) * (1.f - y1)) : 0.f;
        dpe[o] = m ? gl * bce_grad(pe[o], y2, s2 * y2 + (1.f - s2) * (1.f - y2)) : 0.f;
        dpa[o] = m ? 0.5f * gl * bce_grad(pa[o], y3, 1.f) : 0.f;
    }
}

}  // namespace smin

using namespace smin;

extern "C" int smin_loss_fwd(void* stream, const float* pm, const uint8_t* ym, const float* sm, const uint8_t* mm,
                             const float* ps, const uint8_t* ys, const float* ss, const float* pe, const uint8_t* ye, const float* se,
                             const float* pa, const uint8_t* ya, const uint8_t* lm, int B, int L, float* loss, float* part)
{
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(loss_fwd_kernel, dim3(B), dim3(256), 0, st, pm, ym, sm, mm, ps, ys, ss, pe, ye, se, pa, ya, lm, L, part);
    SMIN_LAUNCH_CHECK();
    hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(64), 0, st, part, B, loss);
    SMIN_LAUNCH_CHECK();
    return 0;
}

extern "C" int smin_loss_bwd(void* stream, const float* dloss, const float* part,
                             const float* pm, const uint8_t* ym, const float* sm, const uint8_t* mm,
                             const float* ps, const uint8_t* ys, const float* ss, const float* pe, const uint8_t* ye, const float* se,
                             const float* pa, const uint8_t* ya, const uint8_t* lm, int B, int L,
                             float* dpm, float* dps, float* dpe, float* dpa)
{
    hipLaunchKernelGGL(loss_bwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, dloss, part, pm, ym, sm, mm, ps, ys, ss, pe, ye, se, pa, ya, lm,
                       B, L, dpm, dps, dpe, dpa);
    SMIN_LAUNCH_CHECK();
    return 0;
}
