// Masks and training targets of a batch on the device (reference dataset.py:95-155, per sample on the host there; SURVEY.md
// 8f-4): from (start, end, duration, sampled frames, query length) of every sample one launch writes
//   video_mask [B][T], query_mask [B][Nq], length_mask [B][L], moment_mask [B][L][L]                  (dataset.py:141-149, 173)
//   sm [B][L][L] = IoU of the window (snippet i .. snippet j) with the ground truth, ym = sm > 0.5      (dataset.py:95-110, 151)
//   ss, se [B][L] = unnormalised Gaussians around the true boundaries, sigma = (te - ts) / 5; ys, ye    (dataset.py:112-120, 154-155)
//   ya [B][L] = snippet fully inside the ground-truth window                                            (dataset.py:122-126)
// so that a data-parallel job ships features and five scalars per sample instead of eleven tensors.  fp32 arithmetic in the
// reference's order of operations (s_i = i * dur / L, e_j = (j + 1) * dur / L).
#include "common.h"
#include "smin_hip.h"

namespace smin {

__global__ __launch_bounds__(256)
void build_targets_kernel(const float* __restrict__ times, const float* __restrict__ duration, const int* __restrict__ nfeats, const int* __restrict__ qlen,
                          int T, int L, int Nq, uint8_t* __restrict__ video_mask, uint8_t* __restrict__ query_mask, uint8_t* __restrict__ length_mask,
                          uint8_t* __restrict__ moment_mask, float* __restrict__ sm, uint8_t* __restrict__ ym, float* __restrict__ ss, uint8_t* __restrict__ ys,
                          float* __restrict__ se, uint8_t* __restrict__ ye, uint8_t* __restrict__ ya, const float* __restrict__ two_sigma_sq)
{
    const int b = blockIdx.y;
    const float ts = times[2 * b], te = times[2 * b + 1], dur = duration[b], Lf = (float)L;
    const int nf = min(nfeats[b], T);
    const int n_len = (int)ceil((double)nf / ((double)T / (double)L));      // dataset.py:145: ceil(nfeats / (T / L))
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < L * L) {
        const int i = k / L, j = k - i * L;
        const float s_i = (float)i * dur / Lf, e_j = ((float)j + 1.0f) * dur / Lf;
        const float inter = fmaxf(0.0f, fminf(e_j, te) - fmaxf(s_i, ts));
        const float uni = fmaxf(0.0f, fmaxf(e_j, te) - fminf(s_i, ts));
        const float iou = inter / uni;                                          // 0 / 0 -> NaN exactly where the reference has it
        const size_t o = (size_t)b * L * L + k;
        sm[o] = iou;
        ym[o] = iou > 0.5f;
        moment_mask[o] = (j >= i) && i < n_len && j < n_len;
    }
    if (k < L) {
        const float s_t = (float)k * dur / Lf, e_t = ((float)k + 1.0f) * dur / Lf;
        // dataset.py:116-119 forms sigma = (te - ts) / 5 and 2 sigma^2 in Python doubles from the annotation times and rounds once, at the
        // tensor division: a caller that still has those doubles hands the rounded denominator in (two_sigma_sq); otherwise it is
        // formed in double from the fp32 times (one rounding of each time instead of a chain of fp32 roundings)
        const double sigma = ((double)te - (double)ts) / 5.0;
        const float den = two_sigma_sq ? two_sigma_sq[b] : (float)(2.0 * (sigma * sigma));
        const float a = s_t - ts, c = e_t - te;
        const float vs = expf(-(a * a) / den), ve = expf(-(c * c) / den);
        const size_t o = (size_t)b * L + k;
        ss[o] = vs; ys[o] = vs > 0.5f;
        se[o] = ve; ye[o] = ve > 0.5f;
        ya[o] = (s_t >= ts) && (e_t <= te);
        length_mask[o] = k < n_len;
    }
    for (int t = k; t < T; t += gridDim.x * blockDim.x) video_mask[(size_t)b * T + t] = t < nf;
    if (query_mask) for (int w = k; w < Nq; w += gridDim.x * blockDim.x) query_mask[(size_t)b * Nq + w] = w < qlen[b];
}

}  // namespace smin

extern "C" int smin_build_targets(void* stream, const float* times, const float* duration, const int32_t* nfeats, const int32_t* qlen, int B, int T, int L, int Nq,
                                  uint8_t* video_mask, uint8_t* query_mask, uint8_t* length_mask, uint8_t* moment_mask, float* sm, uint8_t* ym,
                                  float* ss, uint8_t* ys, float* se, uint8_t* ye, uint8_t* ya, const float* two_sigma_sq)
{
    SMIN_REQUIRE(B >= 0 && T >= 1 && L >= 1 && T % L == 0 && (query_mask == nullptr || (qlen != nullptr && Nq >= 1)));
    if (B == 0) return 0;
    hipLaunchKernelGGL(smin::build_targets_kernel, dim3(cdiv(L * L, 256), B), dim3(256), 0, (hipStream_t)stream, times, duration, nfeats, qlen, T, L, Nq,
                       video_mask, query_mask, length_mask, moment_mask, sm, ym, ss, ys, se, ye, ya, two_sigma_sq);
    SMIN_LAUNCH_CHECK();
    return 0;
}
