// Localization (reference models.py:324-344): the 2D start/end proposal score map
//   pm[b,i,j] = sigmoid(<w_pm, f_m[n]> + b_pm) * m          ps/pe/pa[b,l] = sigmoid(<w_k, f_b[b,l]> + b_k) * lmask
// HBM-bound: reads f_m once, writes one float per cell.
#include "common.h"
#include "gemm.h"
#include "smin_hip.h"

namespace smin {

// one wave per cell
__global__ __launch_bounds__(256)
void score_map_fwd_kernel(const float* __restrict__ fm, const int* __restrict__ cells, int N, int L, int D,
                          const float* __restrict__ wm, const float* __restrict__ bm, float* __restrict__ pm)
{
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const int lane = threadIdx.x & 63;
    float dot = 0.f;
    for (int d = lane * 4; d < D; d += 256) {
        const float4 x = ldg4(fm + (size_t)n * D + d), w = ldg4(wm + d);
        dot = fmaf(x.x, w.x, dot); dot = fmaf(x.y, w.y, dot); dot = fmaf(x.z, w.z, dot); dot = fmaf(x.w, w.w, dot);
    }
    dot = wave_sum(dot);
    if (lane == 0) {
        const Cell c = load_cell(cells, n);
        pm[((size_t)c.b * L + c.i) * L + c.j] = (1.0f / (1.0f + expf(-(dot + bm[0])))) * (float)c.m;
    }
}

// one wave per (b, l): the three boundary heads
__global__ __launch_bounds__(256)
// (also clears row (b, l) of the dense score map pm [B][L][L]: the cell kernel that follows writes the valid cells only)
void score_heads_fwd_kernel(const float* __restrict__ fb, int BL, int D, const float* __restrict__ wb, const float* __restrict__ bb,
                            const float* __restrict__ lmask, float* __restrict__ psea, float* __restrict__ pm, int L)
{
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= BL) return;
    const int lane = threadIdx.x & 63;
    for (int j = lane; j < L; j += 64) pm[(size_t)r * L + j] = 0.f;
    float d0 = 0.f, d1 = 0.f, d2 = 0.f;
    for (int d = lane; d < D; d += 64) {
        const float x = fb[(size_t)r * D + d];
        d0 = fmaf(x, wb[d], d0); d1 = fmaf(x, wb[D + d], d1); d2 = fmaf(x, wb[2 * D + d], d2);
    }
    d0 = wave_sum(d0); d1 = wave_sum(d1); d2 = wave_sum(d2);
    if (lane == 0) {
        const float lm = lmask[r];
        psea[r] = lm / (1.0f + expf(-(d0 + bb[0])));
        psea[(size_t)BL + r] = lm / (1.0f + expf(-(d1 + bb[1])));
        psea[(size_t)2 * BL + r] = lm / (1.0f + expf(-(d2 + bb[2])));
    }
}

// chunk of 64 cells per workgroup (a thread walks its chunk serially, one dependent load chain per cell: short chunks
// and many workgroups keep it HBM-bound): dfm = dz * wm ; partial[chunk][d] = sum dz * fm ; partial bias
__global__ __launch_bounds__(128)
void score_map_bwd_kernel(const float* __restrict__ dpm, const float* __restrict__ pm, const float* __restrict__ fm,
                          const int* __restrict__ cells, int N, int L, int D, const float* __restrict__ wm,
                          float* __restrict__ dfm, float* __restrict__ partial, float* __restrict__ bpartial)
{
    const int chunk = blockIdx.x;
    const int n0 = chunk * 64, n1 = min(N, n0 + 64);
    float bsum = 0.f;
    for (int d = threadIdx.x * 4; d < D || d == threadIdx.x * 4; d += 512) {
        const bool dok = d < D;
        const float4 w4 = dok ? ldg4(wm + d) : f4zero();
        float4 acc = f4zero();
        bsum = 0.f;
        for (int n = n0; n < n1; ++n) {
            const Cell c = load_cell(cells, n);
            const size_t o = ((size_t)c.b * L + c.i) * L + c.j;
            const float p = pm[o];
            const float dz = dpm[o] * (float)c.m * p * (1.0f - p);
            bsum += dz;
            if (dok) {
                stg4(dfm + (size_t)n * D + d, f4scale(w4, dz));
                acc = f4fma(ldg4(fm + (size_t)n * D + d), dz, acc);
            }
        }
        if (dok) stg4(partial + (size_t)chunk * D + d, acc);
    }
    if (threadIdx.x == 0) bpartial[chunk] = bsum;
}

// chunk of HEAD_ROWS (b,l) rows per workgroup for the three heads (a thread walks its chunk serially, a dependent chain of scalar
// loads per row: with 64 rows per chunk the 64 workgroups of the ActivityNet shape took 64 us at the opening of the backward pass)
constexpr int HEAD_ROWS = 8;
__global__ __launch_bounds__(128)
void score_heads_bwd_kernel(const float* __restrict__ dpsea, const float* __restrict__ psea, const float* __restrict__ fb,
                            int BL, int D, const float* __restrict__ wb, const float* __restrict__ lmask,
                            float* __restrict__ dfb, float* __restrict__ partial, float* __restrict__ bpartial)
{
    const int chunk = blockIdx.x;
    const int r0 = chunk * HEAD_ROWS, r1 = min(BL, r0 + HEAD_ROWS);
    float bs0 = 0.f, bs1 = 0.f, bs2 = 0.f;
    for (int d = threadIdx.x * 4; d < D || d == threadIdx.x * 4; d += 512) {
        const bool dok = d < D;
        const float4 w0 = dok ? ldg4(wb + d) : f4zero(), w1 = dok ? ldg4(wb + D + d) : f4zero(), w2 = dok ? ldg4(wb + 2 * D + d) : f4zero();
        float4 a0 = f4zero(), a1 = f4zero(), a2 = f4zero();
        bs0 = bs1 = bs2 = 0.f;
        for (int r = r0; r < r1; ++r) {
            const float lm = lmask[r];
            // psea already carries the length mask; the un-masked sigmoid equals it wherever lm == 1
            const float p0 = psea[r], p1 = psea[(size_t)BL + r], p2 = psea[(size_t)2 * BL + r];
            const float z0 = dpsea[r] * lm * p0 * (1.0f - p0);
            const float z1 = dpsea[(size_t)BL + r] * lm * p1 * (1.0f - p1);
            const float z2 = dpsea[(size_t)2 * BL + r] * lm * p2 * (1.0f - p2);
            bs0 += z0; bs1 += z1; bs2 += z2;
            if (dok) {
                const float4 x = ldg4(fb + (size_t)r * D + d);
                stg4(dfb + (size_t)r * D + d, f4fma(w0, z0, f4fma(w1, z1, f4scale(w2, z2))));
                a0 = f4fma(x, z0, a0); a1 = f4fma(x, z1, a1); a2 = f4fma(x, z2, a2);
            }
        }
        if (dok) {
            stg4(partial + ((size_t)chunk * 3 + 0) * D + d, a0);
            stg4(partial + ((size_t)chunk * 3 + 1) * D + d, a1);
            stg4(partial + ((size_t)chunk * 3 + 2) * D + d, a2);
        }
    }
    if (threadIdx.x == 0) { bpartial[chunk * 3] = bs0; bpartial[chunk * 3 + 1] = bs1; bpartial[chunk * 3 + 2] = bs2; }
}

}  // namespace smin

using namespace smin;

extern "C" int smin_score_map_fwd(void* stream, const float* fm, const float* fb, const int32_t* cells, int N, int B, int L, int D,
                                  const float* wm, const float* bm, const float* wb, const float* bb, const float* lmask,
                                  float* pm, float* psea)
{
    hipStream_t st = (hipStream_t)stream;
    SMIN_REQUIRE(D % 4 == 0);
    hipLaunchKernelGGL(score_heads_fwd_kernel, dim3(cdiv(B * L, 4)), dim3(256), 0, st, fb, B * L, D, wb, bb, lmask, psea, pm, L);
    SMIN_LAUNCH_CHECK();
    if (N > 0) {
        hipLaunchKernelGGL(score_map_fwd_kernel, dim3(cdiv(N, 4)), dim3(256), 0, st, fm, cells, N, L, D, wm, bm, pm);
        SMIN_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int smin_score_map_bwd(void* stream, const float* dpm, const float* dpsea, const float* pm, const float* psea,
                                  const float* fm, const float* fb, const int32_t* cells, int N, int B, int L, int D,
                                  const float* wm, const float* wb, const float* lmask,
                                  float* dfm, float* dfb, float* dwm, float* dbm, float* dwb, float* dbb, void* ws, size_t ws_bytes)
{
    hipStream_t st = (hipStream_t)stream;
    SMIN_REQUIRE(D % 4 == 0);
    const int BL = B * L;
    const int nch = cdiv(N > 0 ? N : 1, 64), hch = cdiv(BL, HEAD_ROWS);
    float* w = reinterpret_cast<float*>(ws);
    float* part = w;                                   // [nch][D]
    float* bpart = part + (size_t)nch * D;             // [nch]  (padded to 4)
    float* hpart = bpart + ((nch + 3) & ~3);           // [hch][3][D]
    float* hbpart = hpart + (size_t)hch * 3 * D;       // [hch][3]
    SMIN_REQUIRE((size_t)((hbpart + (size_t)hch * 3) - w) * sizeof(float) <= ws_bytes);
    // two independent halves (the map's score: dpm -> dfm, dwm, dbm; the boundary heads: dpsea -> dfb, dwb, dbb), each with its own part
    // of ws: dpm == NULL or dpsea == NULL skips a half, so that a host can issue them on two streams
    SMIN_REQUIRE(dpm != nullptr || dpsea != nullptr);
    if (dpm) {
        SMIN_REQUIRE(dwm != nullptr && dbm != nullptr && (dfm != nullptr || N == 0));
        if (N > 0) {
            hipLaunchKernelGGL(score_map_bwd_kernel, dim3(nch), dim3(128), 0, st, dpm, pm, fm, cells, N, L, D, wm, dfm, part, bpart);
            SMIN_LAUNCH_CHECK();
            int rc = launch_reduce_slabs2(st, part, dwm, D, bpart, dbm, 1, nch); if (rc) return rc;
        } else {
            (void)hipMemsetAsync(dwm, 0, sizeof(float) * D, st);
            (void)hipMemsetAsync(dbm, 0, sizeof(float), st);
        }
    }
    if (dpsea) {
        SMIN_REQUIRE(dfb != nullptr && dwb != nullptr && dbb != nullptr);
        hipLaunchKernelGGL(score_heads_bwd_kernel, dim3(hch), dim3(128), 0, st, dpsea, psea, fb, BL, D, wb, lmask, dfb, hpart, hbpart);
        SMIN_LAUNCH_CHECK();
        int rc = launch_reduce_slabs2(st, hpart, dwb, 3 * D, hbpart, dbb, 3, hch); if (rc) return rc;
    }
    return 0;
}
