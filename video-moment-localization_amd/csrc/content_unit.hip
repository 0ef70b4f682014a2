// ContentUnit (reference models.py:228-276) + ContentAttention (models.py:198-226), forward and backward.
//
// Per packed cell n with clips c < C (rows r = n*C + c):
//   chat  = (fc Wch^T + bch) * m                                   GEMM  (MFMA, gemm_nt)
//   S     = (chat Mq_b^T + uq_b) / sqrt(dl), masked over words     "QK^T" with W_q/W_k folded per sample
//   P     = softmax_w(S) ; a = P what_b ; q = chat * (a + shat_b)
//   A     = softmax_c(q q^T / sqrt(dl)) ; cchat = A chat           attention core: one wave per cell
//   out   = (cchat Wc^T + bc) * m + fc + sigmoid(fm*fs)*fm         GEMM  (MFMA, gemm_nt) + fused epilogue
// Backward recomputes the attention core from the saved chat and reduces the per-sample word-side
// gradients (dMq, dwhat, dshat, duq) through fixed-order partial slabs (deterministic, no float atomics).
#include "content_attn.h"
#include "smin_hip.h"

namespace smin {

// ------------------------------------------------------------------ epilogues (see gemm.h: tile protocol)
struct EpBiasMask {                 // chat[row][col] = (acc + bias[col]) * m[row / C]
    const float* bias; const int* cells; float* out; int C;
    __device__ __forceinline__ void chunk(const float* Ws, int row0, int col0, int ncols, int M, int N, int lane) const {
        chunk_rows_f4(Ws, row0, col0, ncols, M, N, lane, [&](int row, int col, float4 v) {
            const float m = (float)cells[4 * (size_t)(row / C) + 3];
            stg4(out + (size_t)row * N + col, f4scale(f4add(v, ldg4(bias + col)), m));
        });
    }
};

// out = (acc + bc) * m + fc + hbar[n]      (models.py:269-276; hbar = sigmoid(fm*fs)*fm from gate.hip)
struct EpContentOut {               // any C
    const float* bc; const int* cells; const float* fc; const float* hbar; float* out; int C;
    __device__ __forceinline__ void chunk(const float* Ws, int row0, int col0, int ncols, int M, int N, int lane) const {
        chunk_rows_f4(Ws, row0, col0, ncols, M, N, lane, [&](int row, int col, float4 v) {
            const int n = row / C;
            const float m = (float)cells[4 * (size_t)n + 3];
            const float4 o = f4add(f4add(f4scale(f4add(v, ldg4(bc + col)), m), ldg4(fc + (size_t)row * N + col)), ldg4(hbar + (size_t)n * N + col));
            stg4(out + (size_t)row * N + col, o);
        });
    }
};
struct EpContentOut4 {              // C == 4: one lane owns the 4 clips of a cell -> also emits fcmean = mean_c out
    const float* bc; const int* cells; const float* fc; const float* hbar; float* out; float* fcmean;
    __device__ __forceinline__ void chunk(const float* Ws, int row0, int col0, int ncols, int M, int N, int lane) const {
        chunk_quads_f4(Ws, row0, col0, ncols, M, N, lane, [&](int row, int col, const float4 (&v)[4]) {
            const int n = row >> 2;
            const float m = (float)cells[4 * (size_t)n + 3];
            const float4 b4 = ldg4(bc + col), hb = ldg4(hbar + (size_t)n * N + col);
            float4 sum = f4zero();
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const size_t o = ((size_t)row + c) * N + col;
                const float4 r = f4add(f4add(f4scale(f4add(v[c], b4), m), ldg4(fc + o)), hb);
                stg4(out + o, r);
                sum = f4add(sum, r);
            }
            stg4(fcmean + (size_t)n * N + col, f4scale(sum, 0.25f));
        });
    }
};

// last layer: only mean_c of the output is consumed ->  fcmean = m * (cmean Wc^T + bc) + mean_c fc + hbar   (rows = cells)
struct EpLastMean {
    const float* bc; const int* cells; const float* fcmean_in; const float* hbar; float* fcmean;
    __device__ __forceinline__ void chunk(const float* Ws, int row0, int col0, int ncols, int M, int N, int lane) const {
        chunk_rows_f4(Ws, row0, col0, ncols, M, N, lane, [&](int row, int col, float4 v) {
            const float m = (float)cells[4 * (size_t)row + 3];
            const size_t o = (size_t)row * N + col;
            stg4(fcmean + o, f4add(f4add(f4scale(f4add(v, ldg4(bc + col)), m), ldg4(fcmean_in + o)), ldg4(hbar + o)));
        });
    }
};
struct EpScale {                    // out = acc * s
    float* out; float s;
    __device__ __forceinline__ void chunk(const float* Ws, int row0, int col0, int ncols, int M, int N, int lane) const {
        chunk_rows_f4(Ws, row0, col0, ncols, M, N, lane, [&](int row, int col, float4 v) { stg4(out + (size_t)row * N + col, f4scale(v, s)); });
    }
};

struct EpPlain {
    float* out;
    __device__ __forceinline__ void chunk(const float* Ws, int row0, int col0, int ncols, int M, int N, int lane) const {
        chunk_rows_f4(Ws, row0, col0, ncols, M, N, lane, [&](int row, int col, float4 v) { stg4(out + (size_t)row * N + col, v); });
    }
};

// dfc = acc + dfc_out + dfcmean / C   (residual path, un-masked)
template <bool HAS_DFC>
struct EpAddDout {                  // any C
    const float* dfc_out; const float* dmean; float* out; int C; float invC;
    __device__ __forceinline__ void chunk(const float* Ws, int row0, int col0, int ncols, int M, int N, int lane) const {
        chunk_rows_f4(Ws, row0, col0, ncols, M, N, lane, [&](int row, int col, float4 v) {
            float4 r = f4fma(ldg4(dmean + (size_t)(row / C) * N + col), invC, v);
            if (HAS_DFC) r = f4add(r, ldg4(dfc_out + (size_t)row * N + col));
            stg4(out + (size_t)row * N + col, r);
        });
    }
};
template <bool HAS_DFC>
struct EpAddDout4 {                 // C == 4: also emits dhbar[n] = sum_c dout[n,c] (gradient of the gate term)
    const float* dfc_out; const float* dmean; float* out; float* dhbar;
    __device__ __forceinline__ void chunk(const float* Ws, int row0, int col0, int ncols, int M, int N, int lane) const {
        chunk_quads_f4(Ws, row0, col0, ncols, M, N, lane, [&](int row, int col, const float4 (&v)[4]) {
            const int n = row >> 2;
            const float4 dm = ldg4(dmean + (size_t)n * N + col);
            float4 sum = dm;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const size_t o = ((size_t)row + c) * N + col;
                float4 r = f4fma(dm, 0.25f, v[c]);
                if (HAS_DFC) { const float4 d = ldg4(dfc_out + o); r = f4add(r, d); sum = f4add(sum, d); }
                stg4(out + o, r);
            }
            stg4(dhbar + (size_t)n * N + col, sum);
        });
    }
};

// ------------------------------------------------------------------ small element-wise helpers (C != 4 only)
__global__ void clip_mean_kernel(const float* __restrict__ fc, float* __restrict__ out, int N, int C, int D4, float invC)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)N * D4) return;
    const size_t n = idx / D4; const int d4 = (int)(idx % D4);
    float4 s = f4zero();
    for (int c = 0; c < C; ++c) s = f4add(s, ldg4(fc + ((n * C + c) * D4 + d4) * 4));
    stg4(out + idx * 4, f4scale(s, invC));
}

// dhbar[n] = dfcmean[n] + sum_c dfc_out[n,c]
__global__ void dout_sum_kernel(const float* __restrict__ dfc_out, const float* __restrict__ dmean, float* __restrict__ out, int N, int C, int D4)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)N * D4) return;
    const size_t n = idx / D4; const int d4 = (int)(idx % D4);
    float4 s = ldg4(dmean + idx * 4);
    if (dfc_out)
        for (int c = 0; c < C; ++c) s = f4add(s, ldg4(dfc_out + ((n * C + c) * D4 + d4) * 4));
    stg4(out + idx * 4, s);
}

}  // namespace smin

using namespace smin;

extern "C" int smin_content_unit_fwd(void* stream, const float* fc, const float* hbar, const int32_t* cells, const int32_t* row_ptr,
                                     int N, int B, int L, int C, int D, int dl, int Nq,
                                     const float* Wch, const float* bch, const float* Mq, const float* uq,
                                     const float* what, const float* shat, const float* qmask, const float* Wc, const float* bc,
                                     const float* fcmean_in, int last,
                                     float* fc_out, float* fcmean, float* chat, float* cchat)
{
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    SMIN_REQUIRE(D % 4 == 0 && dl % 16 == 0 && dl >= 16 && dl <= 128 && C >= 2 && C <= 4 && Nq >= 1 && Nq <= 32);
    if (N == 0) return 0;
    const int M = N * C;
    int rc = launch_gemm_nt(st, PlainMat{fc, D}, PlainMat{Wch, D}, EpBiasMask{bch, cells, chat, C}, M, dl, D);
    if (rc) return rc;
    rc = launch_content_attn_fwd(st, chat, cells, row_ptr, N, B, L, C, Mq, uq, what, shat, qmask, last ? nullptr : cchat, last ? cchat : nullptr, dl, Nq);
    if (rc) return rc;
    if (last)                       // cchat holds mean_c cchat [N][dl]; a quarter of the rows, and fc_out is never written
        return launch_gemm_nt(st, PlainMat{cchat, dl}, PlainMat{Wc, dl}, EpLastMean{bc, cells, fcmean_in, hbar, fcmean}, N, D, dl);
    if (C == 4)
        return launch_gemm_nt(st, PlainMat{cchat, dl}, PlainMat{Wc, dl}, EpContentOut4{bc, cells, fc, hbar, fc_out, fcmean}, M, D, dl);
    rc = launch_gemm_nt(st, PlainMat{cchat, dl}, PlainMat{Wc, dl}, EpContentOut{bc, cells, fc, hbar, fc_out, C}, M, D, dl);
    if (rc) return rc;
    const int D4 = D / 4;
    const size_t tot = (size_t)N * D4;
    hipLaunchKernelGGL(clip_mean_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, fc_out, fcmean, N, C, D4, 1.0f / C);
    SMIN_LAUNCH_CHECK();
    return 0;
}

template <bool HAS_DFC>
static int content_unit_bwd_impl(hipStream_t st, const float* dfc_out, const float* dfcmean,
                                 const float* fc, const int32_t* cells, const int32_t* row_ptr,
                                 int N, int B, int L, int C, int D, int dl, int Nq,
                                 const float* WchT, const float* Mq, const float* uq,
                                 const float* what, const float* shat, const float* qmask, const float* WcT,
                                 const float* chat, const float* cchat,
                                 float* dfc, float* dhbar, float* dWch, float* dbch, float* dMq, float* duq,
                                 float* dwhat, float* dshat, float* dWc, float* dbc, void* ws, size_t ws_bytes, int last)
{
    const int M = N * C;
    const float invC = 1.0f / C;
    float* w = reinterpret_cast<float*>(ws);
    size_t off = 0;
    auto take = [&](size_t n) { float* p = w + off; off += (n + 3) & ~(size_t)3; return p; };
    float* dcchat = take((size_t)M * dl);
    float* dchat = take((size_t)M * dl);
    const int sp1 = tn_splits(last ? N : M, D, dl), sp2 = tn_splits(M, dl, D);
    float* slab1 = take((size_t)sp1 * D * dl); float* bslab1 = take((size_t)sp1 * D);
    float* slab2 = take((size_t)sp2 * dl * D); float* bslab2 = take((size_t)sp2 * dl);
    float* aws = take(content_attn_bwd_ws_floats(N, B, dl));
    SMIN_REQUIRE(off * sizeof(float) <= ws_bytes);
    const DoutEffMat<true, HAS_DFC> dout{dfc_out, dfcmean, cells, C, D, invC};

    int rc;
    if (last) {
        // only mean_c of the output was consumed: the gradient row dfcmean/C is shared by the clips of a cell, so
        // (a) and (b) run on N rows with the saved clip mean (cchat = mean_c cchat [N][dl]) instead of N*C rows.
        rc = launch_gemm_nt(st, MaskedRowsMat{dfcmean, D, cells}, PlainMat{WcT, D}, EpScale{dcchat, invC}, N, dl, D);
        if (rc) return rc;
        rc = launch_gemm_tn(st, MaskedRowsMat{dfcmean, D, cells}, PlainMat{cchat, dl}, slab1, bslab1, N, D, dl, sp1);
        if (rc) return rc;
    } else {
        // (a) dcchat = (dout * m) @ Wc          [M, dl], contraction over D
        rc = launch_gemm_nt(st, dout, PlainMat{WcT, D}, EpPlain{dcchat}, M, dl, D);
        if (rc) return rc;
        // (b) dWc[D, dl] = (dout*m)^T @ cchat ; dbc = colsum(dout*m)
        rc = launch_gemm_tn(st, dout, PlainMat{cchat, dl}, slab1, bslab1, M, D, dl, sp1);
        if (rc) return rc;
    }
    rc = launch_reduce_slabs2(st, slab1, dWc, D * dl, bslab1, dbc, D, sp1); if (rc) return rc;
    // (c) attention core backward -> dchat (already multiplied by m: masked cells write 0)
    rc = launch_content_attn_bwd(st, chat, dcchat, cells, row_ptr, N, B, L, C, Mq, uq, what, shat, qmask, dchat, dMq, duq, dwhat, dshat, aws, dl, Nq, last, 1.0f, nullptr, 0.f);
    if (rc) return rc;
    // (d) dfc = dchat @ Wch + dout (residual)     [M, D], contraction over dl;  dhbar = sum_c dout (gate term)
    if (C == 4) {
        rc = launch_gemm_nt(st, PlainMat{dchat, dl}, PlainMat{WchT, dl}, EpAddDout4<HAS_DFC>{dfc_out, dfcmean, dfc, dhbar}, M, D, dl);
        if (rc) return rc;
    } else {
        rc = launch_gemm_nt(st, PlainMat{dchat, dl}, PlainMat{WchT, dl}, EpAddDout<HAS_DFC>{dfc_out, dfcmean, dfc, C, invC}, M, D, dl);
        if (rc) return rc;
        const size_t tot = (size_t)N * (D / 4);
        hipLaunchKernelGGL(dout_sum_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, dfc_out, dfcmean, dhbar, N, C, D / 4);
        SMIN_LAUNCH_CHECK();
    }
    // (e) dWch[dl, D] = dchat^T @ fc ; dbch = colsum(dchat)
    rc = launch_gemm_tn(st, PlainMat{dchat, dl}, PlainMat{fc, D}, slab2, bslab2, M, dl, D, sp2);
    if (rc) return rc;
    rc = launch_reduce_slabs2(st, slab2, dWch, dl * D, bslab2, dbch, dl, sp2); if (rc) return rc;
    return 0;
}

extern "C" int smin_content_unit_bwd(void* stream, const float* dfc_out, const float* dfcmean,
                                     const float* fc, const int32_t* cells, const int32_t* row_ptr,
                                     int N, int B, int L, int C, int D, int dl, int Nq,
                                     const float* WchT, const float* Mq, const float* uq,
                                     const float* what, const float* shat, const float* qmask, const float* WcT,
                                     const float* chat, const float* cchat,
                                     float* dfc, float* dhbar, float* dWch, float* dbch, float* dMq, float* duq,
                                     float* dwhat, float* dshat, float* dWc, float* dbc, void* ws, size_t ws_bytes, int last)
{
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    SMIN_REQUIRE(!(last && dfc_out));                           // "last" means nothing consumed fc_out itself
    SMIN_REQUIRE(D % 4 == 0 && dl % 16 == 0 && dl >= 16 && dl <= 128 && C >= 2 && C <= 4 && Nq >= 1 && Nq <= 32);
    if (N == 0) return 0;
    if (dfc_out)
        return content_unit_bwd_impl<true>(st, dfc_out, dfcmean, fc, cells, row_ptr, N, B, L, C, D, dl, Nq, WchT, Mq, uq, what, shat, qmask,
                                           WcT, chat, cchat, dfc, dhbar, dWch, dbch, dMq, duq, dwhat, dshat, dWc, dbc, ws, ws_bytes, last);
    return content_unit_bwd_impl<false>(st, nullptr, dfcmean, fc, cells, row_ptr, N, B, L, C, D, dl, Nq, WchT, Mq, uq, what, shat, qmask,
                                        WcT, chat, cchat, dfc, dhbar, dWch, dbch, dMq, duq, dwhat, dshat, dWc, dbc, ws, ws_bytes, last);
}
