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
#include "gemm.h"
#include "smin_hip.h"

namespace smin {

// ------------------------------------------------------------------ epilogues (see gemm.h: tile protocol)
struct EpBiasMask {                 // chat[row][col] = (acc + bias[col]) * m[row / C]
    const float* bias; const int* cells; float* out; int C;
    __device__ __forceinline__ void tile(const float* Cs, int row_base, int col_base, int M, int N, int t) const {
        tile_rows_f4(Cs, row_base, col_base, M, N, t, [&](int row, int col, float4 v) {
            const float m = (float)cells[4 * (size_t)(row / C) + 3];
            stg4(out + (size_t)row * N + col, f4scale(f4add(v, ldg4(bias + col)), m));
        });
    }
};

// out = (acc + bc) * m + fc + hbar[n]      (models.py:269-276; hbar = sigmoid(fm*fs)*fm from gate.hip)
struct EpContentOut {               // any C
    const float* bc; const int* cells; const float* fc; const float* hbar; float* out; int C;
    __device__ __forceinline__ void tile(const float* Cs, int row_base, int col_base, int M, int N, int t) const {
        tile_rows_f4(Cs, row_base, col_base, M, N, t, [&](int row, int col, float4 v) {
            const int n = row / C;
            const float m = (float)cells[4 * (size_t)n + 3];
            const float4 o = f4add(f4add(f4scale(f4add(v, ldg4(bc + col)), m), ldg4(fc + (size_t)row * N + col)), ldg4(hbar + (size_t)n * N + col));
            stg4(out + (size_t)row * N + col, o);
        });
    }
};
struct EpContentOut4 {              // C == 4: one lane owns the 4 clips of a cell -> also emits fcmean = mean_c out
    const float* bc; const int* cells; const float* fc; const float* hbar; float* out; float* fcmean;
    __device__ __forceinline__ void tile(const float* Cs, int row_base, int col_base, int M, int N, int t) const {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int idx = t + 256 * it, cl = idx >> 5, c4 = (idx & 31) * 4;
            const int n = (row_base >> 2) + cl, col = col_base + c4;
            if (4 * n < M && col < N) {
                const float m = (float)cells[4 * (size_t)n + 3];
                const float4 b4 = ldg4(bc + col), hb = ldg4(hbar + (size_t)n * N + col);
                float4 sum = f4zero();
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const size_t o = ((size_t)n * 4 + c) * N + col;
                    const float4 v = ldg4(Cs + (cl * 4 + c) * GEMM_LDC + c4);
                    const float4 r = f4add(f4add(f4scale(f4add(v, b4), m), ldg4(fc + o)), hb);
                    stg4(out + o, r);
                    sum = f4add(sum, r);
                }
                stg4(fcmean + (size_t)n * N + col, f4scale(sum, 0.25f));
            }
        }
    }
};

struct EpPlain {
    float* out;
    __device__ __forceinline__ void tile(const float* Cs, int row_base, int col_base, int M, int N, int t) const {
        tile_rows_f4(Cs, row_base, col_base, M, N, t, [&](int row, int col, float4 v) { stg4(out + (size_t)row * N + col, v); });
    }
};

// dfc = acc + dfc_out + dfcmean / C   (residual path, un-masked)
template <bool HAS_DFC>
struct EpAddDout {                  // any C
    const float* dfc_out; const float* dmean; float* out; int C; float invC;
    __device__ __forceinline__ void tile(const float* Cs, int row_base, int col_base, int M, int N, int t) const {
        tile_rows_f4(Cs, row_base, col_base, M, N, t, [&](int row, int col, float4 v) {
            float4 r = f4fma(ldg4(dmean + (size_t)(row / C) * N + col), invC, v);
            if (HAS_DFC) r = f4add(r, ldg4(dfc_out + (size_t)row * N + col));
            stg4(out + (size_t)row * N + col, r);
        });
    }
};
template <bool HAS_DFC>
struct EpAddDout4 {                 // C == 4: also emits dhbar[n] = sum_c dout[n,c] (gradient of the gate term)
    const float* dfc_out; const float* dmean; float* out; float* dhbar;
    __device__ __forceinline__ void tile(const float* Cs, int row_base, int col_base, int M, int N, int t) const {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int idx = t + 256 * it, cl = idx >> 5, c4 = (idx & 31) * 4;
            const int n = (row_base >> 2) + cl, col = col_base + c4;
            if (4 * n < M && col < N) {
                const float4 dm = ldg4(dmean + (size_t)n * N + col);
                float4 sum = dm;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const size_t o = ((size_t)n * 4 + c) * N + col;
                    float4 r = f4fma(dm, 0.25f, ldg4(Cs + (cl * 4 + c) * GEMM_LDC + c4));
                    if (HAS_DFC) { const float4 d = ldg4(dfc_out + o); r = f4add(r, d); sum = f4add(sum, d); }
                    stg4(out + o, r);
                }
                stg4(dhbar + (size_t)n * N + col, sum);
            }
        }
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

// ------------------------------------------------------------------ attention core
// Shared-memory carve (floats): sM[NQP][dl+4] sW[NQP][dl+4] sS[dl] sU[NQP] sQ[NQP]
//                               then per wave: sC[C][dl+4] sD[C][dl+4] sP[C][32] sG[C][32]
struct AttnSmem {
    float *sM, *sW, *sS, *sU, *sQ, *sC, *sD, *sP, *sG;
    int ldw;
    __device__ AttnSmem(float* base, int C, int dl, int NQP, int wave) {
        ldw = dl + 4;
        sM = base; sW = sM + NQP * ldw; sS = sW + NQP * ldw; sU = sS + dl; sQ = sU + NQP;
        float* w0 = sQ + NQP + wave * (2 * C * ldw + 2 * C * 32);
        sC = w0; sD = sC + C * ldw; sP = sD + C * ldw; sG = sP + C * 32;
    }
    static size_t bytes(int C, int dl, int NQP) {
        return sizeof(float) * (size_t)(2 * NQP * (dl + 4) + dl + 2 * NQP + 4 * (2 * C * (dl + 4) + 2 * C * 32));
    }
};

__device__ __forceinline__ void attn_stage_sample(const AttnSmem& sm, const float* Mq, const float* uq, const float* what,
                                                   const float* shat, const float* qmask, int b, int dl, int Nq, int NQP)
{
    const int t = threadIdx.x;
    for (int idx = t; idx < NQP * dl; idx += 256) {
        const int w = idx / dl, d = idx % dl;
        const bool ok = w < Nq;
        sm.sM[w * sm.ldw + d] = ok ? Mq[((size_t)b * Nq + w) * dl + d] : 0.f;
        sm.sW[w * sm.ldw + d] = ok ? what[((size_t)b * Nq + w) * dl + d] : 0.f;
    }
    for (int d = t; d < dl; d += 256) sm.sS[d] = shat[(size_t)b * dl + d];
    for (int w = t; w < NQP; w += 256) {
        sm.sU[w] = w < Nq ? uq[(size_t)b * Nq + w] : 0.f;
        sm.sQ[w] = w < Nq ? qmask[(size_t)b * Nq + w] : 0.f;
    }
}

// dot of two LDS rows of length dl (dl % 4 == 0)
__device__ __forceinline__ float lds_dot(const float* a, const float* b, int dl)
{
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (int k = 0; k < dl; k += 4) {
        const float4 x = ldg4(a + k), y = ldg4(b + k);
        s0 = fmaf(x.x, y.x, s0); s1 = fmaf(x.y, y.y, s1); s2 = fmaf(x.z, y.z, s2); s3 = fmaf(x.w, y.w, s3);
    }
    return (s0 + s1) + (s2 + s3);
}

// Recomputable forward core for one cell (one wave).  ch[c][e] holds chat[c][lane + 64 e].
template <int C, int DPL>
struct CoreState {
    float ch[C][DPL], a[C][DPL], q[C][DPL], A[C][C], P[(C + 1) / 2];
};

template <int C, int DPL>
__device__ __forceinline__ void attn_core_forward(CoreState<C, DPL>& st, const AttnSmem& sm, const float* chat_rows,
                                                   int dl, int Nq, float scale, int lane)
{
    constexpr int NS = (C + 1) / 2;
    const int h = lane >> 5, w = lane & 31;
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
        for (int e = 0; e < DPL; ++e) {
            const int d = lane + 64 * e;
            st.ch[c][e] = d < dl ? chat_rows[(size_t)c * dl + d] : 0.f;
            if (d < dl) sm.sC[c * sm.ldw + d] = st.ch[c][e];
        }
    __builtin_amdgcn_wave_barrier();
    // scores: lane (h, w) of slot k owns (c = 2k + h, word w)
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        const int c = 2 * k + h;
        float s = -INFINITY;
        if (c < C && w < Nq) {
            s = (lds_dot(sm.sC + c * sm.ldw, sm.sM + w * sm.ldw, dl) + sm.sU[w]) * scale;
            const float qm = sm.sQ[w];
            s = (qm == 0.f) ? -1e9f : s * qm;                    // models.py:216-218
        }
        const float mx = half_max(s);
        const float ex = (c < C && w < Nq) ? expf(s - mx) : 0.f;
        const float den = half_sum(ex);
        const float p = (c < C && w < Nq) ? ex / den : 0.f;
        st.P[k] = p;
        if (c < C) sm.sP[c * 32 + w] = p;
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
        for (int e = 0; e < DPL; ++e) st.a[c][e] = 0.f;
    for (int ww = 0; ww < Nq; ++ww) {
#pragma unroll
        for (int e = 0; e < DPL; ++e) {
            const int d = lane + 64 * e;
            const float wv = d < dl ? sm.sW[ww * sm.ldw + d] : 0.f;
#pragma unroll
            for (int c = 0; c < C; ++c) st.a[c][e] = fmaf(sm.sP[c * 32 + ww], wv, st.a[c][e]);
        }
    }
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
        for (int e = 0; e < DPL; ++e) {
            const int d = lane + 64 * e;
            st.q[c][e] = st.ch[c][e] * (st.a[c][e] + (d < dl ? sm.sS[d] : 0.f));
        }
    float Z[C][C];
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
        for (int c2 = c; c2 < C; ++c2) {
            float part = 0.f;
#pragma unroll
            for (int e = 0; e < DPL; ++e) part = fmaf(st.q[c][e], st.q[c2][e], part);
            Z[c][c2] = wave_sum(part) * scale;
            Z[c2][c] = Z[c][c2];
        }
#pragma unroll
    for (int c = 0; c < C; ++c) {                                // models.py:262 (no mask inside this softmax)
        float mx = Z[c][0];
#pragma unroll
        for (int c2 = 1; c2 < C; ++c2) mx = fmaxf(mx, Z[c][c2]);
        float den = 0.f;
#pragma unroll
        for (int c2 = 0; c2 < C; ++c2) { st.A[c][c2] = expf(Z[c][c2] - mx); den += st.A[c][c2]; }
        const float inv = 1.0f / den;
#pragma unroll
        for (int c2 = 0; c2 < C; ++c2) st.A[c][c2] *= inv;
    }
}

template <int C, int DPL>
__global__ __launch_bounds__(256)
void content_attn_fwd_kernel(const float* __restrict__ chat, const int* __restrict__ cells, const int* __restrict__ row_ptr, int L,
                             const float* __restrict__ Mq, const float* __restrict__ uq, const float* __restrict__ what,
                             const float* __restrict__ shat, const float* __restrict__ qmask,
                             float* __restrict__ cchat, int dl, int Nq, int NQP, int cells_per_chunk, float scale)
{
    extern __shared__ __attribute__((aligned(16))) float smem_dyn[];
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int s0 = row_ptr[b * L], s1 = row_ptr[(b + 1) * L];
    const int n_begin = s0 + chunk * cells_per_chunk;
    if (n_begin >= s1) return;
    const int n_end = min(s1, n_begin + cells_per_chunk);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    AttnSmem sm(smem_dyn, C, dl, NQP, wave);
    attn_stage_sample(sm, Mq, uq, what, shat, qmask, b, dl, Nq, NQP);
    __syncthreads();

    CoreState<C, DPL> st;
    for (int n = n_begin + wave; n < n_end; n += 4) {
        float* out = cchat + (size_t)n * C * dl;
        if (cells[4 * (size_t)n + 3] == 0) {                     // masked cell: A = 0 (models.py:263)
#pragma unroll
            for (int c = 0; c < C; ++c)
#pragma unroll
                for (int e = 0; e < DPL; ++e) { const int d = lane + 64 * e; if (d < dl) out[c * dl + d] = 0.f; }
            continue;
        }
        attn_core_forward<C, DPL>(st, sm, chat + (size_t)n * C * dl, dl, Nq, scale, lane);
#pragma unroll
        for (int c = 0; c < C; ++c)
#pragma unroll
            for (int e = 0; e < DPL; ++e) {
                float v = 0.f;
#pragma unroll
                for (int c2 = 0; c2 < C; ++c2) v = fmaf(st.A[c][c2], st.ch[c2][e], v);
                const int d = lane + 64 * e;
                if (d < dl) out[c * dl + d] = v;
            }
        __builtin_amdgcn_wave_barrier();
    }
}

// Backward of the core.  Per-wave partial slab layout (floats): dM[NQP][dl] dW[NQP][dl] dshat[dl] du[32]
template <int C, int DPL, int NQP>
__global__ __launch_bounds__(256)
void content_attn_bwd_kernel(const float* __restrict__ chat, const float* __restrict__ dcchat,
                             const int* __restrict__ cells, const int* __restrict__ row_ptr, int L,
                             const float* __restrict__ Mq, const float* __restrict__ uq, const float* __restrict__ what,
                             const float* __restrict__ shat, const float* __restrict__ qmask,
                             float* __restrict__ dchat, float* __restrict__ slab,
                             int dl, int Nq, int cells_per_chunk, int max_chunks, float scale)
{
    extern __shared__ __attribute__((aligned(16))) float smem_dyn[];
    constexpr int NS = (C + 1) / 2;
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int s0 = row_ptr[b * L], s1 = row_ptr[(b + 1) * L];
    const int n_begin = s0 + chunk * cells_per_chunk;
    if (n_begin >= s1) return;
    const int n_end = min(s1, n_begin + cells_per_chunk);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, h = lane >> 5, w = lane & 31;
    AttnSmem sm(smem_dyn, C, dl, NQP, wave);
    attn_stage_sample(sm, Mq, uq, what, shat, qmask, b, dl, Nq, NQP);
    __syncthreads();

    float dM[NQP][DPL], dW[NQP][DPL], dsh[DPL], du[NS];
#pragma unroll
    for (int ww = 0; ww < NQP; ++ww)
#pragma unroll
        for (int e = 0; e < DPL; ++e) { dM[ww][e] = 0.f; dW[ww][e] = 0.f; }
#pragma unroll
    for (int e = 0; e < DPL; ++e) dsh[e] = 0.f;
#pragma unroll
    for (int k = 0; k < NS; ++k) du[k] = 0.f;

    CoreState<C, DPL> st;
    for (int n = n_begin + wave; n < n_end; n += 4) {
        float* out = dchat + (size_t)n * C * dl;
        if (cells[4 * (size_t)n + 3] == 0) {
#pragma unroll
            for (int c = 0; c < C; ++c)
#pragma unroll
                for (int e = 0; e < DPL; ++e) { const int d = lane + 64 * e; if (d < dl) out[c * dl + d] = 0.f; }
            continue;
        }
        attn_core_forward<C, DPL>(st, sm, chat + (size_t)n * C * dl, dl, Nq, scale, lane);

        float g[C][DPL], dch[C][DPL];
#pragma unroll
        for (int c = 0; c < C; ++c)
#pragma unroll
            for (int e = 0; e < DPL; ++e) {
                const int d = lane + 64 * e;
                g[c][e] = d < dl ? dcchat[((size_t)n * C + c) * dl + d] : 0.f;
            }
        // cchat = A chat
        float dA[C][C];
#pragma unroll
        for (int c = 0; c < C; ++c)
#pragma unroll
            for (int c2 = 0; c2 < C; ++c2) {
                float part = 0.f;
#pragma unroll
                for (int e = 0; e < DPL; ++e) part = fmaf(g[c][e], st.ch[c2][e], part);
                dA[c][c2] = wave_sum(part);
            }
#pragma unroll
        for (int c2 = 0; c2 < C; ++c2)
#pragma unroll
            for (int e = 0; e < DPL; ++e) {
                float v = 0.f;
#pragma unroll
                for (int c = 0; c < C; ++c) v = fmaf(st.A[c][c2], g[c][e], v);
                dch[c2][e] = v;
            }
        // A = softmax(Z), Z = q q^T * scale
        float dZ[C][C];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            float rd = 0.f;
#pragma unroll
            for (int c2 = 0; c2 < C; ++c2) rd = fmaf(st.A[c][c2], dA[c][c2], rd);
#pragma unroll
            for (int c2 = 0; c2 < C; ++c2) dZ[c][c2] = st.A[c][c2] * (dA[c][c2] - rd);
        }
        float da[C][DPL];
#pragma unroll
        for (int c = 0; c < C; ++c)
#pragma unroll
            for (int e = 0; e < DPL; ++e) {
                float dq = 0.f;
#pragma unroll
                for (int c2 = 0; c2 < C; ++c2) dq = fmaf((dZ[c][c2] + dZ[c2][c]) * scale, st.q[c2][e], dq);
                const int d = lane + 64 * e;
                const float sh = d < dl ? sm.sS[d] : 0.f;
                dch[c][e] = fmaf(dq, st.a[c][e] + sh, dch[c][e]);     // q = chat * (a + shat)
                da[c][e] = dq * st.ch[c][e];
                dsh[e] += da[c][e];
                if (d < dl) sm.sD[c * sm.ldw + d] = da[c][e];
            }
        __builtin_amdgcn_wave_barrier();
        // a = P what ; P = softmax(S)
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const int c = 2 * k + h;
            float dp = 0.f;
            if (c < C && w < Nq) dp = lds_dot(sm.sD + c * sm.ldw, sm.sW + w * sm.ldw, dl);
            const float pd = half_sum(st.P[k] * dp);
            float ds = st.P[k] * (dp - pd);
            ds = (c < C && w < Nq) ? ds * sm.sQ[w] * scale : 0.f;   // S = raw * scale * qmask
            du[k] += ds;
            if (c < C) sm.sG[c * 32 + w] = ds;
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int ww = 0; ww < NQP; ++ww) {
            float gs[C], ps[C];
#pragma unroll
            for (int c = 0; c < C; ++c) { gs[c] = sm.sG[c * 32 + ww]; ps[c] = sm.sP[c * 32 + ww]; }
#pragma unroll
            for (int e = 0; e < DPL; ++e) {
                const int d = lane + 64 * e;
                const float mv = d < dl ? sm.sM[ww * sm.ldw + d] : 0.f;
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    dch[c][e] = fmaf(gs[c], mv, dch[c][e]);
                    dM[ww][e] = fmaf(gs[c], st.ch[c][e], dM[ww][e]);
                    dW[ww][e] = fmaf(ps[c], da[c][e], dW[ww][e]);
                }
            }
        }
#pragma unroll
        for (int c = 0; c < C; ++c)
#pragma unroll
            for (int e = 0; e < DPL; ++e) { const int d = lane + 64 * e; if (d < dl) out[c * dl + d] = dch[c][e]; }
        __builtin_amdgcn_wave_barrier();
    }

    const size_t slab_sz = (size_t)2 * NQP * dl + dl + 32;
    float* sl = slab + (((size_t)b * max_chunks + chunk) * 4 + wave) * slab_sz;
#pragma unroll
    for (int ww = 0; ww < NQP; ++ww)
#pragma unroll
        for (int e = 0; e < DPL; ++e) {
            const int d = lane + 64 * e;
            if (d < dl) { sl[(size_t)ww * dl + d] = dM[ww][e]; sl[(size_t)(NQP + ww) * dl + d] = dW[ww][e]; }
        }
#pragma unroll
    for (int e = 0; e < DPL; ++e) { const int d = lane + 64 * e; if (d < dl) sl[(size_t)2 * NQP * dl + d] = dsh[e]; }
    float dut = 0.f;
#pragma unroll
    for (int k = 0; k < NS; ++k) dut += du[k];
    dut += __shfl_xor(dut, 32);
    if (lane < 32) sl[(size_t)2 * NQP * dl + dl + lane] = dut;
}

// Fixed-order sum of the per-wave slabs of each sample, scattered to the un-padded outputs.
__global__ void content_attn_reduce_kernel(const float* __restrict__ slab, const int* __restrict__ row_ptr, int L,
                                           int dl, int Nq, int NQP, int cells_per_chunk, int max_chunks,
                                           float* __restrict__ dMq, float* __restrict__ dwhat, float* __restrict__ dshat, float* __restrict__ duq)
{
    const int b = blockIdx.y;
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int slab_sz = 2 * NQP * dl + dl + 32;
    if (x >= slab_sz) return;
    const int ncell = row_ptr[(b + 1) * L] - row_ptr[b * L];
    const int nslab = ((ncell + cells_per_chunk - 1) / cells_per_chunk) * 4;
    const float* p = slab + (size_t)b * max_chunks * 4 * slab_sz + x;
    float s = 0.f;
    for (int k = 0; k < nslab; ++k) s += p[(size_t)k * slab_sz];
    if (x < NQP * dl) { const int w = x / dl, d = x % dl; if (w < Nq) dMq[((size_t)b * Nq + w) * dl + d] = s; }
    else if (x < 2 * NQP * dl) { const int y = x - NQP * dl, w = y / dl, d = y % dl; if (w < Nq) dwhat[((size_t)b * Nq + w) * dl + d] = s; }
    else if (x < 2 * NQP * dl + dl) dshat[(size_t)b * dl + (x - 2 * NQP * dl)] = s;
    else { const int w = x - 2 * NQP * dl - dl; if (w < Nq) duq[(size_t)b * Nq + w] = s; }
}

static inline int nq_pad(int Nq) { return Nq <= 8 ? 8 : Nq <= 16 ? 16 : Nq <= 24 ? 24 : 32; }

template <int C, int DPL>
static int launch_attn_fwd(hipStream_t st, const float* chat, const int* cells, const int* row_ptr, int B, int L,
                           const float* Mq, const float* uq, const float* what, const float* shat, const float* qmask,
                           float* cchat, int dl, int Nq)
{
    int cpc, mc; chunking(L, &cpc, &mc);
    const int NQP = nq_pad(Nq);
    const size_t smem = AttnSmem::bytes(C, dl, NQP);
    hipLaunchKernelGGL((content_attn_fwd_kernel<C, DPL>), dim3(mc, B), dim3(256), smem, st, chat, cells, row_ptr, L,
                       Mq, uq, what, shat, qmask, cchat, dl, Nq, NQP, cpc, 1.0f / sqrtf((float)dl));
    SMIN_LAUNCH_CHECK();
    return 0;
}

template <int C, int DPL, int NQP>
static int launch_attn_bwd(hipStream_t st, const float* chat, const float* dcchat, const int* cells, const int* row_ptr, int B, int L,
                           const float* Mq, const float* uq, const float* what, const float* shat, const float* qmask,
                           float* dchat, float* slab, int dl, int Nq)
{
    int cpc, mc; chunking(L, &cpc, &mc);
    const size_t smem = AttnSmem::bytes(C, dl, NQP);
    hipLaunchKernelGGL((content_attn_bwd_kernel<C, DPL, NQP>), dim3(mc, B), dim3(256), smem, st, chat, dcchat, cells, row_ptr, L,
                       Mq, uq, what, shat, qmask, dchat, slab, dl, Nq, cpc, mc, 1.0f / sqrtf((float)dl));
    SMIN_LAUNCH_CHECK();
    return 0;
}

template <int C, int DPL>
static int dispatch_attn_bwd_nq(int NQP, hipStream_t st, const float* chat, const float* dcchat, const int* cells, const int* row_ptr,
                                int B, int L, const float* Mq, const float* uq, const float* what, const float* shat,
                                const float* qmask, float* dchat, float* slab, int dl, int Nq)
{
    switch (NQP) {
    case 8: return launch_attn_bwd<C, DPL, 8>(st, chat, dcchat, cells, row_ptr, B, L, Mq, uq, what, shat, qmask, dchat, slab, dl, Nq);
    case 16: return launch_attn_bwd<C, DPL, 16>(st, chat, dcchat, cells, row_ptr, B, L, Mq, uq, what, shat, qmask, dchat, slab, dl, Nq);
    case 24: return launch_attn_bwd<C, DPL, 24>(st, chat, dcchat, cells, row_ptr, B, L, Mq, uq, what, shat, qmask, dchat, slab, dl, Nq);
    default: return launch_attn_bwd<C, DPL, 32>(st, chat, dcchat, cells, row_ptr, B, L, Mq, uq, what, shat, qmask, dchat, slab, dl, Nq);
    }
}

}  // namespace smin

using namespace smin;

extern "C" int smin_content_unit_fwd(void* stream, const float* fc, const float* hbar, const int32_t* cells, const int32_t* row_ptr,
                                     int N, int B, int L, int C, int D, int dl, int Nq,
                                     const float* Wch, const float* bch, const float* Mq, const float* uq,
                                     const float* what, const float* shat, const float* qmask, const float* Wc, const float* bc,
                                     float* fc_out, float* fcmean, float* chat, float* cchat)
{
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    SMIN_REQUIRE(D % 4 == 0 && dl % 4 == 0 && dl <= 128 && C >= 2 && C <= 4 && Nq >= 1 && Nq <= 32);
    if (N == 0) return 0;
    const int M = N * C;
    int rc = launch_gemm_nt(st, PlainMat{fc, D}, PlainMat{Wch, D}, EpBiasMask{bch, cells, chat, C}, M, dl, D);
    if (rc) return rc;
    const int DPL = cdiv(dl, 64);
#define ATTN_FWD(CC, DD) rc = launch_attn_fwd<CC, DD>(st, chat, cells, row_ptr, B, L, Mq, uq, what, shat, qmask, cchat, dl, Nq)
    if (C == 4) { if (DPL == 1) ATTN_FWD(4, 1); else ATTN_FWD(4, 2); }
    else if (C == 3) { if (DPL == 1) ATTN_FWD(3, 1); else ATTN_FWD(3, 2); }
    else { if (DPL == 1) ATTN_FWD(2, 1); else ATTN_FWD(2, 2); }
#undef ATTN_FWD
    if (rc) return rc;
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
                                 float* dwhat, float* dshat, float* dWc, float* dbc, void* ws, size_t ws_bytes)
{
    const int M = N * C;
    const int NQP = nq_pad(Nq);
    int cpc, mc; chunking(L, &cpc, &mc);
    const float invC = 1.0f / C;
    float* w = reinterpret_cast<float*>(ws);
    size_t off = 0;
    auto take = [&](size_t n) { float* p = w + off; off += (n + 3) & ~(size_t)3; return p; };
    float* dcchat = take((size_t)M * dl);
    float* dchat = take((size_t)M * dl);
    const int sp1 = tn_splits(M, D, dl), sp2 = tn_splits(M, dl, D);
    float* slab1 = take((size_t)sp1 * D * dl); float* bslab1 = take((size_t)sp1 * D);
    float* slab2 = take((size_t)sp2 * dl * D); float* bslab2 = take((size_t)sp2 * dl);
    const size_t attn_slab_sz = (size_t)2 * NQP * dl + dl + 32;
    float* aslab = take((size_t)B * mc * 4 * attn_slab_sz);
    SMIN_REQUIRE(off * sizeof(float) <= ws_bytes);
    const DoutEffMat<true, HAS_DFC> dout{dfc_out, dfcmean, cells, C, D, invC};

    // (a) dcchat = (dout * m) @ Wc          [M, dl], contraction over D
    int rc = launch_gemm_nt(st, dout, PlainMat{WcT, D}, EpPlain{dcchat}, M, dl, D);
    if (rc) return rc;
    // (b) dWc[D, dl] = (dout*m)^T @ cchat ; dbc = colsum(dout*m)
    rc = launch_gemm_tn(st, dout, PlainMat{cchat, dl}, slab1, bslab1, M, D, dl, sp1);
    if (rc) return rc;
    rc = launch_reduce_slabs(st, slab1, dWc, D * dl, sp1); if (rc) return rc;
    rc = launch_reduce_slabs(st, bslab1, dbc, D, sp1); if (rc) return rc;
    // (c) attention core backward -> dchat (already multiplied by m: masked cells write 0)
    const int DPL = cdiv(dl, 64);
#define ATTN_BWD(CC, DD) rc = dispatch_attn_bwd_nq<CC, DD>(NQP, st, chat, dcchat, cells, row_ptr, B, L, Mq, uq, what, shat, qmask, dchat, aslab, dl, Nq)
    if (C == 4) { if (DPL == 1) ATTN_BWD(4, 1); else ATTN_BWD(4, 2); }
    else if (C == 3) { if (DPL == 1) ATTN_BWD(3, 1); else ATTN_BWD(3, 2); }
    else { if (DPL == 1) ATTN_BWD(2, 1); else ATTN_BWD(2, 2); }
#undef ATTN_BWD
    if (rc) return rc;
    hipLaunchKernelGGL(content_attn_reduce_kernel, dim3(cdiv((int)attn_slab_sz, 256), B), dim3(256), 0, st, aslab, row_ptr, L,
                       dl, Nq, NQP, cpc, mc, dMq, dwhat, dshat, duq);
    SMIN_LAUNCH_CHECK();
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
    rc = launch_reduce_slabs(st, slab2, dWch, dl * D, sp2); if (rc) return rc;
    rc = launch_reduce_slabs(st, bslab2, dbch, dl, sp2); if (rc) return rc;
    return 0;
}

extern "C" int smin_content_unit_bwd(void* stream, const float* dfc_out, const float* dfcmean,
                                     const float* fc, const int32_t* cells, const int32_t* row_ptr,
                                     int N, int B, int L, int C, int D, int dl, int Nq,
                                     const float* WchT, const float* Mq, const float* uq,
                                     const float* what, const float* shat, const float* qmask, const float* WcT,
                                     const float* chat, const float* cchat,
                                     float* dfc, float* dhbar, float* dWch, float* dbch, float* dMq, float* duq,
                                     float* dwhat, float* dshat, float* dWc, float* dbc, void* ws, size_t ws_bytes)
{
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    SMIN_REQUIRE(D % 4 == 0 && dl % 4 == 0 && dl <= 128 && C >= 2 && C <= 4 && Nq >= 1 && Nq <= 32);
    if (N == 0) return 0;
    if (dfc_out)
        return content_unit_bwd_impl<true>(st, dfc_out, dfcmean, fc, cells, row_ptr, N, B, L, C, D, dl, Nq, WchT, Mq, uq, what, shat, qmask,
                                           WcT, chat, cchat, dfc, dhbar, dWch, dbch, dMq, duq, dwhat, dshat, dWc, dbc, ws, ws_bytes);
    return content_unit_bwd_impl<false>(st, nullptr, dfcmean, fc, cells, row_ptr, N, B, L, C, D, dl, Nq, WchT, Mq, uq, what, shat, qmask,
                                        WcT, chat, cchat, dfc, dhbar, dWch, dbch, dMq, duq, dwhat, dshat, dWc, dbc, ws, ws_bytes);
}
