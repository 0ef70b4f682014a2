// Word-side operands of every layer's content attention in one launch per direction.
//
// ContentUnit.forward (reference models.py:249-251) projects the query words and the sentence feature into the dl space,
// ContentAttention.forward (models.py:209-211) applies W_k to the words and W_q to every clip; the clip side of W_q is folded
// into the word side here (W_q(c) . W_k(w)^T = c . (W_k(w) W_q.weight)^T + W_k(w) . W_q.bias).  Per layer k and sample b:
//   what = (f_w WH^T + bWH) * qmask      [Nq][dl]      (WH, bWH = linear_w_hat)
//   shat =  f_s SH^T + bSH               [dl]          (linear_s_hat)
//   kb   =  what AK^T + bAK              [Nq][dl]      (attn_layer.W_k)
//   Mq   =  kb AQ                        [Nq][dl]      (attn_layer.W_q.weight)
//   uq   =  kb . bAQ                     [Nq]          (attn_layer.W_q.bias)
// O(B Nq D dl) work, ~0.8 GFLOP per step at the benchmark shape: as torch library calls this was ~60 launches forward and
// ~150 backward per step (linear / matmul / mul nodes, their transposes, and the gradient accumulations of parameters used
// several times); here one workgroup per (sample, layer) does the chain through LDS.  Backward: the same grid computes the
// row-side gradients (d f_w, d f_s) and per-sample partial weight gradients, a second launch sums them over the batch in
// fixed order (deterministic).
#include "common.h"
#include "smin_hip.h"

namespace smin {

constexpr int WP_MAXL = 8;                                    // layers per launch
struct WordParams { const float* p[WP_MAXL * 8]; };           // per layer: WH, bWH, SH, bSH, AK, bAK, AQ, bAQ
struct WordGrads { float* p[WP_MAXL * 8]; };
struct WordOuts { const float* dwhat[WP_MAXL]; const float* dshat[WP_MAXL]; const float* dMq[WP_MAXL]; const float* duq[WP_MAXL]; };

// out[w][o] = sum_k in[w][k] * W[o][k] (+ bias[o]) for the rows w = g, g + G, ... of a thread group; W rows from global (each
// thread walks one row with float4 loads, the LDS rows broadcast).  256 threads, O <= 256, K % 4 == 0, at most 16 rows per thread.
template <class F>
__device__ __forceinline__ void rows_matT(const float* in, int ldi, int nrows, const float* __restrict__ W, int O, int K, F store)
{
    const int t = threadIdx.x, G = 256 / O > 0 ? 256 / O : 1;
    const int o = t % O, g = t / O;
    if (g >= G || o >= O) return;
    float acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const float* wrow = W + (size_t)o * K;
    // eight weight loads in flight per thread (K % 32 == 0 in every use; the tail loop covers the rest): with one load per trip
    // the loop was a chain of L2 round trips, ~100 us per workgroup
    int k = 0;
    for (; k + 32 <= K; k += 32) {
        float4 w4[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) w4[u] = ldg4(wrow + k + 4 * u);
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int w = g + i * G;
                if (w < nrows) {
                    const float4 x = ldg4(in + w * ldi + k + 4 * u);
                    acc[i] = fmaf(x.x, w4[u].x, fmaf(x.y, w4[u].y, fmaf(x.z, w4[u].z, fmaf(x.w, w4[u].w, acc[i]))));
                }
            }
    }
    for (; k < K; k += 4) {
        const float4 w4 = ldg4(wrow + k);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int w = g + i * G;
            if (w < nrows) {
                const float4 x = ldg4(in + w * ldi + k);
                acc[i] = fmaf(x.x, w4.x, fmaf(x.y, w4.y, fmaf(x.z, w4.z, fmaf(x.w, w4.w, acc[i]))));
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) { const int w = g + i * G; if (w < nrows) store(w, o, acc[i]); }
}
// out[w][o] = sum_k in[w][k] * W[k][o]   (W row-major [K][O]: lanes walk consecutive columns); O may exceed 256
template <class F>
__device__ __forceinline__ void rows_mat(const float* in, int ldi, int nrows, const float* __restrict__ W, int K, int O, F store)
{
    for (int o = threadIdx.x; o < O; o += 256) {
        for (int w0 = 0; w0 < nrows; w0 += 16) {
            float acc[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
            int k = 0;
            for (; k + 16 <= K; k += 16) {                        // sixteen weight loads in flight per thread
                float wv[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) wv[u] = W[(size_t)(k + u) * O + o];
#pragma unroll
                for (int u = 0; u < 16; ++u)
#pragma unroll
                    for (int i = 0; i < 16; ++i) if (w0 + i < nrows) acc[i] = fmaf(in[(w0 + i) * ldi + k + u], wv[u], acc[i]);
            }
            for (; k < K; ++k) {
                const float wv = W[(size_t)k * O + o];
#pragma unroll
                for (int i = 0; i < 16; ++i) if (w0 + i < nrows) acc[i] = fmaf(in[(w0 + i) * ldi + k], wv, acc[i]);
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) if (w0 + i < nrows) store(w0 + i, o, acc[i]);
        }
    }
}

__global__ __launch_bounds__(256)
void word_prep_fwd_kernel(const float* __restrict__ fw, const float* __restrict__ fs, const float* __restrict__ qmask, WordParams P,
                          int B, int Nq, int D, int dl, int rpp, float* __restrict__ what, float* __restrict__ shat, float* __restrict__ kb,
                          float* __restrict__ Mq, float* __restrict__ uq)
{
    // grid (B, layers, row parts): every stage is row-wise, so a sample's words are cut into parts of rpp rows -- small batches
    // still fill the chip (tacos.yml: 2 samples x 3 layers would be 6 workgroups)
    extern __shared__ __attribute__((aligned(16))) float wp_smem[];
    const int b = blockIdx.x, k = blockIdx.y, t = threadIdx.x;
    const int r0 = blockIdx.z * rpp, nr = min(Nq, r0 + rpp) - r0;
    if (nr <= 0) return;
    const float *WH = P.p[8 * k], *bWH = P.p[8 * k + 1], *SH = P.p[8 * k + 2], *bSH = P.p[8 * k + 3], *AK = P.p[8 * k + 4], *bAK = P.p[8 * k + 5],
                *AQ = P.p[8 * k + 6], *bAQ = P.p[8 * k + 7];
    float* fwS = wp_smem;                                     // [nr][D]  (+ one row: f_s)
    float* whS = fwS + (rpp + 1) * D;                         // [nr][dl]
    float* kbS = whS + rpp * dl;                              // [nr][dl]
    for (int idx = t * 4; idx < nr * D; idx += 1024) stg4(fwS + idx, ldg4(fw + ((size_t)b * Nq + r0) * D + idx));
    for (int idx = t * 4; idx < D; idx += 1024) stg4(fwS + rpp * D + idx, ldg4(fs + (size_t)b * D + idx));
    __syncthreads();
    const size_t ob = (((size_t)k * B + b) * Nq + r0) * dl;
    rows_matT(fwS, D, nr, WH, dl, D, [&](int w, int e, float v) {
        v = (v + bWH[e]) * qmask[(size_t)b * Nq + r0 + w];
        whS[w * dl + e] = v; what[ob + (size_t)w * dl + e] = v;
    });
    if (blockIdx.z == 0) rows_matT(fwS + rpp * D, D, 1, SH, dl, D, [&](int, int e, float v) { shat[((size_t)k * B + b) * dl + e] = v + bSH[e]; });
    __syncthreads();
    rows_matT(whS, dl, nr, AK, dl, dl, [&](int w, int e, float v) {
        v += bAK[e];
        kbS[w * dl + e] = v; kb[ob + (size_t)w * dl + e] = v;
    });
    __syncthreads();
    rows_mat(kbS, dl, nr, AQ, dl, dl, [&](int w, int d, float v) { Mq[ob + (size_t)w * dl + d] = v; });
    if (t < nr) {
        float s = 0.f;
        for (int e = 0; e < dl; ++e) s = fmaf(kbS[t * dl + e], bAQ[e], s);
        uq[((size_t)k * B + b) * Nq + r0 + t] = s;
    }
}

// slab of one (layer, sample): [dAQ dl*dl | dbAQ dl | dAK dl*dl | dbAK dl | dWH dl*D | dbWH dl]
__host__ __device__ inline size_t wp_slab_floats(int D, int dl) { return (size_t)2 * dl * dl + (size_t)dl * D + 3 * (size_t)dl; }

__global__ __launch_bounds__(256)
void word_prep_bwd_kernel(WordOuts G, const float* __restrict__ fw, const float* __restrict__ fs, const float* __restrict__ qmask,
                          const float* __restrict__ what, const float* __restrict__ kb, WordParams P, int B, int Nq, int D, int dl, int rpp,
                          float* __restrict__ dfw_part, float* __restrict__ dfs_part, float* __restrict__ slab)
{
    extern __shared__ __attribute__((aligned(16))) float wp_smem[];
    const int b = blockIdx.x, k = blockIdx.y, t = threadIdx.x;
    const int r0 = blockIdx.z * rpp, Nw = min(Nq, r0 + rpp) - r0;       // this part's words r0 .. r0 + Nw
    if (Nw <= 0) return;
    const float *WH = P.p[8 * k], *SH = P.p[8 * k + 2], *AK = P.p[8 * k + 4], *AQ = P.p[8 * k + 6], *bAQ = P.p[8 * k + 7];
    const int nd = Nw * dl, ndp = rpp * dl;
    float* fwS = wp_smem;                                     // [Nw][D]
    float* dMqS = fwS + rpp * D;                              // [Nw][dl]
    float* kbS = dMqS + ndp;
    float* whS = kbS + ndp;
    float* dkbS = whS + ndp;
    float* dpreS = dkbS + ndp;
    float* duS = dpreS + ndp;                                 // [32] duq, then [dl] dshat
    float* dshS = duS + 32;
    const size_t ob = (((size_t)k * B + b) * Nq + r0) * dl, ogb = ((size_t)b * Nq + r0) * dl;
    const float* dMq = G.dMq[k]; const float* duq = G.duq[k]; const float* dwh = G.dwhat[k]; const float* dsh = G.dshat[k];
    for (int idx = t * 4; idx < Nw * D; idx += 1024) stg4(fwS + idx, ldg4(fw + ((size_t)b * Nq + r0) * D + idx));
    for (int idx = t * 4; idx < nd; idx += 1024) {
        stg4(dMqS + idx, dMq ? ldg4(dMq + ogb + idx) : f4zero());
        stg4(kbS + idx, ldg4(kb + ob + idx));
        stg4(whS + idx, ldg4(what + ob + idx));
    }
    if (t < 32) duS[t] = (duq && t < Nw) ? duq[(size_t)b * Nq + r0 + t] : 0.f;
    for (int e = t; e < dl; e += 256) dshS[e] = dsh ? dsh[(size_t)b * dl + e] : 0.f;
    __syncthreads();
    // dkb = dMq AQ^T + duq (x) bAQ
    rows_matT(dMqS, dl, Nw, AQ, dl, dl, [&](int w, int e, float v) { dkbS[w * dl + e] = fmaf(duS[w], bAQ[e], v); });
    __syncthreads();
    // dwhat = dwhat_ext + dkb AK ; dpre = dwhat * qmask
    rows_mat(dkbS, dl, Nw, AK, dl, dl, [&](int w, int c, float v) {
        if (dwh) v += dwh[ogb + w * dl + c];
        dpreS[w * dl + c] = v * qmask[(size_t)b * Nq + r0 + w];
    });
    __syncthreads();
    // row-side gradients of this layer: dfw = dpre WH, dfs = dshat SH
    rows_mat(dpreS, dl, Nw, WH, dl, D, [&](int w, int d, float v) { dfw_part[(((size_t)k * B + b) * Nq + r0 + w) * D + d] = v; });
    if (blockIdx.z == 0) rows_mat(dshS, dl, 1, SH, dl, D, [&](int, int d, float v) { dfs_part[((size_t)k * B + b) * D + d] = v; });
    // partial weight gradients of this (sample, part)
    const int Nq_ = Nw;                                       // the sums below run over this part's words
    float* sl = slab + (((size_t)k * B + b) * gridDim.z + blockIdx.z) * wp_slab_floats(D, dl);
    float* sAQ = sl; float* sbAQ = sAQ + dl * dl; float* sAK = sbAQ + dl; float* sbAK = sAK + dl * dl; float* sWH = sbAK + dl; float* sbWH = sWH + (size_t)dl * D;
    for (int idx = t; idx < dl * dl; idx += 256) {
        const int e = idx / dl, c = idx % dl;
        float a = 0.f, a2 = 0.f;
        for (int w = 0; w < Nq_; ++w) { a = fmaf(kbS[w * dl + e], dMqS[w * dl + c], a); a2 = fmaf(dkbS[w * dl + e], whS[w * dl + c], a2); }
        sAQ[idx] = a; sAK[idx] = a2;
    }
    for (int e = t; e < dl; e += 256) {
        float a = 0.f, a2 = 0.f, a3 = 0.f;
        for (int w = 0; w < Nq_; ++w) { a = fmaf(kbS[w * dl + e], duS[w], a); a2 += dkbS[w * dl + e]; a3 += dpreS[w * dl + e]; }
        sbAQ[e] = a; sbAK[e] = a2; sbWH[e] = a3;
    }
    for (int idx = t * 4; idx < dl * D; idx += 1024) {
        const int e = idx / D, d = idx % D;
        float4 a = f4zero();
        for (int w = 0; w < Nq_; ++w) a = f4fma(ldg4(fwS + w * D + d), dpreS[w * dl + e], a);
        stg4(sWH + idx, a);
    }
}

// sums over the batch in sample order; grid (chunks, nl [+1 for the row-side sums])
__global__ __launch_bounds__(256)
void word_prep_reduce_kernel(const float* __restrict__ slab, const float* __restrict__ dfw_part, const float* __restrict__ dfs_part, WordOuts G,
                             const float* __restrict__ fs, WordGrads Q, int nl, int B, int Nq, int D, int dl, int parts, float* __restrict__ dfw,
                             float* __restrict__ dfs)
{
    const int k = blockIdx.y;
    const size_t x = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (k == nl) {                                            // d f_w, d f_s: sums over the layers
        const size_t nfw = (size_t)B * Nq * D, nfs = (size_t)B * D;
        if (x < nfw) { float s = 0.f; for (int l = 0; l < nl; ++l) s += dfw_part[(size_t)l * nfw + x]; dfw[x] = s; }
        else if (x < nfw + nfs) { const size_t y = x - nfw; float s = 0.f; for (int l = 0; l < nl; ++l) s += dfs_part[(size_t)l * nfs + y]; dfs[y] = s; }
        return;
    }
    const size_t ss = wp_slab_floats(D, dl);
    float* dWH = Q.p[8 * k]; float* dbWH = Q.p[8 * k + 1]; float* dSH = Q.p[8 * k + 2]; float* dbSH = Q.p[8 * k + 3]; float* dAK = Q.p[8 * k + 4];
    float* dbAK = Q.p[8 * k + 5]; float* dAQ = Q.p[8 * k + 6]; float* dbAQ = Q.p[8 * k + 7];
    if (x < ss) {
        // eight partial sums keep eight loads in flight (a single running sum is one HBM round trip per slab: with ~128 slabs per
        // layer that chain alone was ~350 us); the association is fixed, so the result is still bitwise reproducible
        const float* p = slab + (size_t)k * B * parts * ss + x;
        const int P = B * parts;
        float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int b = 0;
        for (; b + 8 <= P; b += 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] += p[(size_t)(b + u) * ss];
        }
        for (; b < P; ++b) a[0] += p[(size_t)b * ss];
        const float s = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
        const size_t o1 = (size_t)dl * dl, o2 = o1 + dl, o3 = o2 + o1, o4 = o3 + dl, o5 = o4 + (size_t)dl * D;
        if (x < o1) dAQ[x] = s; else if (x < o2) dbAQ[x - o1] = s; else if (x < o3) dAK[x - o2] = s; else if (x < o4) dbAK[x - o3] = s;
        else if (x < o5) dWH[x - o4] = s; else dbWH[x - o5] = s;
    } else if (x < ss + (size_t)dl * D) {                     // dSH[e][d] = sum_b dshat[b][e] fs[b][d]
        const size_t y = x - ss; const int e = (int)(y / D), d = (int)(y % D);
        const float* dsh = G.dshat[k];
        float s = 0.f;
        if (dsh) for (int b = 0; b < B; ++b) s = fmaf(dsh[(size_t)b * dl + e], fs[(size_t)b * D + d], s);
        dSH[y] = s;
    } else if (x < ss + (size_t)dl * D + dl) {
        const int e = (int)(x - ss - (size_t)dl * D);
        const float* dsh = G.dshat[k];
        float s = 0.f;
        if (dsh) for (int b = 0; b < B; ++b) s += dsh[(size_t)b * dl + e];
        dbSH[e] = s;
    }
}

}  // namespace smin

using namespace smin;

static size_t wp_fwd_lds(int rpp, int D, int dl) { return sizeof(float) * ((size_t)(rpp + 1) * D + 2 * (size_t)rpp * dl); }
static size_t wp_bwd_lds(int rpp, int D, int dl) { return sizeof(float) * ((size_t)rpp * D + 5 * (size_t)rpp * dl + 32 + dl); }
// rows per part: enough (sample, layer, part) workgroups to fill the chip, at least one word each
static int wp_rows_per_part(int nl, int B, int Nq)
{
    int parts = cdiv(384, nl * B);
    if (parts > Nq) parts = Nq;
    if (parts < 1) parts = 1;
    return cdiv(Nq, parts);
}

extern "C" int smin_word_prep_fwd(void* stream, const float* fw, const float* fs, const float* qmask, const float* const* params, int nl, int B, int Nq, int D,
                                  int dl, float* what, float* shat, float* kb, float* Mq, float* uq)
{
    if (B == 0) return 0;
    const int rpp = wp_rows_per_part(nl, B, Nq), parts = cdiv(Nq, rpp);
    SMIN_REQUIRE(nl >= 1 && nl <= WP_MAXL && Nq >= 1 && Nq <= 32 && D % 4 == 0 && dl % 4 == 0 && dl <= 128 && wp_fwd_lds(rpp, D, dl) <= 160 * 1024);
    WordParams P;
    for (int i = 0; i < WP_MAXL * 8; ++i) P.p[i] = params[i < nl * 8 ? i : 0];
    const size_t lds = wp_fwd_lds(rpp, D, dl);
    static size_t lds_set = 0;
    if (lds > 64 * 1024 && lds > lds_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&word_prep_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        lds_set = lds;
    }
    hipLaunchKernelGGL(word_prep_fwd_kernel, dim3(B, nl, parts), dim3(256), lds, (hipStream_t)stream, fw, fs, qmask, P, B, Nq, D, dl, rpp, what, shat, kb, Mq, uq);
    SMIN_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t smin_word_prep_bwd_workspace_bytes(int nl, int B, int Nq, int D, int dl)
{
    const int parts = B > 0 ? cdiv(Nq, wp_rows_per_part(nl, B, Nq)) : 1;
    return sizeof(float) * ((size_t)nl * B * parts * wp_slab_floats(D, dl) + (size_t)nl * B * Nq * D + (size_t)nl * B * D + 64);
}

extern "C" int smin_word_prep_bwd(void* stream, const float* const* dwhat, const float* const* dshat, const float* const* dMq, const float* const* duq,
                                  const float* fw, const float* fs, const float* qmask, const float* what, const float* kb, const float* const* params,
                                  int nl, int B, int Nq, int D, int dl, float* dfw, float* dfs, float* const* dparams, void* ws, size_t ws_bytes)
{
    hipStream_t st = (hipStream_t)stream;
    if (B == 0) return 0;
    const int rpp = wp_rows_per_part(nl, B, Nq), parts = cdiv(Nq, rpp);
    SMIN_REQUIRE(nl >= 1 && nl <= WP_MAXL && Nq >= 1 && Nq <= 32 && D % 4 == 0 && dl % 4 == 0 && dl <= 128 && wp_bwd_lds(rpp, D, dl) <= 160 * 1024);
    SMIN_REQUIRE(ws_bytes >= smin_word_prep_bwd_workspace_bytes(nl, B, Nq, D, dl));
    WordParams P; WordGrads Q; WordOuts G;
    for (int i = 0; i < WP_MAXL * 8; ++i) { P.p[i] = params[i < nl * 8 ? i : 0]; Q.p[i] = dparams[i < nl * 8 ? i : 0]; }
    for (int k = 0; k < WP_MAXL; ++k) {
        G.dwhat[k] = k < nl ? dwhat[k] : nullptr; G.dshat[k] = k < nl ? dshat[k] : nullptr; G.dMq[k] = k < nl ? dMq[k] : nullptr; G.duq[k] = k < nl ? duq[k] : nullptr;
    }
    float* slab = reinterpret_cast<float*>(ws);
    float* dfw_part = slab + (size_t)nl * B * parts * wp_slab_floats(D, dl);
    float* dfs_part = dfw_part + (size_t)nl * B * Nq * D;
    const size_t lds = wp_bwd_lds(rpp, D, dl);
    static size_t lds_set = 0;
    if (lds > 64 * 1024 && lds > lds_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&word_prep_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        lds_set = lds;
    }
    hipLaunchKernelGGL(word_prep_bwd_kernel, dim3(B, nl, parts), dim3(256), lds, st, G, fw, fs, qmask, what, kb, P, B, Nq, D, dl, rpp, dfw_part, dfs_part, slab);
    SMIN_LAUNCH_CHECK();
    const size_t per_layer = wp_slab_floats(D, dl) + (size_t)dl * D + dl, rows = (size_t)B * Nq * D + (size_t)B * D;
    const size_t mx = per_layer > rows ? per_layer : rows;
    hipLaunchKernelGGL(word_prep_reduce_kernel, dim3((unsigned)((mx + 255) / 256), nl + 1), dim3(256), 0, st, slab, dfw_part, dfs_part, G, fs, Q, nl, B, Nq, D, dl, parts,
                       dfw, dfs);
    SMIN_LAUNCH_CHECK();
    return 0;
}
