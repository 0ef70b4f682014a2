// ContentAttention core on the matrix cores (reference models.py:207-226 and 253-266), fp32 MFMA 16x16x4.
//
// One wave owns a tile of 4 cells = 16 "rows" (row n = 4*cell + clip; clips >= C are zero padding).  Rows live on
// the MFMA column index (lane & 15); lane (n, kg = lane >> 4) holds the features d = 16*j + 4*kg + q of row n, so
// every per-row vector (chat, a, q, gradients) is DL/4 registers in exactly the accumulator layout:
//   S^T[w][n]  = sum_d Mq[w][d] chat[n][d]          MFMA  A = Mq (LDS, rows = words), B = chat (registers)
//   P          = softmax over words                  in-register: 8 values per lane + two cross-lane exchanges
//   a^T[d][n]  = sum_w what[w][d] P^T[w][n]         MFMA  A = what^T (LDS), B = the P accumulator as is
//   q, Z = q q^T over the 4 clips of a cell, A = softmax(Z), cchat = A chat        VALU + DPP quad permutes
// The backward pass mirrors it (dP^T = what . da^T and dchat^T += Mq^T . dS^T on MFMA) and streams da, dS and P
// to HBM; the per-sample word-side reductions (dMq, dwhat, dshat, duq) are a second, MFMA "TN" kernel that reads
// them back coalesced -- fixed-order partial slabs keep everything deterministic.
// (A first version used 32-row tiles on the 32x32x2 MFMA: its backward needed 426 registers, one wave per SIMD.)
#include "content_attn.h"
#include "smin_hip.h"

namespace smin {

constexpr int LDW = 36;                                       // row stride of the transposed [d][w] LDS images

template <int CTRL>
__device__ __forceinline__ float qperm(float v) {             // quad permute (DPP): neighbour lane j ^ o
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
template <int O> __device__ __forceinline__ float nb(float v);
template <> __device__ __forceinline__ float nb<0>(float v) { return v; }
template <> __device__ __forceinline__ float nb<1>(float v) { return qperm<0xB1>(v); }    // [1,0,3,2]
template <> __device__ __forceinline__ float nb<2>(float v) { return qperm<0x4E>(v); }    // [2,3,0,1]
template <> __device__ __forceinline__ float nb<3>(float v) { return qperm<0x1B>(v); }    // [3,2,1,0]

// acc_o += nb<o>(x) * y_o for o = 1, 2, 3 as three fused DPP FMAs.  Written as one asm statement because hipcc
// otherwise materialises all neighbour values of a whole tile first (hundreds of live VGPRs, scratch spills).
// The leading s_nop 1 covers the "VALU write -> DPP read" hazard for x (asm is not padded by the compiler).
__device__ __forceinline__ void fmac_nb3(float& a1, float& a2, float& a3, float x, float y1, float y2, float y3) {
    asm volatile("s_nop 1\n\t"
                 "v_fmac_f32_dpp %0, %3, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                 "v_fmac_f32_dpp %1, %3, %5 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                 "v_fmac_f32_dpp %2, %3, %6 quad_perm:[3,2,1,0] row_mask:0xf bank_mask:0xf bound_ctrl:1"
                 : "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "v"(y1), "v"(y2), "v"(y3));
}
// acc += sum_{o=1..3} nb<o>(x) * y_o
__device__ __forceinline__ void fmac_nb_sum(float& acc, float x, float y1, float y2, float y3) {
    asm volatile("s_nop 1\n\t"
                 "v_fmac_f32_dpp %0, %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                 "v_fmac_f32_dpp %0, %1, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                 "v_fmac_f32_dpp %0, %1, %4 quad_perm:[3,2,1,0] row_mask:0xf bank_mask:0xf bound_ctrl:1"
                 : "+v"(acc) : "v"(x), "v"(y1), "v"(y2), "v"(y3));
}

__device__ __forceinline__ int wmap(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// LDS carve (floats).  sM/sW: [32][DL+4] word-major; sMT/sWT: [DLP][36] feature-major (DLP = DL rounded to 32).
template <int DL>
struct AttnLds {
    static constexpr int LDM = DL + 4;
    static constexpr int DLP = (DL + 31) / 32 * 32;
    float *sM, *sW, *sMT, *sWT, *sS, *sU, *sQ;
    __device__ AttnLds(float* base, bool bwd) {
        sM = base; sWT = sM + 32 * LDM; sS = sWT + DLP * LDW; sU = sS + DL; sQ = sU + 32;
        sW = sQ + 32; sMT = sW + (bwd ? 32 * LDM : 0);
    }
    static size_t bytes(bool bwd) { return sizeof(float) * (size_t)(32 * LDM + DLP * LDW + DL + 64 + (bwd ? 32 * LDM + DLP * LDW : 0)); }
};

template <int DL>
__device__ __forceinline__ void stage_sample(const AttnLds<DL>& s, bool bwd, const float* Mq, const float* uq, const float* what,
                                             const float* shat, const float* qmask, int b, int dl, int Nq)
{
    constexpr int LDM = AttnLds<DL>::LDM, DLP = AttnLds<DL>::DLP;
    const int t = threadIdx.x;
    for (int idx = t; idx < 32 * LDM; idx += 256) {
        const int w = idx / LDM, d = idx % LDM;
        const bool ok = w < Nq && d < dl;
        s.sM[idx] = ok ? Mq[((size_t)b * Nq + w) * dl + d] : 0.f;
        if (bwd) s.sW[idx] = ok ? what[((size_t)b * Nq + w) * dl + d] : 0.f;
    }
    for (int idx = t; idx < DLP * LDW; idx += 256) {
        const int d = idx / LDW, w = idx % LDW;
        const bool ok = w < Nq && d < dl;
        s.sWT[idx] = ok ? what[((size_t)b * Nq + w) * dl + d] : 0.f;
        if (bwd) s.sMT[idx] = ok ? Mq[((size_t)b * Nq + w) * dl + d] : 0.f;
    }
    for (int d = t; d < DL; d += 256) s.sS[d] = d < dl ? shat[(size_t)b * dl + d] : 0.f;
    if (t < 32) {
        s.sU[t] = t < Nq ? uq[(size_t)b * Nq + t] : 0.f;
        s.sQ[t] = t < Nq ? qmask[(size_t)b * Nq + t] : 0.f;
    }
}

// per-lane geometry of a tile
struct RowGeom {
    int row;            // global row index n*C + c (clamped to a valid row when !ok)
    bool ok;            // this lane carries a real (cell, clip)
    float m;            // cell mask
    bool nbok[4];       // neighbour clip c ^ o exists
};
// ---- 16-row tiles on v_mfma_f32_16x16x4_f32 ---------------------------------------------------------------------
// Accumulator layout of the 16x16x4 MFMA: register r of lane (n, kg) is element [4*kg + r][n], so
//   S^T block b (words 16b .. 16b+15): lane holds words 16b + 4kg + r           -> P[4b + r]
//   a^T block j (features 16j .. 16j+15): lane holds features 16j + 4kg + r      -> aligned with ch[j][r]
// and the P / dS accumulators feed the next MFMA as B operands without any data movement.  The backward fits 204
// registers -> two waves per SIMD; a row's 16-byte loads of four neighbouring lanes form 64-byte segments.
typedef float f32x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4v mfma16(float a, float b, f32x4v c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ float kg_sum(float v) { v += __shfl_xor(v, 16); return v + __shfl_xor(v, 32); }
__device__ __forceinline__ float kg_max(float v) { v = fmaxf(v, __shfl_xor(v, 16)); return fmaxf(v, __shfl_xor(v, 32)); }

__device__ __forceinline__ RowGeom row_geom16(const int* cells, int n0, int n_end, int C, int lane) {
    const int j = lane & 15, cell = n0 + (j >> 2), c = j & 3;
    RowGeom g;
    g.ok = cell < n_end && c < C;
    const int cc = g.ok ? cell : n0;
    g.row = cc * C + (g.ok ? c : 0);
    g.m = g.ok ? (float)cells[4 * (size_t)cc + 3] : 0.f;
#pragma unroll
    for (int o = 0; o < 4; ++o) g.nbok[o] = (c ^ o) < C;
    return g;
}

template <int DL>
__device__ __forceinline__ void load_rows16(float (&v)[DL / 16][4], const float* src, const RowGeom& g, int dl, int kg) {
#pragma unroll
    for (int j = 0; j < DL / 16; ++j) {
        const int d = 16 * j + 4 * kg;
        const float4 x = ldg4(src + (size_t)g.row * dl + min(d, dl - 4));
        const bool ok = g.ok && d < dl;
        v[j][0] = ok ? x.x : 0.f; v[j][1] = ok ? x.y : 0.f; v[j][2] = ok ? x.z : 0.f; v[j][3] = ok ? x.w : 0.f;
    }
}

// P[4b + r] = softmax over words of row n, word 16b + 4kg + r
template <int DL>
__device__ __forceinline__ void scores_softmax16(float (&P)[8], const float (&ch)[DL / 16][4], const AttnLds<DL>& s, int Nq, float scale, int lane)
{
    constexpr int LDM = AttnLds<DL>::LDM;
    const int l15 = lane & 15, kg = lane >> 4;
    f32x4v S0 = {0.f, 0.f, 0.f, 0.f}, S1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < DL / 16; ++j) {
        const float4 a0 = ldg4(s.sM + l15 * LDM + 16 * j + 4 * kg);
        S0 = mfma16(a0.x, ch[j][0], S0); S0 = mfma16(a0.y, ch[j][1], S0); S0 = mfma16(a0.z, ch[j][2], S0); S0 = mfma16(a0.w, ch[j][3], S0);
        if (Nq > 16) {
            const float4 a1 = ldg4(s.sM + (16 + l15) * LDM + 16 * j + 4 * kg);
            S1 = mfma16(a1.x, ch[j][0], S1); S1 = mfma16(a1.y, ch[j][1], S1); S1 = mfma16(a1.z, ch[j][2], S1); S1 = mfma16(a1.w, ch[j][3], S1);
        }
        if ((j & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int w = 16 * (r >> 2) + 4 * kg + (r & 3);
        float v = ((r < 4 ? S0[r & 3] : S1[r & 3]) + s.sU[w]) * scale;
        const float qm = s.sQ[w];
        v = (qm == 0.f) ? -1e9f : v * qm;                         // models.py:216-218
        v = (w < Nq) ? v : -INFINITY;
        P[r] = v;
        mx = fmaxf(mx, v);
    }
    mx = kg_max(mx);
    float den = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) { P[r] = expf(P[r] - mx); den += P[r]; }
    den = kg_sum(den);
    const float inv = 1.0f / den;
#pragma unroll
    for (int r = 0; r < 8; ++r) P[r] *= inv;
}

// a^T block j: acc[r] = a[n][16j + 4kg + r] = sum_w what[w][16j + 4kg + r] P[n][w]
template <int DL>
__device__ __forceinline__ f32x4v attend_tile16(int j, const float (&P)[8], const AttnLds<DL>& s, int Nq, int lane)
{
    const int l15 = lane & 15, kg = lane >> 4;
    f32x4v acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        if (16 * b < Nq) {
            const float4 w4 = ldg4(s.sWT + (16 * j + l15) * LDW + 16 * b + 4 * kg);
            acc = mfma16(w4.x, P[4 * b], acc); acc = mfma16(w4.y, P[4 * b + 1], acc);
            acc = mfma16(w4.z, P[4 * b + 2], acc); acc = mfma16(w4.w, P[4 * b + 3], acc);
        }
    }
    return acc;
}

template <int DL>
__device__ __forceinline__ void clip_attention16(float (&Ao)[4], const float (&ch)[DL / 16][4], const float (&P)[8],
                                                 const AttnLds<DL>& s, const RowGeom& g, int Nq, float scale, int lane)
{
    const int kg = lane >> 4;
    float z[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < DL / 16; ++j) {
        const f32x4v acc = attend_tile16<DL>(j, P, s, Nq, lane);
        const float4 sh = ldg4(s.sS + 16 * j + 4 * kg);
        const float shv[4] = {sh.x, sh.y, sh.z, sh.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float qv = ch[j][q] * (acc[q] + shv[q]);
            z[0] = fmaf(qv, qv, z[0]);
            fmac_nb3(z[1], z[2], z[3], qv, qv, qv, qv);
        }
        if ((j & 1) == 1) __builtin_amdgcn_sched_barrier(0);
    }
    float mx = -INFINITY;
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        z[o] = kg_sum(z[o]) * scale;
        z[o] = g.nbok[o] ? z[o] : -INFINITY;
        mx = fmaxf(mx, z[o]);
    }
    float den = 0.f;
#pragma unroll
    for (int o = 0; o < 4; ++o) { Ao[o] = expf(z[o] - mx); den += Ao[o]; }
    const float inv = g.m / den;                                  // models.py:262-263: softmax, then * mask
#pragma unroll
    for (int o = 0; o < 4; ++o) Ao[o] *= inv;
}

template <int DL>
__global__ __launch_bounds__(256, 3)
void content_attn_fwd16_kernel(const float* __restrict__ chat, const int* __restrict__ cells, const int* __restrict__ row_ptr, int L, int C,
                               const float* __restrict__ Mq, const float* __restrict__ uq, const float* __restrict__ what,
                               const float* __restrict__ shat, const float* __restrict__ qmask,
                               float* __restrict__ cchat, float* __restrict__ ccmean, int dl, int Nq, int cells_per_chunk, float scale)
{
    extern __shared__ __attribute__((aligned(16))) float smem_dyn[];
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int s0 = row_ptr[b * L], s1 = row_ptr[(b + 1) * L];
    const int n_begin = s0 + chunk * cells_per_chunk;
    if (n_begin >= s1) return;
    const int n_end = min(s1, n_begin + cells_per_chunk);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, kg = lane >> 4;
    const float invC = 1.0f / C;
    AttnLds<DL> s(smem_dyn, false);
    stage_sample<DL>(s, false, Mq, uq, what, shat, qmask, b, dl, Nq);
    __syncthreads();

    for (int n0 = n_begin + 4 * wave; n0 < n_end; n0 += 16) {
        const RowGeom g = row_geom16(cells, n0, n_end, C, lane);
        float ch[DL / 16][4], P[8], Ao[4];
        load_rows16<DL>(ch, chat, g, dl, kg);
        scores_softmax16<DL>(P, ch, s, Nq, scale, lane);
        __builtin_amdgcn_sched_barrier(0);
        clip_attention16<DL>(Ao, ch, P, s, g, Nq, scale, lane);
#pragma unroll
        for (int j = 0; j < DL / 16; ++j) {                         // cchat = A chat
            float o4[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float x = ch[j][q];
                o4[q] = Ao[0] * x;
                fmac_nb_sum(o4[q], x, Ao[1], Ao[2], Ao[3]);
            }
            const int d = 16 * j + 4 * kg;
            if (cchat) {
                if (g.ok && d < dl) stg4(cchat + (size_t)g.row * dl + d, make_float4(o4[0], o4[1], o4[2], o4[3]));
            }
            if (ccmean) {                                           // mean over the clips of the quad (padding lanes hold 0)
#pragma unroll
                for (int q = 0; q < 4; ++q) o4[q] = (o4[q] + nb<1>(o4[q]) + nb<2>(o4[q]) + nb<3>(o4[q])) * invC;
                if (g.ok && (lane & 3) == 0 && d < dl)
                    stg4(ccmean + (size_t)(g.row / C) * dl + d, make_float4(o4[0], o4[1], o4[2], o4[3]));
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int DL>
__global__ __launch_bounds__(256, 2)
void content_attn_bwd16_kernel(const float* __restrict__ chat, const float* __restrict__ dcchat,
                               const int* __restrict__ cells, const int* __restrict__ row_ptr, int L, int C,
                               const float* __restrict__ Mq, const float* __restrict__ uq, const float* __restrict__ what,
                               const float* __restrict__ shat, const float* __restrict__ qmask,
                               float* __restrict__ dchat, float* __restrict__ da_out, float* __restrict__ ds_out, float* __restrict__ p_out,
                               int dl, int Nq, int cells_per_chunk, float scale, int g_per_cell, float gscale,
                               const float* __restrict__ dmean2, float mscale)
{
    extern __shared__ __attribute__((aligned(16))) float smem_dyn[];
    constexpr int LDM = AttnLds<DL>::LDM, KJ = DL / 16;
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int s0 = row_ptr[b * L], s1 = row_ptr[(b + 1) * L];
    const int n_begin = s0 + chunk * cells_per_chunk;
    if (n_begin >= s1) return;
    const int n_end = min(s1, n_begin + cells_per_chunk);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, kg = lane >> 4, l15 = lane & 15;
    AttnLds<DL> s(smem_dyn, true);
    stage_sample<DL>(s, true, Mq, uq, what, shat, qmask, b, dl, Nq);
    __syncthreads();

    for (int n0 = n_begin + 4 * wave; n0 < n_end; n0 += 16) {
        const RowGeom g = row_geom16(cells, n0, n_end, C, lane);
        float ch[KJ][4], P[8], Ao[4];
        load_rows16<DL>(ch, chat, g, dl, kg);
        float4 gq[KJ];                                              // gradient rows: requested now, consumed after the recompute
#pragma unroll
        for (int j = 0; j < KJ; ++j)
            gq[j] = ldg4(dcchat + (size_t)(g_per_cell ? g.row / C : g.row) * dl + min(16 * j + 4 * kg, dl - 4));
        float4 gm[KJ];                                              // second consumer's gradient (per cell), when there is one
        if (dmean2) {
#pragma unroll
            for (int j = 0; j < KJ; ++j) gm[j] = ldg4(dmean2 + (size_t)(g.row / C) * dl + min(16 * j + 4 * kg, dl - 4));
        }
        scores_softmax16<DL>(P, ch, s, Nq, scale, lane);
        __builtin_amdgcn_sched_barrier(0);
        clip_attention16<DL>(Ao, ch, P, s, g, Nq, scale, lane);

        // cchat = A chat :  dA[c][c^o] = <g_c, chat_{c^o}> ,  dchat_c = sum_o A[c^o][c] g_{c^o}
        float dch[KJ][4];
        float dAo[4] = {0.f, 0.f, 0.f, 0.f};
        const float An1 = nb<1>(Ao[1]), An2 = nb<2>(Ao[2]), An3 = nb<3>(Ao[3]);
#pragma unroll
        for (int j = 0; j < KJ; ++j) {
            const int d = 16 * j + 4 * kg;
            const bool dok = g.ok && d < dl;
            const float gs = dok ? gscale : 0.f;
            float gv[4] = {gq[j].x * gs, gq[j].y * gs, gq[j].z * gs, gq[j].w * gs};
            if (dmean2) {
                const float ms = dok ? mscale : 0.f;
                gv[0] = fmaf(gm[j].x, ms, gv[0]); gv[1] = fmaf(gm[j].y, ms, gv[1]); gv[2] = fmaf(gm[j].z, ms, gv[2]); gv[3] = fmaf(gm[j].w, ms, gv[3]);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float x = ch[j][q], y = gv[q];
                dAo[0] = fmaf(y, x, dAo[0]);
                fmac_nb3(dAo[1], dAo[2], dAo[3], x, y, y, y);
                dch[j][q] = Ao[0] * y;
                fmac_nb_sum(dch[j][q], y, An1, An2, An3);
            }
            if ((j & 1) == 1) __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_sched_barrier(0);
        // A = softmax(Z) * m ; Z symmetric in (c, c')
        float sym[4];
        {
            float rd = 0.f, dZ[4];
#pragma unroll
            for (int o = 0; o < 4; ++o) { dAo[o] = kg_sum(dAo[o]); rd = fmaf(Ao[o], dAo[o], rd); }
#pragma unroll
            for (int o = 0; o < 4; ++o) dZ[o] = g.nbok[o] ? Ao[o] * (dAo[o] - rd) : 0.f;
            sym[0] = 2.0f * dZ[0] * scale;
            sym[1] = (dZ[1] + nb<1>(dZ[1])) * scale;
            sym[2] = (dZ[2] + nb<2>(dZ[2])) * scale;
            sym[3] = (dZ[3] + nb<3>(dZ[3])) * scale;
        }
        // per 16-feature block: a (MFMA) -> q = chat*(a+shat) -> dq = sum_o sym[o] q_{c^o} -> dchat += dq (a+shat), da = dq chat
        //                       -> da to HBM and straight into  dP^T[w][n] += sum_d what[w][d] da^T[d][n]  (MFMA)
        f32x4v dP0 = {0.f, 0.f, 0.f, 0.f}, dP1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < KJ; ++j) {
            const f32x4v acc = attend_tile16<DL>(j, P, s, Nq, lane);
            const float4 sh = ldg4(s.sS + 16 * j + 4 * kg);
            const float shv[4] = {sh.x, sh.y, sh.z, sh.w};
            float da4[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float tq = acc[q] + shv[q];
                const float qv = ch[j][q] * tq;
                float dq = sym[0] * qv;
                fmac_nb_sum(dq, qv, sym[1], sym[2], sym[3]);
                dch[j][q] = fmaf(dq, tq, dch[j][q]);
                da4[q] = dq * ch[j][q];
            }
            const int d = 16 * j + 4 * kg;
            if (g.ok && d < dl) stg4(da_out + (size_t)g.row * dl + d, make_float4(da4[0], da4[1], da4[2], da4[3]));
            const float4 w0 = ldg4(s.sW + l15 * LDM + d);
            dP0 = mfma16(w0.x, da4[0], dP0); dP0 = mfma16(w0.y, da4[1], dP0); dP0 = mfma16(w0.z, da4[2], dP0); dP0 = mfma16(w0.w, da4[3], dP0);
            if (Nq > 16) {
                const float4 w1 = ldg4(s.sW + (16 + l15) * LDM + d);
                dP1 = mfma16(w1.x, da4[0], dP1); dP1 = mfma16(w1.y, da4[1], dP1); dP1 = mfma16(w1.z, da4[2], dP1); dP1 = mfma16(w1.w, da4[3], dP1);
            }
            if ((j & 1) == 1) __builtin_amdgcn_sched_barrier(0);
        }
        // P = softmax(S), S = (raw + u) * scale * qmask
        float pd = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) pd = fmaf(P[r], (r < 4 ? dP0[r & 3] : dP1[r & 3]), pd);
        pd = kg_sum(pd);
        float dS[8];
#pragma unroll
        for (int r = 0; r < 8; ++r)
            dS[r] = P[r] * ((r < 4 ? dP0[r & 3] : dP1[r & 3]) - pd) * s.sQ[16 * (r >> 2) + 4 * kg + (r & 3)] * scale;
        if (g.ok) {
#pragma unroll
            for (int b2 = 0; b2 < 2; ++b2) {
                stg4(ds_out + (size_t)g.row * 32 + 16 * b2 + 4 * kg, make_float4(dS[4 * b2], dS[4 * b2 + 1], dS[4 * b2 + 2], dS[4 * b2 + 3]));
                stg4(p_out + (size_t)g.row * 32 + 16 * b2 + 4 * kg, make_float4(P[4 * b2], P[4 * b2 + 1], P[4 * b2 + 2], P[4 * b2 + 3]));
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // raw = chat Mq^T :  dchat^T[d][n] += sum_w Mq[w][d] dS^T[w][n]   -> dchat = (...) * m   (chat = linear(fc) * m)
#pragma unroll
        for (int j = 0; j < KJ; ++j) {
            f32x4v acc = {dch[j][0], dch[j][1], dch[j][2], dch[j][3]};
#pragma unroll
            for (int b2 = 0; b2 < 2; ++b2) {
                if (16 * b2 < Nq) {
                    const float4 m4 = ldg4(s.sMT + (16 * j + l15) * LDW + 16 * b2 + 4 * kg);
                    acc = mfma16(m4.x, dS[4 * b2], acc); acc = mfma16(m4.y, dS[4 * b2 + 1], acc);
                    acc = mfma16(m4.z, dS[4 * b2 + 2], acc); acc = mfma16(m4.w, dS[4 * b2 + 3], acc);
                }
            }
            const int d = 16 * j + 4 * kg;
            if (g.ok && d < dl) stg4(dchat + (size_t)g.row * dl + d, make_float4(acc[0] * g.m, acc[1] * g.m, acc[2] * g.m, acc[3] * g.m));
            if ((j & 1) == 1) __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// ---- per-sample word-side reductions on MFMA (operands straight from HBM, coalesced) ---------------------------
//   dMq[w][d]  = sum_rows dS[row][w] chat[row][d]      dwhat[w][d] = sum_rows P[row][w] da[row][d]
//   dshat[d]   = sum_rows da[row][d]                   duq[w]      = sum_rows dS[row][w]
// grid (SPLITS, B); each wave reduces a contiguous row range and writes one partial slab
//   [dM 32 x dl | dW 32 x dl | dshat dl | du 32]; content_attn_reduce_kernel sums the slabs in fixed order.
template <int DL>
__global__ __launch_bounds__(256, 2)
void content_attn_wordgrad_kernel(const float* __restrict__ chat, const float* __restrict__ da, const float* __restrict__ dS, const float* __restrict__ P,
                                  const int* __restrict__ row_ptr, int L, int C, int dl, int splits, float* __restrict__ slab)
{
    constexpr int DT = (DL + 31) / 32;
    const int b = blockIdx.y, sp = blockIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, h = lane >> 5, l31 = lane & 31;
    const int r0 = row_ptr[b * L] * C, r1 = row_ptr[(b + 1) * L] * C;
    const int parts = splits * 4, part = sp * 4 + wave;
    const int per = ((r1 - r0 + parts - 1) / parts + 1) & ~1;            // even number of rows per wave
    const int rb = r0 + part * per, re = min(r1, rb + per);
    f32x16 aM[DT], aW[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { aM[dt][r] = 0.f; aW[dt][r] = 0.f; }
    float sh[DT], du = 0.f;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) sh[dt] = 0.f;
    // two row pairs per iteration: all loads of an iteration are issued before the first MFMA consumes them
    for (int row2 = rb; row2 < re; row2 += 4) {                  // wave-uniform trip count: MFMA needs every lane live
        float gs[2], ps[2], cv[2][DT], dv[2][DT];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const bool ok = row2 + 2 * u + h < re;
            const size_t row = (size_t)min(row2 + 2 * u + h, re - 1);
            gs[u] = ok ? dS[row * 32 + l31] : 0.f;
            ps[u] = ok ? P[row * 32 + l31] : 0.f;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const int d = min(32 * dt + l31, dl - 1);
                const bool dok = ok && 32 * dt + l31 < dl;
                cv[u][dt] = dok ? chat[row * dl + d] : 0.f;
                dv[u][dt] = dok ? da[row * dl + d] : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            du += gs[u];
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                sh[dt] += dv[u][dt];
                aM[dt] = mfma32(gs[u], cv[u][dt], aM[dt]);
                aW[dt] = mfma32(ps[u], dv[u][dt], aW[dt]);
            }
        }
    }
    const size_t slab_sz = (size_t)2 * 32 * dl + dl + 32;
    float* sl = slab + ((size_t)b * parts + part) * slab_sz;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int w = wmap(r, h), d = 32 * dt + l31;
            if (d < dl) { sl[(size_t)w * dl + d] = aM[dt][r]; sl[(size_t)(32 + w) * dl + d] = aW[dt][r]; }
        }
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
        const float v = sh[dt] + __shfl_xor(sh[dt], 32);
        const int d = 32 * dt + l31;
        if (h == 0 && d < dl) sl[(size_t)64 * dl + d] = v;
    }
    du += __shfl_xor(du, 32);
    if (h == 0) sl[(size_t)64 * dl + dl + l31] = du;
}

__global__ void content_attn_reduce_kernel(const float* __restrict__ slab, int dl, int Nq, int parts,
                                           float* __restrict__ dMq, float* __restrict__ dwhat, float* __restrict__ dshat, float* __restrict__ duq)
{
    const int b = blockIdx.y;
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int slab_sz = 2 * 32 * dl + dl + 32;
    if (x >= slab_sz) return;
    const float* p = slab + (size_t)b * parts * slab_sz + x;
    float s = 0.f;
    for (int k = 0; k < parts; ++k) s += p[(size_t)k * slab_sz];
    if (x < 32 * dl) { const int w = x / dl, d = x % dl; if (w < Nq) dMq[((size_t)b * Nq + w) * dl + d] = s; }
    else if (x < 64 * dl) { const int y = x - 32 * dl, w = y / dl, d = y % dl; if (w < Nq) dwhat[((size_t)b * Nq + w) * dl + d] = s; }
    else if (x < 64 * dl + dl) dshat[(size_t)b * dl + (x - 64 * dl)] = s;
    else { const int w = x - 64 * dl - dl; if (w < Nq) duq[(size_t)b * Nq + w] = s; }
}

// ---- launchers --------------------------------------------------------------------------------------------------
template <int DL>
static int fwd_t(hipStream_t st, const float* chat, const int* cells, const int* row_ptr, int B, int L, int C,
                 const float* Mq, const float* uq, const float* what, const float* shat, const float* qmask,
                 float* cc_rows, float* cc_mean, int dl, int Nq)
{
    int cpc, mc; chunking(L, &cpc, &mc);
    hipLaunchKernelGGL((content_attn_fwd16_kernel<DL>), dim3(mc, B), dim3(256), AttnLds<DL>::bytes(false), st, chat, cells, row_ptr, L, C,
                       Mq, uq, what, shat, qmask, cc_rows, cc_mean, dl, Nq, cpc, 1.0f / sqrtf((float)dl));
    SMIN_LAUNCH_CHECK();
    return 0;
}

int launch_content_attn_fwd(hipStream_t st, const float* chat, const int* cells, const int* row_ptr, int B, int L, int C,
                            const float* Mq, const float* uq, const float* what, const float* shat, const float* qmask,
                            float* cc_rows, float* cc_mean, int dl, int Nq)
{
    if (dl <= 16) return fwd_t<16>(st, chat, cells, row_ptr, B, L, C, Mq, uq, what, shat, qmask, cc_rows, cc_mean, dl, Nq);
    if (dl <= 32) return fwd_t<32>(st, chat, cells, row_ptr, B, L, C, Mq, uq, what, shat, qmask, cc_rows, cc_mean, dl, Nq);
    if (dl <= 64) return fwd_t<64>(st, chat, cells, row_ptr, B, L, C, Mq, uq, what, shat, qmask, cc_rows, cc_mean, dl, Nq);
    return fwd_t<128>(st, chat, cells, row_ptr, B, L, C, Mq, uq, what, shat, qmask, cc_rows, cc_mean, dl, Nq);
}

size_t content_attn_bwd_ws_floats(int M, int B, int dl)
{
    return (size_t)M * dl + 2 * (size_t)M * 32 + (size_t)B * ATTN_SPLITS * 4 * ((size_t)64 * dl + dl + 32) + 64;
}

template <int DL>
static int bwd_t(hipStream_t st, const float* chat, const float* dcchat, const int* cells, const int* row_ptr, int M, int B, int L, int C,
                 const float* Mq, const float* uq, const float* what, const float* shat, const float* qmask,
                 float* dchat, float* dMq, float* duq, float* dwhat, float* dshat, float* ws, int dl, int Nq, int g_per_cell, float gscale,
                 const float* dmean2, float mscale)
{
    int cpc, mc; chunking(L, &cpc, &mc);
    float* da = ws;
    float* dS = da + (size_t)M * dl;
    float* P = dS + (size_t)M * 32;
    float* slab = P + (size_t)M * 32;
    hipLaunchKernelGGL((content_attn_bwd16_kernel<DL>), dim3(mc, B), dim3(256), AttnLds<DL>::bytes(true), st, chat, dcchat, cells, row_ptr, L, C,
                       Mq, uq, what, shat, qmask, dchat, da, dS, P, dl, Nq, cpc, 1.0f / sqrtf((float)dl), g_per_cell, gscale, dmean2, mscale);
    SMIN_LAUNCH_CHECK();
    hipLaunchKernelGGL((content_attn_wordgrad_kernel<DL>), dim3(ATTN_SPLITS, B), dim3(256), 0, st, chat, da, dS, P, row_ptr, L, C, dl, ATTN_SPLITS, slab);
    SMIN_LAUNCH_CHECK();
    const int slab_sz = 64 * dl + dl + 32;
    hipLaunchKernelGGL(content_attn_reduce_kernel, dim3(cdiv(slab_sz, 256), B), dim3(256), 0, st, slab, dl, Nq, ATTN_SPLITS * 4, dMq, dwhat, dshat, duq);
    SMIN_LAUNCH_CHECK();
    return 0;
}

int launch_content_attn_bwd(hipStream_t st, const float* chat, const float* dcchat, const int* cells, const int* row_ptr, int M, int B, int L, int C,
                            const float* Mq, const float* uq, const float* what, const float* shat, const float* qmask,
                            float* dchat, float* dMq, float* duq, float* dwhat, float* dshat, float* ws, int dl, int Nq, int g_per_cell, float gscale,
                            const float* dmean2, float mscale)
{
    if (dl <= 16) return bwd_t<16>(st, chat, dcchat, cells, row_ptr, M, B, L, C, Mq, uq, what, shat, qmask, dchat, dMq, duq, dwhat, dshat, ws, dl, Nq, g_per_cell, gscale, dmean2, mscale);
    if (dl <= 32) return bwd_t<32>(st, chat, dcchat, cells, row_ptr, M, B, L, C, Mq, uq, what, shat, qmask, dchat, dMq, duq, dwhat, dshat, ws, dl, Nq, g_per_cell, gscale, dmean2, mscale);
    if (dl <= 64) return bwd_t<64>(st, chat, dcchat, cells, row_ptr, M, B, L, C, Mq, uq, what, shat, qmask, dchat, dMq, duq, dwhat, dshat, ws, dl, Nq, g_per_cell, gscale, dmean2, mscale);
    return bwd_t<128>(st, chat, dcchat, cells, row_ptr, M, B, L, C, Mq, uq, what, shat, qmask, dchat, dMq, duq, dwhat, dshat, ws, dl, Nq, g_per_cell, gscale, dmean2, mscale);
}

}  // namespace smin

using namespace smin;

// The attention core of the content unit on its own (content stream: the unit's two linear maps are applied in the
// dl-dimensional space by the caller, see content_stream in the Python host).  chat [N*C][dl] -> cc [N*C][dl] and/or
// ccmean [N][dl] = mean_c cc (either output may be NULL).
extern "C" int smin_content_attn_fwd(void* stream, const float* chat, const int32_t* cells, const int32_t* row_ptr,
                                     int N, int B, int L, int C, int dl, int Nq,
                                     const float* Mq, const float* uq, const float* what, const float* shat, const float* qmask,
                                     float* cc, float* ccmean)
{
    SMIN_REQUIRE(dl % 8 == 0 && dl <= 128 && C >= 2 && C <= 4 && Nq >= 1 && Nq <= 32);
    if (N == 0) return 0;
    SMIN_REQUIRE(cc || ccmean);
    ProfScope prof((hipStream_t)stream, SMIN_PROF_ATTN_FWD);
    return launch_content_attn_fwd((hipStream_t)stream, chat, cells, row_ptr, B, L, C, Mq, uq, what, shat, qmask, cc, ccmean, dl, Nq);
}

extern "C" size_t smin_content_attn_bwd_workspace_bytes(int N, int B, int C, int dl)
{
    return sizeof(float) * (content_attn_bwd_ws_floats(N * C, B, dl) + 64);
}

// Gradients dcc [N*C][dl] and/or dccmean [N][dl] (either may be NULL, not both) -> dchat and the per-sample word-side
// gradients.  Masked cells (m == 0) get dchat = 0.
extern "C" int smin_content_attn_bwd(void* stream, const float* dcc, const float* dccmean, const float* chat,
                                     const int32_t* cells, const int32_t* row_ptr, int N, int B, int L, int C, int dl, int Nq,
                                     const float* Mq, const float* uq, const float* what, const float* shat, const float* qmask,
                                     float* dchat, float* dMq, float* duq, float* dwhat, float* dshat, void* ws, size_t ws_bytes)
{
    hipStream_t st = (hipStream_t)stream;
    SMIN_REQUIRE(dl % 8 == 0 && dl <= 128 && C >= 2 && C <= 4 && Nq >= 1 && Nq <= 32);
    if (N == 0) return 0;
    SMIN_REQUIRE(dcc || dccmean);
    SMIN_REQUIRE(ws_bytes >= smin_content_attn_bwd_workspace_bytes(N, B, C, dl));
    const int M = N * C;
    float* aws = reinterpret_cast<float*>(ws);
    const float* g = dcc; const float* g2 = nullptr; int per_cell = 0; float gscale = 1.0f;
    if (dcc && dccmean) g2 = dccmean;                               // both consumers: summed while the rows are loaded
    else if (!dcc) { g = dccmean; per_cell = 1; gscale = 1.0f / C; }
    ProfScope prof(st, SMIN_PROF_ATTN_BWD);
    return launch_content_attn_bwd(st, chat, g, cells, row_ptr, M, B, L, C, Mq, uq, what, shat, qmask, dchat, dMq, duq, dwhat, dshat, aws, dl, Nq,
                                   per_cell, gscale, g2, 1.0f / C);
}
