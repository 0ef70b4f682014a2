// ContentAttention core on the matrix cores (reference models.py:207-226 and 253-266), fp32 MFMA 16x16x4.
//
// One wave owns a tile of 4 cells = 16 "rows" (row n = 4*cell + clip; clips >= C are zero padding).  Rows live on
// the MFMA column index (lane & 15); lane (n, kg = lane >> 4) holds the features d = 16*j + 4*kg + q of row n, so
// every per-row vector (chat, a, q, gradients) is DL/4 registers in exactly the accumulator layout:
//   S^T[w][n]  = sum_d Mq[w][d] chat[n][d]          MFMA  A = Mq (LDS, rows = words), B = chat (registers)
//   P          = softmax over words                  in-register: 8 values per lane + two cross-lane exchanges
//   a^T[d][n]  = sum_w what[w][d] P^T[w][n]         MFMA  A = what^T (LDS), B = the P accumulator as is
//   q, Z = q q^T over the 4 clips of a cell, A = softmax(Z), cchat = A chat        VALU + DPP quad permutes
// Work is cut into equal ranges of the packed cell list, one per resident workgroup; a workgroup stages the word-side
// tiles (K/V: Mq, what) of a sample once per range segment (a range rarely crosses a sample boundary) -- not once per
// 64-cell chunk, which made the staging, not the attention, the cost of the first version.
// Words live in LDS in "slot" order: slot s = 16b + 4kg + r holds word 16b + 4r + kg, so that accumulator register r of
// a score block (slots 16b + 4kg + r over the four lane groups kg) is the contiguous word group 16b + 4r .. + 3: the
// contractions over words (a = P what, dchat += dS Mq) then skip whole MFMA steps past Nq (20 words: 5 steps, not 8).
// The backward pass mirrors the forward (dP^T = what . da^T and dchat^T += Mq^T . dS^T on MFMA) and reduces the
// per-sample word-side gradients in the same launch: the eight waves of a workgroup deposit the da / dS / P rows of a
// round (8 tiles = 128 rows) in LDS and then contract them over the rows on the 32x32x2 MFMA
// (dMq = dS^T chat, dwhat = P^T da, dshat = colsum da, duq = colsum dS), each wave owning one 32 x 32 block of the result.
// da, dS and P never reach HBM; per-(range, sample) partial slabs are summed in fixed order by a small second launch.
// (History: 32-row tiles on the 32x32x2 MFMA needed 426 registers in the backward; 16-row tiles with a separate
// word-gradient kernel re-read 0.5 GB of da / dS / P per launch.)
#include "content_attn.h"
#include "smin_hip.h"
#include <stdlib.h>

namespace smin {

constexpr int LDW = 36;                                       // row stride of the transposed [d][slot] LDS image

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
    asm("s_nop 1\n\t"
                 "v_fmac_f32_dpp %0, %3, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                 "v_fmac_f32_dpp %1, %3, %5 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                 "v_fmac_f32_dpp %2, %3, %6 quad_perm:[3,2,1,0] row_mask:0xf bank_mask:0xf bound_ctrl:1"
                 : "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "v"(y1), "v"(y2), "v"(y3));
}
// acc += sum_{o=1..3} nb<o>(x) * y_o
__device__ __forceinline__ void fmac_nb_sum(float& acc, float x, float y1, float y2, float y3) {
    asm("s_nop 1\n\t"
                 "v_fmac_f32_dpp %0, %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                 "v_fmac_f32_dpp %0, %1, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                 "v_fmac_f32_dpp %0, %1, %4 quad_perm:[3,2,1,0] row_mask:0xf bank_mask:0xf bound_ctrl:1"
                 : "+v"(acc) : "v"(x), "v"(y1), "v"(y2), "v"(y3));
}

__device__ __forceinline__ int wmap(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
// word held by LDS slot s (and back: the map is an involution on each block of 16)
__device__ __host__ __forceinline__ int slot_word(int s) { return (s & 16) + 4 * (s & 3) + ((s >> 2) & 3); }

// per-lane geometry of a tile
struct RowGeom {
    int row;            // global row index n*C + c (clamped to a valid row when !ok)
    bool ok;            // this lane carries a real (cell, clip)
    float m;            // cell mask
    bool nbok[4];       // neighbour clip c ^ o exists
};
// ---- 16-row tiles on v_mfma_f32_16x16x4_f32 ---------------------------------------------------------------------
// Accumulator layout of the 16x16x4 MFMA: register r of lane (n, kg) is element [4*kg + r][n], so
//   S^T block b (slots 16b .. 16b+15): lane holds slots 16b + 4kg + r            -> P[4b + r]
//   a^T block j (features 16j .. 16j+15): lane holds features 16j + 4kg + r      -> aligned with ch[j][r]
// and the P / dS accumulators feed the next MFMA as B operands without any data movement.
typedef float f32x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4v mfma16(float a, float b, f32x4v c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
// exp and reciprocal of the two softmaxes as single instructions (v_exp_f32 after a multiply by log2 e; v_rcp_f32): 1 ulp each,
// against ~14 instructions per expf and ~10 per IEEE division -- 12 exponentials and 2 divisions per backward tile.  The arguments
// are differences to the row maximum (<= 0; -inf for masked entries gives exactly 0).
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// v_mfma_f32_4x4x1_16b_f32: sixteen independent 4 x 4 outer products, block b = lanes 4b .. 4b+3: lane 4b + j, register i gets
// a[lane 4b + i] * b[lane 4b + j] (layout confirmed on gfx950 by tools/mfma4x4_probe.hip).  Two passes (8 cycles): a quarter of a
// 16x16x4 step for the same FLOPs, all of them useful when only four words are contracted.
__device__ __forceinline__ f32x4v mfma4(float a, float b, f32x4v c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0); }
// all-reduce over the four lane groups kg (lanes n, n+16, n+32, n+48) in registers: gfx950's v_permlane16_swap /
// v_permlane32_swap exchange 16- and 32-lane rows without the LDS crossbar a __shfl_xor goes through.  Every lane ends with
// the same bits: (r0 op r1) op (r2 op r3).
__device__ __forceinline__ float kg_sum(float v) {
    const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    const float s = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(s), __float_as_uint(s), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
__device__ __forceinline__ float kg_max(float v) {
    const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    const float s = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(s), __float_as_uint(s), false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}

// tile of cells n0 .. n0+3 of a segment ending at n_end (n_end >= 1); lanes past the segment (or every lane, when the tile
// itself lies past it) are padding: ok = false, loads clamped to the segment's last cell
__device__ __forceinline__ RowGeom row_geom16(const int* cells, int n0, int n_end, int C, int lane) {
    const int j = lane & 15, cell = n0 + (j >> 2), c = j & 3;
    RowGeom g;
    g.ok = cell < n_end && c < C;
    const int cc = g.ok ? cell : n_end - 1;
    g.row = cc * C + (g.ok ? c : 0);
    g.m = g.ok ? (float)cells[4 * (size_t)cc + 3] : 0.f;
#pragma unroll
    for (int o = 0; o < 4; ++o) g.nbok[o] = (c ^ o) < C;
    return g;
}

// raw row loads (unconditional; masked when consumed)
template <int DL>
__device__ __forceinline__ void fetch_rows16(float4 (&v)[DL / 16], const float* src, int row, int dl, int kg) {
#pragma unroll
    for (int j = 0; j < DL / 16; ++j) v[j] = ldg4(src + (size_t)row * dl + 4 * kg + min(16 * j, dl - 16));      // dl % 16 == 0: a uniform clamp
}
template <int DL>
__device__ __forceinline__ void mask_rows16(float (&v)[DL / 16][4], const float4 (&x)[DL / 16], bool rok, int dl, int kg) {
#pragma unroll
    for (int j = 0; j < DL / 16; ++j) {
        const bool ok = rok && 16 * j < dl;
        v[j][0] = ok ? x[j].x : 0.f; v[j][1] = ok ? x[j].y : 0.f; v[j][2] = ok ? x[j].z : 0.f; v[j][3] = ok ? x[j].w : 0.f;
    }
}

// Words 16 .. 19 (the only valid slots of block 1 when 16 < Nq <= 20) against the rows v of a tile: out[i] = <X[word 16 + i], v[row]>
// for the lane's own row.  A block of the 4x4x1 MFMA is the four clips of a cell at one lane group kg: the a operand is the word
// image (lane & 3 picks the word: slot 16 + 4 (lane & 3) of the slot-major image), the b operand the lane's own row value, and each
// lane group contracts its own features; kg_sum adds the four partial sums.  32 two-pass MFMAs instead of 32 eight-pass ones
// of which 12 of 16 output rows were padding.
template <int DL>
__device__ __forceinline__ float4 extra_words_operand(int j, const float* sX, int lane)
{
    constexpr int LDM = DL + 4;
    return ldg4(sX + (16 + 4 * (lane & 3)) * LDM + 16 * j + 4 * (lane >> 4));
}
// two accumulators, alternating: consecutive MFMAs are independent
__device__ __forceinline__ void extra_words_mfma(f32x4v& accA, f32x4v& accB, float4 a, const float (&v)[4])
{
    accA = mfma4(a.x, v[0], accA); accB = mfma4(a.y, v[1], accB); accA = mfma4(a.z, v[2], accA); accB = mfma4(a.w, v[3], accB);
}
// every lane: the four totals; lane group kg keeps word 16 + kg (its slot 16 + 4 kg + 0)
__device__ __forceinline__ float extra_words_select(f32x4v acc, int kg)
{
    const float t0 = kg_sum(acc[0]), t1 = kg_sum(acc[1]), t2 = kg_sum(acc[2]), t3 = kg_sum(acc[3]);
    return kg == 0 ? t0 : (kg == 1 ? t1 : (kg == 2 ? t2 : t3));
}

// P[4b + r] = softmax over words of row n, slot 16b + 4kg + r.  sM: [32 slots][LDM] (A operand: lane l15 = slot in block)
// WS = number of 4-word contraction steps the kernel is built for (>= ceil(Nq / 4)): compile-time, so that no MFMA sits
// behind a branch on Nq (hipcc turns `if (Nq > ..) mfma` into a basic block per MFMA and stops scheduling across them).
// Slots past Nq hold zeros in LDS and get P = 0, so computing a step that is not needed adds exact zeros.
template <int DL, int WS>
__device__ __forceinline__ void scores_softmax16(float (&P)[8], const float (&ch)[DL / 16][4], const float* sM, const float* sU, const float* sQ,
                                                 int Nq, float scale, int lane)
{
    constexpr int LDM = DL + 4, KJ = DL / 16, G = KJ >= 2 ? 2 : 1, NG = KJ / G;
    constexpr int NR = WS >= 8 ? 8 : WS;                          // score registers that can hold a word: slots 16 + 4kg + r, r < WS - 4
    const int l15 = lane & 15, kg = lane >> 4;
    // two accumulators per block, alternating: a dependent 16x16x4 MFMA issues 40 cycles after its predecessor, an independent one 32
    f32x4v S0 = {0.f, 0.f, 0.f, 0.f}, S0b = {0.f, 0.f, 0.f, 0.f}, S1 = {0.f, 0.f, 0.f, 0.f}, S1b = {0.f, 0.f, 0.f, 0.f};
    // word operands in batches of G blocks, requested one batch ahead and pinned (one wait per batch; left alone hipcc emits
    // read -> wait -> 4 MFMAs for every block and every block pays the LDS latency)
    float4 a0[2][G], a1[2][G];
    auto request = [&](int gI, int buf) {
#pragma unroll
        for (int u = 0; u < G; ++u) {
            const int j = G * gI + u;
            a0[buf][u] = ldg4(sM + l15 * LDM + 16 * j + 4 * kg);
            if (WS == 5) a1[buf][u] = extra_words_operand<DL>(j, sM, lane);
            else if (WS > 4) a1[buf][u] = ldg4(sM + (16 + l15) * LDM + 16 * j + 4 * kg);
        }
    };
    request(0, 0);
#pragma unroll
    for (int gI = 0; gI < NG; ++gI) {
        const int buf = gI & 1;
#pragma unroll
        for (int u = 0; u < G; ++u) {
            asm volatile("" : "+v"(a0[buf][u].x), "+v"(a0[buf][u].y), "+v"(a0[buf][u].z), "+v"(a0[buf][u].w));
            if (WS > 4) asm volatile("" : "+v"(a1[buf][u].x), "+v"(a1[buf][u].y), "+v"(a1[buf][u].z), "+v"(a1[buf][u].w));
        }
        if (gI + 1 < NG) request(gI + 1, buf ^ 1);
#pragma unroll
        for (int u = 0; u < G; ++u) {
            const int j = G * gI + u;
            const float4 x = a0[buf][u];
            S0 = mfma16(x.x, ch[j][0], S0); S0b = mfma16(x.y, ch[j][1], S0b); S0 = mfma16(x.z, ch[j][2], S0); S0b = mfma16(x.w, ch[j][3], S0b);
            if (WS == 5) extra_words_mfma(S1, S1b, a1[buf][u], ch[j]);
            else if (WS > 4) {
                const float4 y = a1[buf][u];
                S1 = mfma16(y.x, ch[j][0], S1); S1b = mfma16(y.y, ch[j][1], S1b); S1 = mfma16(y.z, ch[j][2], S1); S1b = mfma16(y.w, ch[j][3], S1b);
            }
        }
    }
    S0 += S0b; S1 += S1b;
    if (WS == 5) { const float s1 = extra_words_select(S1, kg); S1[0] = s1; S1[1] = 0.f; S1[2] = 0.f; S1[3] = 0.f; }      // slots 16 + 4kg + r, r > 0: words >= 20
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 8; ++r) P[r] = 0.f;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int sl = 16 * (r >> 2) + 4 * kg + (r & 3);          // slot; its word is 16b + 4r + kg
        const int w = 16 * (r >> 2) + 4 * (r & 3) + kg;
        float v = ((r < 4 ? S0[r & 3] : S1[r & 3]) + sU[sl]) * scale;
        const float qm = sQ[sl];
        v = (qm == 0.f) ? -1e9f : v * qm;                         // models.py:216-218
        v = (w < Nq) ? v : -INFINITY;
        P[r] = v;
        mx = fmaxf(mx, v);
    }
    mx = kg_max(mx);
    float den = 0.f;
#pragma unroll
    for (int r = 0; r < NR; ++r) { P[r] = fast_exp(P[r] - mx); den += P[r]; }
    den = kg_sum(den);
    const float inv = fast_rcp(den);
#pragma unroll
    for (int r = 0; r < NR; ++r) P[r] *= inv;
}

// out^T block j: acc[r] += sum_slots X[slot][16j + 4kg' ...]: the contraction over words with the word operand read from the
// transposed image sXT [feature][LDW] (one b128 read feeds the four steps of a slot block).  Steps whose word group
// 16b + 4r .. 16b + 4r + 3 lies past Nq are skipped.
template <int WS>
__device__ __forceinline__ f32x4v words_tile16_T(f32x4v acc, int j, const float (&V)[8], const float* sXT, int lane)
{
    const int l15 = lane & 15, kg = lane >> 4;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        if (4 * b < WS) {
            const float4 w4 = ldg4(sXT + (16 * j + l15) * LDW + 16 * b + 4 * kg);
            acc = mfma16(w4.x, V[4 * b], acc);
            if (4 * b + 1 < WS) acc = mfma16(w4.y, V[4 * b + 1], acc);
            if (4 * b + 2 < WS) acc = mfma16(w4.z, V[4 * b + 2], acc);
            if (4 * b + 3 < WS) acc = mfma16(w4.w, V[4 * b + 3], acc);
        }
    }
    return acc;
}
// the same contraction with the word operand read from the slot-major image sX [slot][LDM] (b32 reads, conflict-free: the
// backward keeps only the slot-major images so that its round tiles fit in LDS)
template <int LDM, int WS>
__device__ __forceinline__ f32x4v words_tile16_S(f32x4v acc, int j, const float (&V)[8], const float* sX, int lane)
{
    const int l15 = lane & 15, kg = lane >> 4;
    const float* base = sX + (4 * kg) * LDM + 16 * j + l15;
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (4 * b + r < WS) acc = mfma16(base[(16 * b + r) * LDM], V[4 * b + r], acc);
    return acc;
}

// The same contraction for two adjacent blocks j0, j0 + 1 in three steps a caller can pipeline: (1) request every word operand of
// both blocks, (2) pin: one wait for the whole batch instead of a read -> wait -> MFMA round trip per step (hipcc, short of
// registers, sinks each LDS read to its use), (3) the two dependent MFMA chains interleaved.
template <int LDM, int WS>
__device__ __forceinline__ void words_pair_load(float (&wa)[WS], float (&wb)[WS], int j0, const float* sX, int lane)
{
    const float* base = sX + (4 * (lane >> 4)) * LDM + 16 * j0 + (lane & 15);
#pragma unroll
    for (int s2 = 0; s2 < WS; ++s2) { const int row = 16 * (s2 >> 2) + (s2 & 3); wa[s2] = base[row * LDM]; wb[s2] = base[row * LDM + 16]; }
}
template <int WS>
__device__ __forceinline__ void words_pair_pin(float (&wa)[WS], float (&wb)[WS])
{
#pragma unroll
    for (int s2 = 0; s2 < WS; ++s2) { asm volatile("" : "+v"(wa[s2])); asm volatile("" : "+v"(wb[s2])); }
}
template <int WS>
__device__ __forceinline__ void words_pair_mfma(f32x4v& accA, f32x4v& accB, const float (&wa)[WS], const float (&wb)[WS], const float (&V)[8])
{
#pragma unroll
    for (int s2 = 0; s2 < WS; ++s2) { accA = mfma16(wa[s2], V[s2], accA); accB = mfma16(wb[s2], V[s2], accB); }
}

// The 4 x 4 clip-by-clip products of a cell on the 4x4x1 MFMA: with a = x, b = y of the same lane, register i of lane (.., clip c)
// accumulates x_i y_c over the lane's features, i.e. D[i] = <x_i, y_c> for the lane group's share of the features.  The per-clip
// code wants the neighbour order out[o] = D[c ^ o]: eight selects on the two clip bits.  (As three fused DPP FMAs per feature
// the same sums cost 4 vector instructions + a hazard nop per feature; here one two-pass MFMA.)
__device__ __forceinline__ void quad_neighbour_order(float (&out)[4], f32x4v D, int lane)
{
    const bool b0 = lane & 1, b1 = lane & 2;
    const float t0 = b0 ? D[1] : D[0], t1 = b0 ? D[0] : D[1], t2 = b0 ? D[3] : D[2], t3 = b0 ? D[2] : D[3];
    out[0] = b1 ? t2 : t0; out[1] = b1 ? t3 : t1; out[2] = b1 ? t0 : t2; out[3] = b1 ? t1 : t3;
}

// z[o] = <q_c, q_{c^o}> partial sums of the lane group -> Ao[o] = softmax over the clips present, times the cell mask
__device__ __forceinline__ void clip_softmax(float (&Ao)[4], float (&z)[4], const RowGeom& g, float scale)
{
    float mx = -INFINITY;
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        z[o] = kg_sum(z[o]) * scale;
        z[o] = g.nbok[o] ? z[o] : -INFINITY;
        mx = fmaxf(mx, z[o]);
    }
    float den = 0.f;
#pragma unroll
    for (int o = 0; o < 4; ++o) { Ao[o] = fast_exp(z[o] - mx); den += Ao[o]; }
    const float inv = g.m * fast_rcp(den);                        // models.py:262-263: softmax, then * mask
#pragma unroll
    for (int o = 0; o < 4; ++o) Ao[o] *= inv;
}

// clip self-attention of a tile: Ao[o] = softmax_c'(q_c . q_c') * m for neighbour c' = c ^ o.  ATT(j) returns a^T block j.
template <int DL, class ATT>
__device__ __forceinline__ void clip_attention16(float (&Ao)[4], const float (&ch)[DL / 16][4], const float* sS, const RowGeom& g, float scale,
                                                 int lane, ATT att)
{
    const int kg = lane >> 4;
    f32x4v Z = {0.f, 0.f, 0.f, 0.f}, Zb = {0.f, 0.f, 0.f, 0.f};  // q q^T over the clips of a cell on the 4x4x1 MFMA (quad_neighbour_order)
#pragma unroll
    for (int j = 0; j < DL / 16; ++j) {
        const f32x4v acc = att(j);
        const float4 sh = ldg4(sS + 16 * j + 4 * kg);
        const float shv[4] = {sh.x, sh.y, sh.z, sh.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float qv = ch[j][q] * (acc[q] + shv[q]);
            if (q & 1) Zb = mfma4(qv, qv, Zb); else Z = mfma4(qv, qv, Z);
        }
        if ((j & 1) == 1) __builtin_amdgcn_sched_barrier(0);
    }
    Z += Zb;
    float z[4];
    quad_neighbour_order(z, Z, lane);
    clip_softmax(Ao, z, g, scale);
}

// backward form: the a^T blocks come in pairs (words_pair_*), the next pair's word operands are requested while this pair's
// scores are formed, and every block is parked in the lane's own row of X (myX) for the dq pass.
template <int DL, int WS>
__device__ __forceinline__ void clip_attention16_bwd(float (&Ao)[4], const float (&ch)[DL / 16][4], const float (&P)[8], const float* sW, const float* sS,
                                                     float* myX, const RowGeom& g, float scale, int lane)
{
    constexpr int LDM = DL + 4, KJ = DL / 16;
    const int kg = lane >> 4;
    f32x4v Z = {0.f, 0.f, 0.f, 0.f}, Zb = {0.f, 0.f, 0.f, 0.f};
    if constexpr (KJ >= 2) {
        float wa[2][WS], wb[2][WS];
        words_pair_load<LDM, WS>(wa[0], wb[0], 0, sW, lane);
#pragma unroll
        for (int jp = 0; jp < KJ / 2; ++jp) {
            const int cur = jp & 1;
            words_pair_pin<WS>(wa[cur], wb[cur]);
            f32x4v acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
            words_pair_mfma<WS>(acc[0], acc[1], wa[cur], wb[cur], P);
            if (jp + 1 < KJ / 2) words_pair_load<LDM, WS>(wa[cur ^ 1], wb[cur ^ 1], 2 * jp + 2, sW, lane);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int j = 2 * jp + u;
                const float4 sh = ldg4(sS + 16 * j + 4 * kg);
                const float tq[4] = {acc[u][0] + sh.x, acc[u][1] + sh.y, acc[u][2] + sh.z, acc[u][3] + sh.w};
                stg4(myX + 16 * j + 4 * kg, make_float4(tq[0], tq[1], tq[2], tq[3]));      // a + shat, parked for the dq pass
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float qv = ch[j][q] * tq[q];
                    if (u) Zb = mfma4(qv, qv, Zb); else Z = mfma4(qv, qv, Z);
                }
            }
        }
    } else {
        f32x4v z4 = {0.f, 0.f, 0.f, 0.f};
        const f32x4v acc = words_tile16_S<LDM, WS>(z4, 0, P, sW, lane);
        const float4 sh = ldg4(sS + 4 * kg);
        const float tq[4] = {acc[0] + sh.x, acc[1] + sh.y, acc[2] + sh.z, acc[3] + sh.w};
        stg4(myX + 4 * kg, make_float4(tq[0], tq[1], tq[2], tq[3]));
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float qv = ch[0][q] * tq[q];
            Z = mfma4(qv, qv, Z);
        }
    }
    Z += Zb;
    float z[4];
    quad_neighbour_order(z, Z, lane);
    clip_softmax(Ao, z, g, scale);
}

// Stage one sample's word-side operands in slot order.  All global loads of the pass are issued before the first LDS store
// (unconditional, clamped addresses; invalid slots / features become zeros).  NT threads.
//   sM / sW : [32 slots][LDM]   (float4 along features)      sMT / sWT : [DLP features][LDW]   (slot-contiguous)
// (tid = threadIdx.x, handed in so that a caller can hide it from loop-invariant code motion: hoisted out of the segment loop the
//  clamped indices and predicates of the staging passes stay live for the whole kernel -- dozens of registers, spilled)
template <int DL, int NT>
__device__ __forceinline__ void stage_slot_major(float* sA, const float* src, int b, int dl, int Nq, int tid)
{
    constexpr int LDM = DL + 4, Q = DL / 4, TOT = 32 * Q, IT = (TOT + NT - 1) / NT;
    float4 v[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int idx = min(tid + NT * it, TOT - 1), sl = idx / Q, d = (idx % Q) * 4;
        const int w = min(slot_word(sl), Nq - 1);
        v[it] = ldg4(src + ((size_t)b * Nq + w) * dl + min(d, dl - 4));
    }
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int idx = tid + NT * it, sl = idx / Q, d = (idx % Q) * 4;
        if (idx < TOT) stg4(sA + sl * LDM + d, f4sel(slot_word(sl) < Nq && d < dl, v[it]));
    }
}
template <int DL, int NT>
__device__ __forceinline__ void stage_feature_major(float* sT, const float* src, int b, int dl, int Nq, int tid)
{
    constexpr int Q = DL / 4, TOT = 32 * Q, IT = (TOT + NT - 1) / NT;
    float4 v[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {                             // slot fastest: the transposed scalar stores are conflict-free
        const int idx = min(tid + NT * it, TOT - 1), sl = idx & 31, d = (idx >> 5) * 4;
        const int w = min(slot_word(sl), Nq - 1);
        v[it] = ldg4(src + ((size_t)b * Nq + w) * dl + min(d, dl - 4));
    }
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int idx = tid + NT * it, sl = idx & 31, d = (idx >> 5) * 4;
        if (idx < TOT) {
            const float4 x = f4sel(slot_word(sl) < Nq && d < dl, v[it]);
            sT[(d + 0) * LDW + sl] = x.x; sT[(d + 1) * LDW + sl] = x.y; sT[(d + 2) * LDW + sl] = x.z; sT[(d + 3) * LDW + sl] = x.w;
        }
    }
}
template <int DL, int NT>
__device__ __forceinline__ void stage_vectors(float* sS, float* sU, float* sQ, const float* shat, const float* uq, const float* qmask, int b, int dl, int Nq, int tid)
{
    const int t = tid;
    for (int d = t; d < DL; d += NT) sS[d] = d < dl ? shat[(size_t)b * dl + d] : 0.f;
    if (t < 32) {
        const int w = slot_word(t);
        sU[t] = w < Nq ? uq[(size_t)b * Nq + w] : 0.f;
        sQ[t] = w < Nq ? qmask[(size_t)b * Nq + w] : 0.f;
    }
}

// ---- forward ----------------------------------------------------------------------------------------------------
// LDS (floats): sM [32][DL+4] | sWT [DL][36] | sS [DL] | sU [32] | sQ [32]
template <int DL>
static size_t fwd_lds_bytes() { return sizeof(float) * (size_t)(32 * (DL + 4) + DL * LDW + DL + 64); }

// ROWS / MEAN: which outputs exist.  Compile-time: a runtime `if (pointer)` inside the unrolled output loop costs a branch
// per 16-feature block and stops hipcc from scheduling across the blocks.
// ROWS: 0 none, 1 fp32 rows, 2 rows stored as bf16 (round to nearest even; cchat then points to 16-bit elements)
// EXACT: dl == DL and C == 4 (every shipped configuration): row strides, feature clamps and clip predicates are constants --
// fewer address instructions and far fewer scalar registers (the general form keeps a clamped offset and a predicate per block)
template <int DL, int WS, int ROWS, bool MEAN, bool EXACT>
__global__ __launch_bounds__(256, 4)
void content_attn_fwd_kernel(const float* __restrict__ chat, const int* __restrict__ cells, const int* __restrict__ row_ptr, int L, int C,
                             const float* __restrict__ Mq, const float* __restrict__ uq, const float* __restrict__ what,
                             const float* __restrict__ shat, const float* __restrict__ qmask,
                             float* __restrict__ cchat, float* __restrict__ ccmean, int dl, int Nq, int N, int cells_per_range, float scale)
{
    extern __shared__ __attribute__((aligned(16))) float smem_dyn[];
    constexpr int KJ = DL / 16;
    if (EXACT) { dl = DL; C = 4; }
    float* sM = smem_dyn; float* sWT = sM + 32 * (DL + 4); float* sS = sWT + DL * LDW; float* sU = sS + DL; float* sQ = sU + 32;
    const int n_lo = blockIdx.x * cells_per_range;
    if (n_lo >= N) return;
    const int n_hi = min(N, n_lo + cells_per_range);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, kg = lane >> 4;
    const float invC = 1.0f / C;

    for (int n = n_lo; n < n_hi;) {                               // one iteration per sample the range touches
        const int b = __builtin_amdgcn_readfirstlane(cells[4 * (size_t)n]);
        const int seg_end = max(n + 1, min(n_hi, row_ptr[(b + 1) * L]));      // (a well-formed list ends its sample past n; never step back)
        __syncthreads();                                          // the previous segment's tiles are done with the LDS images
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));                             // not loop-invariant as far as the compiler knows (see stage_slot_major)
        stage_slot_major<DL, 256>(sM, Mq, b, dl, Nq, tid);
        stage_feature_major<DL, 256>(sWT, what, b, dl, Nq, tid);
        stage_vectors<DL, 256>(sS, sU, sQ, shat, uq, qmask, b, dl, Nq, tid);
        __syncthreads();

        int n0 = n + 4 * wave;
        RowGeom g = row_geom16(cells, n0, seg_end, C, lane);
        float4 raw[KJ];
        fetch_rows16<DL>(raw, chat, g.row, dl, kg);
        for (; n0 < seg_end; n0 += 16) {
            float ch[KJ][4], P[8], Ao[4];
            mask_rows16<DL>(ch, raw, g.ok, dl, kg);
            const RowGeom gc = g;
            // the next tile's rows are requested before this tile's arithmetic (clamped when there is none)
            g = row_geom16(cells, n0 + 16, seg_end, C, lane);
            fetch_rows16<DL>(raw, chat, g.row, dl, kg);
            scores_softmax16<DL, WS>(P, ch, sM, sU, sQ, Nq, scale, lane);
            __builtin_amdgcn_sched_barrier(0);
            clip_attention16<DL>(Ao, ch, sS, gc, scale, lane, [&](int j) {
                f32x4v z4 = {0.f, 0.f, 0.f, 0.f};
                return words_tile16_T<WS>(z4, j, P, sWT, lane);
            });
#pragma unroll
            for (int j = 0; j < KJ; ++j) {                          // cchat = A chat
                float o4[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float x = ch[j][q];
                    o4[q] = Ao[0] * x;
                    fmac_nb_sum(o4[q], x, Ao[1], Ao[2], Ao[3]);
                }
                const int d = 16 * j + 4 * kg;
                if (ROWS == 1) {
                    if (gc.ok && d < dl) stg4(cchat + (size_t)gc.row * dl + d, make_float4(o4[0], o4[1], o4[2], o4[3]));
                }
                if (ROWS == 2) {
                    unsigned short u[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) u[q] = __builtin_bit_cast(unsigned short, (__bf16)o4[q]);
                    if (gc.ok && d < dl)
                        *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(cchat) + (size_t)gc.row * dl + d) =
                            make_uint2((unsigned)u[0] | ((unsigned)u[1] << 16), (unsigned)u[2] | ((unsigned)u[3] << 16));
                }
                if (MEAN) {                                         // mean over the clips of the quad (padding lanes hold 0)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {                 // quad total in two DPP steps
                        const float t1 = o4[q] + nb<1>(o4[q]);
                        o4[q] = (t1 + nb<2>(t1)) * invC;
                    }
                    if (gc.ok && (lane & 3) == 0 && d < dl)
                        stg4(ccmean + (size_t)(gc.row / C) * dl + d, make_float4(o4[0], o4[1], o4[2], o4[3]));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        n = seg_end;
    }
}

// ---- backward with the word-side reductions fused ---------------------------------------------------------------------
// 256 threads = 4 waves (one per SIMD), TWO workgroups per CU: the two run out of phase, so that one workgroup's matrix-dense
// reduction phase fills the matrix pipe while the other sits in the dependent chains of its tile phase (round 2 ran one
// 512-thread workgroup per CU whose eight waves were all in the same phase: 35 % matrix-pipe occupancy).  LDS (floats):
//   sM [32][DL+4] | sW [32][DL+4] | sS [DL] | sU [32] | sQ [32] | X [64][DL+4] | tDs [64][LDP] | tP [64][LDP] | 32 slack
// A round = 4 tiles (one per wave) = 16 cells = 64 rows:
//   T   tile phase, per wave: forward recompute + gradients; dchat to HBM; the a rows are parked in X between their two uses
//       (clip-attention scores, then dq) instead of being recomputed; X then takes the da rows; dS / P rows into tDs / tP
//   R1  wave fq accumulates the 32 x 32 block [32 columns] x [features 32fq ..] of dwhat = P^T da (+ dshat = colsum da)
//       over the 64 rows with 32 v_mfma_f32_32x32x2_f32
//   W   every wave writes its chat rows (still in registers) over the da rows of X
//   R0  the same for dMq = dS^T chat (+ duq = colsum dS): the chat rows come from LDS, not a second time from L2 / HBM
// tDs / tP keep only the columns that can hold a word: the 16 slots of block 0, then the valid slots 16 + 4kg + r (r < WS - 4)
// of block 1 as column 16 + kg (WS - 4) + r; columns past that read the next row (finite garbage) and their results are dropped.
// Per (range, sample) segment the accumulators go to slab[range + sample]:  [dM 32 x dl | dW 32 x dl | dshat dl | du 32]
// in column order; content_attn_reduce_kernel sums a sample's slabs in range order (fixed order: deterministic).
#ifdef SMIN_ATTN_STAMPS
// diagnostic build only (tools/attn_stamps.sh): s_memtime at the phase boundaries of a few rounds of two workgroups
__device__ unsigned long long g_attn_stamps[2 * 4 * 8 * 16];
#define STAMP(k) do { if (stamp_on && (unsigned)stamp_round < 8u) { g_attn_stamps[((stamp_blk * 4 + wave) * 8 + stamp_round) * 16 + (k)] = __builtin_amdgcn_s_memtime(); } } while (0)
#else
#define STAMP(k) do { } while (0)
#endif
template <int WS> constexpr int bwd_ldp() { return WS <= 5 ? 20 : (WS == 6 ? 28 : 36); }
template <int WS> constexpr int bwd_ncols() { return 16 + 4 * (WS - 4); }
template <int DL, int WS>
static size_t bwd_lds_bytes() { return sizeof(float) * (size_t)(64 * (DL + 4) + DL + 64 + 64 * (DL + 4) + 128 * bwd_ldp<WS>() + 32); }
// word held by column c of the round tiles / slabs (nb1 = WS - 4 valid slots per lane group in block 1)
__device__ __host__ __forceinline__ int col_word(int c, int nb1) {
    if (c < 16) return 4 * (c & 3) + (c >> 2);
    const int e = c - 16, d = nb1 > 0 ? nb1 : 1;
    return 16 + 4 * (e % d) + e / d;
}

// geometry of a tile without its mask load (the mask is requested with the prefetched rows: nothing waits for it before the stores)
__device__ __forceinline__ RowGeom row_geom16_nm(int n0, int n_end, int C, int lane, int& cellc) {
    const int j = lane & 15, cell = n0 + (j >> 2), c = j & 3;
    RowGeom g;
    g.ok = cell < n_end && c < C;
    cellc = g.ok ? cell : n_end - 1;
    g.row = cellc * C + (g.ok ? c : 0);
    g.m = 0.f;
#pragma unroll
    for (int o = 0; o < 4; ++o) g.nbok[o] = (c ^ o) < C;
    return g;
}

// One reduction phase of a round: wave fq adds the 64 rows of the round to its [columns] x [features 32fq .. 32fq+31] block of
// dwhat = P^T da (KIND 1, + dshat = colsum da) or dMq = dS^T chat (KIND 0, + duq = colsum dS).  Column block 0 (16 slots) runs on
// v_mfma_f32_16x16x4_f32 (a = tile value [row 4t + kg][column l15], b = X[row 4t + kg][feature l15]; result register r of lane
// (feature l15, kg) = [column 4kg + r][feature]); with 16 < Nq <= 20 the four extra columns are four FMAs per step against the
// b value the lane already holds (the 32x32x2 MFMA of the first version spent 12 of its 32 columns on padding), with more
// words a second column block.  HOOK(t) is called once per step t = 0 .. 15: the next round's row loads are spread over the loop.
template <int KIND, int DL, int WS, class HOOK>
__device__ __forceinline__ void reduce_round(f32x4v (&rA)[2][2], f32x4v (&rE)[2], float (&cs)[2], float& csa, float& csb, float& csx,
                                             const float* T, const float* X, int fq, int lane, HOOK hook)
{
    constexpr int LDA = DL + 4, LDP = bwd_ldp<WS>(), NB1 = WS - 4;
    constexpr bool TWO = NB1 >= 2, XV = NB1 == 1;
    const int l15 = lane & 15, kg = lane >> 4;
    const float* Tr = T + kg * LDP;
    const float* X0 = X + kg * LDA + min(32 * fq + l15, DL - 1);
    const float* X1 = X + kg * LDA + min(32 * fq + 16 + l15, DL - 1);
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        hook(t);
        const float a0 = Tr[4 * t * LDP + l15];
        const float x0 = X0[4 * t * LDA], x1 = X1[4 * t * LDA];
        rA[0][0] = mfma16(a0, x0, rA[0][0]);
        rA[0][1] = mfma16(a0, x1, rA[0][1]);
        if (TWO) {
            const float a1 = Tr[4 * t * LDP + 16 + l15];
            rA[1][0] = mfma16(a1, x0, rA[1][0]);
            rA[1][1] = mfma16(a1, x1, rA[1][1]);
            if (KIND == 0) csb += a1;
        }
        if (XV) {
            // the four extra columns as 4 x 4 outer products: a block of the 4x4x1 MFMA is four adjacent features of one row, its a
            // operand the row's extra-column values (lane & 3 picks the column): register i of lane (feature, kg) = column 16 + i
            const float e = Tr[4 * t * LDP + 16 + (lane & 3)];
            rE[0] = mfma4(e, x0, rE[0]);
            rE[1] = mfma4(e, x1, rE[1]);
            if (KIND == 0) csx += e;                                // lane (.., c): column 16 + c
        }
        if (KIND == 1) { cs[0] += x0; cs[1] += x1; } else csa += a0;
        if ((t & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
}

template <int DL, int WS, bool MEAN2, bool PERCELL, bool EXACT>
__global__ __launch_bounds__(256, 2)
void content_attn_bwd_kernel(const float* __restrict__ chat, const float* __restrict__ dcchat,
                             const int* __restrict__ cells, const int* __restrict__ row_ptr, int L, int C,
                             const float* __restrict__ Mq, const float* __restrict__ uq, const float* __restrict__ what,
                             const float* __restrict__ shat, const float* __restrict__ qmask,
                             float* __restrict__ dchat, float* __restrict__ slab,
                             int dl, int Nq, int N, int cells_per_range, float scale, int g_per_cell, float gscale,
                             const float* __restrict__ dmean2, float mscale)
{
    extern __shared__ __attribute__((aligned(16))) float smem_dyn[];
    constexpr int LDM = DL + 4, KJ = DL / 16, LDA = DL + 4, DT = (DL + 31) / 32, LDP = bwd_ldp<WS>(), NB1 = WS - 4, NCOLS = bwd_ncols<WS>();
    constexpr bool TWO = NB1 >= 2, XV = NB1 == 1;
    if (EXACT) { dl = DL; C = 4; }
    float* sM = smem_dyn; float* sW = sM + 32 * LDM; float* sS = sW + 32 * LDM; float* sU = sS + DL; float* sQ = sU + 32;
    float* X = sQ + 32; float* tDs = X + 64 * LDA; float* tP = tDs + 64 * LDP;
    const int rg = blockIdx.x;
    const int n_lo = rg * cells_per_range;
    if (n_lo >= N) return;
    const int n_hi = min(N, n_lo + cells_per_range);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, kg = lane >> 4, l15 = lane & 15;
    const int fq = wave;
    const size_t slab_sz = (size_t)64 * dl + dl + 32;
    if (threadIdx.x < 32) tP[64 * LDP + threadIdx.x] = 0.f;       // slack behind the tiles (a b128 read of the last row's extra columns ends here)
    float* myX = X + (16 * wave + l15) * LDA;
#ifdef SMIN_ATTN_STAMPS
    const bool stamp_on = (rg == 3 || rg == 300) && lane == 0;
    const int stamp_blk = rg == 3 ? 0 : 1;
    int stamp_round = -3;                                         // the first two rounds are not recorded
#endif

    for (int n = n_lo; n < n_hi;) {                               // one iteration per sample the range touches
        const int b = __builtin_amdgcn_readfirstlane(cells[4 * (size_t)n]);
        const int seg_end = max(n + 1, min(n_hi, row_ptr[(b + 1) * L]));      // (a well-formed list ends its sample past n; never step back)
        __syncthreads();
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));                             // not loop-invariant as far as the compiler knows (see stage_slot_major)
        stage_slot_major<DL, 256>(sM, Mq, b, dl, Nq, tid);
        stage_slot_major<DL, 256>(sW, what, b, dl, Nq, tid);
        stage_vectors<DL, 256>(sS, sU, sQ, shat, uq, qmask, b, dl, Nq, tid);
        __syncthreads();
        // accumulators of the segment, [kind][column block][feature block]; rE: the four extra columns; cs*: column sums
        f32x4v rA0[2][2], rA1[2][2], rE0[2], rE1[2];
        float cs1[2] = {0.f, 0.f}, cs0a = 0.f, cs0b = 0.f, cs0x = 0.f, csdummy[2] = {0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { rA0[i][k][r] = 0.f; rA1[i][k][r] = 0.f; rE0[i][r] = 0.f; rE1[i][r] = 0.f; }
            }

        // the rows of a round (chat, output gradients) are requested one round ahead, spread over the previous round's reduction
        // phases, so that their HBM latency hides behind them
        float4 raw[KJ], gq[KJ], gm[KJ];
        int cellc, mraw;
        RowGeom g = row_geom16_nm(n + 4 * wave, seg_end, C, lane, cellc);
        mraw = cells[4 * (size_t)cellc + 3];
        fetch_rows16<DL>(raw, chat, g.row, dl, kg);
        fetch_rows16<DL>(gq, dcchat, PERCELL ? g.row / C : g.row, dl, kg);
        if (MEAN2) fetch_rows16<DL>(gm, dmean2, g.row / C, dl, kg);

        for (int c0 = n; c0 < seg_end; c0 += 16) {                // rounds
            float ch[KJ][4];
#ifdef SMIN_ATTN_STAMPS
            ++stamp_round;
#endif
            STAMP(0);
            {
                float P[8], Ao[4];
                mask_rows16<DL>(ch, raw, g.ok, dl, kg);
                g.m = g.ok ? (float)mraw : 0.f;                     // requested a round ago with the rows
                __builtin_amdgcn_sched_barrier(0);                  // the prefetched registers die here
                scores_softmax16<DL, WS>(P, ch, sM, sU, sQ, Nq, scale, lane);
                __builtin_amdgcn_sched_barrier(0);
                STAMP(1);
                __syncthreads();                                    // the previous round's R0 is done with X / tDs (first writes below)
                STAMP(2);
                clip_attention16_bwd<DL, WS>(Ao, ch, P, sW, sS, myX, g, scale, lane);      // a rows parked in X for the dq pass

                STAMP(3);
                // cchat = A chat :  dA[c][c^o] = <g_c, chat_{c^o}> ,  dchat_c = sum_o A[c^o][c] g_{c^o}
                float dch[KJ][4];
                float dAo[4];
                f32x4v dA4 = {0.f, 0.f, 0.f, 0.f}, dA4b = {0.f, 0.f, 0.f, 0.f};
                const float An1 = nb<1>(Ao[1]), An2 = nb<2>(Ao[2]), An3 = nb<3>(Ao[3]);
#pragma unroll
                for (int j = 0; j < KJ; ++j) {
                    // effective output-gradient rows (both consumers summed).  Rows of padding lanes and blocks past dl hold finite
                    // values of other rows (clamped loads): they meet chat = 0, attention weights 0 or are never stored
                    float gvj[4] = {gq[j].x, gq[j].y, gq[j].z, gq[j].w};
                    if (PERCELL) { gvj[0] *= gscale; gvj[1] *= gscale; gvj[2] *= gscale; gvj[3] *= gscale; }      // rows: gscale = 1
                    if (MEAN2) {                                    // the per-cell gradient arrives last (requested in R0)
                        gvj[0] = fmaf(gm[j].x, mscale, gvj[0]); gvj[1] = fmaf(gm[j].y, mscale, gvj[1]);
                        gvj[2] = fmaf(gm[j].z, mscale, gvj[2]); gvj[3] = fmaf(gm[j].w, mscale, gvj[3]);
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float x = ch[j][q], y = gvj[q];
                        if (j & 1) dA4b = mfma4(x, y, dA4b); else dA4 = mfma4(x, y, dA4);      // register i: <chat_i, g_c> = dA[c][i]
                        dch[j][q] = Ao[0] * y;
                        fmac_nb_sum(dch[j][q], y, An1, An2, An3);
                    }
                    if ((j & 1) == 1) __builtin_amdgcn_sched_barrier(0);
                }
                __builtin_amdgcn_sched_barrier(0);
                STAMP(4);
                // A = softmax(Z) * m ; Z symmetric in (c, c')
                float sym[4];
                {
                    float rd = 0.f, dZ[4];
                    dA4 += dA4b;
                    quad_neighbour_order(dAo, dA4, lane);
#pragma unroll
                    for (int o = 0; o < 4; ++o) { dAo[o] = kg_sum(dAo[o]); rd = fmaf(Ao[o], dAo[o], rd); }
#pragma unroll
                    for (int o = 0; o < 4; ++o) dZ[o] = g.nbok[o] ? Ao[o] * (dAo[o] - rd) : 0.f;
                    sym[0] = 2.0f * dZ[0] * scale;
                    sym[1] = (dZ[1] + nb<1>(dZ[1])) * scale;
                    sym[2] = (dZ[2] + nb<2>(dZ[2])) * scale;
                    sym[3] = (dZ[3] + nb<3>(dZ[3])) * scale;
                }
                // per 16-feature block: a (back from X) -> q = chat*(a+shat) -> dq = sum_o sym[o] q_{c^o} -> dchat += dq (a+shat), da = dq chat
                //                       -> da into X and straight into  dP^T[slot][n] += sum_d what[slot][d] da^T[d][n]  (MFMA)
                f32x4v dP0 = {0.f, 0.f, 0.f, 0.f}, dP1 = {0.f, 0.f, 0.f, 0.f}, dP0b = {0.f, 0.f, 0.f, 0.f}, dP1b = {0.f, 0.f, 0.f, 0.f};
                {
                    // LDS operands of block j + 1 are requested before block j's arithmetic and pinned before their use (one wait)
                    float4 a4b[2], w0b[2], w1b[2];
                    auto request = [&](int j, int buf) {
                        const int d = 16 * j + 4 * kg;
                        a4b[buf] = ldg4(myX + d);                    // a + shat (clip_attention16_bwd)
                        w0b[buf] = ldg4(sW + l15 * LDM + d);
                        if (WS == 5) w1b[buf] = extra_words_operand<DL>(j, sW, lane);
                        else if (WS > 4) w1b[buf] = ldg4(sW + (16 + l15) * LDM + d);
                    };
                    request(0, 0);
#pragma unroll
                    for (int j = 0; j < KJ; ++j) {
                        const int buf = j & 1, d = 16 * j + 4 * kg;
                        asm volatile("" : "+v"(a4b[buf].x), "+v"(a4b[buf].y), "+v"(a4b[buf].z), "+v"(a4b[buf].w));
                        asm volatile("" : "+v"(w0b[buf].x), "+v"(w0b[buf].y), "+v"(w0b[buf].z), "+v"(w0b[buf].w));
                        if (WS > 4) asm volatile("" : "+v"(w1b[buf].x), "+v"(w1b[buf].y), "+v"(w1b[buf].z), "+v"(w1b[buf].w));
                        if (j + 1 < KJ) request(j + 1, buf ^ 1);
                        const float av4[4] = {a4b[buf].x, a4b[buf].y, a4b[buf].z, a4b[buf].w};
                        float da4[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float tq = av4[q];
                            const float qv = ch[j][q] * tq;
                            float dq = sym[0] * qv;
                            fmac_nb_sum(dq, qv, sym[1], sym[2], sym[3]);
                            dch[j][q] = fmaf(dq, tq, dch[j][q]);
                            da4[q] = dq * ch[j][q];                 // 0 on padding lanes and past dl (ch is 0 there)
                        }
                        stg4(myX + d, make_float4(da4[0], da4[1], da4[2], da4[3]));
                        const float4 w0 = w0b[buf];
                        dP0 = mfma16(w0.x, da4[0], dP0); dP0b = mfma16(w0.y, da4[1], dP0b); dP0 = mfma16(w0.z, da4[2], dP0); dP0b = mfma16(w0.w, da4[3], dP0b);
                        if (WS == 5) extra_words_mfma(dP1, dP1b, w1b[buf], da4);
                        else if (WS > 4) {
                            const float4 w1 = w1b[buf];
                            dP1 = mfma16(w1.x, da4[0], dP1); dP1b = mfma16(w1.y, da4[1], dP1b); dP1 = mfma16(w1.z, da4[2], dP1); dP1b = mfma16(w1.w, da4[3], dP1b);
                        }
                    }
                }
                dP0 += dP0b; dP1 += dP1b;
                if (WS == 5) { const float d1 = extra_words_select(dP1, kg); dP1[0] = d1; dP1[1] = 0.f; dP1[2] = 0.f; dP1[3] = 0.f; }
                STAMP(5);
                // P = softmax(S), S = (raw + u) * scale * qmask
                float pd = 0.f;
#pragma unroll
                for (int r = 0; r < 8; ++r) pd = fmaf(P[r], (r < 4 ? dP0[r & 3] : dP1[r & 3]), pd);
                pd = kg_sum(pd);
                float dS[8];
#pragma unroll
                for (int r = 0; r < 8; ++r)
                    dS[r] = g.ok ? P[r] * ((r < 4 ? dP0[r & 3] : dP1[r & 3]) - pd) * sQ[16 * (r >> 2) + 4 * kg + (r & 3)] * scale : 0.f;
                {
                    float* myDs = tDs + (16 * wave + l15) * LDP;
                    float* myP = tP + (16 * wave + l15) * LDP;
                    stg4(myDs + 4 * kg, make_float4(dS[0], dS[1], dS[2], dS[3]));
                    stg4(myP + 4 * kg, g.ok ? make_float4(P[0], P[1], P[2], P[3]) : f4zero());
                    if (NB1 == 1) { myDs[16 + kg] = dS[4]; myP[16 + kg] = g.ok ? P[4] : 0.f; }
                    if (NB1 == 2) {
                        *reinterpret_cast<float2*>(myDs + 16 + 2 * kg) = make_float2(dS[4], dS[5]);
                        *reinterpret_cast<float2*>(myP + 16 + 2 * kg) = g.ok ? make_float2(P[4], P[5]) : make_float2(0.f, 0.f);
                    }
                    if (NB1 == 4) {
                        stg4(myDs + 16 + 4 * kg, make_float4(dS[4], dS[5], dS[6], dS[7]));
                        stg4(myP + 16 + 4 * kg, g.ok ? make_float4(P[4], P[5], P[6], P[7]) : f4zero());
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                STAMP(6);
                // raw = chat Mq^T :  dchat^T[d][n] += sum_slots Mq[slot][d] dS^T[slot][n]   -> dchat = (...) * m   (chat = linear(fc) * m)
                const float gm_ = g.m;
                if constexpr (KJ >= 2) {
                    float wa[2][WS], wb[2][WS];
                    words_pair_load<LDM, WS>(wa[0], wb[0], 0, sM, lane);
#pragma unroll
                    for (int jp = 0; jp < KJ / 2; ++jp) {
                        const int cur = jp & 1;
                        words_pair_pin<WS>(wa[cur], wb[cur]);
                        f32x4v acc[2] = {{dch[2 * jp][0], dch[2 * jp][1], dch[2 * jp][2], dch[2 * jp][3]},
                                         {dch[2 * jp + 1][0], dch[2 * jp + 1][1], dch[2 * jp + 1][2], dch[2 * jp + 1][3]}};
                        words_pair_mfma<WS>(acc[0], acc[1], wa[cur], wb[cur], dS);
                        if (jp + 1 < KJ / 2) words_pair_load<LDM, WS>(wa[cur ^ 1], wb[cur ^ 1], 2 * jp + 2, sM, lane);
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const int d = 16 * (2 * jp + u) + 4 * kg;
                            if (g.ok && 16 * (2 * jp + u) < dl)
                                stg4(dchat + (size_t)g.row * dl + d, make_float4(acc[u][0] * gm_, acc[u][1] * gm_, acc[u][2] * gm_, acc[u][3] * gm_));
                        }
                    }
                } else {
                    f32x4v acc = {dch[0][0], dch[0][1], dch[0][2], dch[0][3]};
                    acc = words_tile16_S<LDM, WS>(acc, 0, dS, sM, lane);
                    const int d = 4 * kg;
                    if (g.ok && d < dl) stg4(dchat + (size_t)g.row * dl + d, make_float4(acc[0] * gm_, acc[1] * gm_, acc[2] * gm_, acc[3] * gm_));
                }
            }
            STAMP(7);
            // next round's rows (past the segment: clamped to its last row, never used): requested inside the reduction loops below
            g = row_geom16_nm(c0 + 16 + 4 * wave, seg_end, C, lane, cellc);
            const float* nx_chat = chat + (size_t)g.row * dl + 4 * kg;
            const float* nx_g = dcchat + (size_t)(PERCELL ? g.row / C : g.row) * dl + 4 * kg;
            const float* nx_m = dmean2 + (size_t)(g.row / C) * dl + 4 * kg;
            STAMP(8);
            __syncthreads();                                        // the round's da / dS / P rows are in LDS
            STAMP(9);
            if (fq < DT) {                                          // R1: dwhat += P^T da, dshat += colsum da; the chat rows of the next round
                float dcsa = 0.f, dcsb = 0.f, dcsx = 0.f;
                reduce_round<1, DL, WS>(rA1, rE1, cs1, dcsa, dcsb, dcsx, tP, X, fq, lane, [&](int t) {
                    if (t == 0) mraw = cells[4 * (size_t)cellc + 3];
                    if ((t & 1) == 0 && t / 2 < KJ) raw[t / 2] = ldg4(nx_chat + min(16 * (t / 2), dl - 16));
                    if ((t & 1) == 1 && t / 2 < KJ) gq[t / 2] = ldg4(nx_g + min(16 * (t / 2), dl - 16));
                });
            } else {
                mraw = cells[4 * (size_t)cellc + 3];
                fetch_rows16<DL>(raw, chat, g.row, dl, kg);
                fetch_rows16<DL>(gq, dcchat, PERCELL ? g.row / C : g.row, dl, kg);
            }
            STAMP(10);
            __syncthreads();                                        // every wave is done with the da rows
            STAMP(11);
#pragma unroll
            for (int j = 0; j < KJ; ++j) stg4(myX + 16 * j + 4 * kg, make_float4(ch[j][0], ch[j][1], ch[j][2], ch[j][3]));      // W
            STAMP(12);
            __syncthreads();
            STAMP(13);
            if (fq < DT) {                                          // R0: dMq += dS^T chat, duq += colsum dS; the gradient rows of the next round
                reduce_round<0, DL, WS>(rA0, rE0, csdummy, cs0a, cs0b, cs0x, tDs, X, fq, lane, [&](int t) {
                    if (MEAN2 && (t & 1) == 0 && t / 2 < KJ) gm[t / 2] = ldg4(nx_m + min(16 * (t / 2), dl - 16));
                });
            } else if (MEAN2) fetch_rows16<DL>(gm, dmean2, g.row / C, dl, kg);
            STAMP(14);
        }
        // the segment's partial result: slab rows are tile columns (content_attn_reduce_kernel maps them to words)
        float* sl = slab + (size_t)(rg + b) * slab_sz;
        asm volatile("" : "+v"(sl));                              // the store addresses below are formed here, not ahead of the segment loop
        if (fq < DT) {
#pragma unroll
            for (int fb = 0; fb < 2; ++fb) {
                const int feat = 32 * fq + 16 * fb + l15;
                const bool fok = feat < dl;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int col = 4 * kg + r;
                    if (fok) {
                        sl[(size_t)col * dl + feat] = rA0[0][fb][r];
                        sl[(size_t)(32 + col) * dl + feat] = rA1[0][fb][r];
                        if (TWO && 16 + col < NCOLS) {
                            sl[(size_t)(16 + col) * dl + feat] = rA0[1][fb][r];
                            sl[(size_t)(48 + col) * dl + feat] = rA1[1][fb][r];
                        }
                    }
                    if (XV) {                                       // extra column 16 + r: partial sums of the four lane groups
                        const float e0 = kg_sum(rE0[fb][r]), e1 = kg_sum(rE1[fb][r]);
                        if (fok && kg == 0) { sl[(size_t)(16 + r) * dl + feat] = e0; sl[(size_t)(48 + r) * dl + feat] = e1; }
                    }
                }
                const float c1 = kg_sum(cs1[fb]);
                if (fok && kg == 0) sl[(size_t)64 * dl + feat] = c1;
            }
            // duq: column sums of dS (every wave holds the same sums: wave 0 writes)
            const float ca = kg_sum(cs0a), cb = kg_sum(cs0b), cx = kg_sum(cs0x);
            if (fq == 0 && kg == 0) {
                sl[(size_t)64 * dl + dl + l15] = ca;
                if (TWO) sl[(size_t)64 * dl + dl + 16 + l15] = cb;
                if (XV && l15 < 4) sl[(size_t)64 * dl + dl + 16 + l15] = cx;      // lanes 0 .. 3: columns 16 .. 19
            }
        }
        n = seg_end;
    }
}

// dMq / dwhat / dshat / duq of sample b = sum of its segments' slabs, in range order
__global__ void content_attn_reduce_kernel(const float* __restrict__ slab, const int* __restrict__ row_ptr, int L, int dl, int Nq, int cells_per_range,
                                           int nb1, float* __restrict__ dMq, float* __restrict__ dwhat, float* __restrict__ dshat, float* __restrict__ duq)
{
    const int b = blockIdx.y;
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int slab_sz = 2 * 32 * dl + dl + 32, ncols = 16 + 4 * nb1;
    if (x >= slab_sz) return;
    // which output element this is (columns that hold no word are never summed: their slab entries are not written)
    float* out = nullptr;
    if (x < 64 * dl) {
        const int y = x < 32 * dl ? x : x - 32 * dl, c = y / dl, d = y % dl, w = col_word(c, nb1);
        if (c < ncols && w < Nq) out = (x < 32 * dl ? dMq : dwhat) + ((size_t)b * Nq + w) * dl + d;
    } else if (x < 64 * dl + dl) out = dshat + (size_t)b * dl + (x - 64 * dl);
    else { const int c = x - 64 * dl - dl, w = col_word(c, nb1); if (c < ncols && w < Nq) out = duq + (size_t)b * Nq + w; }
    if (!out) return;
    const int s0 = row_ptr[b * L], s1 = row_ptr[(b + 1) * L];
    float s = 0.f;
    if (s1 > s0) {
        const int g_lo = s0 / cells_per_range, g_hi = (s1 - 1) / cells_per_range;
        for (int g = g_lo; g <= g_hi; ++g) s += slab[(size_t)(g + b) * slab_sz + x];
    }
    *out = s;
}

// ---- launchers --------------------------------------------------------------------------------------------------
static int attn_num_cus()
{
    static int n = 0;
    if (n == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n = v;
        else n = 256;
    }
    return n;
}
// equal ranges of the cell list, one per workgroup slot; whole tiles (4 cells), at least `min_cells`
static int range_cells(int N, int slots, int min_cells)
{
    int c = cdiv(cdiv(N, slots), 4) * 4;
    return c < min_cells ? min_cells : c;
}
int content_attn_bwd_range_cells(int N) { return range_cells(N, 2 * attn_num_cus(), 16); }      // two 256-thread workgroups per CU, whole rounds

template <int DL, int WS, bool EXACT>
static int fwd_t(hipStream_t st, const float* chat, const int* cells, const int* row_ptr, int N, int B, int L, int C,
                 const float* Mq, const float* uq, const float* what, const float* shat, const float* qmask,
                 float* cc_rows, float* cc_mean, int dl, int Nq, bool rows_bf16)
{
    (void)B;
    static const int wgs_per_cu = getenv("SMIN_ATTN_FWD_WGS") ? atoi(getenv("SMIN_ATTN_FWD_WGS")) : 3;
    const int cpr = range_cells(N, wgs_per_cu * attn_num_cus(), 16);      // three 256-thread workgroups per CU (118 registers, 36 KB of LDS each; four measured no faster: the launch is bound by HBM)
    const dim3 grid(cdiv(N, cpr));
    const size_t lds = fwd_lds_bytes<DL>();
    const float scale = 1.0f / sqrtf((float)dl);
    if (rows_bf16)                                                // (rows + mean: the only bf16-rows combination a host asks for)
        hipLaunchKernelGGL((content_attn_fwd_kernel<DL, WS, 2, true, EXACT>), grid, dim3(256), lds, st, chat, cells, row_ptr, L, C, Mq, uq, what, shat, qmask,
                           cc_rows, cc_mean, dl, Nq, N, cpr, scale);
    else if (cc_rows && cc_mean)
        hipLaunchKernelGGL((content_attn_fwd_kernel<DL, WS, 1, true, EXACT>), grid, dim3(256), lds, st, chat, cells, row_ptr, L, C, Mq, uq, what, shat, qmask,
                           cc_rows, cc_mean, dl, Nq, N, cpr, scale);
    else if (cc_rows)
        hipLaunchKernelGGL((content_attn_fwd_kernel<DL, WS, 1, false, EXACT>), grid, dim3(256), lds, st, chat, cells, row_ptr, L, C, Mq, uq, what, shat, qmask,
                           cc_rows, cc_mean, dl, Nq, N, cpr, scale);
    else
        hipLaunchKernelGGL((content_attn_fwd_kernel<DL, WS, 0, true, EXACT>), grid, dim3(256), lds, st, chat, cells, row_ptr, L, C, Mq, uq, what, shat, qmask,
                           cc_rows, cc_mean, dl, Nq, N, cpr, scale);
    SMIN_LAUNCH_CHECK();
    return 0;
}

// kernels exist for these (feature width, word steps) pairs: the production width 128 with 4 / 5 / 6 / 8 steps of four
// words (Nq <= 16 / 20 / 24 / 32), narrower widths with 4 or 8; shapes with dl below the width or fewer than four clips
// run the general form at the widest word count
#define SMIN_ATTN_DISPATCH(FN, ...)                                                                        \
    do {                                                                                                    \
        const int nws__ = (Nq + 3) / 4;                                                                      \
        if (!(C == 4 && (dl == 16 || dl == 32 || dl == 64 || dl == 128))) {                                   \
            if (dl <= 16) return FN<16, 8, false>(__VA_ARGS__);                                                \
            if (dl <= 32) return FN<32, 8, false>(__VA_ARGS__);                                                \
            if (dl <= 64) return FN<64, 8, false>(__VA_ARGS__);                                                \
            return FN<128, 8, false>(__VA_ARGS__);                                                          \
        }                                                                                                   \
        if (dl == 16) return nws__ <= 4 ? FN<16, 4, true>(__VA_ARGS__) : FN<16, 8, true>(__VA_ARGS__);         \
        if (dl == 32) return nws__ <= 4 ? FN<32, 4, true>(__VA_ARGS__) : FN<32, 8, true>(__VA_ARGS__);         \
        if (dl == 64) return nws__ <= 4 ? FN<64, 4, true>(__VA_ARGS__) : FN<64, 8, true>(__VA_ARGS__);         \
        if (nws__ <= 4) return FN<128, 4, true>(__VA_ARGS__);                                                  \
        if (nws__ == 5) return FN<128, 5, true>(__VA_ARGS__);                                                  \
        if (nws__ == 6) return FN<128, 6, true>(__VA_ARGS__);                                                  \
        return FN<128, 8, true>(__VA_ARGS__);                                                               \
    } while (0)

int launch_content_attn_fwd(hipStream_t st, const float* chat, const int* cells, const int* row_ptr, int N, int B, int L, int C,
                            const float* Mq, const float* uq, const float* what, const float* shat, const float* qmask,
                            float* cc_rows, float* cc_mean, int dl, int Nq)
{
    if (N <= 0) return 0;
    SMIN_ATTN_DISPATCH(fwd_t, st, chat, cells, row_ptr, N, B, L, C, Mq, uq, what, shat, qmask, cc_rows, cc_mean, dl, Nq, false);
}

// the same with the rows stored as bf16 (cc_rows_h [N*C][dl] 16-bit) beside the fp32 clip mean
int launch_content_attn_fwd_h(hipStream_t st, const float* chat, const int* cells, const int* row_ptr, int N, int B, int L, int C,
                              const float* Mq, const float* uq, const float* what, const float* shat, const float* qmask,
                              unsigned short* cc_rows_h, float* cc_mean, int dl, int Nq)
{
    if (N <= 0) return 0;
    float* cc_rows = reinterpret_cast<float*>(cc_rows_h);
    SMIN_ATTN_DISPATCH(fwd_t, st, chat, cells, row_ptr, N, B, L, C, Mq, uq, what, shat, qmask, cc_rows, cc_mean, dl, Nq, true);
}

// slabs: one per (range, sample) segment, indexed range + sample
size_t content_attn_bwd_ws_floats(int N, int B, int dl)
{
    const int cpr = content_attn_bwd_range_cells(N > 0 ? N : 1);
    return ((size_t)cdiv(N > 0 ? N : 1, cpr) + B + 1) * ((size_t)64 * dl + dl + 32) + 64;
}

template <int DL, int WS, bool MEAN2, bool PERCELL, bool EXACT>
static int bwd_v(hipStream_t st, const float* chat, const float* dcchat, const int* cells, const int* row_ptr, int N, int B, int L, int C,
                 const float* Mq, const float* uq, const float* what, const float* shat, const float* qmask,
                 float* dchat, float* dMq, float* duq, float* dwhat, float* dshat, float* ws, int dl, int Nq, int g_per_cell, float gscale,
                 const float* dmean2, float mscale)
{
    const int cpr = content_attn_bwd_range_cells(N);
    static bool attr_set = false;                                // > 64 KB of dynamic LDS needs the opt-in, once per instantiation
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&content_attn_bwd_kernel<DL, WS, MEAN2, PERCELL, EXACT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)(bwd_lds_bytes<DL, WS>()));
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const size_t lds = bwd_lds_bytes<DL, WS>();
    hipLaunchKernelGGL((content_attn_bwd_kernel<DL, WS, MEAN2, PERCELL, EXACT>), dim3(cdiv(N, cpr)), dim3(256), lds, st, chat, dcchat, cells, row_ptr, L, C,
                       Mq, uq, what, shat, qmask, dchat, ws, dl, Nq, N, cpr, 1.0f / sqrtf((float)dl), g_per_cell, gscale, dmean2, mscale);
    SMIN_LAUNCH_CHECK();
    const int slab_sz = 64 * dl + dl + 32;
    hipLaunchKernelGGL(content_attn_reduce_kernel, dim3(cdiv(slab_sz, 256), B), dim3(256), 0, st, ws, row_ptr, L, dl, Nq, cpr, WS - 4, dMq, dwhat, dshat, duq);
    SMIN_LAUNCH_CHECK();
    return 0;
}

template <int DL, int WS, bool EXACT>
static int bwd_t(hipStream_t st, const float* chat, const float* dcchat, const int* cells, const int* row_ptr, int N, int B, int L, int C,
                 const float* Mq, const float* uq, const float* what, const float* shat, const float* qmask,
                 float* dchat, float* dMq, float* duq, float* dwhat, float* dshat, float* ws, int dl, int Nq, int g_per_cell, float gscale,
                 const float* dmean2, float mscale)
{
    if (dmean2)
        return bwd_v<DL, WS, true, false, EXACT>(st, chat, dcchat, cells, row_ptr, N, B, L, C, Mq, uq, what, shat, qmask, dchat, dMq, duq, dwhat, dshat, ws, dl, Nq, 0, gscale, dmean2, mscale);
    if (g_per_cell)
        return bwd_v<DL, WS, false, true, EXACT>(st, chat, dcchat, cells, row_ptr, N, B, L, C, Mq, uq, what, shat, qmask, dchat, dMq, duq, dwhat, dshat, ws, dl, Nq, 1, gscale, nullptr, mscale);
    return bwd_v<DL, WS, false, false, EXACT>(st, chat, dcchat, cells, row_ptr, N, B, L, C, Mq, uq, what, shat, qmask, dchat, dMq, duq, dwhat, dshat, ws, dl, Nq, 0, gscale, nullptr, mscale);
}

int launch_content_attn_bwd(hipStream_t st, const float* chat, const float* dcchat, const int* cells, const int* row_ptr, int N, int B, int L, int C,
                            const float* Mq, const float* uq, const float* what, const float* shat, const float* qmask,
                            float* dchat, float* dMq, float* duq, float* dwhat, float* dshat, float* ws, int dl, int Nq, int g_per_cell, float gscale,
                            const float* dmean2, float mscale)
{
    SMIN_ATTN_DISPATCH(bwd_t, st, chat, dcchat, cells, row_ptr, N, B, L, C, Mq, uq, what, shat, qmask, dchat, dMq, duq, dwhat, dshat, ws, dl, Nq, g_per_cell, gscale, dmean2, mscale);
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
    SMIN_REQUIRE(dl % 16 == 0 && dl >= 16 && dl <= 128 && C >= 2 && C <= 4 && Nq >= 1 && Nq <= 32);
    if (N == 0) return 0;
    SMIN_REQUIRE(cc || ccmean);
    ProfScope prof((hipStream_t)stream, SMIN_PROF_ATTN_FWD);
    return launch_content_attn_fwd((hipStream_t)stream, chat, cells, row_ptr, N, B, L, C, Mq, uq, what, shat, qmask, cc, ccmean, dl, Nq);
}

extern "C" int smin_content_attn_fwd_cch(void* stream, const float* chat, const int32_t* cells, const int32_t* row_ptr,
                                         int N, int B, int L, int C, int dl, int Nq,
                                         const float* Mq, const float* uq, const float* what, const float* shat, const float* qmask,
                                         uint16_t* cc_h, float* ccmean)
{
    SMIN_REQUIRE(dl % 16 == 0 && dl >= 16 && dl <= 128 && C >= 2 && C <= 4 && Nq >= 1 && Nq <= 32);
    if (N == 0) return 0;
    SMIN_REQUIRE(cc_h && ccmean);
    return launch_content_attn_fwd_h((hipStream_t)stream, chat, cells, row_ptr, N, B, L, C, Mq, uq, what, shat, qmask, cc_h, ccmean, dl, Nq);
}

#ifdef SMIN_ATTN_STAMPS
extern "C" int smin_debug_attn_stamps(unsigned long long* host_out, int n)
{
    if (n > 2 * 4 * 8 * 16) n = 2 * 4 * 8 * 16;
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_attn_stamps), sizeof(unsigned long long) * n);
}
#endif

extern "C" size_t smin_content_attn_bwd_workspace_bytes(int N, int B, int C, int dl)
{
    (void)C;
    return sizeof(float) * (content_attn_bwd_ws_floats(N, B, dl) + 64);
}

// Gradients dcc [N*C][dl] and/or dccmean [N][dl] (either may be NULL, not both) -> dchat and the per-sample word-side
// gradients.  Masked cells (m == 0) get dchat = 0.
extern "C" int smin_content_attn_bwd(void* stream, const float* dcc, const float* dccmean, const float* chat,
                                     const int32_t* cells, const int32_t* row_ptr, int N, int B, int L, int C, int dl, int Nq,
                                     const float* Mq, const float* uq, const float* what, const float* shat, const float* qmask,
                                     float* dchat, float* dMq, float* duq, float* dwhat, float* dshat, void* ws, size_t ws_bytes)
{
    hipStream_t st = (hipStream_t)stream;
    SMIN_REQUIRE(dl % 16 == 0 && dl >= 16 && dl <= 128 && C >= 2 && C <= 4 && Nq >= 1 && Nq <= 32);
    if (N == 0) return 0;
    SMIN_REQUIRE(dcc || dccmean);
    SMIN_REQUIRE(ws_bytes >= smin_content_attn_bwd_workspace_bytes(N, B, C, dl));
    float* aws = reinterpret_cast<float*>(ws);
    const float* g = dcc; const float* g2 = nullptr; int per_cell = 0; float gscale = 1.0f;
    if (dcc && dccmean) g2 = dccmean;                               // both consumers: summed while the rows are loaded
    else if (!dcc) { g = dccmean; per_cell = 1; gscale = 1.0f / C; }
    ProfScope prof(st, SMIN_PROF_ATTN_BWD);
    return launch_content_attn_bwd(st, chat, g, cells, row_ptr, N, B, L, C, Mq, uq, what, shat, qmask, dchat, dMq, duq, dwhat, dshat, aws, dl, Nq,
                                   per_cell, gscale, g2, 1.0f / C);
}
