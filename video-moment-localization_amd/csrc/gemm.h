// fp32 MFMA GEMM engines for the SMIN hot path (gfx950).
//
// Two kernels, both built on v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD):
//   gemm_nt : C[M,N] = A[M,K] * B[N,K]^T            forward maps and input-gradients
//   gemm_tn : C[I,J] = sum_m A[m,I]^T * B[m,J]      weight-gradients, split over m (deterministic slabs)
// A and B are "virtual matrices": small functors that produce float4 elements on the fly, so masks,
// the boundary pair product f_b[i]*f_b[j] and the broadcast clip gradients never touch HBM.
// Tiles: 128x128 output per 256-thread workgroup (4 waves as 2x2, each 64x64 = 2x2 MFMA tiles),
// K-step 32, register-staged double-buffered LDS, one barrier per K-step.  All global loads are
// unconditional (indices are clamped, out-of-range values are selected away) so that hipcc keeps them
// in flight together instead of branching and waiting per load.  The accumulator tile goes through LDS
// once more at the end so that epilogues read/write HBM in whole 512-byte row segments (float4 per lane).
#pragma once
#include "common.h"

namespace smin {

// ------------------------------------------------------------------ virtual matrices
// Protocol:  Row row(int r) const                   (r in range, guaranteed by the caller)
//            float4 at(const Row&, int c) const     (c % 4 == 0, c in range; must be branch-free)
//            Key key(int r) / Row resolve(Key, r)   row() split in two: key() only issues the descriptor loads,
//                                                   resolve() does the dependent pointer arithmetic later

struct PlainMat {                       // row-major [rows][ld]
    const float* p; int ld;
    struct Row { const float* p; };
    struct Key {};
    __device__ __forceinline__ Row row(int r) const { return Row{p + (size_t)r * ld}; }
    __device__ __forceinline__ Key key(int) const { return Key{}; }
    __device__ __forceinline__ Row resolve(const Key&, int r) const { return row(r); }
    __device__ __forceinline__ float4 at(const Row& r, int c) const { return ldg4(r.p + c); }
};

// Row (b, pos) of the LSTM output Hout [B][Nq][2H] as the recurrence saw it one step earlier: direction 0 -> columns [0, H) of
// position pos - 1, direction 1 -> columns [H, 2H) of position pos + 1, zero where that position does not exist (the operand of
// dW_hh = dG^T h_prev; a kernel of its own wrote this matrix out before)
struct ShiftRowsMat {
    const float* p; int Nq, H, d;
    struct Row { const float* p; float m; };
    struct Key {};
    __device__ __forceinline__ Row row(int r) const {
        const int pos = r % Nq, q = d == 0 ? pos - 1 : pos + 1;
        const bool ok = q >= 0 && q < Nq;
        return Row{p + (size_t)(ok ? r + (d == 0 ? -1 : 1) : r) * 2 * H + (size_t)d * H, ok ? 1.f : 0.f};
    }
    __device__ __forceinline__ Key key(int) const { return Key{}; }
    __device__ __forceinline__ Row resolve(const Key&, int r) const { return row(r); }
    __device__ __forceinline__ float4 at(const Row& r, int c) const { return f4scale(ldg4(r.p + c), r.m); }
};

struct MaskedRowsMat {                  // row n of [N][ld] scaled by the cell mask m[n]
    const float* p; int ld; const int* cells;
    struct Row { const float* p; float m; };
    struct Key { int m; };
    __device__ __forceinline__ Row row(int r) const { return Row{p + (size_t)r * ld, (float)cells[4 * (size_t)r + 3]}; }
    __device__ __forceinline__ Key key(int r) const { return Key{cells[4 * (size_t)r + 3]}; }
    __device__ __forceinline__ Row resolve(const Key& k, int r) const { return Row{p + (size_t)r * ld, (float)k.m}; }
    __device__ __forceinline__ float4 at(const Row& r, int c) const { return f4scale(ldg4(r.p + c), r.m); }
};

// X[n] = [ f_b[b,i,:] * f_b[b,j,:]  |  mean_c f_c[n,:] ]   (models.py:292-301), width 2D
struct PairMeanMat {
    const float* fb; const float* fcmean; const int* cells; int L, D;
    struct Row { const float* bi; const float* bj; const float* cm; };
    struct Key { int4 c; };
    __device__ __forceinline__ Row row(int r) const { return resolve(key(r), r); }
    __device__ __forceinline__ Key key(int r) const { return Key{*reinterpret_cast<const int4*>(cells + 4 * (size_t)r)}; }
    __device__ __forceinline__ Row resolve(const Key& k, int r) const {
        const float* base = fb + (size_t)k.c.x * L * D;
        return Row{base + (size_t)k.c.y * D, base + (size_t)k.c.z * D, fcmean + (size_t)r * D};
    }
    __device__ __forceinline__ float4 at(const Row& r, int c) const {
        const bool first = c < D;
        const float4 a = ldg4(first ? r.bi + c : r.cm + (c - D));
        const float4 b = ldg4(r.bj + (first ? c : 0));
        return first ? f4mul(a, b) : a;
    }
};

// Effective gradient of the content-unit output, row = n*C + c:
//   dout[n,c,:] = m[n] * ( dfc_out[n,c,:] (HAS_DFC) + dfcmean[n,:] / C )      (MASK: apply m)
template <bool MASK, bool HAS_DFC>
struct DoutEffMat {
    const float* dfc; const float* dmean; const int* cells; int C, D; float invC;
    struct Row { const float* a; const float* b; float m; };
    struct Key { int m; };
    __device__ __forceinline__ Row row(int r) const { return resolve(key(r), r); }
    __device__ __forceinline__ Key key(int r) const { return Key{MASK ? cells[4 * (size_t)(r / C) + 3] : 1}; }
    __device__ __forceinline__ Row resolve(const Key& k, int r) const {
        return Row{HAS_DFC ? dfc + (size_t)r * D : nullptr, dmean + (size_t)(r / C) * D, (float)k.m};
    }
    __device__ __forceinline__ float4 at(const Row& r, int c) const {
        float4 v = f4scale(ldg4(r.b + c), invC);
        if (HAS_DFC) v = f4add(v, ldg4(r.a + c));
        return MASK ? f4scale(v, r.m) : v;
    }
};

// up to four row-major [rows][w] matrices side by side along the contraction (or output) index
struct CatMat {
    const float* p[4]; int w;
    struct Row { size_t off; };
    struct Key {};
    __device__ __forceinline__ Row row(int r) const { return Row{(size_t)r * w}; }
    __device__ __forceinline__ Key key(int) const { return Key{}; }
    __device__ __forceinline__ Row resolve(const Key&, int r) const { return row(r); }
    __device__ __forceinline__ float4 at(const Row& r, int c) const {
        const int sg = (c >= w) + (c >= 2 * w) + (c >= 3 * w);
        const float* q = sg == 0 ? p[0] : sg == 1 ? p[1] : sg == 2 ? p[2] : p[3];
        return ldg4(q + r.off + (c - sg * w));
    }
};
__device__ __forceinline__ float4 ldg4_bf16(const unsigned short* p) {
    const uint2 v = *reinterpret_cast<const uint2*>(p);
    return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16), __uint_as_float(v.y & 0xffff0000u));
}
// [x1 | fcmean], both [rows][w], x1 stored as bf16 (the moment unit's left operand when the contractions round their operands to
// bf16 anyway -- smin_set_gemm_mode(2): the stored values are exactly what the loader would have produced from fp32 storage)
struct PairCatH {
    const unsigned short* x1; const float* fcmean; int w;
    struct Row { size_t off; };
    struct Key {};
    __device__ __forceinline__ Row row(int r) const { return Row{(size_t)r * w}; }
    __device__ __forceinline__ Key key(int) const { return Key{}; }
    __device__ __forceinline__ Row resolve(const Key&, int r) const { return row(r); }
    __device__ __forceinline__ float4 at(const Row& r, int c) const {
        if (c >= w) return ldg4(fcmean + r.off + (c - w));
        return ldg4_bf16(x1 + r.off + c);
    }
};
// bf16-stored operands of the same shapes as PlainMat / CatMat (contraction-only tensors under smin_set_gemm_mode(2), see PairCatH)
struct PlainMatH {
    const unsigned short* p; int ld;
    struct Row { const unsigned short* p; };
    struct Key {};
    __device__ __forceinline__ Row row(int r) const { return Row{p + (size_t)r * ld}; }
    __device__ __forceinline__ Key key(int) const { return Key{}; }
    __device__ __forceinline__ Row resolve(const Key&, int r) const { return row(r); }
    __device__ __forceinline__ float4 at(const Row& r, int c) const { return ldg4_bf16(r.p + c); }
};
struct CatMatH {
    const unsigned short* p[4]; int w;
    struct Row { size_t off; };
    struct Key {};
    __device__ __forceinline__ Row row(int r) const { return Row{(size_t)r * w}; }
    __device__ __forceinline__ Key key(int) const { return Key{}; }
    __device__ __forceinline__ Row resolve(const Key&, int r) const { return row(r); }
    __device__ __forceinline__ float4 at(const Row& r, int c) const {
        const int sg = (c >= w) + (c >= 2 * w) + (c >= 3 * w);
        const unsigned short* q = sg == 0 ? p[0] : sg == 1 ? p[1] : sg == 2 ? p[2] : p[3];
        return ldg4_bf16(q + r.off + (c - sg * w));
    }
};
// ------------------------------------------------------------------ helpers
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ float4 f4sel(bool ok, float4 v) {
    return make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
}

constexpr int GEMM_LDW = 68;            // row stride (floats) of a wave's staged 32 x 64 accumulator chunk

// Visit a wave's staged chunk row-major, one float4 per lane: 16 lanes cover one 256-byte row segment.
template <class F>
__device__ __forceinline__ void chunk_rows_f4(const float* Ws, int row0, int col0, int ncols, int M, int N, int lane, F f)
{
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int idx = lane + 64 * it, r = idx >> 4, c4 = (idx & 15) * 4;
        const int row = row0 + r, col = col0 + c4;
        if (c4 < ncols && row < M && col < N) f(row, col, ldg4(Ws + r * GEMM_LDW + c4));
    }
}
// Two-phase visit for epilogues that add tensors from HBM: pre(row, col) -> A requests an element's addends (called for all
// eight elements of the lane first, unconditionally, with clamped indices), post(row, col, v, a) consumes them.  With a
// single-phase lambda the loads of element k+1 sit behind the store of element k (hipcc cannot move a load above a store
// through unrelated pointers), so a chunk paid eight HBM round trips in a row: K = 128 contractions with a residual ran at
// 36 TF against 97 TF for the same shape with a plain store.
template <class A, class PRE, class POST>
__device__ __forceinline__ void chunk_rows_f4_pre(const float* Ws, int row0, int col0, int ncols, int M, int N, int lane, PRE pre, POST post)
{
    A add[8];
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int idx = lane + 64 * it, r = idx >> 4, c4 = (idx & 15) * 4;
        add[it] = pre(min(row0 + r, M - 1), min(col0 + c4, N - 4));
    }
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int idx = lane + 64 * it, r = idx >> 4, c4 = (idx & 15) * 4;
        const int row = row0 + r, col = col0 + c4;
        if (c4 < ncols && row < M && col < N) post(row, col, ldg4(Ws + r * GEMM_LDW + c4), add[it]);
    }
}
// Same chunk as 8 groups of 4 consecutive rows (row0 % 4 == 0): f(first_row, col, v[4]) -- one lane owns the 4 clips of a cell.
template <class F>
__device__ __forceinline__ void chunk_quads_f4(const float* Ws, int row0, int col0, int ncols, int M, int N, int lane, F f)
{
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int idx = lane + 64 * it, g = idx >> 4, c4 = (idx & 15) * 4;
        const int row = row0 + 4 * g, col = col0 + c4;
        if (c4 < ncols && row < M && col < N) {
            const float4 v[4] = {ldg4(Ws + (4 * g) * GEMM_LDW + c4), ldg4(Ws + (4 * g + 1) * GEMM_LDW + c4),
                                 ldg4(Ws + (4 * g + 2) * GEMM_LDW + c4), ldg4(Ws + (4 * g + 3) * GEMM_LDW + c4)};
            f(row, col, v);
        }
    }
}

// ------------------------------------------------------------------ NT kernel
// Epilogue protocol: void chunk(const float* Ws, int row0, int col0, int ncols, int M, int N, int lane) const
//   Ws is a wave-private 32 x ncols (64 or 32) piece of the accumulator tile in LDS (row stride GEMM_LDW) whose
//   first element is C[row0][col0]; rows >= M / cols >= N must be skipped.  No workgroup barrier is involved:
//   each wave transposes its own accumulators through its own LDS region and streams them out.
// Tail handling: a grid of equal tiles runs ceil(tiles / slots) rounds, the last one mostly empty (3152 tiles on
// 512 slots = 6.16 -> 7 rounds).  The first `main_tiles_m` row tiles are 128 rows; the remaining rows are cut into
// 32-row "mini" tiles (one 32x32 MFMA tile per wave) so that the last round costs a quarter of a full one.
template <bool MINI, bool KFULL, class AM, class BM_, class EP>
__device__ __forceinline__ void gemm_nt_body(float* smem, const AM& am, const BM_& bm, const EP& ep, int M, int N, int K,
                                             int row_base, int col_base)
{
    // K-step 16: 40 KB of LDS per workgroup and <= 168 VGPRs -> 3 workgroups (3 waves per SIMD) per CU, which keeps the
    // matrix pipe fed while one workgroup sits at its barrier or in its epilogue.
    constexpr int BM = 128, BN = 128, BK = 16, LDT = BK + 4, KQ = BK / 4, RPP = 256 / KQ, NP = BM / RPP;
    float* As = smem;
    float* Bs = smem + 2 * BM * LDT;
    const int tile_rows = MINI ? 32 : BM;
    // thread -> (row, 16-byte chunk) of a staging pass: the 8 lanes that one ds_write_b128 group serves write rows R and
    // R+4 (bank offsets 0 and 16 mod 32); with consecutive rows (20 dwords apart) 4 of the 16 banks would collide.
    const int t = threadIdx.x, u = (t & 63) >> 2, kq = (t & 3) * 4;
    const int lr = 16 * (t >> 6) + ((u >> 1) & 3) + 8 * (u >> 3) + 4 * (u & 1);

    // out-of-range rows are clamped: they only feed accumulator rows/columns that are never stored
    typename AM::Row arow[NP];
    typename BM_::Row brow[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        arow[p] = am.row(min(row_base + (MINI ? (lr & 31) : lr + RPP * p), M - 1));
        brow[p] = bm.row(min(col_base + lr + RPP * p, N - 1));
    }

    const int wave = t >> 6, lane = t & 63, wm = wave >> 1, wn = wave & 1, l31 = lane & 31, h = lane >> 5;
    const int wrow = MINI ? 0 : wm * 64, wcol = MINI ? wave * 32 : wn * 64;
    constexpr int NMI = MINI ? 1 : 2;
    f32x16 acc[NMI][NMI];
#pragma unroll
    for (int a = 0; a < NMI; ++a)
#pragma unroll
        for (int b = 0; b < NMI; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // Two register stages: the loads of tile kt+3 are issued while tile kt is computed and consumed two iterations later.
    struct Stage { float4 a[NP], b[NP]; };
    const int nk = (K + BK - 1) / BK;
    auto g_load = [&](Stage& st, int kt) {
        const int k = min(kt, nk - 1) * BK + kq;                   // past-the-end tiles reload the last one (never used)
        if (KFULL) {
#pragma unroll
            for (int p = 0; p < NP; ++p) { st.a[p] = am.at(arow[p], k); st.b[p] = bm.at(brow[p], k); }
        } else {
            const bool kok = k < K;
            const int kc = min(k, K - 4);
#pragma unroll
            for (int p = 0; p < NP; ++p) { st.a[p] = f4sel(kok, am.at(arow[p], kc)); st.b[p] = f4sel(kok, bm.at(brow[p], kc)); }
        }
    };
    auto s_store = [&](const Stage& st, int buf) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            stg4(As + buf * BM * LDT + (lr + RPP * p) * LDT + kq, st.a[p]);
            stg4(Bs + buf * BN * LDT + (lr + RPP * p) * LDT + kq, st.b[p]);
        }
    };
    auto compute = [&](int cur) {
        const float* ab = As + cur * BM * LDT + (wrow + l31) * LDT + 4 * h;
        const float* bb = Bs + cur * BN * LDT + (wcol + l31) * LDT + 4 * h;
#pragma unroll
        for (int kg = 0; kg < BK / 8; ++kg) {
            // lane half h supplies k = 8kg+4h+q at MFMA step q: A and B use the same k permutation
            float4 a4[NMI], b4[NMI];
#pragma unroll
            for (int i = 0; i < NMI; ++i) { a4[i] = ldg4(ab + i * 32 * LDT + kg * 8); b4[i] = ldg4(bb + i * 32 * LDT + kg * 8); }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < NMI; ++i)
#pragma unroll
                    for (int j = 0; j < NMI; ++j) {
                        const float av = q == 0 ? a4[i].x : q == 1 ? a4[i].y : q == 2 ? a4[i].z : a4[i].w;
                        const float bv = q == 0 ? b4[j].x : q == 1 ? b4[j].y : q == 2 ? b4[j].z : b4[j].w;
                        acc[i][j] = mfma32(av, bv, acc[i][j]);
                    }
        }
    };

    Stage s0, s1;
    g_load(s0, 0);
    s_store(s0, 0);
    g_load(s0, 1);
    g_load(s1, 2);
    __syncthreads();
    int kt = 0;
    for (; kt + 2 < nk; kt += 2) {                                 // tiles kt+1 and kt+2 exist: no conditions inside
        compute(0);                                                // tile kt
        s_store(s0, 1);                                            // tile kt+1
        g_load(s0, kt + 3);
        __syncthreads();
        compute(1);                                                // tile kt+1
        s_store(s1, 0);                                            // tile kt+2
        g_load(s1, kt + 4);
        __syncthreads();
    }
    compute(0);                                                    // tile kt (nk - kt is 1 or 2)
    if (kt + 1 < nk) {
        s_store(s0, 1);
        __syncthreads();
        compute(1);
    }
    __syncthreads();                                               // the accumulator staging below reuses the operand buffers

    // Epilogue: C/D layout of a 32x32 MFMA is col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).  Each wave
    // writes one 32-row slab of its tile to its private LDS region and reads it back row-major (float4 per lane).
    float* Ws = smem + wave * (32 * GEMM_LDW);
#pragma unroll
    for (int mi = 0; mi < NMI; ++mi) {
#pragma unroll
        for (int ni = 0; ni < NMI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                Ws[((r & 3) + 8 * (r >> 2) + 4 * h) * GEMM_LDW + ni * 32 + l31] = acc[mi][ni][r];
        __builtin_amdgcn_wave_barrier();
        ep.chunk(Ws, row_base + wrow + mi * 32, col_base + wcol, 32 * NMI, min(M, row_base + tile_rows), N, lane);
        __builtin_amdgcn_wave_barrier();
    }
}

template <bool KFULL, class AM, class BM_, class EP>
__global__ __launch_bounds__(256, 3)
void gemm_nt_kernel(AM am, BM_ bm, EP ep, int M, int N, int K, int main_tiles_m, int tiles_n, int main_blocks)
{
    __shared__ __attribute__((aligned(16))) float smem[2 * (128 + 128) * 20];      // 40,960 B >= 4 waves * 32 * GEMM_LDW * 4
    if ((int)blockIdx.x < main_blocks) {
        // XCD-aware tile order: blocks b and b+8 share an XCD (round-robin dispatch), so the tiles_n column
        // tiles of one row tile (which re-read the same A rows) are consecutive slots of one XCD's L2.
        const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
        const int tn = slot % tiles_n, tm = (slot / tiles_n) * 8 + xcd;
        if (tm >= main_tiles_m) return;
        gemm_nt_body<false, KFULL>(smem, am, bm, ep, M, N, K, tm * 128, tn * 128);
    } else {
        const int id = blockIdx.x - main_blocks;
        gemm_nt_body<true, KFULL>(smem, am, bm, ep, M, N, K, main_tiles_m * 128 + (id / tiles_n) * 32, (id % tiles_n) * 128);
    }
}

// ------------------------------------------------------------------ NT kernel on the bf16 matrix cores
// fp32 operands are split on the fly into bf16 pieces and every 16-deep block is accumulated with
// v_mfma_f32_32x32x16_bf16 (fp32 accumulate, 16x the rate of the fp32 MFMA).  NPROD selects the arithmetic:
//   6  x = hi + mid + lo EXACTLY (3 x 8 mantissa bits), products  lo*hi + hi*lo + mid*mid + mid*hi + hi*mid + hi*hi, smallest first.
//      The three dropped cross terms are <= 2^-26 relative per product, below fp32's own product rounding (2^-24): an fp32
//      contraction computed on the bf16 cores ("f32e" mode; the same idea as library BF16x9 fp32 emulation, minus the three
//      terms that cannot reach an fp32 result), at 16/6 = 2.7x the fp32 matrix rate;
//   3  x ~ hi + lo, products  lo*hi + hi*lo + hi*hi: ~2^-16 relative per product (~1e-5 on a dot product), 5.3x the fp32 rate;
//   1  plain bf16: operands rounded once, one product, ~4e-3 relative per product.
// Same tiling, loaders and epilogues as gemm_nt_body.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NIMG>
__device__ __forceinline__ void split_bf16x4(float4 v, uint2 (&out)[NIMG])
{
    const float x[4] = {v.x, v.y, v.z, v.w};
    unsigned short pc[NIMG][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float r = x[q];
#pragma unroll
        for (int i = 0; i < NIMG; ++i) {
            const __bf16 b = (__bf16)r;
            pc[i][q] = __builtin_bit_cast(unsigned short, b);
            r -= (float)b;                                          // exact: the difference of a float and its leading bits
        }
    }
#pragma unroll
    for (int i = 0; i < NIMG; ++i)
        out[i] = make_uint2((unsigned)pc[i][0] | ((unsigned)pc[i][1] << 16), (unsigned)pc[i][2] | ((unsigned)pc[i][3] << 16));
}
__device__ __forceinline__ f32x16 mfma_bf16(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// acc[i][j] += a[i] * b[j] over the pieces, smallest terms first.  Products in the outer loop, the NI x NJ accumulator tiles in the
// inner one: consecutive MFMAs then write different accumulators (six in a row into one accumulator made every MFMA wait for its
// predecessor: 52 % matrix-pipe occupancy in the PMC pass).
template <int NPROD, int NIMG, int NI, int NJ>
__device__ __forceinline__ void mfma_split_tiles(const bf16x8 (&a)[NI][NIMG], const bf16x8 (&b)[NJ][NIMG], f32x16 (&acc)[NI][NJ])
{
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};      // piece indices, smallest product first
#pragma unroll
    for (int p = 6 - NPROD; p < 6; ++p) {
        // (NPROD = 3 uses the last three products of the list, NPROD = 1 the last)
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = mfma_bf16(a[i][PA[p]], b[j][PB[p]], acc[i][j]);
    }
}
constexpr int split_images(int nprod) { return nprod == 6 ? 3 : (nprod == 3 ? 2 : 1); }

template <int NPROD, bool MINI, bool KFULL, class AM, class BM_, class EP>
__device__ __forceinline__ void gemm_nt_x3_body(float* smem, const AM& am, const BM_& bm, const EP& ep, int M, int N, int K,
                                                int row_base, int col_base)
{
    // LDS: NIMG bf16 images per operand and buffer, rows of 16 k-values padded to 24 (48 B: the 16-byte fragment reads of
    // 16 consecutive rows fall on 16 different slots).  2 buffers x 2 NIMG x 128 x 48 B = 24,576 / 49,152 / 73,728 B.
    constexpr int NIMG = split_images(NPROD);
    // NPROD = 6: un-padded 32-byte rows with the two 16-byte halves of rows 8..15 (mod 16) swapped -- 49,152 B, three workgroups per CU
    constexpr bool SWZ = NPROD == 6;
    constexpr int BM = 128, BK = 16, RS = SWZ ? 16 : 24, KQ = BK / 4, RPP = 256 / KQ, NP = BM / RPP, IMG = BM * RS, BUF = 2 * NIMG * IMG;
    unsigned short* lds = reinterpret_cast<unsigned short*>(smem);
    const int tile_rows = MINI ? 32 : BM;
    const int t = threadIdx.x, lr = t / KQ, kq = (t % KQ) * 4;

    typename AM::Row arow[NP];
    typename BM_::Row brow[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        arow[p] = am.row(min(row_base + (MINI ? (lr & 31) : lr + RPP * p), M - 1));
        brow[p] = bm.row(min(col_base + lr + RPP * p, N - 1));
    }
    const int wave = t >> 6, lane = t & 63, wm = wave >> 1, wn = wave & 1, l31 = lane & 31, h = lane >> 5;
    const int wrow = MINI ? 0 : wm * 64, wcol = MINI ? wave * 32 : wn * 64;
    constexpr int NMI = MINI ? 1 : 2;
    f32x16 acc[NMI][NMI];
#pragma unroll
    for (int a = 0; a < NMI; ++a)
#pragma unroll
        for (int b = 0; b < NMI; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // Two register stages: the loads of tile kt+3 are issued while tile kt is computed and are consumed two
    // iterations later -- one iteration of this kernel is far shorter than an HBM round trip.
    struct Stage { float4 a[NP], b[NP]; };
    const int nk = (K + BK - 1) / BK;
    auto g_load = [&](Stage& st, int kt) {
        const int k = min(kt, nk - 1) * BK + kq;                   // past-the-end tiles reload the last one (never used)
        if (KFULL) {
#pragma unroll
            for (int p = 0; p < NP; ++p) { st.a[p] = am.at(arow[p], k); st.b[p] = bm.at(brow[p], k); }
        } else {
            const bool kok = k < K;
            const int kc = min(k, K - 4);
#pragma unroll
            for (int p = 0; p < NP; ++p) { st.a[p] = f4sel(kok, am.at(arow[p], kc)); st.b[p] = f4sel(kok, bm.at(brow[p], kc)); }
        }
    };
    auto s_store = [&](const Stage& st, int buf) {
        unsigned short* base = lds + buf * BUF;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            uint2 pc[NIMG];
            const int o = (lr + RPP * p) * RS + (SWZ ? (kq ^ ((((lr + RPP * p) >> 3) & 1) << 3)) : kq);
            split_bf16x4<NIMG>(st.a[p], pc);
#pragma unroll
            for (int i = 0; i < NIMG; ++i) *reinterpret_cast<uint2*>(base + i * IMG + o) = pc[i];
            split_bf16x4<NIMG>(st.b[p], pc);
#pragma unroll
            for (int i = 0; i < NIMG; ++i) *reinterpret_cast<uint2*>(base + (NIMG + i) * IMG + o) = pc[i];
        }
    };
    auto compute = [&](int cur) {
        const unsigned short* base = lds + cur * BUF;
        const int hs = SWZ ? 8 * (h ^ ((l31 >> 3) & 1)) : 8 * h;              // (tile rows start at multiples of 32)
        const unsigned short* ab = base + (wrow + l31) * RS + hs;             // lane (row, h) holds k = 8h .. 8h+7
        const unsigned short* bb = base + NIMG * IMG + (wcol + l31) * RS + hs;
        bf16x8 af[NMI][NIMG], bf[NMI][NIMG];
#pragma unroll
        for (int i = 0; i < NMI; ++i)
#pragma unroll
            for (int q = 0; q < NIMG; ++q) {
                af[i][q] = *reinterpret_cast<const bf16x8*>(ab + q * IMG + i * 32 * RS);
                bf[i][q] = *reinterpret_cast<const bf16x8*>(bb + q * IMG + i * 32 * RS);
            }
        mfma_split_tiles<NPROD, NIMG, NMI, NMI>(af, bf, acc);
    };

    // One half-step = the matrix products of the resident tile, the split + LDS store of the next one (already in registers) and
    // the loads of a later tile into the freed registers.  Written without branches so that it is ONE scheduling region, and the
    // scheduler is told to issue the split's vector instructions between the matrix instructions (a 32x32x16 MFMA occupies the
    // matrix pipe for 32 cycles during which the wave would otherwise sit idle; left alone, hipcc emits all MFMAs first and the
    // ~120 split instructions after them).
    // NST register stages: a tile's loads are issued NST half-steps before its store (four stages measured no faster than two:
    // the operand latency is not what these kernels wait for).
    constexpr int NST = 2;
    Stage stg[NST];
    auto half_step = [&](int cur, Stage& st, int kt_next_load) {
        compute(cur);
        s_store(st, cur ^ 1);
        g_load(st, kt_next_load);
        constexpr int NMFMA = NMI * NMI * NPROD;
        __builtin_amdgcn_sched_group_barrier(0x100, 2 * NMI * NIMG, 0);          // the fragment reads
#pragma unroll
        for (int q = 0; q < NMFMA; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                   // one MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, MINI ? 10 : 5, 0);       // a slice of the split (VALU)
        }
    };
    // invariant at the top of the main loop: buffer 0 holds tile kt, stage (j % NST) holds tile kt + j for j = 1 .. NST
    g_load(stg[0], 0);
    s_store(stg[0], 0);
#pragma unroll
    for (int j = 1; j <= NST; ++j) g_load(stg[j % NST], j);
    __syncthreads();
    int kt = 0;
    for (; kt + NST < nk; kt += NST) {                              // tiles kt+1 .. kt+NST exist: no conditions inside
#pragma unroll
        for (int hs = 0; hs < NST; ++hs) {
            half_step(hs & 1, stg[(hs + 1) % NST], kt + hs + 1 + NST);   // tile kt+hs; stores kt+hs+1, reloads its stage
            __syncthreads();
        }
    }
#pragma unroll
    for (int hs = 0; hs < NST; ++hs) {                             // the last 1 .. NST tiles
        if (kt + hs < nk) {
            compute(hs & 1);
            if (kt + hs + 1 < nk) {
                s_store(stg[(hs + 1) % NST], (hs + 1) & 1);
                __syncthreads();
            }
        }
    }
    __syncthreads();                                               // the accumulator staging below reuses the operand buffers

    float* Ws = smem + wave * (32 * GEMM_LDW);
#pragma unroll
    for (int mi = 0; mi < NMI; ++mi) {
#pragma unroll
        for (int ni = 0; ni < NMI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                Ws[((r & 3) + 8 * (r >> 2) + 4 * h) * GEMM_LDW + ni * 32 + l31] = acc[mi][ni][r];
        __builtin_amdgcn_wave_barrier();
        ep.chunk(Ws, row_base + wrow + mi * 32, col_base + wcol, 32 * NMI, min(M, row_base + tile_rows), N, lane);
        __builtin_amdgcn_wave_barrier();
    }
}

template <int NPROD, bool KFULL, class AM, class BM_, class EP>
__global__ __launch_bounds__(256, 3)
void gemm_nt_x3_kernel(AM am, BM_ bm, EP ep, int M, int N, int K, int main_tiles_m, int tiles_n, int main_blocks)
{
    constexpr int LDS_FLOATS = 2 * 2 * split_images(NPROD) * 128 * (NPROD == 6 ? 16 : 24) / 2;
    __shared__ __attribute__((aligned(16))) float smem[LDS_FLOATS < 4 * 32 * GEMM_LDW ? 4 * 32 * GEMM_LDW : LDS_FLOATS];
    if ((int)blockIdx.x < main_blocks) {
        const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
        const int tn = slot % tiles_n, tm = (slot / tiles_n) * 8 + xcd;
        if (tm >= main_tiles_m) return;
        gemm_nt_x3_body<NPROD, false, KFULL>(smem, am, bm, ep, M, N, K, tm * 128, tn * 128);
    } else {
        const int id = blockIdx.x - main_blocks;
        gemm_nt_x3_body<NPROD, true, KFULL>(smem, am, bm, ep, M, N, K, main_tiles_m * 128 + (id / tiles_n) * 32, (id % tiles_n) * 128);
    }
}

// 0 = exact fp32 MFMA (default), 1 = split-bf16 (bf16x3), 2 = plain bf16 products, 3 = fp32 emulated on the bf16 cores
// (3-way split, 6 products); every contraction (NT and TN); set by smin_set_gemm_mode()
extern int g_gemm_mode;

constexpr int GEMM_SLOTS = 768;         // resident workgroups: 256 CUs x 3 (40 KB LDS, <= 168 VGPRs each)

template <class AM, class BM_, class EP>
static inline int launch_gemm_nt(hipStream_t st, const AM& am, const BM_& bm, const EP& ep, int M, int N, int K)
{
    if (M <= 0 || N <= 0) return 0;
    const int tiles_m = cdiv(M, 128), tiles_n = cdiv(N, 128);
    int main_tiles_m = tiles_m;
    const int total = tiles_m * tiles_n;
    const int full_rounds = total / GEMM_SLOTS;
    if (full_rounds >= 1 && total % GEMM_SLOTS != 0 && (total % GEMM_SLOTS) < (3 * GEMM_SLOTS) / 4)
        main_tiles_m = (full_rounds * GEMM_SLOTS) / tiles_n;         // whole rounds of 128-row tiles, rest as mini tiles
    else if (total * 3 <= GEMM_SLOTS)
        main_tiles_m = 0;                                            // fewer tiles than CUs: 32-row tiles spread the rows four times wider
    const int rem_rows = M - main_tiles_m * 128;
    const int mini_blocks = rem_rows > 0 ? cdiv(rem_rows, 32) * tiles_n : 0;
    const int main_blocks = cdiv(main_tiles_m, 8) * 8 * tiles_n;
#define SMIN_NT_SPLIT(NPROD)                                                                                                                   \
    do {                                                                                                                                       \
        if (K % 16 == 0)                                                                                                                       \
            hipLaunchKernelGGL((gemm_nt_x3_kernel<NPROD, true, AM, BM_, EP>), dim3(main_blocks + mini_blocks), dim3(256), 0, st, am, bm, ep, M, N, \
                               K, main_tiles_m, tiles_n, main_blocks);                                                                         \
        else                                                                                                                                   \
            hipLaunchKernelGGL((gemm_nt_x3_kernel<NPROD, false, AM, BM_, EP>), dim3(main_blocks + mini_blocks), dim3(256), 0, st, am, bm, ep, M,   \
                               N, K, main_tiles_m, tiles_n, main_blocks);                                                                      \
    } while (0)
    if (g_gemm_mode == 1) {
        SMIN_NT_SPLIT(3);
    } else if (g_gemm_mode == 2) {
        SMIN_NT_SPLIT(1);
    } else if (g_gemm_mode == 3) {
        SMIN_NT_SPLIT(6);
#undef SMIN_NT_SPLIT
    } else if (K % 16 == 0)
        hipLaunchKernelGGL((gemm_nt_kernel<true, AM, BM_, EP>), dim3(main_blocks + mini_blocks), dim3(256), 0, st, am, bm, ep, M, N, K,
                           main_tiles_m, tiles_n, main_blocks);
    else
        hipLaunchKernelGGL((gemm_nt_kernel<false, AM, BM_, EP>), dim3(main_blocks + mini_blocks), dim3(256), 0, st, am, bm, ep, M, N, K,
                           main_tiles_m, tiles_n, main_blocks);
    SMIN_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------ NT product of a FEW rows over a LONG contraction, split along K
// C [M][N] = A [M][K] B[N][K]^T for M of a few tile rows (the LSTM layers' input gradient: 1 280 x 512 x 2 048): the plain engine has
// 40 tiles for 256 CUs and every workgroup walks all of K alone, a chain of K / 16 load -> barrier -> MFMA steps (110 us on the
// dependent chain that closes the backward pass).  Here blockIdx.y cuts K into `splits` ranges: 32-row tiles x column tiles x splits
// workgroups write partial products into slab[split][M][N]; launch_reduce_slabs adds them in split order (fixed: reproducible).
// Exact-fp32 engine only (the bf16-core modes keep the plain call).
struct EpSlabStore {
    float* out;
    __device__ __forceinline__ void chunk(const float* Ws, int row0, int col0, int ncols, int M, int N, int lane) const {
        chunk_rows_f4(Ws, row0, col0, ncols, M, N, lane, [&](int row, int col, float4 v) { stg4(out + (size_t)row * N + col, v); });
    }
};
template <int UNUSED>                                          // (a template only for its linkage: the header is included by every source)
__global__ __launch_bounds__(256, 3)
void gemm_nt_splitk_kernel(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb, float* __restrict__ slab, int M, int N, int Ks, int tiles_n)
{
    __shared__ __attribute__((aligned(16))) float smem[2 * (128 + 128) * 20];
    const int z = blockIdx.y, id = blockIdx.x;
    gemm_nt_body<true, true>(smem, PlainMat{A + (size_t)z * Ks, lda}, PlainMat{B + (size_t)z * Ks, ldb}, EpSlabStore{slab + (size_t)z * M * N}, M, N, Ks,
                             (id / tiles_n) * 32, (id % tiles_n) * 128);
}
int launch_reduce_slabs(hipStream_t st, const float* slab, float* out, int n, int P);
// splits for launch_gemm_nt_splitk (1 = use the plain engine): enough workgroups for the chip, at least 16 K-steps per workgroup
static inline int nt_splitk_splits(int M, int N, int K)
{
    const int blocks = cdiv(M, 32) * cdiv(N, 128);
    if (g_gemm_mode != 0 || blocks * 2 > GEMM_SLOTS || K < 1024) return 1;
    int s = GEMM_SLOTS * 2 / blocks;
    while (s > 1 && (K % (16 * s) != 0 || K / s < 256)) --s;
    return s > 16 ? 16 : s;
}
static inline int launch_gemm_nt_splitk(hipStream_t st, const float* A, int lda, const float* B, int ldb, float* out, int M, int N, int K, int splits, float* slab)
{
    if (M <= 0 || N <= 0) return 0;
    const int tiles_n = cdiv(N, 128);
    hipLaunchKernelGGL(gemm_nt_splitk_kernel<0>, dim3(cdiv(M, 32) * tiles_n, splits), dim3(256), 0, st, A, lda, B, ldb, slab, M, N, K / splits, tiles_n);
    SMIN_LAUNCH_CHECK();
    return launch_reduce_slabs(st, slab, out, M * N, splits);
}

// ------------------------------------------------------------------ TN kernel (weight gradients)
// slab[z][I][J] = sum over rows m in split z of A[m][i] * B[m][j];  optional bias slab[z][I] = sum_m A[m][i].
template <class AM, class BM_, bool BIAS>
__global__ __launch_bounds__(256, 3)
void gemm_tn_kernel(AM am, BM_ bm, float* __restrict__ slab, float* __restrict__ bias_slab,
                    int Mrows, int I, int J, int rows_per_split, int tiles_i, int tiles_j, int splits)
{
    constexpr int BI = 128, BJ = 128, BK = 16, NP = BK / 8;          // 32 KB of LDS -> 3 workgroups per CU
    __shared__ __attribute__((aligned(16))) float smem[2 * BK * (BI + BJ)];
    float* As = smem;
    float* Bs = smem + 2 * BK * BI;

    // XCD-aware block order: blocks b and b+8 share an XCD (round-robin dispatch) and its L2.  All output tiles of one row
    // split read the same A / B rows, so a split's tiles are consecutive slots of ONE XCD: the rows are fetched from HBM
    // once per split instead of once per XCD that happens to hold one of its tiles (3.0x the algorithmic bytes before).
    const int tiles = tiles_i * tiles_j;
    const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
    const int z = (slot / tiles) * 8 + xcd, tile = slot % tiles;
    if (z >= splits) return;
    const int ti = tile % tiles_i, tj = tile / tiles_i;
    const int m_begin = z * rows_per_split;
    const int m_end = min(Mrows, m_begin + rows_per_split);
    const int t = threadIdx.x, lk = t >> 5, c4 = (t & 31) * 4;
    // columns past I / J are clamped: they only feed accumulator entries that are never stored
    const int ia = min(ti * BI + c4, I - 4), jb = min(tj * BJ + c4, J - 4);

    const int wave = t >> 6, lane = t & 63, wm = wave >> 1, wn = wave & 1, l31 = lane & 31, h = lane >> 5;
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    float4 bacc = f4zero();                                          // BIAS: this thread's rows of its four columns of A, summed

    // Two register stages (see gemm_nt_body): row descriptors and operands of tile kt+3 are requested while tile kt
    // is computed; the virtual-matrix rows change every tile here, so their lookups are part of the prefetch.
    struct Stage { float4 a[NP], b[NP]; };
    const int nk = (max(m_end - m_begin, 0) + BK - 1) / BK;
    // The rows of a virtual matrix change with every tile here, and their descriptors (cell lookups) are loads the
    // operand loads depend on.  key() issues those lookups one call ahead; resolve() turns them into pointers only
    // when the operand loads of that tile are issued, so no wave ever waits on a lookup it has just requested.
    typename AM::Key ka[NP];
    typename BM_::Key kb[NP];
    auto row_of = [&](int kt, int p) { return min(m_begin + min(kt, max(nk - 1, 0)) * BK + lk + 8 * p, Mrows - 1); };
    auto k_load = [&](int kt) {
#pragma unroll
        for (int p = 0; p < NP; ++p) { ka[p] = am.key(row_of(kt, p)); kb[p] = bm.key(row_of(kt, p)); }
    };
    auto g_load = [&](Stage& st, int kt) {
        const int m0 = m_begin + min(kt, max(nk - 1, 0)) * BK;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const bool ok = m0 + lk + 8 * p < m_end;
            st.a[p] = f4sel(ok, am.at(am.resolve(ka[p], row_of(kt, p)), ia));
            st.b[p] = f4sel(ok, bm.at(bm.resolve(kb[p], row_of(kt, p)), jb));
        }
        k_load(kt + 1);                                            // lookups for the next call (tiles are requested in order)
    };
    auto s_store = [&](const Stage& st, int buf) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            if (BIAS) bacc = f4add(bacc, st.a[p]);                 // every tile is stored exactly once; rows past the split are zero
            stg4(As + buf * BK * BI + (lk + 8 * p) * BI + c4, st.a[p]);
            stg4(Bs + buf * BK * BJ + (lk + 8 * p) * BJ + c4, st.b[p]);
        }
    };
    auto compute = [&](int cur) {
        const float* ab = As + cur * BK * BI + wm * 64 + l31;
        const float* bb = Bs + cur * BK * BJ + wn * 64 + l31;
#pragma unroll
        for (int s = 0; s < BK / 2; ++s) {
            const int kk = 2 * s + h;
            const float a0 = ab[kk * BI], a1 = ab[kk * BI + 32];
            const float b0 = bb[kk * BJ], b1 = bb[kk * BJ + 32];
            acc[0][0] = mfma32(a0, b0, acc[0][0]);
            acc[0][1] = mfma32(a0, b1, acc[0][1]);
            acc[1][0] = mfma32(a1, b0, acc[1][0]);
            acc[1][1] = mfma32(a1, b1, acc[1][1]);
        }
    };

    Stage s0, s1;
    if (nk > 0) {
        k_load(0);
        g_load(s0, 0);
        s_store(s0, 0);
        g_load(s0, 1);
        g_load(s1, 2);
    }
    __syncthreads();
    int kt = 0;
    for (; kt + 2 < nk; kt += 2) {                                 // tiles kt+1 and kt+2 exist: no conditions inside
        compute(0);
        s_store(s0, 1);
        g_load(s0, kt + 3);
        __syncthreads();
        compute(1);
        s_store(s1, 0);
        g_load(s1, kt + 4);
        __syncthreads();
    }
    if (kt < nk) {
        compute(0);
        if (kt + 1 < nk) {
            s_store(s0, 1);
            __syncthreads();
            compute(1);
        }
    }

    float* out = slab + (size_t)z * I * J;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = ti * BI + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const int j = tj * BJ + wn * 64 + ni * 32 + l31;
                if (i < I && j < J) out[(size_t)i * J + j] = acc[mi][ni][r];
            }
    if (BIAS) {
        // column sums of A over the split's rows: the eight row groups of a column quadruple meet in LDS, in fixed order (summed from
        // the operand registers in every tile, so the main loop carries no branch; written by the tj == 0 tiles)
        __syncthreads();
        float4* red = reinterpret_cast<float4*>(smem);
        red[lk * 32 + (t & 31)] = bacc;
        __syncthreads();
        if (tj == 0 && t < 32) {
            float4 sum = red[t];
#pragma unroll
            for (int g = 1; g < 8; ++g) sum = f4add(sum, red[g * 32 + t]);
            const int i0 = ti * BI + 4 * t;
            if (ia == i0) {                                          // (ia is clamped to I - 4 for a quadruple past the last column)
                float* o = bias_slab + (size_t)z * I + i0;
                o[0] = sum.x; o[1] = sum.y; o[2] = sum.z; o[3] = sum.w;
            }
        }
    }
}


// ------------------------------------------------------------------ TN kernel on the bf16 matrix cores
// Same tiling, splits and XCD-aware order as gemm_tn_kernel.  The row (contraction) index is the MFMA k index while the operands
// arrive row-major, so a lane needs eight consecutive ROWS of one column: the tiles are kept in LDS as they arrive ([row][128
// bf16], written with one 8-byte store per float4) and read with gfx950's transposing LDS read (ds_read_b64_tr_b16: a 16-lane
// group reads 4 rows x 16 columns and each lane receives one column).  Rows are 256 B; 16-byte chunk ch of row r sits at chunk
// ch ^ (((r&3)<<2) | ((r>>2)&3)), which makes both the row-wise stores and the transposed reads conflict-free.
// NPROD as in gemm_nt_x3_body (1 plain bf16, 3 split, 6 emulated fp32); fp32 accumulation; the bias column sums add the pieces
// back (exact for NPROD = 6).
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short short4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int tn_img_off(int row, int ch) { return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); }

__device__ __forceinline__ bf16x8 lds_read_tr2(const unsigned char* p0, const unsigned char* p1)
{
    typedef __attribute__((address_space(3))) short4v* lptr;
    const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(p0));
    const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(p1));
    typedef short short8v __attribute__((ext_vector_type(8)));
    const short8v v = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    return __builtin_bit_cast(bf16x8, v);
}

template <int NPROD, class AM, class BM_, bool BIAS>
__global__ __launch_bounds__(256, NPROD == 3 ? 3 : 2)          // NPROD = 1 stages 32 rows (eight float4 per thread and stage)
void gemm_tn_bf16_kernel(AM am, BM_ bm, float* __restrict__ slab, float* __restrict__ bias_slab,
                         int Mrows, int I, int J, int rows_per_split, int tiles_i, int tiles_j, int splits)
{
    constexpr int NIMG = split_images(NPROD);
    constexpr int BI = 128, BJ = 128, BK = NIMG == 1 ? 32 : 16, NPK = BK / 8;
    constexpr int TILE = BK * 256;                                       // bytes of one [BK][128] bf16 tile
    constexpr int IMG = 2 * TILE;                                        // one image: both buffers
    __shared__ __attribute__((aligned(16))) unsigned char tsm[2 * NIMG * IMG];     // 32,768 / 32,768 / 49,152 B
    unsigned char* As = tsm;
    unsigned char* Bs = tsm + NIMG * IMG;
    const int tiles = tiles_i * tiles_j;
    const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
    const int z = (slot / tiles) * 8 + xcd, tile = slot % tiles;
    if (z >= splits) return;
    const int ti = tile % tiles_i, tj = tile / tiles_i;
    const int m_begin = z * rows_per_split;
    const int m_end = min(Mrows, m_begin + rows_per_split);
    const int t = threadIdx.x, lk = t >> 5, c4 = (t & 31) * 4;          // thread: rows lk + 8p, columns c4 .. c4+3
    const int ia = min(ti * BI + c4, I - 4), jb = min(tj * BJ + c4, J - 4);
    const int wave = t >> 6, lane = t & 63, wm = wave >> 1, wn = wave & 1, l31 = lane & 31, h = lane >> 5;
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    float4 bacc = f4zero();                                              // BIAS: this thread's rows of its four columns of A, summed
    const int nk = (max(m_end - m_begin, 0) + BK - 1) / BK;
    struct Stage { float4 a[NPK], b[NPK]; };
    auto g_load = [&](Stage& st, int kt) {
        const int m0 = m_begin + min(kt, max(nk - 1, 0)) * BK;
#pragma unroll
        for (int p = 0; p < NPK; ++p) {
            const int m = m0 + lk + 8 * p;
            const bool ok = m < m_end;
            const int mc = min(m, Mrows - 1);
            st.a[p] = f4sel(ok, am.at(am.row(mc), ia));
            st.b[p] = f4sel(ok, bm.at(bm.row(mc), jb));
        }
    };
    // store offsets of this thread's four columns in rows lk + 8p (the swizzle depends on the row)
    int woff[NPK];
#pragma unroll
    for (int p = 0; p < NPK; ++p) woff[p] = tn_img_off(lk + 8 * p, c4 >> 3) + 8 * ((c4 >> 2) & 1);
    auto s_store = [&](const Stage& st, int buf) {
#pragma unroll
        for (int p = 0; p < NPK; ++p) {
            uint2 pc[NIMG];
            if (BIAS) bacc = f4add(bacc, st.a[p]);                   // every tile is stored exactly once; rows past the split are zero
            split_bf16x4<NIMG>(st.a[p], pc);
#pragma unroll
            for (int i = 0; i < NIMG; ++i) *reinterpret_cast<uint2*>(As + i * IMG + buf * TILE + woff[p]) = pc[i];
            split_bf16x4<NIMG>(st.b[p], pc);
#pragma unroll
            for (int i = 0; i < NIMG; ++i) *reinterpret_cast<uint2*>(Bs + i * IMG + buf * TILE + woff[p]) = pc[i];
        }
    };
    // transposed-read addresses: lane 4q+p of a 16-lane group g supplies row q, columns 4p..4p+3 of the group's 4 x 16 block;
    // group g covers columns 16 (g&1) .. +15 of the 32-column operand and rows 8 (g>>1) + 4u .. +3 of the 16-deep step (u = 0, 1)
    const int gq = (lane & 15) >> 2, gp = lane & 3, gcb = (lane >> 4) & 1, ghh = lane >> 5;
    int roff[BK / 16][2][2][2];                                          // [step][operand A/B][mi/ni][u]
#pragma unroll
    for (int sst = 0; sst < BK / 16; ++sst)
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int row = 16 * sst + 8 * ghh + 4 * u + gq;
                    const int col = (o == 0 ? wm : wn) * 64 + mi * 32 + 16 * gcb + 4 * gp;
                    roff[sst][o][mi][u] = tn_img_off(row, col >> 3) + 8 * ((col >> 2) & 1);
                }
    auto compute = [&](int cur) {
#pragma unroll
        for (int sst = 0; sst < BK / 16; ++sst) {
            bf16x8 af[2][NIMG], bf[2][NIMG];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int q = 0; q < NIMG; ++q) {
                    const unsigned char* pa = As + q * IMG + cur * TILE;
                    const unsigned char* pb = Bs + q * IMG + cur * TILE;
                    af[i][q] = lds_read_tr2(pa + roff[sst][0][i][0], pa + roff[sst][0][i][1]);
                    bf[i][q] = lds_read_tr2(pb + roff[sst][1][i][0], pb + roff[sst][1][i][1]);
                }
            mfma_split_tiles<NPROD, NIMG, 2, 2>(af, bf, acc);
        }
    };
    // as in gemm_nt_x3_body: branch-free half-steps, the split's vector instructions issued between the matrix instructions
    auto half_step = [&](int cur, Stage& st, int kt_next_load) {
        compute(cur);
        s_store(st, cur ^ 1);
        g_load(st, kt_next_load);
        constexpr int NMFMA = 4 * NPROD * (BK / 16);
#pragma unroll
        for (int q = 0; q < NMFMA; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
        }
    };
    Stage s0, s1;
    if (nk > 0) { g_load(s0, 0); s_store(s0, 0); g_load(s0, 1); g_load(s1, 2); }
    __syncthreads();
    int kt = 0;
    for (; kt + 2 < nk; kt += 2) {
        half_step(0, s0, kt + 3);
        __syncthreads();
        half_step(1, s1, kt + 4);
        __syncthreads();
    }
    if (kt < nk) {
        compute(0);
        if (kt + 1 < nk) {
            s_store(s0, 1);
            __syncthreads();
            compute(1);
        }
    }
    float* out = slab + (size_t)z * I * J;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = ti * BI + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const int j = tj * BJ + wn * 64 + ni * 32 + l31;
                if (i < I && j < J) out[(size_t)i * J + j] = acc[mi][ni][r];
            }
    if (BIAS) {
        // column sums of A over the split's rows (exact fp32 sums of the operand values): the eight row groups of a column quadruple
        // meet in LDS, in fixed order.  Summed in every tile so that the main loop carries no branch; written by the tj == 0 tiles.
        __syncthreads();
        float4* red = reinterpret_cast<float4*>(tsm);
        red[lk * 32 + (t & 31)] = bacc;
        __syncthreads();
        if (tj == 0 && t < 32) {
            float4 sum = red[t];
#pragma unroll
            for (int g = 1; g < 8; ++g) sum = f4add(sum, red[g * 32 + t]);
            const int i0 = ti * BI + 4 * t;
            if (ia == i0) {                                              // (ia is clamped to I - 4 for the last partial quadruple)
                float* o = bias_slab + (size_t)z * I + i0;
                o[0] = sum.x; o[1] = sum.y; o[2] = sum.z; o[3] = sum.w;
            }
        }
    }
}

// number of m-splits so that the grid fills the chip (>= ~2 workgroups per CU)
static inline int tn_splits(int Mrows, int I, int J)
{
    const int tiles = cdiv(I, 128) * cdiv(J, 128);
    int s = cdiv(GEMM_SLOTS, tiles);
    const int max_s = cdiv(Mrows, 256);
    if (s > max_s) s = max_s;
    if (s < 1) s = 1;
    if (s > GEMM_SLOTS) s = GEMM_SLOTS;                 // single-tile outputs (dl x dl) still fill the chip
    return s;
}

int tn_extra_lds();                                    // util.hip: bytes of unused dynamic LDS per weight-gradient workgroup (occupancy cap)

template <class AM, class BM_>
static inline int launch_gemm_tn(hipStream_t st, const AM& am, const BM_& bm, float* slab, float* bias_slab,
                                 int Mrows, int I, int J, int splits)
{
    const int rows_per_split = cdiv(cdiv(Mrows, splits), 32) * 32;
    const int tiles_i = cdiv(I, 128), tiles_j = cdiv(J, 128);
    dim3 grid(cdiv(splits, 8) * 8 * tiles_i * tiles_j);
    if (g_gemm_mode != 0) {
#define SMIN_TN_SPLIT(NPROD)                                                                                                                   \
    do {                                                                                                                                       \
        if (bias_slab)                                                                                                                         \
            hipLaunchKernelGGL((gemm_tn_bf16_kernel<NPROD, AM, BM_, true>), grid, dim3(256), 0, st, am, bm, slab, bias_slab, Mrows, I, J,      \
                               rows_per_split, tiles_i, tiles_j, splits);                                                                      \
        else                                                                                                                                   \
            hipLaunchKernelGGL((gemm_tn_bf16_kernel<NPROD, AM, BM_, false>), grid, dim3(256), 0, st, am, bm, slab, bias_slab, Mrows, I, J,     \
                               rows_per_split, tiles_i, tiles_j, splits);                                                                      \
    } while (0)
        if (g_gemm_mode == 1) SMIN_TN_SPLIT(3);
        else if (g_gemm_mode == 2) SMIN_TN_SPLIT(1);
        else SMIN_TN_SPLIT(6);
#undef SMIN_TN_SPLIT
        SMIN_LAUNCH_CHECK();
        return 0;
    }
    // weight gradients run on a low-priority stream beside the step's critical chain: unused dynamic LDS caps how many of
    // their workgroups a CU takes, so that the chain's kernels find registers free (see tn_extra_lds)
    const size_t xl = (size_t)tn_extra_lds();
    if (bias_slab)
        hipLaunchKernelGGL((gemm_tn_kernel<AM, BM_, true>), grid, dim3(256), xl, st, am, bm, slab, bias_slab, Mrows, I, J, rows_per_split,
                           tiles_i, tiles_j, splits);
    else
        hipLaunchKernelGGL((gemm_tn_kernel<AM, BM_, false>), grid, dim3(256), xl, st, am, bm, slab, bias_slab, Mrows, I, J, rows_per_split,
                           tiles_i, tiles_j, splits);
    SMIN_LAUNCH_CHECK();
    return 0;
}

// out[idx] = sum_z slab[z][idx]   (fixed order: deterministic)
__global__ void reduce_slabs_kernel(const float* __restrict__ slab, float* __restrict__ out, int n, int P);
int launch_reduce_slabs(hipStream_t st, const float* slab, float* out, int n, int P);
int launch_reduce_slabs2(hipStream_t st, const float* slabA, float* outA, int nA, const float* slabB, float* outB, int nB, int P, float* outB2 = nullptr);

}  // namespace smin
