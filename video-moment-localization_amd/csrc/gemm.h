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

struct PlainMat {                       // row-major [rows][ld]
    const float* p; int ld;
    struct Row { const float* p; };
    __device__ __forceinline__ Row row(int r) const { return Row{p + (size_t)r * ld}; }
    __device__ __forceinline__ float4 at(const Row& r, int c) const { return ldg4(r.p + c); }
};

struct MaskedRowsMat {                  // row n of [N][ld] scaled by the cell mask m[n]
    const float* p; int ld; const int* cells;
    struct Row { const float* p; float m; };
    __device__ __forceinline__ Row row(int r) const { return Row{p + (size_t)r * ld, (float)cells[4 * (size_t)r + 3]}; }
    __device__ __forceinline__ float4 at(const Row& r, int c) const { return f4scale(ldg4(r.p + c), r.m); }
};

// X[n] = [ f_b[b,i,:] * f_b[b,j,:]  |  mean_c f_c[n,:] ]   (models.py:292-301), width 2D
struct PairMeanMat {
    const float* fb; const float* fcmean; const int* cells; int L, D;
    struct Row { const float* bi; const float* bj; const float* cm; };
    __device__ __forceinline__ Row row(int r) const {
        const Cell c = load_cell(cells, r);
        const float* base = fb + (size_t)c.b * L * D;
        return Row{base + (size_t)c.i * D, base + (size_t)c.j * D, fcmean + (size_t)r * D};
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
    __device__ __forceinline__ Row row(int r) const {
        const int n = r / C;
        return Row{HAS_DFC ? dfc + (size_t)r * D : nullptr, dmean + (size_t)n * D, MASK ? (float)cells[4 * (size_t)n + 3] : 1.0f};
    }
    __device__ __forceinline__ float4 at(const Row& r, int c) const {
        float4 v = f4scale(ldg4(r.b + c), invC);
        if (HAS_DFC) v = f4add(v, ldg4(r.a + c));
        return MASK ? f4scale(v, r.m) : v;
    }
};

// ------------------------------------------------------------------ helpers
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ float4 f4sel(bool ok, float4 v) {
    return make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
}

constexpr int GEMM_LDC = 132;           // row stride (floats) of the staged 128x128 accumulator tile

// Visit the staged tile row-major, one float4 per lane: 32 lanes cover one 512-byte row segment.
template <class F>
__device__ __forceinline__ void tile_rows_f4(const float* Cs, int row_base, int col_base, int M, int N, int t, F f)
{
#pragma unroll 4
    for (int it = 0; it < 16; ++it) {
        const int idx = t + 256 * it, r = idx >> 5, c4 = (idx & 31) * 4;
        const int row = row_base + r, col = col_base + c4;
        if (row < M && col < N) f(row, col, ldg4(Cs + r * GEMM_LDC + c4));
    }
}

// ------------------------------------------------------------------ NT kernel
// Epilogue protocol: void tile(const float* Cs, int row_base, int col_base, int M, int N, int t) const
//   Cs is the 128 x 128 accumulator tile in LDS (row stride GEMM_LDC); t = threadIdx.x.
template <class AM, class BM_, class EP>
__global__ __launch_bounds__(256, 2)
void gemm_nt_kernel(AM am, BM_ bm, EP ep, int M, int N, int K, int tiles_m, int tiles_n)
{
    constexpr int BM = 128, BN = 128, BK = 32, LDT = BK + 4;
    __shared__ __attribute__((aligned(16))) float smem[2 * (BM + BN) * LDT];      // 73,728 B >= 128 * GEMM_LDC * 4
    float* As = smem;
    float* Bs = smem + 2 * BM * LDT;

    // XCD-aware tile order: blocks b and b+8 share an XCD (round-robin dispatch), so the tiles_n column
    // tiles of one row tile (which re-read the same A rows) are consecutive slots of one XCD's L2.
    const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
    const int tn = slot % tiles_n, tm = (slot / tiles_n) * 8 + xcd;
    if (tm >= tiles_m) return;

    const int t = threadIdx.x, lr = t >> 3, kq = (t & 7) * 4;
    const int row_base = tm * BM, col_base = tn * BN;

    // out-of-range rows are clamped: they only feed accumulator rows/columns that are never stored
    typename AM::Row arow[4];
    typename BM_::Row brow[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        arow[p] = am.row(min(row_base + lr + 32 * p, M - 1));
        brow[p] = bm.row(min(col_base + lr + 32 * p, N - 1));
    }

    const int wave = t >> 6, lane = t & 63, wm = wave >> 1, wn = wave & 1, l31 = lane & 31, h = lane >> 5;
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    float4 ra4[4], rb4[4];
    auto g_load = [&](int k0) {
        const int k = k0 + kq;
        const bool kok = k < K;
        const int kc = min(k, K - 4);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            ra4[p] = f4sel(kok, am.at(arow[p], kc));
            rb4[p] = f4sel(kok, bm.at(brow[p], kc));
        }
    };
    auto s_store = [&](int buf) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            stg4(As + buf * BM * LDT + (lr + 32 * p) * LDT + kq, ra4[p]);
            stg4(Bs + buf * BN * LDT + (lr + 32 * p) * LDT + kq, rb4[p]);
        }
    };

    const int nk = (K + BK - 1) / BK;
    g_load(0);
    s_store(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) g_load((kt + 1) * BK);
        const float* ab = As + cur * BM * LDT + (wm * 64 + l31) * LDT + 4 * h;
        const float* bb = Bs + cur * BN * LDT + (wn * 64 + l31) * LDT + 4 * h;
#pragma unroll
        for (int kg = 0; kg < BK / 8; ++kg) {
            // lane half h supplies k = 8kg+4h+q at MFMA step q: A and B use the same k permutation
            const float4 a0 = ldg4(ab + kg * 8), a1 = ldg4(ab + 32 * LDT + kg * 8);
            const float4 b0 = ldg4(bb + kg * 8), b1 = ldg4(bb + 32 * LDT + kg * 8);
            const float av0[4] = {a0.x, a0.y, a0.z, a0.w}, av1[4] = {a1.x, a1.y, a1.z, a1.w};
            const float bv0[4] = {b0.x, b0.y, b0.z, b0.w}, bv1[4] = {b1.x, b1.y, b1.z, b1.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                acc[0][0] = mfma32(av0[q], bv0[q], acc[0][0]);
                acc[0][1] = mfma32(av0[q], bv1[q], acc[0][1]);
                acc[1][0] = mfma32(av1[q], bv0[q], acc[1][0]);
                acc[1][1] = mfma32(av1[q], bv1[q], acc[1][1]);
            }
        }
        if (kt + 1 < nk) s_store(cur ^ 1);
        __syncthreads();
    }

    // stage the accumulators: C/D layout of a 32x32 MFMA is col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    float* Cs = smem;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                Cs[row * GEMM_LDC + wn * 64 + ni * 32 + l31] = acc[mi][ni][r];
            }
    __syncthreads();
    ep.tile(Cs, row_base, col_base, M, N, t);
}

template <class AM, class BM_, class EP>
static inline int launch_gemm_nt(hipStream_t st, const AM& am, const BM_& bm, const EP& ep, int M, int N, int K)
{
    if (M <= 0 || N <= 0) return 0;
    const int tiles_m = cdiv(M, 128), tiles_n = cdiv(N, 128);
    const int blocks = cdiv(tiles_m, 8) * 8 * tiles_n;
    hipLaunchKernelGGL((gemm_nt_kernel<AM, BM_, EP>), dim3(blocks), dim3(256), 0, st, am, bm, ep, M, N, K, tiles_m, tiles_n);
    SMIN_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------ TN kernel (weight gradients)
// slab[z][I][J] = sum over rows m in split z of A[m][i] * B[m][j];  optional bias slab[z][I] = sum_m A[m][i].
template <class AM, class BM_, bool BIAS>
__global__ __launch_bounds__(256, 2)
void gemm_tn_kernel(AM am, BM_ bm, float* __restrict__ slab, float* __restrict__ bias_slab,
                    int Mrows, int I, int J, int rows_per_split)
{
    constexpr int BI = 128, BJ = 128, BK = 32;
    __shared__ __attribute__((aligned(16))) float smem[2 * BK * (BI + BJ)];
    float* As = smem;
    float* Bs = smem + 2 * BK * BI;

    const int ti = blockIdx.x, tj = blockIdx.y, z = blockIdx.z;
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
    float bsum = 0.f;

    float4 ra4[4], rb4[4];
    auto g_load = [&](int m0) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int m = m0 + lk + 8 * p;
            const bool ok = m < m_end;
            const int mc = min(m, Mrows - 1);
            ra4[p] = f4sel(ok, am.at(am.row(mc), ia));
            rb4[p] = f4sel(ok, bm.at(bm.row(mc), jb));
        }
    };
    auto s_store = [&](int buf) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            stg4(As + buf * BK * BI + (lk + 8 * p) * BI + c4, ra4[p]);
            stg4(Bs + buf * BK * BJ + (lk + 8 * p) * BJ + c4, rb4[p]);
        }
    };

    const int nk = (max(m_end - m_begin, 0) + BK - 1) / BK;
    if (nk > 0) {
        g_load(m_begin);
        s_store(0);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) g_load(m_begin + (kt + 1) * BK);
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
        if (BIAS && tj == 0 && t < BI) {
            const float* col = As + cur * BK * BI + t;
#pragma unroll
            for (int kk = 0; kk < BK; ++kk) bsum += col[kk * BI];
        }
        if (kt + 1 < nk) s_store(cur ^ 1);
        __syncthreads();
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
    if (BIAS && tj == 0 && t < BI && ti * BI + t < I) bias_slab[(size_t)z * I + ti * BI + t] = bsum;
}

// number of m-splits so that the grid fills the chip (>= ~2 workgroups per CU)
static inline int tn_splits(int Mrows, int I, int J)
{
    const int tiles = cdiv(I, 128) * cdiv(J, 128);
    int s = cdiv(512, tiles);
    const int max_s = cdiv(Mrows, 256);
    if (s > max_s) s = max_s;
    if (s < 1) s = 1;
    if (s > 64) s = 64;
    return s;
}

template <class AM, class BM_>
static inline int launch_gemm_tn(hipStream_t st, const AM& am, const BM_& bm, float* slab, float* bias_slab,
                                 int Mrows, int I, int J, int splits)
{
    const int rows_per_split = cdiv(cdiv(Mrows, splits), 32) * 32;
    dim3 grid(cdiv(I, 128), cdiv(J, 128), splits);
    if (bias_slab)
        hipLaunchKernelGGL((gemm_tn_kernel<AM, BM_, true>), grid, dim3(256), 0, st, am, bm, slab, bias_slab, Mrows, I, J, rows_per_split);
    else
        hipLaunchKernelGGL((gemm_tn_kernel<AM, BM_, false>), grid, dim3(256), 0, st, am, bm, slab, bias_slab, Mrows, I, J, rows_per_split);
    SMIN_LAUNCH_CHECK();
    return 0;
}

// out[idx] = sum_z slab[z][idx]   (fixed order: deterministic)
__global__ void reduce_slabs_kernel(const float* __restrict__ slab, float* __restrict__ out, int n, int P);
int launch_reduce_slabs(hipStream_t st, const float* slab, float* out, int n, int P);

}  // namespace smin
