// BoundaryUnit, map-sized part (reference models.py:190-194): the gated row reduction of the moment map
//   fbm[b,i,:] = sum_j A_b[b,i,j] * hbar[(b,i,j),:],   hbar = sigmoid(f_m * f_s) * f_m  (gate.hip)
// HBM-bound (reads hbar once).  The L x L boundary self-attention that produces A_b (models.py:164-188) is
// O(B L^2 D) and runs on the host side as library GEMMs; this file owns everything that touches the map.
#include "gemm.h"
#include "smin_hip.h"

namespace smin {

// fbm[b,i,:] = sum over the cells (i, j) of row i of  A_b[b,i,j] * hbar[n,:]
__global__ __launch_bounds__(128)
void boundary_reduce_fwd_kernel(const float* __restrict__ Ab, const float* __restrict__ hbar, const float* base,
                                const int* __restrict__ cells, const int* __restrict__ row_ptr, int L, int D, float* fbm)
{
    const int i = blockIdx.x, b = blockIdx.y;
    const int r0 = row_ptr[b * L + i], r1 = row_ptr[b * L + i + 1];
    const float* arow = Ab + ((size_t)b * L + i) * L;
    for (int d = threadIdx.x * 4; d < D; d += 512) {
        float4 acc = base ? ldg4(base + ((size_t)b * L + i) * D + d) : f4zero();
        for (int n = r0; n < r1; ++n)
            acc = f4fma(ldg4(hbar + (size_t)n * D + d), arow[cells[4 * (size_t)n + 2]], acc);
        stg4(fbm + ((size_t)b * L + i) * D + d, acc);
    }
}

// one workgroup (4 waves) per row (b, i); each wave owns cells r0 + wave, r0 + wave + 4, ...
//   dAb[b,i,j] = <dfbm[b,i,:], hbar[n,:]>          dhbar[n,:] = Ab[b,i,j] * dfbm[b,i,:]
__global__ __launch_bounds__(256)
void boundary_reduce_bwd_kernel(const float* __restrict__ dfbm, const float* __restrict__ Ab, const float* __restrict__ hbar,
                                const int* __restrict__ cells, const int* __restrict__ row_ptr,
                                int L, int D, float* __restrict__ dAb, float* __restrict__ dhbar)
{
    const int i = blockIdx.x, b = blockIdx.y;
    const int r0 = row_ptr[b * L + i], r1 = row_ptr[b * L + i + 1];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float* arow = Ab + ((size_t)b * L + i) * L;
    float* darow = dAb + ((size_t)b * L + i) * L;
    // the row's entries outside the cell list are zero: written here (the row belongs to this workgroup), not by a memset launch
    for (int j = threadIdx.x; j < L; j += 256) darow[j] = 0.f;
    __syncthreads();
    for (int n = r0 + wave; n < r1; n += 4) {
        const int j = cells[4 * (size_t)n + 2];
        const float a = arow[j];
        float dot = 0.f;
        for (int d = lane * 4; d < D; d += 256) {
            const float4 dy = ldg4(dfbm + ((size_t)b * L + i) * D + d);
            const float4 x = ldg4(hbar + (size_t)n * D + d);
            dot = fmaf(dy.x, x.x, dot); dot = fmaf(dy.y, x.y, dot); dot = fmaf(dy.z, x.z, dot); dot = fmaf(dy.w, x.w, dot);
            if (dhbar) stg4(dhbar + (size_t)n * D + d, f4scale(dy, a));      // (NULL: the caller forms it itself, smin_gate_bwd)
        }
        dot = wave_sum(dot);
        if (lane == 0) darow[j] = dot;
    }
}

// ------------------------------------------------------------------------------------------------------------
// The L x L part of BoundaryUnit (reference models.py:164-188) with its word attention (models.py:137-154).
// O(B L^2 D) work, <0.1% of the path; one 256-thread workgroup per boundary row (b, i), operands streamed from L2.
//   Qb = fb Wq^T + bq, Kb = fw Wk^T + bk                         (gemm_nt, rows = B*L and B*Nq)
//   P  = softmax_w(mask(Qb Kb^T / sqrt(D)))   baq = (P fw) * lm   bqv = fb * (baq + fs)          boundary_rows_fwd
//   A  = softmax_j(mask(bqv bqv^T / sqrt(D))) * lm_i              base = (A fb) * lm + fb        boundary_self_fwd
// The map-sized term f_bm is added by boundary_reduce_fwd (base + fbm).

__device__ __forceinline__ float block_dot_row(const float* __restrict__ a, const float* __restrict__ b, int D, int lane)
{
    float s = 0.f;
    for (int d = lane * 4; d < D; d += 256) {
        const float4 x = ldg4(a + d), y = ldg4(b + d);
        s = fmaf(x.x, y.x, s); s = fmaf(x.y, y.y, s); s = fmaf(x.z, y.z, s); s = fmaf(x.w, y.w, s);
    }
    return wave_sum(s);
}

// four dot products of one row with four others at once: the four rows' loads are in flight together (one dot per trip left a
// wave with one L2 round trip per score)
__device__ __forceinline__ float4 block_dot_row4(const float* __restrict__ a, const float* __restrict__ b0, const float* __restrict__ b1,
                                                 const float* __restrict__ b2, const float* __restrict__ b3, int D, int lane)
{
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (int d = lane * 4; d < D; d += 256) {
        const float4 x = ldg4(a + d), y0 = ldg4(b0 + d), y1 = ldg4(b1 + d), y2 = ldg4(b2 + d), y3 = ldg4(b3 + d);
        s0 = fmaf(x.x, y0.x, s0); s0 = fmaf(x.y, y0.y, s0); s0 = fmaf(x.z, y0.z, s0); s0 = fmaf(x.w, y0.w, s0);
        s1 = fmaf(x.x, y1.x, s1); s1 = fmaf(x.y, y1.y, s1); s1 = fmaf(x.z, y1.z, s1); s1 = fmaf(x.w, y1.w, s1);
        s2 = fmaf(x.x, y2.x, s2); s2 = fmaf(x.y, y2.y, s2); s2 = fmaf(x.z, y2.z, s2); s2 = fmaf(x.w, y2.w, s2);
        s3 = fmaf(x.x, y3.x, s3); s3 = fmaf(x.y, y3.y, s3); s3 = fmaf(x.z, y3.z, s3); s3 = fmaf(x.w, y3.w, s3);
    }
    return make_float4(wave_sum(s0), wave_sum(s1), wave_sum(s2), wave_sum(s3));
}

// softmax over n entries of sv[] (LDS) in place by wave 0; entries must already be masked
__device__ __forceinline__ void block_softmax(float* sv, int n, int t)
{
    __syncthreads();
    if (t < 64) {
        float mx = -INFINITY;
        for (int k = t; k < n; k += 64) mx = fmaxf(mx, sv[k]);
        for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        float den = 0.f;
        for (int k = t; k < n; k += 64) { const float e = expf(sv[k] - mx); sv[k] = e; den += e; }
        den = wave_sum(den);
        const float inv = 1.0f / den;
        for (int k = t; k < n; k += 64) sv[k] *= inv;
    }
    __syncthreads();
}

__global__ __launch_bounds__(256)
void boundary_rows_fwd_kernel(const float* __restrict__ Qb, const float* __restrict__ Kb, const float* __restrict__ fw,
                              const float* __restrict__ fb, const float* __restrict__ fs, const float* __restrict__ qmask,
                              const float* __restrict__ lmask, int L, int Nq, int D, float scale,
                              float* __restrict__ P, float* __restrict__ baq, float* __restrict__ bqv)
{
    __shared__ float sv[64];
    const int i = blockIdx.x, b = blockIdx.y, t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const size_t r = (size_t)b * L + i;
    for (int w0 = wave * 4; w0 < Nq; w0 += 16) {
        const float* kb = Kb + (size_t)b * Nq * D;
        const float4 d4 = block_dot_row4(Qb + r * D, kb + (size_t)w0 * D, kb + (size_t)min(w0 + 1, Nq - 1) * D, kb + (size_t)min(w0 + 2, Nq - 1) * D,
                                         kb + (size_t)min(w0 + 3, Nq - 1) * D, D, lane);
        if (lane < 4 && w0 + lane < Nq) {
            const float d = lane == 0 ? d4.x : lane == 1 ? d4.y : lane == 2 ? d4.z : d4.w;
            const float qm = qmask[(size_t)b * Nq + w0 + lane];
            sv[w0 + lane] = qm == 0.f ? -1e9f : d * scale * qm;        // models.py:141-148
        }
    }
    block_softmax(sv, Nq, t);
    if (t < Nq) P[r * Nq + t] = sv[t];
    const float lm = lmask[r];
    for (int d = t * 4; d < D; d += 1024) {
        float4 a = f4zero();
        for (int w = 0; w < Nq; ++w) a = f4fma(ldg4(fw + ((size_t)b * Nq + w) * D + d), sv[w], a);
        a = f4scale(a, lm);                                             // models.py:170
        stg4(baq + r * D + d, a);
        stg4(bqv + r * D + d, f4mul(ldg4(fb + r * D + d), f4add(a, ldg4(fs + (size_t)b * D + d))));
    }
}

__global__ __launch_bounds__(256)
void boundary_self_fwd_kernel(const float* __restrict__ bqv, const float* __restrict__ fb, const float* __restrict__ lmask,
                              int L, int D, float scale, float* __restrict__ A, float* __restrict__ base)
{
    extern __shared__ float sz[];                                       // [L]
    const int i = blockIdx.x, b = blockIdx.y, t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const size_t r = (size_t)b * L + i;
    for (int j0 = wave * 4; j0 < L; j0 += 16) {
        const float* rows = bqv + (size_t)b * L * D;
        const float4 d4 = block_dot_row4(bqv + r * D, rows + (size_t)j0 * D, rows + (size_t)min(j0 + 1, L - 1) * D, rows + (size_t)min(j0 + 2, L - 1) * D,
                                         rows + (size_t)min(j0 + 3, L - 1) * D, D, lane);
        if (lane < 4 && j0 + lane < L) {
            const float d = lane == 0 ? d4.x : lane == 1 ? d4.y : lane == 2 ? d4.z : d4.w;
            const float lj = lmask[(size_t)b * L + j0 + lane];
            sz[j0 + lane] = lj == 0.f ? -1e9f : d * scale * lj;         // models.py:174-181
        }
    }
    block_softmax(sz, L, t);
    const float lm = lmask[r];
    for (int j = t; j < L; j += 256) { sz[j] *= lm; A[r * L + j] = sz[j]; }       // models.py:184
    __syncthreads();
    for (int d = t * 4; d < D; d += 1024) {
        float4 a = f4zero();
        for (int j = 0; j < L; ++j) a = f4fma(ldg4(fb + ((size_t)b * L + j) * D + d), sz[j], a);
        stg4(base + r * D + d, f4add(f4scale(a, lm), ldg4(fb + r * D + d)));     // f_bb * mask + f_b
    }
}

// ---- backward ----------------------------------------------------------------------------------------------------
// row i:  dA[j] = <dout[i]*lm_i, fb[j]> + dA_bm[i][j] ;  dZ = A * (dA - <A, dA>) ;  draw[i][j] = dZ[j] * lm_j * scale
__global__ __launch_bounds__(256)
void boundary_self_bwd_rows_kernel(const float* __restrict__ dout, const float* __restrict__ dAbm, const float* __restrict__ A,
                                   const float* __restrict__ fb, const float* __restrict__ lmask, int L, int D, float scale,
                                   float* __restrict__ draw)
{
    extern __shared__ float sz[];                                       // [L] dA, then dZ
    __shared__ float red;
    const int i = blockIdx.x, b = blockIdx.y, t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const size_t r = (size_t)b * L + i;
    const float lm = lmask[r];
    for (int j0 = wave * 4; j0 < L; j0 += 16) {
        const float* rows = fb + (size_t)b * L * D;
        const float4 d4 = block_dot_row4(dout + r * D, rows + (size_t)j0 * D, rows + (size_t)min(j0 + 1, L - 1) * D, rows + (size_t)min(j0 + 2, L - 1) * D,
                                         rows + (size_t)min(j0 + 3, L - 1) * D, D, lane);
        if (lane < 4 && j0 + lane < L) {
            const float d = lane == 0 ? d4.x : lane == 1 ? d4.y : lane == 2 ? d4.z : d4.w;
            sz[j0 + lane] = d * lm + dAbm[r * L + j0 + lane];
        }
    }
    __syncthreads();
    if (t < 64) {
        float s = 0.f;
        for (int j = t; j < L; j += 64) s = fmaf(A[r * L + j], sz[j], s);
        s = wave_sum(s);
        if (t == 0) red = s;
    }
    __syncthreads();
    for (int j = t; j < L; j += 256)
        draw[r * L + j] = A[r * L + j] * (sz[j] - red) * lmask[(size_t)b * L + j] * scale;
}

// row i (also column i of A / draw):
//   dfb[i]  = dout[i] + sum_i' A[i'][i] lm_i' dout[i'] + dbq (baq[i] + fs)        dbq = sum_j (draw[i][j] + draw[j][i]) bqv[j]
//   dbaq_lm = dbq * fb[i] * lm_i ; dfs partial[i] = dbq * fb[i]
//   dP[w] = <dbaq_lm, fw[w]> ; dS = P (dP - <P, dP>) ; dQK[i][w] = dS[w] qm[w] scale ; dQb[i] = sum_w dQK[w] Kb[w]
__global__ __launch_bounds__(256)
void boundary_self_bwd_cols_kernel(const float* __restrict__ dout, const float* __restrict__ draw, const float* __restrict__ A,
                                   const float* __restrict__ bqv, const float* __restrict__ baq, const float* __restrict__ fb,
                                   const float* __restrict__ fs, const float* __restrict__ fw, const float* __restrict__ Kb,
                                   const float* __restrict__ P, const float* __restrict__ qmask, const float* __restrict__ lmask,
                                   int L, int Nq, int D, float scale,
                                   float* __restrict__ dfb, float* __restrict__ dbaq_lm, float* __restrict__ dfs_part,
                                   float* __restrict__ dQK, float* __restrict__ dQb)
{
    extern __shared__ float sh[];                                       // [2L] column of A*lm, symmetric draw ; then [64] words
    float* sA = sh; float* sR = sh + L; float* sW = sh + 2 * L;
    __shared__ float red;
    const int i = blockIdx.x, b = blockIdx.y, t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const size_t r = (size_t)b * L + i, rb = (size_t)b * L;
    const float lm = lmask[r];
    for (int j = t; j < L; j += 256) {
        sA[j] = A[(rb + j) * L + i] * lmask[rb + j];
        sR[j] = draw[r * L + j] + draw[(rb + j) * L + i];
    }
    __syncthreads();
    for (int d = t * 4; d < D; d += 1024) {
        float4 acc = ldg4(dout + r * D + d), dq = f4zero();
        for (int j = 0; j < L; ++j) {
            acc = f4fma(ldg4(dout + (rb + j) * D + d), sA[j], acc);
            dq = f4fma(ldg4(bqv + (rb + j) * D + d), sR[j], dq);
        }
        const float4 f = ldg4(fb + r * D + d);
        const float4 tq = f4add(ldg4(baq + r * D + d), ldg4(fs + (size_t)b * D + d));
        stg4(dfb + r * D + d, f4add(acc, f4mul(dq, tq)));
        const float4 df = f4mul(dq, f);
        stg4(dfs_part + r * D + d, df);
        stg4(dbaq_lm + r * D + d, f4scale(df, lm));
    }
    __syncthreads();                                                    // dbaq_lm row visible to the whole block (global, same block)
    __threadfence_block();
    for (int w0 = wave * 4; w0 < Nq; w0 += 16) {
        const float* words = fw + (size_t)b * Nq * D;
        const float4 d4 = block_dot_row4(dbaq_lm + r * D, words + (size_t)w0 * D, words + (size_t)min(w0 + 1, Nq - 1) * D, words + (size_t)min(w0 + 2, Nq - 1) * D,
                                         words + (size_t)min(w0 + 3, Nq - 1) * D, D, lane);
        if (lane < 4 && w0 + lane < Nq) sW[w0 + lane] = lane == 0 ? d4.x : lane == 1 ? d4.y : lane == 2 ? d4.z : d4.w;
    }
    __syncthreads();
    if (t < 64) {
        float s = 0.f;
        for (int w = t; w < Nq; w += 64) s = fmaf(P[r * Nq + w], sW[w], s);
        s = wave_sum(s);
        if (t == 0) red = s;
    }
    __syncthreads();
    if (t < Nq) {
        const float v = P[r * Nq + t] * (sW[t] - red) * qmask[(size_t)b * Nq + t] * scale;
        sW[t] = v;
        dQK[r * Nq + t] = v;
    }
    __syncthreads();
    for (int d = t * 4; d < D; d += 1024) {
        float4 a = f4zero();
        for (int w = 0; w < Nq; ++w) a = f4fma(ldg4(Kb + ((size_t)b * Nq + w) * D + d), sW[w], a);
        stg4(dQb + r * D + d, a);
    }
}

// word w:  dKb[w] = sum_i dQK[i][w] Qb[i] ;  dfw[w] = sum_i P[i][w] dbaq_lm[i]
__global__ __launch_bounds__(256)
void boundary_words_bwd_kernel(const float* __restrict__ dQK, const float* __restrict__ P, const float* __restrict__ Qb,
                               const float* __restrict__ dbaq_lm, int L, int Nq, int D, float* __restrict__ dKb, float* __restrict__ dfw)
{
    const int w = blockIdx.x, b = blockIdx.y, t = threadIdx.x;
    const size_t rb = (size_t)b * L;
    for (int d = t * 4; d < D; d += 1024) {
        float4 k = f4zero(), f = f4zero();
        for (int i = 0; i < L; ++i) {
            k = f4fma(ldg4(Qb + (rb + i) * D + d), dQK[(rb + i) * Nq + w], k);
            f = f4fma(ldg4(dbaq_lm + (rb + i) * D + d), P[(rb + i) * Nq + w], f);
        }
        stg4(dKb + ((size_t)b * Nq + w) * D + d, k);
        stg4(dfw + ((size_t)b * Nq + w) * D + d, f);
    }
}

struct EpBias {                     // out = acc + bias
    const float* bias; float* out;
    __device__ __forceinline__ void chunk(const float* Ws, int row0, int col0, int ncols, int M, int N, int lane) const {
        chunk_rows_f4(Ws, row0, col0, ncols, M, N, lane, [&](int row, int col, float4 v) {
            stg4(out + (size_t)row * N + col, f4add(v, ldg4(bias + col)));
        });
    }
};
struct EpAccum {                    // out += acc
    float* out;
    __device__ __forceinline__ void chunk(const float* Ws, int row0, int col0, int ncols, int M, int N, int lane) const {
        chunk_rows_f4(Ws, row0, col0, ncols, M, N, lane, [&](int row, int col, float4 v) {
            float* o = out + (size_t)row * N + col;
            stg4(o, f4add(v, ldg4(o)));
        });
    }
};

// out[b][s][:] = sum over the rows of slice s of partial[b][i][:]; grid (D / 256, slices, B), 256 threads = 64 float4 columns x 4
// row groups, fixed-order combine through LDS.  Long maps (L = 512) are reduced in two passes of this kernel: 32 workgroups
// walking 512 rows each took 4.3 ms per call in the long-video regime.
__global__ __launch_bounds__(256)
void rows_reduce_kernel(const float* __restrict__ partial, int L, int rows_per_slice, int D, float* __restrict__ out)
{
    __shared__ float4 sh[4][64];
    const int b = blockIdx.z, sl = blockIdx.y, c = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int d = (blockIdx.x * 64 + c) * 4;
    const int r0 = sl * rows_per_slice, r1 = min(L, r0 + rows_per_slice);
    float4 s = f4zero();
    if (d < D) {
        const float* p = partial + (size_t)b * L * D + d;
        int i = r0 + g;
        for (; i + 12 < r1; i += 16) {                                  // four rows in flight per thread
            const float4 a0 = ldg4(p + (size_t)i * D), a1 = ldg4(p + (size_t)(i + 4) * D), a2 = ldg4(p + (size_t)(i + 8) * D), a3 = ldg4(p + (size_t)(i + 12) * D);
            s = f4add(f4add(f4add(f4add(s, a0), a1), a2), a3);
        }
        for (; i < r1; i += 4) s = f4add(s, ldg4(p + (size_t)i * D));
    }
    sh[g][c] = s;
    __syncthreads();
    if (g == 0 && d < D) stg4(out + ((size_t)b * gridDim.y + sl) * D + d, f4add(f4add(sh[0][c], sh[1][c]), f4add(sh[2][c], sh[3][c])));
}

}  // namespace smin

using namespace smin;

extern "C" int smin_boundary_reduce_fwd(void* stream, const float* Ab, const float* hbar, const int32_t* cells,
                                        const int32_t* row_ptr, int N, int B, int L, int D, float* fbm)
{
    (void)N;
    SMIN_REQUIRE(D % 4 == 0);
    hipLaunchKernelGGL(boundary_reduce_fwd_kernel, dim3(L, B), dim3(128), 0, (hipStream_t)stream, Ab, hbar, nullptr, cells, row_ptr, L, D, fbm);
    SMIN_LAUNCH_CHECK();
    return 0;
}

extern "C" int smin_boundary_reduce_bwd(void* stream, const float* dfbm, const float* Ab, const float* hbar,
                                        const int32_t* cells, const int32_t* row_ptr, int N, int B, int L, int D,
                                        float* dAb, float* dhbar)
{
    (void)N;
    hipStream_t st = (hipStream_t)stream;
    SMIN_REQUIRE(D % 4 == 0);
    hipLaunchKernelGGL(boundary_reduce_bwd_kernel, dim3(L, B), dim3(256), 0, st, dfbm, Ab, hbar, cells, row_ptr, L, D, dAb, dhbar);
    SMIN_LAUNCH_CHECK();
    return 0;
}


// ---- whole BoundaryUnit -------------------------------------------------------------------------------------------
static size_t bu_fwd_saved_floats(int B, int L, int Nq, int D)
{
    return (size_t)B * L * D * 3 + (size_t)B * Nq * D + (size_t)B * L * Nq + (size_t)B * L * L;
}

extern "C" int smin_boundary_unit_fwd(void* stream, const float* fb, const float* fw, const float* fs, const float* hbar,
                                      const int32_t* cells, const int32_t* row_ptr, int N, int B, int L, int Nq, int D,
                                      const float* Wq, const float* bq, const float* Wk, const float* bk,
                                      const float* qmask, const float* lmask,
                                      float* out, float* Qb, float* Kb, float* P, float* baq, float* bqv, float* A)
{
    (void)N;
    hipStream_t st = (hipStream_t)stream;
    SMIN_REQUIRE(D % 4 == 0 && Nq >= 1 && Nq <= 64 && L >= 1 && L <= 8192);
    const float scale = 1.0f / sqrtf((float)D);
    int rc = launch_gemm_nt(st, PlainMat{fb, D}, PlainMat{Wq, D}, EpBias{bq, Qb}, B * L, D, D);
    if (rc) return rc;
    rc = launch_gemm_nt(st, PlainMat{fw, D}, PlainMat{Wk, D}, EpBias{bk, Kb}, B * Nq, D, D);
    if (rc) return rc;
    hipLaunchKernelGGL(boundary_rows_fwd_kernel, dim3(L, B), dim3(256), 0, st, Qb, Kb, fw, fb, fs, qmask, lmask, L, Nq, D, scale, P, baq, bqv);
    SMIN_LAUNCH_CHECK();
    hipLaunchKernelGGL(boundary_self_fwd_kernel, dim3(L, B), dim3(256), sizeof(float) * L, st, bqv, fb, lmask, L, D, scale, A, out);
    SMIN_LAUNCH_CHECK();
    // out currently holds f_bb * mask + f_b; add the gated row reduction of the map in place
    hipLaunchKernelGGL(boundary_reduce_fwd_kernel, dim3(L, B), dim3(128), 0, st, A, hbar, out, cells, row_ptr, L, D, out);
    SMIN_LAUNCH_CHECK();
    return 0;
}

extern "C" int smin_boundary_unit_bwd(void* stream, const float* dout, const float* fb, const float* fw, const float* fs, const float* hbar,
                                      const int32_t* cells, const int32_t* row_ptr, int N, int B, int L, int Nq, int D,
                                      const float* WqT, const float* WkT, const float* qmask, const float* lmask,
                                      const float* Qb, const float* Kb, const float* P, const float* baq, const float* bqv, const float* A,
                                      float* dfb, float* dfw, float* dfs, float* dhbar, float* dWq, float* dbq, float* dWk, float* dbk,
                                      void* ws, size_t ws_bytes)
{
    (void)N;
    hipStream_t st = (hipStream_t)stream;
    SMIN_REQUIRE(D % 4 == 0 && Nq >= 1 && Nq <= 64);
    const float scale = 1.0f / sqrtf((float)D);
    float* w = reinterpret_cast<float*>(ws);
    size_t off = 0;
    auto take = [&](size_t n) { float* p = w + off; off += (n + 3) & ~(size_t)3; return p; };
    float* dAbm = take((size_t)B * L * L);
    float* draw = take((size_t)B * L * L);
    float* dbaq_lm = take((size_t)B * L * D);
    float* dfs_part = take((size_t)B * L * D);
    float* dQK = take((size_t)B * L * Nq);
    float* dQb = take((size_t)B * L * D);
    float* dKb = take((size_t)B * Nq * D);
    const int sp1 = tn_splits(B * L, D, D), sp2 = tn_splits(B * Nq, D, D);
    float* slab1 = take((size_t)sp1 * D * D); float* bslab1 = take((size_t)sp1 * D);
    float* slab2 = take((size_t)sp2 * D * D); float* bslab2 = take((size_t)sp2 * D);
    SMIN_REQUIRE(off * sizeof(float) <= ws_bytes);

    // f_bm = sum_j A hbar :  dA (map term), dhbar
    hipLaunchKernelGGL(boundary_reduce_bwd_kernel, dim3(L, B), dim3(256), 0, st, dout, A, hbar, cells, row_ptr, L, D, dAbm, dhbar);
    SMIN_LAUNCH_CHECK();
    hipLaunchKernelGGL(boundary_self_bwd_rows_kernel, dim3(L, B), dim3(256), sizeof(float) * L, st, dout, dAbm, A, fb, lmask, L, D, scale, draw);
    SMIN_LAUNCH_CHECK();
    hipLaunchKernelGGL(boundary_self_bwd_cols_kernel, dim3(L, B), dim3(256), sizeof(float) * (2 * L + 64), st, dout, draw, A, bqv, baq, fb, fs, fw, Kb,
                       P, qmask, lmask, L, Nq, D, scale, dfb, dbaq_lm, dfs_part, dQK, dQb);
    SMIN_LAUNCH_CHECK();
    {
        const int slices = L > 96 ? cdiv(L, 32) : 1;                    // D % 4 == 0
        if (slices == 1) {
            hipLaunchKernelGGL(rows_reduce_kernel, dim3(cdiv(D, 256), 1, B), dim3(256), 0, st, dfs_part, L, L, D, dfs);
        } else {
            float* mid = draw;                                          // [B][slices][D] <= [B][L][L] (D <= 2048 < 32 L); draw was consumed by the kernel above
            hipLaunchKernelGGL(rows_reduce_kernel, dim3(cdiv(D, 256), slices, B), dim3(256), 0, st, dfs_part, L, 32, D, mid);
            SMIN_LAUNCH_CHECK();
            hipLaunchKernelGGL(rows_reduce_kernel, dim3(cdiv(D, 256), 1, B), dim3(256), 0, st, mid, slices, slices, D, dfs);
        }
    }
    SMIN_LAUNCH_CHECK();
    hipLaunchKernelGGL(boundary_words_bwd_kernel, dim3(Nq, B), dim3(256), 0, st, dQK, P, Qb, dbaq_lm, L, Nq, D, dKb, dfw);
    SMIN_LAUNCH_CHECK();
    // projections: dfb += dQb Wq ; dWq = dQb^T fb ; dbq = colsum dQb ; dfw += dKb Wk ; dWk = dKb^T fw ; dbk = colsum dKb
    int rc = launch_gemm_nt(st, PlainMat{dQb, D}, PlainMat{WqT, D}, EpAccum{dfb}, B * L, D, D); if (rc) return rc;
    rc = launch_gemm_tn(st, PlainMat{dQb, D}, PlainMat{fb, D}, slab1, bslab1, B * L, D, D, sp1); if (rc) return rc;
    rc = launch_reduce_slabs2(st, slab1, dWq, D * D, bslab1, dbq, D, sp1); if (rc) return rc;
    rc = launch_gemm_nt(st, PlainMat{dKb, D}, PlainMat{WkT, D}, EpAccum{dfw}, B * Nq, D, D); if (rc) return rc;
    rc = launch_gemm_tn(st, PlainMat{dKb, D}, PlainMat{fw, D}, slab2, bslab2, B * Nq, D, D, sp2); if (rc) return rc;
    rc = launch_reduce_slabs2(st, slab2, dWk, D * D, bslab2, dbk, D, sp2); if (rc) return rc;
    return 0;
}
