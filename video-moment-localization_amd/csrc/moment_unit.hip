// MomentUnit (reference models.py:278-303): the two 1x1 Conv2d(D, D) are per-cell linear maps; they are fused into
// one K = 2D contraction whose left operand [f_b[i]*f_b[j] | mean_c f_c] is generated on the fly (never stored):
//   mu[n,:] = m[n] * ( X[n,:] @ Wcat^T + bcat ) + fm[n,:]
// This is ~46% of the path's FLOPs (4*D^2 per cell) and runs on the fp32 MFMA GEMM engine.
#include "gemm.h"
#include "smin_hip.h"

namespace smin {

struct EpMomentOut {                // mu = (acc + bcat) * m + fm
    const float* bcat; const int* cells; const float* fm; float* out;
    struct Add { float4 r; float m; };
    __device__ __forceinline__ void chunk(const float* Ws, int row0, int col0, int ncols, int M, int N, int lane) const {
        chunk_rows_f4_pre<Add>(Ws, row0, col0, ncols, M, N, lane,
            [&](int row, int col) { return Add{ldg4(fm + (size_t)row * N + col), (float)cells[4 * (size_t)row + 3]}; },
            [&](int row, int col, float4 v, const Add& a) {
                stg4(out + (size_t)row * N + col, f4add(f4scale(f4add(v, ldg4(bcat + col)), a.m), a.r));
            });
    }
};

struct EpSplitStore {               // columns [0, D) -> dX1 (pair-product gradient), [D, 2D) -> dfcmean (+ acc: another
    float* dx1; float* dmean; int D; const float* acc;       //  consumer's gradient of fcmean, summed here)
    __device__ __forceinline__ void chunk(const float* Ws, int row0, int col0, int ncols, int M, int N, int lane) const {
        chunk_rows_f4_pre<float4>(Ws, row0, col0, ncols, M, N, lane,
            [&](int row, int col) { return (acc && col >= D) ? ldg4(acc + (size_t)row * D + (col - D)) : f4zero(); },
            [&](int row, int col, float4 v, const float4& a) {
                if (col < D) { stg4(dx1 + (size_t)row * D + col, v); return; }
                stg4(dmean + (size_t)row * D + (col - D), f4add(v, a));
            });
    }
};

// dfb[b][l][:] = sum_{cells (l, j)} dX1[n] * fb[b][j]  +  sum_{cells (i, l)} dX1[n] * fb[b][i]
// (gradient of the outer product f_b[i]*f_b[j]; the diagonal cell (l, l) contributes through both sums).
// One 128-thread workgroup per (b, l); gather only -> deterministic.
__global__ __launch_bounds__(128)
void moment_dfb_kernel(const float* __restrict__ dx1, const float* __restrict__ fb, const int* __restrict__ cells,
                       const int* __restrict__ row_ptr, const int* __restrict__ cellmap, int B, int L, int D, float* __restrict__ dfb,
                       const float* __restrict__ dfb_acc)
{
    // every row of dX1 is read twice, by the workgroup of its start snippet and by that of its end snippet: all workgroups of a
    // sample sit on ONE XCD (ids id and id + 8 share an XCD under round-robin placement: speed only), so that the second read is
    // an L2 hit while the sample's 4 MB of dX1 pass through (dealt over all eight L2s the launch fetched 2.0x the tensor)
    // (sample 8g + k on XCD class (k + g) mod 8: see proposal_map_bwd_events2_kernel)
    const int id = blockIdx.x, slot = id >> 3, sg = slot / L;
    const int l = slot % L, b = sg * 8 + (((id & 7) - sg) & 7);
    if (b >= B) return;
    const float* fbb = fb + (size_t)b * L * D;
    const int r0 = row_ptr[b * L + l], r1 = row_ptr[b * L + l + 1];
    const int* cmap = cellmap + (size_t)b * L * L;
    for (int d = threadIdx.x * 4; d < D; d += 512) {
        float4 acc = f4zero();
        for (int n = r0; n < r1; ++n) {                                   // row l: partner is the end snippet j
            const int j = cells[4 * (size_t)n + 2];
            acc = f4add(acc, f4mul(ldg4(dx1 + (size_t)n * D + d), ldg4(fbb + (size_t)j * D + d)));
        }
        for (int i = 0; i < L; ++i) {                                     // column l: partner is the start snippet i
            const int n = cmap[i * L + l];
            if (n >= 0) acc = f4add(acc, f4mul(ldg4(dx1 + (size_t)n * D + d), ldg4(fbb + (size_t)i * D + d)));
        }
        if (dfb_acc) acc = f4add(acc, ldg4(dfb_acc + ((size_t)b * L + l) * D + d));    // another consumer's gradient of f_b, summed here
        stg4(dfb + ((size_t)b * L + l) * D + d, acc);
    }
}

// x1[n][:] = f_b[b][i][:] * f_b[b][j][:]   (the pair half of the moment unit's left operand, materialised once per layer
// so that the forward contraction and the weight gradient read two plain matrices)
__global__ void pair_product_kernel(const float* __restrict__ fb, const int* __restrict__ cells, size_t N, int L, int D4, float* __restrict__ x1)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * D4) return;
    const size_t n = idx / D4; const int d4 = (int)(idx % D4);
    const Cell cl = load_cell(cells, n);
    const float* base = fb + (size_t)cl.b * L * D4 * 4;
    stg4(x1 + idx * 4, f4mul(ldg4(base + ((size_t)cl.i * D4 + d4) * 4), ldg4(base + ((size_t)cl.j * D4 + d4) * 4)));
}

// the same product rounded to bf16 (round to nearest even: the conversion the bf16 contraction modes apply to their operands)
__global__ void pair_product_bf16_kernel(const float* __restrict__ fb, const int* __restrict__ cells, size_t N, int L, int D4, unsigned short* __restrict__ x1)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * D4) return;
    const size_t n = idx / D4; const int d4 = (int)(idx % D4);
    const Cell cl = load_cell(cells, n);
    const float* base = fb + (size_t)cl.b * L * D4 * 4;
    const float4 v = f4mul(ldg4(base + ((size_t)cl.i * D4 + d4) * 4), ldg4(base + ((size_t)cl.j * D4 + d4) * 4));
    const __bf16 h[4] = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
    unsigned short u[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) u[q] = __builtin_bit_cast(unsigned short, h[q]);
    *reinterpret_cast<uint2*>(x1 + idx * 4) = make_uint2((unsigned)u[0] | ((unsigned)u[1] << 16), (unsigned)u[2] | ((unsigned)u[3] << 16));
}

static CatMat pair_cat(const float* x1, const float* fcmean, int D)
{
    CatMat m;
    m.p[0] = x1; m.p[1] = fcmean; m.p[2] = fcmean; m.p[3] = fcmean;
    m.w = D;
    return m;
}

}  // namespace smin

using namespace smin;

static int moment_unit_fwd(void* stream, const float* fcmean, const float* fm, const float* fb, const int32_t* cells,
                           int N, int B, int L, int D, const float* Wcat, const float* bcat, float* mu, const float* x1, const unsigned short* x1h)
{
    (void)B;
    hipStream_t st = (hipStream_t)stream;
    SMIN_REQUIRE(D % 4 == 0);
    ProfScope prof(st, SMIN_PROF_MOMENT_FWD);
    if (x1h && N > 0)
        return launch_gemm_nt(st, PairCatH{x1h, fcmean, D}, PlainMat{Wcat, 2 * D}, EpMomentOut{bcat, cells, fm, mu}, N, D, 2 * D);
    if (!x1 || N == 0)
        return launch_gemm_nt(st, PairMeanMat{fb, fcmean, cells, L, D}, PlainMat{Wcat, 2 * D}, EpMomentOut{bcat, cells, fm, mu}, N, D, 2 * D);
    return launch_gemm_nt(st, pair_cat(x1, fcmean, D), PlainMat{Wcat, 2 * D}, EpMomentOut{bcat, cells, fm, mu}, N, D, 2 * D);
}

extern "C" int smin_moment_unit_fwd(void* stream, const float* fcmean, const float* fm, const float* fb, const int32_t* cells,
                                    int N, int B, int L, int D, const float* Wcat, const float* bcat, float* mu, const float* x1)
{
    return moment_unit_fwd(stream, fcmean, fm, fb, cells, N, B, L, D, Wcat, bcat, mu, x1, nullptr);
}

extern "C" int smin_moment_unit_fwd_x1h(void* stream, const float* fcmean, const float* fm, const float* fb, const int32_t* cells,
                                        int N, int B, int L, int D, const float* Wcat, const float* bcat, float* mu, const uint16_t* x1h)
{
    SMIN_REQUIRE(x1h != nullptr || N == 0);
    return moment_unit_fwd(stream, fcmean, fm, fb, cells, N, B, L, D, Wcat, bcat, mu, nullptr, x1h);
}

extern "C" int smin_pair_product_bf16(void* stream, const float* fb, const int32_t* cells, int N, int L, int D, uint16_t* x1h)
{
    SMIN_REQUIRE(D % 4 == 0);
    if (N == 0) return 0;
    const size_t tot = (size_t)N * (D / 4);
    hipLaunchKernelGGL(pair_product_bf16_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, fb, cells, (size_t)N, L, D / 4, x1h);
    SMIN_LAUNCH_CHECK();
    return 0;
}

extern "C" int smin_pair_product(void* stream, const float* fb, const int32_t* cells, int N, int L, int D, float* x1)
{
    SMIN_REQUIRE(D % 4 == 0);
    if (N == 0) return 0;
    const size_t tot = (size_t)N * (D / 4);
    hipLaunchKernelGGL(pair_product_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, fb, cells, (size_t)N, L, D / 4, x1);
    SMIN_LAUNCH_CHECK();
    return 0;
}

static int moment_unit_bwd(void* stream, const float* dmu, const float* fcmean, const float* fb, const int32_t* cells,
                           const int32_t* row_ptr, const int32_t* cellmap, int N, int B, int L, int D, const float* WcatT,
                           float* dfcmean, float* dfb, float* dWcat, float* dbcat, void* ws, size_t ws_bytes, int all_valid,
                           const float* dfcmean_acc, const float* x1, const unsigned short* x1h, const float* dfb_acc)
{
    hipStream_t st = (hipStream_t)stream;
    SMIN_REQUIRE(D % 4 == 0 && D <= 2048);
    float* w = reinterpret_cast<float*>(ws);
    const int sp = N > 0 ? tn_splits(N, D, 2 * D) : 1;
    float* dx1 = w;
    float* slab = dx1 + (((size_t)N * D + 3) & ~(size_t)3);
    float* bslab = slab + (size_t)sp * D * 2 * D;
    SMIN_REQUIRE((size_t)((bslab + (size_t)sp * D) - w) * sizeof(float) <= ws_bytes);
    // either half may be skipped (NULL outputs): the two halves share only dmu, so a host can run them on two streams
    const bool want_in = dfb != nullptr, want_w = dWcat != nullptr;      // (dfcmean [N][D] is legitimately NULL when N == 0)
    SMIN_REQUIRE(!want_in || N == 0 || dfcmean != nullptr);
    SMIN_REQUIRE(!want_w || dbcat != nullptr);
    if (N > 0) {
        // dX = (m * dmu) @ Wcat        [N, 2D], contraction over D
        // dWcat[D, 2D] = (m * dmu)^T @ X ; dbcat = colsum(m * dmu)
        // all_valid (mask-driven cell list, m == 1 everywhere): dmu is read as a plain matrix, without the per-row
        // mask lookups the operand loads would otherwise wait for
        int rc = 0;
        if (want_in) {
            ProfScope prof(st, SMIN_PROF_MOMENT_DX);
            if (all_valid) rc = launch_gemm_nt(st, PlainMat{dmu, D}, PlainMat{WcatT, D}, EpSplitStore{dx1, dfcmean, D, dfcmean_acc}, N, 2 * D, D);
            else rc = launch_gemm_nt(st, MaskedRowsMat{dmu, D, cells}, PlainMat{WcatT, D}, EpSplitStore{dx1, dfcmean, D, dfcmean_acc}, N, 2 * D, D);
        }
        if (rc) return rc;
        if (want_w) {
            {
                ProfScope prof(st, SMIN_PROF_MOMENT_DW);
                if (!all_valid) rc = launch_gemm_tn(st, MaskedRowsMat{dmu, D, cells}, PairMeanMat{fb, fcmean, cells, L, D}, slab, bslab, N, D, 2 * D, sp);
                else if (x1h) rc = launch_gemm_tn(st, PlainMat{dmu, D}, PairCatH{x1h, fcmean, D}, slab, bslab, N, D, 2 * D, sp);
                else if (x1) rc = launch_gemm_tn(st, PlainMat{dmu, D}, pair_cat(x1, fcmean, D), slab, bslab, N, D, 2 * D, sp);
                else rc = launch_gemm_tn(st, PlainMat{dmu, D}, PairMeanMat{fb, fcmean, cells, L, D}, slab, bslab, N, D, 2 * D, sp);
            }
            if (rc) return rc;
            rc = launch_reduce_slabs2(st, slab, dWcat, D * 2 * D, bslab, dbcat, D, sp); if (rc) return rc;
        }
    } else if (want_w) {
        (void)hipMemsetAsync(dWcat, 0, sizeof(float) * (size_t)D * 2 * D, st);
        (void)hipMemsetAsync(dbcat, 0, sizeof(float) * (size_t)D, st);
    }
    if (want_in) {
        hipLaunchKernelGGL(moment_dfb_kernel, dim3(L * 8 * cdiv(B, 8)), dim3(128), 0, st, dx1, fb, cells, row_ptr, cellmap, B, L, D, dfb, dfb_acc);
        SMIN_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int smin_moment_unit_bwd(void* stream, const float* dmu, const float* fcmean, const float* fb, const int32_t* cells,
                                    const int32_t* row_ptr, const int32_t* cellmap, int N, int B, int L, int D, const float* WcatT,
                                    float* dfcmean, float* dfb, float* dWcat, float* dbcat, void* ws, size_t ws_bytes, int all_valid,
                                    const float* dfcmean_acc, const float* x1, const float* dfb_acc)
{
    return moment_unit_bwd(stream, dmu, fcmean, fb, cells, row_ptr, cellmap, N, B, L, D, WcatT, dfcmean, dfb, dWcat, dbcat, ws, ws_bytes, all_valid, dfcmean_acc, x1, nullptr,
                           dfb_acc);
}

extern "C" int smin_moment_unit_bwd_x1h(void* stream, const float* dmu, const float* fcmean, const float* fb, const int32_t* cells,
                                        const int32_t* row_ptr, const int32_t* cellmap, int N, int B, int L, int D, const float* WcatT,
                                        float* dfcmean, float* dfb, float* dWcat, float* dbcat, void* ws, size_t ws_bytes, int all_valid,
                                        const float* dfcmean_acc, const uint16_t* x1h, const float* dfb_acc)
{
    SMIN_REQUIRE(all_valid && (x1h != nullptr || N == 0));          // the stored product is a plain matrix: mask-driven cell lists only
    return moment_unit_bwd(stream, dmu, fcmean, fb, cells, row_ptr, cellmap, N, B, L, D, WcatT, dfcmean, dfb, dWcat, dbcat, ws, ws_bytes, all_valid, dfcmean_acc, nullptr, x1h, dfb_acc);
}
