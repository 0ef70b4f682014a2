// Parameter-only operands of the content stream and of the moment unit, for every SMI layer in ONE launch per direction.
//
// The content stream composes the content unit's two linear maps in the dl-dimensional space, which needs, per layer k:
//   Pcat_k   = [Wch_k Wc_0 | Wch_k Wc_1 | .. | Wch_k Wc_{k-1}]        (dl x k*dl; linear_c_hat of layer k after linear_c of layer l,
//                                                                    reference models.py:247, 269), in parts of <= 4 blocks
//   consts_k = bch_k + Wch_k (bc_0 + .. + bc_{k-1})                  (dl)
//   Wch_all  = [Wch_0; Wch_1; ..]                                    (nl*dl x D)
// and the moment unit (models.py:288-303) its fused weight  Wcat_k = [Wfb_k | Wfc_k] (D x 2D),  bcat_k = bfb_k + bfc_k.
// As torch calls these were ~25 launches forward and ~45 backward per step (hipBLASLt products, rocBLAS matvecs, concatenations).
// Plain fp32 FMAs (the products are 17 MFLOP each): exact arithmetic in a fixed order, no matrix cores, no packed arithmetic.
#include "common.h"
#include "smin_hip.h"

namespace smin {

constexpr int PP_MAX_LAYERS = 8;
constexpr int PP_MAX_PARTS = 2;                     // ceil((PP_MAX_LAYERS - 1) / 4)

struct PrepParams {                                 // per layer: Wch (dl,D), bch (dl), Wc (D,dl), bc (D), Wfb (D,D), bfb (D), Wfc (D,D), bfc (D)
    const float* p[PP_MAX_LAYERS][8];
    float* Pcat[PP_MAX_LAYERS][PP_MAX_PARTS];       // [dl][nseg*dl], nseg = min(4, k - 4*part)
    float* consts;                                  // [nl][dl]
    float* Wcat;                                    // [nl][D][2D]
    float* bcat;                                    // [nl][D]
    float* Wch_all;                                 // [nl*dl][D]
    int nl, D, dl;
    int prod_tiles0, const_blocks0, copy_blocks0;   // first block of each job class in the 1-D grid
    int ntiles_per_prod;                            // (dl/32)^2
};

// one 32x32 tile of one product block  C[i][j] = sum_d Wch_k[i][d] Wc_l[d][j];  256 threads, thread (ty, tx) -> rows ty, ty+16; cols tx, tx+16
__device__ __forceinline__ void prep_product_tile(const PrepParams& P, int prod, int tile, float* sA, float* sB)
{
    // enumerate (k, l) with l < k in the order k = 1: (1,0); k = 2: (2,0), (2,1); ...
    int k = 1, first = 0;
    while (prod >= first + k) { first += k; ++k; }
    const int l = prod - first;
    const int D = P.D, dl = P.dl, tcols = dl / 32;
    const int i0 = (tile / tcols) * 32, j0 = (tile % tcols) * 32;
    const float* Wch = P.p[k][0];
    const float* Wc = P.p[l][2];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    float c00 = 0.f, c01 = 0.f, c10 = 0.f, c11 = 0.f;
    for (int d0 = 0; d0 < D; d0 += 32) {
        // A tile: Wch[i0 + r][d0 + c] (rows contiguous in d);  B tile: Wc[d0 + r][j0 + c] (rows contiguous in j)
        for (int e = threadIdx.x; e < 32 * 32; e += 256) {
            const int r = e >> 5, c = e & 31;
            sA[r * 33 + c] = (d0 + c < D) ? Wch[(size_t)(i0 + r) * D + d0 + c] : 0.f;
            sB[r * 33 + c] = (d0 + r < D) ? Wc[(size_t)(d0 + r) * dl + j0 + c] : 0.f;
        }
        __syncthreads();
#pragma unroll 8
        for (int d = 0; d < 32; ++d) {
            const float a0 = sA[ty * 33 + d], a1 = sA[(ty + 16) * 33 + d], b0 = sB[d * 33 + tx], b1 = sB[d * 33 + tx + 16];
            c00 = fmaf(a0, b0, c00); c01 = fmaf(a0, b1, c01); c10 = fmaf(a1, b0, c10); c11 = fmaf(a1, b1, c11);
        }
        __syncthreads();
    }
    const int part = l / 4, sgm = l % 4, nseg = min(4, k - 4 * part), ld = nseg * dl;
    float* C = P.Pcat[k][part] + sgm * dl;
    C[(size_t)(i0 + ty) * ld + j0 + tx] = c00; C[(size_t)(i0 + ty) * ld + j0 + tx + 16] = c01;
    C[(size_t)(i0 + ty + 16) * ld + j0 + tx] = c10; C[(size_t)(i0 + ty + 16) * ld + j0 + tx + 16] = c11;
}

__global__ __launch_bounds__(256)
void param_prep_fwd_kernel(PrepParams P)
{
    __shared__ float sA[32 * 33], sB[32 * 33];
    const int blk = blockIdx.x, D = P.D, dl = P.dl, nl = P.nl;
    if (blk >= P.prod_tiles0 && blk < P.const_blocks0) {
        const int q = blk - P.prod_tiles0;
        prep_product_tile(P, q / P.ntiles_per_prod, q % P.ntiles_per_prod, sA, sB);
        return;
    }
    if (blk >= P.const_blocks0 && blk < P.copy_blocks0) {
        // consts_k = bch_k + Wch_k bsum_k, bsum_k = bc_0 + .. + bc_{k-1} (summed in layer order); one block per layer
        const int k = blk - P.const_blocks0;
        float* bs = sA;                                          // [D] (D <= 1056 fits the two tiles: 2 * 1056 floats)
        for (int d = threadIdx.x; d < D; d += 256) {
            float s = 0.f;
            for (int l = 0; l < k; ++l) s += P.p[l][3][d];
            bs[d] = s;
        }
        __syncthreads();
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        for (int i = wave; i < dl; i += 4) {
            float s = 0.f;
            if (k > 0) {
                const float* w = P.p[k][0] + (size_t)i * D;
                for (int d = lane; d < D; d += 64) s = fmaf(w[d], bs[d], s);
                s = wave_sum(s);
            }
            if (lane == 0) P.consts[(size_t)k * dl + i] = P.p[k][1][i] + s;
        }
        return;
    }
    // copies: grid-stride over the element-wise jobs of all layers
    const size_t nblk = gridDim.x - P.copy_blocks0, b = blk - P.copy_blocks0;
    const size_t per_layer = (size_t)D * 2 * D / 4 + (size_t)dl * D / 4 + (size_t)D / 4;     // float4 units: Wcat, Wch_all rows, bcat
    const size_t tot = per_layer * nl;
    for (size_t e = b * 256 + threadIdx.x; e < tot; e += nblk * 256) {
        const int k = (int)(e / per_layer);
        size_t r = e % per_layer;
        if (r < (size_t)D * 2 * D / 4) {                         // Wcat_k[row][c4] = c4 < D ? Wfb_k[row][c4] : Wfc_k[row][c4 - D]
            const size_t row = r / (2 * D / 4); const int c4 = (int)(r % (2 * D / 4)) * 4;
            const float4 v = c4 < D ? ldg4(P.p[k][4] + row * D + c4) : ldg4(P.p[k][6] + row * D + (c4 - D));
            stg4(P.Wcat + ((size_t)k * D + row) * 2 * D + c4, v);
            continue;
        }
        r -= (size_t)D * 2 * D / 4;
        if (r < (size_t)dl * D / 4) {                            // Wch_all[k*dl + i][:] = Wch_k[i][:]
            stg4(P.Wch_all + (size_t)k * dl * D + r * 4, ldg4(P.p[k][0] + r * 4));
            continue;
        }
        r -= (size_t)dl * D / 4;
        stg4(P.bcat + (size_t)k * D + r * 4, f4add(ldg4(P.p[k][5] + r * 4), ldg4(P.p[k][7] + r * 4)));
    }
}


// ------------------------------------------------------------------ backward
struct PrepGrads {
    const float* p[PP_MAX_LAYERS][8];               // the parameters, as forward
    const float* dPcat[PP_MAX_LAYERS][PP_MAX_PARTS];
    const float* dWch_base[PP_MAX_LAYERS];          // gradient of Wch_k from its direct use (gate term of chat_k), or NULL
    const float* dWc_base[PP_MAX_LAYERS];           // gradient of Wc_l / bc_l from the clip-mean update
    const float* dbc_base[PP_MAX_LAYERS];
    float* g[PP_MAX_LAYERS][8];                     // gradients of the eight parameters of a layer, every entry written
    const float* dconsts; const float* dWcat; const float* dbcat; const float* dWch_all;
    int nl, D, dl;
    int wc_blocks0, bc_blocks0, copy_blocks0;
};

__global__ __launch_bounds__(256)
void param_prep_bwd_kernel(PrepGrads P)
{
    __shared__ float sA[32 * 33], sB[32 * 33];
    const int blk = blockIdx.x, D = P.D, dl = P.dl, nl = P.nl;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int tiles = (dl / 32) * (D / 32);                     // per layer, for dWch (dl x D) and for dWc (D x dl)
    if (blk < P.wc_blocks0) {
        // dWch_k[i][d] = base + dWch_all[k dl + i][d] + dconst_k[i] bsum_k[d] + sum_{l<k} sum_j dP_kl[i][j] Wc_l[d][j]
        const int k = blk / tiles, tile = blk % tiles, tcols = D / 32;
        const int i0 = (tile / tcols) * 32, d0 = (tile % tcols) * 32;
        float c[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
        for (int l = 0; l < k; ++l) {
            const int part = l / 4, sgm = l % 4, nseg = min(4, k - 4 * part), ld = nseg * dl;
            const float* dP = P.dPcat[k][part] + sgm * dl;
            const float* Wc = P.p[l][2];
            for (int j0 = 0; j0 < dl; j0 += 32) {
                for (int e = threadIdx.x; e < 32 * 32; e += 256) {
                    const int r = e >> 5, cc = e & 31;
                    sA[r * 33 + cc] = dP[(size_t)(i0 + r) * ld + j0 + cc];
                    sB[r * 33 + cc] = Wc[(size_t)(d0 + r) * dl + j0 + cc];
                }
                __syncthreads();
#pragma unroll 8
                for (int j = 0; j < 32; ++j) {
                    const float a0 = sA[ty * 33 + j], a1 = sA[(ty + 16) * 33 + j], b0 = sB[tx * 33 + j], b1 = sB[(tx + 16) * 33 + j];
                    c[0][0] = fmaf(a0, b0, c[0][0]); c[0][1] = fmaf(a0, b1, c[0][1]); c[1][0] = fmaf(a1, b0, c[1][0]); c[1][1] = fmaf(a1, b1, c[1][1]);
                }
                __syncthreads();
            }
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int i = i0 + ty + 16 * a, d = d0 + tx + 16 * b;
                float bs = 0.f;
                for (int l = 0; l < k; ++l) bs += P.p[l][3][d];
                float v = P.dWch_all[((size_t)k * dl + i) * D + d] + c[a][b];
                if (k > 0) v = fmaf(P.dconsts[(size_t)k * dl + i], bs, v);
                if (P.dWch_base[k]) v += P.dWch_base[k][(size_t)i * D + d];
                P.g[k][0][(size_t)i * D + d] = v;
            }
        return;
    }
    if (blk < P.bc_blocks0) {
        // dWc_l[d][j] = base + sum_{k>l} sum_i Wch_k[i][d] dP_kl[i][j]
        const int q = blk - P.wc_blocks0, l = q / tiles, tile = q % tiles, tcols = dl / 32;
        const int d0 = (tile / tcols) * 32, j0 = (tile % tcols) * 32;
        float c[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
        for (int k = l + 1; k < nl; ++k) {
            const int part = l / 4, sgm = l % 4, nseg = min(4, k - 4 * part), ld = nseg * dl;
            const float* dP = P.dPcat[k][part] + sgm * dl;
            const float* Wch = P.p[k][0];
            for (int i0 = 0; i0 < dl; i0 += 32) {
                for (int e = threadIdx.x; e < 32 * 32; e += 256) {
                    const int r = e >> 5, cc = e & 31;
                    sA[r * 33 + cc] = Wch[(size_t)(i0 + r) * D + d0 + cc];
                    sB[r * 33 + cc] = dP[(size_t)(i0 + r) * ld + j0 + cc];
                }
                __syncthreads();
#pragma unroll 8
                for (int i = 0; i < 32; ++i) {
                    const float a0 = sA[i * 33 + ty], a1 = sA[i * 33 + ty + 16], b0 = sB[i * 33 + tx], b1 = sB[i * 33 + tx + 16];
                    c[0][0] = fmaf(a0, b0, c[0][0]); c[0][1] = fmaf(a0, b1, c[0][1]); c[1][0] = fmaf(a1, b0, c[1][0]); c[1][1] = fmaf(a1, b1, c[1][1]);
                }
                __syncthreads();
            }
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int d = d0 + ty + 16 * a, j = j0 + tx + 16 * b;
                P.g[l][2][(size_t)d * dl + j] = P.dWc_base[l][(size_t)d * dl + j] + c[a][b];
            }
        return;
    }
    if (blk < P.copy_blocks0) {
        // dbc_l[d] = base + sum_{k>l} sum_i Wch_k[i][d] dconst_k[i] ;  dbch_l = dconst_l
        const int l = blk - P.bc_blocks0;
        for (int d = threadIdx.x; d < D; d += 256) {
            float s = P.dbc_base[l][d];
            for (int k = l + 1; k < nl; ++k) {
                const float* Wch = P.p[k][0];
                const float* dc = P.dconsts + (size_t)k * dl;
                float t = 0.f;
                for (int i = 0; i < dl; ++i) t = fmaf(Wch[(size_t)i * D + d], dc[i], t);
                s += t;
            }
            P.g[l][3][d] = s;
        }
        for (int i = threadIdx.x; i < dl; i += 256) P.g[l][1][i] = P.dconsts[(size_t)l * dl + i];
        return;
    }
    // moment-unit weights: dWfb_k = dWcat_k[:, :D], dWfc_k = dWcat_k[:, D:], dbfb_k = dbfc_k = dbcat_k
    const size_t nblk = gridDim.x - P.copy_blocks0, b = blk - P.copy_blocks0;
    const size_t per_layer = (size_t)D * 2 * D / 4 + (size_t)D / 4;
    for (size_t e = b * 256 + threadIdx.x; e < per_layer * nl; e += nblk * 256) {
        const int k = (int)(e / per_layer);
        size_t r = e % per_layer;
        if (r < (size_t)D * 2 * D / 4) {
            const size_t row = r / (2 * D / 4); const int c4 = (int)(r % (2 * D / 4)) * 4;
            const float4 v = ldg4(P.dWcat + ((size_t)k * D + row) * 2 * D + c4);
            if (c4 < D) stg4(P.g[k][4] + row * D + c4, v); else stg4(P.g[k][6] + row * D + (c4 - D), v);
            continue;
        }
        r -= (size_t)D * 2 * D / 4;
        const float4 v = ldg4(P.dbcat + (size_t)k * D + r * 4);
        stg4(P.g[k][5] + r * 4, v); stg4(P.g[k][7] + r * 4, v);
    }
}

}  // namespace smin

using namespace smin;

extern "C" int smin_param_prep_fwd(void* stream, const float* const* params, int nl, int D, int dl, float* const* Pcat, float* consts, float* Wcat,
                                   float* bcat, float* Wch_all)
{
    SMIN_REQUIRE(nl >= 1 && nl <= PP_MAX_LAYERS && D % 4 == 0 && D <= 1056 && dl % 32 == 0 && dl >= 32);
    PrepParams P;
    for (int k = 0; k < nl; ++k) {
        for (int q = 0; q < 8; ++q) P.p[k][q] = params[k * 8 + q];
        for (int part = 0; part < PP_MAX_PARTS; ++part) P.Pcat[k][part] = (k > 4 * part) ? Pcat[k * PP_MAX_PARTS + part] : nullptr;
    }
    P.consts = consts; P.Wcat = Wcat; P.bcat = bcat; P.Wch_all = Wch_all;
    P.nl = nl; P.D = D; P.dl = dl;
    P.ntiles_per_prod = (dl / 32) * (dl / 32);
    const int nprod = nl * (nl - 1) / 2;
    P.prod_tiles0 = 0;
    P.const_blocks0 = nprod * P.ntiles_per_prod;
    P.copy_blocks0 = P.const_blocks0 + nl;
    const int copy_blocks = 256;
    hipLaunchKernelGGL(param_prep_fwd_kernel, dim3(P.copy_blocks0 + copy_blocks), dim3(256), 0, (hipStream_t)stream, P);
    SMIN_LAUNCH_CHECK();
    return 0;
}

extern "C" int smin_param_prep_bwd(void* stream, const float* const* params, int nl, int D, int dl, const float* const* dPcat, const float* dconsts,
                                   const float* dWcat, const float* dbcat, const float* dWch_all, const float* const* dWch_base,
                                   const float* const* dWc_base, const float* const* dbc_base, float* const* grads)
{
    SMIN_REQUIRE(nl >= 1 && nl <= PP_MAX_LAYERS && D % 32 == 0 && dl % 32 == 0 && dl >= 32);
    PrepGrads P;
    for (int k = 0; k < nl; ++k) {
        for (int q = 0; q < 8; ++q) { P.p[k][q] = params[k * 8 + q]; P.g[k][q] = grads[k * 8 + q]; SMIN_REQUIRE(grads[k * 8 + q] != nullptr); }
        for (int part = 0; part < PP_MAX_PARTS; ++part) {
            P.dPcat[k][part] = (k > 4 * part) ? dPcat[k * PP_MAX_PARTS + part] : nullptr;
            SMIN_REQUIRE(k <= 4 * part || P.dPcat[k][part] != nullptr);
        }
        P.dWch_base[k] = dWch_base[k]; P.dWc_base[k] = dWc_base[k]; P.dbc_base[k] = dbc_base[k];
        SMIN_REQUIRE(dWc_base[k] != nullptr && dbc_base[k] != nullptr);
    }
    P.dconsts = dconsts; P.dWcat = dWcat; P.dbcat = dbcat; P.dWch_all = dWch_all;
    P.nl = nl; P.D = D; P.dl = dl;
    const int tiles = (dl / 32) * (D / 32);
    P.wc_blocks0 = nl * tiles;
    P.bc_blocks0 = P.wc_blocks0 + nl * tiles;
    P.copy_blocks0 = P.bc_blocks0 + nl;
    hipLaunchKernelGGL(param_prep_bwd_kernel, dim3(P.copy_blocks0 + 256), dim3(256), 0, (hipStream_t)stream, P);
    SMIN_LAUNCH_CHECK();
    return 0;
}
