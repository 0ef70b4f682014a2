// VideoEncoder + the Hadamard fusion of the backbone (reference models.py:25-36 and 81-83):
//   f_v[b][t][:] = (x[b][t][:] W^T + bias + pe[t][:]) * vmask[b][t]         f[b][t][:] = f_v[b][t][:] * f_s[b][:]
// as one contraction with a fused epilogue; backward: one element-wise pass forms the masked gradient
// of the projection and reduces df_s, a second one reduces the position-embedding gradient, then the weight gradient
// is a TN contraction.  Reductions are per-thread loops in fixed order (deterministic).
#include "gemm.h"
#include "smin_hip.h"

namespace smin {

struct EpVideoEnc {                 // row = b*T + t
    const float* bias; const float* pe; const float* vmask; const float* fs; float* fv; float* f; int T;
    __device__ __forceinline__ void chunk(const float* Ws, int row0, int col0, int ncols, int M, int N, int lane) const {
        chunk_rows_f4(Ws, row0, col0, ncols, M, N, lane, [&](int row, int col, float4 v) {
            const int b = row / T, t = row - b * T;
            const float vm = vmask[row];
            const float4 e = f4scale(f4add(f4add(v, ldg4(bias + col)), ldg4(pe + (size_t)t * N + col)), vm);
            stg4(fv + (size_t)row * N + col, e);
            if (f) stg4(f + (size_t)row * N + col, f4mul(e, ldg4(fs + (size_t)b * N + col)));     // (NULL: smin_video_encoder_gate forms f later)
        });
    }
};

// dv[b][t][:] = df * fs[b] * vmask[b][t] ;  dfs[b][:] = sum_t df[b][t][:] * fv[b][t][:]
// grid (D/4 / 64, B), 64 x VE_PH threads: lane = feature group, threadIdx.y = time phase (t = ph, ph + VE_PH, ..); the phases'
// partial sums meet in LDS in phase order (fixed order: deterministic).  One wave per (b, 64 feature groups) left the chip empty
// for 0.12 ms at ActivityNet size.
constexpr int VE_PH = 8;
__global__ __launch_bounds__(64 * VE_PH)
void video_enc_bwd_rows_kernel(const float* __restrict__ df, const float* __restrict__ fv, const float* __restrict__ fs,
                               const float* __restrict__ vmask, int T, int D4, float* __restrict__ dv, float* __restrict__ dfs)
{
    __shared__ float4 part[VE_PH][64];
    const int d4 = blockIdx.x * 64 + threadIdx.x, b = blockIdx.y, ph = threadIdx.y;
    float4 acc = f4zero();
    if (d4 < D4) {
        const float4 s4 = ldg4(fs + ((size_t)b * D4 + d4) * 4);
        for (int t = ph; t < T; t += VE_PH) {
            const size_t o = (((size_t)b * T + t) * D4 + d4) * 4;
            const float4 g = ldg4(df + o);
            acc = f4add(acc, f4mul(g, ldg4(fv + o)));
            stg4(dv + o, f4scale(f4mul(g, s4), vmask[(size_t)b * T + t]));
        }
    }
    part[ph][threadIdx.x] = acc;
    __syncthreads();
    if (ph == 0 && d4 < D4) {
        for (int q = 1; q < VE_PH; ++q) acc = f4add(acc, part[q][threadIdx.x]);
        stg4(dfs + ((size_t)b * D4 + d4) * 4, acc);
    }
}

// dpe[t][:] = sum_b dv[b][t][:]                                                              grid (D/4 / 64, T)
__global__ void video_enc_bwd_pe_kernel(const float* __restrict__ dv, int B, int T, int D4, float* __restrict__ dpe)
{
    const int d4 = blockIdx.x * blockDim.x + threadIdx.x, t = blockIdx.y;
    if (d4 >= D4) return;
    float4 acc = f4zero();
    for (int b = 0; b < B; ++b) acc = f4add(acc, ldg4(dv + (((size_t)b * T + t) * D4 + d4) * 4));
    stg4(dpe + ((size_t)t * D4 + d4) * 4, acc);
}

}  // namespace smin

using namespace smin;

extern "C" int smin_video_encoder_fwd(void* stream, const float* x, const float* W, const float* bias, const float* pe, const float* vmask,
                                      const float* fs, int B, int T, int Din, int D, float* fv, float* f)
{
    SMIN_REQUIRE(Din % 4 == 0 && D % 4 == 0 && B >= 1 && T >= 1);
    SMIN_REQUIRE((f == nullptr) == (fs == nullptr));
    return launch_gemm_nt((hipStream_t)stream, PlainMat{x, Din}, PlainMat{W, Din}, EpVideoEnc{bias, pe, vmask, fs, fv, f, T}, B * T, D, Din);
}

// f[b][t][:] = fv[b][t][:] * fs[b][:] -- the Hadamard product alone, for a host that runs the projection (fs == f == NULL above)
// beside the query encoder, which produces fs
namespace smin {
__global__ void video_enc_gate_kernel(const float* __restrict__ fv, const float* __restrict__ fs, int T, int D4, size_t total, float* __restrict__ f)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int d4 = (int)(idx % D4);
    const size_t b = idx / D4 / T;
    stg4(f + idx * 4, f4mul(ldg4(fv + idx * 4), ldg4(fs + (b * D4 + d4) * 4)));
}
}  // namespace smin

extern "C" int smin_video_encoder_gate(void* stream, const float* fv, const float* fs, int B, int T, int D, float* f)
{
    SMIN_REQUIRE(D % 4 == 0 && B >= 1 && T >= 1);
    const size_t total = (size_t)B * T * (D / 4);
    hipLaunchKernelGGL(video_enc_gate_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, fv, fs, T, D / 4, total, f);
    SMIN_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t smin_video_encoder_bwd_workspace_bytes(int B, int T, int Din, int D)
{
    const size_t sp = (size_t)tn_splits(B * T, D, Din);
    return sizeof(float) * ((size_t)B * T * D + sp * ((size_t)D * Din + D) + 256);
}

// df [B*T][D] -> dW [D][Din], dbias [D], dpe [T][D], dfs [B][D]   (x receives no gradient: it is the input feature).
// In two calls on the same ws: dW == NULL -> the inputs half only (dfs; the masked gradient stays in ws); df == NULL -> the weights half
// (dW, dbias, dpe from ws as the inputs half left it).
extern "C" int smin_video_encoder_bwd(void* stream, const float* df, const float* fv, const float* fs, const float* vmask, const float* x,
                                      int B, int T, int Din, int D, float* dW, float* dbias, float* dpe, float* dfs, void* ws, size_t ws_bytes)
{
    hipStream_t st = (hipStream_t)stream;
    SMIN_REQUIRE(Din % 4 == 0 && D % 4 == 0 && B >= 1 && T >= 1);
    SMIN_REQUIRE(ws_bytes >= smin_video_encoder_bwd_workspace_bytes(B, T, Din, D));
    float* dv = reinterpret_cast<float*>(ws);
    const int R = B * T, D4 = D / 4, sp = tn_splits(R, D, Din);
    float* slab = dv + (size_t)R * D;
    float* bslab = slab + (size_t)sp * D * Din;
    SMIN_REQUIRE(df != nullptr || dW != nullptr);
    if (df) {                                                          // inputs half: dv (kept in ws for the weights half), dfs
        hipLaunchKernelGGL(video_enc_bwd_rows_kernel, dim3(cdiv(D4, 64), B), dim3(64, VE_PH), 0, st, df, fv, fs, vmask, T, D4, dv, dfs);
        SMIN_LAUNCH_CHECK();
    }
    if (!dW) return 0;
    hipLaunchKernelGGL(video_enc_bwd_pe_kernel, dim3(cdiv(D4, 64), T), dim3(64), 0, st, dv, B, T, D4, dpe);
    SMIN_LAUNCH_CHECK();
    int rc = launch_gemm_tn(st, PlainMat{dv, D}, PlainMat{x, Din}, slab, bslab, R, D, Din, sp); if (rc) return rc;
    return launch_reduce_slabs2(st, slab, dW, D * Din, bslab, dbias, D, sp);
}
