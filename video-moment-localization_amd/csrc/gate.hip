// Gated moment feature shared by the content and boundary units (reference models.py:191, 272-274):
//   hbar[n,:] = sigmoid(f_m[n,:] * f_s[b,:]) * f_m[n,:]
// computed once per layer (element-wise, HBM-bound) instead of inside every consumer.  The backward kernel also sums
// the gradients of hbar's two consumers and of the residual copy of f_m, which autograd would otherwise add in
// separate full-size passes.
#include "common.h"
#include "smin_hip.h"

namespace smin {

// SUM: also hsum_out = hsum_in + hbar (the running sum of the layers' gated features that the content stream's gate term reads;
// a separate full-size add per layer otherwise)
template <bool SUM>
__global__ void gate_fwd_kernel(const float* __restrict__ fm, const float* __restrict__ fs, const int* __restrict__ cells,
                                int N, int D4, float* __restrict__ hbar, const float* __restrict__ hsum_in, float* __restrict__ hsum_out)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)N * D4) return;
    const size_t n = idx / D4; const int d4 = (int)(idx % D4);
    const int b = cells[4 * n];
    const float4 x = ldg4(fm + idx * 4), s = ldg4(fs + ((size_t)b * D4 + d4) * 4);
    const float4 h = make_float4(x.x / (1.0f + expf(-x.x * s.x)), x.y / (1.0f + expf(-x.y * s.y)),
                                 x.z / (1.0f + expf(-x.z * s.z)), x.w / (1.0f + expf(-x.w * s.w)));
    stg4(hbar + idx * 4, h);
    if (SUM) stg4(hsum_out + idx * 4, f4add(ldg4(hsum_in + idx * 4), h));
}

// dfm = dh * (g + fm*g*(1-g)*fs) ; partial[b][chunk][:] = sum over the chunk's cells of dh * fm^2 * g*(1-g)
// grid (chunks, B), 128 threads, float4 columns.
struct GradList { const float* p[4]; int n; };   // up to four gradient tensors of one value, summed on the fly

__device__ __forceinline__ float4 grad_sum(const GradList& g, size_t off) {
    float4 v = ldg4(g.p[0] + off);
    if (g.n > 1) v = f4add(v, ldg4(g.p[1] + off));
    if (g.n > 2) v = f4add(v, ldg4(g.p[2] + off));
    if (g.n > 3) v = f4add(v, ldg4(g.p[3] + off));
    return v;
}

__global__ __launch_bounds__(128)
void gate_bwd_kernel(GradList dh, GradList dres, const float* __restrict__ fm, const float* __restrict__ fs,
                     const int* __restrict__ row_ptr, int L, int D, int cells_per_chunk, int max_chunks,
                     float* __restrict__ dfm, float* __restrict__ partial,
                     const int* __restrict__ cells, const float* __restrict__ bA, const float* __restrict__ dbm)
{
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int s0 = row_ptr[b * L], s1 = row_ptr[(b + 1) * L];
    const int n_begin = s0 + chunk * cells_per_chunk;
    if (n_begin >= s1) return;
    const int n_end = min(s1, n_begin + cells_per_chunk);
    for (int d = threadIdx.x * 4; d < D; d += 512) {
        const float4 s4 = ldg4(fs + (size_t)b * D + d);
        float4 acc = f4zero();
        // four cells per trip: all their loads (up to eight gradient tensors + fm each) are requested before the first is used --
        // one cell per trip left a thread with one HBM round trip per cell (4.1 TB/s on 1.2 GB)
        for (int n0 = n_begin; n0 < n_end; n0 += 4) {
            float4 ds[4], x[4], dr[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int nn = min(n0 + u, n_end - 1);
                const size_t off = (size_t)nn * D + d;
                ds[u] = grad_sum(dh, off);                          // every consumer of hbar (summed here, not by autograd)
                if (bA) {
                    // ... the boundary unit's gated row reduction among them (models.py:191-194: f_bm[b,i] = sum_j A[b,i,j] hbar[b,i,j]):
                    // its gradient A[b,i,j] * dbm[b,i,:] is a broadcast of a per-row vector, formed here from [B][L][L] + [B][L][D]
                    // operands instead of being written by the boundary unit and read back (2 x 206 MB per layer at the bench shape)
                    const int ci = cells[4 * (size_t)nn + 1], cj = cells[4 * (size_t)nn + 2];
                    const float a = bA[((size_t)b * L + ci) * L + cj];
                    ds[u] = f4fma(ldg4(dbm + ((size_t)b * L + ci) * D + d), a, ds[u]);
                }
                x[u] = ldg4(fm + off);
                dr[u] = dres.n > 0 ? grad_sum(dres, off) : f4zero();   // gradients of the pass-through copies of f_m
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (n0 + u >= n_end) break;
                float4 o;
#define GATE1(F)                                                                  \
                {                                                                 \
                    const float g = 1.0f / (1.0f + expf(-x[u].F * s4.F));         \
                    const float gg = g * (1.0f - g);                              \
                    o.F = ds[u].F * (g + x[u].F * gg * s4.F);                     \
                    acc.F = fmaf(ds[u].F, x[u].F * x[u].F * gg, acc.F);           \
                }
                GATE1(x) GATE1(y) GATE1(z) GATE1(w)
#undef GATE1
                stg4(dfm + (size_t)(n0 + u) * D + d, f4add(o, dr[u]));
            }
        }
        stg4(partial + ((size_t)b * max_chunks + chunk) * D + d, acc);
    }
}

// second stage of the per-sample reduction: grid (D/4 / 64, B), 64 x SPR_PH threads; lane = feature group, threadIdx.y = phase over the
// chunks (k = ph, ph + SPR_PH, ..), the phases' sums meet in LDS in phase order (fixed order: deterministic).  One thread per feature
// summing all chunks serially ran 53 us on 128 workgroups, three times per step on the critical path.
constexpr int SPR_PH = 8;
__global__ __launch_bounds__(64 * SPR_PH)
void sample_partial_reduce_kernel(const float* __restrict__ partial, const int* __restrict__ row_ptr, int L, int D,
                                  int cells_per_chunk, int max_chunks, float* __restrict__ out)
{
    __shared__ float4 part[SPR_PH][64];
    const int b = blockIdx.y, ph = threadIdx.y;
    const int d = (blockIdx.x * 64 + threadIdx.x) * 4;
    const int ncell = row_ptr[(b + 1) * L] - row_ptr[b * L];
    const int nch = (ncell + cells_per_chunk - 1) / cells_per_chunk;
    float4 s = f4zero();
    if (d < D)
        for (int k = ph; k < nch; k += SPR_PH) s = f4add(s, ldg4(partial + ((size_t)b * max_chunks + k) * D + d));
    part[ph][threadIdx.x] = s;
    __syncthreads();
    if (ph == 0 && d < D) {
        for (int q = 1; q < SPR_PH; ++q) s = f4add(s, part[q][threadIdx.x]);
        stg4(out + (size_t)b * D + d, s);
    }
}

}  // namespace smin

using namespace smin;

extern "C" int smin_gate_fwd(void* stream, const float* fm, const float* fs, const int32_t* cells, int N, int D, float* hbar)
{
    SMIN_REQUIRE(D % 4 == 0);
    if (N == 0) return 0;
    const size_t tot = (size_t)N * (D / 4);
    hipLaunchKernelGGL(gate_fwd_kernel<false>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, fm, fs, cells, N, D / 4, hbar,
                       (const float*)nullptr, (float*)nullptr);
    SMIN_LAUNCH_CHECK();
    return 0;
}

extern "C" int smin_gate_fwd_sum(void* stream, const float* fm, const float* fs, const int32_t* cells, int N, int D, float* hbar, const float* hsum_in,
                                 float* hsum_out)
{
    SMIN_REQUIRE(D % 4 == 0 && hsum_in != nullptr && hsum_out != nullptr);
    if (N == 0) return 0;
    const size_t tot = (size_t)N * (D / 4);
    hipLaunchKernelGGL(gate_fwd_kernel<true>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, fm, fs, cells, N, D / 4, hbar, hsum_in,
                       hsum_out);
    SMIN_LAUNCH_CHECK();
    return 0;
}

extern "C" int smin_gate_bwd(void* stream, const float* const* dhbar, int n_dhbar, const float* const* dres, int n_dres,
                             const float* fm, const float* fs, const int32_t* row_ptr,
                             int N, int B, int L, int D, float* dfm, float* dfs, void* ws, size_t ws_bytes,
                             const int32_t* cells, const float* boundary_A, const float* boundary_dout)
{
    SMIN_REQUIRE((boundary_A == nullptr) == (boundary_dout == nullptr) && (boundary_A == nullptr || cells != nullptr || N == 0));
    hipStream_t st = (hipStream_t)stream;
    SMIN_REQUIRE(D % 4 == 0 && n_dhbar >= 1 && n_dhbar <= 4 && n_dres >= 0 && n_dres <= 4);
    int cpc, mc; chunking_fine(L, &cpc, &mc);
    SMIN_REQUIRE(ws_bytes >= sizeof(float) * (size_t)B * mc * D);
    float* partial = reinterpret_cast<float*>(ws);
    GradList gh, gr;
    for (int k = 0; k < 4; ++k) { gh.p[k] = dhbar[k < n_dhbar ? k : 0]; gr.p[k] = n_dres > 0 ? dres[k < n_dres ? k : 0] : nullptr; }
    gh.n = n_dhbar; gr.n = n_dres;
    hipLaunchKernelGGL(gate_bwd_kernel, dim3(mc, B), dim3(128), 0, st, gh, gr, fm, fs, row_ptr, L, D, cpc, mc, dfm, partial, cells, boundary_A, boundary_dout);
    SMIN_LAUNCH_CHECK();
    hipLaunchKernelGGL(sample_partial_reduce_kernel, dim3(cdiv(D / 4, 64), B), dim3(64, SPR_PH), 0, st, partial, row_ptr, L, D, cpc, mc, dfs);
    SMIN_LAUNCH_CHECK();
    return 0;
}
