// Row-wise linear maps of the content stream:  y[r][:] = x[r][:] W^T + bias + add_rows[r][:] + add_cells[r / C][:]
// on the fp32 MFMA engine (gemm.h), forward and backward.  In the content stream (see the Python host) the content
// unit's two linear maps (reference models.py:247, 269) are composed in the dl-dimensional space, so every
// contraction here has K or N equal to dl instead of D.
#include "gemm.h"
#include "smin_hip.h"

namespace smin {

// HAS_*: which addends exist -- template parameters, not runtime pointer tests: a per-element "load or zero" select on a runtime
// condition makes hipcc branch around every load of the unrolled chunk (and wait for each), cdna_hip_programming.md 5, item 4(c)
template <bool HAS_BIAS, bool HAS_ROWS, bool HAS_CELLS>
struct EpLinearRows {
    const float* bias; const float* add_rows; const float* add_cells; int C; float* out;
    struct Add { float4 r, c; };
    __device__ __forceinline__ void chunk(const float* Ws, int row0, int col0, int ncols, int M, int N, int lane) const {
        chunk_rows_f4_pre<Add>(Ws, row0, col0, ncols, M, N, lane,
            [&](int row, int col) {
                Add a;
                a.r = HAS_ROWS ? ldg4(add_rows + (size_t)row * N + col) : f4zero();
                a.c = HAS_CELLS ? ldg4(add_cells + (size_t)(row / C) * N + col) : f4zero();
                return a;
            },
            [&](int row, int col, float4 v, const Add& a) {
                if (HAS_BIAS) v = f4add(v, ldg4(bias + col));
                if (HAS_ROWS) v = f4add(v, a.r);
                if (HAS_CELLS) v = f4add(v, a.c);
                stg4(out + (size_t)row * N + col, v);
            });
    }
};
template <bool ACC = false>
struct EpStoreRows {                // out = acc (ACC: out += acc, the second and later consumers of one tensor's gradient)
    float* out;
    __device__ __forceinline__ void chunk(const float* Ws, int row0, int col0, int ncols, int M, int N, int lane) const {
        chunk_rows_f4(Ws, row0, col0, ncols, M, N, lane, [&](int row, int col, float4 v) {
            float* o = out + (size_t)row * N + col;
            stg4(o, ACC ? f4add(v, ldg4(o)) : v);
        });
    }
};

template <bool ACC = false>
struct EpSplitCols {                // column block s of the result goes to out[s] [rows][w]
    float* out[4]; int w;
    __device__ __forceinline__ void chunk(const float* Ws, int row0, int col0, int ncols, int M, int N, int lane) const {
        chunk_rows_f4(Ws, row0, col0, ncols, M, N, lane, [&](int row, int col, float4 v) {
            const int sg = (col >= w) + (col >= 2 * w) + (col >= 3 * w);
            float* q = sg == 0 ? out[0] : sg == 1 ? out[1] : sg == 2 ? out[2] : out[3];
            float* o = q + (size_t)row * w + (col - sg * w);
            stg4(o, ACC ? f4add(v, ldg4(o)) : v);
        });
    }
};

// out[g][:] = sum_{c < C} x[g*C + c][:]
__global__ void group_sum_kernel(const float* __restrict__ x, float* __restrict__ out, size_t groups, int C, int W4)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= groups * W4) return;
    const size_t g = idx / W4; const int d4 = (int)(idx % W4);
    float4 s = f4zero();
    for (int c = 0; c < C; ++c) s = f4add(s, ldg4(x + ((g * C + c) * W4 + d4) * 4));
    stg4(out + idx * 4, s);
}

}  // namespace smin

using namespace smin;

static CatMat cat_of(const float* const* xs, int nseg, int K)
{
    CatMat m;
    for (int k = 0; k < 4; ++k) m.p[k] = xs[k < nseg ? k : 0];
    m.w = K;
    return m;
}

// y = [x_0 | .. | x_{nseg-1}] W^T + bias + add_rows + add_cells[r / C];  xs: HOST array of nseg <= 4 device pointers,
// each x_s [R][K]; W [O][nseg*K].
extern "C" int smin_linear_rows_fwd(void* stream, const float* const* xs, int nseg, const float* W, const float* bias, const float* add_rows,
                                    const float* add_cells, int C, int R, int O, int K, float* y)
{
    SMIN_REQUIRE(O % 4 == 0 && K % 4 == 0 && C >= 1 && nseg >= 1 && nseg <= 4);
    if (R == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    auto run = [&](auto ep) {
        if (nseg == 1) return launch_gemm_nt(st, PlainMat{xs[0], K}, PlainMat{W, K}, ep, R, O, K);
        return launch_gemm_nt(st, cat_of(xs, nseg, K), PlainMat{W, nseg * K}, ep, R, O, nseg * K);
    };
    // the combinations the hosts use: plain, bias, bias + rows + cells, rows only, bias + rows
    if (!add_rows && !add_cells) return bias ? run(EpLinearRows<true, false, false>{bias, add_rows, add_cells, C, y})
                                             : run(EpLinearRows<false, false, false>{bias, add_rows, add_cells, C, y});
    if (add_rows && add_cells) return bias ? run(EpLinearRows<true, true, true>{bias, add_rows, add_cells, C, y})
                                           : run(EpLinearRows<false, true, true>{bias, add_rows, add_cells, C, y});
    if (add_rows) return bias ? run(EpLinearRows<true, true, false>{bias, add_rows, add_cells, C, y})
                              : run(EpLinearRows<false, true, false>{bias, add_rows, add_cells, C, y});
    return bias ? run(EpLinearRows<true, false, true>{bias, add_rows, add_cells, C, y})
                : run(EpLinearRows<false, false, true>{bias, add_rows, add_cells, C, y});
}

extern "C" size_t smin_linear_rows_bwd_workspace_bytes(int R, int O, int Ktot)
{
    const int sp = R > 0 ? tn_splits(R, O, Ktot) : 1;
    return sizeof(float) * ((size_t)sp * O * Ktot + (size_t)sp * O + 64);
}

// dx_s = dy W_s (dxs: HOST array of nseg device pointers, or NULL to skip), dW [O][nseg*K] = dy^T [x_0 | ..],
// dbias = colsum(dy) (NULL to skip).  WT is W^T [nseg*K][O], row-major.
extern "C" int smin_linear_rows_bwd(void* stream, const float* dy, const float* const* xs, int nseg, const float* WT, int R, int O, int K,
                                    float* const* dxs, float* dW, float* dbias, void* ws, size_t ws_bytes)
{
    hipStream_t st = (hipStream_t)stream;
    SMIN_REQUIRE(O % 4 == 0 && K % 4 == 0 && nseg >= 1 && nseg <= 4);
    const int Kt = nseg * K;
    if (R == 0) {
        if (dW) (void)hipMemsetAsync(dW, 0, sizeof(float) * (size_t)O * Kt, st);
        if (dW && dbias) (void)hipMemsetAsync(dbias, 0, sizeof(float) * (size_t)O, st);
        return 0;
    }
    SMIN_REQUIRE(ws_bytes >= smin_linear_rows_bwd_workspace_bytes(R, O, Kt));
    SMIN_REQUIRE(dW != nullptr || dbias == nullptr);                   // the column sum rides on the weight-gradient pass
    int rc;
    if (dxs) {
        if (nseg == 1) {
            rc = launch_gemm_nt(st, PlainMat{dy, O}, PlainMat{WT, O}, EpStoreRows<false>{dxs[0]}, R, K, O);
        } else {
            EpSplitCols<false> ep;
            for (int k = 0; k < 4; ++k) ep.out[k] = dxs[k < nseg ? k : 0];
            ep.w = K;
            rc = launch_gemm_nt(st, PlainMat{dy, O}, PlainMat{WT, O}, ep, R, Kt, O);
        }
        if (rc) return rc;
    }
    if (!dW) return 0;                                                 // input gradients only (the weight half runs elsewhere)
    const int sp = tn_splits(R, O, Kt);
    float* slab = reinterpret_cast<float*>(ws);
    float* bslab = slab + (size_t)sp * O * Kt;
    if (nseg == 1) rc = launch_gemm_tn(st, PlainMat{dy, O}, PlainMat{xs[0], K}, slab, bslab, R, O, Kt, sp);
    else rc = launch_gemm_tn(st, PlainMat{dy, O}, cat_of(xs, nseg, K), slab, bslab, R, O, Kt, sp);
    if (rc) return rc;
    rc = launch_reduce_slabs2(st, slab, dW, O * Kt, bslab, dbias, O, sp); if (rc) return rc;
    return 0;
}

// The two calls above that read xs, for xs stored as bf16 (contraction-only operands under smin_set_gemm_mode(2): bit-identical results,
// half the operand bytes).  Forward: the combination the content stream uses for chat_k (rows and cells present, bias optional).
// Backward: the weights half only (input gradients never read xs: smin_linear_rows_bwd / _dx_acc with xs == NULL serve them).
static CatMatH cat_of_h(const uint16_t* const* xs, int nseg, int K)
{
    CatMatH m;
    for (int k = 0; k < 4; ++k) m.p[k] = xs[k < nseg ? k : 0];
    m.w = K;
    return m;
}

extern "C" int smin_linear_rows_fwd_xh(void* stream, const uint16_t* const* xs, int nseg, const float* W, const float* bias, const float* add_rows,
                                       const float* add_cells, int C, int R, int O, int K, float* y)
{
    SMIN_REQUIRE(O % 4 == 0 && K % 4 == 0 && C >= 1 && nseg >= 1 && nseg <= 4);
    if (R == 0) return 0;
    SMIN_REQUIRE(add_rows != nullptr && add_cells != nullptr);
    hipStream_t st = (hipStream_t)stream;
    auto run = [&](auto ep) {
        if (nseg == 1) return launch_gemm_nt(st, PlainMatH{xs[0], K}, PlainMat{W, K}, ep, R, O, K);
        return launch_gemm_nt(st, cat_of_h(xs, nseg, K), PlainMat{W, nseg * K}, ep, R, O, nseg * K);
    };
    return bias ? run(EpLinearRows<true, true, true>{bias, add_rows, add_cells, C, y}) : run(EpLinearRows<false, true, true>{bias, add_rows, add_cells, C, y});
}

extern "C" int smin_linear_rows_bwd_xh(void* stream, const float* dy, const uint16_t* const* xs, int nseg, int R, int O, int K,
                                       float* dW, float* dbias, void* ws, size_t ws_bytes)
{
    hipStream_t st = (hipStream_t)stream;
    SMIN_REQUIRE(O % 4 == 0 && K % 4 == 0 && nseg >= 1 && nseg <= 4 && dW != nullptr);
    const int Kt = nseg * K;
    if (R == 0) {
        (void)hipMemsetAsync(dW, 0, sizeof(float) * (size_t)O * Kt, st);
        if (dbias) (void)hipMemsetAsync(dbias, 0, sizeof(float) * (size_t)O, st);
        return 0;
    }
    SMIN_REQUIRE(ws_bytes >= smin_linear_rows_bwd_workspace_bytes(R, O, Kt));
    const int sp = tn_splits(R, O, Kt);
    float* slab = reinterpret_cast<float*>(ws);
    float* bslab = slab + (size_t)sp * O * Kt;
    int rc;
    if (nseg == 1) rc = launch_gemm_tn(st, PlainMat{dy, O}, PlainMatH{xs[0], K}, slab, bslab, R, O, Kt, sp);
    else rc = launch_gemm_tn(st, PlainMat{dy, O}, cat_of_h(xs, nseg, K), slab, bslab, R, O, Kt, sp);
    if (rc) return rc;
    return launch_reduce_slabs2(st, slab, dW, O * Kt, bslab, dbias, O, sp);
}

// dx_s += dy W_s for every segment: the input gradients of smin_linear_rows_bwd accumulated into tensors that already hold the
// gradient of an earlier consumer (a separate full-size add per step otherwise)
extern "C" int smin_linear_rows_dx_acc(void* stream, const float* dy, int nseg, const float* WT, int R, int O, int K, float* const* dxs)
{
    hipStream_t st = (hipStream_t)stream;
    SMIN_REQUIRE(O % 4 == 0 && K % 4 == 0 && nseg >= 1 && nseg <= 4 && dxs != nullptr);
    if (R == 0) return 0;
    if (nseg == 1) return launch_gemm_nt(st, PlainMat{dy, O}, PlainMat{WT, O}, EpStoreRows<true>{dxs[0]}, R, K, O);
    EpSplitCols<true> ep;
    for (int k = 0; k < 4; ++k) ep.out[k] = dxs[k < nseg ? k : 0];
    ep.w = K;
    return launch_gemm_nt(st, PlainMat{dy, O}, PlainMat{WT, O}, ep, R, nseg * K, O);
}

extern "C" int smin_group_sum(void* stream, const float* x, int groups, int C, int W, float* out)
{
    SMIN_REQUIRE(W % 4 == 0 && C >= 1);
    if (groups == 0) return 0;
    const size_t tot = (size_t)groups * (W / 4);
    hipLaunchKernelGGL(group_sum_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, out, (size_t)groups, C, W / 4);
    SMIN_LAUNCH_CHECK();
    return 0;
}
