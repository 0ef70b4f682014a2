// Bidirectional LSTM layer of the query encoder (reference models.py:38-64: nn.LSTM(300, H, num_layers=2,
// bidirectional, batch_first) over pack_padded_sequence / pad_packed_sequence), forward and backward.
//
// The library path (MIOpen through torch) issues ~8 launches per time step and direction -- ~600 launches per train
// step, each a few microseconds of work, so the host cannot keep the GPU fed (4.3 ms of wall time per step for 0.3 ms
// of arithmetic).  Here one launch runs the whole recurrence of a layer, both directions:
//   * the input projections of all time steps and both directions are one GEMM on the MFMA engine (gemm.h);
//   * the recurrence is split over the batch (samples are independent), never over hidden units, so no workgroup
//     ever waits for another: grid = (ceil(B / 4), 2 directions), 4 threads per hidden unit (a quarter of the
//     contraction each, 4 waves per SIMD hide the weight stream's latency).  W_hh (1 MB at H = 256) is re-streamed
//     from L2 every step by every workgroup; h lives in LDS, c in registers;
//   * sequence lengths are honoured in-kernel: the forward direction stops at len_b, the reverse direction starts at
//     position len_b - 1, padded positions are written as zeros -- exactly the packed-sequence result, without
//     gathers, masks or a host copy of the lengths.
// Backward: one launch for the recurrence of the gate gradients, then the weight gradients as GEMMs over all
// (sample, position) rows.
#include "gemm.h"
#include "smin_hip.h"

namespace smin {

// the same recurrences with W_hh resident in LDS, split over clusters of workgroups (bilstm_cluster.hip); H a multiple of 32
bool bilstm_cluster_ok(int B, int Nq, int H);
int launch_bilstm_cluster_fwd(hipStream_t st, float* G, const float* W4, const int* len, int B, int Nq, int H, float* Hout, float* Cs);
int launch_bilstm_cluster_bwd(hipStream_t st, const float* dHout, const float* G, const float* Cs, const float* Whh, const int* len, int B, int Nq, int H,
                              float* dG);

constexpr int LSTM_BS = 4;          // samples per workgroup

__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }

struct EpBiasRows {
    const float* bias; float* out;
    __device__ __forceinline__ void chunk(const float* Ws, int row0, int col0, int ncols, int M, int N, int lane) const {
        chunk_rows_f4(Ws, row0, col0, ncols, M, N, lane, [&](int row, int col, float4 v) {
            stg4(out + (size_t)row * N + col, f4add(v, ldg4(bias + col)));
        });
    }
};
struct EpStoreLstm {
    float* out;
    __device__ __forceinline__ void chunk(const float* Ws, int row0, int col0, int ncols, int M, int N, int lane) const {
        chunk_rows_f4(Ws, row0, col0, ncols, M, N, lane, [&](int row, int col, float4 v) { stg4(out + (size_t)row * N + col, v); });
    }
};

typedef float f32x2 __attribute__((ext_vector_type(2)));

// Thread roles (H <= 256, Hp = H rounded up to 64, 4*Hp threads): in the contraction phase thread (kq, u) sums its
// quarter of the contraction index for unit u and all LSTM_BS samples (one float4 of W per step of the loop: the four
// gates of unit u, so the weight stream is 16-byte loads coalesced over u); the partial sums meet in LDS and thread
// (b, u) = (kq, u) finishes sample b -- the cell state of (b, u) lives in that thread's registers for the whole launch.
//
// G    [B][Nq][2][4H]  in: input projections + biases (gate order i, f, g, o); out: the gate activations
// W4   [2][H][H][4]    W4[d][k][u][g] = W_hh_d[g*H + u][k]
// Hout [B][Nq][2H]     Cs [B][Nq][2][H]
__global__ __launch_bounds__(1024)
void bilstm_fwd_kernel(float* __restrict__ G, const float* __restrict__ W4, const int* __restrict__ len, int B, int Nq, int H,
                       float* __restrict__ Hout, float* __restrict__ Cs)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int BS = LSTM_BS;
    const int Hp = blockDim.x >> 2;
    float* hs = lds;                                               // [H][BS]
    float* part = lds + (size_t)H * BS;                            // [4 kq][4 g][BS][Hp]
    const int d = blockIdx.y, b0 = blockIdx.x * BS, u = threadIdx.x % Hp, kq = threadIdx.x / Hp;
    const bool own = u < H;
    const int uu = own ? u : 0, H4 = 4 * H, kn = H / 4, k0 = kq * kn;
    const int bme = b0 + kq;                                       // the sample this thread finishes
    const int L = bme < B ? min(len[bme], Nq) : 0;
    float c = 0.f;
    if (kq == 0 && own) *reinterpret_cast<float4*>(hs + u * BS) = f4zero();
    __syncthreads();
    const float4* w = reinterpret_cast<const float4*>(W4) + ((size_t)d * H + k0) * H + uu;
    for (int s = 0; s < Nq; ++s) {
        const bool act = s < L;
        const int p = d == 0 ? s : L - 1 - s;
        const size_t row = (size_t)bme * Nq + (act ? p : 0);
        float gx[4];
        {
            const float* g = G + (row * 2 + d) * H4 + uu;
            const bool ld = act && own;
#pragma unroll
            for (int q = 0; q < 4; ++q) gx[q] = ld ? g[q * H] : 0.f;
        }
        f32x2 a[BS][2];
#pragma unroll
        for (int b = 0; b < BS; ++b) { a[b][0] = f32x2{0.f, 0.f}; a[b][1] = f32x2{0.f, 0.f}; }
#pragma unroll 16
        for (int k = 0; k < kn; ++k) {
            const float4 h4 = *reinterpret_cast<const float4*>(hs + (k0 + k) * BS);
            const float4 w4 = w[(size_t)k * H];
            const f32x2 w01 = {w4.x, w4.y}, w23 = {w4.z, w4.w};
            const float hv[4] = {h4.x, h4.y, h4.z, h4.w};
#pragma unroll
            for (int b = 0; b < BS; ++b) {
                const f32x2 hb = {hv[b], hv[b]};
                a[b][0] = __builtin_elementwise_fma(hb, w01, a[b][0]);
                a[b][1] = __builtin_elementwise_fma(hb, w23, a[b][1]);
            }
        }
#pragma unroll
        for (int b = 0; b < BS; ++b) {
            part[((kq * 4 + 0) * BS + b) * Hp + u] = a[b][0].x; part[((kq * 4 + 1) * BS + b) * Hp + u] = a[b][0].y;
            part[((kq * 4 + 2) * BS + b) * Hp + u] = a[b][1].x; part[((kq * 4 + 3) * BS + b) * Hp + u] = a[b][1].y;
        }
        __syncthreads();                                            // partial sums visible; everyone is done reading hs
        if (own && bme < B) {
            if (act) {
                float z[4];
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    z[q] = gx[q] + ((part[((0 * 4 + q) * BS + kq) * Hp + u] + part[((1 * 4 + q) * BS + kq) * Hp + u]) +
                                    (part[((2 * 4 + q) * BS + kq) * Hp + u] + part[((3 * 4 + q) * BS + kq) * Hp + u]));
                const float ig = sigm(z[0]), fg = sigm(z[1]), gg = tanhf(z[2]), og = sigm(z[3]);
                c = fmaf(fg, c, ig * gg);
                const float hn = og * tanhf(c);
                float* g = G + (row * 2 + d) * H4 + u;
                g[0] = ig; g[H] = fg; g[2 * H] = gg; g[3 * H] = og;
                Cs[(row * 2 + d) * H + u] = c;
                Hout[row * 2 * H + d * H + u] = hn;
                hs[u * BS + kq] = hn;
            } else {
                Hout[((size_t)bme * Nq + s) * 2 * H + d * H + u] = 0.f;            // padded position s >= len
            }
        }
        __syncthreads();
    }
}

// dG [B][Nq][2][4H] = gradient of the pre-activation gates (zero at padded positions);  Whh [2][4H][H] as nn.LSTM stores it.
// Gate phase: thread (b, u) = (tid / Hp, tid % Hp) as in the forward.  Contraction phase dh[b][u] = sum_j dg[b][j] W[j][u]:
// thread (jc, uq) takes a sixteenth of the 4H rows j for the four units 4uq .. 4uq+3 -- one float4 of W (coalesced over
// uq) and one float4 of dg (the four samples, an LDS broadcast) feed 16 FMAs, so the LDS pipe issues a quarter of the
// reads a one-unit-per-thread split needs (it was the bound of this kernel).
__global__ __launch_bounds__(1024)
void bilstm_bwd_kernel(const float* __restrict__ dHout, const float* __restrict__ G, const float* __restrict__ Cs,
                       const float* __restrict__ Whh, const int* __restrict__ len, int B, int Nq, int H, float* __restrict__ dG)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int BS = LSTM_BS;
    const int Hp = blockDim.x >> 2;
    float* dgs = lds;                                              // [4H][BS]
    float* part = lds + (size_t)4 * H * BS;                        // [16 jc][BS][Hp]
    const int d = blockIdx.y, b0 = blockIdx.x * BS, u = threadIdx.x % Hp, jq = threadIdx.x / Hp;
    const bool own = u < H;
    const int H4 = 4 * H;
    const int bme = b0 + jq;
    const int L = bme < B ? min(len[bme], Nq) : 0;
    const int nq = Hp >> 2, uq = threadIdx.x % nq, jc = threadIdx.x / nq;       // contraction roles: 16 row chunks x Hp/4 unit quads
    const int rows = H4 / 16, jr0 = jc * rows;
    const bool uok = 4 * uq < H;
    const float4* wq = reinterpret_cast<const float4*>(Whh + (size_t)d * H4 * H) + (uok ? uq : 0);
    const int ldw = H / 4;
    float dhn = 0.f, dcn = 0.f;
    for (int s = Nq - 1; s >= 0; --s) {
        const bool act = s < L;
        float dg[4] = {0.f, 0.f, 0.f, 0.f};
        if (own && bme < B) {
            if (act) {
                const int p = d == 0 ? s : L - 1 - s;
                const size_t row = (size_t)bme * Nq + p;
                const float* g = G + (row * 2 + d) * H4 + u;
                const float ig = g[0], fg = g[H], gg = g[2 * H], og = g[3 * H];
                const float ct = Cs[(row * 2 + d) * H + u];
                const float cp = s > 0 ? Cs[(((size_t)bme * Nq + (d == 0 ? p - 1 : p + 1)) * 2 + d) * H + u] : 0.f;
                const float dh = dHout[row * 2 * H + d * H + u] + dhn;
                const float tc = tanhf(ct);
                const float dc = fmaf(dh * og, 1.0f - tc * tc, dcn);
                dg[0] = dc * gg * ig * (1.0f - ig);
                dg[1] = dc * cp * fg * (1.0f - fg);
                dg[2] = dc * ig * (1.0f - gg * gg);
                dg[3] = dh * tc * og * (1.0f - og);
                dcn = dc * fg;
                float* o = dG + (row * 2 + d) * H4 + u;
                o[0] = dg[0]; o[H] = dg[1]; o[2 * H] = dg[2]; o[3 * H] = dg[3];
            } else {
                float* o = dG + ((((size_t)bme * Nq + s) * 2 + d) * H4) + u;       // padded position s >= len
                o[0] = 0.f; o[H] = 0.f; o[2 * H] = 0.f; o[3 * H] = 0.f;
            }
        }
        if (own) {
#pragma unroll
            for (int q = 0; q < 4; ++q) dgs[(size_t)(q * H + u) * BS + jq] = dg[q];
        }
        __syncthreads();
        f32x2 a[BS][2];
#pragma unroll
        for (int b = 0; b < BS; ++b) { a[b][0] = f32x2{0.f, 0.f}; a[b][1] = f32x2{0.f, 0.f}; }
#pragma unroll 8
        for (int j = 0; j < rows; ++j) {
            const float4 w4 = wq[(size_t)(jr0 + j) * ldw];
            const float4 g4 = *reinterpret_cast<const float4*>(dgs + (size_t)(jr0 + j) * BS);
            const f32x2 w01 = {w4.x, w4.y}, w23 = {w4.z, w4.w};
            const float gv[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
            for (int b = 0; b < BS; ++b) {
                const f32x2 gb = {gv[b], gv[b]};
                a[b][0] = __builtin_elementwise_fma(gb, w01, a[b][0]);
                a[b][1] = __builtin_elementwise_fma(gb, w23, a[b][1]);
            }
        }
        if (uok) {
#pragma unroll
            for (int b = 0; b < BS; ++b)
                *reinterpret_cast<float4*>(part + (size_t)(jc * BS + b) * Hp + 4 * uq) = make_float4(a[b][0].x, a[b][0].y, a[b][1].x, a[b][1].y);
        }
        __syncthreads();
        if (act && own) {
            float sum = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) sum += part[(size_t)(k * BS + jq) * Hp + u];
            dhn = sum;
        }
        // no third barrier: dgs is rewritten only after every thread has passed the barrier above (its reads are over),
        // and part only after the next step's first barrier, which every thread reaches after the reads of this line
    }
}

// The operand layouts of the layer kernels from nn.LSTM's eight parameter tensors of a layer, in one launch (as torch calls: two
// concatenations, two additions, a stack and a permuted copy per layer, ahead of the first kernel of the step's critical path):
//   Wih [8H][In] = [w_ih; w_ih_reverse],  bias [8H] = [b_ih + b_hh; b_ih_reverse + b_hh_reverse],
//   Whh [2][4H][H] = {w_hh, w_hh_reverse},  W4 [2][H][H][4] with W4[d][k][u][g] = w_hh_d[g H + u][k]
struct LstmRaw { const float* w[8]; };              // w_ih, w_hh, b_ih, b_hh, then the same four of the reverse direction
// (pack_layer: one layer's tensors; the kernels below are thin wrappers -- one layer, or every layer of the encoder in one launch)
__device__ __forceinline__ void pack_layer(const LstmRaw& R, int In, int H, float* __restrict__ Wih, float* __restrict__ bias, float* __restrict__ Whh,
                                           float* __restrict__ W4)
{
    const size_t nWih = (size_t)8 * H * In, nB = (size_t)8 * H, nWhh = (size_t)2 * 4 * H * H;
    const size_t tot = nWih + nB + 2 * nWhh;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (size_t)gridDim.x * blockDim.x) {
        if (e < nWih) {
            const size_t half = (size_t)4 * H * In;
            Wih[e] = e < half ? R.w[0][e] : R.w[4][e - half];
        } else if (e < nWih + nB) {
            const size_t j = e - nWih;
            const int dir = j >= (size_t)4 * H;
            const size_t jj = j - (size_t)dir * 4 * H;
            bias[j] = R.w[dir * 4 + 2][jj] + R.w[dir * 4 + 3][jj];
        } else if (e < nWih + nB + nWhh) {
            const size_t j = e - nWih - nB, half = (size_t)4 * H * H;
            Whh[j] = j < half ? R.w[1][j] : R.w[5][j - half];
        } else {
            const size_t j = e - nWih - nB - nWhh;                  // W4 index ((d H + k) H + u) 4 + g
            const int g = (int)(j & 3);
            const size_t r = j >> 2;
            const int u = (int)(r % H), k = (int)((r / H) % H), d = (int)(r / ((size_t)H * H));
            W4[j] = R.w[d * 4 + 1][((size_t)g * H + u) * H + k];
        }
    }
}
__global__ void lstm_pack_kernel(LstmRaw R, int In, int H, float* __restrict__ Wih, float* __restrict__ bias, float* __restrict__ Whh,
                                 float* __restrict__ W4)
{
    pack_layer(R, In, H, Wih, bias, Whh, W4);
}
constexpr int LSTM_PACK_MAXL = 4;
struct LstmPackAll { LstmRaw raw[LSTM_PACK_MAXL]; int In[LSTM_PACK_MAXL]; float* out[LSTM_PACK_MAXL][4]; };
__global__ void lstm_pack_layers_kernel(LstmPackAll A, int H)
{
    const int l = blockIdx.y;                                    // a select chain, not an indexed read of the argument block
    LstmRaw R = A.raw[0]; int In = A.In[0]; float* o0 = A.out[0][0]; float* o1 = A.out[0][1]; float* o2 = A.out[0][2]; float* o3 = A.out[0][3];
#pragma unroll
    for (int k = 1; k < LSTM_PACK_MAXL; ++k)
        if (l == k) { R = A.raw[k]; In = A.In[k]; o0 = A.out[k][0]; o1 = A.out[k][1]; o2 = A.out[k][2]; o3 = A.out[k][3]; }
    pack_layer(R, In, H, o0, o1, o2, o3);
}


// Sentence feature of the query encoder (models.py:60-62): f_s[b] = [h_fwd at the last word | h_bwd at the first word]
//   fs[b][d] = d < H ? fw[b][max(len_b - 1, 0)][d] : fw[b][0][d];   backward: the same entries of dfw += dfs
__global__ void sentence_feature_kernel(const float* __restrict__ fw, const int* __restrict__ len, int B, int Nq, int H, float* __restrict__ fs)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * 2 * H) return;
    const int b = idx / (2 * H), d = idx % (2 * H);
    const int p = d < H ? max(min(len[b], Nq) - 1, 0) : 0;
    fs[idx] = fw[((size_t)b * Nq + p) * 2 * H + d];
}
__global__ void sentence_feature_bwd_kernel(const float* __restrict__ dfs, const int* __restrict__ len, int B, int Nq, int H, float* __restrict__ dfw)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * 2 * H) return;
    const int b = idx / (2 * H), d = idx % (2 * H);
    const int p = d < H ? max(min(len[b], Nq) - 1, 0) : 0;
    dfw[((size_t)b * Nq + p) * 2 * H + d] += dfs[idx];             // one writer per entry
}

}  // namespace smin

using namespace smin;

extern "C" int smin_bilstm_layer_fwd(void* stream, const float* X, const float* Wih_cat, const float* bias_cat, const float* W4,
                                     const int32_t* len, int B, int Nq, int In, int H, float* G, float* Hout, float* Cs)
{
    hipStream_t st = (hipStream_t)stream;
    SMIN_REQUIRE(In % 4 == 0 && H % 4 == 0 && H >= 4 && H <= 256 && B >= 1 && Nq >= 1);
    int rc = launch_gemm_nt(st, PlainMat{X, In}, PlainMat{Wih_cat, In}, EpBiasRows{bias_cat, G}, B * Nq, 8 * H, In);
    if (rc) return rc;
    if (bilstm_cluster_ok(B, Nq, H)) return launch_bilstm_cluster_fwd(st, G, W4, len, B, Nq, H, Hout, Cs);
    const int Hp = cdiv(H, 64) * 64;
    const size_t lds = sizeof(float) * ((size_t)H * LSTM_BS + (size_t)16 * LSTM_BS * Hp);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(bilstm_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(bilstm_fwd_kernel, dim3(cdiv(B, LSTM_BS), 2), dim3(4 * Hp), lds, st, G, W4, len, B, Nq, H, Hout, Cs);
    SMIN_LAUNCH_CHECK();
    return 0;
}

extern "C" int smin_sentence_feature_fwd(void* stream, const float* fw, const int32_t* len, int B, int Nq, int H, float* fs)
{
    SMIN_REQUIRE(B >= 1 && Nq >= 1 && H >= 1);
    hipLaunchKernelGGL(sentence_feature_kernel, dim3(cdiv(B * 2 * H, 256)), dim3(256), 0, (hipStream_t)stream, fw, len, B, Nq, H, fs);
    SMIN_LAUNCH_CHECK();
    return 0;
}
extern "C" int smin_sentence_feature_bwd(void* stream, const float* dfs, const int32_t* len, int B, int Nq, int H, float* dfw)
{
    SMIN_REQUIRE(B >= 1 && Nq >= 1 && H >= 1);
    hipLaunchKernelGGL(sentence_feature_bwd_kernel, dim3(cdiv(B * 2 * H, 256)), dim3(256), 0, (hipStream_t)stream, dfs, len, B, Nq, H, dfw);
    SMIN_LAUNCH_CHECK();
    return 0;
}

extern "C" int smin_lstm_pack(void* stream, const float* const* w, int In, int H, float* Wih, float* bias, float* Whh, float* W4)
{
    SMIN_REQUIRE(In >= 1 && H >= 1);
    LstmRaw R;
    for (int q = 0; q < 8; ++q) { SMIN_REQUIRE(w[q] != nullptr); R.w[q] = w[q]; }
    hipLaunchKernelGGL(lstm_pack_kernel, dim3(512), dim3(256), 0, (hipStream_t)stream, R, In, H, Wih, bias, Whh, W4);
    SMIN_LAUNCH_CHECK();
    return 0;
}

// every layer of the encoder in one launch (w: HOST array of 8 * nlayers device pointers, layer by layer in smin_lstm_pack's order;
// In: HOST array of the layers' input widths; Wih / bias / Whh / W4: HOST arrays of nlayers device pointers)
extern "C" int smin_lstm_pack_layers(void* stream, int nlayers, const float* const* w, const int* In, int H, float* const* Wih, float* const* bias,
                                     float* const* Whh, float* const* W4)
{
    SMIN_REQUIRE(nlayers >= 1 && nlayers <= LSTM_PACK_MAXL && H >= 1);
    LstmPackAll A;
    for (int l = 0; l < LSTM_PACK_MAXL; ++l) {
        const int s = l < nlayers ? l : 0;
        for (int q = 0; q < 8; ++q) { SMIN_REQUIRE(w[8 * s + q] != nullptr); A.raw[l].w[q] = w[8 * s + q]; }
        SMIN_REQUIRE(In[s] >= 1);
        A.In[l] = In[s];
        A.out[l][0] = Wih[s]; A.out[l][1] = bias[s]; A.out[l][2] = Whh[s]; A.out[l][3] = W4[s];
    }
    hipLaunchKernelGGL(lstm_pack_layers_kernel, dim3(256, nlayers), dim3(256), 0, (hipStream_t)stream, A, H);
    SMIN_LAUNCH_CHECK();
    return 0;
}

// floats of the layer backward's workspace up to (not including) the split-K slab of the input-gradient contraction
static size_t lstm_bwd_ws_floats(int B, int Nq, int In, int H)
{
    const int R = B * Nq;
    const size_t sp1 = (size_t)tn_splits(R, 8 * H, In), sp2 = (size_t)tn_splits(R, 4 * H, H);
    return (size_t)R * 8 * H + (size_t)2 * R * H + sp1 * ((size_t)8 * H * In + 8 * H) + 2 * sp2 * (size_t)4 * H * H + 256;
}
extern "C" size_t smin_bilstm_layer_bwd_workspace_bytes(int B, int Nq, int In, int H)
{
    return sizeof(float) * (lstm_bwd_ws_floats(B, Nq, In, H) + (size_t)16 * B * Nq * In);     // + up to 16 partial products of dX
}

// dHout [B][Nq][2H] -> dX [B*Nq][In] (NULL to skip), dWih_cat [8H][In], dbias_cat [8H], dWhh [2][4H][H].
// In two calls on the same ws: dWih_cat == NULL -> the inputs half (recurrence and dX; the gate gradients stay in ws);
// dHout == NULL -> the weights half (dWih_cat, dbias_cat, dWhh from the gate gradients in ws).
extern "C" int smin_bilstm_layer_bwd(void* stream, const float* dHout, const float* X, const float* Hout, const float* G, const float* Cs,
                                     const float* Wih_catT, const float* Whh, const int32_t* len, int B, int Nq, int In, int H,
                                     float* dX, float* dWih_cat, float* dbias_cat, float* dWhh, void* ws, size_t ws_bytes)
{
    hipStream_t st = (hipStream_t)stream;
    SMIN_REQUIRE(In % 4 == 0 && H % 4 == 0 && H >= 4 && H <= 256 && B >= 1 && Nq >= 1);
    SMIN_REQUIRE(ws_bytes >= smin_bilstm_layer_bwd_workspace_bytes(B, Nq, In, H));
    const int R = B * Nq, H4 = 4 * H, H8 = 8 * H;
    float* w = reinterpret_cast<float*>(ws);
    float* dG = w;
    const int Hp = cdiv(H, 64) * 64;
    const size_t lds = sizeof(float) * ((size_t)H4 * LSTM_BS + (size_t)16 * LSTM_BS * Hp);
    SMIN_REQUIRE(H4 % 16 == 0);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(bilstm_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    SMIN_REQUIRE(dHout != nullptr || dWih_cat != nullptr);
    int rc;
    if (dHout) {                                                       // inputs half: the recurrence (gate gradients stay in ws), dX
        if (bilstm_cluster_ok(B, Nq, H)) {
            rc = launch_bilstm_cluster_bwd(st, dHout, G, Cs, Whh, len, B, Nq, H, dG);
            if (rc) return rc;
        } else {
            hipLaunchKernelGGL(bilstm_bwd_kernel, dim3(cdiv(B, LSTM_BS), 2), dim3(4 * Hp), lds, st, dHout, G, Cs, Whh, len, B, Nq, H, dG);
            SMIN_LAUNCH_CHECK();
        }
        if (dX) {
            const int sk = nt_splitk_splits(R, In, H8);
            if (sk > 1) rc = launch_gemm_nt_splitk(st, dG, H8, Wih_catT, H8, dX, R, In, H8, sk, w + lstm_bwd_ws_floats(B, Nq, In, H));
            else rc = launch_gemm_nt(st, PlainMat{dG, H8}, PlainMat{Wih_catT, H8}, EpStoreLstm{dX}, R, In, H8);
            if (rc) return rc;
        }
    }
    if (!dWih_cat) return 0;
    return smin_bilstm_layer_bwd_weights(stream, 7, X, Hout, B, Nq, In, H, dWih_cat, dbias_cat, nullptr, dWhh, ws, ws_bytes);
}

// The weights half in up to three independent pieces (which: bit 0 = dWih_cat + dbias_cat, bit 1 = dWhh[0], bit 2 = dWhh[1]), each
// with its own part of ws: issued on three streams they run side by side -- they are the last kernels of a train step, one after
// the other they were ~0.2 ms of seven short launches behind the last recurrence.
// dbias_cat2 (NULL to skip): a second copy of dbias_cat (nn.LSTM has two bias vectors per direction with the same gradient; each
// parameter's .grad needs memory of its own).
extern "C" int smin_bilstm_layer_bwd_weights(void* stream, int which, const float* X, const float* Hout, int B, int Nq, int In, int H,
                                             float* dWih_cat, float* dbias_cat, float* dbias_cat2, float* dWhh, void* ws, size_t ws_bytes)
{
    hipStream_t st = (hipStream_t)stream;
    SMIN_REQUIRE(In % 4 == 0 && H % 4 == 0 && H >= 4 && H <= 256 && B >= 1 && Nq >= 1 && which >= 1 && which <= 7);
    SMIN_REQUIRE(ws_bytes >= smin_bilstm_layer_bwd_workspace_bytes(B, Nq, In, H));
    const int R = B * Nq, H4 = 4 * H, H8 = 8 * H;
    float* w = reinterpret_cast<float*>(ws);
    float* dG = w;
    float* slab = dG + (size_t)R * H8 + (size_t)2 * R * H;       // (2 R H floats behind dG: formerly the shifted copy of Hout, unused)
    const int sp1 = tn_splits(R, H8, In), sp2 = tn_splits(R, H4, H);
    float* bslab = slab + (size_t)sp1 * H8 * In;
    float* slab2 = bslab + (size_t)sp1 * H8;
    int rc;
    if (which & 1) {
        SMIN_REQUIRE(dWih_cat != nullptr && dbias_cat != nullptr);
        rc = launch_gemm_tn(st, PlainMat{dG, H8}, PlainMat{X, In}, slab, bslab, R, H8, In, sp1); if (rc) return rc;
        rc = launch_reduce_slabs2(st, slab, dWih_cat, H8 * In, bslab, dbias_cat, H8, sp1, dbias_cat2); if (rc) return rc;
    }
    for (int d = 0; d < 2; ++d) {
        if (!(which & (2 << d))) continue;
        SMIN_REQUIRE(dWhh != nullptr);
        // h_{t-1} is read straight out of Hout by the operand loader (ShiftRowsMat): no shifted copy, no launch for it
        float* sl = slab2 + (size_t)d * sp2 * H4 * H;
        rc = launch_gemm_tn(st, PlainMat{dG + (size_t)d * H4, H8}, ShiftRowsMat{Hout, Nq, H, d}, sl, (float*)nullptr, R, H4, H, sp2);
        if (rc) return rc;
        rc = launch_reduce_slabs(st, sl, dWhh + (size_t)d * H4 * H, H4 * H, sp2); if (rc) return rc;
    }
    return 0;
}
