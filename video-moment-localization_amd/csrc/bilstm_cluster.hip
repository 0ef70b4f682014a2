// The LSTM recurrence with W_hh resident on the chip (reference models.py:38-64, the nn.LSTM of the query encoder).
//
// bilstm.hip runs the recurrence of a (sample group, direction) in ONE workgroup that streams the whole W_hh (1 MB at H = 256)
// from L2 every time step: ~11 us per step at one CU's L2 bandwidth, 0.22 ms per layer and pass on 32 of the 256 CUs, four
// times on the critical path of a train step.  Here a *cluster* of P = H / 32 workgroups shares a (sample group, direction):
// workgroup p owns 32 hidden units -- their 4 x 32 gate rows of W_hh, 128 KB, stay in its LDS for the whole launch -- and the
// workgroups exchange what the next step needs through global memory:
//   forward   every workgroup needs all of h_t [H][4 samples]: each publishes its 32 x 4 values, all gather the rest;
//   backward  dh_{t-1}[u'] = sum_j dg_t[j] W_hh[j][u']: each workgroup contracts its own 128 gate rows against all H columns and
//             sends workgroup p' the partial sums of p's units; the owner adds the P partial sums in workgroup order (fixed: deterministic).
// Hand-off form (MI355X_MICROARCH.md, "Valid forms", R2): data-tagged 8-byte granules {value, step tag}, each written by ONE
// relaxed agent-scope atomic store (global_store_dwordx2 sc1) and polled with relaxed agent-scope atomic loads (sc1: served past
// the L1) -- no flags, no fences, no L2 write-back.  Two granule buffers alternate by step parity: a workgroup writes step s + 2
// only after it has read every other workgroup's step s + 1, which they publish only after reading all of step s -- so the
// buffer it overwrites has been read by everyone.  The exchange buffer is cleared by a memset node ahead of each launch (tags
// start at 1), so replays of a captured step see no stale tags.  Every poll is bounded: on expiry an error word is set and the
// launch finishes with wrong values instead of hanging.
// Residency: the grid never exceeds one workgroup per CU (clusters loop over their sample groups), so every workgroup of a
// cluster becomes resident without waiting for another workgroup of this launch to exit.
#include "gemm.h"
#include "smin_hip.h"
#include <stdlib.h>
#include <map>
#include <mutex>
#include <tuple>

namespace smin {

constexpr int CL_BS = 4;            // samples per cluster pass
constexpr int CL_U = 32;            // hidden units per workgroup
constexpr unsigned CL_SPIN = 1u << 22;

__device__ unsigned int g_lstm_cluster_error;

// -DSMIN_LSTM_STAMPS (tools/lstm_stamps.sh, a diagnostic copy of the library): s_memtime at the phase boundaries of every time step of
// workgroup 0, [kernel 0 = forward / 1 = backward][step < 64][8 stamps]
#ifdef SMIN_LSTM_STAMPS
__device__ unsigned long long g_lstm_stamps[2 * 64 * 8];
#define LSTM_STAMP(kern, step, k) do { if (blockIdx.x == 0 && threadIdx.x == 0 && (step) < 64) g_lstm_stamps[((kern) * 64 + (step)) * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define LSTM_STAMP(kern, step, k) do {} while (0)
#endif

__device__ __forceinline__ float csigm(float x) { return 1.0f / (1.0f + expf(-x)); }

__device__ __forceinline__ void granule_store(unsigned long long* p, float v, unsigned tag) {
    __hip_atomic_store(p, ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// waits for the granule to carry `tag`; returns its value (bounded: see the header)
__device__ __forceinline__ float granule_wait(const unsigned long long* p, unsigned tag) {
    unsigned long long g = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while ((unsigned)(g >> 32) != tag) {
        if (++spins > CL_SPIN) { g_lstm_cluster_error = 1u; break; }
        __builtin_amdgcn_s_sleep(1);
        g = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return __uint_as_float((unsigned)g);
}

// N granules at once: every load is in flight before the first tag is looked at (N dependent round trips otherwise: the owner of a
// unit waited for its P - 1 partial sums one after the other, ~1 us each)
template <int N>
__device__ __forceinline__ void granules_wait(float (&out)[N], const unsigned long long* const (&p)[N], const bool (&want)[N], unsigned tag) {
    unsigned long long g[N];
#pragma unroll
    for (int i = 0; i < N; ++i) g[i] = want[i] ? __hip_atomic_load(p[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ((unsigned long long)tag << 32);
    unsigned spins = 0;
    for (;;) {
        bool all = true;
#pragma unroll
        for (int i = 0; i < N; ++i) all = all && (unsigned)(g[i] >> 32) == tag;
        if (all) break;
        if (++spins > CL_SPIN) { g_lstm_cluster_error = 1u; break; }
        __builtin_amdgcn_s_sleep(1);
#pragma unroll
        for (int i = 0; i < N; ++i)
            if ((unsigned)(g[i] >> 32) != tag) g[i] = __hip_atomic_load(p[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma unroll
    for (int i = 0; i < N; ++i) out[i] = __uint_as_float((unsigned)g[i]);
}

// 16-byte granules {v0, tag, v1, tag}: two values per L2 transaction.  The backward exchange moves 1 024 values per workgroup and
// step in each direction and is bound by the number of transactions the L2 retires, not by their bytes (tools/lstm_stamps.sh: with
// 8-byte granules a step waited 6.9 k cycles for its partial sums, 41 % of the step).  The tag sits in both 8-byte halves, so a
// reader that sees both tags has both values whether or not the 16-byte store is performed as one piece.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void granule2_store(unsigned long long* p, float v0, float v1, unsigned tag) {
    const u32x4 g = {__float_as_uint(v0), tag, __float_as_uint(v1), tag};
    // s_nop: gfx9's store-data hazard -- a VMEM store of more than 8 bytes must not be followed at once by an instruction that rewrites
    // its data registers.  The compiler inserts that wait state for its own stores; it does not look into an asm statement, and the next
    // granule is built in the same registers (without it: a rare wrong VALUE under a right tag, tools/flaky_probe.py 8 of 239 runs).
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(g) : "memory");
}
// seven granules at once (P <= 8: the other workgroups of a cluster): all loads in flight before the first tag is looked at
__device__ __forceinline__ void granules2_load7(u32x4 (&g)[7], const unsigned long long* const (&p)[7]) {
    asm volatile("global_load_dwordx4 %0, %7, off sc1\n\t"
                 "global_load_dwordx4 %1, %8, off sc1\n\t"
                 "global_load_dwordx4 %2, %9, off sc1\n\t"
                 "global_load_dwordx4 %3, %10, off sc1\n\t"
                 "global_load_dwordx4 %4, %11, off sc1\n\t"
                 "global_load_dwordx4 %5, %12, off sc1\n\t"
                 "global_load_dwordx4 %6, %13, off sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(g[0]), "=&v"(g[1]), "=&v"(g[2]), "=&v"(g[3]), "=&v"(g[4]), "=&v"(g[5]), "=&v"(g[6])
                 : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6])
                 : "memory");
}

// workgroup id -> (cluster, member): the P members of a cluster are ids with the same id % 8, i.e. one XCD under the observed
// round-robin placement (speed only: the exchange is then served by one L2)
__device__ __forceinline__ void cluster_of(int id, int P, int& cluster, int& member) {
    const int xcd = id & 7, slot = id >> 3;
    cluster = (slot / P) * 8 + xcd;
    member = slot % P;
}

// G [B][Nq][2][4H] in: input projections + biases; out: gate activations.  W4 [2][H][H][4] (W4[d][k][u][g] = W_hh_d[gH + u][k]).
// xch: [nclus][2 parities][H][CL_BS] granules.  grid = nclus_pad * P workgroups of 256 threads; ngroups = ceil(B / CL_BS) * 2 (group, direction) passes
// are dealt round-robin over the nclus clusters.
typedef float f32x4m __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256)
void bilstm_cluster_fwd_kernel(float* __restrict__ G, const float* __restrict__ W4, const int* __restrict__ len, int B, int Nq, int H, int P, int nclus,
                               float* __restrict__ Hout, float* __restrict__ Cs, unsigned long long* __restrict__ xch)
{
    // The step's contraction z[u][gate][b] = sum_k W[k][u][gate] h[k][b] is 32 units x 4 gates x 4 samples of outputs over K = H: as
    // v_mfma_f32_4x4x1 -- sixteen independent 4 x 4 outer products per instruction, block = unit, A = the unit's four gate weights of
    // column k (four consecutive floats of the LDS image), B = h[k] of the four samples -- 2 instructions per k cover the workgroup's
    // 32 units; the four waves split K and meet in LDS.  (As scalar FMAs the phase was bound by the unpacked fp32 VALU rate: 512 FMAs
    // per lane and step, 3.6 k of a step's 7.7 k cycles; tools/lstm_stamps.sh.)  Lane (blk, j) of instruction m ends up with the four
    // gates (registers) of unit 16 m + blk for sample j: the finishing thread of (unit, sample) -- cell state in its registers.
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int BS = CL_BS, U = CL_U;
    float* Wl = lds;                                               // [H][U][4 gates]
    float* hs = lds + (size_t)H * U * 4;                           // [H][BS]
    float* part = hs + (size_t)H * BS;                             // [4 waves][2 instr][4 gates][64 lanes]
    int cluster, p;
    cluster_of(blockIdx.x, P, cluster, p);
    if (cluster >= nclus) return;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int u0 = p * U, H4 = 4 * H, kn = H / 4, k0 = wave * kn;
    // finishing threads: waves 0 and 1 (tid < 128): unit 16 * wave + lane / 4, sample lane % 4
    const bool fin = tid < 2 * 64;
    const int ul = 16 * (wave & 1) + (lane >> 2), bq = lane & 3, u = u0 + ul;
    const int ngroups = ((B + BS - 1) / BS) * 2;
    unsigned long long* X = xch + (size_t)cluster * 2 * H * BS;
    unsigned tagbase = 0;                                          // tags grow over the passes of a cluster: no clearing between passes

    for (int grp = cluster; grp < ngroups; grp += nclus) {
        const int d = grp & 1, b0 = (grp >> 1) * BS;
        __syncthreads();                                           // the previous pass is done with the LDS images
        for (int e = tid; e < H * U; e += 256) {                   // this workgroup's W_hh slice: W4[d][k][u0 .. u0+U)
            const int k = e / U, x = e % U;
            *reinterpret_cast<float4*>(Wl + (size_t)e * 4) = *reinterpret_cast<const float4*>(W4 + (((size_t)d * H + k) * H + u0 + x) * 4);
        }
        for (int e = tid; e < H * BS; e += 256) hs[e] = 0.f;
        const int bme = b0 + bq;                                   // the sample a finishing thread works for
        const int L = (fin && bme < B) ? min(len[bme], Nq) : 0;
        float c = 0.f;
        // the input projections of step s + 1 are requested while step s contracts (they do not depend on the recurrence): a step that
        // started with their L2 / HBM round trip spent ~3 k of its 7 k cycles waiting for it, whatever the contraction cost
        struct Gx { float v[4]; };
        auto fetch = [&](int s2) {
            Gx r = {{0.f, 0.f, 0.f, 0.f}};
            if (s2 < L) {
                const int pos2 = d == 0 ? s2 : L - 1 - s2;
                const float* g = G + ((((size_t)bme * Nq + pos2) * 2 + d) * H4) + u;
#pragma unroll
                for (int q = 0; q < 4; ++q) r.v[q] = g[q * H];
            }
            return r;
        };
        Gx gcur = fetch(0);
        __syncthreads();
        for (int s = 0; s < Nq; ++s) {
            LSTM_STAMP(0, s, 0);
            const bool act = s < L;
            const int pos = d == 0 ? s : L - 1 - s;
            const size_t row = (size_t)(fin && bme < B ? bme : 0) * Nq + (act ? pos : 0);
            const Gx gnext = fetch(s + 1);
            float gx[4] = {gcur.v[0], gcur.v[1], gcur.v[2], gcur.v[3]};
            // four accumulator chains per instruction slot: back-to-back MFMAs into ONE accumulator wait for each other (two chains ran
            // at the dependent-issue latency: the phase took as long as the scalar version)
            f32x4m accA[4], accB[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) { accA[q] = f32x4m{0.f, 0.f, 0.f, 0.f}; accB[q] = f32x4m{0.f, 0.f, 0.f, 0.f}; }
            {
                const float* wk = Wl + (size_t)k0 * U * 4 + lane;   // units 0..15 of column k: 64 consecutive floats; units 16..31: + 64
                const float* hk = hs + (size_t)k0 * BS + (lane & 3);
                for (int kb = 0; kb < kn; kb += 8) {                // kn = H / 4 is a multiple of 8 (H % 32 == 0): eight columns' operands in flight
                    float a0[8], a1[8], hb[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) { a0[q] = wk[(size_t)(kb + q) * U * 4]; a1[q] = wk[(size_t)(kb + q) * U * 4 + 64]; hb[q] = hk[(size_t)(kb + q) * BS]; }
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        accA[q & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(a0[q], hb[q], accA[q & 3], 0, 0, 0);
                        accB[q & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(a1[q], hb[q], accB[q & 3], 0, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                part[((wave * 2 + 0) * 4 + q) * 64 + lane] = (accA[0][q] + accA[1][q]) + (accA[2][q] + accA[3][q]);
                part[((wave * 2 + 1) * 4 + q) * 64 + lane] = (accB[0][q] + accB[1][q]) + (accB[2][q] + accB[3][q]);
            }
            LSTM_STAMP(0, s, 1);
            __syncthreads();                                        // partial sums visible; everyone is done reading hs
            LSTM_STAMP(0, s, 2);
            // the last step's h feeds nothing -- but unless this is the cluster's last pass the exchange still runs: it is what keeps a fast
            // workgroup from overwriting, in its next pass, granules a slow one has not read yet (see the header)
            const bool xchg = P > 1 && (s + 1 < Nq || grp + nclus < ngroups);
            const unsigned gs = tagbase + (unsigned)s, tag = gs + 1u;
            unsigned long long* Xs = X + (size_t)(gs & 1u) * H * BS;
            if (fin) {
                float hn = 0.f;
                if (act) {
                    float z[4];
                    const int m = wave & 1;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float sum = 0.f;
#pragma unroll
                        for (int w = 0; w < 4; ++w) sum += part[((w * 2 + m) * 4 + q) * 64 + lane];
                        z[q] = gx[q] + sum;
                    }
                    const float ig = csigm(z[0]), fg = csigm(z[1]), gg = tanhf(z[2]), og = csigm(z[3]);
                    c = fmaf(fg, c, ig * gg);
                    hn = og * tanhf(c);
                    float* g = G + (row * 2 + d) * H4 + u;
                    g[0] = ig; g[H] = fg; g[2 * H] = gg; g[3 * H] = og;
                    Cs[(row * 2 + d) * H + u] = c;
                    Hout[row * 2 * H + d * H + u] = hn;
                } else if (bme < B) {
                    Hout[((size_t)bme * Nq + s) * 2 * H + d * H + u] = 0.f;        // padded position s >= len
                }
                hs[u * BS + bq] = hn;                               // own units: straight into the local image
                if (xchg) granule_store(Xs + (size_t)u * BS + bq, hn, tag);
            }
            LSTM_STAMP(0, s, 3);
            if (xchg) {                                             // gather the other workgroups' units (H * BS <= 1024 granules: four per thread)
                float v[4];
                const unsigned long long* ptr[4];
                bool want[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int e = tid + 256 * i;
                    want[i] = e < H * BS && (e / BS) / U != p;
                    ptr[i] = Xs + (want[i] ? e : 0);
                }
                granules_wait<4>(v, ptr, want, tag);
#pragma unroll
                for (int i = 0; i < 4; ++i) if (want[i]) hs[tid + 256 * i] = v[i];
            }
            gcur = gnext;
            LSTM_STAMP(0, s, 4);
            __syncthreads();
            LSTM_STAMP(0, s, 5);
        }
        tagbase += (unsigned)Nq;
    }
}

// Backward recurrence.  Whh [2][4H][H] as nn.LSTM stores it; dG [B][Nq][2][4H] out (zero at padded positions).
// xch: [nclus][2 parities][P dest][P src][U][CL_BS] granules.  256 threads: gate phase thread (b, ul) = (tid / U, tid % U) for tid < 128;
// contraction phase thread u' = tid (+ 256 ..) over the workgroup's 128 gate rows.
__global__ __launch_bounds__(256)
void bilstm_cluster_bwd_kernel(const float* __restrict__ dHout, const float* __restrict__ G, const float* __restrict__ Cs, const float* __restrict__ Whh,
                               const int* __restrict__ len, int B, int Nq, int H, int P, int nclus, float* __restrict__ dG,
                               unsigned long long* __restrict__ xch)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int BS = CL_BS, U = CL_U;
    float* Wt = lds;                                               // [4U][H]: row g*U + ul = W_hh[g H + u0 + ul][:]
    float* dgs = lds + (size_t)4 * U * H;                          // [4U][BS]
    float* own = dgs + (size_t)4 * U * BS;                         // [U][BS] this workgroup's partial sums for its own units
    int cluster, p;
    cluster_of(blockIdx.x, P, cluster, p);
    if (cluster >= nclus) return;
    const int tid = threadIdx.x, ul = tid % U, bq = tid / U;
    const int u0 = p * U, u = u0 + ul, H4 = 4 * H;
    const int ngroups = ((B + BS - 1) / BS) * 2;
    unsigned long long* X = xch + (size_t)cluster * 2 * P * P * U * BS;
    unsigned tagbase = 0;

    for (int grp = cluster; grp < ngroups; grp += nclus) {
        const int d = grp & 1, b0 = (grp >> 1) * BS;
        __syncthreads();
        for (int e = tid; e < 4 * U * (H / 4); e += 256) {         // the slice, float4 along u'
            const int j = e / (H / 4), c4 = e % (H / 4), g = j / U, x = j % U;
            *reinterpret_cast<float4*>(Wt + (size_t)j * H + 4 * c4) =
                *reinterpret_cast<const float4*>(Whh + ((size_t)d * H4 + (size_t)g * H + u0 + x) * H + 4 * c4);
        }
        const bool fin = bq < BS;
        const int bme = b0 + bq;
        const int L = (fin && bme < B) ? min(len[bme], Nq) : 0;
        float dhn = 0.f, dcn = 0.f;
        // the gate phase's operands (saved gates, cell states, the output gradient) do not depend on the recurrence: those of step
        // s - 1 are requested while step s contracts, so no step starts with an L2 round trip
        struct GateIn { float ig, fg, gg, og, ct, cp, dho; };
        auto fetch = [&](int s2) {
            GateIn r = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (fin && s2 >= 0 && s2 < L) {
                const int pos = d == 0 ? s2 : L - 1 - s2;
                const size_t row = (size_t)bme * Nq + pos;
                const float* g = G + (row * 2 + d) * H4 + u;
                r.ig = g[0]; r.fg = g[H]; r.gg = g[2 * H]; r.og = g[3 * H];
                r.ct = Cs[(row * 2 + d) * H + u];
                r.cp = s2 > 0 ? Cs[(((size_t)bme * Nq + (d == 0 ? pos - 1 : pos + 1)) * 2 + d) * H + u] : 0.f;
                r.dho = dHout[row * 2 * H + d * H + u];
            }
            return r;
        };
        GateIn cur = fetch(Nq - 1);
        __syncthreads();
        int it = 0;
        for (int s = Nq - 1; s >= 0; --s, ++it) {
            LSTM_STAMP(1, it, 0);
            const bool act = s < L;
            const GateIn nxt = fetch(s - 1);
            if (fin) {
                float dg[4] = {0.f, 0.f, 0.f, 0.f};
                if (act) {
                    const int pos = d == 0 ? s : L - 1 - s;
                    const size_t row = (size_t)bme * Nq + pos;
                    const float ig = cur.ig, fg = cur.fg, gg = cur.gg, og = cur.og;
                    const float ct = cur.ct;
                    const float cp = cur.cp;
                    const float dh = cur.dho + dhn;
                    const float tc = tanhf(ct);
                    const float dc = fmaf(dh * og, 1.0f - tc * tc, dcn);
                    dg[0] = dc * gg * ig * (1.0f - ig);
                    dg[1] = dc * cp * fg * (1.0f - fg);
                    dg[2] = dc * ig * (1.0f - gg * gg);
                    dg[3] = dh * tc * og * (1.0f - og);
                    dcn = dc * fg;
                    float* o = dG + (row * 2 + d) * H4 + u;
                    o[0] = dg[0]; o[H] = dg[1]; o[2 * H] = dg[2]; o[3 * H] = dg[3];
                } else if (bme < B) {
                    float* o = dG + ((((size_t)bme * Nq + s) * 2 + d) * H4) + u;   // padded position s >= len
                    o[0] = 0.f; o[H] = 0.f; o[2 * H] = 0.f; o[3 * H] = 0.f;
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) dgs[(size_t)(q * U + ul) * BS + bq] = dg[q];
            }
            LSTM_STAMP(1, it, 1);
            __syncthreads();
            LSTM_STAMP(1, it, 2);
            if (s == 0 && !(P > 1 && grp + nclus < ngroups)) break;   // dh of the step before the first feeds nothing (exchange kept between passes, as in the forward)
            const unsigned gs = tagbase + (unsigned)it, tag = gs + 1u;
            unsigned long long* Xs = X + (size_t)(gs & 1u) * P * P * U * BS;
            for (int up = tid; up < H; up += 256) {                 // partial dh[b][u'] over this workgroup's 4U gate rows
                float a[BS] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
                for (int j = 0; j < 4 * U; ++j) {
                    const float w = Wt[(size_t)j * H + up];
                    const float4 g4 = *reinterpret_cast<const float4*>(dgs + (size_t)j * BS);
                    a[0] = fmaf(g4.x, w, a[0]); a[1] = fmaf(g4.y, w, a[1]); a[2] = fmaf(g4.z, w, a[2]); a[3] = fmaf(g4.w, w, a[3]);
                }
                const int dest = up / U, x = up % U;
                if (dest == p) {
#pragma unroll
                    for (int b = 0; b < BS; ++b) own[x * BS + b] = a[b];
                } else {
                    unsigned long long* o = Xs + (((size_t)dest * P + p) * U + x) * BS;       // BS = 4 values = two 16-byte granules
                    granule2_store(o, a[0], a[1], tag);
                    granule2_store(o + 2, a[2], a[3], tag);
                }
            }
            LSTM_STAMP(1, it, 3);
            __syncthreads();                                        // own[] visible; dgs free for the next step
            LSTM_STAMP(1, it, 4);
            if (fin) {                                              // the owner adds the P partial sums in workgroup order (P <= 8)
                // threads of the even samples load the 16-byte granules (their sample and the next one's); the odd samples' threads sit
                // 32 lanes further in the same wave and take their value from there
                const bool loader = (bq & 1) == 0;
                u32x4 g[7];
                const unsigned long long* ptr[7];
                int nsrc = 0;
#pragma unroll
                for (int k = 0; k < 7; ++k) {
                    const int src = k < p ? k : k + 1;               // the k-th other workgroup
                    const bool want = src < P;
                    ptr[k] = Xs + (((size_t)p * P + (want ? src : (p == 0 ? 1 : 0))) * U + ul) * BS + (bq & 2);
                    nsrc += want ? 1 : 0;
                }
                if (loader) {
                    unsigned spins = 0;
                    for (;;) {
                        granules2_load7(g, ptr);
                        bool all = true;
#pragma unroll
                        for (int k = 0; k < 7; ++k) all = all && (k >= nsrc || (g[k][1] == tag && g[k][3] == tag));
                        if (all) break;
                        if (++spins > CL_SPIN) { g_lstm_cluster_error = 1u; break; }
                        __builtin_amdgcn_s_sleep(1);
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < 7; ++k) g[k] = u32x4{0u, 0u, 0u, 0u};
                }
                float sum = 0.f;
#pragma unroll
                for (int src = 0; src < 8; ++src) {
                    if (src >= P) continue;
                    float v;
                    if (src == p) v = own[ul * BS + bq];
                    else {
                        const int k = src < p ? src : src - 1;
                        const float odd = __shfl(__uint_as_float(g[k][2]), (threadIdx.x & 63) ^ 32);     // the loader's second value
                        v = loader ? __uint_as_float(g[k][0]) : odd;
                    }
                    sum += v;
                }
                dhn = sum;
            }
            LSTM_STAMP(1, it, 5);
            cur = nxt;
            // (own[] is rewritten only after the next step's first barrier)
        }
        tagbase += (unsigned)Nq;
    }
}

// ---- launchers ----------------------------------------------------------------------------------------------------------------
static int cl_num_cus()
{
    static int n = 0;
    if (n == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n = v;
        else n = 256;
    }
    return n;
}
// Exchange buffers, one per (device, direction), allocated ONCE at the size the largest geometry needs on this device (the cluster
// count is capped by the CU count, so the bound does not depend on the batch): forward <= 256 CUs + 64 H granules, backward
// <= 2048 CUs + 2048 P^2.  Never reallocated, so a captured graph's launches keep a valid address.  The step runs its recurrences
// on one stream.  Two cluster recurrences in flight at once on one device (two streams) would share granules AND wait for each
// other's CUs (measured: 4.8 s per step until the bounded polls expire): cl_turn below orders launches from different streams
// behind each other.  (Not inside a stream capture -- a captured step has one recurrence stream; two graphs that both hold cluster
// recurrences must not be replayed side by side.)
static unsigned long long* cl_exchange(int which, size_t granules)
{
    static std::mutex mu;
    static std::map<std::pair<int, int>, std::pair<unsigned long long*, size_t>> bufs;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lk(mu);
    auto& b = bufs[std::make_pair(dev, which)];
    if (!b.first) {
        const size_t ncu = (size_t)cl_num_cus();
        const size_t cap = which == 0 ? 256 * ncu + 64 * 256 : 2048 * ncu + 2048 * 64;
        if (hipMalloc(reinterpret_cast<void**>(&b.first), cap * sizeof(unsigned long long)) != hipSuccess) { b.first = nullptr; return nullptr; }
        b.second = cap;
    }
    return granules <= b.second ? b.first : nullptr;
}

// One cluster recurrence at a time per device: a launch waits for the previous launch's end (an event per device; a no-op when both
// are on one stream).  `end` = false before the launch (wait), true after it (record).
static int cl_turn(hipStream_t st, bool end)
{
    static std::mutex mu;
    static std::map<int, std::pair<hipEvent_t, bool>> last;          // event, recorded at least once
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess) return -4;
    if (cs != hipStreamCaptureStatusNone) return 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -4;
    std::lock_guard<std::mutex> lk(mu);
    auto& e = last[dev];
    if (!e.first && hipEventCreateWithFlags(&e.first, hipEventDisableTiming) != hipSuccess) { e.first = nullptr; return -4; }
    if (end) { e.second = true; return hipEventRecord(e.first, st) == hipSuccess ? 0 : -4; }
    return (!e.second || hipStreamWaitEvent(st, e.first, 0) == hipSuccess) ? 0 : -4;
}

bool bilstm_cluster_ok(int B, int Nq, int H)
{
    (void)B;
    return H % CL_U == 0 && H >= CL_U && H <= 256 && Nq < (1 << 20) && !getenv("SMIN_LSTM_STREAMED");
}
static void cl_geometry(int B, int H, int& P, int& nclus, int& grid)
{
    P = H / CL_U;
    const int ngroups = cdiv(B, CL_BS) * 2;
    int maxclus = cl_num_cus() / P;                                  // never more than one workgroup per CU: see the header
    if (maxclus < 1) maxclus = 1;
    nclus = ngroups < maxclus ? ngroups : maxclus;
    grid = cdiv(nclus, 8) * 8 * P;                                   // cluster_of deals ids over 8 XCD classes
}

int launch_bilstm_cluster_fwd(hipStream_t st, float* G, const float* W4, const int* len, int B, int Nq, int H, float* Hout, float* Cs)
{
    int P, nclus, grid;
    cl_geometry(B, H, P, nclus, grid);
    const size_t lds = sizeof(float) * ((size_t)H * CL_U * 4 + (size_t)H * CL_BS + (size_t)4 * 2 * 4 * 64);
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(bilstm_cluster_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return -1;
        attr = true;
    }
    const size_t gran = (size_t)cdiv(nclus, 8) * 8 * 2 * H * CL_BS;
    unsigned long long* xch = cl_exchange(0, gran);
    if (!xch) return -2;
    if (cl_turn(st, false)) return -4;
    if (P > 1 && hipMemsetAsync(xch, 0, gran * sizeof(unsigned long long), st) != hipSuccess) return -3;
    hipLaunchKernelGGL(bilstm_cluster_fwd_kernel, dim3(grid), dim3(256), lds, st, G, W4, len, B, Nq, H, P, nclus, Hout, Cs, xch);
    SMIN_LAUNCH_CHECK();
    return cl_turn(st, true);
}

int launch_bilstm_cluster_bwd(hipStream_t st, const float* dHout, const float* G, const float* Cs, const float* Whh, const int* len, int B, int Nq, int H,
                              float* dG)
{
    int P, nclus, grid;
    cl_geometry(B, H, P, nclus, grid);
    const size_t lds = sizeof(float) * ((size_t)4 * CL_U * H + (size_t)4 * CL_U * CL_BS + (size_t)CL_U * CL_BS);
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(bilstm_cluster_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return -1;
        attr = true;
    }
    const size_t gran = (size_t)cdiv(nclus, 8) * 8 * 2 * P * P * CL_U * CL_BS;
    unsigned long long* xch = cl_exchange(1, gran);
    if (!xch) return -2;
    if (cl_turn(st, false)) return -4;
    if (P > 1 && hipMemsetAsync(xch, 0, gran * sizeof(unsigned long long), st) != hipSuccess) return -3;
    hipLaunchKernelGGL(bilstm_cluster_bwd_kernel, dim3(grid), dim3(256), lds, st, dHout, G, Cs, Whh, len, B, Nq, H, P, nclus, dG, xch);
    SMIN_LAUNCH_CHECK();
    return cl_turn(st, true);
}

}  // namespace smin

#ifdef SMIN_LSTM_STAMPS
extern "C" int smin_debug_lstm_stamps(unsigned long long* host_out, int n)
{
    if (n > 2 * 64 * 8) n = 2 * 64 * 8;
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(smin::g_lstm_stamps), sizeof(unsigned long long) * n);
}
#endif

// non-zero once a bounded poll of the cluster recurrence has expired (the results of that launch are wrong); clears the word
extern "C" int smin_lstm_cluster_error(void)
{
    unsigned int v = 0, z = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(smin::g_lstm_cluster_error), sizeof(v)) != hipSuccess) return -1;
    if (v) (void)hipMemcpyToSymbol(HIP_SYMBOL(smin::g_lstm_cluster_error), &z, sizeof(z));
    return (int)v;
}
