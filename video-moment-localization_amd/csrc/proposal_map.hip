// 2D temporal-adjacency proposal map: ProposalGeneration.forward (reference models.py:115-126) with
// compute_content_matrix (models.py:88-98) restated as index arithmetic -- the dense (L,L,C,T) averaging
// matrix is never built.  HBM-bound: the map is written once (C*D + D floats per cell).
//
// Forward: an fp64 running sum of f over time per sample (B*(T+1)*D doubles, L2/MALL resident) turns every clip
// mean into two row reads, independent of the window length (the reference spends 2*L^2*C*T*D FLOPs here).
// Backward: the transpose, as a difference array -- every clip adds +g/cs at its first frame and -g/cs one past
// its last frame; a running sum over time gives df.  Events are *gathered* per (b, t) by solving the clip
// equations, so the result is deterministic (no float atomics).
#include "common.h"
#include "smin_hip.h"

namespace smin {

struct double4_ { double x, y, z, w; };

// Blocked running sum over time for both directions of the proposal map (forward: the fp64 prefix Pf of f; backward:
// df = running sum of the event array E).  A serial form keeps one thread per (b, d) busy for T dependent
// steps (130 us at T = 256 with half the CUs idle); here a 1024-thread workgroup owns 64 feature columns of a sample
// and cuts time into 16 segments: per-segment sums, a 16-entry exclusive scan through LDS, then the segment's running
// sums (the second read hits L2).  fp64 accumulation throughout; the order of additions is fixed.
template <bool PREFIX>
__global__ __launch_bounds__(1024)
void time_scan_kernel(const float* __restrict__ in, const float* __restrict__ dfb, int T, int L, int D,
                      double* __restrict__ Pf, float* __restrict__ df)
{
    __shared__ double part[16][64];
    const int col = threadIdx.x & 63, seg = threadIdx.x >> 6, b = blockIdx.y;
    const int d = blockIdx.x * 64 + col;
    const bool ok = d < D;
    const int TL = (T + 15) / 16, ta = min(T, seg * TL), tb = min(T, ta + TL);
    const float* src = in + (size_t)b * T * D + (ok ? d : 0);
    // A segment of at most 16 frames (T <= 256) is held in registers: its loads are all in flight at once and nothing is read twice.
    // (The loops below waited for one load per frame -- the stores to df / Pf may alias src for all the compiler knows -- and a step's
    // closing scans ran 350-500 us beside the weight contractions for 60 us alone.)
    const bool small = TL <= 16;
    float v[16], e[16];
    double local = 0.0;
    if (small) {
#pragma unroll
        for (int i = 0; i < 16; ++i) { const int t = ta + i; v[i] = t < tb ? src[(size_t)t * D] : 0.f; }
        if (!PREFIX) {
            const int r = T / L;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int t = ta + i;
                e[i] = (dfb && ok && t < tb && t / r < L) ? dfb[((size_t)b * L + t / r) * D + d] : 0.f;
            }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) local += (double)v[i];
    } else {
        for (int t = ta; t < tb; ++t) local += (double)src[(size_t)t * D];
    }
    part[seg][col] = local;
    __syncthreads();
    double run = 0.0;
    for (int k = 0; k < seg; ++k) run += part[k][col];
    if (!ok) return;
    if (PREFIX) {
        double* dst = Pf + (size_t)b * (T + 1) * D + d;
        if (small) {
#pragma unroll
            for (int i = 0; i < 16; ++i) { const int t = ta + i; if (t < tb) { dst[(size_t)t * D] = run; run += (double)v[i]; } }
        } else {
            for (int t = ta; t < tb; ++t) { dst[(size_t)t * D] = run; run += (double)src[(size_t)t * D]; }
        }
        if (tb == T && ta < T) dst[(size_t)T * D] = run;
        if (T == 0 && seg == 0) dst[0] = 0.0;
    } else {
        const int r = T / L;
        const float fr = (float)r;
        if (small) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int t = ta + i;
                if (t < tb) {
                    run += (double)v[i];
                    float o = (float)run;
                    if (dfb && t / r < L) o += e[i] / fr;
                    df[((size_t)b * T + t) * D + d] = o;
                }
            }
        } else {
            for (int t = ta; t < tb; ++t) {
                run += (double)src[(size_t)t * D];
                float o = (float)run;
                if (dfb && t / r < L) o += dfb[((size_t)b * L + t / r) * D + d] / fr;
                df[((size_t)b * T + t) * D + d] = o;
            }
        }
    }
}

// one 128-thread workgroup per cell
__global__ __launch_bounds__(128)
void proposal_map_fwd_kernel(const double* __restrict__ Pf, const int* __restrict__ cells, int T, int L, int C, int D,
                             float* __restrict__ fc, float* __restrict__ fm)
{
    const int n = blockIdx.x;
    const Cell cl = load_cell(cells, n);
    const int r = T / L;
    const int w = cl.j - cl.i + 1;
    const int nf = w * r;
    const int cs = max(1, nf / C);
    const int nclip = (cl.m != 0 && w >= 1) ? min(C, nf) : 0;
    const float inv32 = 1.0f / (float)cs;                         // the reference stores 1/clip_size in fp32
    const double inv = (double)inv32;
    const double* P = Pf + (size_t)cl.b * (T + 1) * D;
    if (!fc) {
        // f_m alone: the cell's clips are consecutive windows of cs frames starting at frame i*r, so the sum of their means telescopes to
        // ONE difference of the prefix -- two row reads per cell instead of C + 1 (the kernel is bound by the L2 reads of the fp64 rows).
        // One rounding instead of C + 1: within an ulp or two of the clip-by-clip sum the reference forms (models.py:117-119).
        const double scale = inv / (double)C;
        for (int d = threadIdx.x * 4; d < D; d += 512) {
            float4 v = f4zero();
            if (nclip > 0) {
                const double* pb = P + (size_t)(cl.i * r) * D + d;
                const double* pe = pb + (size_t)(nclip * cs) * D;
                v = make_float4((float)((pe[0] - pb[0]) * scale), (float)((pe[1] - pb[1]) * scale), (float)((pe[2] - pb[2]) * scale), (float)((pe[3] - pb[3]) * scale));
            }
            stg4(fm + (size_t)n * D + d, v);
        }
        return;
    }
    for (int d = threadIdx.x * 4; d < D; d += 512) {
        float4 sum = f4zero();
        // consecutive clips share a boundary row of the prefix: nclip + 1 row reads instead of 2 * nclip
        const double* pb = P + (size_t)(cl.i * r) * D + d;
        double4_ lo = {pb[0], pb[1], pb[2], pb[3]};
        for (int c = 0; c < C; ++c) {
            float4 v = f4zero();
            if (c < nclip) {
                const double* pe = pb + (size_t)((c + 1) * cs) * D;
                const double4_ hi = {pe[0], pe[1], pe[2], pe[3]};
                v.x = (float)((hi.x - lo.x) * inv); v.y = (float)((hi.y - lo.y) * inv);
                v.z = (float)((hi.z - lo.z) * inv); v.w = (float)((hi.w - lo.w) * inv);
                lo = hi;
            }
            stg4(fc + ((size_t)n * C + c) * D + d, v);
            sum = f4add(sum, v);
        }
        if (fm) stg4(fm + (size_t)n * D + d, make_float4(sum.x / C, sum.y / C, sum.z / C, sum.w / C));
    }
}

// fb[b][l][:] = mean_{t < r} f[b][l*r + t][:]        (AvgPool1d(r, r), models.py:121-125)
__global__ void boundary_pool_fwd_kernel(const float* __restrict__ f, int T, int L, int D4, float* __restrict__ fb, size_t total)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int d4 = (int)(idx % D4);
    const size_t bl = idx / D4;
    const int l = (int)(bl % L);
    const size_t b = bl / L;
    const int r = T / L;
    float4 s = f4zero();
    for (int t = 0; t < r; ++t) s = f4add(s, ldg4(f + (((size_t)b * T + l * r + t) * D4 + d4) * 4));
    const float fr = (float)r;
    stg4(fb + idx * 4, make_float4(s.x / fr, s.y / fr, s.z / fr, s.w / fr));
}

// where the clip gradient rows come from: one [N*C][D] tensor, or nseg tensors [N*C][W] side by side (D = nseg*W)
struct DenseClipGrad {
    const float* dfc;
    __device__ __forceinline__ bool any() const { return dfc != nullptr; }
    __device__ __forceinline__ int width(int D) const { return D; }
    __device__ __forceinline__ float4 load(size_t row, int d, int D) const { return ldg4(dfc + row * D + d); }
};
struct SegClipGrad {
    const float* seg[8]; int W;
    __device__ __forceinline__ bool any() const { return true; }
    __device__ __forceinline__ int width(int) const { return W; }
    __device__ __forceinline__ float4 load(size_t row, int d, int) const {
        const int sg = d / W;                                       // a select chain, not an indexed read of the argument block
        const float* p = seg[0];
#pragma unroll
        for (int k = 1; k < 8; ++k) p = sg == k ? seg[k] : p;
        return ldg4(p + row * W + (d - sg * W));
    }
};

// E[b][t][:] = sum over clips starting at t of g/cs  -  sum over clips ending at t of g/cs,
// g = m * (dfc[n][c][:] + dfm[n][:] / C).  One workgroup per (b, t); deterministic (pure gather).
template <class Src>
__global__ __launch_bounds__(128)
void proposal_map_bwd_events_kernel(Src src, const float* __restrict__ dfm,
                                    const int* __restrict__ cells, const int* __restrict__ cellmap,
                                    int T, int L, int C, int D, float* __restrict__ E)
{
    const int t = blockIdx.x, b = blockIdx.y;
    const int r = T / L;
    const int* cmap = cellmap + (size_t)b * L * L;
    const float invC = 1.0f / C;
    float4 acc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] = f4zero();

    // with_m: the mean-path gradient dfm/C is the same for all clips of a cell, and the clips tile the window back to
    // back, so it survives only at the first clip's start and the last clip's end (interior start/end pairs cancel).
    auto add_event = [&](int i, int w, int c, int cs, float sign, bool with_m) {
        const int j = i + w - 1;
        if (j >= L) return;
        const int n = cmap[i * L + j];
        if (n < 0) return;
        if (cells[4 * (size_t)n + 3] == 0) return;
        const float sc = sign * (1.0f / (float)cs);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int d = threadIdx.x * 4 + 512 * k;
            if (d < D) {
                float4 g = f4zero();
                if (src.any()) g = src.load((size_t)n * C + c, d, D);
                if (dfm && with_m) g = f4fma(ldg4(dfm + (size_t)n * D + d), invC, g);
                acc[k] = f4fma(g, sc, acc[k]);
            }
        }
    };
    // all window widths w whose clip size is cs and that own clip c:  max(1, w*r/C) == cs  and  c < min(C, w*r)
    auto for_widths = [&](int i, int c, int cs, float sign) {
        int lo = (cs * C + r - 1) / r;
        int hi = ((cs + 1) * C + r - 1) / r - 1;
        if (cs == 1) lo = 1;
        lo = max(lo, c / r + 1);
        hi = min(hi, L - i);
        for (int w = lo; w <= hi; ++w) add_event(i, w, c, cs, sign, sign < 0.f && c + 1 == min(C, w * r));
    };

    const int imax = min(L - 1, t / r);
    for (int i = 0; i <= imax; ++i) {
        const int base = t - i * r;                         // offset of frame t inside the window of row i
        if (base == 0) {                                    // clip 0 of every width starts here
            for (int w = 1; w <= L - i; ++w) add_event(i, w, 0, max(1, (w * r) / C), 1.0f, true);
        } else {
            for (int c = 1; c < C; ++c)                     // clip c >= 1 starts at t:  c * cs == base
                if (base % c == 0) for_widths(i, c, base / c, 1.0f);
            for (int cc = 1; cc <= C; ++cc)                 // clip cc-1 ends at t:  cc * cs == base
                if (base % cc == 0) for_widths(i, cc - 1, base / cc, -1.0f);
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int d = threadIdx.x * 4 + 512 * k;
        if (d < D) stg4(E + ((size_t)b * T + t) * D + d, acc[k]);
    }
}

// Same result as proposal_map_bwd_events_kernel, organised for the machine.
//
// Which clips of which cells (i, j) start or end at frame t depends only on (T, L, C) -- it is the sparsity pattern of
// the reference's content matrix Wc (models.py:88-98, a module buffer built once).  clip_event_table_kernel lists it
// once per geometry: a cell (i, w = j-i+1) has a clip boundary at t exactly when base = t - i*r is a multiple q of its
// clip size cs: clip q starts there if q < nclip, clip q-1 ends there if 1 <= q <= nclip.  The table is tiny
// (2 * sum nclip entries of 8 bytes) and shared by every sample and every call.
// Per call, one workgroup per (b, t) reads its frame's entries, resolves (i, j) through the sample's cellmap (and the
// mask) and compacts the hits into per-wave LDS lists by ballot rank -- a fixed order, so the sum is bitwise
// reproducible.  Streaming phase: a gradient row is `lanes` float4 wide per slot (slot = one 512-float chunk of a
// dense row, or one segment); the 128 threads form 128 / lanes groups that take alternate batches of UNR events, each
// with UNR x NS independent row loads in flight -- the phase is bound by HBM latency per step, not by bytes.  The
// groups' partial sums are combined through LDS in group order.
constexpr int EV_CAP = 512;                            // events per wave list per round
struct Ev { int nc; float sc; };                       // table: nc = (i << 18) | (j << 6) | (with_mean << 3) | clip
                                                       // resolved: nc = (cell << 4) | (with_mean << 3) | clip

// grid T, one wave per frame.  table == nullptr: counts[t] = number of entries; else fill table[offsets[t] ...].
__global__ __launch_bounds__(64)
void clip_event_table_kernel(int T, int L, int C, int* __restrict__ counts, const int* __restrict__ offsets, Ev* __restrict__ table)
{
    const int t = blockIdx.x, lane = threadIdx.x;
    const int r = T / L;
    const int imax = min(L - 1, t / r);
    const int npairs = (imax + 1) * L;
    Ev* out = table ? table + offsets[t] : nullptr;
    int wn = 0;
    for (int p0 = 0; p0 < npairs; p0 += 64) {
        const int p = p0 + lane;
        bool s_ev = false, e_ev = false;
        int i = 0, j = 0, q = 0, nclip = 0;
        float inv = 0.f;
        if (p < npairs) {
            i = p / L;
            const int w = p - i * L + 1;
            j = i + w - 1;
            if (j < L) {
                const int nf = w * r, cs = max(1, nf / C), base = t - i * r;
                nclip = min(C, nf);
                q = base / cs;
                if (q * cs == base) { s_ev = q < nclip; e_ev = q >= 1 && q <= nclip; }
                inv = 1.0f / (float)cs;
            }
        }
        const int key = (i << 18) | (j << 6);
        unsigned long long m = __ballot(s_ev);
        if (s_ev && out) out[wn + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0))] =
                             Ev{key | ((q == 0) << 3) | q, inv};
        wn += __popcll(m);
        m = __ballot(e_ev);
        if (e_ev && out) out[wn + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0))] =
                             Ev{key | ((q == nclip) << 3) | (q - 1), -inv};
        wn += __popcll(m);
    }
    if (!table && lane == 0) counts[t] = wn;
}

template <class Src, int NS, int UNR, bool HAS_M>
__global__ __launch_bounds__(128)
void proposal_map_bwd_events2_kernel(Src src, const float* __restrict__ dfm,
                                     const int* __restrict__ cells, const int* __restrict__ cellmap,
                                     const int* __restrict__ ev_off, const Ev* __restrict__ ev_tab,
                                     int B, int T, int L, int C, int D, float* __restrict__ E, int check_mask)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    Ev* list = reinterpret_cast<Ev*>(lds_raw);                      // [2][EV_CAP]
    int* wcnt = reinterpret_cast<int*>(list + 2 * EV_CAP);          // [2] events in each wave's list
    float4* red = reinterpret_cast<float4*>(lds_raw + 2 * EV_CAP * sizeof(Ev) + 16);   // [128][NS]
    // All frames of a sample on ONE XCD (workgroups id and id + 8 share an XCD under the observed round-robin placement: speed
    // only): a clip's gradient row is read at its start frame and again at its end frame, by two workgroups of the same sample
    // that are in flight together -- with (t, b) dealt over all eight L2s the second read went to the fabric (1.24 GB fetched for
    // 0.62 GB of rows).
    // (sample 8g + k goes to XCD class (k + g) mod 8, not k: lengths often alternate with the sample index -- the synthetic batch's
    //  even samples are full length, its odd ones ragged -- and a fixed class per residue left half of the XCDs with a quarter of the work)
    const int id = blockIdx.x, slot = id >> 3, sg = slot / T;
    const int t = slot % T, b = sg * 8 + (((id & 7) - sg) & 7), tid = threadIdx.x;
    if (b >= B) return;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int* cmap = cellmap + (size_t)b * L * L;
    const float invC = 1.0f / C;
    const int e0 = ev_off[t], ne = ev_off[t + 1] - e0;

    const int W = src.width(D);
    const int lanes = min(128, W / 4), groups = 128 / lanes;
    const int grp = tid / lanes, ln = tid - grp * lanes;
    const int nslot = (D + 4 * lanes - 1) / (4 * lanes);
    const bool active = grp < groups;
    float4 acc[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) acc[k] = f4zero();

    for (int base = 0; base < ne; base += 2 * EV_CAP) {
        // resolve this round's table entries: (i, j) -> packed cell index, dropping absent and masked cells
        int wn = 0;
        Ev* mine = list + wave * EV_CAP;
        // the two waves take alternate 64-entry chunks of the round (a frame has ~65 entries: with one contiguous half per wave the
        // second wave idled through both dependent-load trips of the first)
        const int hi = min(ne, base + 2 * EV_CAP);
        for (int x0 = base + 64 * wave; x0 < hi; x0 += 128) {
            const int x = x0 + lane;
            Ev e = Ev{0, 0.f};
            int n = -1;
            if (x < hi) {
                e = ev_tab[e0 + x];
                n = cmap[(e.nc >> 18) * L + ((e.nc >> 6) & 4095)];
                if (n >= 0 && check_mask && cells[4 * (size_t)n + 3] == 0) n = -1;
            }
            const unsigned long long m = __ballot(n >= 0);
            if (n >= 0) mine[wn + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0))] =
                            Ev{(n << 4) | (e.nc & 15), e.sc};
            wn += __popcll(m);
        }
        if (lane == 0) wcnt[wave] = wn;
        __syncthreads();
        const int n0 = wcnt[0], n1 = wcnt[1];
        auto event = [&](int e) -> Ev { return e < n0 ? list[e] : list[EV_CAP + (e - n0)]; };
        if (active)
            for (int q0 = grp * UNR; q0 < n0 + n1; q0 += groups * UNR) {
                const int cntk = min(UNR, n0 + n1 - q0);
                float4 gc[UNR][NS], gm[UNR][NS];
                float sc[UNR];
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    const Ev e = event(q0 + min(u, cntk - 1));
                    const bool ok = u < cntk && e.nc >= 0;
                    const int n = ok ? e.nc >> 4 : 0, c = e.nc & 7;
                    sc[u] = ok ? e.sc : 0.f;
#pragma unroll
                    for (int k = 0; k < NS; ++k) {
                        const int d = (k * lanes + ln) * 4;
                        const bool dok = ok && k < nslot && d < D;
                        gc[u][k] = (src.any() && dok) ? src.load((size_t)n * C + c, d, D) : f4zero();
                        if (HAS_M) gm[u][k] = (dok && (e.nc & 8)) ? ldg4(dfm + (size_t)n * D + d) : f4zero();
                    }
                }
#pragma unroll
                for (int u = 0; u < UNR; ++u)
#pragma unroll
                    for (int k = 0; k < NS; ++k)
                        acc[k] = f4fma(HAS_M ? f4fma(gm[u][k], invC, gc[u][k]) : gc[u][k], sc[u], acc[k]);
            }
        __syncthreads();                                            // the lists are rebuilt in the next round
    }
    if (groups > 1) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NS; ++k) red[tid * NS + k] = acc[k];
        __syncthreads();
        if (grp == 0)
            for (int g = 1; g < groups; ++g)
#pragma unroll
                for (int k = 0; k < NS; ++k) acc[k] = f4add(acc[k], red[(g * lanes + ln) * NS + k]);
    }
    if (grp == 0) {
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const int d = (k * lanes + ln) * 4;
            if (k < nslot && d < D) stg4(E + ((size_t)b * T + t) * D + d, acc[k]);
        }
    }
}

// df[b][t][d] = running sum of E over t (fp64 accumulator) + dfb[b][t / r][d] / r
__global__ void proposal_map_bwd_scan_kernel(const float* __restrict__ E, const float* __restrict__ dfb, int B, int T, int L, int D,
                                             float* __restrict__ df)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * D) return;
    const int b = idx / D, d = idx % D;
    const int r = T / L;
    const float fr = (float)r;
    double run = 0.0;
    for (int t = 0; t < T; ++t) {
        const size_t o = ((size_t)b * T + t) * D + d;
        if (E) run += (double)E[o];
        float v = (float)run;
        if (dfb && t / r < L) v += dfb[((size_t)b * L + t / r) * D + d] / fr;
        df[o] = v;
    }
}

}  // namespace smin

using namespace smin;

extern "C" int smin_proposal_map_fwd(void* stream, const float* f, const int32_t* cells, int N, int B, int T, int L, int C, int D,
                                        float* fc, float* fm, float* fb, void* ws, size_t ws_bytes)
{
    hipStream_t st = (hipStream_t)stream;
    SMIN_REQUIRE(D % 4 == 0 && L >= 1 && T >= L && T % L == 0 && C >= 1);
    SMIN_REQUIRE(D <= 2048);
    SMIN_REQUIRE(ws_bytes >= sizeof(double) * (size_t)B * (T + 1) * D);
    double* Pf = reinterpret_cast<double*>(ws);
    hipLaunchKernelGGL((time_scan_kernel<true>), dim3(cdiv(D, 64), B), dim3(1024), 0, st, f, (const float*)nullptr, T, L, D, Pf, (float*)nullptr);
    SMIN_LAUNCH_CHECK();
    if (N > 0 && (fc || fm)) {
        hipLaunchKernelGGL(proposal_map_fwd_kernel, dim3(N), dim3(128), 0, st, Pf, cells, T, L, C, D, fc, fm);
        SMIN_LAUNCH_CHECK();
    }
    if (fb) {
        const size_t tot = (size_t)B * L * (D / 4);
        hipLaunchKernelGGL(boundary_pool_fwd_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, f, T, L, D / 4, fb, tot);
        SMIN_LAUNCH_CHECK();
    }
    return 0;
}

static bool events2_ok(int N, int T, int L, int C, size_t* lds)
{
    (void)T;
    *lds = 2 * EV_CAP * sizeof(Ev) + 16;                           // + the group-reduction buffer, see launch_events2
    return C <= 8 && L <= 4096 && N < (1 << 27);
}

template <class Src, int NS, int UNR, bool HAS_M>
static int launch_events2_t(hipStream_t st, const Src& src, const float* dfm, const int32_t* cells, const int32_t* cellmap,
                            const int32_t* ev_off, const Ev* ev_tab, int B, int T, int L, int C, int D, float* E, size_t lds)
{
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(proposal_map_bwd_events2_kernel<Src, NS, UNR, HAS_M>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL((proposal_map_bwd_events2_kernel<Src, NS, UNR, HAS_M>), dim3(T * 8 * cdiv(B, 8)), dim3(128), lds, st, src, dfm, cells, cellmap, ev_off, ev_tab,
                       B, T, L, C, D, E, 1);
    SMIN_LAUNCH_CHECK();
    return 0;
}

// W = width of one gradient row segment (D for a dense source)
template <class Src>
static int launch_events2(hipStream_t st, const Src& src, const float* dfm, const int32_t* cells, const int32_t* cellmap,
                          const int32_t* ev_off, const Ev* ev_tab, int B, int T, int L, int C, int D, int W, float* E, size_t lds)
{
    const int lanes = W / 4 < 128 ? W / 4 : 128;
    const int nslot = cdiv(D, 4 * lanes);
    if (128 / lanes > 1) lds += (size_t)128 * 16 * (nslot == 1 ? 1 : nslot <= 4 ? 4 : 8);
    if (dfm) {
        if (nslot == 1) return launch_events2_t<Src, 1, 8, true>(st, src, dfm, cells, cellmap, ev_off, ev_tab, B, T, L, C, D, E, lds);
        if (nslot <= 4) return launch_events2_t<Src, 4, 2, true>(st, src, dfm, cells, cellmap, ev_off, ev_tab, B, T, L, C, D, E, lds);
        return launch_events2_t<Src, 8, 1, true>(st, src, dfm, cells, cellmap, ev_off, ev_tab, B, T, L, C, D, E, lds);
    }
    if (nslot == 1) return launch_events2_t<Src, 1, 8, false>(st, src, dfm, cells, cellmap, ev_off, ev_tab, B, T, L, C, D, E, lds);
    if (nslot <= 4) return launch_events2_t<Src, 4, 4, false>(st, src, dfm, cells, cellmap, ev_off, ev_tab, B, T, L, C, D, E, lds);
    return launch_events2_t<Src, 8, 2, false>(st, src, dfm, cells, cellmap, ev_off, ev_tab, B, T, L, C, D, E, lds);
}

// The clip-boundary table of a geometry (T, L, C): call once with table == NULL to get counts[T], build the exclusive
// prefix offsets[T + 1] on the caller's side, then call again to fill table[offsets[T]] (8-byte entries).
extern "C" int smin_clip_event_table(void* stream, int T, int L, int C, int32_t* counts, const int32_t* offsets, void* table)
{
    SMIN_REQUIRE(L >= 1 && T >= L && T % L == 0 && C >= 1 && C <= 8 && L <= 4096);
    SMIN_REQUIRE((table == nullptr) != (counts == nullptr));
    hipLaunchKernelGGL(clip_event_table_kernel, dim3(T), dim3(64), 0, (hipStream_t)stream, T, L, C, counts, offsets, reinterpret_cast<Ev*>(table));
    SMIN_LAUNCH_CHECK();
    return 0;
}

extern "C" int smin_proposal_map_bwd(void* stream, const float* dfc, const float* dfm, const float* dfb,
                                        const int32_t* cells, const int32_t* row_ptr, const int32_t* cellmap,
                                        int N, int B, int T, int L, int C, int D, float* df, void* ws, size_t ws_bytes,
                                        const int32_t* ev_offsets, const void* ev_table)
{
    (void)row_ptr;
    hipStream_t st = (hipStream_t)stream;
    SMIN_REQUIRE(D % 4 == 0 && L >= 1 && T >= L && T % L == 0 && C >= 1 && D <= 2048);
    float* E = nullptr;
    if (N > 0 && (dfc || dfm)) {
        SMIN_REQUIRE(ws_bytes >= sizeof(float) * (size_t)B * T * D);
        E = reinterpret_cast<float*>(ws);
        size_t lds;
        if (ev_table && ev_offsets && events2_ok(N, T, L, C, &lds)) {
            const int rc = launch_events2(st, DenseClipGrad{dfc}, dfm, cells, cellmap, ev_offsets, reinterpret_cast<const Ev*>(ev_table), B, T, L, C, D, D, E, lds);
            if (rc) return rc;
        } else {
            hipLaunchKernelGGL((proposal_map_bwd_events_kernel<DenseClipGrad>), dim3(T, B), dim3(128), 0, st, DenseClipGrad{dfc}, dfm, cells, cellmap, T, L, C, D, E);
            SMIN_LAUNCH_CHECK();
        }
    }
    if (E) hipLaunchKernelGGL((time_scan_kernel<false>), dim3(cdiv(D, 64), B), dim3(1024), 0, st, E, dfb, T, L, D, (double*)nullptr, df);
    else hipLaunchKernelGGL(proposal_map_bwd_scan_kernel, dim3(cdiv(B * D, 128)), dim3(128), 0, st, E, dfb, B, T, L, D, df);
    SMIN_LAUNCH_CHECK();
    return 0;
}

// Clip means of per-frame features g [B][T][nseg*W] over every cell's clip windows, plus a bias on every clip row
// (empty clips included), one output tensor per segment of W features:
//   out[s][n*C + c][:] = m * (mean_{t in clip c of cell n} g[b][t][s*W : (s+1)*W] + bias[s*W : (s+1)*W])
// (bias covers the first bias_len features only: later segments get their constant where its gradient is free)
// With g = f [Wch_1; ..; Wch_k]^T this is every layer's linear_c_hat applied to ProposalGeneration's f_c without ever
// forming f_c (content stream).  Backward: smin_clip_window_means_bwd.
__global__ __launch_bounds__(128)
void clip_window_means_kernel(const double* __restrict__ Pf, const float* __restrict__ bias, int bias_len, const int* __restrict__ cells,
                              int T, int L, int C, int W, int nseg, size_t rows, float* __restrict__ out)
{
    // one workgroup per cell; a thread owns float4 columns of the nseg*W features and walks the cell's clips along the
    // shared boundary rows of the prefix
    const int n = blockIdx.x, D = W * nseg;
    const Cell cl = load_cell(cells, n);
    const int r = T / L, w = cl.j - cl.i + 1, nf = w * r, cs = max(1, nf / C);
    const int nclip = (cl.m != 0 && w >= 1) ? min(C, nf) : 0;
    const double inv = (double)(1.0f / (float)cs);
    for (int d = threadIdx.x * 4; d < D; d += 512) {
        const int sg = d / W, dw = d - sg * W;
        const float4 b4 = (d < bias_len && cl.m != 0) ? ldg4(bias + d) : f4zero();
        const double* pb = Pf + ((size_t)cl.b * (T + 1) + cl.i * r) * D + d;
        double4_ lo = {pb[0], pb[1], pb[2], pb[3]};
        float* o = out + ((size_t)sg * rows + (size_t)n * C) * W + dw;
        for (int c = 0; c < C; ++c) {
            float4 v = b4;
            if (c < nclip) {
                const double* pe = pb + (size_t)((c + 1) * cs) * D;
                const double4_ hi = {pe[0], pe[1], pe[2], pe[3]};
                v = f4add(v, make_float4((float)((hi.x - lo.x) * inv), (float)((hi.y - lo.y) * inv),
                                         (float)((hi.z - lo.z) * inv), (float)((hi.w - lo.w) * inv)));
                lo = hi;
            }
            stg4(o + (size_t)c * W, v);
        }
    }
}


extern "C" int smin_clip_window_means_fwd(void* stream, const float* g, const float* bias, int bias_len, const int32_t* cells, int N, int B, int T, int L, int C,
                                          int W, int nseg, float* out, void* ws, size_t ws_bytes)
{
    hipStream_t st = (hipStream_t)stream;
    const int D = W * nseg;
    SMIN_REQUIRE(W % 4 == 0 && nseg >= 1 && nseg <= 8 && L >= 1 && T >= L && T % L == 0 && C >= 1 && D <= 2048);
    SMIN_REQUIRE(bias_len % 4 == 0 && bias_len >= 0 && bias_len <= D);
    SMIN_REQUIRE(ws_bytes >= sizeof(double) * (size_t)B * (T + 1) * D);
    if (N == 0) return 0;
    double* Pf = reinterpret_cast<double*>(ws);
    hipLaunchKernelGGL((time_scan_kernel<true>), dim3(cdiv(D, 64), B), dim3(1024), 0, st, g, (const float*)nullptr, T, L, D, Pf, (float*)nullptr);
    SMIN_LAUNCH_CHECK();
    const size_t rows = (size_t)N * C;
    hipLaunchKernelGGL(clip_window_means_kernel, dim3(N), dim3(128), 0, st, Pf, bias, bias ? bias_len : 0, cells, T, L, C, W, nseg, rows, out);
    SMIN_LAUNCH_CHECK();
    return 0;
}

// dg [B][T][nseg*W] from the nseg gradient tensors dout[s] [N*C][W] (host array of device pointers).
extern "C" int smin_clip_window_means_bwd(void* stream, const float* const* dout, const int32_t* cells, const int32_t* row_ptr,
                                          const int32_t* cellmap, int N, int B, int T, int L, int C, int W, int nseg, float* dg,
                                          void* ws, size_t ws_bytes, const int32_t* ev_offsets, const void* ev_table)
{
    (void)row_ptr;
    hipStream_t st = (hipStream_t)stream;
    const int D = W * nseg;
    SMIN_REQUIRE(W % 4 == 0 && nseg >= 1 && nseg <= 8 && L >= 1 && T >= L && T % L == 0 && C >= 1 && D <= 2048);
    float* E = nullptr;
    if (N > 0) {
        SMIN_REQUIRE(ws_bytes >= sizeof(float) * (size_t)B * T * D);
        E = reinterpret_cast<float*>(ws);
        size_t lds;
        SegClipGrad src;
        for (int k = 0; k < 8; ++k) src.seg[k] = dout[k < nseg ? k : 0];
        src.W = W;
        if (ev_table && ev_offsets && events2_ok(N, T, L, C, &lds) && W <= 512) {
            const int rc = launch_events2(st, src, nullptr, cells, cellmap, ev_offsets, reinterpret_cast<const Ev*>(ev_table), B, T, L, C, D, W, E, lds);
            if (rc) return rc;
        } else {
            hipLaunchKernelGGL((proposal_map_bwd_events_kernel<SegClipGrad>), dim3(T, B), dim3(128), 0, st, src, (const float*)nullptr, cells, cellmap, T, L, C, D, E);
            SMIN_LAUNCH_CHECK();
        }
    }
    if (E) hipLaunchKernelGGL((time_scan_kernel<false>), dim3(cdiv(D, 64), B), dim3(1024), 0, st, E, (const float*)nullptr, T, L, D, (double*)nullptr, dg);
    else hipLaunchKernelGGL(proposal_map_bwd_scan_kernel, dim3(cdiv(B * D, 128)), dim3(128), 0, st, E, (const float*)nullptr, B, T, L, D, dg);
    SMIN_LAUNCH_CHECK();
    return 0;
}
