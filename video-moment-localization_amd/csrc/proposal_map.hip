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

// Pf[b][t][d] = sum_{t' < t} f[b][t'][d], t = 0..T
__global__ void time_prefix_kernel(const float* __restrict__ f, double* __restrict__ Pf, int B, int T, int D)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * D) return;
    const int b = idx / D, d = idx % D;
    const float* src = f + (size_t)b * T * D + d;
    double* dst = Pf + (size_t)b * (T + 1) * D + d;
    double run = 0.0;
    dst[0] = 0.0;
    for (int t = 0; t < T; ++t) { run += (double)src[(size_t)t * D]; dst[(size_t)(t + 1) * D] = run; }
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
    for (int d = threadIdx.x * 4; d < D; d += 512) {
        float4 sum = f4zero();
        for (int c = 0; c < C; ++c) {
            float4 v = f4zero();
            if (c < nclip) {
                const int s = cl.i * r + c * cs;
                const double* ps = P + (size_t)s * D + d;
                const double* pe = P + (size_t)(s + cs) * D + d;
                v.x = (float)((pe[0] - ps[0]) * inv); v.y = (float)((pe[1] - ps[1]) * inv);
                v.z = (float)((pe[2] - ps[2]) * inv); v.w = (float)((pe[3] - ps[3]) * inv);
            }
            stg4(fc + ((size_t)n * C + c) * D + d, v);
            sum = f4add(sum, v);
        }
        stg4(fm + (size_t)n * D + d, make_float4(sum.x / C, sum.y / C, sum.z / C, sum.w / C));
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

// E[b][t][:] = sum over clips starting at t of g/cs  -  sum over clips ending at t of g/cs,
// g = m * (dfc[n][c][:] + dfm[n][:] / C).  One workgroup per (b, t); deterministic (pure gather).
__global__ __launch_bounds__(128)
void proposal_map_bwd_events_kernel(const float* __restrict__ dfc, const float* __restrict__ dfm,
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
                if (dfc) g = ldg4(dfc + ((size_t)n * C + c) * D + d);
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

// Same result as proposal_map_bwd_events_kernel, organised for the machine: the clip equations of the start-snippet
// rows are solved by one LANE per row (the version above has every thread of the workgroup repeat the whole scalar
// search: ~25k integer instructions per workgroup), each lane leaving its events {cell, clip, +-1/cs, with-mean} in a
// fixed slot range of an LDS list; the row that starts exactly at frame t (one event per window width) gets a list
// of its own.  Then all threads stream the lists in (row, slot) order -- the order is fixed, so the sum is
// bitwise reproducible -- with the dependent cellmap / mask lookups already resolved.
constexpr int EV_SLOTS = 16;
struct Ev { int nc; float sc; };                      // nc = (cell << 3) | (with_mean << 2 ... ) see pack below

__global__ __launch_bounds__(128)
void proposal_map_bwd_events2_kernel(const float* __restrict__ dfc, const float* __restrict__ dfm,
                                     const int* __restrict__ cells, const int* __restrict__ cellmap,
                                     int T, int L, int C, int D, float* __restrict__ E)
{
    extern __shared__ __attribute__((aligned(8))) unsigned char lds_raw[];
    Ev* rowev = reinterpret_cast<Ev*>(lds_raw);                    // [L][EV_SLOTS]
    Ev* zeroev = rowev + (size_t)L * EV_SLOTS;                      // [L]      events of the row with base == 0
    int* cnt = reinterpret_cast<int*>(zeroev + L);                  // [L]
    const int t = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int r = T / L;
    const int* cmap = cellmap + (size_t)b * L * L;
    const float invC = 1.0f / C;
    const int imax = min(L - 1, t / r);
    const int i0 = (t % r == 0 && t / r < L) ? t / r : -1;          // row whose window starts at frame t

    auto lookup = [&](int i, int w) -> int {                        // cell id of (i, i+w-1) if present and unmasked
        const int j = i + w - 1;
        if (j >= L) return -1;
        const int n = cmap[i * L + j];
        if (n < 0 || cells[4 * (size_t)n + 3] == 0) return -1;
        return n;
    };
    for (int i = tid; i < L; i += 128) {
        int k = 0;
        if (i <= imax && i != i0) {
            const int base = t - i * r;
            auto widths = [&](int c, int cs, float sign) {
                int lo = (cs * C + r - 1) / r, hi = ((cs + 1) * C + r - 1) / r - 1;
                if (cs == 1) lo = 1;
                lo = max(lo, c / r + 1);
                hi = min(hi, L - i);
                for (int w = lo; w <= hi; ++w) {
                    const int n = lookup(i, w);
                    if (n >= 0 && k < EV_SLOTS) {
                        const int with_m = (sign < 0.f && c + 1 == min(C, w * r)) ? 1 : 0;
                        rowev[i * EV_SLOTS + k++] = Ev{(n << 4) | (with_m << 3) | c, sign / (float)cs};
                    }
                }
            };
            for (int c = 1; c < C; ++c) if (base % c == 0) widths(c, base / c, 1.0f);
            for (int cc = 1; cc <= C; ++cc) if (base % cc == 0) widths(cc - 1, base / cc, -1.0f);
        }
        cnt[i] = k;
    }
    if (i0 >= 0)
        for (int w = tid + 1; w <= L - i0; w += 128) {
            const int n = lookup(i0, w);
            zeroev[w - 1] = Ev{n >= 0 ? ((n << 4) | 8) : -1, 1.0f / (float)max(1, (w * r) / C)};
        }
    __syncthreads();

    float4 acc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] = f4zero();
    // four events per step: their row loads are independent, so eight HBM requests are in flight per lane
    auto apply4 = [&](const Ev* ev, int cnt4) {
        float4 gc[4], gm[4];
        float sc[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = acc[k];
        for (int kd = 0; kd < 4; ++kd) {
            const int d = tid * 4 + 512 * kd;
            if (d >= D) break;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const Ev e = ev[min(u, cnt4 - 1)];
                const bool ok = u < cnt4 && e.nc >= 0;
                const int n = ok ? e.nc >> 4 : 0, c = e.nc & 7;
                sc[u] = ok ? e.sc : 0.f;
                gc[u] = (dfc && ok) ? ldg4(dfc + ((size_t)n * C + c) * D + d) : f4zero();
                gm[u] = (dfm && ok && (e.nc & 8)) ? ldg4(dfm + (size_t)n * D + d) : f4zero();
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[kd] = f4fma(f4fma(gm[u], invC, gc[u]), sc[u], acc[kd]);
        }
    };
    for (int i = 0; i <= imax; ++i) {
        const Ev* list = i == i0 ? zeroev : rowev + (size_t)i * EV_SLOTS;
        const int k = i == i0 ? L - i0 : cnt[i];
        for (int q = 0; q < k; q += 4) apply4(list + q, min(4, k - q));
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int d = tid * 4 + 512 * k;
        if (d < D) stg4(E + ((size_t)b * T + t) * D + d, acc[k]);
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
    hipLaunchKernelGGL(time_prefix_kernel, dim3(cdiv(B * D, 128)), dim3(128), 0, st, f, Pf, B, T, D);
    SMIN_LAUNCH_CHECK();
    if (N > 0) {
        hipLaunchKernelGGL(proposal_map_fwd_kernel, dim3(N), dim3(128), 0, st, Pf, cells, T, L, C, D, fc, fm);
        SMIN_LAUNCH_CHECK();
    }
    const size_t tot = (size_t)B * L * (D / 4);
    hipLaunchKernelGGL(boundary_pool_fwd_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, f, T, L, D / 4, fb, tot);
    SMIN_LAUNCH_CHECK();
    return 0;
}

extern "C" int smin_proposal_map_bwd(void* stream, const float* dfc, const float* dfm, const float* dfb,
                                        const int32_t* cells, const int32_t* row_ptr, const int32_t* cellmap,
                                        int N, int B, int T, int L, int C, int D, float* df, void* ws, size_t ws_bytes)
{
    hipStream_t st = (hipStream_t)stream;
    SMIN_REQUIRE(D % 4 == 0 && L >= 1 && T >= L && T % L == 0 && C >= 1 && D <= 2048);
    float* E = nullptr;
    if (N > 0 && (dfc || dfm)) {
        SMIN_REQUIRE(ws_bytes >= sizeof(float) * (size_t)B * T * D);
        E = reinterpret_cast<float*>(ws);
        const int r = T / L;
        const size_t lds = (size_t)L * EV_SLOTS * 8 + (size_t)L * 8 + (size_t)L * 4;
        if (2 * C * cdiv(C, r) <= EV_SLOTS && C <= 8 && N < (1 << 27) && lds <= 150 * 1024) {
            if (lds > 48 * 1024) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(proposal_map_bwd_events2_kernel),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                if (e != hipSuccess) return (int)e;
            }
            hipLaunchKernelGGL(proposal_map_bwd_events2_kernel, dim3(T, B), dim3(128), lds, st, dfc, dfm, cells, cellmap, T, L, C, D, E);
        } else {
            hipLaunchKernelGGL(proposal_map_bwd_events_kernel, dim3(T, B), dim3(128), 0, st, dfc, dfm, cells, cellmap, T, L, C, D, E);
        }
        SMIN_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(proposal_map_bwd_scan_kernel, dim3(cdiv(B * D, 128)), dim3(128), 0, st, E, dfb, B, T, L, D, df);
    SMIN_LAUNCH_CHECK();
    return 0;
}
