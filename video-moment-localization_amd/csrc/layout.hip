// Packed valid-cell layout from a (B, L, L) mask (the reference keeps the dense map and multiplies by
// moment_mask after every op, models.py:117-303): cells[n] = {b, i, j, m} sorted by (b, i, j), row_ptr[b*L + i] = first
// cell of start-snippet row (b, i), cellmap[b][i][j] = cell id or -1.  Two launches instead of the ~15 of a
// nonzero / cumsum / index_put formulation; the cell count N is known to the caller (it sizes the outputs).
#include "common.h"
#include "smin_hip.h"

namespace smin {

// row_ptr[r + 1] = inclusive running count of listed cells over rows r = b*L + i; one workgroup, rows in chunks of 1024.
// all_cells: every (b, i, j) is listed (count = L per row).
__global__ __launch_bounds__(1024)
void layout_rows_kernel(const uint8_t* __restrict__ mask, int rows, int L, int all_cells, int* __restrict__ row_ptr)
{
    __shared__ int part[1024];
    __shared__ int carry;
    const int t = threadIdx.x;
    if (t == 0) { carry = 0; row_ptr[0] = 0; }
    __syncthreads();
    for (int r0 = 0; r0 < rows; r0 += 1024) {
        const int r = r0 + t;
        int c = 0;
        if (r < rows) {
            if (all_cells) c = L;
            else {
                const uint8_t* m = mask + (size_t)r * L;
                if ((L & 15) == 0) {                                // 16 mask bytes per load (rows are L bytes apart: 16-byte aligned with the base)
                    for (int j = 0; j < L; j += 16) {
                        const uint4 v = *reinterpret_cast<const uint4*>(m + j);
                        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                        for (int q = 0; q < 4; ++q) {               // bytes are 0 / non-zero: count the non-zero ones
                            const unsigned nz = (w[q] | (w[q] >> 4)) & 0x0f0f0f0fu, t2 = (nz | (nz >> 2)) & 0x03030303u, t1 = (t2 | (t2 >> 1)) & 0x01010101u;
                            c += __popc(t1);
                        }
                    }
                } else {
                    for (int j = 0; j < L; ++j) c += m[j] != 0;
                }
            }
        }
        part[t] = c;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {                       // Hillis-Steele inclusive scan
            const int v = t >= o ? part[t - o] : 0;
            __syncthreads();
            part[t] += v;
            __syncthreads();
        }
        if (r < rows) row_ptr[r + 1] = carry + part[t];
        __syncthreads();
        if (t == 1023) carry += part[1023];
        __syncthreads();
    }
}

// one wave per row (b, i): compacts the listed j's by ballot rank
// n_cap >= 0: the caller sized `cells` for exactly n_cap entries without asking the device (a captured step replays with the count
// it was captured for): nothing is written past n_cap, entries the mask does not fill are set to an inert in-range cell, and
// *status becomes 1 when the mask lists a different number of cells -- the step's results are then meaningless, but in bounds.
__global__ __launch_bounds__(256)
void layout_fill_kernel(const uint8_t* __restrict__ mask, const int* __restrict__ row_ptr, int rows, int L, int all_cells,
                        int* __restrict__ cells, int* __restrict__ cellmap, int n_cap, int* __restrict__ status)
{
    if (n_cap >= 0 && blockIdx.x == 0) {
        const int total = row_ptr[rows];
        if (threadIdx.x == 0 && total != n_cap) *status = 1;
        for (int id = total + (int)threadIdx.x; id < n_cap; id += 256) *reinterpret_cast<int4*>(cells + 4 * (size_t)id) = make_int4(0, 0, 0, 0);
    }
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= rows) return;
    const int b = r / L, i = r - b * L;
    int n = row_ptr[r];
    for (int j0 = 0; j0 < L; j0 += 64) {
        const int j = j0 + lane;
        const bool in = j < L;
        const bool flag = in && mask[(size_t)r * L + j] != 0;
        const bool listed = in && (all_cells || flag);
        const unsigned long long bal = __ballot(listed);
        const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0));
        const bool stored = listed && (n_cap < 0 || n + rank < n_cap);
        if (stored) {
            const int id = n + rank;
            *reinterpret_cast<int4*>(cells + 4 * (size_t)id) = make_int4(b, i, j, flag ? 1 : 0);
        }
        if (in) cellmap[(size_t)r * L + j] = stored ? n + rank : -1;
        n += __popcll(bal);
    }
}

}  // namespace smin

using namespace smin;

// mask: uint8 / bool [B][L][L] (non-zero = valid).  all_cells = 0: list the valid cells (m = 1); 1: list every cell with
// m = mask.  cells [N][4], row_ptr [B*L + 1], cellmap [B][L][L]; N must equal the number of listed cells.
extern "C" int smin_build_cells(void* stream, const uint8_t* mask, int B, int L, int all_cells,
                                int32_t* cells, int32_t* row_ptr, int32_t* cellmap)
{
    hipStream_t st = (hipStream_t)stream;
    SMIN_REQUIRE(B >= 1 && L >= 1);
    const int rows = B * L;
    hipLaunchKernelGGL(layout_rows_kernel, dim3(1), dim3(1024), 0, st, mask, rows, L, all_cells, row_ptr);
    SMIN_LAUNCH_CHECK();
    hipLaunchKernelGGL(layout_fill_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, st, mask, row_ptr, rows, L, all_cells, cells, cellmap, -1, (int*)nullptr);
    SMIN_LAUNCH_CHECK();
    return 0;
}

// The same for a caller that knows the number of listed cells already (n_expected, e.g. a captured step: no device -> host round
// trip inside the step).  `cells` holds exactly n_expected entries; *status (device, int32) is set to 1 if the mask lists another
// number -- see layout_fill_kernel.  The caller clears *status once and reads it when convenient.
extern "C" int smin_build_cells_n(void* stream, const uint8_t* mask, int B, int L, int all_cells, int n_expected,
                                  int32_t* cells, int32_t* row_ptr, int32_t* cellmap, int32_t* status)
{
    hipStream_t st = (hipStream_t)stream;
    SMIN_REQUIRE(B >= 1 && L >= 1 && n_expected >= 0 && status != nullptr);
    const int rows = B * L;
    hipLaunchKernelGGL(layout_rows_kernel, dim3(1), dim3(1024), 0, st, mask, rows, L, all_cells, row_ptr);
    SMIN_LAUNCH_CHECK();
    hipLaunchKernelGGL(layout_fill_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, st, mask, row_ptr, rows, L, all_cells, cells, cellmap, n_expected, status);
    SMIN_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------- the step's mask prologue
// Everything the train step derives from its four masks and the localization head's parameters before its first real kernel, in
// one launch (as torch calls: two reductions, four conversions, a stack and a concatenation -- eight launches in front of the
// query encoder, which opens the step's critical path):
//   len32[b] = number of words of query b, qmf / vmaskf / lmf = the masks as fp32, count = number of valid cells (moment_mask),
//   wb [3][D] / bb [3] = the three boundary heads' weights and biases side by side (Localization, models.py:318-333).
namespace smin {
struct PrologueArgs {
    const unsigned char *qmask, *vmask, *lmask, *mmask;
    const float* w[3]; const float* b[3];
    int B, Nq, T, L, D;
    int* len32; float *qmf, *vmaskf, *lmf, *wb, *bb;
    long long* count; unsigned long long* acc;                   // acc[0] = running sum, acc[1] = ticket (both zero between launches)
};

__global__ __launch_bounds__(256)
void prologue_kernel(PrologueArgs a)
{
    const size_t nq = (size_t)a.B * a.Nq, nv = (size_t)a.B * a.T, nl = (size_t)a.B * a.L, nw = (size_t)3 * a.D, nm = (size_t)a.B * a.L * a.L;
    const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, nth = (size_t)gridDim.x * 256;
    for (size_t i = tid; i < nq; i += nth) a.qmf[i] = a.qmask[i] ? 1.f : 0.f;
    for (size_t i = tid; i < nv; i += nth) a.vmaskf[i] = a.vmask[i] ? 1.f : 0.f;
    for (size_t i = tid; i < nl; i += nth) a.lmf[i] = a.lmask[i] ? 1.f : 0.f;
    for (size_t i = tid; i < nw; i += nth) { const int h = (int)(i / a.D); a.wb[i] = (h == 0 ? a.w[0] : h == 1 ? a.w[1] : a.w[2])[i - (size_t)h * a.D]; }
    if (tid < 3) a.bb[tid] = (tid == 0 ? a.b[0] : tid == 1 ? a.b[1] : a.b[2])[0];
    for (size_t b = tid; b < (size_t)a.B; b += nth) {
        int s = 0;
        for (int q = 0; q < a.Nq; ++q) s += a.qmask[b * a.Nq + q] ? 1 : 0;
        a.len32[b] = s;
    }
    // valid cells: integer sums (any order gives the same count); the last workgroup to arrive publishes the total and clears the words
    unsigned int part = 0;
    for (size_t i = tid; i < nm; i += nth) part += a.mmask[i] ? 1u : 0u;
    __shared__ unsigned int red[256];
    red[threadIdx.x] = part;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s]; __syncthreads(); }
    if (threadIdx.x == 0) {
        atomicAdd(&a.acc[0], (unsigned long long)red[0]);
        __threadfence();
        const unsigned long long t = atomicAdd(&a.acc[1], 1ull);
        if (t == (unsigned long long)gridDim.x - 1) {
            __threadfence();
            a.count[0] = (long long)atomicAdd(&a.acc[0], 0ull);
            atomicExch(&a.acc[0], 0ull);
            atomicExch(&a.acc[1], 0ull);
        }
    }
}
}  // namespace smin

// masks: one byte per element (bool / uint8), non-zero = set.  w / b: HOST arrays of the three heads' weight [D] and bias [1] device
// pointers.  count: device int64; acc: two device words that are zero on entry (the launch leaves them zero).
extern "C" int smin_step_prologue(void* stream, const uint8_t* query_mask, const uint8_t* video_mask, const uint8_t* length_mask, const uint8_t* moment_mask,
                                  const float* const* w, const float* const* b, int B, int Nq, int T, int L, int D, int32_t* len32, float* qmf, float* vmaskf,
                                  float* lmf, float* wb, float* bb, int64_t* count, void* acc)
{
    SMIN_REQUIRE(B >= 1 && Nq >= 1 && T >= 1 && L >= 1 && D >= 1 && acc != nullptr && count != nullptr);
    PrologueArgs a;
    a.qmask = query_mask; a.vmask = video_mask; a.lmask = length_mask; a.mmask = moment_mask;
    for (int h = 0; h < 3; ++h) { a.w[h] = w[h]; a.b[h] = b[h]; }
    a.B = B; a.Nq = Nq; a.T = T; a.L = L; a.D = D;
    a.len32 = len32; a.qmf = qmf; a.vmaskf = vmaskf; a.lmf = lmf; a.wb = wb; a.bb = bb;
    a.count = reinterpret_cast<long long*>(count); a.acc = reinterpret_cast<unsigned long long*>(acc);
    const size_t work = (size_t)B * L * L > (size_t)B * T ? (size_t)B * L * L : (size_t)B * T;
    int grid = (int)((work + 256 * 8 - 1) / (256 * 8));
    grid = grid < 1 ? 1 : grid > 256 ? 256 : grid;
    hipLaunchKernelGGL(prologue_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    SMIN_LAUNCH_CHECK();
    return 0;
}
