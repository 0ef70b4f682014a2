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
