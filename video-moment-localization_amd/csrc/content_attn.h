// Launchers of the MFMA attention core (content_attn.hip), used by content_unit.hip.
// cc_rows [N*C][dl] and/or cc_mean [N][dl] = mean_c (either may be NULL).
// g_per_cell: dcchat is [N][dl], the same gradient row for every clip of a cell; gscale multiplies the gradient rows.
#pragma once
#include "gemm.h"

namespace smin {

int launch_content_attn_fwd(hipStream_t st, const float* chat, const int* cells, const int* row_ptr, int N, int B, int L, int C,
                            const float* Mq, const float* uq, const float* what, const float* shat, const float* qmask,
                            float* cc_rows, float* cc_mean, int dl, int Nq);

// scratch floats needed by launch_content_attn_bwd for N cells (partial slabs of the word-side gradients)
size_t content_attn_bwd_ws_floats(int N, int B, int dl);

int launch_content_attn_bwd(hipStream_t st, const float* chat, const float* dcchat, const int* cells, const int* row_ptr, int N, int B, int L, int C,
                            const float* Mq, const float* uq, const float* what, const float* shat, const float* qmask,
                            float* dchat, float* dMq, float* duq, float* dwhat, float* dshat, float* ws, int dl, int Nq, int g_per_cell, float gscale,
                            const float* dmean2 /* nullable [N][dl]: a second, per-cell gradient added as dmean2 * mscale */, float mscale);

}  // namespace smin
