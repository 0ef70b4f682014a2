/* smin_hip.h -- C ABI of the MI355X (gfx950) SMIN hot path: cross-modal fusion + 2D temporal
 * proposal scoring of ChanukyaVardhan/Video-Moment-Localization (reference models.py).
 *
 * The reference has no FFI of its own (pure PyTorch): its de-facto operator boundary is
 * the nn.Module surface of models.py.  Each entry point below replaces the body of one of those
 * modules' forward() (cited per function) for fwd and for the autograd backward torch would derive.
 * The Python host (video-moment-localization_amd/) binds these with ctypes and keeps the reference's
 * module/ctor/forward/state_dict surface.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to contiguous fp32 unless stated; `stream` is a hipStream_t.
 *   - all calls are asynchronous on `stream`, allocate nothing and never synchronise (graph-capturable);
 *     scratch comes from the caller-provided workspace `ws` (size from smin_workspace_bytes).
 *   - return 0 on success, a positive hipError_t, or a negative code for a rejected argument.
 *   - requirements: D % 4 == 0, dl % 16 == 0, 16 <= dl <= 128, 2 <= C <= 4, Nq <= 32.
 *
 * Packed valid-cell layout: the L x L map is stored as a list of N cells sorted by
 * (b, i, j):  cells[n] = {b, i, j, m} (int32 x4, m = moment_mask[b,i,j]);
 *             row_ptr[b*L+i] .. row_ptr[b*L+i+1] = the cells of start-snippet row (b, i)  (B*L+1 ints);
 *             cellmap[b][i][j] = n or -1.
 * Per-cell tensors: f_c [N][C][D], f_m [N][D].  With moment-mask driven lists (m == 1 everywhere) this
 * is bit-equivalent to the reference's dense (B,L,L,..) tensors, whose masked cells are exactly zero;
 * with a list of all B*L*L cells it reproduces the dense sub-module seams for arbitrary inputs.
 */
#ifndef SMIN_HIP_H
#define SMIN_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2 (round 3): smin_build_cells_n, smin_step_prologue, smin_lstm_pack_layers, smin_bilstm_layer_bwd_weights, smin_video_encoder_gate
 * (smin_video_encoder_fwd with fs == f == NULL), smin_lstm_cluster_error; new trailing arguments of smin_gate_bwd (cells, boundary
 * A, boundary dout), smin_moment_unit_bwd[_x1h] (dfb_acc) and smin_build_targets; smin_score_map_bwd may be issued as two halves;
 * content attention requires dl % 16 == 0; round-2 changes that were not versioned: smin_linear_rows_bwd accepts dW == NULL,
 * smin_video_encoder_bwd / smin_bilstm_layer_bwd may be issued as two halves */
#define SMIN_HIP_ABI_VERSION 2

int smin_abi_version(void);
/* Arithmetic of the dense contractions (forward maps, input gradients, weight gradients):
 *   0 (default) exact fp32 on v_mfma_f32_32x32x2_f32;
 *   1 split-bf16: operands split into hi+lo bf16 on the fly, hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16 with
 *     fp32 accumulation (~2^-16 relative per product, ~1e-5 on a dot product);
 *   2 plain bf16: operands rounded to bf16 once, fp32 accumulation (~4e-3 relative per product; BASELINE.json configs[1]);
 *   3 fp32 emulated on the bf16 matrix cores: operands split EXACTLY into three bf16 pieces (3 x 8 mantissa bits), six
 *     products (hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid) accumulated in fp32, smallest first; the three dropped terms
 *     are <= 2^-26 relative per product, below fp32's own product rounding -- results agree with mode 0 to fp32 rounding.
 * Process-wide; returns 0 or -1 for an unknown mode. */
int smin_set_gemm_mode(int mode);
int smin_get_gemm_mode(void);
/* Launch timing for benchmarks: while enabled, chosen launches inside the entry points below are bracketed by HIP events
 * recorded on the stream the launch goes to (tags: SMIN_PROF_*).  smin_prof_enable(1) clears earlier records;
 * smin_prof_read waits for the recorded events, writes up to `cap` (tag, milliseconds) pairs in launch order and returns
 * how many it wrote (or a negative code).  Off by default; costs two event records per tagged launch when on. */
#define SMIN_PROF_MOMENT_FWD 1   /* moment unit, forward contraction   mu = [x1 | mean_c f_c] Wcat^T          */
#define SMIN_PROF_MOMENT_DX 2    /* moment unit, input gradient        dX = dmu Wcat                            */
#define SMIN_PROF_MOMENT_DW 3    /* moment unit, weight gradient       dWcat = dmu^T [x1 | mean_c f_c]         */
#define SMIN_PROF_ATTN_FWD 4     /* content attention core, forward                                             */
#define SMIN_PROF_ATTN_BWD 5     /* content attention core, backward (all its launches)                         */
int smin_prof_enable(int on);
int smin_prof_read(int32_t* tags, float* ms, int cap);
/* "gfx950" -- the only code object in the library */
const char* smin_target_arch(void);
/* bytes of scratch any single call below may need for N cells (C, D, dl, Nq, B as given) */
size_t smin_workspace_bytes(int N, int B, int C, int D, int dl, int Nq);

/* ---- ProposalGeneration.forward (models.py:115-126) with compute_content_matrix (models.py:88-98)
 * as index arithmetic: fc[n,c,:] = mean of f[b, start..start+cs) per clip, fm = mean_c fc (divides by C),
 * fb = AvgPool1d(T/L) over time.  f [B][T][D]; fc [N][C][D]; fm [N][D]; fb [B][L][D]. */
int smin_proposal_map_fwd(void* stream, const float* f, const int32_t* cells, int N, int B, int T, int L, int C, int D,
                          float* fc, float* fm, float* fb, void* ws, size_t ws_bytes /* >= 8*B*(T+1)*D */);
/* (any of fc/fm/fb may be NULL = not wanted) */
/* The clip-boundary table of a geometry (T, L, C): which clip of which (i, j) starts / ends at each frame -- the
 * sparsity pattern of the reference's cached content matrix (models.py:88-98, 110).  Build once per geometry: call
 * with table == NULL to get counts [T]; form the exclusive prefix offsets [T+1]; call with counts == NULL to fill
 * table (8 bytes per entry, offsets[T] entries). */
int smin_clip_event_table(void* stream, int T, int L, int C, int32_t* counts, const int32_t* offsets, void* table);
/* df [B][T][D] = d/df of the three outputs (any of dfc/dfm/dfb may be NULL = zero).  ws_bytes >= 4*B*T*D.
 * ev_offsets / ev_table: the geometry's clip-boundary table (NULL, NULL = derive the boundaries in-kernel, slower). */
int smin_proposal_map_bwd(void* stream, const float* dfc, const float* dfm, const float* dfb,
                          const int32_t* cells, const int32_t* row_ptr, const int32_t* cellmap,
                          int N, int B, int T, int L, int C, int D, float* df, void* ws, size_t ws_bytes,
                          const int32_t* ev_offsets, const void* ev_table);

/* ---- Gated moment feature shared by ContentUnit (models.py:272-274) and BoundaryUnit (models.py:191):
 *   hbar[n,:] = sigmoid(f_m[n,:] * f_s[b,:]) * f_m[n,:]          hbar [N][D]
 * backward: dfm [N][D], dfs [B][D].  hbar and f_m usually have several consumers (content unit, boundary unit, the
 * content stream's running sum; f_m also passes through to the moment unit's residual): dhbar / dres are HOST arrays of
 * n_dhbar (1..4) / n_dres (0..4) device pointers [N][D] whose sum is the gradient of hbar / is added to dfm. */
int smin_gate_fwd(void* stream, const float* fm, const float* fs, const int32_t* cells, int N, int D, float* hbar);
/* the same, and hsum_out = hsum_in + hbar [N][D] (running sum over layers, read by the next layers' gate term of chat) */
int smin_gate_fwd_sum(void* stream, const float* fm, const float* fs, const int32_t* cells, int N, int D, float* hbar, const float* hsum_in,
                      float* hsum_out);
/* boundary_A [B][L][L], boundary_dout [B][L][D] (both or neither; cells [N][4] then required): the boundary unit's consumer of hbar,
 * f_bm[b,i] = sum_j A[b,i,j] hbar[b,i,j] (models.py:191-194), enters as A[b,i,j] * boundary_dout[b,i,:] formed on the fly -- pass the
 * unit's saved attention A and the gradient of its output, and call smin_boundary_unit_bwd with dhbar == NULL. */
int smin_gate_bwd(void* stream, const float* const* dhbar, int n_dhbar, const float* const* dres, int n_dres,
                  const float* fm, const float* fs, const int32_t* row_ptr,
                  int N, int B, int L, int D, float* dfm, float* dfs, void* ws, size_t ws_bytes,
                  const int32_t* cells, const float* boundary_A, const float* boundary_dout);

/* ---- ContentUnit.forward (models.py:242-276) incl. ContentAttention.forward (models.py:207-226).
 * Query-side per-sample operands are prepared by the host (O(B*Nq*dl) work):
 *   what [B][Nq][dl] = linear_w_hat(f_w) * query_mask          shat [B][dl] = linear_s_hat(f_s)
 *   Mq   [B][Nq][dl] = W_k(what) @ W_q.weight                   uq   [B][Nq] = W_k(what) @ W_q.bias
 *   (so that  W_q(c_hat) . W_k(what)^T  ==  c_hat . Mq^T + uq  -- the per-cell dl x dl projection folds away)
 *   qmask [B][Nq] fp32 0/1;  hbar from smin_gate_fwd.
 * Outputs: fc_out [N][C][D], fcmean [N][D] = mean_c fc_out (consumed by the moment unit);
 * saved for backward: chat [N*C][dl], cchat [N*C][dl].
 * last != 0 (final SMI layer: nothing consumes fc_out itself, models.py:372-375): fc_out is not written (may be
 * NULL), cchat receives mean_c cchat [N][dl] and linear_c runs on N rows; fcmean_in [N][D] = mean_c fc (the
 * previous layer's fcmean, or the proposal map's f_m) is required. */
int smin_content_unit_fwd(void* stream, const float* fc, const float* hbar, const int32_t* cells, const int32_t* row_ptr,
                          int N, int B, int L, int C, int D, int dl, int Nq,
                          const float* Wch, const float* bch, const float* Mq, const float* uq,
                          const float* what, const float* shat, const float* qmask, const float* Wc, const float* bc,
                          const float* fcmean_in, int last,
                          float* fc_out, float* fcmean, float* chat, float* cchat);
/* Backward.  dfc_out may be NULL (last layer: only the moment unit consumes fc_out, through fcmean); pass the
 * same `last` as in forward (then dfc_out must be NULL and cchat is the [N][dl] clip mean).
 * WcT [dl][D] and WchT [D][dl] are transposed copies of the weights.
 * Gradients: dfc [N][C][D], dhbar [N][D], dWch [dl][D], dbch [dl], dMq [B][Nq][dl], duq [B][Nq],
 * dwhat [B][Nq][dl], dshat [B][dl], dWc [D][dl], dbc [D]. */
int smin_content_unit_bwd(void* stream, const float* dfc_out, const float* dfcmean,
                          const float* fc, const int32_t* cells, const int32_t* row_ptr,
                          int N, int B, int L, int C, int D, int dl, int Nq,
                          const float* WchT, const float* Mq, const float* uq,
                          const float* what, const float* shat, const float* qmask, const float* WcT,
                          const float* chat, const float* cchat,
                          float* dfc, float* dhbar, float* dWch, float* dbch, float* dMq, float* duq,
                          float* dwhat, float* dshat, float* dWc, float* dbc, void* ws, size_t ws_bytes, int last);

/* ---- BoundaryUnit.forward, the map-sized part (models.py:190-194):
 *   fbm[b,i,:] = sum_j A_b[b,i,j] * hbar[(b,i,j),:]
 * A_b [B][L][L] comes from the L x L boundary self-attention (models.py:164-188), which the host runs
 * as plain library GEMMs.  fbm [B][L][D]. */
int smin_boundary_reduce_fwd(void* stream, const float* Ab, const float* hbar, const int32_t* cells,
                             const int32_t* row_ptr, int N, int B, int L, int D, float* fbm);
/* dAb [B][L][L] (zero where no cell), dhbar [N][D]. */
int smin_boundary_reduce_bwd(void* stream, const float* dfbm, const float* Ab, const float* hbar,
                             const int32_t* cells, const int32_t* row_ptr, int N, int B, int L, int D,
                             float* dAb, float* dhbar);

/* ---- BoundaryUnit.forward, whole unit (models.py:164-196) with its word attention Attention.forward (models.py:137-154):
 *   out = (A_b f_b) * lm + f_b + sum_j A_b[i,j] hbar[(i,j)],  A_b = softmax(mask(bq bq^T / sqrt(D))) * lm,
 *   bq = f_b * (softmax(mask(W_q f_b . (W_k f_w)^T / sqrt(D))) f_w * lm + f_s).
 * fb [B][L][D], fw [B][Nq][D], fs [B][D], qmask [B][Nq], lmask [B][L] fp32 0/1; Wq, Wk [D][D] (+ biases).
 * Saved for backward (caller-allocated): Qb, baq, bqv [B][L][D], Kb [B][Nq][D], P [B][L][Nq], A [B][L][L]. */
int smin_boundary_unit_fwd(void* stream, const float* fb, const float* fw, const float* fs, const float* hbar,
                           const int32_t* cells, const int32_t* row_ptr, int N, int B, int L, int Nq, int D,
                           const float* Wq, const float* bq, const float* Wk, const float* bk,
                           const float* qmask, const float* lmask,
                           float* out, float* Qb, float* Kb, float* P, float* baq, float* bqv, float* A);
/* WqT, WkT: transposed weights.  Gradients: dfb, dfw, dfs, dhbar [N][D], dWq, dbq, dWk, dbk.
 * ws_bytes >= 4 * (2*B*L*L + 3*B*L*D + B*L*Nq + B*Nq*D + 2*64*(D*D + D)). */
int smin_boundary_unit_bwd(void* stream, const float* dout, const float* fb, const float* fw, const float* fs, const float* hbar,
                           const int32_t* cells, const int32_t* row_ptr, int N, int B, int L, int Nq, int D,
                           const float* WqT, const float* WkT, const float* qmask, const float* lmask,
                           const float* Qb, const float* Kb, const float* P, const float* baq, const float* bqv, const float* A,
                           float* dfb, float* dfw, float* dfs, float* dhbar, float* dWq, float* dbq, float* dWk, float* dbk,
                           void* ws, size_t ws_bytes);

/* ---- MomentUnit.forward (models.py:288-303): two 1x1 convs fused into one K = 2D contraction
 *   mu[n,:] = m * ( [fb[b,i]*fb[b,j] | fcmean[n]] @ Wcat^T + bcat ) + fm[n,:]
 * Wcat [D][2D] = [conv_layer_fb.weight | conv_layer_fc.weight], bcat [D] = sum of the two biases. */
int smin_moment_unit_fwd(void* stream, const float* fcmean, const float* fm, const float* fb, const int32_t* cells,
                         int N, int B, int L, int D, const float* Wcat, const float* bcat, float* mu,
                         const float* x1 /* nullable [N][D] = f_b[i]*f_b[j] from smin_pair_product: the contraction then
                                            reads two plain matrices (pass it to smin_moment_unit_bwd as well) */);
/* x1[n][:] = f_b[b][i][:] * f_b[b][j][:]  -- the pair half of the moment unit's left operand (models.py:292-294) */
int smin_pair_product(void* stream, const float* fb, const int32_t* cells, int N, int L, int D, float* x1);
/* WcatT [2D][D].  dfcmean [N][D], dfb [B][L][D], dWcat [D][2D], dbcat [D]; the residual gradient
 * d mu / d fm is the identity and is left to the caller (dfm += dmu).  Either half may be skipped: dfcmean == dfb == NULL
 * computes only the weight gradients, dWcat == dbcat == NULL only the input gradients (the halves share nothing but
 * dmu, so a host may run them on two streams; each call needs its own workspace). */
int smin_moment_unit_bwd(void* stream, const float* dmu, const float* fcmean, const float* fb, const int32_t* cells,
                         const int32_t* row_ptr, const int32_t* cellmap, int N, int B, int L, int D, const float* WcatT,
                         float* dfcmean, float* dfb, float* dWcat, float* dbcat, void* ws, size_t ws_bytes,
                         int all_valid /* 1: every listed cell has m == 1 (mask-driven list): skips the mask lookups */,
                         const float* dfcmean_acc /* nullable [N][D]: added into dfcmean (a second consumer of fcmean) */,
                         const float* x1 /* nullable: the pair product saved by smin_moment_unit_fwd */,
                         const float* dfb_acc /* nullable [B][L][D]: added into dfb (another consumer's gradient of f_b) */);
/* The same three with the pair product stored as bf16 (uint16_t bit patterns, round to nearest even), x1h [N][D]: half the bytes of the
 * largest saved tensor of a layer.  Under smin_set_gemm_mode(2) -- where every contraction rounds its operands to bf16 as it loads them --
 * the results equal the fp32-storage calls bit for bit; in the other modes the product loses its low bits.  all_valid must be 1. */
int smin_pair_product_bf16(void* stream, const float* fb, const int32_t* cells, int N, int L, int D, uint16_t* x1h);
int smin_moment_unit_fwd_x1h(void* stream, const float* fcmean, const float* fm, const float* fb, const int32_t* cells,
                             int N, int B, int L, int D, const float* Wcat, const float* bcat, float* mu, const uint16_t* x1h);
int smin_moment_unit_bwd_x1h(void* stream, const float* dmu, const float* fcmean, const float* fb, const int32_t* cells,
                             const int32_t* row_ptr, const int32_t* cellmap, int N, int B, int L, int D, const float* WcatT,
                             float* dfcmean, float* dfb, float* dWcat, float* dbcat, void* ws, size_t ws_bytes,
                             int all_valid, const float* dfcmean_acc, const uint16_t* x1h, const float* dfb_acc);

/* ---- Localization.forward (models.py:335-344): score heads.
 *   pm [B][L][L] dense, zero-filled outside the cell list;  wb [3][D], bb [3] = (ps, pe, pa) heads;
 *   psea [3][B][L];  lmask [B][L] fp32 0/1. */
int smin_score_map_fwd(void* stream, const float* fm, const float* fb, const int32_t* cells, int N, int B, int L, int D,
                       const float* wm, const float* bm, const float* wb, const float* bb, const float* lmask,
                       float* pm, float* psea);
/* dpm [B][L][L], dpsea [3][B][L] -> dfm [N][D], dfb [B][L][D], dwm [D], dbm [1], dwb [3][D], dbb [3].  Two independent halves
 * (map score: dpm -> dfm, dwm, dbm; boundary heads: dpsea -> dfb, dwb, dbb): dpm == NULL or dpsea == NULL skips one (two streams). */
int smin_score_map_bwd(void* stream, const float* dpm, const float* dpsea, const float* pm, const float* psea,
                       const float* fm, const float* fb, const int32_t* cells, int N, int B, int L, int D,
                       const float* wm, const float* wb, const float* lmask,
                       float* dfm, float* dfb, float* dwm, float* dbm, float* dwb, float* dbb, void* ws, size_t ws_bytes);

/* ---- Restated train-step loss (reference main.py:89-116 with reduction='none'):
 *   loss = L_m + L_s + L_e + 0.5 L_a, scaled BCE, masked, per-sample mean then batch mean.
 * pm/sm [B][L][L], ps/pe/pa/ss/se [B][L] fp32; ym/mm [B][L][L], ys/ye/ya/lm [B][L] as bytes (bool / uint8).
 * loss [1]; part [B][6] scratch kept for backward.  Backward writes dpm [B][L][L], dps/dpe/dpa [B][L]. */
int smin_loss_fwd(void* stream, const float* pm, const uint8_t* ym, const float* sm, const uint8_t* mm,
                  const float* ps, const uint8_t* ys, const float* ss, const float* pe, const uint8_t* ye, const float* se,
                  const float* pa, const uint8_t* ya, const uint8_t* lm, int B, int L, float* loss, float* part);
int smin_loss_bwd(void* stream, const float* dloss, const float* part,
                  const float* pm, const uint8_t* ym, const float* sm, const uint8_t* mm,
                  const float* ps, const uint8_t* ys, const float* ss, const float* pe, const uint8_t* ye, const float* se,
                  const float* pa, const uint8_t* ya, const uint8_t* lm, int B, int L,
                  float* dpm, float* dps, float* dpe, float* dpa);

/* ---- masks and training targets of a batch (reference dataset.py:95-155: AbstractDataset.get_iou, get_boundary_penalties,
 * get_snippet_label and the mask construction of __getitem__, per sample on the host there).
 * times [B][2] = ground-truth (start, end) seconds, duration [B], nfeats [B] sampled frames (clamped to T), qlen [B] query
 * words.  Outputs (bytes are 0/1): video_mask [B][T], query_mask [B][Nq] (NULL with qlen NULL: not wanted), length_mask [B][L],
 * moment_mask [B][L][L], sm [B][L][L] fp32, ym, ss / se [B][L] fp32, ys, ye, ya.  Requires L | T. */
int smin_build_targets(void* stream, const float* times, const float* duration, const int32_t* nfeats, const int32_t* qlen, int B, int T, int L, int Nq,
                       uint8_t* video_mask, uint8_t* query_mask, uint8_t* length_mask, uint8_t* moment_mask, float* sm, uint8_t* ym,
                       float* ss, uint8_t* ys, float* se, uint8_t* ye, uint8_t* ya,
                       const float* two_sigma_sq /* [B] or NULL: 2 sigma^2 of the boundary Gaussians as computed in double from the unrounded
                                                    annotation times (dataset.py:116-119); NULL: formed in double from the fp32 times */);

/* ---- compute_ious (reference utils.py:10-31): counts [8] = number of samples with a hit for
 * R@1 x IoU {0.1, 0.3, 0.5, 0.7} then R@5 x the same; ws [B][8] scratch.  Any L with L*L >= 5 (the reference's topk(5) needs as many). */
int smin_compute_ious(void* stream, const float* pm, const float* ps, const float* pe, const uint8_t* mm, const float* sm,
                      int B, int L, float* counts, float* ws);

/* ---- content stream (reference models.py:242-276 + 115-119, re-associated): the content unit's output
 *   f_c' = m*(cc Wc^T + bc) + f_c + hbar   (models.py:269-276)
 * is consumed only by the next unit's linear_c_hat (models.py:247) and, through mean_c, by the moment unit
 * (models.py:295).  Both are linear, so the (N*C) x D tensor f_c is never formed: with g = f Wch^T,
 *   chat_k = window_mean(g_k) + sum_{l<k} cc_l (Wch_k Wc_l)^T + (sum_{l<k} hbar_l) Wch_k^T + const
 *   mean_c f_c^k = mean_c f_c^{k-1} + (mean_c cc_k) Wc_k^T + bc_k + hbar_k
 * and every contraction over N*C rows has K, N <= dl.  The entry points below are the pieces; the Python host
 * (modules.py: SMIN.forward) composes them under autograd.  Requires a mask-driven cell list (m == 1 everywhere). */

/* out[s][n*C + c][:] = m * (mean over clip c of cell n of g[b][t][s*W:(s+1)*W] + bias[s*W:(s+1)*W])
 * g [B][T][nseg*W] (nseg <= 8: one segment per layer), out [nseg][N*C][W]; ws >= 8*B*(T+1)*nseg*W bytes. */
int smin_clip_window_means_fwd(void* stream, const float* g, const float* bias /* [bias_len], first features only */, int bias_len,
                               const int32_t* cells, int N, int B, int T, int L, int C,
                               int W, int nseg, float* out, void* ws, size_t ws_bytes);
/* dout: HOST array of nseg device pointers, dout[s] [N*C][W] -> dg [B][T][nseg*W]; ws >= 4*B*T*nseg*W bytes. */
int smin_clip_window_means_bwd(void* stream, const float* const* dout, const int32_t* cells, const int32_t* row_ptr,
                               const int32_t* cellmap, int N, int B, int T, int L, int C, int W, int nseg, float* dg,
                               void* ws, size_t ws_bytes, const int32_t* ev_offsets, const void* ev_table);

/* attention core of the content unit alone (models.py:252-267): chat [N*C][dl] -> cc [N*C][dl] and/or ccmean [N][dl]
 * (either output may be NULL). */
int smin_content_attn_fwd(void* stream, const float* chat, const int32_t* cells, const int32_t* row_ptr,
                          int N, int B, int L, int C, int dl, int Nq,
                          const float* Mq, const float* uq, const float* what, const float* shat, const float* qmask,
                          float* cc, float* ccmean);
/* smin_content_attn_fwd with the rows stored as bf16 (cc_h [N*C][dl], round to nearest even) beside the fp32 clip mean */
int smin_content_attn_fwd_cch(void* stream, const float* chat, const int32_t* cells, const int32_t* row_ptr,
                              int N, int B, int L, int C, int dl, int Nq,
                              const float* Mq, const float* uq, const float* what, const float* shat, const float* qmask,
                              uint16_t* cc_h, float* ccmean);
size_t smin_content_attn_bwd_workspace_bytes(int N, int B, int C, int dl);
/* dcc [N*C][dl] and/or dccmean [N][dl] (one may be NULL) -> dchat [N*C][dl], dMq, duq, dwhat, dshat. */
int smin_content_attn_bwd(void* stream, const float* dcc, const float* dccmean, const float* chat,
                          const int32_t* cells, const int32_t* row_ptr, int N, int B, int L, int C, int dl, int Nq,
                          const float* Mq, const float* uq, const float* what, const float* shat, const float* qmask,
                          float* dchat, float* dMq, float* duq, float* dwhat, float* dshat, void* ws, size_t ws_bytes);

/* y[r][:] = [x_0[r] | .. | x_{nseg-1}[r]] W^T + bias + add_rows[r][:] + add_cells[r / C][:]
 * xs: HOST array of nseg <= 4 device pointers, x_s [R][K]; W [O][nseg*K]; y [R][O]; bias, add_rows, add_cells may
 * be NULL.  Backward: dx_s = dy W_s (dxs: HOST array of nseg device pointers, or NULL; WT = W^T [nseg*K][O]),
 * dW [O][nseg*K] = dy^T [x_0 | ..] (NULL, with dbias NULL, to skip the weight half), dbias = colsum(dy) (may be NULL);
 * the gradient of add_rows is dy itself, of add_cells smin_group_sum(dy). */
int smin_linear_rows_fwd(void* stream, const float* const* xs, int nseg, const float* W, const float* bias, const float* add_rows,
                         const float* add_cells, int C, int R, int O, int K, float* y);
size_t smin_linear_rows_bwd_workspace_bytes(int R, int O, int Ktot);
int smin_linear_rows_bwd(void* stream, const float* dy, const float* const* xs, int nseg, const float* WT, int R, int O, int K,
                         float* const* dxs, float* dW, float* dbias, void* ws, size_t ws_bytes);
/* smin_linear_rows_fwd / the weights half of smin_linear_rows_bwd for xs stored as bf16 (uint16_t bit patterns; contraction-only operands
 * under smin_set_gemm_mode(2), where the results equal the fp32-storage calls bit for bit).  Forward: add_rows and add_cells required. */
int smin_linear_rows_fwd_xh(void* stream, const uint16_t* const* xs, int nseg, const float* W, const float* bias, const float* add_rows,
                            const float* add_cells, int C, int R, int O, int K, float* y);
int smin_linear_rows_bwd_xh(void* stream, const float* dy, const uint16_t* const* xs, int nseg, int R, int O, int K,
                            float* dW, float* dbias, void* ws, size_t ws_bytes);
/* dx_s += dy W_s for every segment (WT as above): input gradients accumulated into tensors that already hold another consumer's gradient */
int smin_linear_rows_dx_acc(void* stream, const float* dy, int nseg, const float* WT, int R, int O, int K, float* const* dxs);
/* out[g][:] = sum_{c<C} x[g*C + c][:]   (x [groups*C][W]) */
int smin_group_sum(void* stream, const float* x, int groups, int C, int W, float* out);

/* ---- word-side operands of every layer's content attention in one launch (ContentUnit.forward models.py:249-251 and the
 * word half of ContentAttention.forward models.py:209-211, with W_q folded onto the words).  Per layer k:
 *   what_k = (f_w WH_k^T + bWH_k) * qmask,  shat_k = f_s SH_k^T + bSH_k,  kb_k = what_k AK_k^T + bAK_k,  Mq_k = kb_k AQ_k,  uq_k = kb_k . bAQ_k
 * params: HOST array of nl*8 device pointers, per layer [WH (dl,D), bWH, SH (dl,D), bSH, AK (dl,dl), bAK, AQ (dl,dl), bAQ]
 * (= linear_w_hat, linear_s_hat, attn_layer.W_k, attn_layer.W_q).  fw [B][Nq][D], fs [B][D], qmask [B][Nq] fp32.
 * Outputs, layer-major: what, kb, Mq [nl][B][Nq][dl], shat [nl][B][dl], uq [nl][B][Nq].  nl <= 8, Nq <= 32, dl <= 128. */
int smin_word_prep_fwd(void* stream, const float* fw, const float* fs, const float* qmask, const float* const* params, int nl, int B, int Nq, int D,
                       int dl, float* what, float* shat, float* kb, float* Mq, float* uq);
size_t smin_word_prep_bwd_workspace_bytes(int nl, int B, int Nq, int D, int dl);
/* dwhat / dshat / dMq / duq: HOST arrays of nl device pointers (per-layer gradients [B][..]; NULL entries = zero).
 * -> dfw [B][Nq][D], dfs [B][D], dparams: HOST array of nl*8 device pointers (same order as params), all written. */
int smin_word_prep_bwd(void* stream, const float* const* dwhat, const float* const* dshat, const float* const* dMq, const float* const* duq,
                       const float* fw, const float* fs, const float* qmask, const float* what, const float* kb, const float* const* params,
                       int nl, int B, int Nq, int D, int dl, float* dfw, float* dfs, float* const* dparams, void* ws, size_t ws_bytes);

/* ---- VideoEncoder (models.py:25-36) fused with the backbone's Hadamard product (models.py:81-83):
 *   fv[b][t][:] = (x[b][t][:] W^T + bias + pe[t][:]) * vmask[b][t]     f[b][t][:] = fv[b][t][:] * fs[b][:]
 * x [B*T][Din], W [D][Din], pe [>=T][D] (rows 0..T-1 are used), vmask [B*T] fp32, fs [B][D]; outputs fv, f [B*T][D].
 * fs == f == NULL: the projection alone (fv); smin_video_encoder_gate then forms f = fv * fs -- a host can run the projection
 * beside the query encoder (which produces fs) and pay only the product behind it. */
int smin_video_encoder_fwd(void* stream, const float* x, const float* W, const float* bias, const float* pe, const float* vmask,
                           const float* fs, int B, int T, int Din, int D, float* fv, float* f);
int smin_video_encoder_gate(void* stream, const float* fv, const float* fs, int B, int T, int D, float* f);
size_t smin_video_encoder_bwd_workspace_bytes(int B, int T, int Din, int D);
/* df [B*T][D] -> dW [D][Din], dbias [D], dpe [T][D], dfs [B][D].  May be issued as two calls on the same ws (e.g. on two streams, the
 * second ordered behind the first): dW == NULL -> inputs half (dfs; the masked gradient stays in ws); df == NULL -> weights half. */
int smin_video_encoder_bwd(void* stream, const float* df, const float* fv, const float* fs, const float* vmask, const float* x,
                           int B, int T, int Din, int D, float* dW, float* dbias, float* dpe, float* dfs, void* ws, size_t ws_bytes);

/* ---- QueryEncoder's bidirectional LSTM layer (models.py:38-64: nn.LSTM over a packed, padded batch), one layer per
 * call, both directions.  X [B*Nq][In]; Wih_cat [8H][In] = [W_ih; W_ih_reverse]; bias_cat [8H] = b_ih + b_hh per
 * direction; W4 [2][H][H][4] with W4[d][k][u][g] = W_hh_d[g*H+u][k]; len [B] valid lengths (device).  Outputs: Hout
 * [B][Nq][2H] (zero at padded positions, as pad_packed_sequence gives), and for backward G [B][Nq][2][4H] (gate
 * activations i,f,g,o) and Cs [B][Nq][2][H] (cell states).  Requires In % 4 == 0, H % 4 == 0, H <= 256. */
/* sentence feature (models.py:60-62): fs [B][2H], fs[b] = [fw[b][len_b - 1][0:H] | fw[b][0][H:2H]] with fw [B][Nq][2H] (len clamped to
 * 1..Nq); backward adds dfs into the same entries of dfw (in place) */
int smin_sentence_feature_fwd(void* stream, const float* fw, const int32_t* len, int B, int Nq, int H, float* fs);
int smin_sentence_feature_bwd(void* stream, const float* dfs, const int32_t* len, int B, int Nq, int H, float* dfw);
/* the operand layouts above from nn.LSTM's eight parameter tensors of a layer (w: HOST array of 8 device pointers: weight_ih, weight_hh,
 * bias_ih, bias_hh of the forward direction, then of the reverse direction), one launch; also writes Whh [2][4H][H] for the backward call */
int smin_lstm_pack(void* stream, const float* const* w, int In, int H, float* Wih, float* bias, float* Whh, float* W4);
/* every layer of the encoder (nlayers <= 4) in one launch, ahead of the first recurrence: w = HOST array of 8 * nlayers device pointers
 * (layer by layer, each in smin_lstm_pack's order), In = HOST array of the layers' input widths, outputs = HOST arrays of nlayers pointers */
int smin_lstm_pack_layers(void* stream, int nlayers, const float* const* w, const int* In, int H, float* const* Wih, float* const* bias,
                          float* const* Whh, float* const* W4);
int smin_bilstm_layer_fwd(void* stream, const float* X, const float* Wih_cat, const float* bias_cat, const float* W4,
                          const int32_t* len, int B, int Nq, int In, int H, float* G, float* Hout, float* Cs);
size_t smin_bilstm_layer_bwd_workspace_bytes(int B, int Nq, int In, int H);
/* dHout [B][Nq][2H] -> dX [B*Nq][In] (NULL to skip), dWih_cat [8H][In], dbias_cat [8H] (= d b_ih = d b_hh),
 * dWhh [2][4H][H].  Wih_catT [In][8H]; Whh [2][4H][H] (W_hh per direction, as nn.LSTM stores it).  May be issued as two calls on the
 * same ws: dWih_cat == NULL -> inputs half (the recurrence and dX; gate gradients stay in ws); dHout == NULL -> weights half. */
int smin_bilstm_layer_bwd(void* stream, const float* dHout, const float* X, const float* Hout, const float* G, const float* Cs,
                          const float* Wih_catT, const float* Whh, const int32_t* len, int B, int Nq, int In, int H,
                          float* dX, float* dWih_cat, float* dbias_cat, float* dWhh, void* ws, size_t ws_bytes);
/* The weights half in up to three independent pieces, each with its own part of ws (which: bit 0 = dWih_cat + dbias_cat, bit 1 =
 * dWhh[0], bit 2 = dWhh[1]; 7 = what smin_bilstm_layer_bwd's weights half does): pieces issued on different streams run side by
 * side behind the inputs half (they are the last kernels of a train step, models.py:46-62 backward).
 * dbias_cat2 (NULL to skip): a second copy of dbias_cat -- b_ih and b_hh have the same gradient and each needs memory of its own. */
int smin_bilstm_layer_bwd_weights(void* stream, int which, const float* X, const float* Hout, int B, int Nq, int In, int H,
                                  float* dWih_cat, float* dbias_cat, float* dbias_cat2, float* dWhh, void* ws, size_t ws_bytes);

/* The recurrences above run, for H a multiple of 32, as clusters of workgroups that keep W_hh in LDS and exchange h / dh through
 * tagged granules in global memory (csrc/bilstm_cluster.hip); every poll there is bounded.  Returns 1 once a poll has expired since
 * the last call (the launch that hit it produced wrong values), 0 otherwise; clears the word.  Synchronises the device. */
int smin_lstm_cluster_error(void);

/* ---- packed valid-cell layout from a (B, L, L) mask (uint8 / bool, non-zero = valid).  all_cells = 0: list the valid
 * cells (m = 1); 1: list every (b, i, j) with m = mask.  cells [N][4] = {b, i, j, m} sorted by (b, i, j); row_ptr
 * [B*L + 1]; cellmap [B][L][L] = cell id or -1.  The caller sizes cells from the count of listed cells. */
int smin_build_cells(void* stream, const uint8_t* mask, int B, int L, int all_cells,
                     int32_t* cells, int32_t* row_ptr, int32_t* cellmap);
/* The same for a caller that already knows the number of listed cells (a captured step replays with the count it was captured
 * for: no device -> host round trip inside the step).  cells holds exactly n_expected entries; nothing is written past it, and
 * *status (device int32, cleared once by the caller) is set to 1 if the mask lists a different number of cells. */
int smin_build_cells_n(void* stream, const uint8_t* mask, int B, int L, int all_cells, int n_expected,
                       int32_t* cells, int32_t* row_ptr, int32_t* cellmap, int32_t* status);

/* What the train step derives from its masks and the boundary heads' parameters before its first real kernel, in one launch
 * (main.py:141-160 hands the batch over as it comes from the dataset: byte masks): len32 [B] = words per query, qmf [B][Nq] /
 * vmaskf [B][T] / lmf [B][L] = the masks as fp32, *count (device int64) = number of set cells of moment_mask [B][L][L],
 * wb [3][D] / bb [3] = weights and biases of the start / end / "all" heads side by side (Localization, models.py:318-333; w, b:
 * HOST arrays of three device pointers).  Masks: one byte per element.  acc: 16 bytes of device memory, zero on entry and on exit. */
int smin_step_prologue(void* stream, const uint8_t* query_mask, const uint8_t* video_mask, const uint8_t* length_mask, const uint8_t* moment_mask,
                       const float* const* w, const float* const* b, int B, int Nq, int T, int L, int D, int32_t* len32, float* qmf, float* vmaskf,
                       float* lmf, float* wb, float* bb, int64_t* count, void* acc);

/* ---- layout helpers: dense (B,L,L,W) <-> packed [N][W] rows (W floats per cell). */
int smin_pack_cells(void* stream, const float* dense, const int32_t* cells, int N, int L, int W, float* packed);
int smin_unpack_cells(void* stream, const float* packed, const int32_t* cells, int N, int L, int W, float* dense /* pre-zeroed */);

/* ---- parameter-only operands of the content stream and the moment unit, every layer in one launch per direction
 * (csrc/param_prep.hip).  params: HOST array of nl*8 device pointers, per layer [Wch (dl,D) = linear_c_hat.weight, bch, Wc (D,dl) =
 * linear_c.weight, bc, Wfb (D,D) = conv_layer_fb.weight, bfb, Wfc (D,D) = conv_layer_fc.weight, bfc]  (models.py:247, 269, 288-303).
 *   Pcat: HOST array of nl*2 device pointers; entry [k*2 + part] (k > 4*part) is [dl][nseg*dl], nseg = min(4, k - 4 part), column block
 *         s = Wch_k Wc_{4 part + s};   consts [nl][dl] = bch_k + Wch_k (bc_0 + .. + bc_{k-1});   Wcat [nl][D][2D] = [Wfb_k | Wfc_k];
 *         bcat [nl][D] = bfb_k + bfc_k;   Wch_all [nl*dl][D].       nl <= 8, D % 32 == 0, D <= 1056, dl % 32 == 0.
 * Backward: dPcat (as Pcat), dconsts, dWcat, dbcat, dWch_all; *_base: HOST arrays of nl device pointers with the gradient a parameter
 * already has from its direct uses (dWch_base[k] may be NULL); grads: HOST array of nl*8 device pointers, every entry written. */
int smin_param_prep_fwd(void* stream, const float* const* params, int nl, int D, int dl, float* const* Pcat, float* consts, float* Wcat,
                        float* bcat, float* Wch_all);
int smin_param_prep_bwd(void* stream, const float* const* params, int nl, int D, int dl, const float* const* dPcat, const float* dconsts,
                        const float* dWcat, const float* dbcat, const float* dWch_all, const float* const* dWch_base,
                        const float* const* dWc_base, const float* const* dbc_base, float* const* grads);

/* ---- host helpers of the fused step: many small tensors in one launch.  src/dst/rows/cols/srcs are HOST arrays.
 *   smin_transpose_batch: dst[m] [cols][rows] = src[m]^T for n <= SMIN_BATCH_MAX row-major matrices src[m] [rows][cols]
 *     (the W^T operands of the input-gradient contractions; the reference transposes inside autograd, models.py every nn.Linear);
 *   smin_sum_lists: out[i] = sum_k srcs[k][i] in list order (gradients of f_s / f_w, which every layer consumes). */
#define SMIN_BATCH_MAX 32
int smin_transpose_batch(void* stream, const float* const* src, float* const* dst, const int32_t* rows, const int32_t* cols, int n);
int smin_sum_lists(void* stream, const float* const* srcs, int n, size_t numel, float* out);
/* out[c] = sum_r x[r][c]  (x [R][W], W % 4 == 0, W <= 1024): the bias gradient of a linear map whose weight gradient has no pass of its own
 * to ride on (content stream: layer 0's constant).  Two fixed-order stages. */
size_t smin_col_sum_workspace_bytes(int R, int W);
int smin_col_sum(void* stream, const float* x, int R, int W, float* out, void* ws, size_t ws_bytes);

/* ---- stand-alone fp32 MFMA GEMM  C[M][N] = A[M][K] * B[N][K]^T  (used by tests and bench.py's
 * roofline probe; same engine as every contraction above). */
int smin_gemm_nt(void* stream, const float* A, const float* Bm, float* Cm, int M, int N, int K);
/* C += A * B^T through the engine's accumulate epilogue (tests) */
int smin_gemm_nt_acc(void* stream, const float* A, const float* Bm, float* Cm, int M, int N, int K);

#ifdef __cplusplus
}
#endif
#endif /* SMIN_HIP_H */
