/*
 * gssgcn.h -- C ABI of libgssgcn.so, the MI355X (gfx950) implementation of the GSS-GCN
 * training hot path of bowang-lab/gcn-drug-repurposing.
 *
 * The reference has no native code: its hot path is a sequence of torch calls.  Each entry
 * point below replaces the torch call(s) named in its comment (file:line relative to the
 * reference root).  A reference maintainer binds these with ctypes.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller unless its name starts with h_
 *   - matrices are dense row-major fp32 [rows][d]; CSR uses int32 rowptr/col and fp32 values
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); calls are
 *     asynchronous on it and never synchronise the device
 *   - return 0 on success, a negative GSS_E* code on failure; gss_last_error() gives the text
 *   - d (feature width == --hidden-units, modules/model.py:142) must be a multiple of 16, <= 1024
 */
#ifndef GSSGCN_H
#define GSSGCN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GSS_ABI_VERSION 6   /* 6 (round 6): gss_source_hash; the access-shape knobs whose sweeps said "default holds" twice are gone; 5 (round 5): gss_rowsum_check, gss_plan_sync_stats, gss_comm_local_mode / gss_comm_local_log, gss_csr_giant_rows; 4 (round 4): gss_shard_desc gained a_loc_t, gss_plan_comm_stats, gss_knn_topk_rows */

#define GSS_OK 0
#define GSS_EINVAL (-22)   /* bad argument (shape, null pointer, unsupported d) */
#define GSS_ENOMEM (-12)   /* hipMalloc failed */
#define GSS_EHIP (-5)      /* a HIP call or kernel launch failed */
#define GSS_ENOTCONV (-34)  /* gss_ppr_run: max_iter reached */
#define GSS_ECOMM (-104)    /* a collective failed or the communicator was aborted (RCCL error, dead peer) */
#define GSS_ETIMEOUT (-110) /* gss_comm_sync: the stream did not drain before the deadline; the communicator was aborted */

typedef struct gss_csr gss_csr;   /* a CSR operand plus its launch schedule (row bins) */
typedef struct gss_plan gss_plan; /* activations + workspace of one training replica/shard */
typedef struct gss_comm gss_comm; /* the communicator of a node-range sharded job (RCCL over xGMI, or in-process ranks) */

int gss_abi_version(void);
/* sha256 (lower-case hex) of a source file the library was built from -- "spmm.hip", "dense.hip", "common.h", "gssgcn.h" ... --, of all of
 * them concatenated in name order ("*" or NULL), or NULL for any other name.  Evidence under profiles/ records the hashes of the kernels it
 * measured; bench.py drops a counter file whose recorded spmm.hip hash is not this library's, __graft_entry__.build() rebuilds a library
 * whose hashes are not the tree's (a stale .so with fresh timestamps cannot pass for a current one). */
const char *gss_source_hash(const char *file);
/* Diagnostic builds of a measurement only (tools/gemm_stamps.py): while a device buffer is set, every wave of a projection launch
 * (gss_dense_fwd) stores its wall-clock stamps {start, loop begin, loop end, end} + {workgroup id, HW_ID} there (6 x 8 bytes per wave).
 * NULL (the default) switches it off; no production path sets it. */
int gss_debug_set_stamp_buffer(void *device_buffer);
/* measurement aid (tools/ab_live.py): changes one KERNEL-SELECTION knob ("gemm_variant", "gemm_ws", "spmm_slices", "spmm_pin", "spmm_list_blocks") in a live
 * plan's snapshot, so that one plan -- the same buffers at the same addresses --
 * can be timed under alternating settings; knobs that size a workspace or steer the plan's bookkeeping are refused (GSS_EINVAL).
 * Not thread-safe against gss_debug_set_option on another thread. */
int gss_plan_debug_set_option(gss_plan *plan, const char *name, int value);
const char *gss_last_error(void);

/* ---- K11  preprocess_graph, helpers/helper.py:82-89 (+ fp32 cast helper.py:95) ----------
 * In: CSR of (A + I) with fp64 values (diagonal already inserted, columns sorted).
 * Out: val_out[e] = (float)(dinv[row] * val[e] * dinv[col[e]]), dinv = rowsum^-1/2 in fp64;
 *      rowsum_out (nullable) = fp64 row sums D_ii.  Row sums <= 0 give NaN/inf like the reference. */
int gss_normalize_adj(int32_t n, const int32_t *rowptr, const int32_t *col, const double *val,
                      float *val_out, double *rowsum_out, void *stream);

/* The same for one node-range shard: the shard holds rows [row0, row0 + n) of A + I -- or, transposed != 0, of
 * (A + I)^T -- with GLOBAL column ids.  gss_rowsum_dinv: D_ii and D_ii^-1/2 of the shard's rows of A + I;
 * gss_scale_adj_shard: the normalised values from the all-gathered D^-1/2 of every node, with the rounding sequence of
 * gss_normalize_adj, so a shard's values are bit-identical to the matching entries of the single-GPU A_hat / A_hat^T. */
int gss_rowsum_dinv(int32_t n, const int32_t *rowptr, const double *val, double *dinv_out, double *rowsum_out, void *stream);
int gss_scale_adj_shard(int32_t n, int32_t row0, const int32_t *rowptr, const int32_t *col, const double *val, const double *dinv_global,
                        int32_t transposed, float *val_out, void *stream);
/* The guard the reference lacks (helpers/helper.py:85, SURVEY a3's hazard): counts the rows whose D_ii is not > 0 -- negative
 * similarities in kNN mode can do that -- i.e. the rows whose D_ii^-1/2 is NaN / inf and poisons every entry of their row and column.
 * rowsum = the fp64 row sums gss_normalize_adj / gss_rowsum_dinv wrote (device); *h_count_out and *h_first_out (nullable; the lowest
 * such row, -1 if none) are HOST values: the call waits for the stream.  The normalisation itself stays the reference's arithmetic; what
 * to do about a positive count is the caller's decision (trainer.py refuses to train unless --allow-nan). */
int gss_rowsum_check(int32_t n, const double *rowsum, int64_t *h_count_out, int32_t *h_first_out, void *stream);

/* ---- CSR handle ---------------------------------------------------------------------------
 * Borrows d_rowptr/d_col/d_val (caller keeps them alive).  h_rowptr is a HOST copy of rowptr used
 * once to bin rows by length (long rows get a whole workgroup).  n_cols is the height of the dense
 * operand (== n_rows on one GPU, the global N for a row shard). */
int gss_csr_create(gss_csr **out, int32_t n_rows, int32_t n_cols, int64_t nnz, const int32_t *h_rowptr,
                   const int32_t *d_rowptr, const int32_t *d_col, const float *d_val);
void gss_csr_destroy(gss_csr *a);
/* Optional, for graphs whose nodes were relabelled hub-first (descending degree): declare which rows of the dense operand
 * belong to the hubs -- rows [0, own_hot) and [halo_begin, halo_end) (the second range is for a shard, whose operand holds
 * the other shards' boundary rows behind its own).  Where the operand is far larger than the caches the SpMM then fetches
 * every other row with the non-temporal policy, so that once-read rows do not evict the hubs' rows.  own_hot = -1: no split
 * (the default).  Speed only: results are unchanged. */
int gss_csr_set_hot(gss_csr *a, int32_t own_hot, int32_t halo_begin, int32_t halo_end);
/* Rows with more stored entries than knob "spmm_giant" (default 32,768; 0 = never) are not one workgroup's job: the dense products cut
 * them into chunks of a quarter of that, sum the chunks in a pass of their own and add a row's chunks in order in a finish pass that runs
 * the product's epilogue (three launches of the same kernel; results agree with the unchunked sum to rounding, rows below the threshold
 * keep their bits).  The hubs of a 10^7-node scale-free graph hold ~3 x 10^5 entries: as one workgroup's job such a row alone outlasts
 * the rest of a shard's launch.  A handle looks at the knob with its first product (or with this call): n_rows_out = its giant rows,
 * n_chunks_out = their chunks. */
int gss_csr_giant_rows(const gss_csr *a, int32_t *n_rows_out, int32_t *n_chunks_out);

/* ---- K1/K2  torch.sparse.mm + torch.mul, modules/model.py:163,168-169 -----------------------
 * y = A x  (x: [n_cols][d], y: [n_rows][d]).  If m != NULL also m = y (.) h, h: [n_rows][d]
 * (the Hadamard of model.py:168 fused into the epilogue of the first SpMM). */
int gss_spmm(const gss_csr *a, int32_t d, const float *x, float *y, const float *h, float *m, void *stream);
/* The second pass of a two-pass product: y = y_in + A x (then m = y (.) h as above).  The entries of the rows may be split over two
 * CSR handles of the same rows (e.g. a shard's own-column and boundary-column entries, gss_shard_desc): gss_spmm with the first,
 * gss_spmm_add with the second.  y_in may be y itself. */
int gss_spmm_add(const gss_csr *a, int32_t d, const float *x, const float *y_in, float *y, const float *h, float *m, void *stream);

/* backward SpMMs (autograd of model.py:163-169 for layers >= 2), A here is CSR(A_hat^T):
 *   gss_spmm_bwd1: dm = A g_am;  u = g_ax + dm (.) x_in;  t = dm (.) ax
 *   gss_spmm_bwd2: gx = t + A u; dp = c * gx (.) elu'(p) (+ res if res != NULL); gx_out nullable */
int gss_spmm_bwd1(const gss_csr *at, int32_t d, const float *g_am, const float *g_ax, const float *x_in,
                  const float *ax, float *u, float *t, void *stream);
int gss_spmm_bwd2(const gss_csr *at, int32_t d, const float *u, const float *t, const float *p, float c,
                  const float *res, float *dp, float *gx_out, void *stream);
/* gss_spmm_bwd1 for row-sparse gradients (the top layer: dLoss/dE is non-zero on the B batch rows only):
 * g_am_b / g_ax_b are compact [B][d]; pos_col[c] is the compact row of column id c (or -1), pos_row[r] the
 * compact row of output row r (or -1).  Neighbours with pos_col < 0 are skipped. */
int gss_spmm_bwd1_sparse(const gss_csr *at, int32_t d, const float *g_am_b, const float *g_ax_b,
                         const int32_t *pos_col, const int32_t *pos_row, const float *x_in, const float *ax,
                         float *u, float *t, void *stream);

/* ---- K3/K4  nn.Linear x2 + add + F.elu + residual, modules/model.py:165,170-173,201-203 -------
 * p = ax W1^T + b1 + am W2^T + b2;  o = elu(p);  x_next = p_prev ? p_prev + decay*o : o.
 * W1, W2: [d][d] row-major ([out][in], the nn.Linear layout).  fp32 MFMA (v_mfma_f32_16x16x4_f32). */
int gss_dense_fwd(int32_t n, int32_t d, const float *ax, const float *am, const float *w1, const float *b1,
                  const float *w2, const float *b2, const float *p_prev, float decay, float *p, float *x_next,
                  void *stream);

/* input gradients of the two Linear layers: g_ax = dp W1, g_am = dp W2 (w1t/w2t are the TRANSPOSED
 * weights, [in][out]).  If rows != NULL, row r of the compact [n][d] input is written to row rows[r]
 * of the outputs (scatter of batch-row gradients into zeroed [N][d] buffers). */
int gss_dense_bwd_input(int32_t n, int32_t d, const float *dp, const float *w1t, const float *w2t,
                        const int32_t *rows, float *g_ax, float *g_am, void *stream);

/* weight gradients: gw1 (+)= dp^T ax, gw2 (+)= dp^T am, gb (+)= colsum(dp), fixed-order two-stage
 * reduction (bitwise reproducible).  rows != NULL: dp is compact [n][d] and ax/am rows are gathered
 * at rows[r].  accumulate != 0 adds to the existing gw1/gw2/gb.  ws: workspace of
 * gss_wgrad_workspace_bytes(n, d) bytes. */
size_t gss_wgrad_workspace_bytes(int32_t n, int32_t d);
int gss_dense_bwd_weight(int32_t n, int32_t d, const float *dp, const float *ax, const float *am,
                         const int32_t *rows, float *gw1, float *gw2, float *gb, int accumulate, void *ws,
                         void *stream);

/* ---- K5  F.normalize(x, dim=1), modules/model.py:205 (eps 1e-12) -----------------------------
 * e = x / max(||x||, eps); inv_den[i] = 1 / max(||x_i||, eps) is kept for the backward pass. */
int gss_rownorm_fwd(int32_t n, int32_t d, const float *x, float *e, float *inv_den, void *stream);

/* ---- K6/K7  GSS_loss.gss_loss + its autograd, modules/model.py:214-221 ------------------------
 * E_B = e[idx]; S = E_B E_B^T; loss = mean(-alpha/2 (relu(S) - beta)^2) written to loss_out[0];
 * de_b[b] = dLoss/dE_B[b] = sum_j (G_bj + G_jb) E_B[j], G = -alpha/B^2 (relu(S)-beta) 1[S>0].
 * S is never materialised.  idx: int32[B], unique.  ws: gss_loss_workspace_bytes(B, d) bytes. */
size_t gss_loss_workspace_bytes(int32_t b, int32_t d);
/* The size is NOT monotone in B (fewer row tiles get more column slabs: B = 2032 needs more than B = 2048): a caller that keeps one
 * workspace for batches of up to b_max rows -- e.g. the shorter last batch of an epoch -- sizes it with this. */
size_t gss_loss_workspace_bytes_max(int32_t b_max, int32_t d);
int gss_loss_fwd_bwd(int32_t n, int32_t d, const float *e, const int32_t *idx, int32_t b, float beta,
                     float alpha, float *loss_out, float *de_b, void *ws, void *stream);

/* backward of K5 + K4 on the batch rows: dx_b = (de_b - e_b (e_b . de_b)) * inv_den[idx];
 * dp_b = c * dx_b (.) elu'(p[idx]).  All [B][d] compact. */
int gss_rownorm_elu_bwd(int32_t d, const float *de_b, const int32_t *idx, int32_t b, const float *e,
                        const float *inv_den, const float *p, float c, float *dx_b, float *dp_b, void *stream);

/* dst[rows[r]] += src[r] for r < b (rows unique; a negative row is skipped -- a batch row another shard owns) */
int gss_scatter_add_rows(int32_t d, const float *src, const int32_t *rows, int32_t b, float *dst, void *stream);

/* ---- C1-C3  collectives of the node-range sharded trainer (the reference is single-device,
 * train.py:68,118-122, so there is nothing to match -- these are the exchange points 1-D sharding of its step needs).
 * One process per GPU: rank 0 calls gss_comm_unique_id, the host distributes the GSS_COMM_ID_BYTES bytes (torch.distributed
 * store, a file ...), every rank calls gss_comm_create_rccl with its current HIP device set.  gss_comm_create_local makes
 * `world` communicators for ranks that are THREADS of one process (each with its own stream, possibly all on one GPU):
 * device-to-device copies behind a timed host barrier, so that a sharded plan can be run at world 2..8 on a one-GPU box.
 * Collectives are enqueued on `stream` (the local backend also synchronises it); every rank must call them in the same order. */
#define GSS_COMM_ID_BYTES 128
int gss_comm_unique_id(void *id_out);
int gss_comm_create_rccl(gss_comm **out, int32_t world, int32_t rank, const void *id);
int gss_comm_create_local(gss_comm **out /* [world] */, int32_t world);
/* Measurement aid of the in-process backend (tools/shard_emulation.py --serial): rank threads share ONE GPU, so a step timed with all
 * of them running says nothing about one rank's kernels.  mode 1 = record: keep a device copy of what every collective DELIVERS to this
 * rank, in call order; mode 2 = replay: serve those copies again (same call sequence and sizes, checked) by a device-to-device copy --
 * no peers, no barrier, no host wait -- so that one rank's step can run alone on the GPU and be timed; mode 0 (default) = normal, frees
 * the log.  gss_comm_local_log: the delivered bytes of every recorded collective (n_out = how many there are; at most cap are written). */
int gss_comm_local_mode(gss_comm *c, int32_t mode);
int gss_comm_local_log(gss_comm *c, int64_t *bytes_out, int32_t cap, int32_t *n_out);
/* A third backend for boxes where RCCL cannot run the job: one PROCESS per rank as with RCCL, but every collective is staged through
 * pinned host memory and handed to a transport callback of the host language (the Python package passes torch.distributed's gloo;
 * GSS_COMM_BACKEND=host).  RCCL refuses two ranks on one device, so this is how `train.py --ngpus N` / `bench.py --gpus N` run as N
 * real processes on a ONE-GPU test box (all ranks on that GPU).  The callback receives pinned host buffers:
 *   GSS_HOST_ALLGATHER: `count` bytes at send from every rank, rank r's at recv + r * count;
 *   GSS_HOST_ALLTOALLV: bytes [send_off[q], send_off[q+1]) of send go to rank q, the bytes rank q sends land at [recv_off[q], recv_off[q+1])
 *   of recv (byte offsets, world + 1 entries each).
 * (the own range is empty, as in gss_exchange_rows).  It returns 0 on success.  Sums (gss_allreduce_sum) are taken on the host in rank order: the same bits on every rank. */
enum { GSS_HOST_ALLGATHER = 0, GSS_HOST_ALLTOALLV = 1 };
typedef int (*gss_host_xfer_fn)(void *user, int kind, const void *send, const int64_t *send_off, void *recv, const int64_t *recv_off, int64_t count);
int gss_comm_create_host(gss_comm **out, int32_t world, int32_t rank, gss_host_xfer_fn fn, void *user);
void gss_comm_destroy(gss_comm *c);
/* Make every rank that is (or will be) blocked in a collective of this group return an error -- call it from a rank that failed.
 * In-process backend: releases the host barrier (the peers do not wait for its 120 s timeout).  RCCL: ncclCommAbort -- kernels
 * already enqueued stop waiting for the peer, the communicator is unusable afterwards (every later call returns GSS_ECOMM). */
void gss_comm_abort(gss_comm *c);
/* Failure detection (the reference is single-process and has none).  Every collective entry point polls
 * ncclCommGetAsyncError after enqueueing and, on an error, aborts the communicator and returns GSS_ECOMM; gss_comm_check is the same
 * poll on its own.  gss_comm_sync waits until `stream` has drained (hipStreamQuery + the same poll, no busy device wait) and, if
 * that takes longer than timeout_s (> 0) -- a peer that stopped calling, mismatched collectives --, aborts the communicator and
 * returns GSS_ETIMEOUT instead of hanging; callers then exit non-zero.  gss_comm_count: the number of ranks the backend itself
 * reports (ncclCommCount) -- bench.py prints it as `rccl_ranks`. */
int gss_comm_check(gss_comm *c);
int gss_comm_count(gss_comm *c, int32_t *count_out);
int gss_comm_sync(gss_comm *c, void *stream, double timeout_s);
int32_t gss_comm_world(const gss_comm *c);
int32_t gss_comm_rank(const gss_comm *c);
/* C1: src [max_rows][d] (this rank's rows first, the rest don't-care) -> dst_padded [world * max_rows][d], rank r's rows at
 * row r * max_rows.  In place when src == dst_padded + rank * max_rows * d. */
int gss_allgather_rows(gss_comm *c, int32_t d, int32_t max_rows, const float *src, float *dst_padded, void *stream);
int gss_allgather_bytes(gss_comm *c, const void *src, void *dst, size_t bytes_per_rank, void *stream);
/* C2/C3: buf <- sum over ranks, in place, the same bits on every rank */
int gss_allreduce_sum(gss_comm *c, float *buf, int64_t count, void *stream);
/* C1, boundary form: rows [h_send_off[q], h_send_off[q+1]) of `send` ([..][d]) go to rank q; the rows rank q sends land at
 * rows [h_recv_off[q], h_recv_off[q+1]) of `recv`.  Offsets are HOST arrays of world + 1 entries (own range empty); one
 * fused group of ncclSend / ncclRecv, pairs with nothing to exchange are skipped. */
int gss_exchange_rows(gss_comm *c, int32_t d, const float *send, const int64_t *h_send_off, float *recv, const int64_t *h_recv_off,
                      void *stream);

/* ---- K10  torch.optim.Adam.step, train.py:139-141,184 -----------------------------------------
 * One tensor of `count` floats; step is the 1-based step number.  lr, betas, eps as torch defaults.
 * If wt != NULL the updated parameter, viewed as [dim][dim], is also written transposed to wt. */
int gss_adam_step(int64_t count, float *param, const float *grad, float *exp_avg, float *exp_avg_sq,
                  int32_t step, float lr, float beta1, float beta2, float eps, float *wt, int32_t dim,
                  void *stream);

/* ---- K12  np.percentile(E E^T, q), train.py:165-167 -------------------------------------------
 * Exact q-th percentile (linear interpolation) of all n*n inner products of the rows of e, by
 * 3-pass radix select on device.  Result to h_out (host float), synchronises the stream. */
int gss_percentile(int32_t n, int32_t d, const float *e, double q, float *h_out, void *stream);

/* ---- a1  gen_graph's similarity + top-k, helpers/helper.py:39-44 --------------------------------------
 * x: [n][d] fp64 features (the reference works in fp64 here).  For every row the k largest inner products
 * x_i . x_j over all j (self included, as np.argpartition(x_sim, -k, 1)[:, -k:] returns them; order within the
 * k is unspecified) -> top_val [n][k] fp64, top_idx [n][k] int32.  fp64 MFMA; d multiple of 8, k <= 64. */
int gss_knn_topk(int32_t n, int32_t d, const double *x, int32_t k, double *top_val, int32_t *top_idx, void *stream);
/* the same for rows [row_lo, row_hi) only (against all n columns): top_val / top_idx [row_hi - row_lo][k].  What a rank of a sharded
 * job computes for its own row window (shards.KnnSource); a row's result does not depend on the window. */
int gss_knn_topk_rows(int32_t n, int32_t d, const double *x, int32_t k, int32_t row_lo, int32_t row_hi, double *top_val,
                      int32_t *top_idx, void *stream);

/* ---- f4  diffusion profiles, multiscale/diff_prof/diffusion_profiles.py:30-90 ---------------------------
 * Personalised PageRank from every drug / indication, all start nodes at once: column c of the fp64 matrix
 * x[n][kpad] is the visit-probability vector of start node start[c].  The reference builds one matrix M_s per start
 * node s (edges from every other drug / indication to its proteins cut, edges from s's proteins into s cut, rows
 * renormalised; :30-56) and runs  x <- alpha (x M_s + dangling(x) e_s) + (1 - alpha) e_s  until
 * ||x - x_last||_1 < n tol (:65-90).  Here ONE shared matrix M' -- every drug / indication row in its "not selected"
 * form, i.e. cut and renormalised (normally empty: a pure sink) -- is multiplied into all columns by one fp64 SpMM per
 * iteration; what differs per start node is applied around the product:
 *   - ovr_*: entry (ovr_row[e], column ovr_col[e]) of x is multiplied by ovr_ratio[e] for the product.  Proteins of
 *     the start node: their row of M_s is renormalised without the cut edge (ratio 0 = the row became empty; listed in
 *     zero_* and counted as dangling).  The start node's own entry gets ratio 0 (not dangling) when its "not
 *     selected" row is not empty, so that row does not act in its own column;
 *   - sel_*: the start node's row in its "selected" form (all its out-edges over their sum):
 *     y[sel_row[e]][sel_col[e]] += sel_val[e] * x[start][sel_col[e]];
 *   - keep_*: in-edges of a start node that survive the cut (none in the plain MSI, where drugs and indications only
 *     touch proteins): value M'[keep_row][start[c]]; the product's own value at (start[c], c) is replaced by their sum;
 *   - z_rows: the empty rows of M' (sinks, isolated nodes): dangling in every column except, for the start node's
 *     own row, its column (start_dangling[c] = 1 if even its "selected" row is empty).
 * Columns converge independently and are frozen at the iteration the reference would have returned them.
 * All pointers are device pointers except h_rowptr.  kpad (the row stride of x) must be a multiple of 64. */
typedef struct gss_ppr gss_ppr;
typedef struct gss_ppr_desc {
  int32_t n, k, kpad;
  int64_t nnz;
  const int32_t *h_rowptr;                 /* host copy of t_rowptr */
  const int32_t *t_rowptr, *t_col;         /* CSR of M'^T: row j lists in-neighbours i ascending */
  const double *t_val;                     /* M'[i][j] */
  const int32_t *start;                    /* [k] */
  const int32_t *start_dangling;           /* [k] */
  int32_t n_z;
  const int32_t *z_rows;                   /* [n_z] ascending */
  int64_t n_ovr;
  const int32_t *ovr_col, *ovr_row;        /* [n_ovr] */
  const double *ovr_ratio;                 /* [n_ovr] */
  const int32_t *zero_ptr, *zero_ovr;      /* [k+1], indices e of dangling overrides, grouped by column */
  int64_t n_sel;
  const int32_t *sel_col, *sel_row;        /* [n_sel], (row, column) pairs unique */
  const double *sel_val;
  const int32_t *keep_ptr, *keep_row;      /* [k+1], rows */
  const double *keep_val;
  const int32_t *ovr_ptr;                  /* optional, [k+1]: the overrides of column c are entries [ovr_ptr[c], ovr_ptr[c+1]) (ovr_col
                                              non-decreasing).  With it (or with n_ovr == 0) the update x <- alpha (...) + (1 - alpha) e_s and
                                              the column errors run in the SpMM's epilogue instead of in a pass of their own */
} gss_ppr_desc;
int gss_ppr_create(gss_ppr **out, const gss_ppr_desc *desc);
void gss_ppr_destroy(gss_ppr *p);
size_t gss_ppr_device_bytes(const gss_ppr *p);
/* every buffer the handle carves from its slab is followed by a 256-byte guard no kernel may touch; this synchronises the device and
 * verifies them all (as gss_plan_check_guards).  For tests. */
int gss_ppr_check_guards(gss_ppr *p);
/* x <- 1/n, then iterate.  iters_out: host [k], the iteration at which each column converged.  Returns
 * GSS_ENOTCONV (x holds the last iterate) if some column needs more than max_iter iterations -- where the reference
 * raises (:90).  Synchronises the stream once per iteration (reads the number of unconverged columns). */
int gss_ppr_run(gss_ppr *p, double alpha, double tol, int32_t max_iter, double *x, int32_t *iters_out, void *stream);
/* one product y = M'^T x on the handle's schedule (fp64; the kernel the iteration is built on) */
int gss_ppr_spmm(gss_ppr *p, const double *x, double *y, void *stream);

/* ---- a13  np.savetxt('graph_embs.txt', hidden_emb), train.py:193 (host-side; h_emb is a HOST pointer) ---------------------
 * Every value of the float32 matrix as Python prints it with '%.18e' after widening to double (exact decimal expansion, round
 * half to even -- byte-identical to np.savetxt's output), ' ' between the values of a row, '\n' after each row.  Rows are
 * formatted by `threads` host threads (0 = all cores) and written in order.  gss_format_e18: one value into a buffer of >= 26
 * bytes, returns its length (the unit the writer is tested by). */
int gss_write_embs_text(const char *path, const float *h_emb, int64_t n, int32_t d, int32_t threads);
int gss_format_e18(float value, char *out26);
/* ---- a14  the '.embs.txt' reader, train.py:79-80 (np.loadtxt(..., skiprows=1, dtype=object)[:, 1:].astype(float)) -----------
 * First line '<N> <d>' (multiscale/openne/node2vec.py:42), then '<node> v1 ... vd' per line; blank lines skipped.  Values are
 * parsed with strtod (correctly rounded: the doubles Python's float() gives) by `threads` host threads (0 = all cores).
 * gss_embs_copy: x_out HOST fp64 [rows][cols]; names_out: the node names joined by '\n' (gss_embs_names_bytes bytes);
 * header_n: the N of the header line or -1.  A ragged or non-numeric line makes gss_embs_open fail. */
typedef struct gss_embs_file gss_embs_file;
int gss_embs_open(gss_embs_file **out, const char *path, int32_t threads);
int64_t gss_embs_rows(const gss_embs_file *e);
int32_t gss_embs_cols(const gss_embs_file *e);
int64_t gss_embs_names_bytes(const gss_embs_file *e);
int gss_embs_copy(const gss_embs_file *e, double *x_out, char *names_out, int64_t names_cap, int64_t *header_n);
void gss_embs_close(gss_embs_file *e);

/* ---- a2  the weighted edgelist 'u v w' (nx.write_weighted_edgelist, predict_drug.py:224-226) as --adj-file reads it ----------
 * names: the node ids of the .embs.txt in row order, joined by '\n' (names_bytes bytes, n_names ids): every u / v is mapped to its
 * row.  Lines are parsed by `threads` host threads; a missing weight is 1.0; blank and '#' lines are skipped.
 * gss_edgelist_bad_line >= 0: the first line (0-based among the data lines) with an unknown node id or a malformed weight. */
typedef struct gss_edgelist_file gss_edgelist_file;
int gss_edgelist_open(gss_edgelist_file **out, const char *path, const char *names, int64_t names_bytes, int64_t n_names, int32_t threads);
int64_t gss_edgelist_edges(const gss_edgelist_file *e);
int64_t gss_edgelist_bad_line(const gss_edgelist_file *e);
int gss_edgelist_copy(const gss_edgelist_file *e, int32_t *src, int32_t *dst, double *w);
void gss_edgelist_close(gss_edgelist_file *e);

/* ---- whole training step (train.py:158-184) ---------------------------------------------------
 * A plan owns every activation/gradient buffer of one replica so that a step is ONE host call that
 * enqueues all kernels.  a / at: CSR(A_hat) and CSR(A_hat^T) (at may be NULL for num_layers == 1). */
typedef struct gss_plan_desc {
  int32_t n;          /* nodes (rows of this replica) */
  int32_t d;          /* hidden units */
  int32_t num_layers; /* model.py:199 */
  int32_t max_batch;  /* largest batch the plan will see */
  float layer_decay;  /* model.py:202 */
  float alpha;        /* model.py:220 */
  float lr, beta1, beta2, eps; /* Adam */
  int32_t cache_layer1; /* 1: keep AX/AM of layer 1 (inputs are constant) across steps */
  int32_t pipeline_layer1; /* 1: gss_plan_step runs the NEXT step's layer-1 SpMMs (constant inputs) on an internal second
                              stream underneath this step's MFMA-bound kernels; all work still executes every step and
                              the results are bitwise unchanged */
  const int32_t *node_map; /* NULL, or device int32 [N] (borrowed): the row of every node id a batch may name.  For graphs whose
                              nodes were relabelled (hub-first, for gather locality): callers keep naming nodes by their
                              original ids, the plan looks the batch up in this map first */
} gss_plan_desc;

/* caller-owned tensors the plan reads and writes (all device pointers, fp32) */
typedef struct gss_plan_io {
  const float *x;             /* [n][d] input features (train.py:125) */
  float *w1, *b1, *w2, *b2;   /* gcn_layer.dense{,2}.{weight,bias}; updated in place by gss_plan_adam */
  float *emb;                 /* [n][d] out: unit-norm embeddings of the last forward */
  float *loss;                /* [1]    out: loss of the last gss_plan_loss_backward */
  float *gw1, *gb1, *gw2, *gb2; /* out: gradients of the last backward */
} gss_plan_io;

int gss_plan_create(gss_plan **out, const gss_plan_desc *desc, const gss_csr *a, const gss_csr *at,
                    const gss_plan_io *io);

/* One shard of a node-range sharded replica.  Rank r owns the node range [bounds[r], bounds[r+1]) -- the rows
 * of A_hat and A_hat^T and the matching rows of every activation and gradient; weights are replicated.
 *
 * Operand layout ("halo" form of C1): an SpMM operand of this shard is a [n + n_halo][d] buffer -- its own n rows first,
 * then the rows of other shards that its CSR actually references (the boundary features), grouped by owner in ascending
 * node order.  The CSR handles carry LOCAL OPERAND ROW ids as column ids (n_cols = n + n_halo).  A_hat and A_hat^T
 * reference different nodes, so each has its own halo.  Before a hop every rank packs the rows its peers reference
 * (d_send_rows) and one fused group of point-to-point transfers (gss_exchange_rows) delivers exactly those rows:
 * nothing is padded and pairs of shards that share no edge exchange nothing.
 *   desc->n = rows of THIS shard; io->x / io->emb: this shard's rows; the weights / gradients / loss in io are full-size
 *   and end up identical on every rank.
 * Every gss_plan_* call then is a collective: all ranks call it with the same batch.  Per step the plan enqueues, on the
 * caller's stream, 2L - 2 + max(0, 2L - 3) halo exchanges (C1; the boundary rows of the input features and of layer 1's M --
 * constants -- are fetched once; every shard still recomputes its own rows of them each step), one all-reduce of
 * the B gathered batch rows and one of their 2 B d input gradients (C3), and one grouped all-reduce of the four weight
 * gradients (C2); no host round trip in between.  world == 1 is exactly gss_plan_create.  gss_plan_backward (external
 * upstream gradient) is not available on a sharded plan. */
typedef struct gss_halo_desc {
  const int64_t *h_recv_off;   /* host [world + 1]: halo rows [recv_off[q], recv_off[q+1]) are owned by rank q; n_halo = recv_off[world] */
  const int64_t *h_send_off;   /* host [world + 1]: packed rows [send_off[q], send_off[q+1]) go to rank q */
  const int32_t *d_send_rows;  /* device [send_off[world]]: the local row behind every packed row (borrowed: keep it alive) */
} gss_halo_desc;
typedef struct gss_shard_desc {
  int32_t world, rank;
  const int64_t *h_bounds;     /* host, [world + 1], bounds[0] = 0, bounds[world] = N */
  gss_halo_desc halo_a;        /* operand halo of A_hat's columns */
  gss_halo_desc halo_at;       /* operand halo of A_hat^T's columns (ignored when num_layers == 1) */
  const int32_t *d_gid2op_t;   /* device [N], borrowed: node id -> operand row of A_hat^T's column space ([0, n) own rows, then
                                  halo), -1 where this shard never reads the node.  NULL when num_layers == 1 */
  /* Optional: every matrix once more, split by column -- *_own holds the entries whose column is one of the shard's own rows
   * (operand rows [0, n)), *_halo the entries that reference boundary rows; same rows, same operand-row column ids, entries in
   * their original order.  With them a hop is overlapped with its exchange: the boundary rows travel on a second stream while the
   * own-column entries are multiplied, the boundary-column entries are added afterwards (gss_spmm_add) and the epilogue runs
   * there.  A row is then summed as (own entries) + (boundary entries) instead of in column order: results differ from the
   * single-GPU plan by rounding (~1e-7 relative), no longer bit for bit.  All NULL: exchange, then one pass (the default). */
  const gss_csr *a_own, *a_halo, *at_own, *at_halo;
  /* Optional (round 4): the shard's A_hat transposed IN PLACE -- rows = its operand rows (own rows, then the boundary rows), columns =
   * its own rows, [n + n_halo_a] x [n], the same values.  With it (and halo_recompute) the LAST backward hop, whose result only feeds
   * the bottom layer's weight gradient, runs in scatter-by-owner form: this shard multiplies ITS rows of u into every row they touch
   * (own and boundary), applies ELU'(P) there (P of the boundary rows is what halo_recompute computes anyway) and sums the weight
   * gradient over own + boundary rows; the ranks' all-reduce of the weight gradients completes the sum.  No exchange of u's boundary
   * rows in that hop: 3 collectives per step at two layers.  The weight gradients then add per-rank partial sums (rounding-level
   * differences, as with any other change of their summation order).  NULL: the hop fetches u's boundary rows (the default). */
  const gss_csr *a_loc_t;
} gss_shard_desc;
int gss_plan_create_sharded(gss_plan **out, const gss_plan_desc *desc, const gss_shard_desc *shard, gss_comm *comm,
                            const gss_csr *a, const gss_csr *at, const gss_plan_io *io);
/* this shard's rows of the last embeddings gathered from every shard: out [N][d] in node order (collective) */
int gss_plan_gather_embeddings(gss_plan *p, float *out, void *stream);
void gss_plan_destroy(gss_plan *p);
/* forward only (model.py:197-207): writes io.emb */
int gss_plan_forward(gss_plan *p, void *stream);
/* loss + backward for batch idx[0..b): writes io.loss and io.g* */
int gss_plan_loss_backward(gss_plan *p, const int32_t *idx, int32_t b, float beta, void *stream);
/* backward only, for an arbitrary upstream gradient that is non-zero on `b` unique rows: de_rows is the
 * compact [b][d] dLoss/dEmbedding of rows[0..b) (the autograd-glue path; NULL = the plan's own de_b) */
int gss_plan_backward(gss_plan *p, const int32_t *rows, int32_t b, const float *de_rows, void *stream);
/* Adam on the four tensors with the plan's gradients; step numbers are counted by the plan */
int gss_plan_adam(gss_plan *p, void *stream);
/* forward + loss + backward + Adam = one iteration of train.py:155-184.  idx[0..b): the batch's node ids (device), DISTINCT -- what the
 * reference's sampler draws (a permutation cut into batches, method/dataset.py:5-28); the batch-position map holds one position per
 * node, so a repeated id would leave the other position's rows unwritten.  The same holds for gss_plan_step_lazy and
 * gss_plan_loss_backward. */
int gss_plan_step(gss_plan *p, const int32_t *idx, int32_t b, float beta, void *stream);
/* gss_plan_step with the top layer evaluated on the b batch rows only -- the rows of it that the loss (model.py:216-221) and the
 * backward pass read; the reference computes all N every step (train.py:158-161) and reads B of them.  Loss, gradients and
 * parameters equal gss_plan_step's bit for bit (a computed row takes the same path through the same kernels).  Afterwards io.emb
 * holds the step's embeddings on the batch rows only: call gss_plan_forward when all of them are wanted (the beta percentile of
 * step 0, the embeddings that are written out).  On a sharded plan every shard evaluates the top layer on the batch rows it owns.
 * One-layer plans run the full step. */
int gss_plan_step_lazy(gss_plan *p, const int32_t *idx, int32_t b, float beta, void *stream);
/* Sharded plans, knob lazy_halo (default: graphs of >= 262,144 nodes): two hops fetch a SUBSET of their operand's boundary rows.
 * (1) Lazy steps, the top layer's M: only the boundary rows that the batch rows of this shard reference (the receiver marks them over
 * its halo slots and sends the bitmaps to the owners).  What is needed depends on the batch ids and the graph alone, so the request
 * phase is the FIRST thing gss_plan_step_lazy enqueues, on the plan's own request stream: it runs underneath layer 1, the row counts
 * reach the host through an event-gated async copy, and the host waits for that event -- never for the caller's stream -- just before
 * it enqueues the transfer (gss_plan_sync_stats).  On by default on every transport.
 * (2) Every step, u -- the operand of the top layer's second backward hop, zero outside the batch's neighbourhood --: only the rows
 * that can be non-zero (the owners send them and the bitmap that says which).  That bitmap is an OUTPUT of the hop before it, so the
 * host drains the caller's stream once per step for the counts; knob lazy_halo_u (-1, the default: with lazy_halo on the host-side
 * transports, not over RCCL -- there a two-layer plan runs the hop exchange-free on gss_shard_desc.a_loc_t instead; 0 never; 1 with lazy_halo).
 * Same bits as the full exchanges.  out6 reports what the LAST such exchanges moved: {rows fetched, rows sent, rows of the whole
 * halo} for (1) and then for (2); fetched / sent are -1 when the plan exchanges whole halos. */
int gss_plan_lazy_halo_rows(const gss_plan *p, int64_t *out6);
/* Collectives a sharded plan has enqueued since the last call (then reset): out3 = {boundary-row exchanges, batch-row all-reduces,
 * weight-gradient all-reduces}.  A steady full step at L layers: 2L - 2 + 2L - 3 exchanges (one less with halo_recompute: layer 2's
 * boundary input rows are recomputed from layer 1's constant AX / AM), 1 batch-row all-reduce ([E_B | P_B | inv_B] as one buffer; 2
 * at widths outside {64, 128, 256} or with the row-slab loss sweep, knob loss_slab), 1 weight-gradient all-reduce.  Zeros on one GPU. */
int gss_plan_comm_stats(gss_plan *p, int64_t *out3);
/* Host-side waits of a sharded plan since the last call (then reset): out2 = {waits that DRAIN the caller's stream (the device idles
 * until the host has enqueued again): the sender-driven subset exchange of u, knob lazy_halo_u; waits for an EVENT of the plan's
 * request stream while the caller's stream keeps running: the request phase of the lazy step's subset exchange}.  An RCCL job with
 * default knobs: {0, 0} per full step, {0, 1} per lazy step from 262,144 nodes on, {0, 0} below.  Zeros on one GPU. */
int gss_plan_sync_stats(gss_plan *p, int64_t *out2);
/* layer activations for parity tests: which 0 AX, 1 AM, 2 P of layer `layer` (0-based); 3 the bottom layer's dP [n][d] and 4 the top layer's dP on
 * the batch rows [b][d] as the last backward pass left them; 5 / 6 u and t of the top layer's first backward hop, 7 the uint32 bitmap of the rows of
 * u / t it wrote (NULL on plans without it: then every row is written), 8 the batch rows' input gradients [g_ax ; g_am] as [2 b][d] -- `layer`
 * is ignored for 3 .. 8 */
const float *gss_plan_activation(const gss_plan *p, int layer, int which);
/* bytes of the plan's slab.  Not counted: what a gss_csr handle caches for itself -- its segment descriptors, the chunk scratch of giant rows (one
 * buffer per stream the handle is used on, grown to the widest d seen), the live-workgroup list of row-filtered products. */
size_t gss_plan_device_bytes(const gss_plan *p);
/* every buffer the plan carves from its slab is followed by a 256-byte guard no kernel may touch; this synchronises the device and
 * verifies them all (GSS_EINVAL names the first one that was overwritten).  For tests. */
int gss_plan_check_guards(gss_plan *p);
/* set the 1-based Adam step counter (resume; also marks every buffer derived from the weights stale) / read it */
void gss_plan_set_step(gss_plan *p, int32_t step);
int32_t gss_plan_get_step(const gss_plan *p);
/* checkpoint / resume: device pointer of Adam's exp_avg (moment 0) or exp_avg_sq (moment 1) of tensor 0..3 =
 * W1, b1, W2, b2 (same shapes as the parameters); NULL on a bad argument */
float *gss_plan_adam_buffer(gss_plan *p, int32_t moment, int32_t tensor);
/* Per-kernel-class timing with HIP events recorded on the caller's stream around every launch of a
 * plan call (bench.py's live roofline measurement).  gss_plan_profile_read synchronises the stream, returns
 * the accumulated milliseconds and launch counts per class (arrays of GSS_PROF_CLASSES) and resets them. */
enum {
  GSS_PROF_SPMM_FWD_HAD = 0, /* AX = A x with fused Hadamard epilogue */
  GSS_PROF_SPMM_FWD = 1,     /* AM = A M */
  GSS_PROF_SPMM_BWD1 = 2,    /* the top layer's first backward hop, batch-sparse (SPMM_BWD1S) */
  GSS_PROF_SPMM_BWD2 = 3,    /* the top layer's second backward hop with the batch rows' residual folded in (SPMM_BWD2S) */
  GSS_PROF_DENSE_FWD = 4,
  GSS_PROF_DGRAD = 5,
  GSS_PROF_WGRAD = 6,        /* N-row weight gradient (+ its reduce) */
  GSS_PROF_WGRAD_BATCH = 7,  /* batch-row weight gradient of the top layer */
  GSS_PROF_LOSS = 8,
  GSS_PROF_ROWNORM = 9,
  GSS_PROF_ELEMENTWISE = 10, /* norm/ELU backward on batch rows, transposes, memsets, scatter */
  GSS_PROF_ADAM = 11,
  GSS_PROF_COMM = 12,        /* sharded plans: boundary-row exchanges of the SpMM hops (pack kernel + grouped send / recv) */
  GSS_PROF_COMM_BATCH = 13,  /* sharded plans: the batch-row all-reduce(s) of the loss */
  GSS_PROF_COMM_GRADS = 14,  /* sharded plans: the all-reduce of the four weight gradients */
  GSS_PROF_SPMM_BWD1_DENSE = 15,  /* (round 6) an N-row first backward hop: layers below the top one at L >= 3, the phase-wise entry points */
  GSS_PROF_SPMM_BWD2_DENSE = 16,  /* (round 6) an N-row second backward hop */
  GSS_PROF_CLASSES = 17
};
int gss_plan_profile(gss_plan *p, int enable);
int gss_plan_profile_read(gss_plan *p, double *ms_out, int64_t *count_out, void *stream);
/* tuning/debug knobs (A/B runs inside one process; 18 of them since round 6 -- the access-shape variants whose sweeps said "default holds" in
 * two or more rounds were removed from the kernels): "spmm_list_blocks" = workgroups from which a ROW-FILTERED balanced SpMM (the lazy step's
 * top-layer products, the batch-sparse backward hop) lists the workgroups that hold a passing row and walks the list with persistent
 * workgroups instead of dispatching every workgroup (default 2048; 0 = never; same bits); "spmm_variant" = 1 (whole-row gather, wave per row) or
 * 2 (nnz-balanced segments, default); "spmm_slices" = 0 (automatic, default) or 1..8 feature slices in the balanced SpMM, "spmm_pin" = with a
 * manual "spmm_slices": slices time-separated (0, default) or pinned to XCDs (1) -- the automatic policy pins operands of <= 64 MB;
 * "spmm_hot_rows" = -1 (default: what gss_csr_set_hot declared) or a row count; "spmm_seg_edges" = entries per SpMM segment (default 32;
 * applies to gss_csr handles created afterwards); "spmm_giant" = stored entries above which a row is summed chunk by chunk across workgroups
 * (default 32768, 0 = never; gss_csr_giant_rows); "gemm_variant" = projection tile shape: 2 (by width and row count; default), 3 (128-node
 * tiles of four waves forced), 5 (128-node tiles of eight waves forced); "gemm_ws" = -1 (default: from 32,769 rows on -- more 128-node
 * tiles than CUs) / 0 / 1: the d = 128 forward projection as the weight-stationary persistent kernel (same bits);
 * "wgrad_wgs", "loss_wgs" = workgroups of a full-size weight-gradient launch / the loss sweep (default 256 = one per CU; set
 * before plans are created); "sparse_bits_rows" = operand rows from which a plan keeps the bitmaps of the sparsity-aware backward hops
 * (default 100000); "ppr_fused" = 1 (default) / 0 (separate update pass of the diffusion profiles);
 * "lazy_halo" = -1 (default: graphs of >= 262,144 nodes, but never over RCCL, where it stays opt-in) / 0 / 1: sharded plans fetch subsets of the
 * boundary rows where a hop reads a subset (gss_plan_lazy_halo_rows), "lazy_halo_u" = -1 / 0 / 1 the same for u in the second backward hop (see gss_plan_lazy_halo_rows); "halo_recompute" = -1 (default: on) / 0 / 1: sharded plans recompute layer 2's boundary input rows from layer 1's constant AX / AM (fetched once)
 * instead of exchanging them every step (same bits); "loss_dgrad" = -1 (default: on shards only) / 0 / 1: the
 * loss finish and the batch rows' input gradient in one launch instead of two (same bits; on a shard it spares a collective); "prep_side" = 1 (default) / 0: on one GPU a step's batch preparation rides in its
 * first forward SpMM launch and E_B comes out of the top layer's projection (no batch_prepare / gather launch; same bits);
 * JOB-WIDE knobs -- "lazy_halo", "lazy_halo_u", "halo_recompute", "loss_slab" -- must have the same value on every rank:
 * gss_plan_create_sharded compares them across the ranks (one all-reduce of min / max) and fails by name when they differ;
 * "loss_slab" = -1 (default: batches of >= 8192
 * rows) / 0 / 1: sharded plans sweep the B x B loss as row slabs (rank r the i tiles r, r + P, ...; one more all-reduce of B d + 1
 * floats) instead of replicating it on every rank (every rank of a job must use the same value; results agree to rounding).  Every setting computes the same results (some in a different summation order); the defaults are
 * the measured optima recorded in DESIGN.md section 4.  The values are process-wide DEFAULTS: a plan (and a gss_ppr handle) takes a
 * snapshot when it is created and runs under it from then on, so changing a knob never re-shapes a live plan -- in particular not
 * the plans of other rank threads of the same process; per-op entry points read the current defaults. */
int gss_debug_set_option(const char *name, int value);
/* plain device-to-device copy on `stream` (lets a ctypes host read plan-owned activations) */
int gss_memcpy_d2d(void *dst, const void *src, size_t bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* GSSGCN_H */
