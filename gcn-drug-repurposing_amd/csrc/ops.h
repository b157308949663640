// ops.h -- internal launchers shared between the per-op C entry points and the plan.
#pragma once
#include <vector>

#include "common.h"

struct gss_csr {
  int32_t n_rows, n_cols;
  int64_t nnz;
  const int32_t *rowptr, *col;
  const float *val;
  int32_t n_long;
  int32_t *d_long_rows;
  int32_t max_row;
  int32_t hot_own, hot_halo0, hot_halo1;  // gss_csr_set_hot: operand rows [0, hot_own) and [hot_halo0, hot_halo1) are the hubs' (-1: no split)
  std::vector<int32_t> h_rowptr;  // host copy, used to build segment descriptors lazily
  int32_t *d_segs[5];             // balanced SpMM: int4 descriptors per lane group, by log2(groups per wave)
  int32_t n_seg_blocks[5];
  int32_t *d_live[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};   // row-filtered launches over many workgroups: [0] = count, [1 ..] = the workgroups (segment blocks) that hold a row the filter lets through (spmm.hip live_blocks_kernel)
  // a VIEW (spmm.hip GiantRows): the schedule covers the listed rows / entry ranges of the borrowed arrays instead of every row of rowptr
  bool by_items = false;
  std::vector<int32_t> item_row, item_first, item_len;
  struct gss_giant_rows *giant = nullptr;   // rows too long for one workgroup, cut into chunks (built on first use; spmm.hip)
  int32_t giant_threshold = -1;             // the knob value `giant` was built (or found empty) for; -1 = not looked at yet
};

// communicator interface of the sharded plan (comm.hip): RCCL, or threads of one process
struct gss_comm {
  int world = 1, rank = 0;
  virtual ~gss_comm() {}
  // make every rank that is (or will be) blocked in a collective of this group return an error: the local backend releases its
  // host barrier, RCCL calls ncclCommAbort (the communicator is unusable afterwards)
  virtual void abort() = 0;
  virtual int check_async() = 0;                        // GSS_ECOMM once the backend has seen a failure (RCCL: ncclCommGetAsyncError)
  virtual int count(int32_t *out) = 0;                  // ranks the backend itself reports (RCCL: ncclCommCount)
  virtual int sync(hipStream_t st, double timeout_s) = 0;  // wait for `st` while watching for errors; past the deadline: abort + GSS_ETIMEOUT
  // rank r's `bytes_per_rank` land at recv + r * bytes_per_rank; in place when send == recv + rank * bytes_per_rank
  virtual int all_gather(const void *send, void *recv, size_t bytes_per_rank, hipStream_t st) = 0;
  // true for a transport that moves device memory between devices (RCCL): its host round trips (sync) stall a whole GPU
  virtual bool device_transport() const { return false; }
  // in place, nbuf tensors as one fused operation; identical bits on every rank
  virtual int all_reduce_sum(float *const *bufs, const size_t *counts, int nbuf, hipStream_t st) = 0;
  // halo exchange: rows [send_off[q], send_off[q+1]) of `send` go to rank q, rows from rank q land at [recv_off[q], recv_off[q+1])
  // of `recv` (row = d floats; offsets are host arrays of world + 1 entries; own entries are empty)
  virtual int exchange_rows(const float *send, const int64_t *send_off, float *recv, const int64_t *recv_off, int d, hipStream_t st) = 0;
  // measurement aid of the in-process backend (gss_comm_local_mode): record what every collective delivers / serve it again without peers
  virtual int set_mode(int /*mode*/) { return -1; }
  virtual int log_bytes(int64_t * /*out*/, int32_t /*cap*/, int32_t * /*n_out*/) { return -1; }
};

namespace gss {
// What batch_prepare does, as a SIDE JOB of a forward SpMM launch (round 4): one extra workgroup of the launch translates the batch ids
// and sets the batch-position map while the other workgroups multiply -- a launch of ~5 us (two dependent round trips around 2048
// integers) disappears from the step's chain.  Nothing in that SpMM or before the top layer's projection reads what it writes.
struct BatchPrep {
  const int32_t *idx;        // NULL: no side job
  int32_t b;
  const int32_t *node_map;
  int32_t lo, nl;
  const int32_t *gid2op;
  int32_t *rloc, *pid;
  float *keep;
  int32_t *pos, *rlist;
};
// nnz-balanced segment descriptors of a CSR for 2^gpw_log2 lane groups per wave (spmm.hip; cached in the handle)
int csr_segments(const gss_csr *a, int gpw_log2, const int4 **out, int *n_blocks);
// Two-pass products (a shard's hop overlapped with its halo exchange, plan.hip): the rows' entries are split over two CSRs of the
// same rows; the first pass is a plain product into `y`, the second pass -- any mode -- takes it as y_in, adds its own sums and
// runs the epilogue.  y_in may be the buffer the second pass writes.
// What the caller expects a row-filtered product to compute, in rows (an upper estimate; 0 = unknown): a launch lists its live workgroups
// (spmm.hip live_blocks_kernel) only when that is a small part of the matrix's rows -- walking a list with persistent workgroups costs ~15 % over
// the hardware's own dispatch when most workgroups are live (a hub shard's lazy products: every row is a neighbour of some batch row).
struct LiveHint {
  int64_t prev;
  explicit LiveHint(int64_t rows);
  ~LiveHint();
  LiveHint(const LiveHint &) = delete;
  LiveHint &operator=(const LiveHint &) = delete;
};
int spmm_fwd(const gss_csr *a, int32_t d, const float *x, float *y, const float *h, float *m, void *stream,
             const int32_t *row_pos = nullptr,    // plain product only: compute rows with row_pos[row] >= 0 only
             const uint32_t *row_bits = nullptr,  // Hadamard-fused product only: compute rows whose bit is set only
             const float *y_in = nullptr,
             const uint32_t *gather_bits = nullptr,   // plain product only: neighbours whose bit is clear are skipped (their rows are zero)
             const BatchPrep *prep = nullptr);        // balanced SpMM only: the batch preparation as a side job of this launch
int mark_rows_and_neighbours(const gss_csr *a, const int32_t *rows, int32_t b, uint32_t *bits, void *stream);
int spmm_bwd1(const gss_csr *at, int32_t d, const float *g_am, const float *g_ax, const float *x_in, const float *ax,
              float *u, float *t, void *stream, const float *y_in = nullptr);
int spmm_bwd2(const gss_csr *at, int32_t d, const float *u, const float *t, const float *p, float c, const float *res,
              float *dp, float *gx_out, void *stream, const float *y_in = nullptr,
              int32_t own_row_limit = 0);   // > 0: t and res exist for output rows below it only (the in-place transposed shard, plan.hip)
int dense_fwd(int32_t n, int32_t d, const float *ax, const float *am, const float *w1, const float *b1, const float *w2,
              const float *b2, const float *p_prev, float decay, float *p, float *x_next, void *stream,
              const int32_t *row_list = nullptr);  // row_list: the n tile rows are node rows row_list[0..n) of every operand; a negative
                                                   // entry is skipped (nothing of it is written)
bool dense_fwd_norm_available(int32_t d);
int dense_fwd_norm(int32_t n, int32_t d, const float *ax, const float *am, const float *w1, const float *b1, const float *w2,
                   const float *b2, const float *p_prev, float decay, float *p, float *e, float *inv_den, void *stream,
                   const int32_t *row_list = nullptr, float *rows_out = nullptr,    // rows_out: tile row t's unit-norm row also to rows_out[t]
                   const int32_t *rows_out_pos = nullptr);   // without a row list: node row r's unit-norm row also to rows_out[rows_out_pos[r]]
                                                             // where that is >= 0 (the batch-position map: E_B without a gather launch)
float *loss_workspace_e_b(int32_t d, int32_t b, void *ws);   // where loss_gather_rows* would put E_B for a batch of b rows
int dense_bwd_input(int32_t n, int32_t d, const float *dp, const float *w1t, const float *w2t, const int32_t *rows,
                    float *g_ax, float *g_am, void *stream);
size_t wgrad_workspace_bytes(int32_t n, int32_t d);
int dense_bwd_weight(int32_t n, int32_t d, const float *dp, const float *ax, const float *am, const int32_t *rows,
                     float *gw1, float *gw2, float *gb, float *gb2, int accumulate, void *ws, void *stream);
int wgrad_partial(int32_t n, int32_t d, const float *dp, const float *ax, const float *am, const int32_t *rows, void *ws,
                  int total_slices, int slice0, int *nslices_out, void *stream);
int wgrad_partial_pair(int32_t d, int32_t n0, const float *dp0, const float *ax0, const float *am0, const int32_t *rows0, int slice0_0,
                       int32_t n1, const float *dp1, const float *ax1, const float *am1, const int32_t *rows1, int slice0_1, void *ws,
                       int total_slices, int *ns0_out, int *ns1_out, void *stream);
int wgrad_reduce(int32_t d, void *ws, int total_slices, int nslices, float *gw1, float *gw2, float *gb, float *gb2, int accumulate,
                 void *stream);
int wgrad_slices(int32_t n, int32_t d);
int wgrad_slices_max(int32_t n_max, int32_t d);  // bound over every n in [1, n_max]
int wgrad_reduce_adam(int32_t d, void *ws, int total_slices, int nslices, float *const grad[4], float *const param[4],
                      float *const m[4], float *const v[4], int32_t step, float lr, float beta1, float beta2, float eps, float *w1t,
                      float *w2t, int32_t *pos_clear, const int32_t *idx, int32_t b, void *stream);
int rownorm_fwd(int32_t n, int32_t d, const float *x, float *e, float *inv_den, void *stream, const int32_t *rows = nullptr);
int rownorm_elu_bwd(int32_t d, const float *de_b, const int32_t *idx, int32_t b, const float *e, const float *inv_den,
                    const float *p, float c, float *dx_b, float *dp_b, int32_t *pos_set, void *stream);
// dst[rows[r]] += src[r] unless rows[r] < 0 or keep[r] == 0 (keep nullable); pos_clear != NULL: also pos_clear[pos_ids[r]] = -1
int scatter_add_rows(int32_t d, const float *src, const int32_t *rows, const float *keep, int32_t b, float *dst, int32_t *pos_clear,
                     const int32_t *pos_ids, void *stream);
int pack_rows(int32_t d, const float *src, const int32_t *rows, int64_t n, float *out, void *stream);
int unpack_rows(int32_t d, const float *src, const int32_t *rows, int64_t n, float *dst, void *stream);   // dst[rows[k]] = src[k]
// lazy halo (plan.hip): mark the boundary rows the listed rows of `a` read, per owner in word-aligned bit ranges; list the set bits
int halo_need_mark(const gss_csr *a, const int32_t *rows, int32_t b, int32_t n, int P, const int64_t *d_recv_off, const int64_t *d_wrecv_off,
                   uint32_t *needw, void *stream);
int send_slot_bits(const uint32_t *bits, const int32_t *send_rows, int P, const int64_t *d_send_off, const int64_t *d_wsend_off, int64_t n_words,
                   uint32_t *out, void *stream);
int bits_clear(uint32_t *bits, int64_t first, int64_t last, void *stream);          // bits [first, last) := 0
int bits_set_list(uint32_t *bits, const int32_t *list, int64_t n, void *stream);    // bits[list[k]] := 1
int bits_compact(const uint32_t *words, int P, const int64_t *d_woff, const int64_t *h_woff, const int64_t *d_slot_off, const int32_t *map, int32_t add,
                 int32_t *out, int64_t *d_out_off, void *stream, void *scratch = nullptr, size_t scratch_bytes = 0);
size_t bits_compact_scratch_bytes(int P, const int64_t *h_woff);   // what bits_compact needs as scratch for these word offsets
int spmm_bwd1_sparse(const gss_csr *at, int32_t d, const float *g_am_b, const float *g_ax_b, const int32_t *pos,
                     const int32_t *pos_row, const float *x_in, const float *ax, float *u, float *t, void *stream,
                     const uint32_t *posbits = nullptr, uint32_t *nzbits_out = nullptr, int skip_zero_rows = 0, const uint32_t *live_rows = nullptr);
// bitmap companion of a batch-position map: set (value 1) or clear (value 0) the words of ids[0..b) (negative ids skipped)
int batch_bits(const int32_t *ids, int32_t b, uint32_t *bits, int set, void *stream);
bool spmm_sparse_available();
int spmm_bwd2_sparse_res(const gss_csr *at, int32_t d, const float *u, const float *t, const float *p, float c, const float *res_b,
                         const int32_t *pos_row, float *dp, float *gx_out, void *stream, const uint32_t *nzbits = nullptr,
                         const float *y_in = nullptr, int32_t pos_row_limit = 0);   // pos_row_limit > 0: pos_row covers output rows below it only
// rlist (nullable): per member the local row when this shard owns it, -1 otherwise -- the row list of the lazy top layer
int batch_prepare(const int32_t *idx, int32_t b, const int32_t *node_map, int32_t lo, int32_t nl, const int32_t *gid2op, int32_t *rloc,
                  int32_t *pid, float *keep, int32_t *pos, void *stream, int32_t *rlist = nullptr);
// bits [first, last) of a bitmap := 1 (whole and partial words; other bits untouched)
int bits_fill(uint32_t *bits, int64_t first, int64_t last, void *stream);
int adam_step(int64_t count, float *param, const float *grad, float *m, float *v, int32_t step, float lr, float beta1,
              float beta2, float eps, float *wt, int32_t dim, void *stream);
struct AdamTensor {
  float *param;
  const float *grad;
  float *m, *v;
  int64_t count;
};
int adam_step4(const AdamTensor (&t)[4], int32_t step, float lr, float beta1, float beta2, float eps, float *w1t, float *w2t,
               int32_t dim, void *stream, int32_t *pos_clear = nullptr, const int32_t *ids = nullptr, int32_t b = 0);
int loss_fwd_bwd_fused(int32_t n, int32_t d, const float *e, const int32_t *idx, int32_t b, float beta, float alpha, float *loss_out,
                       const float *inv_den, const float *p, float c, float *dx_b, float *dp_b, int32_t *pos_set, void *ws,
                       void *stream);
int loss_gather_rows(int32_t d, const float *e, const int32_t *rows, const float *keep, int32_t b, void *ws, float **e_b_out, void *stream);
int loss_gather_rows_mapped(int32_t d, const float *e, const int32_t *idx, const int32_t *node_map, int32_t lo, int32_t nl, const int32_t *gid2op,
                            int32_t *pid, int32_t *rloc, float *keep, int32_t b, void *ws, float **e_b_out, void *stream);
int loss_gather_batch(int32_t d, const float *e, const float *p, const float *inv_den, const int32_t *idx, const int32_t *node_map, int32_t lo,
                      int32_t nl, const int32_t *gid2op, int32_t *pid, int32_t *rloc, float *keep, const int32_t *rows, int32_t b, float *out,
                      void *stream);
int loss_fused_gathered(int32_t d, int32_t b, float beta, float alpha, float *loss_out, const int32_t *idx, const int32_t *pos_ids,
                        const float *keep, const float *inv_den, const float *p, float c, float *dx_b, float *dp_b, int32_t *pos_set,
                        void *ws, void *stream);
// One step's loss on the plan's path (loss.hip loss_step), over batch rows gathered beforehand: the sweep + the finish, with the batch
// rows' input gradient in the finish launch where the width allows
struct LossStep {
  int32_t d, b;
  float beta, alpha;
  float *loss_out;
  const float *e_b;                 // [b][d] gathered rows (NULL: the workspace's E_B, written by loss_gather_rows*)
  const int32_t *rows;              // row of inv_den / p per member (NULL: the member's position -- a shard's gathered batch)
  const int32_t *pos_ids;           // key of the batch-position map per member (NULL: as rows)
  int32_t *pos_set;                 // nullable
  const float *keep;                // nullable: 0 where another shard owns the member
  const float *inv_den, *p;
  float c;
  float *dx_b, *dp_b;
  // the batch rows' input gradient [g_ax | g_am] = dP [W1 ; W2] (nullable: w1t == NULL)
  const float *w1t, *w2t;
  float *gax_b, *gam_b;
  bool dgrad_all;                   // the input gradient of EVERY member (a shard that holds p / inv_den of the whole batch)
  // row-slab sweep of a shard: the i tiles slab_rank, slab_rank + slab_parts, ... (loss_step_slab_sweep), the ranks' sums in de_x
  int slab_rank, slab_parts;
  const float *de_x;                // loss_step after the ranks summed de_x: [b][d] = sum_js de_part of every row (NULL: sweep here)
};
int loss_step(const LossStep &s, void *ws, void *stream, bool *dgrad_done);
int loss_step_slab_sweep(const LossStep &s, void *ws, float *de_x, void *stream);
bool loss_dgrad_available(int32_t d, int32_t b, bool sharded);
int transpose2(int32_t dim, const float *a, const float *b, float *at, float *bt, void *stream);
size_t loss_workspace_bytes(int32_t b, int32_t d, int parts = 1);
size_t loss_workspace_bytes_max(int32_t b_max, int32_t d, int parts = 1);   // enough for every batch of 1..b_max rows (the size is not monotone in b);
                                                                              // parts > 1: also for the row-slab sweep over that many ranks
int loss_fwd_bwd(int32_t n, int32_t d, const float *e, const int32_t *idx, int32_t b, float beta, float alpha,
                 float *loss_out, float *de_b, void *ws, void *stream);
}  // namespace gss
