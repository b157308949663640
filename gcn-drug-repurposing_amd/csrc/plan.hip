// plan.hip -- one training replica, or one node-range shard of one: owns every activation / gradient buffer and
// enqueues a whole iteration of train.py:155-184 (forward, gss_loss, backward, Adam) from a single host call, so the
// kernels of a step (14 launches at L = 2) and, on a shard, the collectives between them (comm.hip: halo exchanges and
// all-reduces over RCCL) are issued back to back from C++ with no Python in between.
//
// Layer l (0-based) keeps x_l (input), AX_l, AM_l, P_l for the backward pass (model.py:163-173).
// The backward pass exploits that dLoss/dE is non-zero only on the B batch rows: normalise-bwd,
// ELU-bwd, the top layer's weight gradient and its input gradient run on B rows instead of N.
//
// Sharding: rank r owns rows [bounds[r], bounds[r+1]) of A_hat, A_hat^T and of every activation.  An SpMM
// operand is a [n + n_halo][d] buffer: the shard's own rows, written in place by the producing kernel, followed by the
// rows of other shards its CSR references (the boundary features), which one fused group of point-to-point transfers
// fetches before the hop.  The CSR column ids are operand rows.  A single-GPU plan is the case P = 1 with empty halos.
#include <algorithm>
#include <chrono>
#include <thread>
#include <vector>

#include "ops.h"

struct gss_plan {
  gss::Knobs knobs;   // the tuning knobs as they stood when the plan was created: every call on the plan runs under them
  gss_plan_desc desc;
  const gss_csr *a, *at;
  const float *x;
  float *w1, *b1, *w2, *b2;
  // one slab, carved
  char *slab;
  size_t slab_bytes;
  std::vector<float *> ax, am, p, xin;  // xin[l] = input of layer l as an operand of A_hat (own rows, then halo)
  float *m_tmp, *x_last, *emb, *inv_den;
  float *g_ax, *g_am, *u, *t, *dp, *gx[2];
  float *de_b, *dx_b, *dp_b, *gax_b, *gam_b;
  int32_t *pos;  // operand row of A_hat^T's column space -> batch position (-1 outside the batch), for the sparsity-aware backward SpMM
  uint32_t *posbits;  // bitmap of pos >= 0, kept only around the sparse backward SpMM and only for huge operands (else NULL)
  std::vector<size_t> guard_off;   // slab offsets of the guards behind the carved buffers
  uint32_t *needbits; // huge operands only (else NULL): gss_plan_step_lazy -- the rows of the top layer's AX / M that anything reads (batch + neighbours)
  bool needbits_valid = false;   // needbits holds the set of the batch the running step was given (plan_step_impl's forward or backward built it)
  uint32_t *nzbits;   // huge operands only (else NULL): bit r set <=> row r of u (the top layer's A_hat^T operand) may be non-zero; written by the
                      // sparse hop, read by the hop after it (which then skips the zero rows); halo rows are always set
  float *w1t, *w2t;
  float *grad[4];
  float *adam_m[4], *adam_v[4];
  float *loss;
  void *loss_ws, *wgrad_ws;
  int32_t step;
  bool layer1_valid;
  // software pipelining of layer 1 across steps (desc.pipeline_layer1): its two SpMMs read only constants (A_hat, X), so
  // step t+1's run on `side` underneath step t's MFMA-bound kernels, into the other half of a double buffer
  hipStream_t side;
  hipEvent_t ev_main_ready, ev_side_done;
  float *ax0[2], *am0[2], *m_side;
  int cur0;
  bool prefetched;
  bool wt_valid;      // w1t/w2t match w1/w2 (maintained by the fused Adam inside gss_plan_step only)
  int wg_total;       // slices of the shared weight-gradient partial buffer
  // optional per-kernel-class timing with HIP events on the caller's stream (bench.py roofline leg)
  bool prof_on;
  std::vector<hipEvent_t> ev;      // pairs: 2 k = start, 2 k + 1 = stop
  std::vector<int> ev_cls;
  size_t ev_used;
  double prof_ms[GSS_PROF_CLASSES];
  int64_t prof_cnt[GSS_PROF_CLASSES];
  // node-range sharding
  gss_comm *comm;     // NULL on one GPU
  int P, rank;
  int lo;             // first global node id of this shard
  std::vector<int64_t> h_bounds;
  struct Halo {       // operand halo of one matrix (gss_halo_desc with the offsets copied)
    int64_t n_halo = 0, n_send = 0;
    std::vector<int64_t> recv_off, send_off;
    const int32_t *d_send_rows = nullptr;
  } halo_a, halo_t;
  size_t rows_a, rows_t;   // operand rows of A_hat / A_hat^T: n + n_halo
  const int32_t *gid2op_t; // node id -> operand row of A_hat^T's column space (borrowed), NULL on one GPU
  float *sendbuf;          // packed rows for the peers, [max n_send][d]
  float *x0op;             // the input features with their halo (constant: exchanged once); io.x itself on one GPU
  bool x0_ready;
  float *m0op;             // layer 1's M = AX (.) X as an operand: its OWN rows are recomputed every step, its halo rows -- the same
  bool m0_ready;           // constants computed by their owners -- are fetched once (m_tmp itself on one GPU)
  int32_t *pid, *rloc;     // per batch: operand row in A_hat^T's column space (or -1) / local row (clamped) of every member
  int32_t *rlist;          // shards, gss_plan_step_lazy: per member its local row when this shard owns it, -1 otherwise (the top layer's row list)
  float *keep;             // per batch: 1.0 where this shard owns the member
  float *gab;              // [2 * max_batch][d]: the top layer's compact input gradients (one buffer: one all-reduce where one is needed)
  float *bx;               // shards: [max_batch (2 d + 1)] = [E_B | P_B | inv_B], the ONE batch collective of a step (loss.hip gather_batch_kernel)
  float *dex;              // shards: [max_batch d + 1], row-slab sweep (knob loss_slab): the ranks' rows of sum_js de_part + their loss shares
  // halo_recompute (knob, on by default): layer 1's AX / AM are constants, so their boundary rows are fetched ONCE and layer 1's
  // projection runs over own + boundary rows -- layer 2's boundary input rows are then computed here (by the same kernel from the same
  // operands: the owner's bits) instead of exchanged every step.  ax[0] / am[0] / p[0] are operand-sized then.  The trade per boundary
  // row and step: 2 * 2 d * d flops of fp32 MFMA (0.75 ns at d = 128, measured projection rate) against d * 4 bytes over xGMI (1.5 ns
  // at the ~340 GB/s a rank's seven links deliver together, plus the collective's latency): recomputing wins at every size.
  bool recompute;
  bool l0h_ready;          // the boundary rows of AX_0 / AM_0 have been fetched
  int64_t n_coll[3];       // collectives enqueued since the last gss_plan_comm_stats: halo exchanges / batch-row / weight-gradient
  const gss::BatchPrep *prep_pending;   // gss_plan_step on one GPU: the batch preparation rides in the step's first forward SpMM (consumed there)
  int32_t eb_scatter_b;    // > 0 (a full step whose batch was prepared that way): the top layer's projection also scatters the unit-norm
                           // rows of the batch members into E_B through the batch-position map -- no gather launch
  int32_t eb_rows;         // > 0: this step's lazy forward already wrote E_B for a batch of that many rows into the loss workspace (one GPU:
                           // the row-list projection's tile rows are the batch positions), so the loss needs no gather launch
  // overlapped hops (gss_shard_desc a_own / a_halo / at_own / at_halo): the boundary rows of a hop travel on `xs` while the entries
  // that reference the shard's own rows are multiplied on the caller's stream; the boundary-column entries are added afterwards
  const gss_csr *a_own, *a_halo, *at_own, *at_halo;
  // the shard's A_hat transposed in place (gss_shard_desc.a_loc_t; used with halo_recompute at two layers): the last backward hop in
  // scatter-by-owner form, no exchange of u
  const gss_csr *tloc;
  hipStream_t xs;
  hipEvent_t ev_ready, ev_halo;
  // lazy halo (knob lazy_halo; gss_plan_step_lazy on shards): of the top layer's M only the boundary rows that the batch rows this
  // shard owns reference are fetched.  Requests travel as bitmaps over the (static) halo slots -- word-aligned per owner, so their
  // sizes are host constants --; both sides list the set bits in the same order; the per-peer counts come back to the host once per
  // step (ncclSend / ncclRecv take host-side counts).
  struct LazyHalo {
    bool on = false;
    std::vector<int64_t> w_recv_off, w_send_off;   // word offsets per owner / per requester
    int64_t *d_meta = nullptr;                     // device copies: recv_off | w_recv_off | send_off | w_send_off, P + 1 entries each
    uint32_t *needw = nullptr, *reqw = nullptr;    // what I need of each owner / what each requester needs of me
    int32_t *recv_list = nullptr, *send_list = nullptr;   // operand rows the fetched rows land on / own rows to pack, in exchange order
    char *cscratch = nullptr;                      // bits_compact's block counts / bases (both lists use it in turn, on one stream)
    size_t cscratch_bytes = 0;
    int64_t *d_cnt = nullptr;                      // send offsets (P + 1) | recv offsets (P + 1) of this step's exchange
    int64_t *h_cnt = nullptr;                      // pinned host copy
    float *recvbuf = nullptr;                      // [n_halo][d] staging of the fetched rows
    int64_t last_recv = -1, last_send = -1;        // rows moved by the last exchange (gss_plan_lazy_halo_rows)
  } lz, lzt;
  // The request phase of lz depends on the batch ids and the graph only (never on values), so gss_plan_step_lazy runs it FIRST, on the
  // plan's request stream `rq`, while the caller's stream works through layer 1; the 2 (P + 1) counts reach the host by an event-gated
  // async copy and the host waits for that EVENT just before it enqueues the transfer -- the caller's stream is never drained for it
  // (round 5; until then: comm->sync on the caller's stream, a device bubble per step, which kept the subset exchange off over RCCL).
  hipStream_t rq = nullptr;
  hipEvent_t ev_rq_in = nullptr, ev_rq_cnt = nullptr;
  bool lz_pending = false;     // a request phase is in flight on rq
  int64_t n_sync[2] = {0, 0};  // since the last gss_plan_sync_stats: host waits that DRAIN the caller's stream / host waits for an event of rq
  // lzt: the same machinery for A_hat^T's halo, driven by the SENDER: of u -- the operand of the top layer's second backward hop, non-zero
  // only in the batch's neighbourhood (nzbits) -- the owners send the rows that can be non-zero and the bitmap that says which; the receiver
  // sets exactly those bits behind its own rows' in nzbits, so the hop never reads a row that was not sent.  Full and lazy steps alike.
};

// debug knob "sparse_bits_rows": operand rows from which a plan keeps the two bitmaps of the sparsity-aware backward hops (the
// batch-membership bitmap in front of the position map, the non-zero-row bitmap of u); plans created afterwards
// measured with RMAT graphs at B = 2048 (tools/ab_sparse_bits.sh): 60k rows -2.5 %, 120k +2 %, 250k +4.4 %, 450k +6.4 % of a step   [knob sparse_bits_rows, common.h Knobs]

using namespace gss;

namespace {
// RAII pair of events around one launcher call
struct ProfScope {
  gss_plan *p;
  hipStream_t st;
  size_t slot;
  bool on;
  ProfScope(gss_plan *plan, int cls, void *stream) : p(plan), st(as_stream(stream)), slot(0), on(plan->prof_on) {
    if (!on) return;
    if (p->ev_used * 2 + 2 > p->ev.size()) {
      hipEvent_t a, b;
      if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) {
        on = false;
        return;
      }
      p->ev.push_back(a);
      p->ev.push_back(b);
      p->ev_cls.push_back(cls);
    }
    slot = p->ev_used++;
    p->ev_cls[slot] = cls;
    (void)hipEventRecord(p->ev[2 * slot], st);
  }
  ~ProfScope() {
    if (on) (void)hipEventRecord(p->ev[2 * slot + 1], st);
  }
};
#define PROF(cls) ProfScope prof_scope_##__LINE__(p, cls, stream)

// Every carved buffer is followed by a 256-byte guard (kGuardByte everywhere, written once at creation).  No kernel may touch one:
// gss_plan_check_guards reads them back -- a buffer sized for the wrong worst case shows up there instead of silently running into
// its neighbour (what the loss workspace did until round 2)
constexpr size_t kGuardBytes = 256;
constexpr int kGuardByte = 0xA5;
struct Carver {
  size_t off = 0;
  char *base = nullptr;
  std::vector<size_t> *guards = nullptr;   // offsets of the guards, recorded by the carving pass
  template <typename T>
  T *take(size_t count) {
    off = (off + 255) / 256 * 256;
    T *p = base ? reinterpret_cast<T *>(base + off) : nullptr;
    off += sizeof(T) * count;
    off = (off + 255) / 256 * 256;
    if (guards) guards->push_back(off);
    off += kGuardBytes;
    return p;
  }
};

void carve(gss_plan *p, Carver &c) {
  const gss_plan_desc &D = p->desc;
  const size_t n1 = (size_t)(D.n > 0 ? D.n : 1);
  const size_t nd = n1 * D.d;                                  // a local [n][d] buffer
  const size_t nd_a = (p->rows_a ? p->rows_a : 1) * D.d;       // an operand of A_hat: own rows + halo
  const size_t nd_t = (p->rows_t ? p->rows_t : 1) * D.d;       // an operand of A_hat^T
  const size_t bd = (size_t)D.max_batch * D.d;
  const int L = D.num_layers;
  const bool sharded = p->P > 1;
  const size_t n_send = (size_t)std::max(p->halo_a.n_send, p->halo_t.n_send);
  p->sendbuf = sharded ? c.take<float>((n_send ? n_send : 1) * D.d) : nullptr;
  auto carve_lazy = [&](gss_plan::LazyHalo &z, const gss_plan::Halo &h) {
    if (!z.on) return;
    const size_t P1 = (size_t)p->P + 1;
    z.d_meta = c.take<int64_t>(4 * P1);
    z.d_cnt = c.take<int64_t>(2 * P1);
    z.needw = c.take<uint32_t>((size_t)z.w_recv_off.back() + 1);
    z.reqw = c.take<uint32_t>((size_t)z.w_send_off.back() + 1);
    z.recv_list = c.take<int32_t>((size_t)h.n_halo + 1);
    z.send_list = c.take<int32_t>((size_t)h.n_send + 1);
    z.recvbuf = c.take<float>(((size_t)h.n_halo + 1) * D.d);
    z.cscratch_bytes = std::max(bits_compact_scratch_bytes(p->P, z.w_send_off.data()), bits_compact_scratch_bytes(p->P, z.w_recv_off.data()));
    z.cscratch = c.take<char>(z.cscratch_bytes);
  };
  carve_lazy(p->lz, p->halo_a);
  carve_lazy(p->lzt, p->halo_t);
  p->x0op = sharded ? c.take<float>(nd_a) : const_cast<float *>(p->x);
  const bool maps = sharded || D.node_map != nullptr;   // per-batch index maps are needed
  p->pid = maps ? c.take<int32_t>(D.max_batch) : nullptr;
  p->rloc = maps ? c.take<int32_t>(D.max_batch) : nullptr;
  p->keep = sharded ? c.take<float>(D.max_batch) : nullptr;
  p->rlist = sharded ? c.take<int32_t>(D.max_batch) : nullptr;
  p->ax.assign(L, nullptr);
  p->am.assign(L, nullptr);
  p->p.assign(L, nullptr);
  p->xin.assign(L, nullptr);
  for (int l = 0; l < L; ++l) {
    const size_t nd_l = (l == 0 && p->recompute) ? nd_a : nd;   // halo_recompute: layer 1's AX / AM / P also on the boundary rows
    p->ax[l] = c.take<float>(nd_l);
    p->am[l] = c.take<float>(nd_l);
    p->p[l] = c.take<float>(nd_l);
    p->xin[l] = l == 0 ? p->x0op : c.take<float>(nd_a);       // layer inputs are operands of A_hat
  }
  p->m_tmp = c.take<float>(nd_a);
  p->m0op = sharded ? c.take<float>(nd_a) : p->m_tmp;
  p->ax0[0] = p->ax[0];
  p->am0[0] = p->am[0];
  const bool pipe = D.pipeline_layer1 != 0;
  p->ax0[1] = pipe ? c.take<float>(nd) : nullptr;
  p->am0[1] = pipe ? c.take<float>(nd) : nullptr;
  p->m_side = pipe ? c.take<float>(nd_a) : nullptr;
  p->x_last = c.take<float>(nd);
  p->inv_den = c.take<float>(n1);
  if (L > 1) {
    p->g_ax = c.take<float>(nd);
    p->g_am = c.take<float>(nd_t);                             // operands of A_hat^T
    p->u = c.take<float>(nd_t);
    p->t = c.take<float>(nd);
    p->dp = c.take<float>(p->tloc ? nd_a : nd);              // (tloc: the last hop's dP also on the boundary rows)
    p->gx[0] = L > 2 ? c.take<float>(nd) : nullptr;
    p->gx[1] = L > 2 ? c.take<float>(nd) : nullptr;
  } else {
    p->g_ax = p->g_am = p->u = p->t = p->dp = p->gx[0] = p->gx[1] = nullptr;
  }
  p->de_b = c.take<float>(bd);
  p->dx_b = c.take<float>(bd);
  p->dp_b = c.take<float>(bd);
  p->gab = L > 1 ? c.take<float>(2 * bd) : nullptr;            // gax_b = gab, gam_b = gab + b * d: one all-reduce
  p->bx = sharded ? c.take<float>(2 * bd + (size_t)D.max_batch) : nullptr;
  p->dex = sharded ? c.take<float>(bd + 1) : nullptr;
  p->gax_b = p->gam_b = nullptr;
  p->pos = L > 1 ? c.take<int32_t>(p->rows_t ? p->rows_t : 1) : nullptr;
  // from sparse_bits_rows operand rows on: the sparse SpMM tests a bitmap before the 4-byte-per-node map (zero-initialised slab)
  // (the sizing pass carves from a null base: conditions must not look at the pointers it hands out)
  // (with the lazy halo the bitmaps are kept whatever the size of THIS shard: whether a hop's exchange is a subset is a decision all
  // ranks must share, and the sender-driven subset of u's halo is read off nzbits)
  const bool bitmaps = L > 1 && (p->rows_t >= (size_t)K().sparse_bits_rows || p->lzt.on || p->lz.on);
  p->posbits = bitmaps ? c.take<uint32_t>(p->rows_t / 32 + 1) : nullptr;
  p->nzbits = bitmaps ? c.take<uint32_t>(p->rows_t / 32 + 1) : nullptr;
  // (a shard needs the peers' requests to know which of its rows anything reads: with the lazy halo only)
  p->needbits = (bitmaps && (!sharded || p->lz.on)) ? c.take<uint32_t>((p->rows_a ? p->rows_a : 1) / 32 + 1) : nullptr;
  p->w1t = c.take<float>((size_t)D.d * D.d);
  p->w2t = c.take<float>((size_t)D.d * D.d);
  const size_t cnt[4] = {(size_t)D.d * D.d, (size_t)D.d, (size_t)D.d * D.d, (size_t)D.d};
  for (int k = 0; k < 4; ++k) {
    p->adam_m[k] = c.take<float>(cnt[k]);
    p->adam_v[k] = c.take<float>(cnt[k]);
  }
  p->loss_ws = c.take<char>(loss_workspace_bytes_max(D.max_batch, D.d, p->P));   // not monotone in the batch size: the worst case over 1..max_batch
  p->wgrad_ws = c.take<char>(sizeof(float) * (size_t)p->wg_total * ((size_t)D.d * 2 * D.d + D.d));
}

bool halo_ok(const gss_halo_desc &h, int P, int rank) {
  if (!h.h_recv_off || !h.h_send_off) return false;
  if (h.h_recv_off[0] != 0 || h.h_send_off[0] != 0) return false;
  for (int q = 0; q < P; ++q)
    if (h.h_recv_off[q + 1] < h.h_recv_off[q] || h.h_send_off[q + 1] < h.h_send_off[q]) return false;
  if (h.h_recv_off[rank + 1] != h.h_recv_off[rank] || h.h_send_off[rank + 1] != h.h_send_off[rank]) return false;  // nothing from / to itself
  return h.h_send_off[P] == 0 || h.d_send_rows != nullptr;
}

void take_halo(gss_plan::Halo &dst, const gss_halo_desc &h, int P) {
  dst.recv_off.assign(h.h_recv_off, h.h_recv_off + P + 1);
  dst.send_off.assign(h.h_send_off, h.h_send_off + P + 1);
  dst.n_halo = dst.recv_off[(size_t)P];
  dst.n_send = dst.send_off[(size_t)P];
  dst.d_send_rows = h.d_send_rows;
}

// The job-wide knobs steer WHICH collectives a step enqueues: ranks that disagree would wait for each other in different collectives
// (a hang, not an error).  Every rank contributes its values to one all-gather at plan creation and fails BY NAME when they differ
// (VERDICT round 5, item 8).  Collective: every rank of the job creates its plan, as gss_plan_create_sharded requires anyway.
static int check_job_knobs(gss_comm *comm, const Knobs &k) {
  static const char *const kNames[4] = {"lazy_halo", "lazy_halo_u", "halo_recompute", "loss_slab"};
  const int P = comm->world, rank = comm->rank;
  const int32_t mine[4] = {k.lazy_halo, k.lazy_halo_u, k.halo_recompute, k.loss_slab};
  std::vector<int32_t> all((size_t)P * 4, 0);
  int32_t *dbuf = nullptr;
  hipStream_t st = nullptr;
  GSS_HIP(hipMalloc((void **)&dbuf, sizeof(int32_t) * 4 * (size_t)P));
  hipError_t e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  int rc = GSS_OK;
  if (e == hipSuccess) e = hipMemcpyAsync(dbuf + 4 * rank, mine, sizeof(mine), hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);     // `mine` is a stack array
  if (e == hipSuccess) rc = comm->all_gather(dbuf + 4 * rank, dbuf, sizeof(mine), st);
  if (e == hipSuccess && rc == GSS_OK) rc = comm->sync(st, 300.0);
  if (e == hipSuccess && rc == GSS_OK) e = hipMemcpy(all.data(), dbuf, sizeof(int32_t) * 4 * (size_t)P, hipMemcpyDeviceToHost);
  if (st) (void)hipStreamDestroy(st);
  (void)hipFree(dbuf);
  if (e != hipSuccess) return fail(GSS_EHIP, "plan_create_sharded: comparing the job-wide knobs -> %s", hipGetErrorString(e));
  if (rc != GSS_OK) return rc;
  for (int j = 0; j < 4; ++j)
    for (int r = 0; r < P; ++r)
      if (all[(size_t)r * 4 + j] != mine[j])
        return fail(GSS_EINVAL, "plan_create_sharded: knob \"%s\" is %d on rank %d and %d on rank %d -- a job-wide knob must have the same value on "
                    "every rank (set it through GSS_OPTIONS, which every rank of a torch.distributed job inherits)", kNames[j], mine[j], rank,
                    all[(size_t)r * 4 + j], r);
  return GSS_OK;
}

int plan_create_impl(gss_plan **out, const gss_plan_desc *desc, const gss_shard_desc *shard, gss_comm *comm, const gss_csr *a,
                     const gss_csr *at, const gss_plan_io *io) {
  GSS_REQUIRE(out && desc && a && io, "plan_create: null argument");
  GSS_REQUIRE(io->w1 && io->b1 && io->w2 && io->b2 && io->loss && io->gw1 && io->gb1 && io->gw2 && io->gb2,
              "plan_create: null pointer in gss_plan_io");
  GSS_REQUIRE((io->x && io->emb) || desc->n == 0, "plan_create: null x / emb in gss_plan_io");  // an empty shard has neither
  if (int rc = check_d(desc->d)) return rc;
  const int P = shard ? shard->world : 1;
  const int rank = shard ? shard->rank : 0;
  GSS_REQUIRE(desc->num_layers >= 1 && desc->num_layers <= 64, "plan_create: num_layers=%d", desc->num_layers);
  GSS_REQUIRE(desc->n > 0 || (P > 1 && desc->n == 0), "plan_create: n=%d", desc->n);
  int64_t n_global = desc->n;
  int64_t ha = 0, hat = 0;
  if (shard) {
    GSS_REQUIRE(P >= 1 && rank >= 0 && rank < P && shard->h_bounds, "plan_create_sharded: bad shard descriptor");
    GSS_REQUIRE(P == 1 || comm, "plan_create_sharded: world %d needs a communicator", P);
    GSS_REQUIRE(!comm || (comm->world == P && comm->rank == rank), "plan_create_sharded: communicator is rank %d of %d, shard is %d of %d",
                comm ? comm->rank : -1, comm ? comm->world : -1, rank, P);
    GSS_REQUIRE(shard->h_bounds[0] == 0, "plan_create_sharded: bounds[0] must be 0");
    for (int r = 0; r < P; ++r) GSS_REQUIRE(shard->h_bounds[r + 1] >= shard->h_bounds[r], "plan_create_sharded: bounds not monotone at %d", r);
    n_global = shard->h_bounds[P];
    GSS_REQUIRE(n_global > 0 && n_global < (int64_t)INT32_MAX, "plan_create_sharded: N=%lld", (long long)n_global);
    GSS_REQUIRE(shard->h_bounds[rank + 1] - shard->h_bounds[rank] == desc->n, "plan_create_sharded: desc->n=%d is not this shard's row count",
                desc->n);
    GSS_REQUIRE(!desc->pipeline_layer1 || P == 1, "plan_create_sharded: pipeline_layer1 is a single-GPU option");
    if (P > 1) {
      GSS_REQUIRE(halo_ok(shard->halo_a, P, rank), "plan_create_sharded: malformed halo descriptor of A_hat");
      ha = shard->halo_a.h_recv_off[P];
      if (desc->num_layers > 1) {
        GSS_REQUIRE(halo_ok(shard->halo_at, P, rank) && shard->d_gid2op_t, "plan_create_sharded: malformed halo descriptor of A_hat^T");
        hat = shard->halo_at.h_recv_off[P];
      }
      GSS_REQUIRE(desc->n + ha < (int64_t)INT32_MAX && desc->n + hat < (int64_t)INT32_MAX, "plan_create_sharded: operand rows overflow int32");
    }
  }
  GSS_REQUIRE(desc->max_batch >= 1 && (int64_t)desc->max_batch <= n_global, "plan_create: max_batch=%d out of [1, N=%lld]", desc->max_batch,
              (long long)n_global);
  GSS_REQUIRE(a->n_rows == desc->n && a->n_cols == desc->n + ha, "plan_create: A is %d x %d, expected %d x %lld", a->n_rows, a->n_cols, desc->n,
              (long long)(desc->n + ha));
  GSS_REQUIRE(desc->num_layers == 1 || (at && at->n_rows == desc->n && at->n_cols == desc->n + hat),
              "plan_create: A^T missing or mis-shaped (needed for num_layers >= 2)");
  if (shard && P > 1) {
    const gss_csr *pairs[2][3] = {{shard->a_own, shard->a_halo, a}, {shard->at_own, shard->at_halo, at}};
    for (int m = 0; m < 2; ++m) {
      const gss_csr *own = pairs[m][0], *hal = pairs[m][1], *full = pairs[m][2];
      GSS_REQUIRE((own == nullptr) == (hal == nullptr), "plan_create_sharded: the own-column and boundary-column halves of a matrix go together");
      if (!own || !full) continue;
      GSS_REQUIRE(own->n_rows == full->n_rows && hal->n_rows == full->n_rows && own->n_cols == full->n_cols && hal->n_cols == full->n_cols &&
                      own->nnz + hal->nnz == full->nnz,
                  "plan_create_sharded: the split halves of %s do not add up to the matrix (%lld + %lld entries of %lld)", m ? "A_hat^T" : "A_hat",
                  (long long)own->nnz, (long long)hal->nnz, (long long)full->nnz);
    }
  }
  if (P > 1)
    if (int rc = check_job_knobs(comm, g_knobs)) return rc;
  gss_plan *p = new gss_plan();
  p->knobs = g_knobs;
  KnobScope knob_scope(&p->knobs);
  p->desc = *desc;
  p->comm = P > 1 ? comm : nullptr;
  p->P = P;
  p->rank = rank;
  p->lo = shard ? (int)shard->h_bounds[rank] : 0;
  if (shard) p->h_bounds.assign(shard->h_bounds, shard->h_bounds + P + 1);
  if (P > 1) {
    take_halo(p->halo_a, shard->halo_a, P);
    if (desc->num_layers > 1) take_halo(p->halo_t, shard->halo_at, P);
  }
  p->rows_a = (size_t)desc->n + (size_t)ha;
  p->rows_t = (size_t)desc->n + (size_t)hat;
  p->gid2op_t = (P > 1 && desc->num_layers > 1) ? shard->d_gid2op_t : nullptr;
  p->a_own = p->a_halo = p->at_own = p->at_halo = nullptr;
  p->tloc = nullptr;
  p->xs = nullptr;
  p->ev_ready = p->ev_halo = nullptr;
  if (P > 1 && shard->a_own && shard->a_halo) {
    p->a_own = shard->a_own;
    p->a_halo = shard->a_halo;
  }
  if (P > 1 && desc->num_layers > 1 && shard->at_own && shard->at_halo) {
    p->at_own = shard->at_own;
    p->at_halo = shard->at_halo;
  }
  p->x0_ready = false;
  p->m0_ready = false;
  p->a = a;
  p->at = at;
  p->x = io->x;
  p->w1 = io->w1;
  p->b1 = io->b1;
  p->w2 = io->w2;
  p->b2 = io->b2;
  p->emb = io->emb;
  p->loss = io->loss;
  p->grad[0] = io->gw1;
  p->grad[1] = io->gb1;
  p->grad[2] = io->gw2;
  p->grad[3] = io->gb2;
  p->step = 0;
  p->layer1_valid = false;
  p->wt_valid = false;

  p->prof_on = false;
  p->ev_used = 0;
  for (int k = 0; k < GSS_PROF_CLASSES; ++k) {
    p->prof_ms[k] = 0.0;
    p->prof_cnt[k] = 0;
  }
  {
    // every rank takes the same decision: it depends on the knobs, the transport and the size of the whole graph only.
    //   lz  (the top layer's M in a lazy step, receiver-driven): automatic from 262,144 nodes on over the host-side transports.  Over
    //       RCCL it is OPT-IN (knob lazy_halo = 1): its request phase enqueues ncclSend / ncclRecv on the plan's request
    //       stream while layer 1's exchanges use the caller's stream of the SAME communicator -- RCCL serialises the operations of one
    //       communicator, so the overlap round 5 built may not materialise, and the configuration has never run between two devices
    //       (tests/test_gpu_dist.py's two-GPU cases are skipped on the one-GPU box).  It becomes the default there once a run on >= 2
    //       GPUs has executed it and matched lazy_halo = 0 bit for bit;
    //   lzt (u in the top layer's second backward hop, sender-driven: the bitmap is the sparse hop's OUTPUT, so the counts cannot be known
    //       ahead and the host drains the stream once per step for them): automatic on the host-side transports only; over RCCL the hop
    //       runs exchange-free on the shard's A_hat transposed in place instead (tloc below) -- opt-in there (knob lazy_halo_u = 1).
    const int knob = K().lazy_halo, knob_u = K().lazy_halo_u;
    const bool rccl = comm && comm->device_transport();
    p->lz.on = P > 1 && desc->num_layers > 1 && (knob == 1 || (knob < 0 && n_global >= 262144 && !rccl));
    p->lzt.on = p->lz.on && spmm_sparse_available() && (knob_u == 1 || (knob_u < 0 && (knob == 1 || !rccl)));   // (the sparse first backward hop writes nzbits; knob spmm_variant is process-wide)
    auto word_offsets = [&](gss_plan::LazyHalo &z, const gss_plan::Halo &h) {
      if (!z.on) return;
      z.w_recv_off.assign((size_t)P + 1, 0);
      z.w_send_off.assign((size_t)P + 1, 0);
      for (int q = 0; q < P; ++q) {
        z.w_recv_off[(size_t)q + 1] = z.w_recv_off[(size_t)q] + (h.recv_off[(size_t)q + 1] - h.recv_off[(size_t)q] + 31) / 32;
        z.w_send_off[(size_t)q + 1] = z.w_send_off[(size_t)q] + (h.send_off[(size_t)q + 1] - h.send_off[(size_t)q] + 31) / 32;
      }
    };
    word_offsets(p->lz, p->halo_a);
    word_offsets(p->lzt, p->halo_t);
  }
  {
    const int knob = K().halo_recompute;
    p->recompute = P > 1 && desc->num_layers > 1 && knob != 0;   // (automatic = on: what it trades is priced below)
    // (used where the hop it replaces would move a WHOLE halo: graphs below 262,144 nodes -- a latency-bound collective -- and big graphs
    //  without the subset exchange of u (RCCL's default).  Where the subset exchange is on, u's few MB are cheaper than the boundary
    //  rows' share of the bottom layer's weight gradient: RMAT 10M, world 8, all ranks on one GPU: 119.6 -> 128.0 ms per step with it)
    const bool tloc_pays = n_global < 262144 || !p->lzt.on;
    if (p->recompute && desc->num_layers >= 2 && shard->a_loc_t && tloc_pays) {
      const gss_csr *t = shard->a_loc_t;
      if (!(t->n_rows == desc->n + ha && t->n_cols == desc->n && t->nnz == a->nnz)) {
        const int r_ = t->n_rows, c_ = t->n_cols;
        const long long z_ = (long long)t->nnz;
        delete p;
        return fail(GSS_EINVAL, "plan_create_sharded: a_loc_t is %d x %d with %lld entries, expected the transpose of A_hat's shard (%lld x %d, %lld)",
                    r_, c_, z_, (long long)(desc->n + ha), desc->n, (long long)a->nnz);
      }
      p->tloc = t;
    }
    p->l0h_ready = false;
    p->n_coll[0] = p->n_coll[1] = p->n_coll[2] = 0;
    p->eb_rows = 0;
    p->prep_pending = nullptr;
    p->eb_scatter_b = 0;
  }
  // slices of the shared weight-gradient partial buffer: the top layer's batch rows + every layer below over its rows (the bottom layer
  // over own + boundary rows with tloc)
  p->wg_total = wgrad_slices_max(desc->max_batch, desc->d) + (desc->num_layers - 1) * wgrad_slices(desc->n, desc->d);
  if (p->tloc) p->wg_total += wgrad_slices((int32_t)p->rows_a, desc->d);
  Carver sizing;
  carve(p, sizing);
  p->slab_bytes = sizing.off + 256;
  hipError_t e = hipMalloc((void **)&p->slab, p->slab_bytes);
  if (e != hipSuccess) {
    const size_t want = p->slab_bytes;
    delete p;
    return fail(GSS_ENOMEM, "plan_create: hipMalloc(%zu bytes) -> %s", want, hipGetErrorString(e));
  }
  e = hipMemset(p->slab, 0, p->slab_bytes);
  if (e != hipSuccess) {
    (void)hipFree(p->slab);
    delete p;
    return fail(GSS_EHIP, "plan_create: hipMemset -> %s", hipGetErrorString(e));
  }
  Carver real;
  real.base = p->slab;
  real.guards = &p->guard_off;
  carve(p, real);
  if (real.off != sizing.off) {  // the two passes must carve the same sequence (a condition that looked at a pointer would not)
    const size_t a_ = sizing.off, b_ = real.off;
    (void)hipFree(p->slab);
    delete p;
    return fail(GSS_EINVAL, "plan_create: internal error, sizing pass %zu bytes but carving pass %zu", a_, b_);
  }
  if (desc->n == 0) {  // an empty shard: launchers still want non-null operands (they move zero rows)
    if (!p->x) p->x = p->x_last;
    if (!p->emb) p->emb = p->x_last;
  }
  p->side = nullptr;
  p->ev_main_ready = p->ev_side_done = nullptr;
  p->cur0 = 0;
  p->prefetched = false;
  if (desc->pipeline_layer1) {
    e = hipStreamCreateWithFlags(&p->side, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&p->ev_main_ready, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&p->ev_side_done, hipEventDisableTiming);
    if (e != hipSuccess) {
      (void)hipFree(p->slab);
      delete p;
      return fail(GSS_EHIP, "plan_create: side stream/events -> %s", hipGetErrorString(e));
    }
  }
  if (p->a_own || p->at_own) {
    e = hipStreamCreateWithFlags(&p->xs, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&p->ev_ready, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&p->ev_halo, hipEventDisableTiming);
    if (e != hipSuccess) {
      (void)hipFree(p->slab);
      delete p;
      return fail(GSS_EHIP, "plan_create: exchange stream/events -> %s", hipGetErrorString(e));
    }
  }
  if (p->lz.on) {
    e = hipStreamCreateWithFlags(&p->rq, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&p->ev_rq_in, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&p->ev_rq_cnt, hipEventDisableTiming);
    if (e != hipSuccess) {
      (void)hipFree(p->slab);
      delete p;
      return fail(GSS_EHIP, "plan_create: request stream/events -> %s", hipGetErrorString(e));
    }
  }
  if (p->pos) {
    e = hipMemset(p->pos, 0xff, sizeof(int32_t) * (p->rows_t ? p->rows_t : 1));  // all -1
    if (e != hipSuccess) {
      (void)hipFree(p->slab);
      delete p;
      return fail(GSS_EHIP, "plan_create: hipMemset(pos) -> %s", hipGetErrorString(e));
    }
  }
  for (size_t g : p->guard_off) {
    e = hipMemset(p->slab + g, kGuardByte, kGuardBytes);
    if (e != hipSuccess) {
      (void)hipFree(p->slab);
      delete p;
      return fail(GSS_EHIP, "plan_create: hipMemset(guard) -> %s", hipGetErrorString(e));
    }
  }
  for (int m = 0; m < 2; ++m) {
    gss_plan::LazyHalo &z = m ? p->lzt : p->lz;
    const gss_plan::Halo &h = m ? p->halo_t : p->halo_a;
    if (!z.on) continue;
    const size_t P1 = (size_t)P + 1;
    std::vector<int64_t> meta;
    meta.insert(meta.end(), h.recv_off.begin(), h.recv_off.end());
    meta.insert(meta.end(), z.w_recv_off.begin(), z.w_recv_off.end());
    meta.insert(meta.end(), h.send_off.begin(), h.send_off.end());
    meta.insert(meta.end(), z.w_send_off.begin(), z.w_send_off.end());
    e = hipMemcpy(z.d_meta, meta.data(), sizeof(int64_t) * 4 * P1, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipHostMalloc((void **)&z.h_cnt, sizeof(int64_t) * 2 * P1, hipHostMallocDefault);
    if (e != hipSuccess) {
      if (p->lz.h_cnt) (void)hipHostFree(p->lz.h_cnt);
      (void)hipFree(p->slab);
      delete p;
      return fail(GSS_EHIP, "plan_create: lazy-halo tables -> %s", hipGetErrorString(e));
    }
  }
  if (p->nzbits && p->rows_t > (size_t)desc->n) {
    // rows a peer owns: this shard cannot know whether they are zero
    if (int rc = bits_fill(p->nzbits, desc->n, (int64_t)p->rows_t, nullptr)) {
      (void)hipFree(p->slab);
      delete p;
      return rc;
    }
  }
  // the memsets above ran on the null stream; the caller's stream may be a non-blocking one that does not order behind it
  e = hipStreamSynchronize(nullptr);
  if (e != hipSuccess) {
    (void)hipFree(p->slab);
    delete p;
    return fail(GSS_EHIP, "plan_create: hipStreamSynchronize -> %s", hipGetErrorString(e));
  }
  *out = p;
  return GSS_OK;
}

// ---- exchange points of a sharded plan (no-ops on one GPU) ------------------------------------------------------
// C1, boundary form: `op` is an operand whose own rows [0, n) are complete; fetch its halo rows [n, n + n_halo) from their
// owners and serve the peers the rows they reference
int plan_halo(gss_plan *p, const gss_plan::Halo &h, float *op, void *stream) {
  if (p->P == 1) return GSS_OK;
  PROF(GSS_PROF_COMM);
  p->n_coll[0] += 1;
  const int d = p->desc.d;
  if (h.n_send > 0)
    if (int rc = pack_rows(d, op, h.d_send_rows, h.n_send, p->sendbuf, stream)) return rc;
  return p->comm->exchange_rows(p->sendbuf, h.send_off.data(), op + (size_t)p->desc.n * d, h.recv_off.data(), d, as_stream(stream));
}

// A hop that reads a SUBSET of its operand's boundary rows (knob lazy_halo).  Which rows is known on one side only:
//   need_rows != NULL  (the top layer's M in a lazy step): the receiver knows -- the columns of the rows `need_rows[0..b)` of A_hat it
//                      evaluates (-1: a peer's member); it marks them over its halo slots and sends the bitmaps to the owners;
//   src_bits  != NULL  (u, the operand of the top layer's second backward hop): the sender knows -- its rows that can be non-zero
//                      (nzbits); it sends the bitmaps over its send slots to the receivers, which then set exactly those bits behind their
//                      own rows' in the same bitmap, so the hop's non-zero-row filter never lets it read a row that was not sent.
// The bitmaps are word-aligned per peer, their sizes host constants; both sides list the set bits in the same order; the per-peer row counts
// of the step come back to the host once (ncclSend / ncclRecv take host-side counts).  Boundary rows that are not fetched keep whatever
// they held: nothing reads them.
// Request phase: who needs what.  Afterwards z.send_list / z.recv_list and the host-side counts z.h_cnt describe the transfer.
// enqueue only: bitmaps, their exchange, the two lists and the async copy of the counts to the pinned host buffer, all on `stream`
int plan_halo_requests_enqueue(gss_plan *p, gss_plan::LazyHalo &z, const gss_plan::Halo &h, const gss_csr *a, const int32_t *need_rows, int32_t b,
                               const uint32_t *src_bits, void *stream) {
  PROF(GSS_PROF_COMM);
  p->n_coll[0] += 1;
  const int P = p->P, n = p->desc.n;
  const size_t P1 = (size_t)P + 1;
  hipStream_t st = as_stream(stream);
  const int64_t *d_recv_off = z.d_meta, *d_wrecv_off = z.d_meta + P1, *d_send_off = z.d_meta + 2 * P1, *d_wsend_off = z.d_meta + 3 * P1;
  if (need_rows) {
    GSS_HIP(hipMemsetAsync(z.needw, 0, sizeof(uint32_t) * ((size_t)z.w_recv_off.back() + 1), st));
    if (h.n_halo > 0)
      if (int rc = halo_need_mark(a, need_rows, b, n, P, d_recv_off, d_wrecv_off, z.needw, stream)) return rc;
    // my requests are grouped by owner (the layout I receive rows in); what arrives is grouped by requester (the layout I send rows in)
    if (int rc = p->comm->exchange_rows(reinterpret_cast<const float *>(z.needw), z.w_recv_off.data(), reinterpret_cast<float *>(z.reqw),
                                        z.w_send_off.data(), 1, st))
      return rc;
  } else {
    if (int rc = send_slot_bits(src_bits, h.d_send_rows, P, d_send_off, d_wsend_off, z.w_send_off.back(), z.reqw, stream)) return rc;
    if (int rc = p->comm->exchange_rows(reinterpret_cast<const float *>(z.reqw), z.w_send_off.data(), reinterpret_cast<float *>(z.needw),
                                        z.w_recv_off.data(), 1, st))
      return rc;
  }
  if (int rc = bits_compact(z.reqw, P, d_wsend_off, z.w_send_off.data(), d_send_off, h.d_send_rows, 0, z.send_list, z.d_cnt, stream, z.cscratch, z.cscratch_bytes)) return rc;
  if (int rc = bits_compact(z.needw, P, d_wrecv_off, z.w_recv_off.data(), d_recv_off, nullptr, n, z.recv_list, z.d_cnt + P1, stream, z.cscratch, z.cscratch_bytes)) return rc;
  GSS_HIP(hipMemcpyAsync(z.h_cnt, z.d_cnt, sizeof(int64_t) * 2 * P1, hipMemcpyDeviceToHost, st));
  return GSS_OK;
}

// the counts are on the host: check them
int plan_halo_requests_counts(gss_plan *p, gss_plan::LazyHalo &z, const gss_plan::Halo &h) {
  const int P = p->P;
  const int64_t *send_off = z.h_cnt, *recv_off = z.h_cnt + (size_t)P + 1;
  GSS_REQUIRE(send_off[P] >= 0 && send_off[P] <= h.n_send && recv_off[P] >= 0 && recv_off[P] <= h.n_halo,
              "lazy halo: %lld rows to send of %lld, %lld to fetch of %lld", (long long)send_off[P], (long long)h.n_send, (long long)recv_off[P],
              (long long)h.n_halo);
  z.last_send = send_off[P];
  z.last_recv = recv_off[P];
  return GSS_OK;
}

// Sender-driven form (u): the bitmap is the sparse hop's output of THIS step, the transfer follows at once -- the host waits for the
// caller's stream (with the communicator's watchdog): a drain per step, counted in n_sync[0]
int plan_halo_requests(gss_plan *p, gss_plan::LazyHalo &z, const gss_plan::Halo &h, const gss_csr *a, const int32_t *need_rows, int32_t b,
                       const uint32_t *src_bits, void *stream) {
  if (int rc = plan_halo_requests_enqueue(p, z, h, a, need_rows, b, src_bits, stream)) return rc;
  p->n_sync[0] += 1;
  if (int rc = p->comm->sync(as_stream(stream), 300.0)) return rc;          // the counts of this step
  return plan_halo_requests_counts(p, z, h);
}

// Receiver-driven form (the top layer's M), ahead of time: the request phase on the plan's request stream, behind everything the caller's
// stream holds so far (the batch preparation that wrote need_rows; the previous step's readers of the lists)
int plan_lazy_requests_begin(gss_plan *p, const int32_t *need_rows, int32_t b, void *stream) {
  GSS_HIP(hipEventRecord(p->ev_rq_in, as_stream(stream)));
  GSS_HIP(hipStreamWaitEvent(p->rq, p->ev_rq_in, 0));
  if (int rc = plan_halo_requests_enqueue(p, p->lz, p->halo_a, p->a, need_rows, b, nullptr, p->rq)) return rc;
  GSS_HIP(hipEventRecord(p->ev_rq_cnt, p->rq));
  p->lz_pending = true;
  return GSS_OK;
}

// ... and its end: the host waits for the EVENT behind the counts (polled, with the communicator's error poll and a deadline -- never a
// blocking call, never the caller's stream, which keeps running what it has), the caller's stream orders itself behind the lists
int plan_lazy_requests_end(gss_plan *p, void *stream) {
  GSS_REQUIRE(p->lz_pending, "lazy halo: no request phase in flight");
  p->lz_pending = false;
  p->n_sync[1] += 1;
  const auto t0 = std::chrono::steady_clock::now();
  for (int spins = 0;; ++spins) {
    const hipError_t q = hipEventQuery(p->ev_rq_cnt);
    if (q == hipSuccess) break;
    if (q != hipErrorNotReady) return fail(GSS_EHIP, "lazy halo: hipEventQuery -> %s", hipGetErrorString(q));
    if ((spins & 63) == 63) {
      if (int rc = p->comm->check_async()) return rc;
      const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      if (el > 300.0) {
        p->comm->abort();
        return fail(GSS_ETIMEOUT, "lazy halo: rank %d of %d waited %.0f s for the request exchange (a peer stopped taking part?); communicator aborted",
                    p->rank, p->P, el);
      }
    }
    if (spins < 2000) std::this_thread::yield();
    else std::this_thread::sleep_for(std::chrono::microseconds(spins < 20000 ? 20 : 500));
  }
  GSS_HIP(hipStreamWaitEvent(as_stream(stream), p->ev_rq_cnt, 0));
  return plan_halo_requests_counts(p, p->lz, p->halo_a);
}

// Transfer phase: the listed rows of `op` to the peers, theirs into the listed boundary rows.  set_bits (sender-driven form): the
// boundary rows' bits become exactly the rows that arrive.
int plan_halo_transfer(gss_plan *p, gss_plan::LazyHalo &z, const gss_plan::Halo &h, float *op, uint32_t *set_bits, void *stream) {
  PROF(GSS_PROF_COMM);
  p->n_coll[0] += 1;
  const int P = p->P, d = p->desc.d, n = p->desc.n;
  const int64_t *send_off = z.h_cnt, *recv_off = z.h_cnt + (size_t)P + 1;
  if (int rc = pack_rows(d, op, z.send_list, send_off[P], p->sendbuf, stream)) return rc;
  if (int rc = p->comm->exchange_rows(p->sendbuf, send_off, z.recvbuf, recv_off, d, as_stream(stream))) return rc;
  if (set_bits) {
    // (atomics: the word that straddles the own rows is shared with them)
    if (int rc = bits_clear(set_bits, n, n + h.n_halo, stream)) return rc;
    if (int rc = bits_set_list(set_bits, z.recv_list, recv_off[P], stream)) return rc;
  }
  return unpack_rows(d, z.recvbuf, z.recv_list, recv_off[P], op, stream);
}

// One hop over an operand whose boundary rows come from the peers.  Plain form: exchange, then `full` (the product over the shard's
// whole CSR).  Overlapped form (the matrix was given split, gss_shard_desc): the exchange runs on p->xs while `own` multiplies the
// entries that reference the shard's own rows -- complete as soon as the producing kernel is --, then `rest` adds the boundary-column
// entries and runs the epilogue.  Both streams are re-joined before `rest`, so everything after the hop is ordered as before.
template <typename Full, typename Own, typename Rest>
int plan_hop(gss_plan *p, const gss_plan::Halo &h, bool overlapped, float *op, void *stream, Full full, Own own, Rest rest,
             const int32_t *need_rows = nullptr, int32_t need_b = 0, uint32_t *src_bits = nullptr) {
  // need_rows / src_bits: a subset of the boundary rows; h is halo_a with need_rows (whose request phase the caller has run: the
  // top layer's needed-row bitmap comes out of it), halo_t with src_bits
  (void)need_b;
  auto exchange = [&](void *st) {
    if (need_rows) return plan_halo_transfer(p, p->lz, h, op, nullptr, st);
    if (src_bits) {
      if (int rc = plan_halo_requests(p, p->lzt, h, p->at, nullptr, 0, src_bits, st)) return rc;
      return plan_halo_transfer(p, p->lzt, h, op, src_bits, st);
    }
    return plan_halo(p, h, op, st);
  };
  if (p->P == 1 || !overlapped) {
    if (p->P > 1)
      if (int rc = exchange(stream)) return rc;
    return full();
  }
  GSS_HIP(hipEventRecord(p->ev_ready, as_stream(stream)));     // the operand's own rows are complete here
  GSS_HIP(hipStreamWaitEvent(p->xs, p->ev_ready, 0));
  if (int rc = own()) return rc;                               // enqueued first: the in-process backend blocks the host in the exchange
  if (int rc = exchange(p->xs)) return rc;
  GSS_HIP(hipEventRecord(p->ev_halo, p->xs));
  GSS_HIP(hipStreamWaitEvent(as_stream(stream), p->ev_halo, 0));
  return rest();
}

int plan_allreduce(gss_plan *p, float *buf, size_t count, void *stream) {
  if (p->P == 1 || count == 0) return GSS_OK;
  PROF(GSS_PROF_COMM_BATCH);
  p->n_coll[1] += 1;
  return p->comm->all_reduce_sum(&buf, &count, 1, as_stream(stream));
}

// C2: the four weight-gradient tensors summed over the shards, one fused collective
int plan_allreduce_grads(gss_plan *p, void *stream) {
  if (p->P == 1) return GSS_OK;
  PROF(GSS_PROF_COMM_GRADS);
  p->n_coll[2] += 1;
  const gss_plan_desc &D = p->desc;
  const size_t cnt[4] = {(size_t)D.d * D.d, (size_t)D.d, (size_t)D.d * D.d, (size_t)D.d};
  return p->comm->all_reduce_sum(p->grad, cnt, 4, as_stream(stream));
}

// the input features never change: their halo is fetched once
int plan_x0(gss_plan *p, void *stream) {
  if (p->P == 1 || p->x0_ready) return GSS_OK;
  if (p->desc.n > 0)
    GSS_HIP(hipMemcpyAsync(p->x0op, p->x, sizeof(float) * (size_t)p->desc.n * p->desc.d, hipMemcpyDeviceToDevice, as_stream(stream)));
  if (int rc = plan_halo(p, p->halo_a, p->x0op, stream)) return rc;
  p->x0_ready = true;
  return GSS_OK;
}

// enqueue layer 1's SpMMs for the NEXT step on the side stream, to start once the main stream reaches this point
int plan_prefetch_layer1(gss_plan *p, void *main_stream) {
  const gss_plan_desc &D = p->desc;
  const int nxt = p->cur0 ^ 1;
  // everything step t-1 read from buffer set `nxt` precedes this point of the main stream
  GSS_HIP(hipEventRecord(p->ev_main_ready, as_stream(main_stream)));
  GSS_HIP(hipStreamWaitEvent(p->side, p->ev_main_ready, 0));
  {
    void *stream = p->side;
    {
      PROF(GSS_PROF_SPMM_FWD_HAD);
      if (int rc = spmm_fwd(p->a, D.d, p->x, p->ax0[nxt], p->x, p->m_side, stream)) return rc;
    }
    PROF(GSS_PROF_SPMM_FWD);
    if (int rc = spmm_fwd(p->a, D.d, p->m_side, p->am0[nxt], nullptr, nullptr, stream)) return rc;
  }
  GSS_HIP(hipEventRecord(p->ev_side_done, p->side));
  p->prefetched = true;
  return GSS_OK;
}

// lazy_rows != NULL (gss_plan_step_lazy): the top layer's A_hat M, projection, ELU, residual and normalisation are evaluated on the
// lazy_b listed rows only (p->pos is their row mask) -- the rows the loss and the backward pass read
int plan_forward_impl(gss_plan *p, void *stream, const int32_t *lazy_rows = nullptr, int32_t lazy_b = 0) {
  if (p) p->needbits_valid = false;   // (set again below when this pass marks the batch rows and their neighbours)
  GSS_REQUIRE(p, "plan_forward: null plan");
  const gss_plan_desc &D = p->desc;
  const int L = D.num_layers;
  bool have_l0 = false;
  if (p->prefetched) {
    // layer 1 of this step was computed on the side stream during the previous step
    GSS_HIP(hipStreamWaitEvent(as_stream(stream), p->ev_side_done, 0));
    p->prefetched = false;
    p->cur0 ^= 1;
    p->ax[0] = p->ax0[p->cur0];
    p->am[0] = p->am0[p->cur0];
    have_l0 = true;
  }
  for (int l = 0; l < L; ++l) {
    float *xl = p->xin[l];
    const bool cached = (l == 0 && ((D.cache_layer1 && p->layer1_valid) || have_l0));
    if (!cached) {
      // AX = A x ; M = AX (.) x      (model.py:163,168); x's boundary rows come from their owners (C1)
      if (l == 0)
        if (int rc = plan_x0(p, stream)) return rc;
      const bool split_a = p->a_own != nullptr;
      // layer 1's inputs are constants, so are the boundary rows of its M: every shard recomputes its own rows each step
      // (the SpMM is executed), the boundary rows are exchanged once
      float *m = l == 0 ? p->m0op : p->m_tmp;
      const bool lazy_l = lazy_rows && l == L - 1;
      const bool needed_only = lazy_l && p->P > 1 && p->lz.on;   // sharded: A_hat M fetches only the boundary rows of M its rows read
      if (needed_only) {  // who reads which of M's rows: the peers' requests also say which of THIS shard's rows anything reads
        if (!p->lz_pending)   // (gss_plan_step_lazy started the request phase before layer 1; anything else that gets here starts it now)
          if (int rc = plan_lazy_requests_begin(p, lazy_rows, lazy_b, stream)) return rc;
        if (int rc = plan_lazy_requests_end(p, stream)) return rc;
      }
      if (lazy_l && p->needbits) {
        // huge graphs: AX and M of the top layer on the batch rows and their neighbours only (what A_hat M on the batch rows, the
        // batch-row weight gradient and the backward hop's epilogue read); a shard adds the rows its peers' batch rows reference
        PROF(GSS_PROF_ELEMENTWISE);
        GSS_HIP(hipMemsetAsync(p->needbits, 0, sizeof(uint32_t) * (p->rows_a / 32 + 1), as_stream(stream)));
        if (int rc = mark_rows_and_neighbours(p->a, lazy_rows, lazy_b, p->needbits, stream)) return rc;
        if (needed_only)
          if (int rc = bits_set_list(p->needbits, p->lz.send_list, p->lz.h_cnt[p->P], stream)) return rc;
        p->needbits_valid = p->P == 1 || needed_only;   // own rows: complete (a shard without the request phase does not know its peers' batch rows)
      }
      // (what the row-filtered products of a lazy top layer are expected to compute: the batch rows and their neighbours / the batch rows)
      const int64_t deg_a = p->a->n_rows > 0 ? p->a->nnz / p->a->n_rows : 0;
      LiveHint hint_nb(lazy_l ? (int64_t)lazy_b * (1 + deg_a) : 0);
      {
        const uint32_t *rbits = (lazy_l && p->needbits) ? p->needbits : nullptr;
        auto full = [&]() {
          PROF(GSS_PROF_SPMM_FWD_HAD);
          const BatchPrep *prep = p->prep_pending;     // the step's batch preparation as a side job of its first forward SpMM
          p->prep_pending = nullptr;
          return spmm_fwd(p->a, D.d, xl, p->ax[l], xl, m, stream, nullptr, rbits, nullptr, nullptr, prep);
        };
        if (l == 0 || (l == 1 && p->recompute)) {
          // the boundary rows of X_0 are constants, fetched once by plan_x0; those of X_1 were computed by layer 1's projection
          // (halo_recompute): nothing to exchange, nothing to overlap
          if (int rc = full()) return rc;
        } else {
          auto own = [&]() {
            PROF(GSS_PROF_SPMM_FWD_HAD);
            return spmm_fwd(p->a_own, D.d, xl, p->ax[l], nullptr, nullptr, stream, nullptr, rbits);
          };
          auto rest = [&]() {
            PROF(GSS_PROF_SPMM_FWD_HAD);
            return spmm_fwd(p->a_halo, D.d, xl, p->ax[l], xl, m, stream, nullptr, rbits, p->ax[l]);
          };
          if (int rc = plan_hop(p, p->halo_a, split_a, xl, stream, full, own, rest)) return rc;
        }
      }
      // AM = A M                      (model.py:169)
      {
        const int32_t *rpos = lazy_l ? p->pos : nullptr;
        LiveHint hint_b(lazy_l ? (int64_t)lazy_b : 0);   // (A_hat M on the batch rows only)
        auto full = [&]() {
          PROF(GSS_PROF_SPMM_FWD);
          return spmm_fwd(p->a, D.d, m, p->am[l], nullptr, nullptr, stream, rpos);
        };
        if (l == 0 && p->m0_ready) {   // layer 1's M: boundary rows are constants, fetched in the first step
          if (int rc = full()) return rc;
        } else {
          auto own = [&]() {
            PROF(GSS_PROF_SPMM_FWD);
            return spmm_fwd(p->a_own, D.d, m, p->am[l], nullptr, nullptr, stream, rpos);
          };
          auto rest = [&]() {
            PROF(GSS_PROF_SPMM_FWD);
            return spmm_fwd(p->a_halo, D.d, m, p->am[l], nullptr, nullptr, stream, rpos, nullptr, p->am[l]);
          };
          if (int rc = plan_hop(p, p->halo_a, split_a, m, stream, full, own, rest, needed_only ? lazy_rows : nullptr, lazy_b)) return rc;
          if (l == 0) p->m0_ready = true;
        }
      }
      if (l == 0) p->layer1_valid = true;   // (the lazy top layer is never layer 1: gss_plan_step_lazy needs two layers)
      if (l == 0 && p->recompute && !p->l0h_ready) {
        // AX_0 / AM_0 are functions of A_hat and X alone: their boundary rows are fetched once (the own rows above them are rewritten
        // with the same values every step)
        if (int rc = plan_halo(p, p->halo_a, p->ax[0], stream)) return rc;
        if (int rc = plan_halo(p, p->halo_a, p->am[0], stream)) return rc;
        p->l0h_ready = true;
      }
    }
    PROF(GSS_PROF_DENSE_FWD);
    if (l == L - 1 && dense_fwd_norm_available(D.d)) {
      // last layer: F.normalize fused into the GEMM epilogue (no x_last round trip, no extra launch)
      if (lazy_rows) {
        // one GPU: every listed row is a batch member in batch order -- the epilogue also drops its unit-norm row into E_B
        float *e_b = (p->P == 1 && lazy_b > 0) ? loss_workspace_e_b(D.d, lazy_b, p->loss_ws) : nullptr;
        if (int rc = dense_fwd_norm(lazy_b, D.d, p->ax[l], p->am[l], p->w1, p->b1, p->w2, p->b2, l > 0 ? p->p[l - 1] : nullptr, D.layer_decay,
                                    p->p[l], p->emb, p->inv_den, stream, lazy_rows, e_b))
          return rc;
        p->eb_rows = e_b ? lazy_b : 0;
        return GSS_OK;
      }
      // a full step whose batch was prepared by the first SpMM's side job: the batch members' unit-norm rows also go to E_B
      float *e_b = (p->eb_scatter_b > 0 && p->pos) ? loss_workspace_e_b(D.d, p->eb_scatter_b, p->loss_ws) : nullptr;
      if (int rc = dense_fwd_norm(D.n, D.d, p->ax[l], p->am[l], p->w1, p->b1, p->w2, p->b2, l > 0 ? p->p[l - 1] : nullptr, D.layer_decay,
                                  p->p[l], p->emb, p->inv_den, stream, nullptr, e_b, e_b ? p->pos : nullptr))
        return rc;
      p->eb_rows = e_b ? p->eb_scatter_b : 0;
      return GSS_OK;
    }

    float *xn = (l == L - 1) ? p->x_last : p->xin[l + 1];
    const bool listed = lazy_rows && l == L - 1;
    // halo_recompute: layer 1's projection also produces the boundary rows of layer 2's input (l == 0 < L - 1 there)
    const int32_t n_proj = (l == 0 && p->recompute) ? (int32_t)p->rows_a : D.n;
    if (int rc = dense_fwd(listed ? lazy_b : n_proj, D.d, p->ax[l], p->am[l], p->w1, p->b1, p->w2, p->b2, l > 0 ? p->p[l - 1] : nullptr,
                           D.layer_decay, p->p[l], xn, stream, listed ? lazy_rows : nullptr))
      return rc;
  }
  PROF(GSS_PROF_ROWNORM);
  if (lazy_rows) return rownorm_fwd(lazy_b, D.d, p->x_last, p->emb, p->inv_den, stream, lazy_rows);
  return rownorm_fwd(D.n, D.d, p->x_last, p->emb, p->inv_den, stream);
}

// What the kernels index a batch with.  One GPU: the node ids themselves.  A shard: `rows` = the local row of every
// batch member clamped into the shard (foreign rows get zero gradients through `keep`), `ids` = the member's operand row
// in A_hat^T's column space (the key of the batch-position map; -1 where this shard never reads the node).
struct BatchView {
  const int32_t *rows, *ids;
  const float *keep;
};

// the batch as the plan's kernels see it; for a relabelled graph / a shard the translated ids are WRITTEN by the gather launch of
// plan_loss_backward_impl (loss.hip gather_rows_mapped_kernel), this only says where they live
static bool plan_batch_mapped(const gss_plan *p) { return p->P > 1 || p->desc.node_map; }
static BatchView plan_batch_view(gss_plan *p, const int32_t *idx) {
  if (!plan_batch_mapped(p)) return BatchView{idx, idx, nullptr};
  return BatchView{p->rloc, p->pid, p->keep};   // keep stays NULL on one GPU (every row is owned)
}

int plan_backward_impl(gss_plan *p, const BatchView &bv, int32_t b, const float *de_rows, bool top_done, bool wt_ok, void *stream,
                       int *deferred_slices = nullptr, bool dgrad_done = false);

// prepared (gss_plan_step_lazy): batch_prepare already translated the batch ids and set the batch-position map
int plan_loss_backward_impl(gss_plan *p, const int32_t *idx, int32_t b, float beta, bool wt_ok, void *stream, BatchView &bv,
                            int *deferred_slices = nullptr, bool prepared = false) {
  GSS_REQUIRE(p && idx, "plan_loss_backward: null argument");
  const gss_plan_desc &D = p->desc;
  GSS_REQUIRE(b >= 1 && b <= D.max_batch, "plan_loss_backward: batch %d out of [1, %d]", b, D.max_batch);
  const int L = D.num_layers;
  bv = plan_batch_view(p, idx);
  const bool sparse_top = L > 1 && spmm_sparse_available();
  GSS_REQUIRE(p->P == 1 || L == 1 || sparse_top, "a sharded plan needs the balanced SpMM (spmm_variant 2)");
  const bool mapped = plan_batch_mapped(p);
  // the batch rows' input gradient shares the launch of the loss's finish where the width allows: it needs the transposed weights now
  const bool want_dgrad = sparse_top && loss_dgrad_available(D.d, b, p->P > 1);
  if (want_dgrad && !wt_ok) {
    PROF(GSS_PROF_ELEMENTWISE);
    if (int rc = transpose2(D.d, p->w1, p->w2, p->w1t, p->w2t, stream)) return rc;
    wt_ok = true;
  }
  LossStep s{};
  s.d = D.d;
  s.b = b;
  s.beta = beta;
  s.alpha = D.alpha;
  s.loss_out = p->loss;
  s.inv_den = p->inv_den;
  s.p = p->p[L - 1];
  s.c = L > 1 ? D.layer_decay : 1.f;
  s.dx_b = p->dx_b;
  s.dp_b = p->dp_b;
  s.pos_set = (sparse_top && !prepared) ? p->pos : nullptr;
  s.keep = bv.keep;
  if (want_dgrad) {
    s.w1t = p->w1t;
    s.w2t = p->w2t;
    s.gax_b = p->gab;
    s.gam_b = p->gab + (size_t)b * D.d;
  }
  const bool eb_ready = prepared && p->P == 1 && p->eb_rows == b;   // the lazy forward of this step wrote E_B itself
  p->eb_rows = 0;
  if (eb_ready) {
    s.e_b = loss_workspace_e_b(D.d, b, p->loss_ws);
    s.rows = bv.rows;
    s.pos_ids = bv.ids;
  } else if (p->P == 1) {
    float *e_b = nullptr;
    {
      // E_B = emb[idx] (model.py:216-217), the batch-id translation of a relabelled graph in the same launch
      PROF(GSS_PROF_LOSS);
      if (mapped && !prepared) {
        if (int rc = loss_gather_rows_mapped(D.d, p->emb, idx, D.node_map, p->lo, D.n, p->gid2op_t, p->pid, p->rloc, p->keep, b, p->loss_ws, &e_b,
                                             stream))
          return rc;
      } else if (int rc = loss_gather_rows(D.d, p->emb, bv.rows, bv.keep, b, p->loss_ws, &e_b, stream))
        return rc;
    }
    s.e_b = e_b;
    s.rows = bv.rows;
    s.pos_ids = bv.ids;
  } else if (p->P > 1) {
    // shards: ONE batch collective.  Every shard contributes [E_B | P_B | inv_B] of the members it owns (zeros elsewhere); after the
    // all-reduce (C3; one non-zero contributor per element: exact) every rank holds the whole batch's rows of the embeddings
    // (model.py:216-217), of the top layer's pre-activation and its 1 / ||x||, runs the same B x B sweep (identical bits everywhere,
    // no exchange of the loss) and the finish + input gradient of EVERY member: the [2 B][d] input gradients need no second collective
    {
      PROF(GSS_PROF_LOSS);
      if (int rc = loss_gather_batch(D.d, p->emb, p->p[L - 1], p->inv_den, prepared ? nullptr : idx, D.node_map, p->lo, D.n, p->gid2op_t, p->pid,
                                     p->rloc, p->keep, bv.rows, b, p->bx, stream))
        return rc;
    }
    if (int rc = plan_allreduce(p, p->bx, (size_t)b * (2 * D.d + 1), stream)) return rc;
    s.e_b = p->bx;
    s.p = p->bx + (size_t)b * D.d;
    s.inv_den = p->bx + (size_t)2 * b * D.d;
    s.rows = nullptr;                       // p / inv_den are per member
    s.pos_ids = bv.ids;
    s.dgrad_all = true;
  }
  // Row-slab sweep ("each GPU computes its row-slab of S and of 2 G E_B"; knob loss_slab, automatic from B = 8192): with
  // the whole batch's rows on every rank, rank r sweeps the i tiles r, r + P, ... -- 1 / P of the B^2 d work whoever owns the members
  // -- and the ranks sum their rows of dE (and their shares of the loss) in one more all-reduce.  Below that size the sweep (25 us at
  // B = 2048, d = 128) is replicated instead: a second batch collective costs more than the (P - 1) / P of it that a slab saves.
  const int slab_knob = K().loss_slab;
  const bool slab = p->P > 1 && (slab_knob == 1 || (slab_knob < 0 && b >= 8192));
  if (slab) {
    s.slab_rank = p->rank;
    s.slab_parts = p->P;
    {
      PROF(GSS_PROF_LOSS);
      if (int rc = loss_step_slab_sweep(s, p->loss_ws, p->dex, stream)) return rc;
    }
    if (int rc = plan_allreduce(p, p->dex, (size_t)b * D.d + 1, stream)) return rc;
    GSS_HIP(hipMemcpyAsync(p->loss, p->dex + (size_t)b * D.d, sizeof(float), hipMemcpyDeviceToDevice, as_stream(stream)));
    s.de_x = p->dex;
  }
  bool dgrad_done = false;
  {
    // loss, dLoss/dE_B and the backward of F.normalize / F.elu on the batch rows (replicated form: every shard computes the full
    // B x B sweep -- identical bits everywhere, no exchange of the loss); the gradient rows a shard owns feed its weight gradient
    PROF(GSS_PROF_LOSS);
    if (int rc = loss_step(s, p->loss_ws, stream, &dgrad_done)) return rc;
  }
  // (a shard at a width the fused finish does not cover -- outside {64, 128, 256} -- falls back to the masked input gradient + its all-reduce)
  return plan_backward_impl(p, bv, b, nullptr, true, wt_ok, stream, deferred_slices, dgrad_done);
}

// deferred_slices != NULL (gss_plan_step): the caller finishes with the fused reduce + Adam kernel, which also resets
// the batch-position map; the residual of the top layer then rides in the backward SpMM's epilogue
int plan_backward_impl(gss_plan *p, const BatchView &bv, int32_t b, const float *de_rows, bool top_done, bool wt_ok, void *stream,
                       int *deferred_slices, bool dgrad_done) {
  GSS_REQUIRE(p && bv.rows, "plan_backward: null argument");
  const gss_plan_desc &D = p->desc;
  GSS_REQUIRE(b >= 1 && b <= D.max_batch, "plan_backward: %d rows out of [1, %d]", b, D.max_batch);
  const int L = D.num_layers;
  hipStream_t st = as_stream(stream);
  const float *de_b = de_rows ? de_rows : p->de_b;
  const int32_t *pos_row = p->pos;  // a shard's own rows are the first n operand rows of A_hat^T's column space
  // top layer, batch rows only
  const float c_top = L > 1 ? D.layer_decay : 1.f;
  const bool sparse_top = L > 1 && spmm_sparse_available();
  if (!top_done) {
    PROF(GSS_PROF_ELEMENTWISE);
    if (int rc = rownorm_elu_bwd(D.d, de_b, bv.rows, b, p->emb, p->inv_den, p->p[L - 1], c_top, p->dx_b, p->dp_b,
                                 sparse_top ? p->pos : nullptr, stream))
      return rc;
  }
  // every layer's partial slabs go to consecutive slices of one buffer; one fixed-order reduce at the end
  int wg_used = 0, wg_n = 0;
  // gss_plan_step with L > 1: the top layer's batch-row weight gradient (64 latency-bound workgroups, its inputs stay
  // untouched until the end of the step) is postponed and shares the launch of the next layer's full-N one.  Its
  // partial slabs keep their place (slices [0, wg_top)), so the fixed-order reduce adds the same numbers in the same order.
  const bool merge_top = deferred_slices && L > 1 && D.n > 0;
  const int wg_top = wgrad_slices(b, D.d);
  if (merge_top) {
    wg_used = wg_top;
  } else {
    PROF(GSS_PROF_WGRAD_BATCH);
    if (int rc = wgrad_partial(b, D.d, p->dp_b, p->ax[L - 1], p->am[L - 1], bv.rows, p->wgrad_ws, p->wg_total, wg_used, &wg_n, stream))
      return rc;
    wg_used += wg_n;
  }
  if (L > 1) {
    if (!wt_ok) {
      PROF(GSS_PROF_ELEMENTWISE);
      if (int rc = transpose2(D.d, p->w1, p->w2, p->w1t, p->w2t, stream)) return rc;
    }
    if (sparse_top) {
      // the top layer's input gradients live on the b batch rows only: keep them compact and let the SpMM
      // skip every neighbour that is not a batch row.  Sharded: rows a shard does not own are zero in dp_b, so the
      // sum over the shards (C3) is the batch's gradient
      p->gax_b = p->gab;
      p->gam_b = p->gab + (size_t)b * D.d;
      if (!dgrad_done) {   // (else the loss's finish launch already wrote them)
        PROF(GSS_PROF_DGRAD);
        if (int rc = dense_bwd_input(b, D.d, p->dp_b, p->w1t, p->w2t, nullptr, p->gax_b, p->gam_b, stream)) return rc;
      }
      if (!dgrad_done)     // the finish launch's input gradient covers every member on every rank (LossStep.dgrad_all): nothing to sum
        if (int rc = plan_allreduce(p, p->gab, (size_t)2 * b * D.d, stream)) return rc;
      if (p->posbits) {
        PROF(GSS_PROF_ELEMENTWISE);
        if (int rc = batch_bits(bv.ids, b, p->posbits, 1, stream)) return rc;
      }
      // huge operands: the hop records which rows of u can be non-zero, the hop after it -- fold_res below -- follows only those.
      // One shard: every bit is cleared and rows whose bit stays clear are not even written (nothing reads them).  Several shards: the
      // word that straddles the own rows and the always-set halo rows keeps its bits (stale own bits only cost gathers; all rows are
      // written, a peer may read them)
      const bool track_nz = p->nzbits && deferred_slices;
      const size_t nz_words = p->P == 1 ? ((size_t)D.n + 31) / 32 : (size_t)D.n / 32;
      if (track_nz && nz_words > 0) {
        PROF(GSS_PROF_ELEMENTWISE);
        GSS_HIP(hipMemsetAsync(p->nzbits, 0, sizeof(uint32_t) * nz_words, st));
      }
      // (round 6) huge graphs: the hop walks only the rows that can have a hit -- the batch rows and their neighbours, the set the lazy
      // step's forward marked for this batch; a full step on one GPU marks it here (b rows of A_hat: microseconds against the 2.7 ms the
      // hop spent streaming RMAT 10M's index for 45 k hits).  Whole-step entry points only (the same batch as the forward's).
      const uint32_t *live_rows = nullptr;
      if (p->needbits && deferred_slices) {
        if (!p->needbits_valid && p->P == 1) {
          PROF(GSS_PROF_ELEMENTWISE);
          GSS_HIP(hipMemsetAsync(p->needbits, 0, sizeof(uint32_t) * (p->rows_a / 32 + 1), st));
          if (int rc = mark_rows_and_neighbours(p->a, bv.rows, b, p->needbits, stream)) return rc;
          p->needbits_valid = true;
        }
        if (p->needbits_valid) live_rows = p->needbits;
      }
      {
        PROF(GSS_PROF_SPMM_BWD1);
        LiveHint hint((int64_t)b * (1 + (p->at->n_rows > 0 ? p->at->nnz / p->at->n_rows : 0)));
        if (int rc = spmm_bwd1_sparse(p->at, D.d, p->gam_b, p->gax_b, p->pos, pos_row, p->xin[L - 1], p->ax[L - 1], p->u, p->t, stream, p->posbits,
                                      track_nz ? p->nzbits : nullptr, track_nz && p->P == 1, live_rows))
          return rc;
      }
      if (p->posbits) {
        PROF(GSS_PROF_ELEMENTWISE);
        if (int rc = batch_bits(bv.ids, b, p->posbits, 0, stream)) return rc;
      }
    } else {
      const size_t nd_bytes = sizeof(float) * (size_t)D.n * D.d;
      {
        PROF(GSS_PROF_ELEMENTWISE);
        GSS_HIP(hipMemsetAsync(p->g_ax, 0, nd_bytes, st));
        GSS_HIP(hipMemsetAsync(p->g_am, 0, nd_bytes, st));
      }
      {
        PROF(GSS_PROF_DGRAD);
        if (int rc = dense_bwd_input(b, D.d, p->dp_b, p->w1t, p->w2t, bv.rows, p->g_ax, p->g_am, stream)) return rc;
      }
      PROF(GSS_PROF_SPMM_BWD1_DENSE);
      if (int rc = spmm_bwd1(p->at, D.d, p->g_am, p->g_ax, p->xin[L - 1], p->ax[L - 1], p->u, p->t, stream)) return rc;
    }
    for (int lp = L - 2; lp >= 0; --lp) {
      const float c = lp == 0 ? 1.f : D.layer_decay;
      const float *res = (lp + 2 <= L - 1) ? p->gx[(lp + 2) & 1] : nullptr;
      float *gx_out = (lp >= 1 && L > 2) ? p->gx[(lp + 1) & 1] : nullptr;
      const bool fold_res = deferred_slices && sparse_top && lp + 2 == L;
      bool use_tloc = false;
      {
        // gx = t + A_hat^T u needs u's boundary rows (C1).  fold_res: dP += dx_b on the batch rows inside the SpMM epilogue (no separate
        // scatter-add launch).  Overlapped: the own-column sums go to dp first (with the non-zero-row filter of u where there is one)
        const bool split_t = p->at_own != nullptr;
        const int bwd2_cls = fold_res ? GSS_PROF_SPMM_BWD2 : GSS_PROF_SPMM_BWD2_DENSE;   // (the top layer's sparsity-aware hop / an N-row hop)
        auto full = [&]() {
          PROF(bwd2_cls);
          if (fold_res) return spmm_bwd2_sparse_res(p->at, D.d, p->u, p->t, p->p[lp], c, p->dx_b, pos_row, p->dp, gx_out, stream, p->nzbits);
          return spmm_bwd2(p->at, D.d, p->u, p->t, p->p[lp], c, res, p->dp, gx_out, stream);
        };
        auto own = [&]() {
          PROF(bwd2_cls);
          return spmm_fwd(p->at_own, D.d, p->u, p->dp, nullptr, nullptr, stream, nullptr, nullptr, nullptr, fold_res ? p->nzbits : nullptr);
        };
        auto rest = [&]() {
          PROF(bwd2_cls);
          if (fold_res)
            return spmm_bwd2_sparse_res(p->at_halo, D.d, p->u, p->t, p->p[lp], c, p->dx_b, pos_row, p->dp, gx_out, stream, p->nzbits, p->dp);
          return spmm_bwd2(p->at_halo, D.d, p->u, p->t, p->p[lp], c, res, p->dp, gx_out, stream, p->dp);
        };
        // fold_res with the non-zero-row bitmap: u is zero outside the batch's neighbourhood -- the owners send the rows that are not
        const bool subset = fold_res && p->lzt.on && p->nzbits;
        // tloc (two layers, halo_recompute, the shard's A_hat transposed in place): dP_0 only feeds the bottom layer's weight gradient,
        // a sum over ALL nodes that the ranks' all-reduce completes anyway -- so this shard multiplies ITS rows of u into every row they
        // touch, own and boundary (T_loc u_own), applies c * (t + .) (.) ELU'(P_0) there (t exists on own rows; P_0 of the boundary
        // rows is what layer 1's projection over own + boundary rows left behind) and sums the weight gradient over own + boundary
        // rows below.  Every (entry, u row) pair is counted once, at the owner of the u row: no exchange.
        use_tloc = lp == 0 && p->tloc != nullptr;
        if (use_tloc) {
          PROF(bwd2_cls);
          if (fold_res) {
            if (int rc = spmm_bwd2_sparse_res(p->tloc, D.d, p->u, p->t, p->p[0], c, p->dx_b, pos_row, p->dp, nullptr, stream, p->nzbits, nullptr, D.n))
              return rc;
          } else if (int rc = spmm_bwd2(p->tloc, D.d, p->u, p->t, p->p[0], c, res, p->dp, nullptr, stream, nullptr, D.n)) {
            return rc;    // (deeper nets: the residual of layer 2 lives on own rows; the phase-wise entry points: the batch rows' residual is
          }               //  scattered onto own rows below)
        } else if (int rc = plan_hop(p, p->halo_t, split_t, p->u, stream, full, own, rest, nullptr, 0, subset ? p->nzbits : nullptr)) {
          return rc;
        }
      }
      if (lp + 2 == L && !fold_res) {
        PROF(GSS_PROF_ELEMENTWISE);
        if (int rc = scatter_add_rows(D.d, p->dx_b, bv.rows, bv.keep, b, p->dp, sparse_top ? p->pos : nullptr, bv.ids, stream)) return rc;
      }
      {
        PROF(GSS_PROF_WGRAD);
        if (merge_top && lp == L - 2) {
          int n_top = 0;
          const int32_t n_w = use_tloc ? (int32_t)p->rows_a : D.n;       // tloc: dP_0 lives on own + boundary rows
          if (int rc = wgrad_partial_pair(D.d, n_w, p->dp, p->ax[lp], p->am[lp], nullptr, wg_used, b, p->dp_b, p->ax[L - 1], p->am[L - 1], bv.rows,
                                          0, p->wgrad_ws, p->wg_total, &wg_n, &n_top, stream))
            return rc;
        } else {
          const int32_t n_w = use_tloc ? (int32_t)p->rows_a : D.n;
          if (int rc = wgrad_partial(n_w, D.d, p->dp, p->ax[lp], p->am[lp], nullptr, p->wgrad_ws, p->wg_total, wg_used, &wg_n, stream))
            return rc;
        }
        wg_used += wg_n;
      }
      if (lp >= 1) {
        {
          PROF(GSS_PROF_DGRAD);
          if (int rc = dense_bwd_input(D.n, D.d, p->dp, p->w1t, p->w2t, nullptr, p->g_ax, p->g_am, stream)) return rc;
        }
        // dm = A_hat^T g_am needs g_am's boundary rows (C1); overlapped: the own-column sums go to t first
        const bool split_t = p->at_own != nullptr;
        auto full = [&]() {
          PROF(GSS_PROF_SPMM_BWD1_DENSE);
          return spmm_bwd1(p->at, D.d, p->g_am, p->g_ax, p->xin[lp], p->ax[lp], p->u, p->t, stream);
        };
        auto own = [&]() {
          PROF(GSS_PROF_SPMM_BWD1_DENSE);
          return spmm_fwd(p->at_own, D.d, p->g_am, p->t, nullptr, nullptr, stream);
        };
        auto rest = [&]() {
          PROF(GSS_PROF_SPMM_BWD1_DENSE);
          return spmm_bwd1(p->at_halo, D.d, p->g_am, p->g_ax, p->xin[lp], p->ax[lp], p->u, p->t, stream, p->t);
        };
        if (int rc = plan_hop(p, p->halo_t, split_t, p->g_am, stream, full, own, rest)) return rc;
      }
    }
  }
  // b1 and b2 enter the sum p = ... + b1 + ... + b2 symmetrically (model.py:165,170,172): the reduce kernel
  // writes the same column sums to both bias gradients
  if (deferred_slices) {
    *deferred_slices = wg_used;
    return GSS_OK;
  }
  {
    PROF(GSS_PROF_WGRAD);
    if (int rc = wgrad_reduce(D.d, p->wgrad_ws, p->wg_total, wg_used, p->grad[0], p->grad[2], p->grad[1], p->grad[3], 0, stream)) return rc;
  }
  return plan_allreduce_grads(p, stream);
}

int plan_adam_impl(gss_plan *p, void *stream, int32_t *pos_clear = nullptr, const int32_t *ids = nullptr, int32_t b = 0) {
  GSS_REQUIRE(p, "plan_adam: null plan");
  const gss_plan_desc &D = p->desc;
  p->step += 1;
  PROF(GSS_PROF_ADAM);
  float *params[4] = {p->w1, p->b1, p->w2, p->b2};
  const int64_t cnt[4] = {(int64_t)D.d * D.d, D.d, (int64_t)D.d * D.d, D.d};
  AdamTensor t[4];
  for (int k = 0; k < 4; ++k) t[k] = AdamTensor{params[k], p->grad[k], p->adam_m[k], p->adam_v[k], cnt[k]};
  const bool wt = D.num_layers > 1;
  return adam_step4(t, p->step, D.lr, D.beta1, D.beta2, D.eps, wt ? p->w1t : nullptr, wt ? p->w2t : nullptr, D.d, stream, pos_clear, ids, b);
}
}  // namespace

extern "C" {

int gss_plan_create(gss_plan **out, const gss_plan_desc *desc, const gss_csr *a, const gss_csr *at, const gss_plan_io *io) {
  return plan_create_impl(out, desc, nullptr, nullptr, a, at, io);
}

int gss_plan_create_sharded(gss_plan **out, const gss_plan_desc *desc, const gss_shard_desc *shard, gss_comm *comm, const gss_csr *a,
                            const gss_csr *at, const gss_plan_io *io) {
  GSS_REQUIRE(shard, "plan_create_sharded: null shard descriptor");
  return plan_create_impl(out, desc, shard, comm, a, at, io);
}

void gss_plan_destroy(gss_plan *p) {
  if (!p) return;
  for (hipEvent_t e : p->ev) (void)hipEventDestroy(e);
  if (p->side) {
    (void)hipStreamSynchronize(p->side);
    (void)hipStreamDestroy(p->side);
  }
  if (p->xs) {
    (void)hipStreamSynchronize(p->xs);
    (void)hipStreamDestroy(p->xs);
  }
  if (p->rq) {
    (void)hipStreamSynchronize(p->rq);
    (void)hipStreamDestroy(p->rq);
  }
  if (p->ev_rq_in) (void)hipEventDestroy(p->ev_rq_in);
  if (p->ev_rq_cnt) (void)hipEventDestroy(p->ev_rq_cnt);
  if (p->ev_ready) (void)hipEventDestroy(p->ev_ready);
  if (p->ev_halo) (void)hipEventDestroy(p->ev_halo);
  if (p->ev_main_ready) (void)hipEventDestroy(p->ev_main_ready);
  if (p->ev_side_done) (void)hipEventDestroy(p->ev_side_done);
  if (p->lz.h_cnt) (void)hipHostFree(p->lz.h_cnt);
  if (p->lzt.h_cnt) (void)hipHostFree(p->lzt.h_cnt);
  if (p->slab) (void)hipFree(p->slab);
  delete p;
}

// Measurement aid (tools/ab_live.py): change ONE kernel-selection knob in a LIVE plan's snapshot.  Only knobs that pick between kernels
// over the same buffers are accepted -- nothing that sizes a workspace or steers the plan's own bookkeeping -- so the plan's memory
// layout stays what it was.  Two plans in one process differ by up to +-3.4 us per step in identical settings (their buffers sit at
// different addresses); the same plan under alternating settings does not.
int gss_plan_debug_set_option(gss_plan *p, const char *name, int value) {
  GSS_REQUIRE(p && name, "plan_debug_set_option: null argument");
  static const char *const kLive[] = {"gemm_variant", "spmm_slices", "spmm_pin", "gemm_ws", "spmm_list_blocks"};
  bool ok = false;
  for (const char *k : kLive) ok = ok || strcmp(k, name) == 0;
  GSS_REQUIRE(ok, "plan_debug_set_option: only kernel-selection knobs can change on a live plan");
  return gss::set_knob(p->knobs, name, value);   // the plan's own snapshot: no other thread's plan or per-op call sees it
}

// ---- public entry points.  The separate phases make no assumption about who changed the weights in between, so
// they always re-transpose; gss_plan_step owns the whole iteration and reuses the transposes its own Adam wrote.
int gss_plan_forward(gss_plan *p, void *stream) {
  KnobScope knob_scope(p ? &p->knobs : nullptr);
  if (p) p->wt_valid = false;
  return plan_forward_impl(p, stream);
}
int gss_plan_loss_backward(gss_plan *p, const int32_t *idx, int32_t b, float beta, void *stream) {
  KnobScope knob_scope(p ? &p->knobs : nullptr);
  if (p) p->wt_valid = false;
  BatchView bv{};
  return plan_loss_backward_impl(p, idx, b, beta, false, stream, bv);
}
int gss_plan_backward(gss_plan *p, const int32_t *rows, int32_t b, const float *de_rows, void *stream) {
  KnobScope knob_scope(p ? &p->knobs : nullptr);
  GSS_REQUIRE(p && rows, "plan_backward: null argument");
  GSS_REQUIRE(p->P == 1 && !p->desc.node_map, "plan_backward: an external upstream gradient is not supported on a sharded or relabelled plan");
  p->wt_valid = false;
  const BatchView bv{rows, rows, nullptr};
  return plan_backward_impl(p, bv, b, de_rows, false, false, stream);
}
int gss_plan_adam(gss_plan *p, void *stream) {
  KnobScope knob_scope(p ? &p->knobs : nullptr);
  const int rc = plan_adam_impl(p, stream);
  if (p) p->wt_valid = false;
  return rc;
}
static int plan_step_impl(gss_plan *p, const int32_t *idx, int32_t b, float beta, void *stream, bool lazy);

int gss_plan_step(gss_plan *p, const int32_t *idx, int32_t b, float beta, void *stream) {
  KnobScope knob_scope(p ? &p->knobs : nullptr);
  return plan_step_impl(p, idx, b, beta, stream, false);
}

// The same step with the top layer evaluated on the batch rows only.  The loss reads the top layer's output on the b batch rows and
// its backward pass reads AX of that layer (kept whole); A_hat M, the projection, ELU, the residual and the normalisation of the other
// N - b rows are not read by anything inside a step.  A row that IS computed goes through the same segments, the same summation tree
// and the same MFMA rows as in the full pass, so loss, gradients and parameters equal gss_plan_step's bit for bit; afterwards
// io.emb holds this step's embeddings on the batch rows only (call gss_plan_forward for all of them).  Falls back to the full step
// where the pieces it needs are absent (one layer, spmm_variant 1).  On a sharded plan every shard evaluates the
// top layer on the batch rows it owns.
int gss_plan_step_lazy(gss_plan *p, const int32_t *idx, int32_t b, float beta, void *stream) {
  KnobScope knob_scope(p ? &p->knobs : nullptr);
  GSS_REQUIRE(p, "plan_step_lazy: null plan");
  const gss_plan_desc &D = p->desc;
  const bool can = D.num_layers > 1 && spmm_sparse_available() && !D.pipeline_layer1 && p->pos;
  return plan_step_impl(p, idx, b, beta, stream, can);
}

static int plan_step_impl(gss_plan *p, const int32_t *idx, int32_t b, float beta, void *stream, bool lazy) {
  GSS_REQUIRE(p, "plan_step: null plan");
  const bool pipe = p->desc.pipeline_layer1 && p->side && !p->prof_on && !p->desc.cache_layer1;
  // One GPU: the batch preparation (id translation, batch-position map) does not get a launch of its own where the step's first
  // forward SpMM can carry it as a side job (spmm.hip BatchPrep): always in a full step with two or more layers; in a lazy step unless
  // that first SpMM is the top layer's with a needed-row bitmap, which is derived from the prepared lists (huge graphs, layer 1 kept).
  const gss_plan_desc &D0 = p->desc;
  const bool mapped0 = plan_batch_mapped(p);
  const bool sparse0 = D0.num_layers > 1 && spmm_sparse_available() && p->pos;
  const bool l0_runs = !(D0.cache_layer1 && p->layer1_valid) && !p->prefetched;
  const bool host_prep = p->P == 1 && sparse0 && K().prep_side != 0 && idx && b >= 1 && b <= D0.max_batch &&
                         (l0_runs || !(lazy && p->needbits));
  BatchPrep prep{idx, b, D0.node_map, p->lo, D0.n, p->gid2op_t, mapped0 ? p->rloc : nullptr, mapped0 ? p->pid : nullptr, p->keep, p->pos, p->rlist};
  bool prepared = lazy;
  if (host_prep) {
    p->prep_pending = &prep;
    prepared = true;
    // (a full step: E_B comes out of the top layer's projection through the map the side job sets)
    p->eb_scatter_b = (!lazy && dense_fwd_norm_available(D0.d)) ? b : 0;
  }
  if (lazy) {
    GSS_REQUIRE(idx && b >= 1 && b <= p->desc.max_batch, "plan_step_lazy: batch %d out of [1, %d]", b, p->desc.max_batch);
    const bool mapped = plan_batch_mapped(p);   // a relabelled graph and / or a shard: translated ids in rloc / pid (/ keep)
    if (!host_prep) {
      PROF(GSS_PROF_ELEMENTWISE);
      if (int rc = batch_prepare(idx, b, p->desc.node_map, p->lo, p->desc.n, p->gid2op_t, mapped ? p->rloc : nullptr, mapped ? p->pid : nullptr,
                                 p->keep, p->pos, stream, p->rlist))
        return rc;
    }
    // a shard lists the batch members it OWNS (rlist: -1 for the others, whose tile rows compute on row 0's operands and store
    // nothing): every listed row has one writer, no row outside the batch is touched; an empty shard has no top layer to evaluate
    const int32_t *rows = p->rlist ? p->rlist : (mapped ? p->rloc : idx);
    // the subset exchange's request phase: it needs the prepared lists and the graph, nothing else -- started now, on the request stream,
    // it runs underneath layer 1
    if (p->P > 1 && p->lz.on)
      if (int rc = plan_lazy_requests_begin(p, rows, p->desc.n > 0 ? b : 0, stream)) return rc;
    const int rc_f = plan_forward_impl(p, stream, rows, p->desc.n > 0 ? b : 0);
    if (rc_f && p->lz_pending) {   // the forward failed before it consumed the request phase: do not leave it dangling
      (void)hipStreamSynchronize(p->rq);
      p->lz_pending = false;
    }
    const bool hosted = host_prep && p->prep_pending == nullptr;
    p->prep_pending = nullptr;
    p->eb_scatter_b = 0;
    if (rc_f) return rc_f;
    GSS_REQUIRE(!host_prep || hosted, "plan_step_lazy: internal error, no forward SpMM carried the batch preparation");
  } else {
    const int rc_f = plan_forward_impl(p, stream);
    const bool hosted = host_prep && p->prep_pending == nullptr;
    p->prep_pending = nullptr;
    p->eb_scatter_b = 0;
    if (rc_f) return rc_f;
    GSS_REQUIRE(!host_prep || hosted, "plan_step: internal error, no forward SpMM carried the batch preparation");
  }
  if (pipe)  // the side stream starts when this stream reaches the loss kernel (MFMA-bound, 1 MB working set)
    if (int rc = plan_prefetch_layer1(p, stream)) return rc;
  int slices = 0;
  BatchView bv{};
  if (int rc = plan_loss_backward_impl(p, idx, b, beta, p->wt_valid, stream, bv, &slices, prepared)) return rc;
  const gss_plan_desc &D = p->desc;
  const bool wt = D.num_layers > 1;
  const bool sparse_top = wt && spmm_sparse_available();
  if (p->P == 1) {
    // weight-gradient reduce + Adam on the four tensors (+ transposed weights for the next backward, + reset of the
    // batch-position map) in one launch
    p->step += 1;
    PROF(GSS_PROF_ADAM);
    float *params[4] = {p->w1, p->b1, p->w2, p->b2};
    if (int rc = wgrad_reduce_adam(D.d, p->wgrad_ws, p->wg_total, slices, p->grad, params, p->adam_m, p->adam_v, p->step, D.lr, D.beta1,
                                   D.beta2, D.eps, wt ? p->w1t : nullptr, wt ? p->w2t : nullptr, sparse_top ? p->pos : nullptr, bv.ids, b,
                                   stream))
      return rc;
  } else {
    // sharded: the gradients of the shards are summed between the reduce and the optimizer (C2)
    {
      PROF(GSS_PROF_WGRAD);
      if (int rc = wgrad_reduce(D.d, p->wgrad_ws, p->wg_total, slices, p->grad[0], p->grad[2], p->grad[1], p->grad[3], 0, stream)) return rc;
    }
    if (int rc = plan_allreduce_grads(p, stream)) return rc;
    if (int rc = plan_adam_impl(p, stream, sparse_top ? p->pos : nullptr, bv.ids, b)) return rc;
  }
  p->wt_valid = wt;
  return GSS_OK;
}

int gss_plan_gather_embeddings(gss_plan *p, float *out, void *stream) {
  KnobScope knob_scope(p ? &p->knobs : nullptr);
  GSS_REQUIRE(p && out, "plan_gather_embeddings: null argument");
  const gss_plan_desc &D = p->desc;
  hipStream_t st = as_stream(stream);
  if (p->P == 1) {
    GSS_HIP(hipMemcpyAsync(out, p->emb, sizeof(float) * (size_t)D.n * D.d, hipMemcpyDeviceToDevice, st));
    return GSS_OK;
  }
  // equal-count all-gather through a padded staging buffer (a one-off, not part of a step)
  int64_t maxr = 1;
  for (int r = 0; r < p->P; ++r) maxr = std::max(maxr, p->h_bounds[(size_t)r + 1] - p->h_bounds[(size_t)r]);
  const size_t slot = (size_t)maxr * D.d;
  float *stage = nullptr;
  GSS_HIP(hipMallocAsync((void **)&stage, sizeof(float) * slot * (size_t)p->P, st));
  float *mine = stage + (size_t)p->rank * slot;
  if (D.n > 0) GSS_HIP(hipMemcpyAsync(mine, p->emb, sizeof(float) * (size_t)D.n * D.d, hipMemcpyDeviceToDevice, st));
  int rc = p->comm->all_gather(mine, stage, sizeof(float) * slot, st);
  for (int r = 0; r < p->P && rc == GSS_OK; ++r) {
    const int64_t lo = p->h_bounds[(size_t)r], rows = p->h_bounds[(size_t)r + 1] - lo;
    if (rows > 0 && hipMemcpyAsync(out + (size_t)lo * D.d, stage + (size_t)r * slot, sizeof(float) * (size_t)rows * D.d, hipMemcpyDeviceToDevice, st) != hipSuccess)
      rc = fail(GSS_EHIP, "plan_gather_embeddings: copy of shard %d failed", r);
  }
  (void)hipFreeAsync(stage, st);
  return rc;
}

const float *gss_plan_activation(const gss_plan *p, int layer, int which) {
  if (!p || layer < 0 || layer >= p->desc.num_layers) return nullptr;
  // 3 / 4 (round 6, parity tests at full size): the pre-activation gradient buffers as the last backward pass left them -- dP of the bottom
  // layer on all rows [n][d] (L >= 2) / the top layer's dP on the batch rows, in batch order [b][d]; `layer` is ignored for them
  // 5 / 6: u [rows of A_hat^T's operand][d] and t [n][d] of the top layer's first backward hop; 7: the bitmap (as floats' bytes: uint32 words)
  // of the rows of u / t that hop wrote -- under it a row whose bit is clear was NOT written (it is zero by contract, its bytes are stale)
  if (which == 5) return p->u;
  if (which == 6) return p->t;
  if (which == 7) return reinterpret_cast<const float *>(p->nzbits);
  if (which == 8) return p->gab;   // [2 b][d]: the batch rows' input gradients g_ax (first b rows) and g_am, in batch order
  return which == 0 ? p->ax[layer] : which == 1 ? p->am[layer] : which == 2 ? p->p[layer] : which == 3 ? p->dp : which == 4 ? p->dp_b : nullptr;
}
size_t gss_plan_device_bytes(const gss_plan *p) { return p ? p->slab_bytes : 0; }

int gss_plan_lazy_halo_rows(const gss_plan *p, int64_t *out6) {
  GSS_REQUIRE(p && out6, "plan_lazy_halo_rows: null argument");
  out6[0] = p->lz.on ? p->lz.last_recv : -1;
  out6[1] = p->lz.on ? p->lz.last_send : -1;
  out6[2] = p->halo_a.n_halo;
  out6[3] = p->lzt.on ? p->lzt.last_recv : -1;
  out6[4] = p->lzt.on ? p->lzt.last_send : -1;
  out6[5] = p->halo_t.n_halo;
  return GSS_OK;
}

int gss_plan_comm_stats(gss_plan *p, int64_t *out3) {
  GSS_REQUIRE(p && out3, "plan_comm_stats: null argument");
  for (int k = 0; k < 3; ++k) {
    out3[k] = p->n_coll[k];
    p->n_coll[k] = 0;
  }
  return GSS_OK;
}

int gss_plan_sync_stats(gss_plan *p, int64_t *out2) {
  GSS_REQUIRE(p && out2, "plan_sync_stats: null argument");
  for (int k = 0; k < 2; ++k) {
    out2[k] = p->n_sync[k];
    p->n_sync[k] = 0;
  }
  return GSS_OK;
}

int gss_plan_check_guards(gss_plan *p) {
  GSS_REQUIRE(p, "plan_check_guards: null plan");
  GSS_HIP(hipDeviceSynchronize());
  std::vector<unsigned char> host(kGuardBytes);
  for (size_t k = 0; k < p->guard_off.size(); ++k) {
    GSS_HIP(hipMemcpy(host.data(), p->slab + p->guard_off[k], kGuardBytes, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < kGuardBytes; ++i)
      if (host[i] != (unsigned char)kGuardByte)
        return fail(GSS_EINVAL, "plan_check_guards: the guard behind carved buffer %zu (slab offset %zu) was overwritten at byte %zu", k,
                    p->guard_off[k], i);
  }
  return GSS_OK;
}
void gss_plan_set_step(gss_plan *p, int32_t step) {
  if (!p) return;
  p->step = step;
  p->wt_valid = false;  // a restored state: the transposed weight copies no longer match the weights
}
float *gss_plan_adam_buffer(gss_plan *p, int32_t moment, int32_t tensor) {
  if (!p || moment < 0 || moment > 1 || tensor < 0 || tensor > 3) return nullptr;
  return moment == 0 ? p->adam_m[tensor] : p->adam_v[tensor];
}
int32_t gss_plan_get_step(const gss_plan *p) { return p ? p->step : 0; }
int gss_plan_profile(gss_plan *p, int enable) {
  GSS_REQUIRE(p, "plan_profile: null plan");
  p->prof_on = enable != 0;
  return GSS_OK;
}

int gss_plan_profile_read(gss_plan *p, double *ms_out, int64_t *count_out, void *stream) {
  GSS_REQUIRE(p && ms_out && count_out, "plan_profile_read: null argument");
  GSS_HIP(hipStreamSynchronize(as_stream(stream)));
  for (size_t k = 0; k < p->ev_used; ++k) {
    float ms = 0.f;
    GSS_HIP(hipEventElapsedTime(&ms, p->ev[2 * k], p->ev[2 * k + 1]));
    p->prof_ms[p->ev_cls[k]] += ms;
    p->prof_cnt[p->ev_cls[k]] += 1;
  }
  p->ev_used = 0;
  for (int k = 0; k < GSS_PROF_CLASSES; ++k) {
    ms_out[k] = p->prof_ms[k];
    count_out[k] = p->prof_cnt[k];
    p->prof_ms[k] = 0.0;
    p->prof_cnt[k] = 0;
  }
  return GSS_OK;
}

int gss_memcpy_d2d(void *dst, const void *src, size_t bytes, void *stream) {
  GSS_REQUIRE(dst && src, "memcpy_d2d: null pointer");
  GSS_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, as_stream(stream)));
  return GSS_OK;
}
}
