// comm.hip -- the collectives of the node-range sharded trainer.
//
// The reference is single-process, single-device (train.py:68,118-122); these are the exchange points a
// 1-D node-range sharding of its training step needs:
//   C1  all-gather of the [N][d] operand of every SpMM hop (model.py:163,169 and their autograd)
//   C2  all-reduce (sum) of the 2 (d^2 + d) weight-gradient floats (train.py:183)
//   C3  all-reduce of the B batch rows of the embeddings / of their input gradients (model.py:216-217)
//
// Two backends behind one interface:
//   * RCCL over xGMI: one process per GPU, communicator created from a 128-byte unique id that the host
//     distributes (torch.distributed store, MPI, a file ...).  Collectives are enqueued on the caller's stream.
//   * local: `world` ranks that are THREADS of one process (each with its own stream, possibly all on one
//     GPU).  Exchanges are device-to-device copies fenced by a timed host barrier.  It exists so that the
//     sharded plan can be run and checked at world 2..8 on a box with a single GPU; it is not a fast path.
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "ops.h"

namespace gss {

#define GSS_NCCL(call)                                                                                    \
  do {                                                                                                    \
    ncclResult_t r_ = (call);                                                                             \
    if (r_ != ncclSuccess) return ::gss::fail(GSS_EHIP, "%s:%d %s -> %s", __FILE__, __LINE__, #call, ncclGetErrorString(r_)); \
  } while (0)

// wait for `st` with a deadline (timeout_s <= 0: none): the stream is polled, never blocked on
static int stream_wait(hipStream_t st, double timeout_s, const char *who) {
  const auto t0 = std::chrono::steady_clock::now();
  for (int spins = 0;; ++spins) {
    const hipError_t q = hipStreamQuery(st);
    if (q == hipSuccess) return GSS_OK;
    if (q != hipErrorNotReady) return fail(GSS_EHIP, "%s: hipStreamQuery -> %s", who, hipGetErrorString(q));
    if ((spins & 63) == 63 && timeout_s > 0) {
      const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      if (el > timeout_s) return fail(GSS_ETIMEOUT, "%s: waited %.0f s for the stream", who, el);
    }
    if (spins < 2000) std::this_thread::yield();
    else std::this_thread::sleep_for(std::chrono::microseconds(spins < 20000 ? 20 : 500));
  }
}

// ---- RCCL ------------------------------------------------------------------------------------------------
// Error handling: RCCL reports a failed peer / transport asynchronously.  Every enqueue below ends with a poll of
// ncclCommGetAsyncError; a failure aborts the communicator (ncclCommAbort: kernels already enqueued on the stream stop waiting for
// the dead peer) and is returned to the caller, who exits non-zero -- a process that has touched the GPU never re-execs.  A
// collective that hangs without an error (a rank that stopped calling: mismatched order, a crashed peer whose socket stays open)
// is caught at the caller's synchronisation points by gss_comm_sync, which waits for the stream with a deadline and aborts.
struct RcclComm final : gss_comm {
  // `comm` is read by every enqueue and freed by abort(), which the host may call from another thread (Python's Comm.abort, a
  // watchdog).  `mu` guards the POINTER only and is never held across a call into RCCL: an enqueue can stall on the host inside RCCL
  // (lazy p2p connection setup on first use, a peer that never joins the group), and that is exactly the hang abort() exists to
  // break -- ncclCommAbort raises the communicator's abort flag, which RCCL's host-side wait loops poll, so the stalled call returns
  // with an error.  Enqueues snapshot the pointer under `mu` (Use), abort() takes it out under `mu` and calls ncclCommAbort outside.
  ncclComm_t comm = nullptr;
  std::mutex mu;
  std::atomic<bool> aborted{false};
  std::atomic<bool> failed{false};   // check_async has reported an error: ncclCommDestroy could wait for the dead peer
  std::atomic<int> in_rccl{0};       // enqueues currently inside an RCCL call (the destructor waits for them; abort() for at most 100 ms)
  struct Use {                       // an enqueue's hold on the communicator: pointer snapshot + in-flight count
    RcclComm &c;
    ncclComm_t h = nullptr;
    explicit Use(RcclComm &c_) : c(c_) {
      std::lock_guard<std::mutex> lk(c.mu);
      if (!c.aborted.load()) h = c.comm;
      if (h) c.in_rccl.fetch_add(1);
    }
    ~Use() {
      if (h) c.in_rccl.fetch_sub(1);
    }
  };
  ~RcclComm() override {
    for (int spins = 0; in_rccl.load() > 0 && spins < 5000; ++spins) std::this_thread::sleep_for(std::chrono::milliseconds(1));
    ncclComm_t h = take();
    if (!h) return;
    if (failed.load() || aborted.load())
      (void)ncclCommAbort(h);
    else
      (void)ncclCommDestroy(h);
  }
  bool device_transport() const override { return true; }
  ncclComm_t take() {
    std::lock_guard<std::mutex> lk(mu);
    ncclComm_t h = comm;
    comm = nullptr;
    return h;
  }
  void abort() override { abort_from(0); }
  // ncclCommAbort FREES the communicator, and enqueues on other threads may hold a snapshot of it (Use).  Order: raise `aborted` and take
  // the pointer out first -- from here on every new Use gets no handle --, then give the enqueues that are inside RCCL right now a
  // bounded time to leave (one that is not stalled drains within microseconds: this closes the window between check()'s test of
  // `aborted` and its ncclCommGetAsyncError), and only when that deadline passes abort underneath them -- that is the stalled peer,
  // the case abort() exists for: ncclCommAbort raises the flag RCCL's host-side wait loops poll and the stalled call returns.
  // `own` = enqueues the CALLER itself holds (check() aborts from inside its own Use).
  void abort_from(int own) {
    if (aborted.exchange(true)) return;
    ncclComm_t h = take();
    if (!h) return;
    const auto t0 = std::chrono::steady_clock::now();
    while (in_rccl.load() > own && std::chrono::steady_clock::now() - t0 < std::chrono::milliseconds(100)) std::this_thread::sleep_for(std::chrono::microseconds(50));
    (void)ncclCommAbort(h);
  }
  int dead() const { return fail(GSS_ECOMM, "RCCL communicator of rank %d was aborted after an earlier failure", rank); }
  int check(ncclComm_t h) {
    if (aborted.load()) return dead();   // another thread aborted while this call was inside RCCL: the handle is gone, do not touch it
    ncclResult_t st = ncclSuccess;
    const ncclResult_t r = ncclCommGetAsyncError(h, &st);
    if (r == ncclSuccess && (st == ncclSuccess || st == ncclInProgress)) return GSS_OK;
    if (aborted.load()) return dead();   // another thread aborted while this call was inside RCCL
    const ncclResult_t bad = r != ncclSuccess ? r : st;
    failed.store(true);
    abort_from(1);   // (this call's own Use)
    return fail(GSS_ECOMM, "RCCL asynchronous error on rank %d of %d: %s (communicator aborted)", rank, world, ncclGetErrorString(bad));
  }
  // an RCCL call's own return code: after abort() from another thread it is that abort speaking
  int rccl_rc(ncclResult_t r, const char *what) {
    if (r == ncclSuccess) return GSS_OK;
    if (aborted.load()) return dead();
    return fail(GSS_ECOMM, "%s -> %s", what, ncclGetErrorString(r));
  }
  int check_async() override {
    Use u(*this);
    if (!u.h) return dead();
    return check(u.h);
  }
  int count(int32_t *out) override {
    Use u(*this);
    if (!u.h) return dead();
    int c = 0;
    if (int rc = rccl_rc(ncclCommCount(u.h, &c), "ncclCommCount")) return rc;
    *out = c;
    return GSS_OK;
  }
  int sync(hipStream_t st, double timeout_s) override {
    const auto t0 = std::chrono::steady_clock::now();
    int spins = 0;
    for (;;) {
      const hipError_t q = hipStreamQuery(st);
      if (q == hipSuccess) return check_async();
      if (q != hipErrorNotReady) return fail(GSS_EHIP, "comm_sync: hipStreamQuery -> %s", hipGetErrorString(q));
      if ((++spins & 63) == 0) {
        if (int rc = check_async()) return rc;
        const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (timeout_s > 0 && el > timeout_s) {
          failed.store(true);
          abort();
          return fail(GSS_ETIMEOUT, "comm_sync: rank %d of %d waited %.0f s for its stream (a peer stopped taking part in a collective?); "
                      "communicator aborted", rank, world, el);
        }
      }
      if (spins < 2000) std::this_thread::yield();
      else std::this_thread::sleep_for(std::chrono::microseconds(spins < 20000 ? 20 : 500));
    }
  }
  int all_gather(const void *send, void *recv, size_t bytes_per_rank, hipStream_t st) override {
    Use u(*this);
    if (!u.h) return dead();
    // in place when send == recv + rank * bytes_per_rank (RCCL detects it)
    if (int rc = rccl_rc(ncclAllGather(send, recv, bytes_per_rank, ncclInt8, u.h, st), "ncclAllGather")) return rc;
    return check(u.h);
  }
  int all_reduce_sum(float *const *bufs, const size_t *counts, int nbuf, hipStream_t st) override {
    Use u(*this);
    if (!u.h) return dead();
    // several tensors = one fused RCCL operation (one launch)
    if (nbuf > 1) GSS_NCCL(ncclGroupStart());
    ncclResult_t r = ncclSuccess;
    for (int k = 0; k < nbuf && r == ncclSuccess; ++k) r = ncclAllReduce(bufs[k], bufs[k], counts[k], ncclFloat, ncclSum, u.h, st);
    const ncclResult_t e = nbuf > 1 ? ncclGroupEnd() : ncclSuccess;
    if (int rc = rccl_rc(r != ncclSuccess ? r : e, "ncclAllReduce")) return rc;
    return check(u.h);
  }
  int exchange_rows(const float *send, const int64_t *send_off, float *recv, const int64_t *recv_off, int d, hipStream_t st) override {
    Use u(*this);
    if (!u.h) return dead();
    // halo exchange: one fused group of point-to-point transfers, each over the xGMI link of its pair; pairs with an empty
    // list are skipped
    GSS_NCCL(ncclGroupStart());
    ncclResult_t r = ncclSuccess;
    for (int q = 0; q < world && r == ncclSuccess; ++q) {
      if (q == rank) continue;
      const int64_t ns = send_off[q + 1] - send_off[q], nr = recv_off[q + 1] - recv_off[q];
      if (ns > 0) r = ncclSend(send + (size_t)send_off[q] * d, (size_t)ns * d, ncclFloat, q, u.h, st);
      if (nr > 0 && r == ncclSuccess) r = ncclRecv(recv + (size_t)recv_off[q] * d, (size_t)nr * d, ncclFloat, q, u.h, st);
    }
    const ncclResult_t e = ncclGroupEnd();
    if (int rc = rccl_rc(r != ncclSuccess ? r : e, "halo exchange (ncclSend/ncclRecv group)")) return rc;
    return check(u.h);
  }
};

// ---- local (threads of one process) --------------------------------------------------------------------------
struct LocalShared {
  int world;
  std::mutex mu;
  std::condition_variable cv;
  int arrived = 0;
  uint64_t generation = 0;
  bool broken = false;
  std::vector<const void *> src;
  std::vector<const int64_t *> off;  // exchange_rows: every rank's send offsets (host arrays, valid between the two barriers)
  explicit LocalShared(int w) : world(w), src((size_t)w, nullptr), off((size_t)w, nullptr) {}
  void abort() {
    std::lock_guard<std::mutex> lk(mu);
    broken = true;
    cv.notify_all();
  }
  // timed barrier: a rank that failed elsewhere must not hang the others forever
  int wait() {
    std::unique_lock<std::mutex> lk(mu);
    if (broken) return fail(GSS_EHIP, "local comm: a peer rank failed or timed out");
    const uint64_t gen = generation;
    if (++arrived == world) {
      arrived = 0;
      ++generation;
      cv.notify_all();
      return GSS_OK;
    }
    if (!cv.wait_for(lk, std::chrono::seconds(120), [&] { return generation != gen || broken; })) {
      broken = true;
      cv.notify_all();
      return fail(GSS_EHIP, "local comm: barrier timed out after 120 s (a peer rank never arrived)");
    }
    if (broken) return fail(GSS_EHIP, "local comm: a peer rank failed or timed out");
    return GSS_OK;
  }
};

__global__ __launch_bounds__(256) void local_sum_kernel(size_t count, int world, const float *__restrict__ tmp, float *__restrict__ out) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= count) return;
  float s = tmp[i];
  for (int r = 1; r < world; ++r) s += tmp[(size_t)r * count + i];  // rank order: the same bits on every rank
  out[i] = s;
}

// Record / replay (gss_comm_local_mode; tools/shard_emulation.py --serial): ranks that are threads of one process share ONE GPU, so a step
// timed with all of them running says nothing about one rank's kernels.  Mode 1 keeps a device copy of what every collective DELIVERED to
// this rank, in call order; mode 2 serves those copies again -- same call sequence, same sizes (checked), a device-to-device copy instead of
// the peers, no barrier, no host wait -- so that ONE rank's step can run alone on the GPU and be timed kernel by kernel.  Mode 0 (the
// default) frees the log.  The replayed payloads are the recorded step's (stale by the weight updates since): a timing aid, not a result.
struct CommLog {
  struct Rec {
    void *buf = nullptr;
    size_t bytes = 0;
  };
  int mode = 0;
  std::vector<Rec> recs;
  size_t cursor = 0;
  ~CommLog() { clear(); }
  void clear() {
    for (Rec &r : recs)
      if (r.buf) (void)hipFree(r.buf);
    recs.clear();
    cursor = 0;
  }
  int keep(const void *src, size_t bytes, hipStream_t st) {   // mode 1: after the payload has landed (stream order)
    Rec r;
    r.bytes = bytes;
    if (bytes) {
      GSS_HIP(hipMalloc(&r.buf, bytes));
      const hipError_t e = hipMemcpyAsync(r.buf, src, bytes, hipMemcpyDeviceToDevice, st);
      if (e != hipSuccess) {
        (void)hipFree(r.buf);
        return fail(GSS_EHIP, "local comm record: hipMemcpyAsync -> %s", hipGetErrorString(e));
      }
    }
    recs.push_back(r);
    return GSS_OK;
  }
  int serve(void *dst, size_t bytes, hipStream_t st) {        // mode 2
    if (cursor >= recs.size()) return fail(GSS_EINVAL, "local comm replay: collective %zu was never recorded (%zu in the log)", cursor, recs.size());
    const Rec &r = recs[cursor];
    if (r.bytes != bytes) return fail(GSS_EINVAL, "local comm replay: collective %zu delivers %zu bytes, the recorded one %zu", cursor, bytes, r.bytes);
    ++cursor;
    if (bytes) GSS_HIP(hipMemcpyAsync(dst, r.buf, bytes, hipMemcpyDeviceToDevice, st));
    return GSS_OK;
  }
};

struct LocalComm final : gss_comm {
  std::shared_ptr<LocalShared> sh;
  CommLog log;
  int set_mode(int mode) override {
    if (mode == 0 || mode == 1) log.clear();
    log.cursor = 0;
    log.mode = mode;
    return GSS_OK;
  }
  int log_bytes(int64_t *out, int32_t cap, int32_t *n_out) override {
    *n_out = (int32_t)log.recs.size();
    for (int32_t k = 0; k < cap && k < *n_out; ++k) out[k] = (int64_t)log.recs[(size_t)k].bytes;
    return GSS_OK;
  }
  void abort() override { sh->abort(); }
  int check_async() override {
    std::lock_guard<std::mutex> lk(sh->mu);
    return sh->broken ? fail(GSS_ECOMM, "local comm: a peer rank failed or timed out") : GSS_OK;
  }
  int count(int32_t *out) override {
    *out = world;
    return GSS_OK;
  }
  int sync(hipStream_t st, double timeout_s) override {   // (every collective of this backend already waited behind a timed barrier)
    if (int rc = stream_wait(st, timeout_s, "local comm")) {
      if (rc == GSS_ETIMEOUT) sh->abort();
      return rc;
    }
    return check_async();
  }
  float *tmp = nullptr;
  size_t tmp_floats = 0;
  ~LocalComm() override {
    if (tmp) (void)hipFree(tmp);
  }
  int all_gather(const void *send, void *recv, size_t bytes_per_rank, hipStream_t st) override {
    if (log.mode == 2) return log.serve(recv, bytes_per_rank * (size_t)world, st);
    GSS_HIP(hipStreamSynchronize(st));  // my rows are complete before a peer copies them
    sh->src[(size_t)rank] = send;
    if (int rc = sh->wait()) return rc;
    for (int r = 0; r < world; ++r) {
      char *dst = (char *)recv + (size_t)r * bytes_per_rank;
      if ((const void *)dst != sh->src[(size_t)r]) GSS_HIP(hipMemcpyAsync(dst, sh->src[(size_t)r], bytes_per_rank, hipMemcpyDeviceToDevice, st));
    }
    if (log.mode == 1)
      if (int rc = log.keep(recv, bytes_per_rank * (size_t)world, st)) return rc;
    GSS_HIP(hipStreamSynchronize(st));
    return sh->wait();  // nobody overwrites its send buffer while a peer still reads it
  }
  int exchange_rows(const float *send, const int64_t *send_off, float *recv, const int64_t *recv_off, int d, hipStream_t st) override {
    if (log.mode == 2) return log.serve(recv, sizeof(float) * (size_t)recv_off[world] * d, st);
    GSS_HIP(hipStreamSynchronize(st));  // my packed rows are complete
    sh->src[(size_t)rank] = send;
    sh->off[(size_t)rank] = send_off;
    if (int rc = sh->wait()) return rc;
    for (int q = 0; q < world; ++q) {
      if (q == rank) continue;
      const int64_t nr = recv_off[q + 1] - recv_off[q];
      if (nr <= 0) continue;
      const int64_t theirs = sh->off[(size_t)q][rank + 1] - sh->off[(size_t)q][rank];
      if (theirs != nr) return fail(GSS_EINVAL, "local comm: rank %d expects %lld rows from rank %d, which sends %lld", rank, (long long)nr, q, (long long)theirs);
      const float *src = (const float *)sh->src[(size_t)q] + (size_t)sh->off[(size_t)q][rank] * d;
      GSS_HIP(hipMemcpyAsync(recv + (size_t)recv_off[q] * d, src, sizeof(float) * (size_t)nr * d, hipMemcpyDeviceToDevice, st));
    }
    if (log.mode == 1)
      if (int rc = log.keep(recv, sizeof(float) * (size_t)recv_off[world] * d, st)) return rc;
    GSS_HIP(hipStreamSynchronize(st));
    return sh->wait();
  }
  int all_reduce_sum(float *const *bufs, const size_t *counts, int nbuf, hipStream_t st) override {
    if (log.mode == 2) {
      for (int k = 0; k < nbuf; ++k)
        if (int rc = log.serve(bufs[k], sizeof(float) * counts[k], st)) return rc;
      return GSS_OK;
    }
    for (int k = 0; k < nbuf; ++k) {
      const size_t count = counts[k];
      if (tmp_floats < count * (size_t)world) {
        if (tmp) GSS_HIP(hipFree(tmp));
        tmp = nullptr;
        GSS_HIP(hipMalloc((void **)&tmp, sizeof(float) * count * (size_t)world));
        tmp_floats = count * (size_t)world;
      }
      GSS_HIP(hipStreamSynchronize(st));
      sh->src[(size_t)rank] = bufs[k];
      if (int rc = sh->wait()) return rc;
      for (int r = 0; r < world; ++r)
        GSS_HIP(hipMemcpyAsync(tmp + (size_t)r * count, sh->src[(size_t)r], sizeof(float) * count, hipMemcpyDeviceToDevice, st));
      GSS_HIP(hipStreamSynchronize(st));
      if (int rc = sh->wait()) return rc;  // every rank holds its copy of all contributions: buffers may change now
      if (count) {
        hipLaunchKernelGGL(local_sum_kernel, dim3(ceil_div((int64_t)count, 256)), dim3(256), 0, st, count, world, tmp, bufs[k]);
        GSS_LAUNCH_CHECK("local_sum_kernel");
      }
      if (log.mode == 1)
        if (int rc = log.keep(bufs[k], sizeof(float) * count, st)) return rc;
    }
    return GSS_OK;
  }
};

// ---- host-staged (a transport callback supplied by the host language) --------------------------------------------------------
// One process per rank like RCCL, but the bytes travel device -> pinned host buffer -> callback -> pinned host buffer -> device.
// The callback is whatever the host has (the Python package passes torch.distributed's gloo).  Purpose: RCCL refuses two ranks on
// one device, so on a ONE-GPU box the real multi-PROCESS job -- torch.distributed.run, train.py --ngpus N, bench.py --gpus N, every
// rank its own process and plan -- can still run with all ranks sharing the GPU (GSS_COMM_BACKEND=host).  Same interface, same
// results (exchanges are copies; sums are taken in rank order, identical on every rank); not a fast path.
struct HostComm final : gss_comm {
  gss_host_xfer_fn fn = nullptr;
  void *user = nullptr;
  bool broken = false;
  char *hsend = nullptr, *hrecv = nullptr;
  size_t cap_send = 0, cap_recv = 0;
  ~HostComm() override {
    if (hsend) (void)hipHostFree(hsend);
    if (hrecv) (void)hipHostFree(hrecv);
  }
  void abort() override { broken = true; }
  int check_async() override { return broken ? fail(GSS_ECOMM, "host-staged communicator of rank %d failed earlier", rank) : GSS_OK; }
  int count(int32_t *out) override {
    *out = world;
    return GSS_OK;
  }
  int sync(hipStream_t st, double timeout_s) override {
    if (int rc = stream_wait(st, timeout_s, "host-staged comm")) {
      if (rc == GSS_ETIMEOUT) broken = true;
      return rc;
    }
    return check_async();
  }
  int reserve(size_t ns, size_t nr) {
    if (ns > cap_send) {
      if (hsend) GSS_HIP(hipHostFree(hsend));
      hsend = nullptr;
      GSS_HIP(hipHostMalloc((void **)&hsend, ns, hipHostMallocDefault));
      cap_send = ns;
    }
    if (nr > cap_recv) {
      if (hrecv) GSS_HIP(hipHostFree(hrecv));
      hrecv = nullptr;
      GSS_HIP(hipHostMalloc((void **)&hrecv, nr, hipHostMallocDefault));
      cap_recv = nr;
    }
    return GSS_OK;
  }
  int call(int kind, const int64_t *soff, const int64_t *roff, int64_t count) {
    if (int rc = check_async()) return rc;
    const int rc = fn(user, kind, hsend, soff, hrecv, roff, count);
    if (rc != 0) {
      broken = true;
      return fail(GSS_ECOMM, "host-staged communicator: the transport callback failed (kind %d, code %d)", kind, rc);
    }
    return GSS_OK;
  }
  int all_gather(const void *send, void *recv, size_t bytes_per_rank, hipStream_t st) override {
    if (int rc = reserve(bytes_per_rank ? bytes_per_rank : 1, bytes_per_rank * (size_t)world + 1)) return rc;
    GSS_HIP(hipMemcpyAsync(hsend, send, bytes_per_rank, hipMemcpyDeviceToHost, st));
    GSS_HIP(hipStreamSynchronize(st));
    if (int rc = call(GSS_HOST_ALLGATHER, nullptr, nullptr, (int64_t)bytes_per_rank)) return rc;
    GSS_HIP(hipMemcpyAsync(recv, hrecv, bytes_per_rank * (size_t)world, hipMemcpyHostToDevice, st));
    GSS_HIP(hipStreamSynchronize(st));   // the staging buffer is reused by the next call
    return GSS_OK;
  }
  int all_reduce_sum(float *const *bufs, const size_t *counts, int nbuf, hipStream_t st) override {
    for (int k = 0; k < nbuf; ++k) {
      const size_t bytes = counts[k] * sizeof(float);
      if (bytes == 0) continue;
      if (int rc = reserve(bytes, bytes * (size_t)world)) return rc;
      GSS_HIP(hipMemcpyAsync(hsend, bufs[k], bytes, hipMemcpyDeviceToHost, st));
      GSS_HIP(hipStreamSynchronize(st));
      // every rank's contribution, then the sum in rank order on the host: the same bits on every rank
      if (int rc = call(GSS_HOST_ALLGATHER, nullptr, nullptr, (int64_t)bytes)) return rc;
      float *acc = reinterpret_cast<float *>(hsend);
      const float *all = reinterpret_cast<const float *>(hrecv);
      for (size_t i = 0; i < counts[k]; ++i) {
        float sum = all[i];
        for (int r = 1; r < world; ++r) sum += all[(size_t)r * counts[k] + i];
        acc[i] = sum;
      }
      GSS_HIP(hipMemcpyAsync(bufs[k], hsend, bytes, hipMemcpyHostToDevice, st));
      GSS_HIP(hipStreamSynchronize(st));
    }
    return GSS_OK;
  }
  int exchange_rows(const float *send, const int64_t *send_off, float *recv, const int64_t *recv_off, int d, hipStream_t st) override {
    const size_t row = sizeof(float) * (size_t)d;
    const size_t ns = (size_t)send_off[world] * row, nr = (size_t)recv_off[world] * row;
    if (int rc = reserve(ns ? ns : 1, nr ? nr : 1)) return rc;
    if (ns) GSS_HIP(hipMemcpyAsync(hsend, send, ns, hipMemcpyDeviceToHost, st));
    GSS_HIP(hipStreamSynchronize(st));
    std::vector<int64_t> sb((size_t)world + 1), rb((size_t)world + 1);
    for (int q = 0; q <= world; ++q) {
      sb[(size_t)q] = send_off[q] * (int64_t)row;
      rb[(size_t)q] = recv_off[q] * (int64_t)row;
    }
    if (int rc = call(GSS_HOST_ALLTOALLV, sb.data(), rb.data(), 0)) return rc;
    if (nr) GSS_HIP(hipMemcpyAsync(recv, hrecv, nr, hipMemcpyHostToDevice, st));
    GSS_HIP(hipStreamSynchronize(st));
    return GSS_OK;
  }
};

}  // namespace gss

using namespace gss;

extern "C" {

int gss_comm_create_host(gss_comm **out, int32_t world, int32_t rank, gss_host_xfer_fn fn, void *user) {
  GSS_REQUIRE(out && fn && world >= 1 && rank >= 0 && rank < world, "comm_create_host: bad argument (world=%d rank=%d)", world, rank);
  HostComm *c = new HostComm();
  c->world = world;
  c->rank = rank;
  c->fn = fn;
  c->user = user;
  *out = c;
  return GSS_OK;
}

int gss_comm_unique_id(void *id_out) {
  GSS_REQUIRE(id_out, "comm_unique_id: null pointer");
  static_assert(sizeof(ncclUniqueId) == GSS_COMM_ID_BYTES, "GSS_COMM_ID_BYTES must match ncclUniqueId");
  ncclUniqueId id;
  GSS_NCCL(ncclGetUniqueId(&id));
  memcpy(id_out, &id, sizeof(id));
  return GSS_OK;
}

int gss_comm_create_rccl(gss_comm **out, int32_t world, int32_t rank, const void *id) {
  GSS_REQUIRE(out && id && world >= 1 && rank >= 0 && rank < world, "comm_create_rccl: bad argument (world=%d rank=%d)", world, rank);
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof(uid));
  RcclComm *c = new RcclComm();
  c->world = world;
  c->rank = rank;
  const ncclResult_t r = ncclCommInitRank(&c->comm, world, uid, rank);  // binds to the calling thread's current device
  if (r != ncclSuccess) {
    c->comm = nullptr;
    delete c;
    return fail(GSS_EHIP, "ncclCommInitRank(world=%d, rank=%d) -> %s", world, rank, ncclGetErrorString(r));
  }
  *out = c;
  return GSS_OK;
}

int gss_comm_create_local(gss_comm **out, int32_t world) {
  GSS_REQUIRE(out && world >= 1 && world <= 64, "comm_create_local: bad argument (world=%d)", world);
  auto sh = std::make_shared<LocalShared>(world);
  for (int r = 0; r < world; ++r) {
    LocalComm *c = new LocalComm();
    c->world = world;
    c->rank = r;
    c->sh = sh;
    out[r] = c;
  }
  return GSS_OK;
}

int gss_comm_local_mode(gss_comm *c, int32_t mode) {
  GSS_REQUIRE(c && mode >= 0 && mode <= 2, "comm_local_mode: bad argument");
  GSS_REQUIRE(c->set_mode(mode) == GSS_OK, "comm_local_mode: only the in-process backend (gss_comm_create_local) records and replays");
  return GSS_OK;
}
int gss_comm_local_log(gss_comm *c, int64_t *bytes_out, int32_t cap, int32_t *n_out) {
  GSS_REQUIRE(c && n_out && (cap == 0 || bytes_out), "comm_local_log: bad argument");
  GSS_REQUIRE(c->log_bytes(bytes_out, cap, n_out) == GSS_OK, "comm_local_log: only the in-process backend keeps a log");
  return GSS_OK;
}
void gss_comm_destroy(gss_comm *c) { delete c; }
void gss_comm_abort(gss_comm *c) {
  if (c) c->abort();
}
int gss_comm_check(gss_comm *c) {
  GSS_REQUIRE(c, "comm_check: null communicator");
  return c->check_async();
}
int gss_comm_count(gss_comm *c, int32_t *count_out) {
  GSS_REQUIRE(c && count_out, "comm_count: null argument");
  return c->count(count_out);
}
int gss_comm_sync(gss_comm *c, void *stream, double timeout_s) {
  GSS_REQUIRE(c, "comm_sync: null communicator");
  return c->sync(as_stream(stream), timeout_s);
}
int32_t gss_comm_world(const gss_comm *c) { return c ? c->world : 0; }
int32_t gss_comm_rank(const gss_comm *c) { return c ? c->rank : -1; }

int gss_allgather_rows(gss_comm *c, int32_t d, int32_t max_rows, const float *src, float *dst_padded, void *stream) {
  GSS_REQUIRE(c && src && dst_padded && d > 0 && max_rows >= 0, "allgather_rows: bad argument");
  if (max_rows == 0) return GSS_OK;
  return c->all_gather(src, dst_padded, sizeof(float) * (size_t)max_rows * d, as_stream(stream));
}

int gss_allgather_bytes(gss_comm *c, const void *src, void *dst, size_t bytes_per_rank, void *stream) {
  GSS_REQUIRE(c && src && dst, "allgather_bytes: bad argument");
  if (bytes_per_rank == 0) return GSS_OK;
  return c->all_gather(src, dst, bytes_per_rank, as_stream(stream));
}

int gss_allreduce_sum(gss_comm *c, float *buf, int64_t count, void *stream) {
  GSS_REQUIRE(c && buf && count >= 0, "allreduce_sum: bad argument");
  if (count == 0) return GSS_OK;
  const size_t cnt = (size_t)count;
  return c->all_reduce_sum(&buf, &cnt, 1, as_stream(stream));
}

int gss_exchange_rows(gss_comm *c, int32_t d, const float *send, const int64_t *h_send_off, float *recv, const int64_t *h_recv_off, void *stream) {
  GSS_REQUIRE(c && d > 0 && h_send_off && h_recv_off, "exchange_rows: bad argument");
  GSS_REQUIRE((send || h_send_off[c->world] == h_send_off[0]) && (recv || h_recv_off[c->world] == h_recv_off[0]), "exchange_rows: null buffer");
  if (c->world == 1) return GSS_OK;
  return c->exchange_rows(send, h_send_off, recv, h_recv_off, d, as_stream(stream));
}
}
