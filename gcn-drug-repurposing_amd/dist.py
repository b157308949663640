"""1-D node-range sharding of the GSS-GCN training step over the GPUs of one node.

The reference is single-process, single-device (train.py:68,118-122); this is the multi-GPU design the north
star asks for.  One process per GPU; rank p owns a contiguous, nnz-balanced range of rows of A_hat and A_hat^T
and the matching rows of every activation and gradient.  Before an SpMM hop the BOUNDARY rows of the dense operand
(the rows of other shards this shard's entries reference) are fetched from their owners over xGMI by one grouped
ncclSend / ncclRecv (gss_exchange_rows); the weights are replicated, their gradients all-reduced; the loss needs the B
batch rows of the embeddings, summed into a [B][d] buffer with one all-reduce.  The step itself is C++
(csrc/plan.hip, gss_plan_create_sharded): this module only prepares what the plan borrows -- partition, halo
layout (Halo), descriptors (ShardLayout) -- and owns the communicator handle (Comm).

Everything here that talks to peers goes through the five methods of `Comm` (allgather_bytes, exchange_rows,
all_reduce_sum_, sync, abort), so the same layout code runs over RCCL, over the in-process backend, and -- in
tests/test_dist_cpu.py, with a gloo communicator and numpy kernels -- as separate processes on a box without a GPU.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import scipy.sparse as sp
import torch

from . import _lib


# ---------------------------------------------------------------------------------------------------------
# partitioning (host, numpy)
# ---------------------------------------------------------------------------------------------------------
def nnz_balanced_ranges(indptr, parts):
    """contiguous row ranges with ~equal stored entries (+1 per row so empty rows still count)"""
    n = len(indptr) - 1
    work = np.asarray(indptr[1:], dtype=np.int64) + np.arange(1, n + 1)
    bounds = [0]
    for p in range(1, parts):
        bounds.append(int(np.searchsorted(work, work[-1] * p / parts)))
    bounds.append(n)
    bounds = np.maximum.accumulate(np.asarray(bounds, dtype=np.int64))
    return bounds


class Partition:
    def __init__(self, bounds):
        self.bounds = np.asarray(bounds, dtype=np.int64)
        self.parts = len(bounds) - 1
        self.n = int(bounds[-1])
        self.max_rows = int(np.diff(self.bounds).max())

    def owner(self, ids):
        return np.searchsorted(self.bounds, ids, side="right") - 1

    def rows(self, rank):
        return int(self.bounds[rank]), int(self.bounds[rank + 1])


# ---------------------------------------------------------------------------------------------------------
# native communicators (include/gssgcn.h gss_comm_*) and the sharded plan
# ---------------------------------------------------------------------------------------------------------
class Comm:
    """owner of one gss_comm handle (RCCL, or the in-process backend); device tensors in, device tensors out, everything on
    torch's current stream"""

    def __init__(self, handle, world, rank):
        self.handle, self.world, self.rank = handle, int(world), int(rank)
        self._destroy = _lib.load().gss_comm_destroy

    def __del__(self):
        h = getattr(self, "handle", None)
        if h:
            self._destroy(h)
            self.handle = None

    def abort(self):
        """release the peers blocked in a collective (call from a rank that failed): the in-process backend opens its host
        barrier, RCCL calls ncclCommAbort -- the communicator is unusable afterwards"""
        if self.handle:
            _lib.load().gss_comm_abort(self.handle)

    def check(self):
        """raises GssError once the backend has seen a failure (ncclCommGetAsyncError)"""
        _lib.check(_lib.load().gss_comm_check(self.handle), "gss_comm_check")

    def count(self) -> int:
        """the number of ranks the backend itself reports (ncclCommCount)"""
        out = C.c_int32()
        _lib.check(_lib.load().gss_comm_count(self.handle, C.byref(out)), "gss_comm_count")
        return int(out.value)

    def sync(self, timeout_s: float = 300.0):
        """wait for torch's current stream; a stream that does not drain within timeout_s (a peer that stopped taking part) aborts
        the communicator and raises instead of hanging"""
        _lib.check(_lib.load().gss_comm_sync(self.handle, _lib.current_stream(), float(timeout_s)), "gss_comm_sync")

    def local_mode(self, mode: int):
        """in-process backend only (gss_comm_local_mode): 1 = record what every collective delivers, 2 = replay it without peers, 0 = normal"""
        _lib.check(_lib.load().gss_comm_local_mode(self.handle, int(mode)), "gss_comm_local_mode")

    def local_log(self):
        """delivered bytes of every recorded collective, in call order (gss_comm_local_log)"""
        n = C.c_int32(0)
        _lib.check(_lib.load().gss_comm_local_log(self.handle, None, 0, C.byref(n)), "gss_comm_local_log")
        out = (C.c_int64 * max(1, n.value))()
        _lib.check(_lib.load().gss_comm_local_log(self.handle, out, n.value, C.byref(n)), "gss_comm_local_log")
        return [int(out[k]) for k in range(n.value)]

    def allgather_bytes(self, src: torch.Tensor) -> torch.Tensor:
        """every rank's `src` (same shape and dtype everywhere), concatenated in rank order -> [world * src.numel()]"""
        src = src.contiguous()
        dst = torch.empty(self.world * src.numel(), dtype=src.dtype, device=src.device)
        _lib.check(_lib.load().gss_allgather_bytes(self.handle, src.data_ptr(), dst.data_ptr(), src.numel() * src.element_size(),
                                                   _lib.current_stream()), "gss_allgather_bytes")
        return dst

    def exchange_rows(self, d, send: torch.Tensor, send_off, recv: torch.Tensor, recv_off):
        """C1, boundary form (gss_exchange_rows): rows [send_off[q], send_off[q+1]) of `send` go to rank q, the rows rank q sends
        land at rows [recv_off[q], recv_off[q+1]) of `recv`; rows are d 4-byte words; offsets are host int64 arrays"""
        send_off = np.ascontiguousarray(send_off, dtype=np.int64)
        recv_off = np.ascontiguousarray(recv_off, dtype=np.int64)
        _lib.check(_lib.load().gss_exchange_rows(self.handle, int(d), send.data_ptr(), send_off.ctypes.data, recv.data_ptr(),
                                                 recv_off.ctypes.data, _lib.current_stream()), "gss_exchange_rows")

    def all_gather_rows(self, src, dst_padded):
        """C1, equal-count form: src [max_rows, d] -> dst_padded [world * max_rows, d]"""
        _lib.check(_lib.load().gss_allgather_rows(self.handle, src.shape[1], src.shape[0], src.data_ptr(), dst_padded.data_ptr(),
                                                  _lib.current_stream()), "gss_allgather_rows")

    def all_reduce_sum_(self, t):
        _lib.check(_lib.load().gss_allreduce_sum(self.handle, t.data_ptr(), t.numel(), _lib.current_stream()), "gss_allreduce_sum")


def rccl_comm(world=None, rank=None):
    """RCCL communicator of this process inside a torch.distributed job: rank 0 draws the unique id, the (gloo or
    nccl) process group only carries its 128 bytes; afterwards the data path never touches torch.distributed."""
    import torch.distributed as dist
    lib = _lib.load()
    world = dist.get_world_size() if world is None else world
    rank = dist.get_rank() if rank is None else rank
    buf = (C.c_char * 128)()
    if rank == 0:
        _lib.check(lib.gss_comm_unique_id(buf), "gss_comm_unique_id")
    box = [bytes(buf)]
    if world > 1:
        dist.broadcast_object_list(box, src=0)
    h = C.c_void_p()
    _lib.check(lib.gss_comm_create_rccl(C.byref(h), int(world), int(rank), box[0]), "gss_comm_create_rccl")
    return Comm(h, world, rank)


# gss_host_xfer_fn (include/gssgcn.h): int fn(void *user, int kind, const void *send, const int64_t *send_off, void *recv,
#                                             const int64_t *recv_off, int64_t count)
_HOST_XFER = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_int64), C.c_void_p, C.POINTER(C.c_int64), C.c_int64)


def _host_bytes(ptr, n):
    """a uint8 tensor over n bytes of (pinned) host memory the library owns"""
    return torch.frombuffer((C.c_uint8 * max(int(n), 1)).from_address(ptr), dtype=torch.uint8)[:int(n)]


def host_comm(world=None, rank=None, group=None):
    """Host-staged communicator (gss_comm_create_host) inside a torch.distributed job whose process group is gloo: the library
    stages every collective through pinned host buffers and this module moves those bytes between the processes.  For boxes
    where RCCL cannot run the job -- several ranks sharing ONE GPU, which RCCL refuses -- so that the real multi-process job
    (torch.distributed.run, one plan per process) can be exercised there.  Select with GSS_COMM_BACKEND=host (job_comm)."""
    import torch.distributed as dist
    world = dist.get_world_size(group) if world is None else int(world)
    rank = dist.get_rank(group) if rank is None else int(rank)

    def xfer(_user, kind, send, send_off, recv, recv_off, count):
        try:
            if kind == 0:                                                   # GSS_HOST_ALLGATHER
                out = _host_bytes(recv, count * world)
                if world == 1:
                    out.copy_(_host_bytes(send, count))
                elif count > 0:
                    dist.all_gather(list(out.view(world, count).unbind(0)), _host_bytes(send, count), group=group)
            elif kind == 1:                                                 # GSS_HOST_ALLTOALLV
                src, dst = _host_bytes(send, send_off[world]), _host_bytes(recv, recv_off[world])
                reqs = []
                for q in range(world):
                    s0, s1, r0, r1 = send_off[q], send_off[q + 1], recv_off[q], recv_off[q + 1]
                    if q == rank:                                            # own range: empty by contract, like the other backends
                        continue
                    if s1 > s0:
                        reqs.append(dist.isend(src[s0:s1], q, group=group))
                    if r1 > r0:
                        reqs.append(dist.irecv(dst[r0:r1], q, group=group))
                for r in reqs:
                    r.wait()
            else:
                return 22
            return 0
        except Exception as e:                                              # never unwind through the C frame
            import sys
            print(f"[gss host comm rank {rank}] transport failed: {type(e).__name__}: {e}", file=sys.stderr, flush=True)
            return 5

    cb = _HOST_XFER(xfer)
    h = C.c_void_p()
    _lib.check(_lib.load().gss_comm_create_host(C.byref(h), world, rank, C.cast(cb, C.c_void_p), None), "gss_comm_create_host")
    comm = Comm(h, world, rank)
    comm._keep = cb                                                         # the C side holds the function pointer
    return comm


def job_comm(world=None, rank=None):
    """the communicator of this process in a torch.distributed job: RCCL (one GPU per rank) unless GSS_COMM_BACKEND=host asks for
    the host-staged backend (ranks may share a GPU; the process group must be gloo)"""
    import os
    backend = os.environ.get("GSS_COMM_BACKEND", "rccl").lower()
    if backend == "rccl":
        return rccl_comm(world, rank)
    if backend == "host":
        return host_comm(world, rank)
    raise ValueError(f"GSS_COMM_BACKEND={backend!r}: expected 'rccl' or 'host'")


def job_device(local_rank):
    """the device index of this process: its local rank, or -- host-staged backend on a box with fewer GPUs than ranks -- that
    modulo the number of GPUs"""
    import os
    if os.environ.get("GSS_COMM_BACKEND", "rccl").lower() == "host":
        return int(local_rank) % max(torch.cuda.device_count(), 1)
    return int(local_rank)


def local_comms(world):
    """`world` communicators for ranks that are threads of this process (gss_comm_create_local)"""
    arr = (C.c_void_p * world)()
    _lib.check(_lib.load().gss_comm_create_local(arr, int(world)), "gss_comm_create_local")
    return [Comm(C.c_void_p(arr[r]), world, r) for r in range(world)]


class _ShardGraph:
    """the (a, at) pair GssEngine expects, for one shard"""

    def __init__(self, a, at, n):
        self.a, self.at, self.n = a, at, n

    @property
    def nnz(self):
        return self.a.nnz


def partition_for(a_hat, world):
    """nnz-balanced node ranges over forward + backward entries"""
    work = a_hat.indptr + sp.csr_matrix(a_hat.T).indptr
    return Partition(nnz_balanced_ranges(work, world))


class Halo:
    """Operand halo of one shard matrix (include/gssgcn.h gss_halo_desc): which rows of other shards its columns reference
    (`remote`, ascending = grouped by owner), where they land behind the shard's own rows, and -- after exchange() -- which
    of its own rows every peer wants."""

    def __init__(self, cols_global, part: Partition, rank, uniq=None):
        self.part, self.rank = part, rank
        lo, hi = part.rows(rank)
        self.lo, self.nl = lo, hi - lo
        uniq = np.unique(np.asarray(cols_global, dtype=np.int64)) if uniq is None else np.asarray(uniq, dtype=np.int64)
        self.remote = uniq[(uniq < lo) | (uniq >= hi)]
        owner = part.owner(self.remote) if len(self.remote) else np.zeros(0, np.int64)
        self.recv_off = np.zeros(part.parts + 1, dtype=np.int64)
        self.recv_off[1:] = np.cumsum(np.bincount(owner, minlength=part.parts))
        self.gid2op = np.full(part.n, -1, dtype=np.int32)          # node id -> operand row (own rows first, then the halo)
        self.gid2op[lo:hi] = np.arange(self.nl, dtype=np.int32)
        self.gid2op[self.remote] = self.nl + np.arange(len(self.remote), dtype=np.int32)
        self.n_halo = int(len(self.remote))
        self.send_off = np.zeros(part.parts + 1, dtype=np.int64)
        self.send_rows = None                                       # device int32, set by exchange()

    def local_cols(self, cols_global):
        return self.gid2op[np.asarray(cols_global, dtype=np.int64)]

    def exchange(self, comm: "Comm", device):
        """tell every owner which of its rows this shard reads (a collective over `comm`: counts by all-gather, the id lists
        by the boundary exchange itself with one int32 per row)"""
        P, rank = self.part.parts, self.rank
        if P == 1:
            self.send_rows = torch.zeros(1, dtype=torch.int32, device=device)
            return self
        mine = torch.from_numpy(np.diff(self.recv_off).astype(np.int64)).to(device)            # [P]: rows I want from q
        allc = comm.allgather_bytes(mine)
        comm.sync()
        counts = allc.cpu().numpy().reshape(P, P)                                              # counts[r][q]: r wants from q
        self.send_off[1:] = np.cumsum(counts[:, rank])
        want = torch.from_numpy(self.remote.astype(np.int32)).to(device) if self.n_halo else torch.zeros(1, dtype=torch.int32, device=device)
        n_send = int(self.send_off[-1])
        got = torch.empty(max(n_send, 1), dtype=torch.int32, device=device)
        # my request list is grouped by owner = my recv layout; what I receive is grouped by requester = my send layout
        comm.exchange_rows(1, want, self.recv_off, got, self.send_off)
        comm.sync()
        self.send_rows = (got - self.lo).contiguous()
        if n_send:
            r = self.send_rows[:n_send]
            assert int(r.min()) >= 0 and int(r.max()) < self.nl, "a peer asked for a row this shard does not own"
        return self

    def c_desc(self):
        return _lib.HaloDesc(self.recv_off.ctypes.data, self.send_off.ctypes.data, self.send_rows.data_ptr())


class ShardLayout:
    """everything gss_plan_create_sharded borrows for one shard; keeps the host arrays and device tensors alive"""

    def __init__(self, part: Partition, rank, halo_a: Halo, halo_at, device, split_a=None, split_at=None, a_loc_t=None):
        self.part, self.rank, self.world = part, rank, part.parts
        self.bounds = np.ascontiguousarray(part.bounds, dtype=np.int64)
        self.halo_a, self.halo_at = halo_a, halo_at
        # (own-column CSR, boundary-column CSR) of A_hat / A_hat^T, or None: with them the plan overlaps a hop with its exchange
        self.split_a, self.split_at = split_a, split_at
        # the shard's A_hat transposed in place ([n + n_halo_a] x [n]), or None: with it the last backward hop needs no exchange of u
        self.a_loc_t = a_loc_t
        self.gid2op_t = torch.from_numpy(halo_at.gid2op).to(device) if halo_at is not None else None
        self._empty = np.zeros(part.parts + 1, dtype=np.int64)

    def c_desc(self):
        none = _lib.HaloDesc(self._empty.ctypes.data, self._empty.ctypes.data, None)
        def h(pair, k):
            return pair[k].handle if pair is not None else None
        return _lib.ShardDesc(self.world, self.rank, self.bounds.ctypes.data, self.halo_a.c_desc(),
                              self.halo_at.c_desc() if self.halo_at is not None else none, _lib.ptr(self.gid2op_t),
                              h(self.split_a, 0), h(self.split_a, 1), h(self.split_at, 0), h(self.split_at, 1),
                              self.a_loc_t.handle if self.a_loc_t is not None else None)

    @property
    def overlapped(self):
        return self.split_a is not None

    def halo_fraction(self):
        """rows received per hop / rows owned by the other shards: (A_hat, A_hat^T)"""
        others = max(1, self.part.n - self.halo_a.nl)
        return self.halo_a.n_halo / others, (self.halo_at.n_halo / others if self.halo_at is not None else 0.0)


def sharded_plan_engine(adj, x_host, params_host, comm, num_layers=2, layer_decay=0.3, alpha=1.0, lr=1e-4, max_batch=None,
                        betas=(0.9, 0.999), eps=1e-8, device=None, a_hat=None, cache_layer1=False, relabel="auto", split="auto"):
    """One rank of the node-range sharded trainer on the NATIVE path: a gss_plan created with gss_plan_create_sharded
    that holds the communicator and enqueues kernels and collectives from C++ (no Python between kernels).  The shard's
    CSRs carry operand-row column ids: own rows first, then the boundary rows its entries reference (Halo).  Returns a
    GssEngine (same methods as the single-GPU one; .global_nnz, .part, .layout added)."""
    from .engine import GssEngine
    from .graph import DeviceCSR
    world, rank = comm.world, comm.rank
    dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    if a_hat is None:
        # the regular path: this rank's rows only, normalised on the device (shards.py) -- the same bits as GssGraph's A_hat
        from .shards import ScipySource, build_shard, shard_engine, shard_rows
        shard = build_shard(ScipySource(adj), comm, need_transpose=num_layers > 1, device=dev, relabel=relabel, split=split)
        return shard_engine(shard, shard_rows(shard, x_host), params_host, comm, num_layers=num_layers, layer_decay=layer_decay, alpha=alpha, lr=lr,
                            max_batch=max_batch, betas=betas, eps=eps, cache_layer1=cache_layer1)
    # a_hat given (an already normalised matrix, e.g. the reference's preprocess_graph output in tests): slice it on the host
    a_hat = sp.csr_matrix(a_hat)
    a_hat.sort_indices()
    part = partition_for(a_hat, world)
    lo, hi = part.rows(rank)
    nl = hi - lo

    def shard_of(m):
        sub = sp.csr_matrix(m[lo:hi])
        halo = Halo(sub.indices, part, rank).exchange(comm, dev)
        sub.sort_indices()
        # entries stay in ascending GLOBAL column order (only the ids are replaced), so a row is summed in the same order as
        # on one GPU and its result is bit-identical whatever the sharding
        return DeviceCSR(sub.indptr.astype(np.int32), halo.local_cols(sub.indices).astype(np.int32), sub.data.astype(np.float32), nl,
                         nl + halo.n_halo, dev), halo

    a, halo_a = shard_of(a_hat)
    at, halo_at = (None, None)
    if num_layers > 1:
        at, halo_at = shard_of(sp.csr_matrix(a_hat.T))
    layout = ShardLayout(part, rank, halo_a, halo_at, dev)
    x = torch.from_numpy(np.ascontiguousarray(x_host[lo:hi], dtype=np.float32)).to(dev)
    params = [torch.from_numpy(np.ascontiguousarray(params_host[k], dtype=np.float32)).to(dev) for k in ("W1", "b1", "W2", "b2")]
    eng = GssEngine(_ShardGraph(a, at, nl), x, params, num_layers=num_layers, layer_decay=layer_decay, alpha=alpha, lr=lr,
                    max_batch=max_batch or a_hat.shape[0], cache_layer1=cache_layer1, betas=betas, eps=eps, shard=layout, comm=comm)
    eng.global_nnz, eng.part, eng.layout = int(a_hat.nnz), part, layout
    return eng
