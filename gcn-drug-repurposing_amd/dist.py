"""1-D node-range sharding of the GSS-GCN training step over the GPUs of one node.

The reference is single-process, single-device (train.py:68,118-122); this is the multi-GPU design the north
star asks for.  One process per GPU; rank p owns a contiguous, nnz-balanced range of rows of A_hat and A_hat^T
(all columns) and the matching rows of every activation and gradient.  Per SpMM hop the dense operand is
all-gathered over xGMI (RCCL); the weights are replicated, their gradients all-reduced; the loss needs the B
batch rows of the embeddings, summed into a [B][d] buffer with one all-reduce.

Exchange layout: all-gathers land in a *padded* [P][max_rows][d] buffer (all_gather_into_tensor needs equal
shard sizes), so the column ids of the local CSRs are remapped once to padded ids
    col' = owner(col) * max_rows + (col - lo[owner(col)])
and the SpMM kernels gather straight out of the receive buffer -- no compaction pass.

`ShardedEngine` is written against two small interfaces so that its exchange logic can be tested without
8 GPUs: a communicator (TorchComm = torch.distributed; ThreadComm = P threads in one process) and an op backend
(HipOps = libgssgcn.so; the tests plug a numpy backend to run world_size-2 gloo on CPU).
"""
from __future__ import annotations

import ctypes as C
import threading

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
        # every shard is padded to a multiple of 4 rows only through max_rows; ids below are padded ids

    def owner(self, ids):
        return np.searchsorted(self.bounds, ids, side="right") - 1

    def padded_id(self, ids):
        ids = np.asarray(ids, dtype=np.int64)
        o = self.owner(ids)
        return o * self.max_rows + (ids - self.bounds[o])

    def rows(self, rank):
        return int(self.bounds[rank]), int(self.bounds[rank + 1])


def shard_csr(a_hat, part: Partition, rank):
    """rows [lo, hi) of a scipy CSR with padded column ids -> (indptr, indices, data)"""
    lo, hi = part.rows(rank)
    sub = sp.csr_matrix(a_hat[lo:hi])
    sub.sort_indices()
    return sub.indptr.astype(np.int32), part.padded_id(sub.indices).astype(np.int32), sub.data.astype(np.float32)


# ---------------------------------------------------------------------------------------------------------
# communicators
# ---------------------------------------------------------------------------------------------------------
class TorchComm:
    """torch.distributed (backend "nccl" == RCCL over xGMI on the GPU box, "gloo" in CPU tests)"""

    def __init__(self):
        import torch.distributed as dist
        self.dist = dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()

    def all_gather_rows(self, src, dst_padded):
        """src [max_rows, d] (rows beyond the shard are don't-care) -> dst [world * max_rows, d]"""
        self.dist.all_gather_into_tensor(dst_padded, src)

    def all_reduce_sum_(self, t):
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)

    def barrier(self):
        self.dist.barrier()


class ThreadComm:
    """P threads of one process as P ranks (tests: both 'ranks' may share one GPU)."""

    class Shared:
        def __init__(self, world):
            self.world = world
            self.barrier = threading.Barrier(world)
            self.slots = [None] * world

    def __init__(self, shared, rank):
        self.shared, self.rank, self.world = shared, rank, shared.world

    def _exchange(self, t):
        if t.is_cuda:
            torch.cuda.current_stream().synchronize()
        self.shared.slots[self.rank] = t
        self.shared.barrier.wait()
        got = list(self.shared.slots)
        return got

    def all_gather_rows(self, src, dst_padded):
        got = self._exchange(src)
        m = src.shape[0]
        for r, t in enumerate(got):
            dst_padded[r * m:(r + 1) * m].copy_(t)
        if dst_padded.is_cuda:
            torch.cuda.current_stream().synchronize()
        self.shared.barrier.wait()

    def all_reduce_sum_(self, t):
        got = self._exchange(t.clone())
        acc = got[0].clone()
        for o in got[1:]:
            acc += o          # rank order: identical result on every rank
        t.copy_(acc)
        if t.is_cuda:
            torch.cuda.current_stream().synchronize()
        self.shared.barrier.wait()

    def barrier(self):
        self.shared.barrier.wait()


# ---------------------------------------------------------------------------------------------------------
# native communicators (include/gssgcn.h gss_comm_*) and the sharded plan
# ---------------------------------------------------------------------------------------------------------
class Comm:
    """owner of one gss_comm handle"""

    def __init__(self, handle, world, rank):
        self.handle, self.world, self.rank = handle, int(world), int(rank)
        self._destroy = _lib.load().gss_comm_destroy

    def __del__(self):
        h = getattr(self, "handle", None)
        if h:
            self._destroy(h)
            self.handle = None

    def abort(self):
        """in-process ranks: release the peers blocked in a collective (call from a rank that failed)"""
        if self.handle:
            _lib.load().gss_comm_abort(self.handle)

    def all_gather_rows(self, src, dst_padded):
        """C1: src [max_rows, d] -> dst_padded [world * max_rows, d] on torch's current stream"""
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
    """nnz-balanced node ranges over forward + backward entries (the same rule as ShardedEngine)"""
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
        by gss_exchange_rows with one int32 per row)"""
        P, rank = self.part.parts, self.rank
        lib = _lib.load()
        st = _lib.current_stream
        if P == 1:
            self.send_rows = torch.zeros(1, dtype=torch.int32, device=device)
            return self
        mine = torch.from_numpy(np.diff(self.recv_off).astype(np.int64)).to(device)            # [P]: rows I want from q
        allc = torch.empty(P, P, dtype=torch.int64, device=device)
        _lib.check(lib.gss_allgather_bytes(comm.handle, mine.data_ptr(), allc.data_ptr(), 8 * P, st()), "gss_allgather_bytes")
        torch.cuda.current_stream().synchronize()
        counts = allc.cpu().numpy()                                                            # counts[r][q]: r wants from q
        self.send_off[1:] = np.cumsum(counts[:, rank])
        want = torch.from_numpy(self.remote.astype(np.int32)).to(device) if self.n_halo else torch.zeros(1, dtype=torch.int32, device=device)
        n_send = int(self.send_off[-1])
        got = torch.empty(max(n_send, 1), dtype=torch.int32, device=device)
        # my request list is grouped by owner = my recv layout; what I receive is grouped by requester = my send layout
        _lib.check(lib.gss_exchange_rows(comm.handle, 1, want.data_ptr(), self.recv_off.ctypes.data, got.data_ptr(),
                                         self.send_off.ctypes.data, st()), "gss_exchange_rows")
        torch.cuda.current_stream().synchronize()
        self.send_rows = (got - self.lo).contiguous()
        if n_send:
            r = self.send_rows[:n_send]
            assert int(r.min()) >= 0 and int(r.max()) < self.nl, "a peer asked for a row this shard does not own"
        return self

    def c_desc(self):
        return _lib.HaloDesc(self.recv_off.ctypes.data, self.send_off.ctypes.data, self.send_rows.data_ptr())


class ShardLayout:
    """everything gss_plan_create_sharded borrows for one shard; keeps the host arrays and device tensors alive"""

    def __init__(self, part: Partition, rank, halo_a: Halo, halo_at, device):
        self.part, self.rank, self.world = part, rank, part.parts
        self.bounds = np.ascontiguousarray(part.bounds, dtype=np.int64)
        self.halo_a, self.halo_at = halo_a, halo_at
        self.gid2op_t = torch.from_numpy(halo_at.gid2op).to(device) if halo_at is not None else None
        self._empty = np.zeros(part.parts + 1, dtype=np.int64)

    def c_desc(self):
        none = _lib.HaloDesc(self._empty.ctypes.data, self._empty.ctypes.data, None)
        return _lib.ShardDesc(self.world, self.rank, self.bounds.ctypes.data, self.halo_a.c_desc(),
                              self.halo_at.c_desc() if self.halo_at is not None else none, _lib.ptr(self.gid2op_t))

    def halo_fraction(self):
        """rows received per hop / rows owned by the other shards: (A_hat, A_hat^T)"""
        others = max(1, self.part.n - self.halo_a.nl)
        return self.halo_a.n_halo / others, (self.halo_at.n_halo / others if self.halo_at is not None else 0.0)


def sharded_plan_engine(adj, x_host, params_host, comm, num_layers=2, layer_decay=0.3, alpha=1.0, lr=1e-4, max_batch=None,
                        betas=(0.9, 0.999), eps=1e-8, device=None, a_hat=None, cache_layer1=False, relabel="auto"):
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
        shard = build_shard(ScipySource(adj), comm, need_transpose=num_layers > 1, device=dev, relabel=relabel)
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


# ---------------------------------------------------------------------------------------------------------
# op backend over the C ABI
# ---------------------------------------------------------------------------------------------------------
class HipOps:
    """thin per-op wrappers over include/gssgcn.h; every tensor is a contiguous CUDA fp32/int32 tensor"""

    def __init__(self, device):
        self.lib = _lib.load()
        self.device = torch.device(device)
        self._ws = {}

    def st(self):
        return _lib.current_stream()

    def empty(self, *shape, dtype=torch.float32):
        return torch.empty(*shape, dtype=dtype, device=self.device)

    def zeros(self, *shape, dtype=torch.float32):
        return torch.zeros(*shape, dtype=dtype, device=self.device)

    def tensor(self, a, dtype=None):
        t = torch.from_numpy(np.ascontiguousarray(a))
        return (t.to(dtype) if dtype is not None else t).to(self.device)

    def csr(self, indptr, indices, data, n_rows, n_cols):
        from .graph import DeviceCSR
        return DeviceCSR(indptr, indices, data, n_rows, n_cols, self.device)

    def spmm(self, csr, x_full, h=None):
        d = x_full.shape[1]
        y = self.empty(csr.n_rows, d)
        m = self.empty(csr.n_rows, d) if h is not None else None
        _lib.check(self.lib.gss_spmm(csr.handle, d, x_full.data_ptr(), y.data_ptr(), _lib.ptr(h), _lib.ptr(m), self.st()), "gss_spmm")
        return y, m

    def dense_fwd(self, ax, am, w, p_prev, decay):
        n, d = ax.shape
        p, xn = self.empty(n, d), self.empty(n, d)
        _lib.check(self.lib.gss_dense_fwd(n, d, ax.data_ptr(), am.data_ptr(), w[0].data_ptr(), w[1].data_ptr(), w[2].data_ptr(),
                                          w[3].data_ptr(), _lib.ptr(p_prev), float(decay), p.data_ptr(), xn.data_ptr(), self.st()),
                   "gss_dense_fwd")
        return p, xn

    def rownorm_fwd(self, x):
        n, d = x.shape
        e, inv = self.empty(n, d), self.empty(n)
        _lib.check(self.lib.gss_rownorm_fwd(n, d, x.data_ptr(), e.data_ptr(), inv.data_ptr(), self.st()), "gss_rownorm_fwd")
        return e, inv

    def loss_fwd_bwd(self, e_b, beta, alpha):
        b, d = e_b.shape
        idx = torch.arange(b, dtype=torch.int32, device=self.device)
        loss, de = self.empty(1), self.empty(b, d)
        ws = torch.empty(self.lib.gss_loss_workspace_bytes(b, d), dtype=torch.uint8, device=self.device)
        _lib.check(self.lib.gss_loss_fwd_bwd(b, d, e_b.data_ptr(), idx.data_ptr(), b, float(beta), float(alpha), loss.data_ptr(),
                                             de.data_ptr(), ws.data_ptr(), self.st()), "gss_loss_fwd_bwd")
        return loss, de

    def rownorm_elu_bwd(self, de_rows, rows, e, inv_den, p, c):
        b, d = de_rows.shape
        dx, dp = self.empty(b, d), self.empty(b, d)
        if b:
            _lib.check(self.lib.gss_rownorm_elu_bwd(d, de_rows.data_ptr(), rows.data_ptr(), b, e.data_ptr(), inv_den.data_ptr(),
                                                    p.data_ptr(), float(c), dx.data_ptr(), dp.data_ptr(), self.st()), "gss_rownorm_elu_bwd")
        return dx, dp

    def wgrad(self, dp, ax, am, rows, grads, accumulate):
        n, d = dp.shape
        if n == 0:
            if not accumulate:
                for g in grads:
                    g.zero_()
            return
        need = self.lib.gss_wgrad_workspace_bytes(n, d)
        if self._ws.get("wgrad") is None or self._ws["wgrad"].numel() < need:
            self._ws["wgrad"] = torch.empty(need, dtype=torch.uint8, device=self.device)   # grow-only
        _lib.check(self.lib.gss_dense_bwd_weight(n, d, dp.data_ptr(), ax.data_ptr(), am.data_ptr(), _lib.ptr(rows), grads[0].data_ptr(),
                                                 grads[2].data_ptr(), grads[1].data_ptr(), 1 if accumulate else 0,
                                                 self._ws["wgrad"].data_ptr(), self.st()), "gss_dense_bwd_weight")
        grads[3].copy_(grads[1])

    def dgrad(self, dp, w1t, w2t):
        n, d = dp.shape
        gax, gam = self.empty(n, d), self.empty(n, d)
        if n:
            _lib.check(self.lib.gss_dense_bwd_input(n, d, dp.data_ptr(), w1t.data_ptr(), w2t.data_ptr(), None, gax.data_ptr(),
                                                    gam.data_ptr(), self.st()), "gss_dense_bwd_input")
        return gax, gam

    def spmm_bwd1(self, at, gam_full, gax_local, x_in, ax):
        d = x_in.shape[1]
        u, t = self.empty(at.n_rows, d), self.empty(at.n_rows, d)
        _lib.check(self.lib.gss_spmm_bwd1(at.handle, d, gam_full.data_ptr(), gax_local.data_ptr(), x_in.data_ptr(), ax.data_ptr(),
                                          u.data_ptr(), t.data_ptr(), self.st()), "gss_spmm_bwd1")
        return u, t

    def spmm_bwd1_sparse(self, at, gam_b, gax_b, pos_col, pos_row, x_in, ax):
        d = x_in.shape[1]
        u, t = self.empty(at.n_rows, d), self.empty(at.n_rows, d)
        _lib.check(self.lib.gss_spmm_bwd1_sparse(at.handle, d, gam_b.data_ptr(), gax_b.data_ptr(), pos_col.data_ptr(), pos_row.data_ptr(),
                                                 x_in.data_ptr(), ax.data_ptr(), u.data_ptr(), t.data_ptr(), self.st()),
                   "gss_spmm_bwd1_sparse")
        return u, t

    def spmm_bwd2(self, at, u_full, t, p, c, res, want_gx):
        d = t.shape[1]
        dp = self.empty(at.n_rows, d)
        gx = self.empty(at.n_rows, d) if want_gx else None
        _lib.check(self.lib.gss_spmm_bwd2(at.handle, d, u_full.data_ptr(), t.data_ptr(), p.data_ptr(), float(c), _lib.ptr(res),
                                          dp.data_ptr(), _lib.ptr(gx), self.st()), "gss_spmm_bwd2")
        return dp, gx

    def scatter_add_rows(self, src, rows, dst):
        b, d = src.shape
        if b:
            _lib.check(self.lib.gss_scatter_add_rows(d, src.data_ptr(), rows.data_ptr(), b, dst.data_ptr(), self.st()), "gss_scatter_add_rows")

    def batch_maps(self, idx32, lo, nl, bounds_dev, world, maxr):
        """index maps of one batch on this shard (gss_shard_batch_maps) -> rows_all, rows_own, keep [b,1], pos_col, pos_row"""
        b = idx32.numel()
        rows_all, rows_own = self.empty(b, dtype=torch.int32), self.empty(b, dtype=torch.int32)
        keep = self.empty(b, 1)
        pos_col, pos_row = self.empty(world * maxr, dtype=torch.int32), self.empty(max(nl, 1), dtype=torch.int32)
        _lib.check(self.lib.gss_shard_batch_maps(idx32.data_ptr(), b, int(lo), int(nl), bounds_dev.data_ptr(), int(world), int(maxr),
                                                 rows_all.data_ptr(), rows_own.data_ptr(), keep.data_ptr(), pos_col.data_ptr(),
                                                 pos_row.data_ptr(), self.st()), "gss_shard_batch_maps")
        return rows_all, rows_own, keep, pos_col, pos_row

    def adam(self, params, grads, m, v, step, lr, betas, eps):
        for k in range(4):
            _lib.check(self.lib.gss_adam_step(params[k].numel(), params[k].data_ptr(), grads[k].data_ptr(), m[k].data_ptr(),
                                              v[k].data_ptr(), step, float(lr), float(betas[0]), float(betas[1]), float(eps), None, 0,
                                              self.st()), "gss_adam_step")


# ---------------------------------------------------------------------------------------------------------
# the sharded training step
# ---------------------------------------------------------------------------------------------------------
class ShardedEngine:
    """One rank of the node-range-sharded trainer.  Same step semantics as GssEngine.step (train.py:158-184)."""

    def __init__(self, adj, x_host, params_host, num_layers=2, layer_decay=0.3, alpha=1.0, lr=1e-4, max_batch=None,
                 betas=(0.9, 0.999), eps=1e-8, comm=None, ops=None, device=None, a_hat=None):
        self.comm = comm or TorchComm()
        rank, world = self.comm.rank, self.comm.world
        dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.ops = ops or HipOps(dev)
        n, d = x_host.shape
        self.n, self.d, self.L = n, d, int(num_layers)
        self.decay, self.alpha, self.lr, self.betas, self.eps = float(layer_decay), float(alpha), float(lr), betas, float(eps)
        # A_hat: every rank normalises the (small) host matrix the same way; values as in preprocess_graph
        if a_hat is None:
            a_hat = self._normalize_host(adj)
        a_hat = sp.csr_matrix(a_hat)
        a_hat.sort_indices()
        self.global_nnz = int(a_hat.nnz)
        work = a_hat.indptr + sp.csr_matrix(a_hat.T).indptr      # forward + backward entries per row prefix
        self.part = Partition(nnz_balanced_ranges(work, world))
        self.lo, self.hi = self.part.rows(rank)
        self.nl, self.maxr = self.hi - self.lo, self.part.max_rows
        ops = self.ops
        ip, ix, dv = shard_csr(a_hat, self.part, rank)
        self.a = ops.csr(ip, ix, dv, self.nl, world * self.maxr)
        self.at = None
        if self.L > 1:
            ip, ix, dv = shard_csr(sp.csr_matrix(a_hat.T), self.part, rank)
            self.at = ops.csr(ip, ix, dv, self.nl, world * self.maxr)
        self.x0 = ops.tensor(x_host[self.lo:self.hi].astype(np.float32))
        self.params = [ops.tensor(params_host[k].astype(np.float32)) for k in ("W1", "b1", "W2", "b2")]
        # the four gradients are views of one flat buffer: one all-reduce, no packing
        self._grad_flat = torch.zeros(sum(p.numel() for p in self.params), dtype=torch.float32, device=self.params[0].device)
        self.grads, o = [], 0
        for p in self.params:
            self.grads.append(self._grad_flat[o:o + p.numel()].view_as(p))
            o += p.numel()
        self.m = [torch.zeros_like(p) for p in self.params]
        self.v = [torch.zeros_like(p) for p in self.params]
        self.step_no = 0
        self.loss = ops.zeros(1)
        self.emb = None
        self._send = ops.zeros(self.maxr, d)                 # padded send buffer
        self._full = [ops.empty(world * self.maxr, d) for _ in range(2)]
        self._bounds_dev = ops.tensor(self.part.bounds, dtype=torch.int64)
        self._x0_full = None

    @staticmethod
    def _normalize_host(adj):
        # helpers/helper.py:82-89 in fp64, then the fp32 cast of :95 happens in shard_csr
        adj = sp.csr_matrix(adj, dtype=np.float64)
        a_ = adj + sp.eye(adj.shape[0], dtype=np.float64, format="csr")
        rowsum = np.asarray(a_.sum(1)).reshape(-1)
        dinv = sp.diags(np.power(rowsum, -0.5))
        return sp.csr_matrix(a_.dot(dinv).transpose().dot(dinv).transpose())

    # -- collectives ------------------------------------------------------------------------------------
    def _gather(self, local, which=0):
        """all-gather the local rows [nl, d] into the padded [world*maxr, d] receive buffer"""
        self._send[:self.nl].copy_(local)
        self.comm.all_gather_rows(self._send, self._full[which])
        return self._full[which]

    def _padded_ids(self, ids64):
        o = torch.searchsorted(self._bounds_dev, ids64, right=True) - 1
        return o * self.maxr + (ids64 - self._bounds_dev[o])

    # -- one iteration ------------------------------------------------------------------------------------
    def forward(self):
        ops, L = self.ops, self.L
        self.act = []
        x, p_prev = self.x0, None
        for l in range(L):
            if l == 0:
                if self._x0_full is None:                              # the input features never change: distribute them once
                    self._x0_full = self._gather(x, 0).clone()
                xf = self._x0_full
            else:
                xf = self._gather(x, 0)
            ax, m = ops.spmm(self.a, xf, h=x)                          # model.py:163,168
            mf = self._gather(m, 1)
            am, _ = ops.spmm(self.a, mf)                               # model.py:169
            p, xn = ops.dense_fwd(ax, am, self.params, p_prev, self.decay)   # model.py:165,170-173,201-203
            self.act.append({"x": x, "ax": ax, "am": am, "p": p})
            x, p_prev = xn, p
        self.emb, self.inv_den = ops.rownorm_fwd(x)                    # model.py:205
        return self.emb

    def loss_backward(self, idx32, beta, count=None, offset=0):
        """Fixed-shape on purpose: every rank handles all b batch rows, with the rows it does not own masked to zero
        (a clamped row index, zero gradient), so nothing here depends on a device value and the host never waits for
        the GPU inside a step."""
        ops, L, d = self.ops, self.L, self.d
        b = int(count if count is not None else idx32.numel())
        rows_all, rows_own, keep, pos_col, pos_row = ops.batch_maps(idx32[offset:offset + b], self.lo, self.nl, self._bounds_dev,
                                                                    self.comm.world, self.maxr)
        # E_B on every rank: each rank contributes its rows, one all-reduce (model.py:216-217)
        e_b = self.emb.index_select(0, rows_all.long()) * keep
        self.comm.all_reduce_sum_(e_b)
        loss, de_b = ops.loss_fwd_bwd(e_b, beta, self.alpha)           # model.py:218-221 (+ autograd); same on every rank
        self.loss = loss
        top = self.act[L - 1]
        c_top = self.decay if L > 1 else 1.0
        dx_b, dp_b = ops.rownorm_elu_bwd(de_b * keep, rows_all, self.emb, self.inv_den, top["p"], c_top)   # zero rows where not owned
        ops.wgrad(dp_b, top["ax"], top["am"], rows_all, self.grads, accumulate=False)
        if L > 1:
            w1t, w2t = self.params[0].t().contiguous(), self.params[2].t().contiguous()
            gax_b, gam_b = ops.dgrad(dp_b, w1t, w2t)                    # [b][d] in batch order, zero where not owned
            both = torch.cat([gax_b, gam_b])
            self.comm.all_reduce_sum_(both)                            # batch-row gradients of every rank
            gax_b, gam_b = both[:b], both[b:]
            u, t = ops.spmm_bwd1_sparse(self.at, gam_b, gax_b, pos_col, pos_row, top["x"], top["ax"])
            gx_prev = None       # g_x(lp + 2) as a dense local tensor, once lp + 2 <= L - 1
            for lp in range(L - 2, -1, -1):
                lay = self.act[lp]
                c = 1.0 if lp == 0 else self.decay
                uf = self._gather(u, 0)
                dp, gx = ops.spmm_bwd2(self.at, uf, t, lay["p"], c, gx_prev if lp + 2 <= L - 1 else None, want_gx=lp >= 1)
                if lp + 2 == L:
                    ops.scatter_add_rows(dx_b, rows_own, dp)
                ops.wgrad(dp, lay["ax"], lay["am"], None, self.grads, accumulate=True)
                if lp >= 1:
                    gax, gam = ops.dgrad(dp, w1t, w2t)
                    gf = self._gather(gam, 1)
                    u, t = ops.spmm_bwd1(self.at, gf, gax, lay["x"], lay["ax"])
                gx_prev = gx
        self.comm.all_reduce_sum_(self._grad_flat)                     # C2: 2 (d^2 + d) floats

    def adam(self):
        self.step_no += 1
        self.ops.adam(self.params, self.grads, self.m, self.v, self.step_no, self.lr, self.betas, self.eps)

    def step(self, idx32, beta, count=None, offset=0):
        self.forward()
        self.loss_backward(idx32, beta, count, offset)
        self.adam()

    def gather_embeddings(self):
        """full [N][d] embeddings on every rank (row order == node order)"""
        full = self._gather(self.emb, 0)
        out = [full[r * self.maxr: r * self.maxr + (self.part.rows(r)[1] - self.part.rows(r)[0])] for r in range(self.comm.world)]
        return torch.cat(out)
