"""Building one node-range shard of A_hat / A_hat^T without any rank holding the whole graph.

The reference normalises one scipy matrix on the host (helpers/helper.py:82-89).  Here every rank
  1. learns the per-node work (stored entries of its row of A + I and of (A + I)^T) from a *row source* and derives the
     same nnz-balanced node ranges as every other rank,
  2. takes only ITS rows of A + I and of (A + I)^T from the source (global column ids, fp64 values),
  3. computes D_ii^-1/2 of its rows on the device (gss_rowsum_dinv), all-gathers the N scalars, and scales its entries
     with the single-GPU kernel's rounding sequence (gss_scale_adj_shard) -- bit-identical values to GssGraph's,
  4. finds the boundary rows its entries reference (dist.Halo), tells their owners, and renumbers its columns to
     operand rows (own rows first, then the halo).
Row sources: ScipySource (a matrix in memory: small graphs, tests) and RmatSource (BASELINE config 5: the RMAT edge
stream generated chunk by chunk on the GPU, every rank keeping only the edges of its own rows; host memory per rank is
O(N) scalars, not O(nnz)).
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp
import torch

from . import _lib
from .dist import Comm, Halo, Partition, ShardLayout, nnz_balanced_ranges


# ---------------------------------------------------------------------------------------------------------
# small host collectives over a gss_comm (setup time only; device buffers underneath, so both backends serve them)
# ---------------------------------------------------------------------------------------------------------
def allgather_host(comm: Comm, arr: np.ndarray, device) -> np.ndarray:
    """[world, *arr.shape] of every rank's `arr` (same shape and dtype everywhere)"""
    arr = np.ascontiguousarray(arr)
    if comm.world == 1:
        return arr[None].copy()
    src = torch.from_numpy(arr.view(np.uint8).reshape(-1)).to(device)
    dst = comm.allgather_bytes(src)
    comm.sync()
    return dst.cpu().numpy().view(arr.dtype).reshape((comm.world,) + arr.shape)


def allgather_ranges(comm: Comm, local: torch.Tensor, bounds, device) -> torch.Tensor:
    """local [rows of this rank] (1-D device tensor) of every rank, concatenated in node order -> [N] on the device"""
    P = comm.world
    if P == 1:
        return local
    sizes = np.diff(np.asarray(bounds, dtype=np.int64))
    maxr = int(max(1, sizes.max()))
    pad = torch.zeros(maxr, dtype=local.dtype, device=device)
    pad[:local.numel()] = local
    out = comm.allgather_bytes(pad)
    return torch.cat([out[r * maxr: r * maxr + int(sizes[r])] for r in range(P)])


class NativeShardOps:
    """the device side of build_shard: libgssgcn.so kernels and CSR handles.  (tests/cpu_ops.py has the numpy stand-in with
    which tests/test_dist_cpu.py drives build_shard in gloo processes on a box without a GPU.)"""

    def __init__(self):
        self.lib = _lib.load()

    def rowsum_dinv(self, nl, rowptr, val, dev):
        dinv = torch.empty(max(nl, 1), dtype=torch.float64, device=dev)
        rowsum = torch.empty(max(nl, 1), dtype=torch.float64, device=dev)
        _lib.check(self.lib.gss_rowsum_dinv(nl, rowptr.data_ptr(), _lib.ptr(val), dinv.data_ptr(), rowsum.data_ptr(), _lib.current_stream()),
                   "gss_rowsum_dinv")
        return dinv, rowsum

    def rowsum_check(self, nl, rowsum):
        from .graph import rowsum_check
        return rowsum_check(rowsum, nl)

    def scale_adj(self, nl, lo, rowptr, col, val, dinv, transposed, dev):
        val32 = torch.empty(max(col.numel(), 1), dtype=torch.float32, device=dev)
        _lib.check(self.lib.gss_scale_adj_shard(nl, lo, rowptr.data_ptr(), _lib.ptr(col), _lib.ptr(val), dinv.data_ptr(), transposed,
                                                val32.data_ptr(), _lib.current_stream()), "gss_scale_adj_shard")
        return val32

    def csr(self, rowptr_host, col_local, val32, n_rows, n_cols, dev):
        from .graph import DeviceCSR
        return DeviceCSR(rowptr_host, col_local, val32, n_rows, n_cols, dev)

    def set_hot(self, csr, own_hot, halo_begin, halo_end):
        _lib.check(self.lib.gss_csr_set_hot(csr.handle, own_hot, halo_begin, halo_end), "gss_csr_set_hot")


# ---------------------------------------------------------------------------------------------------------
# row sources
# ---------------------------------------------------------------------------------------------------------
class ScipySource:
    """rows of A + I and of (A + I)^T from a matrix in memory (every rank holds it: for graphs that fit, and for tests)"""

    def __init__(self, adj):
        adj = sp.csr_matrix(adj, dtype=np.float64)
        self.n = adj.shape[0]
        self.a = (adj + sp.eye(self.n, dtype=np.float64, format="csr")).tocsr()     # helper.py:83
        self.a.sort_indices()
        self.at = self.a.T.tocsr()
        self.at.sort_indices()
        self.nnz = int(self.a.nnz)

    def work(self, comm, device):
        return (np.diff(self.a.indptr) + np.diff(self.at.indptr)).astype(np.int64)

    def _rows(self, m, lo, hi, device, relabel):
        if relabel is None:
            sub = m[lo:hi]
            cols = sub.indices
        else:
            # rows lo..hi of the relabelled matrix = old rows perm[lo:hi]; entries stay in ascending OLD column order
            sub = m[relabel.perm[lo:hi]]
            sub.sort_indices()
            cols = relabel.inv[sub.indices]
        return (torch.from_numpy(sub.indptr.astype(np.int32)).to(device), torch.from_numpy(cols.astype(np.int32)).to(device),
                torch.from_numpy(sub.data.astype(np.float64)).to(device))

    def rows(self, lo, hi, device, relabel=None):
        return self._rows(self.a, lo, hi, device, relabel)

    def rows_t(self, lo, hi, device, relabel=None):
        return self._rows(self.at, lo, hi, device, relabel)


def _mix64(x: torch.Tensor) -> torch.Tensor:
    """splitmix64 finaliser on int64 tensors (wrapping multiplies, logical shifts), result masked to 63 bits"""
    def lsr(v, s):
        return (v >> s) & ((1 << (64 - s)) - 1)
    x = (x ^ lsr(x, 30)) * (-4658895280553007687)       # 0xbf58476d1ce4e5b9
    x = (x ^ lsr(x, 27)) * (-7723592293110705685)       # 0x94d049bb133111eb
    x = x ^ lsr(x, 31)
    return x & 0x7FFFFFFFFFFFFFFF


class RmatSource:
    """BASELINE config 5: RMAT (a, b, c, d) on `n` node ids (sampled on the next power of two, endpoints >= n and self loops
    rejected), `m` unique directed entries with unit weights, + the n self loops of A + I.  The sample stream comes from a
    seeded device generator and is therefore the same on every rank; a rank keeps only the samples of its own rows.
    Exactly m entries: samples are drawn in rounds until at least m distinct ones exist (counted over all ranks), then the
    ones with the smallest 63-bit hash of (u, v) are kept -- a cut every rank can apply on its own."""

    def __init__(self, n, m, seed=4, abcd=(0.57, 0.19, 0.19, 0.05), device="cuda", chunk=None):
        self.n, self.m, self.seed, self.abcd = int(n), int(m), int(seed), abcd
        self.device = torch.device(device)
        self.bits = int(np.ceil(np.log2(self.n)))
        self.chunk = int(chunk or min(1 << 26, max(1 << 16, int(self.m * 0.35))))
        self.rounds = None
        self.cut = None            # keep entries with hash < cut
        self.nnz = self.m + self.n
        self._cache = None         # (lo, hi, keys_by_u, keys_by_v) of the preparation pass, reused when the ranges match

    # -- the sample stream ---------------------------------------------------------------------------------
    def _chunks(self, rounds=None):
        g = torch.Generator(device=self.device)
        g.manual_seed(self.seed)
        a, b, c, _ = self.abcd
        r = 0
        while rounds is None or r < rounds:
            k = self.chunk
            u = torch.zeros(k, dtype=torch.int64, device=self.device)
            v = torch.zeros(k, dtype=torch.int64, device=self.device)
            for _ in range(self.bits):
                p = torch.rand(k, generator=g, device=self.device)
                right = ((p >= a) & (p < a + b)) | (p >= a + b + c)
                down = p >= a + b
                u = (u << 1) | down.to(torch.int64)
                v = (v << 1) | right.to(torch.int64)
            keep = (u < self.n) & (v < self.n) & (u != v)
            yield u[keep], v[keep]
            r += 1

    def _collect(self, lo, hi, rounds, comm=None, want=None, inv=None):
        """distinct samples with u in [lo, hi) (keyed u * n + v) and with v in [lo, hi) (keyed v * n + u), over `rounds` rounds,
        or -- rounds None -- until the job holds at least `want` distinct samples.  inv (relabelled graph): the row ids are
        inv[u] / inv[v] (range test and key), the column half of the key stays the generator's id"""
        ku = torch.zeros(0, dtype=torch.int64, device=self.device)
        kv = torch.zeros(0, dtype=torch.int64, device=self.device)
        used = 0
        for u, v in self._chunks(rounds):
            ru, rv = (u, v) if inv is None else (inv[u], inv[v])
            mu, mv = (ru >= lo) & (ru < hi), (rv >= lo) & (rv < hi)
            ku = torch.unique(torch.cat([ku, ru[mu] * self.n + v[mu]]))
            kv = torch.unique(torch.cat([kv, rv[mv] * self.n + u[mv]]))
            used += 1
            if rounds is None:
                total = int(allgather_host(comm, np.array([ku.numel()], dtype=np.int64), self.device).sum())
                if total >= want:
                    break
                if used > 64:
                    raise RuntimeError(f"RMAT: {total} distinct entries after {used} rounds, {want} wanted (graph too dense for its id space)")
        return ku, kv, used

    def _uv_key(self, k_by_v, perm=None):
        """(v * n + u) -> (u * n + v), the key the hash is taken of (generator ids: a relabelled row id goes through perm)"""
        v = k_by_v // self.n
        return (k_by_v % self.n) * self.n + (v if perm is None else perm[v])

    def _u_key(self, k_by_u, perm=None):
        return k_by_u if perm is None else perm[k_by_u // self.n] * self.n + k_by_u % self.n

    def prepare(self, comm: Comm):
        """collective: number of rounds, the hash cut that leaves exactly m entries, and every node's work"""
        P, rank = comm.world, comm.rank
        lo, hi = rank * self.n // P, (rank + 1) * self.n // P
        ku, kv, used = self._collect(lo, hi, None, comm, self.m)
        self.rounds = used
        hu = _mix64(ku)
        # smallest cut with count(hash < cut) >= m, by bisection over the 63-bit range (counts summed over the ranks)
        a_, b_ = 0, 1 << 63
        while a_ < b_:
            mid = (a_ + b_) // 2
            cnt = int(allgather_host(comm, np.array([int((hu < mid).sum())], dtype=np.int64), self.device).sum())
            if cnt >= self.m:
                b_ = mid
            else:
                a_ = mid + 1
        self.cut = a_
        ku = ku[hu < self.cut]
        kv = kv[_mix64(self._uv_key(kv)) < self.cut]
        total = int(allgather_host(comm, np.array([ku.numel()], dtype=np.int64), self.device).sum())
        self.nnz = total + self.n
        deg = (torch.bincount(ku // self.n - lo, minlength=hi - lo) + torch.bincount(kv // self.n - lo, minlength=hi - lo) + 2)
        bounds = np.array([r * self.n // P for r in range(P + 1)], dtype=np.int64)
        self._work = allgather_ranges(comm, deg, bounds, self.device).cpu().numpy().astype(np.int64)
        self._cache = ((lo, hi, False), ku, kv)
        return self

    def work(self, comm, device):
        if self.rounds is None:
            self.prepare(comm)
        return self._work

    def _csr(self, keys, lo, hi, relabel):
        """distinct keys (row * n + generator column id, rows in [lo, hi)) + the diagonal -> CSR of those rows of A + I (unit
        values); a row's entries are in ascending generator-id order, the column ids returned are the relabelled ones"""
        r = torch.arange(lo, hi, dtype=torch.int64, device=self.device)
        diag = r * self.n + (r if relabel is None else relabel.perm_dev(self.device)[r])
        keys = torch.sort(torch.cat([keys, diag])).values
        rows = keys // self.n - lo
        rowptr = torch.zeros(hi - lo + 1, dtype=torch.int64, device=self.device)
        rowptr[1:] = torch.cumsum(torch.bincount(rows, minlength=hi - lo), 0)
        cols = keys % self.n
        if relabel is not None:
            cols = relabel.inv_dev(self.device)[cols]
        return rowptr.to(torch.int32), cols.to(torch.int32), torch.ones(keys.numel(), dtype=torch.float64, device=self.device)

    def _keys(self, lo, hi, relabel):
        tag = (lo, hi, relabel is not None)
        if self._cache is not None and self._cache[0] == tag:
            return self._cache[1], self._cache[2]
        inv = relabel.inv_dev(self.device) if relabel is not None else None
        perm = relabel.perm_dev(self.device) if relabel is not None else None
        ku, kv, _ = self._collect(lo, hi, self.rounds, inv=inv)
        ku = ku[_mix64(self._u_key(ku, perm)) < self.cut]
        kv = kv[_mix64(self._uv_key(kv, perm)) < self.cut]
        self._cache = (tag, ku, kv)
        return ku, kv

    def rows(self, lo, hi, device, relabel=None):
        return self._csr(self._keys(lo, hi, relabel)[0], lo, hi, relabel)

    def rows_t(self, lo, hi, device, relabel=None):
        return self._csr(self._keys(lo, hi, relabel)[1], lo, hi, relabel)

    def release(self):
        self._cache = None


class KnnSource:
    """train.py's own adjacency -- gen_graph's "descriptor" kNN graph (helpers/helper.py:39-53) -- as a row source: every rank
    computes the top-k of ITS row window on the device (gss_knn_topk_rows; fp64 MFMA similarity against all N columns), the
    [N][k] table of (neighbour, similarity) pairs is all-gathered (N k 12 bytes: 1.8 MB at N = 29,960, k = 5), and a rank's rows of
    A + I are assembled from it: row i holds topk(i) and every j with i in topk(j) (x_adj[i, top] = v, x_adj[top, i] = v, the later
    iteration wins; zero diagonal, exact zeros dropped -- graph._symmetrise_topk's rule on the rows of one range).  The graph is
    symmetric, so rows_t = rows.  X itself ([N][d] fp64) is needed whole on every rank: the similarity of a row runs against all
    columns -- that is the reference's algorithm, O(N^2 d), not a property of this builder."""

    def __init__(self, x_all, k, device="cuda"):
        x = np.ascontiguousarray(x_all, dtype=np.float64)
        self.n, d = x.shape
        self.k = int(min(k, self.n))
        if d % 8:
            x = np.concatenate([x, np.zeros((self.n, 8 - d % 8))], axis=1)
        self.device = torch.device(device)
        self._x = torch.from_numpy(x).to(self.device)
        self.ti = self.tv = None
        self.nnz = None

    def prepare(self, comm: Comm):
        P, rank, n, k = comm.world, comm.rank, self.n, self.k
        lo, hi = rank * n // P, (rank + 1) * n // P                     # an even split: this phase is N^2 d / P flops per rank
        lib = _lib.load()
        tv = torch.full((max(hi - lo, 1), k), float("-inf"), dtype=torch.float64, device=self.device)
        ti = torch.full((max(hi - lo, 1), k), -1, dtype=torch.int32, device=self.device)
        _lib.check(lib.gss_knn_topk_rows(n, self._x.shape[1], self._x.data_ptr(), k, lo, hi, tv.data_ptr(), ti.data_ptr(), _lib.current_stream()),
                   "gss_knn_topk_rows")
        bounds = np.array([r * n // P for r in range(P + 1)], dtype=np.int64)
        # (allgather_ranges moves 1-D tensors: k columns, one at a time)
        self.tv = torch.stack([allgather_ranges(comm, tv[:hi - lo, c].contiguous(), bounds, self.device) for c in range(k)], dim=1)
        self.ti = torch.stack([allgather_ranges(comm, ti[:hi - lo, c].contiguous(), bounds, self.device) for c in range(k)], dim=1)
        self._x = None
        # every rank derives the same entry list (N k pairs) and from it every node's work
        i = torch.arange(n, device=self.device).repeat_interleave(k)
        j = self.ti.reshape(-1).long()
        v = self.tv.reshape(-1)
        keep = (j >= 0) & (j != i)
        i, j, v = i[keep], j[keep], v[keep]
        r = torch.cat([i, j])
        c = torch.cat([j, i])
        it = torch.cat([i, i])
        vv = torch.cat([v, v])
        # (row, col) duplicates: the write of the later iteration wins (helper.py:48-50)
        key = r * n + c
        order = torch.argsort(key * n + it)          # by (row, col), then iteration  (n^3 < 2^63 for n < 2 M nodes: kNN graphs are small)
        ks = key[order]
        last = torch.ones_like(ks, dtype=torch.bool)
        last[:-1] = ks[1:] != ks[:-1]
        sel = order[last]
        sel = sel[vv[sel] != 0]                      # adj.eliminate_zeros()
        self._r, self._c, self._v = r[sel], c[sel], vv[sel]          # sorted by (row, col)
        deg = torch.bincount(self._r, minlength=n) + 1                 # + the diagonal of A + I
        self._work = (2 * deg).cpu().numpy().astype(np.int64)
        self.nnz = int(self._r.numel()) + n
        return self

    def work(self, comm, device):
        if self.ti is None:
            self.prepare(comm)
        return self._work

    def rows(self, lo, hi, device, relabel=None):
        n = self.n
        if relabel is None:
            m = (self._r >= lo) & (self._r < hi)
            rr, cc_old, vv = self._r[m] - lo, self._c[m], self._v[m]
        else:
            inv = relabel.inv_dev(self.device)
            nr = inv[self._r]
            m = (nr >= lo) & (nr < hi)
            rr, cc_old, vv = nr[m] - lo, self._c[m], self._v[m]
        # + the diagonal (A + I, helper.py:83); a row's entries in ascending OLD column order, column ids relabelled
        dr = torch.arange(hi - lo, device=self.device)
        d_old = (dr + lo) if relabel is None else relabel.perm_dev(self.device)[dr + lo]
        rr = torch.cat([rr, dr])
        cc_old = torch.cat([cc_old, d_old])
        vv = torch.cat([vv, torch.ones(hi - lo, dtype=torch.float64, device=self.device)])
        order = torch.argsort(rr * n + cc_old)
        rr, cc_old, vv = rr[order], cc_old[order], vv[order]
        rowptr = torch.zeros(hi - lo + 1, dtype=torch.int64, device=self.device)
        rowptr[1:] = torch.cumsum(torch.bincount(rr, minlength=hi - lo), 0)
        cols = cc_old if relabel is None else relabel.inv_dev(self.device)[cc_old]
        return rowptr.to(torch.int32), cols.to(torch.int32), vv.contiguous()

    rows_t = rows                                    # the kNN graph is symmetric by construction


class EdgelistSource:
    """--adj-file: the directed weighted edgelist (embio.read_edgelist: src, dst, w over the .embs.txt rows) as a row source.  A rank
    keeps the entries of its own rows of A + I and of (A + I)^T only; a repeated (u, v) keeps the last weight (DiGraph.add_edge),
    a self loop in the file adds to the diagonal's 1, and an entry whose final value is 0 -- a zero weight in the file, a self loop of
    weight -1 -- is not stored, as scipy's `adj + eye` drops it (graph.edgelist_adj + helper.py:83): the same stored entries, nnz(A_hat),
    row work and partition as ScipySource(edgelist_adj(...))."""

    def __init__(self, src, dst, w, n, device="cuda"):
        self.n = int(n)
        self.device = torch.device(device)
        src, dst, w = np.asarray(src, np.int64), np.asarray(dst, np.int64), np.asarray(w, np.float64)
        key = src * self.n + dst
        order = np.argsort(key, kind="stable")
        ks = key[order]
        last = order[np.r_[ks[1:] != ks[:-1], True]] if len(ks) else order
        src, dst, w = src[last], dst[last], w[last]                                      # distinct (u, v), sorted by (u, v)
        diag = src == dst
        self._loop = np.zeros(self.n, dtype=bool)                                        # rows with a self loop in the file: their diagonal
        self._loop[src[diag]] = True                                                     # entry is w + 1 (or nothing), not the inserted 1
        w = w + diag                                                                     # A + I
        keep = w != 0
        self._src, self._dst, self._w = src[keep], dst[keep], w[keep]
        diag = diag[keep]
        has_diag = ~self._loop
        has_diag[self._src[diag]] = True
        self.nnz = int(len(self._src) - diag.sum() + has_diag.sum())
        deg_out = np.bincount(self._src[~diag], minlength=self.n)
        deg_in = np.bincount(self._dst[~diag], minlength=self.n)
        self._work = (deg_out + deg_in + 2 * has_diag).astype(np.int64)                  # entries of the node's row of A + I and of its transpose

    def work(self, comm, device):
        return self._work

    def _rows(self, rows_of, cols_of, lo, hi, relabel):
        n = self.n
        r_new = rows_of if relabel is None else relabel.inv[rows_of]
        m = (r_new >= lo) & (r_new < hi)
        rr, cc_old, vv = r_new[m] - lo, cols_of[m], self._w[m]
        d_old = np.arange(lo, hi) if relabel is None else relabel.perm[lo:hi]
        miss = np.flatnonzero(~self._loop[d_old])                                        # rows without a self loop in the file: the inserted 1
        rr = np.concatenate([rr, miss])
        cc_old = np.concatenate([cc_old, d_old[miss]])
        vv = np.concatenate([vv, np.ones(len(miss))])
        order = np.argsort(rr * n + cc_old, kind="stable")
        rr, cc_old, vv = rr[order], cc_old[order], vv[order]
        rowptr = np.zeros(hi - lo + 1, dtype=np.int64)
        rowptr[1:] = np.cumsum(np.bincount(rr, minlength=hi - lo))
        cols = cc_old if relabel is None else relabel.inv[cc_old]
        dev = self.device
        return (torch.from_numpy(rowptr.astype(np.int32)).to(dev), torch.from_numpy(cols.astype(np.int32)).to(dev),
                torch.from_numpy(np.ascontiguousarray(vv, dtype=np.float64)).to(dev))

    def rows(self, lo, hi, device, relabel=None):
        return self._rows(self._src, self._dst, lo, hi, relabel)

    def rows_t(self, lo, hi, device, relabel=None):
        return self._rows(self._dst, self._src, lo, hi, relabel)


def gaussian_rows(lo, hi, d, seed, block=1 << 16):
    """rows [lo, hi) of a unit-variance Gaussian feature matrix that any rank can generate for its own range: block k
    (rows k * block ...) comes from RandomState(seed * 1000003 + k)"""
    out = np.empty((hi - lo, d), dtype=np.float32)
    for k in range(lo // block, (max(hi, lo + 1) - 1) // block + 1):
        b0, b1 = k * block, (k + 1) * block
        s0, s1 = max(lo, b0), min(hi, b1)
        if s1 <= s0:
            continue
        rows = np.random.RandomState((seed * 1000003 + k) % (2 ** 32)).randn(block, d).astype(np.float32)
        out[s0 - lo:s1 - lo] = rows[s0 - b0:s1 - b0]
    return out


# ---------------------------------------------------------------------------------------------------------
# the builder
# ---------------------------------------------------------------------------------------------------------
class Relabel:
    """hub-first node order: perm[new id] = old id by descending work (stored entries of the node's row of A + I and of its
    transpose), ties in old-id order; inv = its inverse.  Every rank derives the same permutation from the same work array.
    Why: with the hubs' feature rows contiguous at the head of the operand, the gathers of an SpMM over a table far larger
    than the caches run 20-30 % faster (tools/spmm_hot_cold.py, profiles/r02_spmm_hot_cold_*.txt)."""

    def __init__(self, work):
        work = np.asarray(work, dtype=np.int64)
        self.perm = np.argsort(-work, kind="stable").astype(np.int64)
        self.inv = np.empty_like(self.perm)
        self.inv[self.perm] = np.arange(len(work), dtype=np.int64)
        self._dev = {}

    def perm_dev(self, device):
        key = ("perm", str(device))
        if key not in self._dev:
            self._dev[key] = torch.from_numpy(self.perm).to(device)
        return self._dev[key]

    def inv_dev(self, device):
        key = ("inv", str(device))
        if key not in self._dev:
            self._dev[key] = torch.from_numpy(self.inv).to(device)
        return self._dev[key]


RELABEL_MIN_NODES = 16384   # relabel="auto": from here on the hub-first order pays (config 2, N = 29,960: -3 % per SpMM, -1.7 % per step;
                            # RMAT 10M: -31 % per SpMM); below it the graphs are test-sized
HOT_ROWS = 65536     # the hubs whose rows are kept in the caches (32 MB at d = 128): measured optimum at RMAT 10M / 200M


class Shard:
    """one rank's part of the graph on the device: a / at (DeviceCSR with operand-row column ids), layout, part; relabel
    (or None) + node_map (device int32 [N], original id -> row) when the nodes were relabelled"""

    def __init__(self, a, at, layout, part, nnz_global, rowsum, relabel=None, node_map=None, split_a=None, split_at=None):
        self.a, self.at, self.layout, self.part, self.nnz_global, self.rowsum = a, at, layout, part, nnz_global, rowsum
        self.relabel, self.node_map = relabel, node_map
        # (own-column CSR, boundary-column CSR) of each matrix, or None: what an overlapped hop multiplies while / after the exchange
        self.split_a, self.split_at = split_a, split_at
        self.n = a.n_rows

    @property
    def nnz(self):
        return self.a.nnz


ROW_WEIGHT = 12      # build_shard's default cost of a row besides its stored entries, in stored entries of A_hat + A_hat^T (callers that know
                     # the width and depth pass row_weight_for(d, num_layers) instead)


def row_weight_for(d, num_layers):
    """What a row costs a training step besides its stored entries, in units of one stored entry of A_hat + A_hat^T: the dense kernels
    (projection, input gradient, weight gradient: 3 L - 1 passes over the rows, d^2 work each) against the SpMMs (4 L - 2 passes over the
    entries, d work each).  Measured with each rank alone on the GPU (tools/scaling_forecast.py, RMAT 10M / 200M, d = 128, L = 2, world 8;
    profiles/r05_scaling_forecast_c5*.json): a least-squares fit gives 0.124 ns per entry, 3.6 ns per own row and 1.6 ns per BOUNDARY row
    (halo_recompute projects them, the exchange-free last hop sums the weight gradient over them) -- 29 entries per own row.  The partition
    cannot price boundary rows (they are a result of it), and with the hub-first order they pile up on the low ranks (rank 0: 4 k own rows,
    3.0 M boundary rows), so the weight that balances the measured per-rank times is lower: slowest / mean rank 1.27 at 12 (the round-2
    constant: the ranks with the many low-degree rows are slowest), 1.24 at 29 (the hub ranks are), ~1.1 at 18."""
    L = max(1, int(num_layers))
    return max(4, int(round(18.0 * (d / 128.0) * ((3 * L - 1) / 5.0) / ((4 * L - 2) / 6.0))))


SPLIT_MIN_HALO_ROWS = 65536   # split="auto": overlap a hop with its exchange from this many boundary rows on (32 MB at d = 128); below
                              # it the exchange is latency-bound and the second SpMM launch + two events cost more than they hide


def split_by_column(rowptr, col_local, val32, nl, device):
    """entries of a shard CSR (operand-row column ids) -> (rowptr, col, val) of those with col < nl and of the others; a row's
    entries keep their order"""
    counts = (rowptr[1:] - rowptr[:-1]).long()
    row_of = torch.repeat_interleave(torch.arange(nl, device=device), counts) if nl else torch.zeros(0, dtype=torch.long, device=device)
    own = col_local[:row_of.numel()].long() < nl
    out = []
    for mask in (own, ~own):
        rp = torch.zeros(nl + 1, dtype=torch.int64, device=device)
        if nl:
            rp[1:] = torch.cumsum(torch.bincount(row_of[mask], minlength=nl), 0)
        out.append((rp.to(torch.int32), col_local[:row_of.numel()][mask].contiguous(), val32[:row_of.numel()][mask].contiguous()))
    return out


def transpose_in_place(rowptr, col_local, val32, nl, n_cols, device):
    """the shard's A_hat ([nl] x [n_cols], operand-row column ids) transposed: (rowptr [n_cols + 1], col = own row ids, val), a row's
    entries in ascending own-row order"""
    nnz = int(col_local.numel()) if nl else 0
    counts = (rowptr[1:] - rowptr[:-1]).long()
    row_of = torch.repeat_interleave(torch.arange(nl, device=device), counts) if nl else torch.zeros(0, dtype=torch.long, device=device)
    c = col_local[:row_of.numel()].long()
    order = torch.argsort(c * max(nl, 1) + row_of)
    rp = torch.zeros(n_cols + 1, dtype=torch.int64, device=device)
    if row_of.numel():
        rp[1:] = torch.cumsum(torch.bincount(c, minlength=n_cols), 0)
    return rp.to(torch.int32), row_of[order].to(torch.int32).contiguous(), val32[:row_of.numel()][order].contiguous()


def build_shard(source, comm: Comm, need_transpose=True, device=None, relabel="auto", row_weight=ROW_WEIGHT, ops=None, split="auto",
                local_transpose="auto", allow_nan=False, name_of=None) -> Shard:
    """relabel: True / False / "auto" (hub-first node order when the graph has >= RELABEL_MIN_NODES nodes; the non-temporal
    treatment of the cold rows additionally needs an operand far beyond the caches, gss_csr_set_hot).  Relabelling is invisible in
    the results: a row's entries keep their original
    order, so every sum is taken in the same order, batches name original ids (gss_plan_desc.node_map) and
    GssEngine.gather_embeddings returns original order.  ops: the device side (NativeShardOps unless a test plugs its own).
    split: True / False / "auto" -- also keep every matrix split into its own-column and boundary-column entries, so that the plan
    overlaps each hop with its halo exchange (gss_shard_desc.a_own ...; results then differ from the single-GPU plan by rounding,
    not bit for bit).  "auto": when a hop fetches >= SPLIT_MIN_HALO_ROWS boundary rows on some rank (RMAT scale), never on one rank.
    local_transpose: True / False / "auto" (on unless GSS_LOCAL_TRANSPOSE=0) -- also keep A_hat's shard transposed in place, so that the
    plan's last backward hop needs no exchange of u (gss_shard_desc.a_loc_t).
    allow_nan / name_of: as GssGraph -- a row sum <= 0 anywhere in the graph raises NonPositiveRowSum on EVERY rank (the ranks
    exchange their counts) unless allow_nan is set."""
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    ops = ops or NativeShardOps()
    P, rank = comm.world, comm.rank
    import os
    if local_transpose == "auto":      # job-wide: every rank inherits the environment
        local_transpose = os.environ.get("GSS_LOCAL_TRANSPOSE", "1") != "0"
    if split == "auto" and os.environ.get("GSS_SPLIT") in ("0", "1"):
        split = os.environ["GSS_SPLIT"] == "1"      # job-wide override (every rank inherits the environment): entry points without a flag for it
    work = np.asarray(source.work(comm, dev), dtype=np.int64)
    n = len(work)
    if relabel == "auto":
        relabel = n >= RELABEL_MIN_NODES
    rl = Relabel(work) if relabel else None
    if rl is not None:
        work = work[rl.perm]
    # contiguous node ranges of equal cost: stored entries + a per-row share for the dense kernels
    part = Partition(nnz_balanced_ranges(np.concatenate([[0], np.cumsum(work + max(0, int(row_weight) - 1))]), P))
    lo, hi = part.rows(rank)
    nl = hi - lo

    rowptr, col, val = source.rows(lo, hi, dev, relabel=rl)
    dinv_local, rowsum = ops.rowsum_dinv(nl, rowptr, val, dev)
    # the guard helper.py:85 lacks: every rank learns how many rows of the WHOLE graph have D_ii <= 0 and which comes first
    bad, first = ops.rowsum_check(nl, rowsum)
    first_orig = -1
    if bad:
        first_orig = int(rl.perm[lo + first]) if rl is not None else lo + first
    every = allgather_host(comm, np.array([bad, first_orig], dtype=np.int64), dev).reshape(-1, 2) if P > 1 else np.array([[bad, first_orig]])
    if every[:, 0].sum() > 0:
        from .graph import report_bad_row_sums
        report_bad_row_sums(int(every[:, 0].sum()), int(every[every[:, 0] > 0, 1].min()), n, allow_nan, name_of)
    dinv = allgather_ranges(comm, dinv_local[:nl], part.bounds, dev).contiguous()            # D^-1/2 of every node

    def finish(rowptr, col, val, transposed):
        val32 = ops.scale_adj(nl, lo, rowptr, col, val, dinv, transposed, dev)
        uniq = torch.unique(col).cpu().numpy() if col.numel() else np.zeros(0, np.int64)
        halo = Halo(None, part, rank, uniq=uniq).exchange(comm, dev)
        g2o = torch.from_numpy(halo.gid2op).to(dev)
        col_local = g2o[col.long()].contiguous() if col.numel() else col
        csr = ops.csr(rowptr.cpu().numpy(), col_local, val32[:col.numel()], nl, nl + halo.n_halo, dev)
        hot = None
        if rl is not None:
            # the hubs are nodes [0, HOT_ROWS): this shard's own rows among them, and the head of its halo (ascending ids)
            hot = (int(min(max(HOT_ROWS - lo, 0), nl)), nl, nl + int(np.searchsorted(halo.remote, HOT_ROWS)))
            ops.set_hot(csr, *hot)
        want_split = split
        if split == "auto":
            # every rank takes the same decision: the largest halo of the job
            most = int(allgather_host(comm, np.array([halo.n_halo], dtype=np.int64), dev).max()) if P > 1 else 0
            want_split = most >= SPLIT_MIN_HALO_ROWS
        halves = None
        if want_split and P > 1:
            halves = []
            for rp, cl, vl in split_by_column(rowptr, col_local, val32, nl, dev):
                part_csr = ops.csr(rp.cpu().numpy(), cl, vl if vl.numel() else val32[:1], nl, nl + halo.n_halo, dev)
                if hot is not None:
                    ops.set_hot(part_csr, *hot)
                halves.append(part_csr)
        loc_t = None
        if not transposed and P > 1 and need_transpose and local_transpose:
            # A_hat's shard transposed in place, for the exchange-free last backward hop (gss_shard_desc.a_loc_t)
            rp, cl, vl = transpose_in_place(rowptr, col_local, val32, nl, nl + halo.n_halo, dev)
            loc_t = ops.csr(rp.cpu().numpy(), cl, vl if vl.numel() else val32[:1], nl + halo.n_halo, nl, dev)
            if hot is not None:
                ops.set_hot(loc_t, hot[0], nl, nl)          # its operand is u's OWN rows: the shard's own hubs
        return csr, halo, halves, loc_t

    a, halo_a, split_a, a_loc_t = finish(rowptr, col, val, 0)
    del rowptr, col, val
    at, halo_at, split_at = None, None, None
    if need_transpose:
        rowptr, col, val = source.rows_t(lo, hi, dev, relabel=rl)
        at, halo_at, split_at, _ = finish(rowptr, col, val, 1)
        del rowptr, col, val
    if hasattr(source, "release"):
        source.release()
    layout = ShardLayout(part, rank, halo_a, halo_at, dev, split_a=split_a, split_at=split_at, a_loc_t=a_loc_t)
    node_map = rl.inv_dev(dev).to(torch.int32).contiguous() if rl is not None else None
    return Shard(a, at, layout, part, int(source.nnz), rowsum[:nl], rl, node_map, split_a=split_a, split_at=split_at)


def shard_rows(shard: Shard, x_all):
    """this shard's feature rows out of a full [N][d] host matrix in ORIGINAL node order"""
    lo, hi = shard.part.rows(shard.layout.rank)
    return x_all[lo:hi] if shard.relabel is None else x_all[shard.relabel.perm[lo:hi]]


def shard_engine(shard: Shard, x_local, params_host, comm: Comm, num_layers=2, layer_decay=0.3, alpha=1.0, lr=1e-4, max_batch=None,
                 betas=(0.9, 0.999), eps=1e-8, cache_layer1=False):
    """GssEngine over a built shard (gss_plan_create_sharded): x_local = this shard's feature rows (numpy or device tensor;
    shard_rows() picks them out of a full matrix)"""
    from .engine import GssEngine
    dev = shard.a.rowptr.device
    x = x_local if torch.is_tensor(x_local) else torch.from_numpy(np.ascontiguousarray(x_local, dtype=np.float32))
    x = x.to(dev).contiguous()
    params = [torch.from_numpy(np.ascontiguousarray(params_host[k], dtype=np.float32)).to(dev) for k in ("W1", "b1", "W2", "b2")]
    n_global = int(shard.part.bounds[-1])
    eng = GssEngine(shard, x, params, num_layers=num_layers, layer_decay=layer_decay, alpha=alpha, lr=lr, max_batch=max_batch or n_global,
                    cache_layer1=cache_layer1, betas=betas, eps=eps, shard=shard.layout, comm=comm, node_map=shard.node_map)
    eng.global_nnz, eng.part, eng.layout = shard.nnz_global, shard.part, shard.layout
    return eng
