"""Graph preparation: adjacency (kNN 'descriptor' mode or weighted edgelist) -> on-device CSR of
A_hat = D^-1/2 (A + I) D^-1/2 and of A_hat^T.

Reference: helpers/helper.py:25-58 (gen_graph), :82-89 (preprocess_graph), :92-96 (sparse tensor).
Only index bookkeeping (sorting, inserting the diagonal, transposing the structure) runs on the host;
row sums and the normalisation run in gss_normalize_adj on the GPU (fp64 accumulate, fp32 result).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import scipy.sparse as sp
import torch

from . import _lib


class NonPositiveRowSum(ValueError):
    """A + I has rows whose sum D_ii is not > 0: D^-1/2 is NaN / inf there (helpers/helper.py:85 does not guard)."""

    def __init__(self, count, first, n):
        self.count, self.first, self.n = int(count), int(first), int(n)
        super().__init__(f"{self.count} of {self.n} rows of A + I have a sum <= 0 (first: node {self.first}): D^-1/2 is NaN/inf there and the NaN "
                         f"spreads through every hop.  Negative similarities in kNN mode do this; the reference trains on and writes NaN "
                         f"embeddings without a word.  Pass allow_nan / --allow-nan for that behaviour.")


def report_bad_row_sums(count, first, n, allow_nan, name_of=None):
    """What a positive count from gss_rowsum_check leads to: an exception, or under allow_nan a warning and the reference's NaNs."""
    if count <= 0:
        return
    err = NonPositiveRowSum(count, first, n)
    if name_of is not None:
        err.args = (err.args[0].replace(f"node {first})", f"node {first} = {name_of(first)!r})"),)
    if not allow_nan:
        raise err
    import warnings
    warnings.warn(str(err.args[0]), RuntimeWarning, stacklevel=3)


def rowsum_check(rowsum, n):
    """gss_rowsum_check on a device vector of fp64 row sums: (rows whose D_ii is not > 0, the first of them or -1)"""
    count, first = C.c_int64(0), C.c_int32(-1)
    _lib.check(_lib.load().gss_rowsum_check(int(n), rowsum.data_ptr() if n else None, C.addressof(count), C.addressof(first),
                                            _lib.current_stream()), "gss_rowsum_check")
    return int(count.value), int(first.value)


def knn_descriptor_adj(X, k=5, chunk=4096):
    """helpers/helper.py:39-53 without the two dense N x N host arrays: x_sim rows are produced in
    chunks.  X: [N, d] fp64 features (the reference passes the transpose).  Returns scipy CSR fp64.
    (Host-side setup step; the device builder is SURVEY section 8-f3.)"""
    X = np.ascontiguousarray(X, dtype=np.float64)
    n = X.shape[0]
    k = min(k, n)
    rows, cols, vals = [], [], []
    for s in range(0, n, chunk):
        sim = X[s:s + chunk] @ X.T
        top = np.argpartition(sim, -k, 1)[:, -k:]
        r = np.repeat(np.arange(s, min(n, s + chunk)), k)
        c = top.reshape(-1)
        v = sim[r - s, c]
        keep = r != c                      # x_adj[i, i] = 0  (helper.py:51)
        rows.append(r[keep]); cols.append(c[keep]); vals.append(v[keep])
    rows = np.concatenate(rows); cols = np.concatenate(cols); vals = np.concatenate(vals)
    # x_adj[i, top] = v and x_adj[top, i] = v are assignments in a loop over i (helper.py:48-50):
    # the write of the later iteration wins
    r = np.concatenate([rows, cols]); c = np.concatenate([cols, rows])
    v = np.concatenate([vals, vals]); it = np.concatenate([rows, rows])
    key = r.astype(np.int64) * n + c
    order = np.lexsort((it, key))
    ks = key[order]
    last = order[np.r_[ks[1:] != ks[:-1], True]] if len(ks) else order
    adj = sp.csr_matrix((v[last], (r[last], c[last])), shape=(n, n))
    adj.eliminate_zeros()
    adj.sort_indices()
    return adj


def _symmetrise_topk(top_idx, top_val, n):
    """helper.py:46-53 on (row, top-k) pairs: x_adj[i, top] = v, x_adj[top, i] = v in a loop over i (the later
    iteration wins), zero diagonal, exact zeros dropped"""
    k = top_idx.shape[1]
    rows = np.repeat(np.arange(n), k)
    cols = top_idx.reshape(-1).astype(np.int64)
    vals = top_val.reshape(-1)
    keep = (rows != cols) & (cols >= 0)
    rows, cols, vals = rows[keep], cols[keep], vals[keep]
    r = np.concatenate([rows, cols]); c = np.concatenate([cols, rows])
    v = np.concatenate([vals, vals]); it = np.concatenate([rows, rows])
    key = r.astype(np.int64) * n + c
    order = np.lexsort((it, key))
    ks = key[order]
    last = order[np.r_[ks[1:] != ks[:-1], True]] if len(ks) else order
    adj = sp.csr_matrix((v[last], (r[last], c[last])), shape=(n, n))
    adj.eliminate_zeros()
    adj.sort_indices()
    return adj


def knn_descriptor_adj_device(X, k=5, device="cuda"):
    """gen_graph's kNN adjacency with the N x N similarity and the top-k selection on the GPU (gss_knn_topk, fp64
    MFMA).  X: [N, d] fp64 (d is zero-padded to a multiple of 8).  Same result as knn_descriptor_adj up to the
    order of fp64 summation."""
    X = np.ascontiguousarray(X, dtype=np.float64)
    n, d = X.shape
    k = min(k, n)
    if d % 8:
        X = np.concatenate([X, np.zeros((n, 8 - d % 8))], axis=1)
    lib = _lib.load()
    xd = torch.from_numpy(X).to(device)
    tv = torch.empty(n, k, dtype=torch.float64, device=device)
    ti = torch.empty(n, k, dtype=torch.int32, device=device)
    _lib.check(lib.gss_knn_topk(n, X.shape[1], xd.data_ptr(), k, tv.data_ptr(), ti.data_ptr(), _lib.current_stream()), "gss_knn_topk")
    return _symmetrise_topk(ti.cpu().numpy(), tv.cpu().numpy(), n)


def edgelist_adj(src, dst, w, n):
    """directed weighted edgelist -> CSR; a repeated (u, v) keeps the last weight (DiGraph.add_edge)."""
    key = np.asarray(src, np.int64) * n + np.asarray(dst, np.int64)
    order = np.argsort(key, kind="stable")
    ks = key[order]
    last = order[np.r_[ks[1:] != ks[:-1], True]] if len(ks) else order
    adj = sp.csr_matrix((np.asarray(w, np.float64)[last], (np.asarray(src)[last], np.asarray(dst)[last])), shape=(n, n))
    adj.sort_indices()
    return adj


class DeviceCSR:
    """CSR operand on the GPU + its libgssgcn handle (row bins for the SpMM kernels)."""

    def __init__(self, indptr, indices, values32, n_rows, n_cols, device):
        self.n_rows, self.n_cols = int(n_rows), int(n_cols)
        self.h_indptr = np.ascontiguousarray(indptr, dtype=np.int32)
        self.nnz = int(self.h_indptr[-1])
        self.rowptr = torch.from_numpy(self.h_indptr).to(device)
        self.col = (indices.to(device=device, dtype=torch.int32).contiguous() if torch.is_tensor(indices)
                    else torch.from_numpy(np.ascontiguousarray(indices, dtype=np.int32)).to(device))
        if self.col.numel() == 0:
            self.col = torch.zeros(1, dtype=torch.int32, device=device)     # a non-null pointer for an empty shard
        self.val = values32 if torch.is_tensor(values32) else torch.from_numpy(np.ascontiguousarray(values32, np.float32)).to(device)
        assert self.val.dtype == torch.float32 and self.val.numel() >= self.nnz
        lib = _lib.load()
        h = C.c_void_p()
        _lib.check(lib.gss_csr_create(C.byref(h), self.n_rows, self.n_cols, self.nnz, self.h_indptr.ctypes.data,
                                      self.rowptr.data_ptr(), self.col.data_ptr(), self.val.data_ptr()), "gss_csr_create")
        self.handle = h
        self._destroy = lib.gss_csr_destroy   # bound now: module globals may be gone at interpreter shutdown

    def __del__(self):
        h = getattr(self, "handle", None)
        if h is not None and h.value:
            self._destroy(h)
            self.handle = None

    def to_scipy(self):
        return sp.csr_matrix((self.val.cpu().numpy(), self.col.cpu().numpy(), self.h_indptr), shape=(self.n_rows, self.n_cols))


class GssGraph:
    """A_hat and A_hat^T of one graph on one GPU.

    `adj` is the raw scipy adjacency (fp64, any sparsity format).  Mirrors
    preprocess_graph + convert_sparse_matrix_to_sparse_tensor (helpers/helper.py:82-96)."""

    def __init__(self, adj, device="cuda", need_transpose=True, allow_nan=False, name_of=None):
        """allow_nan: a row sum <= 0 raises NonPositiveRowSum unless set (then: a warning, and NaN / inf exactly where the reference
        has them).  name_of: node id -> name, for the message."""
        adj = sp.csr_matrix(adj, dtype=np.float64)
        n = adj.shape[0]
        assert adj.shape[0] == adj.shape[1]
        a_ = (adj + sp.eye(n, dtype=np.float64, format="csr")).tocsr()   # structure of A + I (helper.py:83)
        a_.sort_indices()
        self.n = n
        lib = _lib.load()
        dev = torch.device(device)
        rowptr = torch.from_numpy(a_.indptr.astype(np.int32)).to(dev)
        col = torch.from_numpy(a_.indices.astype(np.int32)).to(dev)
        val64 = torch.from_numpy(a_.data.astype(np.float64)).to(dev)
        val32 = torch.empty(a_.nnz, dtype=torch.float32, device=dev)
        self.rowsum = torch.empty(n, dtype=torch.float64, device=dev)
        _lib.check(lib.gss_normalize_adj(n, rowptr.data_ptr(), col.data_ptr(), val64.data_ptr(), val32.data_ptr(),
                                         self.rowsum.data_ptr(), _lib.current_stream()), "gss_normalize_adj")
        self.bad_rows, self.first_bad_row = rowsum_check(self.rowsum, n)
        report_bad_row_sums(self.bad_rows, self.first_bad_row, n, allow_nan, name_of)
        self.a = DeviceCSR(a_.indptr, a_.indices, val32, n, n, dev)
        self.at = None
        if need_transpose:
            # structure of (A + I)^T on the host; values are a permutation of A_hat's
            tag = sp.csr_matrix((np.arange(1, a_.nnz + 1, dtype=np.int64), a_.indices, a_.indptr), shape=(n, n)).T.tocsr()
            tag.sort_indices()
            perm = torch.from_numpy((tag.data - 1).astype(np.int64)).to(dev)
            self.at = DeviceCSR(tag.indptr, tag.indices, val32.index_select(0, perm), n, n, dev)

    @classmethod
    def from_normalized(cls, a_hat, device="cuda", need_transpose=True):
        """Wrap an already normalised A_hat (scipy CSR, e.g. the reference's preprocess_graph output)."""
        self = cls.__new__(cls)
        a_hat = sp.csr_matrix(a_hat)
        a_hat.sort_indices()
        n = a_hat.shape[0]
        dev = torch.device(device)
        self.n = n
        self.rowsum = None
        self.bad_rows, self.first_bad_row = 0, -1
        self.a = DeviceCSR(a_hat.indptr, a_hat.indices, a_hat.data.astype(np.float32), n, n, dev)
        self.at = None
        if need_transpose:
            t = sp.csr_matrix(a_hat.T)
            t.sort_indices()
            self.at = DeviceCSR(t.indptr, t.indices, t.data.astype(np.float32), n, n, dev)
        return self

    @property
    def nnz(self):
        return self.a.nnz
