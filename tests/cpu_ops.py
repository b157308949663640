"""numpy stand-ins for the device side of the sharded trainer -- TEST INFRASTRUCTURE.
Lets the PRODUCT's shard preparation (gcn_drug_repurposing_amd/shards.py build_shard, dist.Halo / ShardLayout) and a restatement of
the C++ plan's sharded step (tests/shard_step_mirror.py) run as separate gloo processes on CPU, where there is no GPU for the HIP
kernels:
  NumpyOps       the arithmetic of the per-op kernels (fp64 inside, fp32 between ops)
  NumpyShardOps  what shards.NativeShardOps does with libgssgcn.so (row sums, D^-1/2 scaling with the kernels' rounding sequence, CSR)
  GlooComm       the five methods of dist.Comm over torch.distributed (gloo)"""
import numpy as np
import scipy.sparse as sp
import torch


class _Csr:
    def __init__(self, indptr, indices, data, n_rows, n_cols):
        self.n_rows, self.n_cols = n_rows, n_cols
        self.m = sp.csr_matrix((np.asarray(data, np.float64), np.asarray(indices), np.asarray(indptr)), shape=(n_rows, n_cols))


def _np(t):
    return t.detach().numpy().astype(np.float64)


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def _elu_grad(p):
    return np.where(p > 0, 1.0, np.exp(np.minimum(p, 0)))


class NumpyOps:
    device = torch.device("cpu")

    def empty(self, *shape, dtype=torch.float32):
        return torch.empty(*shape, dtype=dtype)

    def zeros(self, *shape, dtype=torch.float32):
        return torch.zeros(*shape, dtype=dtype)

    def tensor(self, a, dtype=None):
        t = torch.from_numpy(np.ascontiguousarray(a))
        return t.to(dtype) if dtype is not None else t

    def csr(self, indptr, indices, data, n_rows, n_cols):
        return _Csr(indptr, indices, data, n_rows, n_cols)

    def spmm(self, csr, x_full, h=None):
        y = csr.m @ _np(x_full)
        return _t(y), (_t(y * _np(h)) if h is not None else None)

    def dense_fwd(self, ax, am, w, p_prev, decay):
        p = _np(ax) @ _np(w[0]).T + _np(w[1]) + _np(am) @ _np(w[2]).T + _np(w[3])
        o = np.where(p > 0, p, np.expm1(np.minimum(p, 0)))
        xn = o if p_prev is None else _np(p_prev) + decay * o
        return _t(p), _t(xn)

    def rownorm_fwd(self, x):
        x = _np(x)
        inv = 1.0 / np.maximum(np.sqrt((x * x).sum(1)), 1e-12)
        return _t(x * inv[:, None]), _t(inv)

    def loss_fwd_bwd(self, e_b, beta, alpha):
        e = _np(e_b)
        b = e.shape[0]
        s = e @ e.T
        loss = (-0.5 * alpha * (np.maximum(s, 0) - beta) ** 2).mean()
        g = -(alpha / (b * b)) * (np.maximum(s, 0) - beta) * (s > 0)
        return _t(np.array([loss])), _t(g @ e + g.T @ e)

    def rownorm_elu_bwd(self, de_rows, rows, e, inv_den, p, c):
        r = rows.numpy().astype(np.int64)
        de, eb, inv, pb = _np(de_rows), _np(e)[r], _np(inv_den)[r], _np(p)[r]
        dx = (de - eb * (eb * de).sum(1, keepdims=True)) * inv[:, None]
        return _t(dx), _t(c * dx * _elu_grad(pb))

    def wgrad(self, dp, ax, am, rows, grads, accumulate):
        dpn = _np(dp)
        r = rows.numpy().astype(np.int64) if rows is not None else slice(None)
        new = [dpn.T @ _np(ax)[r], dpn.sum(0), dpn.T @ _np(am)[r], dpn.sum(0)]
        for g, v in zip(grads, new):
            g.copy_(_t(v + (_np(g) if accumulate else 0.0)))

    def dgrad(self, dp, w1t, w2t):
        return _t(_np(dp) @ _np(w1t).T), _t(_np(dp) @ _np(w2t).T)

    def spmm_bwd1(self, at, gam_full, gax_local, x_in, ax):
        dm = at.m @ _np(gam_full)
        return _t(_np(gax_local) + dm * _np(x_in)), _t(dm * _np(ax))

    def spmm_bwd1_sparse(self, at, gam_b, gax_b, pos_col, pos_row, x_in, ax):
        pc, pr = pos_col.numpy(), pos_row.numpy()
        gam_full = np.zeros((at.n_cols, gam_b.shape[1]))
        gam_full[pc >= 0] = _np(gam_b)[pc[pc >= 0]]
        gax_local = np.zeros((at.n_rows, gax_b.shape[1]))
        sel = pr[:at.n_rows] >= 0
        gax_local[sel] = _np(gax_b)[pr[:at.n_rows][sel]]
        dm = at.m @ gam_full
        return _t(gax_local + dm * _np(x_in)), _t(dm * _np(ax))

    def spmm_bwd2(self, at, u_full, t, p, c, res, want_gx):
        gx = _np(t) + at.m @ _np(u_full)
        dp = c * gx * _elu_grad(_np(p)) + (_np(res) if res is not None else 0.0)
        return _t(dp), (_t(gx) if want_gx else None)

    def adam(self, params, grads, m, v, step, lr, betas, eps):
        b1, b2 = betas
        bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
        for k in range(4):
            m[k].lerp_(grads[k], 1 - b1)
            v[k].mul_(b2).addcmul_(grads[k], grads[k], value=1 - b2)
            params[k].addcdiv_(m[k], (v[k].sqrt() / bc2 ** 0.5).add_(eps), value=-lr / bc1)


class NumpyShardOps:
    """shards.NativeShardOps on the host: the same results as csrc/spmm.hip rowsum_kernel / scale_shard_kernel (same rounding
    sequence: fp64 products in the kernels' order, one cast to fp32)"""

    def rowsum_dinv(self, nl, rowptr, val, dev):
        rp, v = rowptr.numpy().astype(np.int64), val.numpy()
        rowsum = np.array([v[rp[i]:rp[i + 1]].sum() for i in range(nl)] + ([0.0] if nl == 0 else []), dtype=np.float64)
        with np.errstate(divide="ignore", invalid="ignore"):
            dinv = np.power(rowsum, -0.5)
        return torch.from_numpy(dinv), torch.from_numpy(rowsum)

    def rowsum_check(self, nl, rowsum):
        bad = np.flatnonzero(~(_np(rowsum)[:nl] > 0))
        return len(bad), (int(bad[0]) if len(bad) else -1)

    def scale_adj(self, nl, lo, rowptr, col, val, dinv, transposed, dev):
        rp, c, v, di = rowptr.numpy().astype(np.int64), col.numpy().astype(np.int64), val.numpy(), dinv.numpy()
        row = np.repeat(np.arange(nl, dtype=np.int64), np.diff(rp[:nl + 1])) + lo
        out = ((v * di[row]) * di[c]) if transposed else ((v * di[c]) * di[row])
        out = out.astype(np.float32)
        return torch.from_numpy(out if out.size else np.zeros(1, np.float32))

    def csr(self, rowptr_host, col_local, val32, n_rows, n_cols, dev):
        c = _Csr(rowptr_host, col_local.numpy(), val32.numpy(), n_rows, n_cols)
        c.nnz = int(len(val32))
        c.hot = None
        return c

    def set_hot(self, csr, own_hot, halo_begin, halo_end):
        csr.hot = (own_hot, halo_begin, halo_end)       # speed only on the device; recorded so that a test can look at it


class GlooComm:
    """dist.Comm's methods over torch.distributed (gloo): CPU tensors, blocking"""

    handle = None

    def __init__(self):
        import torch.distributed as dist
        self.dist = dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.rows_received = 0          # rows of all exchange_rows calls (what a step moves over the links)
        self.exchanges = 0
        self.allreduces = 0

    def abort(self):
        pass

    def check(self):
        pass

    def count(self):
        return self.world

    def sync(self, timeout_s=None):
        pass

    def allgather_bytes(self, src):
        src = src.contiguous()
        parts = [torch.empty_like(src) for _ in range(self.world)]
        self.dist.all_gather(parts, src)
        return torch.cat([p.reshape(-1) for p in parts])

    def exchange_rows(self, d, send, send_off, recv, recv_off):
        send_off, recv_off = np.asarray(send_off, np.int64), np.asarray(recv_off, np.int64)
        reqs, landed = [], []
        for q in range(self.world):
            if q == self.rank:
                continue
            ns, nr = int(send_off[q + 1] - send_off[q]), int(recv_off[q + 1] - recv_off[q])
            if ns > 0:
                reqs.append(self.dist.isend(send.reshape(-1)[send_off[q] * d:send_off[q + 1] * d].contiguous(), q))
            if nr > 0:
                buf = torch.empty(nr * d, dtype=recv.dtype)
                reqs.append(self.dist.irecv(buf, q))
                landed.append((int(recv_off[q]), nr, buf))
        for r in reqs:
            r.wait()
        flat = recv.reshape(-1)
        for off, nr, buf in landed:
            flat[off * d:(off + nr) * d] = buf
            self.rows_received += nr
        self.exchanges += 1

    def all_reduce_sum_(self, t):
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        self.allreduces += 1


def emulate_ppr(prob, alpha, tol, max_iter):
    """numpy statement of what gss_ppr_run does on the device (csrc/ppr.hip), step for step, from a PprProblem:
    used on CPU to check the problem's index lists against the oracle.  -> (x [n][k], iterations [k])"""
    n, k = prob.n, prob.k
    x = np.full((n, k), 1.0 / n)
    done = np.zeros(k, dtype=bool)
    iters = np.zeros(k, dtype=np.int32)
    cols = np.arange(k)
    for it in range(1, max_iter + 1):
        held = x[prob.z_rows].copy()                                   # ppr_dangling_kernel
        held[prob.z_rows[:, None] == prob.starts[None, :]] = 0.0
        dsum = held.sum(0)
        xs = x.copy()                                                  # ppr_ovr_scale_kernel
        orig = xs[prob.ovr_row, prob.ovr_col].copy()
        xs[prob.ovr_row, prob.ovr_col] = orig * prob.ovr_ratio
        yself = np.zeros(k)
        for c in range(k):                                             # ppr_column_kernel
            for e in prob.zero_ovr[prob.zero_ptr[c]:prob.zero_ptr[c + 1]]:
                dsum[c] += orig[e]
            if prob.start_dangling[c]:
                dsum[c] += xs[prob.starts[c], c]
            lo, hi = prob.keep_ptr[c], prob.keep_ptr[c + 1]
            yself[c] = (prob.keep_val[lo:hi] * xs[prob.keep_row[lo:hi], c]).sum()
        y = prob.mt @ xs                                               # ppr_spmm_kernel
        np.add.at(y, (prob.sel_row, prob.sel_col), prob.sel_val * x[prob.starts[prob.sel_col], prob.sel_col])   # ppr_sel_kernel
        y[prob.starts, cols] = yself
        p = np.zeros((n, k))
        p[prob.starts, cols] = 1.0
        xn = alpha * (y + dsum[None, :] * p) + (1.0 - alpha) * p       # ppr_update_kernel
        err = np.abs(xn - x).sum(0)
        upd = ~done
        x[:, upd] = xn[:, upd]
        newly = upd & (err < n * tol)                                  # ppr_finish_kernel
        iters[newly] = it
        done |= newly
        if done.all():
            return x, iters
    raise RuntimeError("not converged")
