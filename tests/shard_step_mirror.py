"""A numpy restatement of the SHARDED training step that csrc/plan.hip enqueues (plan_forward_impl, plan_loss_backward_impl,
plan_backward_impl, plan_step_impl with P > 1) -- TEST INFRASTRUCTURE.

It runs on the objects the PRODUCT prepares -- shards.build_shard's Shard (CSRs with operand-row column ids, node_map of a
relabelled graph), dist.ShardLayout / dist.Halo (recv_off, send_off, send_rows, gid2op) -- and talks to its peers through the same
`Comm` methods, so tests/test_dist_cpu.py can run the product's partitioning, halo layout and boundary-row exchange offsets as
separate gloo processes and compare the trajectories with the reference-generated fixtures.  What it shares with the C++ plan is
the data layout and the order of the exchange points, line by line:

  operand = [n own rows | n_halo boundary rows grouped by owner]         plan.hip header, carve()
  plan_halo: pack rows send_rows[0..n_send) -> exchange_rows(send_off, recv_off) -> operand[n:]      plan.hip plan_halo
  X_0's and M_0's boundary rows fetched once (constants)                  plan_x0, m0_ready
  halo_recompute (on by default): AX_0 / AM_0 boundary rows fetched once, layer 1's projection over own + boundary
      rows, so layer 2's boundary input rows are computed here instead of exchanged                      plan_forward_impl (recompute)
  ONE batch collective: [E_B | P_B | inv_B] = sum over shards of (own rows | 0)                           C3, plan_loss_backward_impl
  batch maps: rloc / keep / pid = gid2op_t[node_map[idx]]                 loss.hip gather_batch_kernel
  finish + input gradient of EVERY member on every rank (no second collective), sparse first hop         loss.hip tail, plan_backward_impl
  halo of u (A_hat^T's halo) before every second hop -- except the LAST one (the bottom layer's), which runs in scatter-by-owner form over
      A_hat's shard transposed in place (gss_shard_desc.a_loc_t): no exchange                           plan_backward_impl (use_tloc)
  four weight gradients summed over the shards, then Adam                 plan_step_impl (P > 1 branch)

The arithmetic is NumpyOps' (fp64 inside an op, fp32 between ops)."""
import numpy as np
import torch

from cpu_ops import NumpyOps, _elu_grad


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class ShardStepMirror:
    def __init__(self, shard, x_local, params_host, comm, num_layers=2, layer_decay=0.3, alpha=1.0, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, slab=False):
        lay = shard.layout
        self.shard, self.lay, self.comm = shard, lay, comm
        self.P, self.rank = comm.world, comm.rank
        self.lo, self.hi = shard.part.rows(self.rank)
        self.n = self.hi - self.lo
        self.n_global = int(shard.part.bounds[-1])
        self.L, self.decay, self.alpha, self.lr, self.betas, self.eps = int(num_layers), float(layer_decay), float(alpha), float(lr), betas, float(eps)
        self.a = shard.a.m                                            # scipy [n][n + n_halo_a], operand-row column ids
        self.at = shard.at.m if shard.at is not None else None
        self.ha, self.ht = lay.halo_a, lay.halo_at
        assert self.a.shape == (self.n, self.n + self.ha.n_halo)
        assert self.at is None or self.at.shape == (self.n, self.n + self.ht.n_halo)
        self.node_map = shard.node_map.numpy().astype(np.int64) if shard.node_map is not None else None
        self.gid2op_t = lay.gid2op_t.numpy().astype(np.int64) if (lay.gid2op_t is not None and self.P > 1) else None
        self.x = _f32(x_local)
        assert self.x.shape[0] == self.n
        self.d = self.x.shape[1]
        self.params = [torch.from_numpy(_f32(params_host[k]).copy()) for k in ("W1", "b1", "W2", "b2")]
        self.grads = [torch.zeros_like(p) for p in self.params]
        self.m = [torch.zeros_like(p) for p in self.params]
        self.v = [torch.zeros_like(p) for p in self.params]
        self.step_no = 0
        self.ops = NumpyOps()
        self.x0op = None
        self.m0op = None
        self.recompute = self.P > 1 and self.L > 1      # knob halo_recompute, automatic choice (on)
        self.slab = slab and self.P > 1                 # knob loss_slab (automatic: batches of >= 8192 rows)
        # gss_shard_desc.a_loc_t: A_hat's shard transposed in place -- two layers + halo_recompute: the last backward hop without an exchange
        t_loc = getattr(lay, "a_loc_t", None)
        self.tloc = t_loc.m if (t_loc is not None and self.recompute and self.L >= 2) else None
        self.ax0op = self.am0op = None
        self.loss = None
        self.emb = None

    # -- plan_halo ----------------------------------------------------------------------------------------------
    def _halo(self, h, op):
        """op [n + n_halo][d] fp32 with its own rows complete: fetch the boundary rows, serve the peers"""
        if self.P == 1:
            return
        d = op.shape[1]
        n_send = int(h.send_off[-1])
        rows = h.send_rows[:n_send].numpy().astype(np.int64)
        assert n_send == 0 or (rows.min() >= 0 and rows.max() < self.n)
        send = torch.from_numpy(_f32(op[rows])) if n_send else torch.zeros(1, d)          # pack_rows_kernel
        recv = torch.zeros(max(h.n_halo, 1), d)
        self.comm.exchange_rows(d, send, h.send_off, recv, h.recv_off)
        op[self.n:] = recv[:h.n_halo].numpy()

    def _operand(self, h, own):
        op = np.zeros((self.n + h.n_halo, self.d), dtype=np.float32)
        op[:self.n] = own
        return op

    def _spmm(self, mat, op):
        return _f32(mat @ op.astype(np.float64))

    # -- plan_forward_impl ----------------------------------------------------------------------------------------
    def forward(self):
        L, n = self.L, self.n
        w1, b1, w2, b2 = self.params
        self.act = []
        p_prev = None
        x_own = self.x
        for l in range(L):
            if l == 0:
                if self.x0op is None:                                  # plan_x0: constants, exchanged once
                    self.x0op = self._operand(self.ha, self.x)
                    self._halo(self.ha, self.x0op)
                xl = self.x0op
            elif l == 1 and self.recompute:
                xl = x_op                                              # boundary rows computed by layer 1's projection below
            else:
                xl = self._operand(self.ha, x_own)
                self._halo(self.ha, xl)
            ax = self._spmm(self.a, xl)                                # model.py:163
            m_own = _f32(ax.astype(np.float64) * xl[:n])               # model.py:168 (fused epilogue)
            if l == 0:
                if self.m0op is None:                                  # boundary rows of M_0: constants, exchanged once
                    self.m0op = self._operand(self.ha, m_own)
                    self._halo(self.ha, self.m0op)
                self.m0op[:n] = m_own                                   # own rows are recomputed every step
                m = self.m0op
            else:
                m = self._operand(self.ha, m_own)
                self._halo(self.ha, m)
            am = self._spmm(self.a, m)                                  # model.py:169
            if l == 0 and self.recompute:
                # AX_0 / AM_0 are constants: boundary rows fetched once; the projection runs over own + boundary rows
                if self.ax0op is None:
                    self.ax0op, self.am0op = self._operand(self.ha, ax), self._operand(self.ha, am)
                    self._halo(self.ha, self.ax0op)
                    self._halo(self.ha, self.am0op)
                self.ax0op[:n], self.am0op[:n] = ax, am
                p_op, x_op = self.ops.dense_fwd(torch.from_numpy(self.ax0op), torch.from_numpy(self.am0op), self.params, None, self.decay)
                x_op = x_op.numpy()
                p, xn = p_op[:n], torch.from_numpy(x_op[:n])
            else:
                p, xn = self.ops.dense_fwd(torch.from_numpy(ax), torch.from_numpy(am), self.params, p_prev, self.decay)
            self.act.append({"xin": xl, "ax": ax, "am": am, "p": p.numpy(),
                             "p_op": (p_op.numpy() if (l == 0 and self.recompute) else None)})
            x_own, p_prev = xn.numpy(), p
        e, inv = self.ops.rownorm_fwd(torch.from_numpy(x_own))         # model.py:205
        self.emb, self.inv_den = e.numpy(), inv.numpy()
        return self.emb

    # -- batch maps: loss.hip gather_rows_mapped_kernel / elementwise.hip batch_prepare_kernel ----------------------
    def _batch_maps(self, idx):
        ids = np.asarray(idx, dtype=np.int64)
        if self.node_map is not None:
            ids = self.node_map[ids]
        rel = ids - self.lo
        mine = (rel >= 0) & (rel < self.n)
        rloc = np.clip(rel, 0, max(self.n - 1, 0))
        pid = self.gid2op_t[ids] if self.gid2op_t is not None else np.where(mine, rel, -1)
        return rloc, mine, pid

    # -- plan_loss_backward_impl + plan_backward_impl (sparse top layer) -----------------------------------------------
    def loss_backward(self, idx, beta):
        L, n, d = self.L, self.n, self.d
        b = len(idx)
        rloc, mine, pid = self._batch_maps(idx)
        keep = mine.astype(np.float64)[:, None]
        top = self.act[L - 1]
        # gather_batch_kernel: [E_B | P_B | inv_B] of the members this shard owns, zeros elsewhere; ONE all-reduce (C3) -- one non-zero
        # contributor per element -- leaves every rank with the whole batch's rows
        bx = np.zeros((b, 2 * d + 1), dtype=np.float32)
        if n > 0:
            bx[mine, :d] = self.emb[rloc[mine]]
            bx[mine, d:2 * d] = top["p"][rloc[mine]]
            bx[mine, 2 * d] = self.inv_den[rloc[mine]]
        bx = torch.from_numpy(bx)
        self.comm.all_reduce_sum_(bx)
        e_b = bx[:, :d].contiguous()
        loss, de_b = self.ops.loss_fwd_bwd(e_b, beta, self.alpha)      # model.py:218-221 (+ autograd); the same on every rank
        if self.slab:
            # row-slab sweep (knob loss_slab): rank r keeps the rows of the i tiles r, r + P, ... of dE and 1 / P-th ... its own tiles'
            # share of the loss; one more all-reduce assembles them (loss.hip loss_slab_sum_kernel, plan_loss_backward_impl)
            tiles = np.arange(b) // 16
            mine_t = (tiles % self.P) == self.rank
            dex = torch.zeros(b * d + 1)
            dex[:b * d] = (de_b * torch.from_numpy(mine_t.astype(np.float32))[:, None]).reshape(-1)
            # the loss is a sum over (i, j) pairs: this rank's share = the rows i of its tiles
            s_ = e_b.double() @ e_b.double().t()
            t_ = torch.clamp(s_, min=0.0) - beta
            dex[b * d] = float((-0.5 * self.alpha * (t_ * t_)[torch.from_numpy(mine_t)].sum() / (b * b)))
            self.comm.all_reduce_sum_(dex)
            de_b = dex[:b * d].reshape(b, d).contiguous()
            loss = dex[b * d].clone()
        self.loss = loss
        c_top = self.decay if L > 1 else 1.0
        # the sweep's tail: backward of F.normalize and F.elu on EVERY member (this rank holds p / inv_den of all of them); the rows
        # another shard owns are zero in dx_b / dp_b (the weight gradient and the residual count each member once, at its owner) but
        # take part in the input gradient below
        de, eb = de_b.numpy().astype(np.float64), e_b.numpy().astype(np.float64)
        inv_all = bx[:, 2 * d].numpy().astype(np.float64)[:, None]
        dx_all = _f32((de - eb * (eb * de).sum(1, keepdims=True)) * inv_all)
        dp_all = _f32(c_top * dx_all * _elu_grad(bx[:, d:2 * d].numpy().astype(np.float64)))
        dx_b = _f32(dx_all * keep)
        dp_b = _f32(dp_all * keep)
        w1, b1, w2, b2 = [p.numpy().astype(np.float64) for p in self.params]
        # weight gradients: fixed order top layer (batch rows) first, then the layers below; summed over the shards at the end
        gw1 = np.zeros((d, d))
        gw2 = np.zeros((d, d))
        gb = np.zeros(d)

        def wgrad(dp, ax, am):
            nonlocal gw1, gw2, gb
            dp = dp.astype(np.float64)
            gw1 += dp.T @ ax.astype(np.float64)
            gw2 += dp.T @ am.astype(np.float64)
            gb += dp.sum(0)

        if n > 0:
            wgrad(dp_b, top["ax"][rloc], top["am"][rloc])
        if L > 1:
            # compact input gradients of ALL batch rows, computed locally on every rank (the loss kernel's tail): no collective
            gax_b, gam_b = _f32(dp_all.astype(np.float64) @ w1), _f32(dp_all.astype(np.float64) @ w2)
            # batch-position map over A_hat^T's operand rows (own rows first, then its halo)
            rows_t = n + (self.ht.n_halo if self.P > 1 else 0)
            pos = np.full(max(rows_t, 1), -1, dtype=np.int64)
            pos[pid[pid >= 0]] = np.arange(b)[pid >= 0]
            g_am_op = np.zeros((rows_t, d), dtype=np.float32)
            g_am_op[pos[:rows_t] >= 0] = gam_b[pos[:rows_t][pos[:rows_t] >= 0]]
            g_ax = np.zeros((n, d), dtype=np.float32)
            sel = pos[:n] >= 0
            g_ax[sel] = gax_b[pos[:n][sel]]
            dm = self.at @ g_am_op.astype(np.float64)                  # spmm_bwd1_sparse: only batch-row neighbours contribute
            u_own = _f32(g_ax + dm * top["xin"][:n])
            t = _f32(dm * top["ax"])
            gx = [None, None]
            for lp in range(L - 2, -1, -1):
                lay = self.act[lp]
                c = 1.0 if lp == 0 else self.decay
                if self.tloc is not None and lp == 0:
                    # plan_backward_impl, use_tloc: this shard's rows of u multiplied into every row they touch (own + boundary), ELU'(P_0)
                    # applied there, the weight gradient summed over own + boundary rows -- no exchange of u; the ranks' all-reduce of the
                    # weight gradients completes the sum
                    rows_a = n + self.ha.n_halo
                    g_ext = self.tloc @ u_own.astype(np.float64)
                    g_ext[:n] += t.astype(np.float64)
                    dp_ext = c * g_ext * _elu_grad(lay["p_op"].astype(np.float64))
                    if lp + 2 <= L - 1:                                # deeper nets: the residual gradient of layer lp + 2 lives on own rows
                        dp_ext[:n] += gx[(lp + 2) & 1]
                    if lp + 2 == L and n > 0:
                        np.add.at(dp_ext, rloc[mine], dx_b[mine].astype(np.float64))
                    assert dp_ext.shape[0] == rows_a
                    wgrad(_f32(dp_ext), self.ax0op, self.am0op)
                    continue
                u = self._operand(self.ht, u_own)
                self._halo(self.ht, u)                                 # C1 on A_hat^T's halo
                g = t.astype(np.float64) + self.at @ u.astype(np.float64)
                dp = c * g * _elu_grad(lay["p"].astype(np.float64))
                if lp + 2 <= L - 1:
                    dp = dp + gx[(lp + 2) & 1]
                if lp + 2 == L and n > 0:                              # the top layer's residual: dP += dx_b on the batch rows this shard owns
                    np.add.at(dp, rloc[mine], dx_b[mine].astype(np.float64))
                dp = _f32(dp)
                if lp >= 1 and L > 2:
                    gx[(lp + 1) & 1] = _f32(g)
                wgrad(dp, lay["ax"], lay["am"])
                if lp >= 1:
                    g_ax_f = _f32(dp.astype(np.float64) @ w1)
                    g_am = self._operand(self.ht, _f32(dp.astype(np.float64) @ w2))
                    self._halo(self.ht, g_am)                          # C1
                    dm = self.at @ g_am.astype(np.float64)
                    u_own = _f32(g_ax_f + dm * lay["xin"][:n])
                    t = _f32(dm * lay["ax"])
        for g, v in zip(self.grads, (gw1, gb, gw2, gb)):
            g.copy_(torch.from_numpy(_f32(v)))
        flat = torch.cat([g.reshape(-1) for g in self.grads])
        self.comm.all_reduce_sum_(flat)                                # C2
        o = 0
        for g in self.grads:
            g.copy_(flat[o:o + g.numel()].view_as(g))
            o += g.numel()

    def adam(self):
        self.step_no += 1
        self.ops.adam(self.params, self.grads, self.m, self.v, self.step_no, self.lr, self.betas, self.eps)

    def step(self, idx, beta):
        self.forward()
        self.loss_backward(idx, beta)
        self.adam()

    # -- gss_plan_gather_embeddings + GssEngine.gather_embeddings ----------------------------------------------------
    def gather_embeddings(self):
        maxr = max(1, int(np.diff(self.shard.part.bounds).max()))
        pad = torch.zeros(maxr * self.d)
        pad[:self.n * self.d] = torch.from_numpy(self.emb).reshape(-1)
        allp = self.comm.allgather_bytes(pad).reshape(self.P, maxr, self.d)
        out = torch.cat([allp[r, :int(self.shard.part.bounds[r + 1] - self.shard.part.bounds[r])] for r in range(self.P)]).numpy()
        if self.node_map is not None:
            out = out[self.node_map]
        return out
