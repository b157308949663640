"""The reference's CPU path restated as the same sequence of torch ops -- TEST / BASELINE INFRASTRUCTURE ONLY.

modules/model.py:163-173,197-207 (torch.sparse.mm on COO fp32, nn.Linear, F.elu, F.normalize),
modules/model.py:216-221 (loss), train.py:139-141,182-184 (Adam, backward, step).  Used by bench.py's
cpu_baseline leg (kind "port": the reference's .py files cannot travel to the GPU box) and by tests as a
second opinion with autograd.  Pinned against the golden fixtures in tests/test_oracle_golden.py.
"""
import time

import numpy as np
import torch
import torch.nn.functional as F


class TorchCpuPath:
    def __init__(self, a_hat_csr32, x32, params, num_layers, layer_decay, alpha, lr, dtype=torch.float32):
        """dtype: torch.float32 = the reference's arithmetic; torch.float64 runs the same graph of ops in double precision (the
        fp32 inputs widened) -- tests/test_trajectory_spread.py uses it to measure how far two valid executions drift apart"""
        coo = a_hat_csr32.tocoo()
        i = torch.from_numpy(np.vstack([coo.row, coo.col]).astype(np.int64))
        self.adj = torch.sparse_coo_tensor(i, torch.from_numpy(coo.data.astype(np.float32)).to(dtype), coo.shape)  # helper.py:92-96
        self.x = torch.from_numpy(np.ascontiguousarray(x32, dtype=np.float32)).to(dtype)
        self.p = {k: torch.tensor(np.asarray(v, dtype=np.float32), dtype=dtype, requires_grad=True) for k, v in params.items()}
        self.L, self.decay, self.alpha = num_layers, layer_decay, alpha
        self.opt = torch.optim.Adam([self.p[k] for k in ("W1", "b1", "W2", "b2")], lr=lr, weight_decay=0)
        self.nnz = coo.nnz

    def forward(self):
        x, residual = self.x, None
        for _ in range(self.L):
            ax = torch.sparse.mm(self.adj, x)
            p1 = F.linear(ax, self.p["W1"], self.p["b1"])
            am = torch.sparse.mm(self.adj, torch.mul(ax, x))
            p2 = F.linear(am, self.p["W2"], self.p["b2"])
            pre = p1 + p2
            out = F.elu(pre)
            x = out if residual is None else residual + self.decay * out
            residual = pre
        return F.normalize(x, dim=1)

    def loss(self, emb, beta, idx):
        e = emb[torch.as_tensor(idx, dtype=torch.long)]
        s = F.relu(torch.mm(e, e.transpose(0, 1)))
        return torch.mean(-0.5 * self.alpha * (s - beta) ** 2)

    def step(self, idx, beta):
        emb = self.forward()
        loss = self.loss(emb, beta, idx)
        self.opt.zero_grad()
        loss.backward()
        self.opt.step()
        return emb.detach(), float(loss.detach())

    def time_steps(self, batches, beta, warmup=1):
        for b in batches[:warmup]:
            self.step(b, beta)
        ts = []
        for b in batches[warmup:]:
            t0 = time.perf_counter()
            self.step(b, beta)
            ts.append(time.perf_counter() - t0)
        return ts

    def time_spmm(self, reps=5):
        with torch.no_grad():
            torch.sparse.mm(self.adj, self.x)
            t0 = time.perf_counter()
            for _ in range(reps):
                torch.sparse.mm(self.adj, self.x)
            return (time.perf_counter() - t0) / reps
