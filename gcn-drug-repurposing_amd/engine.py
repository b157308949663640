"""GssEngine: Python owner of one gss_plan (include/gssgcn.h) -- one training replica on one GPU.

torch tensors are only the containers: parameters, embeddings, loss and gradients live in torch
allocations whose device pointers are handed to the plan once; every arithmetic step is a HIP kernel
enqueued by libgssgcn.so on torch's current stream."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib

PARAM_NAMES = ("W1", "b1", "W2", "b2")


class GssEngine:
    def __init__(self, graph, x: torch.Tensor, params, num_layers=2, layer_decay=0.3, alpha=1.0, lr=1e-4,
                 max_batch=None, cache_layer1=False, betas=(0.9, 0.999), eps=1e-8, pipeline_layer1=False, shard=None, comm=None,
                 node_map=None):
        """shard = dist.ShardLayout + comm (dist.Comm): one shard of a node-range sharded replica
        (gss_plan_create_sharded); graph.a / graph.at then hold this shard's rows with operand-row column ids and
        x / emb this shard's rows.  Every method is then a collective over the shards."""
        assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 2
        n, d = x.shape
        assert n == graph.n, f"features have {n} rows, graph has {graph.n} nodes"
        if num_layers > 1 and graph.at is None:
            raise ValueError("num_layers >= 2 needs the transposed CSR (GssGraph(need_transpose=True))")
        self.graph, self.x = graph, x
        self.n, self.d, self.num_layers = n, d, int(num_layers)
        self.shard, self.comm = shard, comm
        # relabelled graphs (shards.build_shard(relabel=...)): node_map[original id] = row the node lives in (device int32 [N]);
        # batches keep naming original ids, gather_embeddings() returns original order
        self.node_map = node_map
        self.n_global = int(shard.bounds[-1]) if shard is not None else n
        self.max_batch = int(max_batch or self.n_global)
        self.params = list(params)
        shapes = [(d, d), (d,), (d, d), (d,)]
        for p, s in zip(self.params, shapes):
            assert p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and tuple(p.shape) == s, (p.shape, s)
        dev = x.device
        self.emb = torch.empty(n, d, dtype=torch.float32, device=dev)
        self.loss = torch.zeros(1, dtype=torch.float32, device=dev)
        self.grads = [torch.zeros(s, dtype=torch.float32, device=dev) for s in shapes]
        self.lib = _lib.load()
        self.desc = _lib.PlanDesc(n, d, self.num_layers, self.max_batch, float(layer_decay), float(alpha), float(lr),
                                  float(betas[0]), float(betas[1]), float(eps), 1 if cache_layer1 else 0,
                                  1 if (pipeline_layer1 and not cache_layer1) else 0, _lib.ptr(node_map))
        io = _lib.PlanIO(x.data_ptr(), *[p.data_ptr() for p in self.params], self.emb.data_ptr(), self.loss.data_ptr(),
                         *[g.data_ptr() for g in self.grads])
        self._param_ptrs = [p.data_ptr() for p in self.params]
        h = C.c_void_p()
        at_h = graph.at.handle if graph.at is not None else None
        if shard is None:
            _lib.check(self.lib.gss_plan_create(C.byref(h), C.byref(self.desc), graph.a.handle, at_h, C.byref(io)), "gss_plan_create")
        else:
            sd = shard.c_desc()           # borrows host arrays / device tensors owned by `shard`, which this engine keeps alive
            _lib.check(self.lib.gss_plan_create_sharded(C.byref(h), C.byref(self.desc), C.byref(sd), comm.handle if comm is not None else None,
                                                        graph.a.handle, at_h, C.byref(io)), "gss_plan_create_sharded")
        self.handle = h
        self._destroy = self.lib.gss_plan_destroy

    def __del__(self):
        h = getattr(self, "handle", None)
        if h is not None and h.value:
            self._destroy(h)
            self.handle = None

    def params_moved(self):
        return [p.data_ptr() for p in self.params] != self._param_ptrs

    # -- the four phases of train.py:158-184 -------------------------------------------------------
    def forward(self):
        _lib.check(self.lib.gss_plan_forward(self.handle, _lib.current_stream()), "gss_plan_forward")
        return self.emb

    def loss_backward(self, idx32: torch.Tensor, beta: float, count=None, offset=0):
        b = int(count if count is not None else idx32.numel())
        _lib.check(self.lib.gss_plan_loss_backward(self.handle, idx32.data_ptr() + 4 * offset, b, float(beta),
                                                   _lib.current_stream()), "gss_plan_loss_backward")

    def backward(self, rows32: torch.Tensor, de_rows: torch.Tensor):
        assert de_rows.is_contiguous() and de_rows.dtype == torch.float32
        _lib.check(self.lib.gss_plan_backward(self.handle, rows32.data_ptr(), int(rows32.numel()), de_rows.data_ptr(),
                                              _lib.current_stream()), "gss_plan_backward")

    def adam(self):
        _lib.check(self.lib.gss_plan_adam(self.handle, _lib.current_stream()), "gss_plan_adam")

    def step(self, idx32: torch.Tensor, beta: float, count=None, offset=0):
        b = int(count if count is not None else idx32.numel())
        _lib.check(self.lib.gss_plan_step(self.handle, idx32.data_ptr() + 4 * offset, b, float(beta), _lib.current_stream()),
                   "gss_plan_step")

    def step_lazy(self, idx32: torch.Tensor, beta: float, count=None, offset=0):
        """step() with the top layer evaluated on the batch rows only (gss_plan_step_lazy): the same loss, gradients and parameters
        bit for bit; self.emb is then valid on the batch rows only -- forward() recomputes all of it"""
        b = int(count if count is not None else idx32.numel())
        _lib.check(self.lib.gss_plan_step_lazy(self.handle, idx32.data_ptr() + 4 * offset, b, float(beta), _lib.current_stream()),
                   "gss_plan_step_lazy")

    def gather_embeddings(self) -> torch.Tensor:
        """the full [N][d] embeddings in node order (a collective on a sharded plan; the plan's own tensor otherwise)"""
        if self.shard is None:
            out = self.emb
        else:
            out = torch.empty(self.n_global, self.d, dtype=torch.float32, device=self.x.device)
            _lib.check(self.lib.gss_plan_gather_embeddings(self.handle, out.data_ptr(), _lib.current_stream()), "gss_plan_gather_embeddings")
        if self.node_map is not None:
            out = out.index_select(0, self.node_map.long())       # row of original node i = row node_map[i] of the relabelled graph
        return out

    def percentile(self, q: float) -> float:
        """beta = np.percentile(E E^T, q) of the current embeddings (train.py:165-167), exact, on device."""
        nm, self.node_map = self.node_map, None               # the percentile of E E^T does not depend on the row order
        try:
            e = self.gather_embeddings()
        finally:
            self.node_map = nm
        out = C.c_float()
        _lib.check(self.lib.gss_percentile(self.n_global, self.d, e.data_ptr(), float(q), C.byref(out), _lib.current_stream()),
                   "gss_percentile")
        return float(out.value)

    def activation(self, layer: int, which: str) -> torch.Tensor:
        """copy of AX / AM / P of a layer, or of the gradient buffers the last backward pass left: "dP" = the bottom layer's pre-activation
        gradient [n][d] (L >= 2), "dP_batch" = the top layer's on the batch rows, in batch order [max_batch][d]; "u" / "t" = the results of the top
        layer's first backward hop, "g_batch" = the batch rows' input gradients [g_ax ; g_am] ([2 max_batch][d]; the first b rows of each half are valid
        when b < max_batch: the halves then start at rows 0 and b) (parity tests)"""
        src = self.lib.gss_plan_activation(self.handle, layer, {"AX": 0, "AM": 1, "P": 2, "dP": 3, "dP_batch": 4, "u": 5, "t": 6, "g_batch": 8}[which])
        assert src, (layer, which)
        rows = {"dP_batch": self.max_batch, "g_batch": 2 * self.max_batch}.get(which, self.n)
        out = torch.empty(rows, self.d, dtype=torch.float32, device=self.x.device)
        _lib.check(self.lib.gss_memcpy_d2d(out.data_ptr(), src, out.numel() * 4, _lib.current_stream()), "gss_memcpy_d2d")
        return out

    def written_rows_bitmap(self):
        """bool [n]: the rows of u / t the top layer's first backward hop wrote in the last whole step (plans over huge operands skip the rows
        that are zero by contract), or None when the plan keeps no such bitmap (every row is written)"""
        src = self.lib.gss_plan_activation(self.handle, 0, 7)
        if not src:
            return None
        words = torch.empty((self.n + 31) // 32, dtype=torch.int32, device=self.x.device)
        _lib.check(self.lib.gss_memcpy_d2d(words.data_ptr(), src, words.numel() * 4, _lib.current_stream()), "gss_memcpy_d2d")
        bits = (words.view(-1, 1) >> torch.arange(32, device=words.device, dtype=torch.int32).view(1, -1)) & 1
        return bits.reshape(-1)[:self.n].bool()

    def profile(self, enable=True):
        _lib.check(self.lib.gss_plan_profile(self.handle, 1 if enable else 0), "gss_plan_profile")

    def profile_read(self):
        """-> {class: (total_ms, launches)} since the last read (synchronises the stream)"""
        k = len(_lib.PROF_CLASSES)
        ms = (C.c_double * k)()
        cnt = (C.c_int64 * k)()
        _lib.check(self.lib.gss_plan_profile_read(self.handle, ms, cnt, _lib.current_stream()), "gss_plan_profile_read")
        return {name: (ms[i], cnt[i]) for i, name in enumerate(_lib.PROF_CLASSES)}

    # -- checkpoint / resume (the reference has none: SURVEY section 5; a few hundred KB per replica) ---------------
    def _adam_view(self, moment: int, k: int) -> torch.Tensor:
        ptr = self.lib.gss_plan_adam_buffer(self.handle, moment, k)
        assert ptr, (moment, k)
        out = torch.empty_like(self.params[k])
        _lib.check(self.lib.gss_memcpy_d2d(out.data_ptr(), ptr, out.numel() * 4, _lib.current_stream()), "gss_memcpy_d2d")
        return out

    def state_dict(self) -> dict:
        """parameters, Adam moments and step count as host arrays"""
        sd = {"step": np.int64(self.lib.gss_plan_get_step(self.handle))}
        for k, name in enumerate(PARAM_NAMES):
            sd[name] = self.params[k].detach().cpu().numpy()
            sd["m_" + name] = self._adam_view(0, k).cpu().numpy()
            sd["v_" + name] = self._adam_view(1, k).cpu().numpy()
        return sd

    def load_state_dict(self, sd) -> None:
        dev = self.x.device
        for k, name in enumerate(PARAM_NAMES):
            self.params[k].copy_(torch.from_numpy(np.ascontiguousarray(sd[name], dtype=np.float32)).to(dev))
            for moment, key in ((0, "m_" + name), (1, "v_" + name)):
                src = torch.from_numpy(np.ascontiguousarray(sd[key], dtype=np.float32)).to(dev)
                assert src.shape == self.params[k].shape, (key, src.shape)
                _lib.check(self.lib.gss_memcpy_d2d(self.lib.gss_plan_adam_buffer(self.handle, moment, k), src.data_ptr(), src.numel() * 4,
                                                   _lib.current_stream()), "gss_memcpy_d2d")
                torch.cuda.current_stream().synchronize()   # src must outlive the copy
        self.lib.gss_plan_set_step(self.handle, int(sd["step"]))

    def check_guards(self) -> None:
        """raises if a kernel wrote behind one of the plan's buffers (gss_plan_check_guards; tests call it)"""
        _lib.check(self.lib.gss_plan_check_guards(self.handle), "gss_plan_check_guards")

    def lazy_halo_rows(self):
        """what the last subset exchanges of a sharded plan moved (gss_plan_lazy_halo_rows): (fetched, sent, whole halo) of the top
        layer's M in a lazy step, then the same of u in the second backward hop; fetched / sent are -1 when the plan exchanges whole halos"""
        import ctypes as C
        out = (C.c_int64 * 6)()
        _lib.check(self.lib.gss_plan_lazy_halo_rows(self.handle, out), "gss_plan_lazy_halo_rows")
        return tuple(int(v) for v in out)

    def comm_stats(self):
        """collectives the plan has enqueued since the last call (gss_plan_comm_stats): (boundary-row exchanges, batch-row all-reduces,
        weight-gradient all-reduces); zeros on one GPU"""
        import ctypes as C
        out = (C.c_int64 * 3)()
        _lib.check(self.lib.gss_plan_comm_stats(self.handle, out), "gss_plan_comm_stats")
        return tuple(int(v) for v in out)

    def sync_stats(self):
        """host-side waits of a sharded plan since the last call (gss_plan_sync_stats): (drains of the caller's stream, waits for an event
        of the plan's request stream); zeros on one GPU"""
        import ctypes as C
        out = (C.c_int64 * 2)()
        _lib.check(self.lib.gss_plan_sync_stats(self.handle, out), "gss_plan_sync_stats")
        return tuple(int(v) for v in out)

    def device_bytes(self) -> int:
        return int(self.lib.gss_plan_device_bytes(self.handle))
