"""Drop-in mirrors of the reference's operator interface (modules/model.py:134-221):

    model = ResidualGraphConvolutionalNetwork(train_batch_size, val_batch_size, num_layers, hidden_units,
                                              init_weights, layer_decay)
    emb   = model(x=features, adj=adj)                  # [N, d], unit-norm rows
    loss  = GSS_loss(alpha).gss_loss(emb, beta, index)  # scalar
    loss.backward(); torch.optim.Adam(model.parameters()).step()

Same constructor, parameter names (gcn_layer.dense{,2}.{weight,bias}) and semantics; the arithmetic is
the HIP path of libgssgcn.so behind torch.autograd.Function glue.  `adj` may be a GssGraph or the
torch sparse COO tensor of A_hat the reference passes (helpers/helper.py:92-96).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import scipy.sparse as sp
import torch
import torch.nn as nn

from . import _lib
from .engine import GssEngine
from .graph import DeviceCSR, GssGraph


class GSS_GNNLayer(nn.Module):
    """Parameter container of the two-hop layer (modules/model.py:137-150): W1 = W2 = eps * randn with
    unit diagonal (numpy RNG), zero biases.  nn.Linear is constructed first so the torch RNG stream is
    consumed exactly as in the reference (the batch sampler draws from it afterwards)."""

    def __init__(self, hidden_units, init_weights=1e-5):
        super().__init__()
        w = np.random.randn(hidden_units, hidden_units) * init_weights
        np.fill_diagonal(w, 1.0)
        w = torch.tensor(w, dtype=torch.float32)
        self.dense = nn.Linear(hidden_units, hidden_units)
        self.dense2 = nn.Linear(hidden_units, hidden_units)
        for lin in (self.dense, self.dense2):
            lin.weight = nn.Parameter(w.clone())
            lin.bias = nn.Parameter(torch.zeros(hidden_units, dtype=torch.float32))


def _graph_from_sparse_tensor(adj: torch.Tensor) -> GssGraph:
    """torch sparse COO tensor holding A_hat -> GssGraph without re-normalising."""
    a = adj.coalesce() if not adj.is_coalesced() else adj
    ij = a.indices().cpu().numpy()
    m = sp.csr_matrix((a.values().cpu().numpy().astype(np.float32), (ij[0], ij[1])), shape=tuple(a.shape))
    m.sort_indices()
    return GssGraph.from_normalized(m)


class _Forward(torch.autograd.Function):
    @staticmethod
    def forward(ctx, engine, w1, b1, w2, b2):
        ctx.engine = engine
        return engine.forward().clone()

    @staticmethod
    def backward(ctx, d_emb):
        eng = ctx.engine
        rows = (d_emb != 0).any(dim=1).nonzero().flatten()
        if rows.numel() == 0:
            return (None,) + tuple(torch.zeros_like(g) for g in eng.grads)
        eng.backward(rows.to(torch.int32), d_emb.index_select(0, rows).contiguous())
        return (None,) + tuple(g.clone() for g in eng.grads)


class ResidualGraphConvolutionalNetwork(nn.Module):
    def __init__(self, train_batch_size, val_batch_size, num_layers=2, hidden_units=2048, init_weights=1e-5,
                 layer_decay=0.4):
        super().__init__()
        self.train_batch_size = train_batch_size
        self.val_batch_size = val_batch_size
        self.num_layers = num_layers
        self.hidden_units = hidden_units
        self.init_weights = init_weights
        self.layer_decay = layer_decay
        self.gcn_layer = GSS_GNNLayer(hidden_units, init_weights)
        self._engine = None
        self._engine_key = None
        self._graphs = {}

    def _params(self):
        g = self.gcn_layer
        return [g.dense.weight, g.dense.bias, g.dense2.weight, g.dense2.bias]

    def engine_for(self, x, adj) -> GssEngine:
        if not torch.cuda.is_available():
            raise _lib.GssError("ResidualGraphConvolutionalNetwork needs an MI355X: there is no CPU path")
        if not isinstance(adj, GssGraph):
            key = id(adj)
            if key not in self._graphs:
                self._graphs[key] = (_graph_from_sparse_tensor(adj), adj)  # keep adj alive so id() stays unique
            adj = self._graphs[key][0]
        params = [p.data for p in self._params()]
        if not all(p.is_cuda for p in params):
            raise _lib.GssError("move the model to the GPU first (model.cuda()), as train.py:118-121 does with --gpu-id")
        if not x.is_cuda:
            x = x.cuda()
        x = x.contiguous().float()
        key = (x.data_ptr(), tuple(x.shape), id(adj), self.num_layers, float(self.layer_decay))
        if self._engine is None or self._engine_key != key or self._engine.params_moved():
            self._engine = GssEngine(adj, x, params, num_layers=self.num_layers, layer_decay=self.layer_decay,
                                     max_batch=x.shape[0])
            self._engine_key = key
        return self._engine

    def forward(self, x, adj):
        eng = self.engine_for(x, adj)
        return _Forward.apply(eng, *self._params())


class _Loss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, embs, index, beta, alpha):
        lib = _lib.load()
        n, d = embs.shape
        e = embs.contiguous()
        idx32 = index.to(device=e.device, dtype=torch.int32).contiguous()
        b = idx32.numel()
        loss = torch.empty(1, dtype=torch.float32, device=e.device)
        de_b = torch.empty(b, d, dtype=torch.float32, device=e.device)
        ws = torch.empty(lib.gss_loss_workspace_bytes(b, d), dtype=torch.uint8, device=e.device)
        _lib.check(lib.gss_loss_fwd_bwd(n, d, e.data_ptr(), idx32.data_ptr(), b, float(beta), float(alpha), loss.data_ptr(),
                                        de_b.data_ptr(), ws.data_ptr(), _lib.current_stream()), "gss_loss_fwd_bwd")
        ctx.save_for_backward(idx32, de_b)
        ctx.shape = (n, d)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        idx32, de_b = ctx.saved_tensors
        d_emb = torch.zeros(ctx.shape, dtype=torch.float32, device=de_b.device)
        d_emb.index_copy_(0, idx32.long(), de_b * g)
        return d_emb, None, None, None


class GSS_loss:
    def __init__(self, alpha):
        self.alpha = alpha

    def gss_loss(self, embs, beta, index=None):
        """If index is None the loss runs over all rows (modules/model.py:214-221)."""
        if index is None:
            index = torch.arange(embs.shape[0], device=embs.device)
        return _Loss.apply(embs, index, float(beta), float(self.alpha))
