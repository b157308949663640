"""CPU oracle for the GSS-GCN hot path -- TEST INFRASTRUCTURE ONLY.

This file is a plain numpy/scipy restatement of the reference algorithm
(bowang-lab/gcn-drug-repurposing).  It exists to *check* the HIP path and to
define the CPU baseline; nothing in the product package may import it.  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
use it.

Parity status: PINNED.  ``tests/golden/*.npz`` were produced by importing the
reference's own ``modules/model.py`` / ``helpers/helper.py`` / ``train.py`` in
the build container (``tests/golden/make_fixtures.py``) and
``tests/test_oracle_golden.py`` checks every function below against them.
The reference itself ships no test or golden vector for this path
(SURVEY.md section 4), so those generated fixtures are the pin.

Every function cites the reference file:line it restates (paths relative to
the reference root).  All maths is written for an arbitrary float dtype so the
same code runs as the fp64 "truth" and as an fp32 model of torch's arithmetic.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp

# --------------------------------------------------------------------------
# graph preparation
# --------------------------------------------------------------------------


def gen_graph_descriptor(X, k=5):
    """kNN ("descriptor") adjacency.  helpers/helper.py:39-53.

    X is (d, N) fp64 exactly as train.py:81 passes it.  Row i keeps its top-k
    inner products (argpartition, self normally among them), the matrix is
    symmetrised by also writing the transposed entry, the diagonal is zeroed.
    Values are the raw inner products.
    """
    x_sim = np.matmul(X.T, X)
    top = np.argpartition(x_sim, -k, 1)[:, -k:]
    n = x_sim.shape[0]
    rows = np.repeat(np.arange(n), k)
    cols = top.reshape(-1)
    vals = x_sim[rows, cols]
    keep = rows != cols
    rows, cols, vals = rows[keep], cols[keep], vals[keep]
    # x_adj[i, top] = v ; x_adj[top, i] = v  is an assignment inside a loop over i
    # (helper.py:48-51): when both i and j select the pair, the write of the later
    # iteration survives.  Keep, per (row, col), the entry of the largest iteration.
    r = np.concatenate([rows, cols])
    c = np.concatenate([cols, rows])
    v = np.concatenate([vals, vals])
    it = np.concatenate([rows, rows])
    key = r.astype(np.int64) * n + c
    order = np.lexsort((it, key))
    key_s = key[order]
    last = order[np.r_[key_s[1:] != key_s[:-1], True]]
    adj = sp.csr_matrix((v[last], (r[last], c[last])), shape=(n, n))
    adj.eliminate_zeros()  # sp.csr_matrix(dense) drops exact zeros, helper.py:53
    adj.sort_indices()
    return adj


def edgelist_to_adj(src, dst, w, n):
    """Directed weighted edgelist -> CSR (u v w per line, predict_drug.py:224-226).

    Duplicate (u, v) pairs keep the last weight, as networkx DiGraph.add_edge does.
    """
    src = np.asarray(src, dtype=np.int64)
    dst = np.asarray(dst, dtype=np.int64)
    w = np.asarray(w, dtype=np.float64)
    key = src * n + dst
    # last occurrence wins
    _, last_rev = np.unique(key[::-1], return_index=True)
    last = len(key) - 1 - last_rev
    adj = sp.csr_matrix((w[last], (src[last], dst[last])), shape=(n, n))
    adj.sort_indices()
    return adj


def preprocess_graph(adj):
    """A_hat = D^-1/2 (A + I) D^-1/2, D = row sums of A + I.  helpers/helper.py:82-89.

    fp64 throughout, like the reference; returns (csr fp64, rowsum fp64[N]).
    """
    adj_ = sp.csr_matrix(adj, dtype=np.float64) + sp.eye(adj.shape[0], dtype=np.float64, format="csr")
    rowsum = np.asarray(adj_.sum(1)).reshape(-1)
    dinv = np.power(rowsum, -0.5)
    d = sp.diags(dinv)
    out = sp.csr_matrix(adj_.dot(d).transpose().dot(d).transpose())
    out.sort_indices()
    return out, rowsum


def to_fp32_csr(adj_hat):
    """The fp32 cast of helpers/helper.py:92-96 (kept CSR instead of COO)."""
    a = sp.csr_matrix(adj_hat)
    a.sort_indices()
    return sp.csr_matrix((a.data.astype(np.float32), a.indices.copy(), a.indptr.copy()), shape=a.shape)


# --------------------------------------------------------------------------
# model
# --------------------------------------------------------------------------


def init_layer_weights(d, init_weights, rng=np.random):
    """W1 = W2 = randn(d,d)*eps with unit diagonal, b1 = b2 = 0.  modules/model.py:137-150."""
    w = rng.randn(d, d) * init_weights
    w[np.where(np.eye(d) != 0)] = 1
    w = w.astype(np.float32)
    z = np.zeros(d, dtype=np.float32)
    return {"W1": w.copy(), "b1": z.copy(), "W2": w.copy(), "b2": z.copy()}


def _elu(p):
    return np.where(p > 0, p, np.expm1(np.minimum(p, 0)))


def forward(X, A, params, num_layers, layer_decay, dtype=np.float64, keep=True):
    """ResidualGraphConvolutionalNetwork.forward.  modules/model.py:163-175,197-207.

    Returns (emb, cache).  cache holds per-layer x_in, AX, AM, P plus the
    pre-normalisation x and its row norms for the backward pass.
    """
    A = sp.csr_matrix(A, dtype=dtype)
    x = np.asarray(X, dtype=dtype)
    W1 = params["W1"].astype(dtype)
    W2 = params["W2"].astype(dtype)
    b1 = params["b1"].astype(dtype)
    b2 = params["b2"].astype(dtype)
    layers = []
    residual = None
    for _ in range(num_layers):
        ax = A @ x                                   # model.py:163
        am = A @ (ax * x)                            # model.py:168-169
        p = (ax @ W1.T + b1) + (am @ W2.T + b2)      # model.py:165,170,172
        o = _elu(p).astype(dtype)                    # model.py:173
        layers.append({"x": x, "AX": ax, "AM": am, "P": p})
        x = o if residual is None else residual + dtype(layer_decay) * o  # model.py:201-202
        residual = p                                  # model.py:203
    nrm = np.sqrt((x * x).sum(1))
    den = np.maximum(nrm, dtype(1e-12))              # F.normalize eps, model.py:205
    emb = x / den[:, None]
    cache = {"layers": layers, "x_out": x, "den": den, "A": A, "emb": emb,
             "params": (W1, b1, W2, b2), "decay": dtype(layer_decay)} if keep else None
    return emb, cache


def gss_loss(emb, beta, index=None, alpha=1.0):
    """mean(-alpha/2 * (relu(E_B E_B^T) - beta)^2).  modules/model.py:214-221."""
    e = emb if index is None else emb[np.asarray(index)]
    s = e @ e.T
    logits = np.maximum(s, 0)
    return (-0.5 * alpha * (logits - beta) ** 2).mean(dtype=np.float64)


def loss_grad_emb(emb, beta, index, alpha=1.0):
    """dL/dE (N x d, only batch rows non-zero): autograd of model.py:216-221."""
    idx = np.arange(emb.shape[0]) if index is None else np.asarray(index)
    e = emb[idx]
    s = e @ e.T
    b = len(idx)
    g = -(alpha / (b * b)) * (np.maximum(s, 0) - beta) * (s > 0)
    de_b = g @ e + g.T @ e
    de = np.zeros_like(emb)
    np.add.at(de, idx, de_b)
    return de


def backward(cache, d_emb):
    """Manual reverse pass through forward(); what torch autograd does at train.py:183."""
    W1, b1, W2, b2 = cache["params"]
    A = cache["A"]
    AT = sp.csr_matrix(A.T)
    decay = cache["decay"]
    emb, den = cache["emb"], cache["den"]
    # F.normalize backward (rows with norm > eps)
    dx = (d_emb - emb * (emb * d_emb).sum(1, keepdims=True)) / den[:, None]
    layers = cache["layers"]
    L = len(layers)
    gW1 = np.zeros_like(W1)
    gW2 = np.zeros_like(W2)
    gb = np.zeros_like(b1)
    d_res = None  # gradient flowing into P_{l} through the residual of layer l+1
    for l in range(L - 1, -1, -1):
        lay = layers[l]
        p = lay["P"]
        if l == 0:
            d_o = dx                      # x_1 = O_1
            d_p = d_o * np.where(p > 0, 1.0, np.exp(np.minimum(p, 0)))
        else:
            d_o = decay * dx              # x_l = P_{l-1} + decay * O_l
            d_p = d_o * np.where(p > 0, 1.0, np.exp(np.minimum(p, 0)))
        if d_res is not None:
            d_p = d_p + d_res
        d_res = dx if l > 0 else None     # into P_{l-1}
        gW1 += d_p.T @ lay["AX"]
        gW2 += d_p.T @ lay["AM"]
        gb += d_p.sum(0)
        if l > 0:
            d_ax = d_p @ W1
            d_am = d_p @ W2
            d_m = AT @ d_am
            d_ax = d_ax + d_m * lay["x"]
            dx = d_m * lay["AX"] + AT @ d_ax
    return {"W1": gW1, "b1": gb.copy(), "W2": gW2, "b2": gb.copy()}


def adam_step(params, grads, state, lr, betas=(0.9, 0.999), eps=1e-8):
    """torch.optim.Adam (no weight decay, no amsgrad), train.py:139-141,184.  In place, fp32."""
    state["t"] = state.get("t", 0) + 1
    t = state["t"]
    b1, b2 = betas
    bc1 = 1.0 - b1 ** t
    bc2 = 1.0 - b2 ** t
    for k in ("W1", "b1", "W2", "b2"):
        g = grads[k].astype(np.float32)
        m = state.setdefault("m_" + k, np.zeros_like(params[k]))
        v = state.setdefault("v_" + k, np.zeros_like(params[k]))
        m += (g - m) * np.float32(1 - b1)          # exp_avg.lerp_(grad, 1-beta1)
        v *= np.float32(b2)
        v += np.float32(1 - b2) * g * g            # addcmul_
        denom = np.sqrt(v) / np.float32(np.sqrt(bc2)) + np.float32(eps)
        params[k] -= np.float32(lr / bc1) * (m / denom)
    return params


def beta_percentile(emb, q):
    """np.percentile(E E^T flattened, q), train.py:165-167 (linear interpolation)."""
    e = np.asarray(emb, dtype=np.float32)
    return np.percentile(np.dot(e, e.T).flatten(), q)


def epoch_batches(n, batch_size, rng):
    """One epoch of the index sampler (method/dataset.py:5-28): a fresh permutation
    cut into ceil(n/B) batches, last one short; batch_size 0 means one full batch.
    The reference draws it from torch's global RNG; parity tests replay recorded lists."""
    b = n if batch_size <= 0 else batch_size
    perm = rng.permutation(n)
    return [perm[i:i + b] for i in range(0, n, b)]


def train(X, A_hat32, params, num_layers, layer_decay, batches, beta=None,
          beta_pct=None, alpha=1.0, lr=1e-4, dtype=np.float32):
    """The train.py:151-193 loop over a recorded list of batches (all epochs flattened).

    Returns (emb_of_last_forward, params, losses, beta).  The returned embeddings come
    from the last forward, i.e. before the final optimizer step (train.py:158,193)."""
    state = {}
    losses = []
    emb = None
    for it, idx in enumerate(batches):
        emb, cache = forward(X, A_hat32, params, num_layers, layer_decay, dtype=dtype)
        if it == 0 and beta is None:
            beta = beta_percentile(emb, beta_pct)
        losses.append(gss_loss(emb, beta, idx, alpha))
        grads = backward(cache, loss_grad_emb(emb, dtype(beta), idx, alpha))
        adam_step(params, grads, state, lr)
    return emb, params, losses, beta


# --------------------------------------------------------------------------
# file formats
# --------------------------------------------------------------------------


def read_embs(path):
    """'.embs.txt' reader: header 'N d', then 'name v1..vd' (openne/node2vec.py:40-47; train.py:79-80)."""
    names, rows = [], []
    with open(path) as f:
        f.readline()
        for line in f:
            parts = line.split()
            if not parts:
                continue
            names.append(parts[0])
            rows.append([float(x) for x in parts[1:]])
    return names, np.asarray(rows, dtype=np.float64)


def write_graph_embs(path, emb):
    """np.savetxt default format, train.py:193."""
    np.savetxt(path, np.asarray(emb))


# --------------------------------------------------------------------------
# embedding consumer contract
# --------------------------------------------------------------------------


def rank_scores(emb, query_row, candidate_rows):
    """predict_drug.py:51-70: row-normalise, score = e_cand . e_query, descending order."""
    e = np.asarray(emb, dtype=np.float64)
    e = e / np.maximum(np.sqrt((e * e).sum(1, keepdims=True)), 1e-12)  # sklearn normalize(axis=1)
    s = e[np.asarray(candidate_rows)] @ e[query_row]
    order = np.argsort(s)[::-1]
    return s, order


def roc_auc(labels, scores):
    """ROC-AUC by the rank-sum formula with average ranks for ties (sklearn.roc_auc_score, evaluate_auc.py:156-170)."""
    labels = np.asarray(labels).astype(bool)
    scores = np.asarray(scores, dtype=np.float64)
    from scipy.stats import rankdata
    r = rankdata(scores)
    n_pos = labels.sum()
    n_neg = len(labels) - n_pos
    if n_pos == 0 or n_neg == 0:
        return float("nan")
    return (r[labels].sum() - n_pos * (n_pos + 1) / 2.0) / (n_pos * n_neg)
