#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by RUNNING THE REFERENCE.

Run once in the build container (the reference lives read-only at /root/reference
and does not travel to the GPU box):

    python tests/golden/make_fixtures.py

It imports the reference's own modules/model.py, helpers/helper.py and runs its
train.py (with the one-line ``np.float = float`` shim numpy>=1.24 needs,
train.py:80) on small seeded inputs and stores inputs + outputs as .npz.  Only
data is stored -- no reference source, bytecode or pickled reference objects.
"""
import os
import runpy
import sys
import tempfile

import numpy as np

REF = os.environ.get("GSS_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
np.float = float  # noqa: shim for train.py:80 / predict_drug.py on numpy >= 1.24

import scipy.sparse as sp  # noqa: E402
import torch  # noqa: E402

import helpers.helper as H  # noqa: E402  (reference)
import modules.model as M  # noqa: E402  (reference)

torch.set_num_threads(4)


def csr_parts(a, prefix):
    a = sp.csr_matrix(a)
    a.sort_indices()
    return {prefix + "_indptr": a.indptr.astype(np.int64), prefix + "_indices": a.indices.astype(np.int64),
            prefix + "_data": a.data.astype(np.float64)}


def run_case(name, adj, X32, L, decay, init_w, alpha, beta_pct, batches, lr, seed, keep_layers):
    """adj: scipy CSR raw adjacency (fp64); X32: (N,d) fp32 features; batches: list of index arrays."""
    n, d = X32.shape
    out = {"meta": np.array([n, d, L], dtype=np.int64), "decay": np.float64(decay), "alpha": np.float64(alpha),
           "lr": np.float64(lr), "init_weights": np.float64(init_w), "seed": np.int64(seed),
           "beta_pct": np.float64(beta_pct)}
    out.update(csr_parts(adj, "A"))
    a_hat = H.preprocess_graph(adj)                                   # helper.py:82-89
    out.update(csr_parts(a_hat, "Ahat"))
    out["rowsum"] = np.asarray((sp.csr_matrix(adj) + sp.eye(n)).sum(1)).reshape(-1)
    adj_t = H.convert_sparse_matrix_to_sparse_tensor(a_hat)            # helper.py:92-96
    out["X"] = X32

    torch.manual_seed(seed)
    np.random.seed(seed)
    model = M.ResidualGraphConvolutionalNetwork(train_batch_size=len(batches[0]), val_batch_size=n, num_layers=L,
                                                hidden_units=d, init_weights=init_w, layer_decay=decay)
    names = {"gcn_layer.dense.weight": "W1", "gcn_layer.dense.bias": "b1",
             "gcn_layer.dense2.weight": "W2", "gcn_layer.dense2.bias": "b2"}
    for k, v in model.state_dict().items():
        out["init_" + names[k]] = v.numpy().copy()
    loss_fn = M.GSS_loss(alpha).gss_loss
    opt = torch.optim.Adam(model.parameters(), lr=lr, weight_decay=0)   # train.py:139-141
    feats = torch.tensor(X32, dtype=torch.float32)

    # per-layer intermediates of the first forward (model.py:163-173), recomputed op by op
    if keep_layers:
        with torch.no_grad():
            x = feats
            residual = None
            for l in range(L):
                ax = torch.sparse.mm(adj_t, x)
                am = torch.sparse.mm(adj_t, ax * x)
                pre, o = model.gcn_layer(x, adj_t)
                out[f"L{l}_AX"] = ax.numpy().copy()
                out[f"L{l}_AM"] = am.numpy().copy()
                out[f"L{l}_P"] = pre.numpy().copy()
                x = o if residual is None else residual + decay * o
                residual = pre

    losses = []
    beta = None
    for it, idx in enumerate(batches):
        emb = model(x=feats, adj=adj_t)
        if it == 0:
            e0 = emb.data
            beta = np.percentile(np.dot(e0, e0.transpose(0, 1)).flatten(), beta_pct)   # train.py:165-167
            out["emb0"] = e0.numpy().copy()
            out["beta"] = np.float64(beta)
        loss = loss_fn(embs=emb, beta=beta, index=torch.as_tensor(idx))
        opt.zero_grad()
        loss.backward()
        if it == 0:
            for k, p in model.named_parameters():
                out["grad0_" + names[k]] = p.grad.numpy().copy()
        opt.step()
        losses.append(float(loss))
        if it == 0:
            for k, v in model.state_dict().items():
                out["step1_" + names[k]] = v.numpy().copy()
    for k, v in model.state_dict().items():
        out["final_" + names[k]] = v.numpy().copy()
    out["emb_last"] = emb.data.numpy().copy()      # from the last forward, before the last step (train.py:193)
    out["losses"] = np.asarray(losses, dtype=np.float64)
    out["batch_sizes"] = np.asarray([len(b) for b in batches], dtype=np.int64)
    out["batches"] = np.concatenate(batches).astype(np.int64)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: N={n} d={d} L={L} nnz(Ahat)={a_hat.nnz} beta={beta:.6f} losses={losses} -> {os.path.getsize(path)/1e6:.2f} MB")


def make_batches(n, b, steps, rng):
    out = []
    while len(out) < steps:
        perm = rng.permutation(n)
        out += [perm[i:i + b] for i in range(0, n, b)]
    return out[:steps]


def knn_adj(X32, k):
    """Reference gen_graph in descriptor mode (helper.py:25-58), x graph only."""
    X = X32.astype(np.float64).T
    _, _, x_adj, _ = H.gen_graph(X[:, 0:3], X, k, k, None, None)
    return x_adj


def toy_sif_adj():
    edges = [l.split() for l in open(os.path.join(REF, "2016data", "toy.sif")) if l.strip()]
    names = sorted({e[0] for e in edges} | {e[2] for e in edges})
    ix = {n: i for i, n in enumerate(names)}
    r = [ix[e[0]] for e in edges] + [ix[e[2]] for e in edges]
    c = [ix[e[2]] for e in edges] + [ix[e[0]] for e in edges]
    return sp.csr_matrix((np.ones(len(r)), (r, c)), shape=(len(names), len(names)))


def case_train_py():
    """Run the reference's train.py end to end and record what it consumed and produced."""
    import method.dataset as D
    n, d, seed = 200, 16, 7
    rng = np.random.RandomState(123)
    X = rng.randn(n, d).astype(np.float32)
    rec_batches, rec_losses, rec_beta = [], [], []
    orig_iter = D.DiffusionDataLoader.__iter__
    orig_loss = M.GSS_loss.gss_loss

    def rec_iter(self):
        for b in orig_iter(self):
            if self.batch_size != len(self.dataset):  # training loader only (validation loader is never iterated)
                rec_batches.append(b.numpy().copy())
            yield b

    def rec_loss(self, embs, beta, index=None):
        out = orig_loss(self, embs, beta, index)
        rec_losses.append(float(out))
        rec_beta.append(float(beta))
        return out

    D.DiffusionDataLoader.__iter__ = rec_iter
    M.GSS_loss.gss_loss = rec_loss
    cwd = os.getcwd()
    argv = sys.argv
    with tempfile.TemporaryDirectory() as tmp:
        emb_path = os.path.join(tmp, "in.embs.txt")
        with open(emb_path, "w") as f:
            f.write(f"{n} {d}\n")
            for i in range(n):
                f.write(f"node{i} " + " ".join(str(x) for x in X[i]) + "\n")   # node2vec.py:40-47 format
        os.chdir(tmp)
        sys.argv = ["train.py", "--emb-file", emb_path, "--num-layers", "2", "--hidden-units", str(d), "--k", "5",
                    "--kq", "5", "--epochs", "3", "--lr", "0.0003", "--graph-mode", "descriptor",
                    "--beta-percentile", "98", "--batch-size", "64", "--seed", str(seed)]
        try:
            runpy.run_path(os.path.join(REF, "train.py"), run_name="__main__")
            out_txt = open(os.path.join(tmp, "graph_embs.txt")).read()
            in_txt = open(emb_path).read()
        finally:
            os.chdir(cwd)
            sys.argv = argv
            D.DiffusionDataLoader.__iter__ = orig_iter
            M.GSS_loss.gss_loss = orig_loss
    path = os.path.join(HERE, "train_py_n200_d16.npz")
    np.savez_compressed(path, X=X, seed=np.int64(seed), batches=np.concatenate(rec_batches).astype(np.int64),
                        batch_sizes=np.asarray([len(b) for b in rec_batches], dtype=np.int64),
                        losses=np.asarray(rec_losses), beta=np.float64(rec_beta[0]),
                        graph_embs_txt=np.frombuffer(out_txt.encode(), dtype=np.uint8),
                        in_embs_txt=np.frombuffer(in_txt.encode(), dtype=np.uint8),
                        torch_version=np.frombuffer(torch.__version__.encode(), dtype=np.uint8))
    print(f"train_py_n200_d16: {len(rec_batches)} steps, beta={rec_beta[0]:.6f}, losses[:3]={rec_losses[:3]} "
          f"-> {os.path.getsize(path)/1e6:.2f} MB")


def case_negative_rowsum():
    """SURVEY a3's hazard, pinned: kNN mode on features with negative similarities.  One node points away from everybody else, so
    every inner product it keeps is negative and its row sum of A + I is < 0: np.power(rowsum, -0.5) (helper.py:85) is NaN
    there, and the NaN spreads to the rows and columns of their neighbours.  The reference does not guard; the fixture
    records what it produces -- the graph, the row sums, A_hat with its NaN / inf pattern, and the first forward's embeddings."""
    import warnings
    rng = np.random.RandomState(11)
    n, d, k = 96, 8, 5
    X = rng.randn(n, d) * 0.3
    X[:, 0] += 1.0                                  # a common direction: similarities among ordinary nodes are positive
    X[17] = -2.5 * X[17]                            # one node that points the other way: every similarity it keeps is negative
    X32 = X.astype(np.float32)
    adj = knn_adj(X32, k)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        a_hat = H.preprocess_graph(adj)                                # helper.py:82-89: NaN / inf, silently
    rowsum = np.asarray((sp.csr_matrix(adj) + sp.eye(n)).sum(1)).reshape(-1)
    bad = np.flatnonzero(~(rowsum > 0))
    assert len(bad) > 0, "the construction must produce a non-positive row sum"
    torch.manual_seed(5)
    np.random.seed(5)
    model = M.ResidualGraphConvolutionalNetwork(train_batch_size=32, val_batch_size=n, num_layers=2, hidden_units=d, init_weights=1e-5,
                                                layer_decay=0.3)
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        emb = model(x=torch.tensor(X32), adj=H.convert_sparse_matrix_to_sparse_tensor(a_hat)).numpy().copy()
    out = {"X": X32, "k": np.int64(k), "rowsum": rowsum, "bad_rows": bad.astype(np.int64), "emb0": emb}
    out.update(csr_parts(adj, "A"))
    a = sp.csr_matrix(a_hat)
    a.sort_indices()
    out.update({"Ahat_indptr": a.indptr.astype(np.int64), "Ahat_indices": a.indices.astype(np.int64), "Ahat_data": a.data.astype(np.float64)})
    path = os.path.join(HERE, "knn_negative_rowsum_n96_d8.npz")
    np.savez_compressed(path, **out)
    print(f"knn_negative_rowsum_n96_d8: rows with D_ii <= 0: {bad.tolist()} (sums {rowsum[bad].tolist()}), "
          f"non-finite entries of A_hat: {int((~np.isfinite(a.data)).sum())} of {a.nnz}, non-finite embedding rows: "
          f"{int((~np.isfinite(emb)).any(1).sum())} of {n} -> {os.path.getsize(path)/1e3:.1f} KB")


def main():
    if sys.argv[1:] == ["negative_rowsum"]:        # one case alone (the others are unchanged since round 1)
        case_negative_rowsum()
        return
    # 1. BASELINE config 1: toy.sif, d=64, L=2, full batch
    rng = np.random.RandomState(0)
    adj = toy_sif_adj()
    X = rng.randn(adj.shape[0], 64).astype(np.float32)
    run_case("toy_sif_d64_L2", adj, X, L=2, decay=0.3, init_w=1e-5, alpha=1.0, beta_pct=98,
             batches=[np.arange(adj.shape[0])] * 3, lr=3e-4, seed=1, keep_layers=True)

    # 2. kNN ("descriptor") mode, the adjacency train.py really builds
    rng = np.random.RandomState(1)
    X = rng.randn(200, 16).astype(np.float32)
    run_case("knn_n200_d16_L2", knn_adj(X, 5), X, L=2, decay=0.3, init_w=1e-5, alpha=1.0, beta_pct=98,
             batches=make_batches(200, 64, 4, rng), lr=3e-4, seed=2, keep_layers=True)

    # 3. three layers, larger init noise so off-diagonal weights matter
    rng = np.random.RandomState(2)
    X = (rng.randn(2000, 64) / 8.0).astype(np.float32)
    run_case("knn_n2000_d64_L3", knn_adj(X, 5), X, L=3, decay=0.4, init_w=1e-2, alpha=1.0, beta_pct=95,
             batches=make_batches(2000, 512, 4, rng), lr=1e-3, seed=3, keep_layers=False)

    # 4. edgelist mode: directed, asymmetric positive weights (MSI-style type-normalised weights, msi.py:255-262)
    rng = np.random.RandomState(3)
    n, m = 600, 6000
    src = rng.randint(0, n, m)
    dst = rng.randint(0, n, m)
    keep = src != dst
    key = src[keep].astype(np.int64) * n + dst[keep]
    _, first = np.unique(key, return_index=True)
    src, dst = src[keep][first], dst[keep][first]
    w = rng.uniform(0.05, 1.0, len(src))
    adj = sp.csr_matrix((w, (src, dst)), shape=(n, n))
    X = (rng.randn(n, 128) / 11.0).astype(np.float32)
    run_case("edge_n600_d128_L2", adj, X, L=2, decay=0.3, init_w=1e-3, alpha=2.0, beta_pct=98,
             batches=make_batches(n, 256, 4, rng), lr=3e-4, seed=4, keep_layers=True)

    # 5. the reference's train.py, end to end
    case_train_py()

    # 6. a non-positive row sum (SURVEY a3 hazard)
    case_negative_rowsum()


if __name__ == "__main__":
    main()
