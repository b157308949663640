"""Pin the numpy oracle (oracle/gss_oracle.py) against fixtures produced by the reference itself."""
import numpy as np
import scipy.sparse as sp

from conftest import golden_batches, golden_csr, golden_params, load_golden
from oracle import gss_oracle as O


def rel(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def test_preprocess_graph(op_case):
    name, g = op_case
    a_hat, rowsum = O.preprocess_graph(golden_csr(g, "A"))
    ref = golden_csr(g, "Ahat")
    assert np.array_equal(a_hat.indptr, ref.indptr) and np.array_equal(a_hat.indices, ref.indices)
    np.testing.assert_allclose(a_hat.data, ref.data, rtol=1e-14, atol=0)
    np.testing.assert_allclose(rowsum, g["rowsum"], rtol=1e-14, atol=0)


def test_non_positive_row_sum_gives_the_reference_nan_pattern():
    """SURVEY a3's hazard, pinned by a reference-generated fixture: a node whose kept similarities are all negative has D_ii < 0, the
    reference's np.power(rowsum, -0.5) is NaN there and every entry of that node's row and column of A_hat is NaN; the oracle (same
    arithmetic, no guard) must produce NaN at exactly the same entries and the same values elsewhere"""
    import warnings
    g = load_golden("knn_negative_rowsum_n96_d8")
    n = g["X"].shape[0]
    adj = O.gen_graph_descriptor(g["X"].astype(np.float64).T, int(g["k"]))
    ref_a = sp.csr_matrix((g["A_data"], g["A_indices"], g["A_indptr"]), shape=(n, n))
    assert np.array_equal(adj.indptr, ref_a.indptr) and np.array_equal(adj.indices, ref_a.indices)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        a_hat, rowsum = O.preprocess_graph(adj)
    np.testing.assert_allclose(rowsum, g["rowsum"], rtol=1e-13)
    assert np.array_equal(np.flatnonzero(~(rowsum > 0)), g["bad_rows"]) and len(g["bad_rows"]) == 1
    assert np.array_equal(a_hat.indptr, g["Ahat_indptr"]) and np.array_equal(a_hat.indices, g["Ahat_indices"])
    nan_ref = ~np.isfinite(g["Ahat_data"])
    assert nan_ref.sum() > 0 and np.array_equal(~np.isfinite(a_hat.data), nan_ref)
    np.testing.assert_allclose(a_hat.data[~nan_ref], g["Ahat_data"][~nan_ref], rtol=1e-13)
    # the NaN entries are the bad node's row and column, nothing else
    rows = np.repeat(np.arange(n), np.diff(a_hat.indptr))
    bad = int(g["bad_rows"][0])
    assert np.array_equal(nan_ref, (rows == bad) | (a_hat.indices == bad))


def test_knn_graph_matches_reference_gen_graph():
    for name in ("knn_n200_d16_L2", "knn_n2000_d64_L3"):
        g = load_golden(name)
        adj = O.gen_graph_descriptor(g["X"].astype(np.float64).T, 5)
        ref = golden_csr(g, "A")
        assert np.array_equal(adj.indptr, ref.indptr) and np.array_equal(adj.indices, ref.indices)
        np.testing.assert_allclose(adj.data, ref.data, rtol=1e-13)


def test_init_weights_follow_numpy_rng(op_case):
    name, g = op_case
    d = int(g["meta"][1])
    np.random.seed(int(g["seed"]))
    p = O.init_layer_weights(d, float(g["init_weights"]))
    for k in ("W1", "b1", "W2", "b2"):
        assert np.array_equal(p[k], g["init_" + k]), k


def test_forward_layers_and_embedding(op_case):
    name, g = op_case
    n, d, L = (int(v) for v in g["meta"])
    a32 = O.to_fp32_csr(golden_csr(g, "Ahat"))
    params = golden_params(g, "init")
    for dtype, tol in ((np.float64, 2e-6), (np.float32, 5e-6)):
        emb, cache = O.forward(g["X"], a32, params, L, float(g["decay"]), dtype=dtype)
        assert rel(emb, g["emb0"]) < tol, (dtype, rel(emb, g["emb0"]))
        if "L0_AX" in g:
            for l in range(L):
                for key in ("AX", "AM", "P"):
                    r = rel(cache["layers"][l][key], g[f"L{l}_{key}"])
                    assert r < tol, (dtype, l, key, r)


def test_beta_percentile(op_case):
    name, g = op_case
    beta = O.beta_percentile(g["emb0"], float(g["beta_pct"]))
    assert abs(beta - float(g["beta"])) < 1e-6


def test_loss_and_grads_first_step(op_case):
    name, g = op_case
    n, d, L = (int(v) for v in g["meta"])
    a32 = O.to_fp32_csr(golden_csr(g, "Ahat"))
    params = golden_params(g, "init")
    idx = golden_batches(g)[0]
    beta, alpha = float(g["beta"]), float(g["alpha"])
    emb, cache = O.forward(g["X"], a32, params, L, float(g["decay"]), dtype=np.float64)
    loss = O.gss_loss(emb, beta, idx, alpha)
    assert abs(loss - g["losses"][0]) < 2e-6 * abs(g["losses"][0]) + 1e-9
    grads = O.backward(cache, O.loss_grad_emb(emb, beta, idx, alpha))
    for k in ("W1", "b1", "W2", "b2"):
        ref = g["grad0_" + k]
        err = np.abs(grads[k] - ref).max()
        # torch fp32 autograd vs fp64 oracle: 1e-5 relative to the largest gradient entry
        assert err < 1e-5 * np.abs(g["grad0_W1"]).max() + 1e-10, (k, err, np.abs(ref).max())


def test_adam_first_step(op_case):
    name, g = op_case
    params = golden_params(g, "init")
    grads = {k: g["grad0_" + k] for k in ("W1", "b1", "W2", "b2")}
    O.adam_step(params, grads, {}, float(g["lr"]))
    for k in ("W1", "b1", "W2", "b2"):
        # Adam's first step is lr*sign(g) wherever |g| >> eps: compare updates, not weights
        upd = params[k] - g["init_" + k]
        ref = g["step1_" + k] - g["init_" + k]
        assert np.abs(upd - ref).max() < 2e-3 * float(g["lr"]) + 1e-9, k


def test_training_trajectory(op_case):
    """Several steps of train.py:151-191 on the recorded batches, fp32 oracle vs torch."""
    name, g = op_case
    n, d, L = (int(v) for v in g["meta"])
    a32 = O.to_fp32_csr(golden_csr(g, "Ahat"))
    params = golden_params(g, "init")
    emb, params, losses, beta = O.train(g["X"], a32, params, L, float(g["decay"]), golden_batches(g),
                                        beta_pct=float(g["beta_pct"]), alpha=float(g["alpha"]), lr=float(g["lr"]),
                                        dtype=np.float64)
    np.testing.assert_allclose(losses, g["losses"], rtol=2e-4, atol=1e-8)
    assert rel(emb, g["emb_last"]) < 2e-3   # weights moved by a few lr; sign-of-tiny-gradient effects allowed
    for k in ("W1", "W2"):
        assert np.abs(params[k] - g["final_" + k]).max() < 2.5 * float(g["lr"])


def test_train_py_end_to_end_fixture():
    """The reference's train.py run (N=200, d=16, 3 epochs): replay its recorded batches through the oracle."""
    g = load_golden("train_py_n200_d16")
    X = g["X"]
    txt = bytes(g["graph_embs_txt"]).decode()
    ref_emb = np.loadtxt(txt.splitlines())
    assert ref_emb.shape == X.shape
    np.testing.assert_allclose(np.sqrt((ref_emb ** 2).sum(1)), 1.0, atol=1e-6)
    # '%.18e' format of np.savetxt (train.py:193)
    assert len(txt.splitlines()[0].split(" ")[0]) == len("%.18e" % 0.5)
    adj = O.gen_graph_descriptor(X.astype(np.float64).T, 5)
    a_hat, _ = O.preprocess_graph(adj)
    np.random.seed(int(g["seed"]))
    params = O.init_layer_weights(16, 1e-5)
    batches, o = [], 0
    for s in g["batch_sizes"]:
        batches.append(g["batches"][o:o + int(s)])
        o += int(s)
    assert len(batches) == 12 and [len(b) for b in batches[:4]] == [64, 64, 64, 8]
    emb, params, losses, beta = O.train(X, O.to_fp32_csr(a_hat), params, 2, 0.3, batches, beta_pct=98.0,
                                        alpha=1.0, lr=3e-4, dtype=np.float64)
    assert abs(beta - float(g["beta"])) < 1e-6
    np.testing.assert_allclose(losses, g["losses"], rtol=5e-4, atol=1e-8)
    assert np.abs(emb - ref_emb).max() < 5e-3


def test_embs_reader_roundtrip(tmp_path):
    g = load_golden("train_py_n200_d16")
    p = tmp_path / "in.embs.txt"
    p.write_bytes(bytes(g["in_embs_txt"]))
    names, x = O.read_embs(str(p))
    assert names[:2] == ["node0", "node1"] and x.shape == (200, 16)
    assert np.array_equal(x.astype(np.float32), g["X"])


def test_auc_helper_matches_sklearn():
    from sklearn.metrics import roc_auc_score
    rng = np.random.RandomState(0)
    y = rng.rand(300) < 0.2
    s = np.round(rng.randn(300), 1)  # ties
    assert abs(O.roc_auc(y, s) - roc_auc_score(y, s)) < 1e-12


def test_torch_cpu_path_matches_fixture(op_case):
    """oracle/torch_cpu_path.py (bench.py's cpu_baseline 'port') reproduces the reference's trajectory"""
    from oracle.torch_cpu_path import TorchCpuPath
    name, g = op_case
    n, d, L = (int(v) for v in g["meta"])
    cpu = TorchCpuPath(O.to_fp32_csr(golden_csr(g, "Ahat")), g["X"], golden_params(g, "init"), L, float(g["decay"]),
                       float(g["alpha"]), float(g["lr"]))
    losses = []
    for idx in golden_batches(g):
        emb, loss = cpu.step(idx, float(g["beta"]))
        losses.append(loss)
    np.testing.assert_allclose(losses, g["losses"], rtol=1e-5, atol=1e-9)
    assert rel(emb.numpy(), g["emb_last"]) < 1e-5
