"""CPU: host-side logic of the product package (graph building, file formats, sampler, CLI errors)."""
import numpy as np
import pytest
import scipy.sparse as sp

from conftest import golden_csr, load_golden
from oracle import gss_oracle as O


def test_knn_builder_matches_reference_gen_graph_chunked():
    from gcn_drug_repurposing_amd.graph import knn_descriptor_adj
    for name in ("knn_n200_d16_L2", "knn_n2000_d64_L3"):
        g = load_golden(name)
        ref = golden_csr(g, "A")
        for chunk in (4096, 77):
            adj = knn_descriptor_adj(g["X"].astype(np.float64), 5, chunk=chunk)
            assert np.array_equal(adj.indptr, ref.indptr) and np.array_equal(adj.indices, ref.indices)
            np.testing.assert_allclose(adj.data, ref.data, rtol=1e-13)


def test_edgelist_adj_last_duplicate_wins():
    from gcn_drug_repurposing_amd.graph import edgelist_adj
    a = edgelist_adj([0, 1, 0, 2], [1, 2, 1, 0], [1.0, 2.0, 5.0, 3.0], 3)
    assert a[0, 1] == 5.0 and a[1, 2] == 2.0 and a[2, 0] == 3.0 and a.nnz == 3
    b = O.edgelist_to_adj([0, 1, 0, 2], [1, 2, 1, 0], [1.0, 2.0, 5.0, 3.0], 3)
    assert abs(a - b).max() == 0


def test_embs_reader_writer_roundtrip(tmp_path):
    from gcn_drug_repurposing_amd import embio
    g = load_golden("train_py_n200_d16")
    p = tmp_path / "in.embs.txt"
    p.write_bytes(bytes(g["in_embs_txt"]))
    names, x = embio.read_embs(str(p))
    names_o, x_o = O.read_embs(str(p))
    assert names == names_o and np.array_equal(x, x_o)
    assert np.array_equal(x.astype(np.float32), g["X"])
    embio.write_embs(str(tmp_path / "w.txt"), names, g["X"])
    _, x2 = embio.read_embs(str(tmp_path / "w.txt"))
    assert np.array_equal(x2.astype(np.float32), g["X"])
    out = tmp_path / "graph_embs.txt"
    embio.write_graph_embs(str(out), g["X"])
    assert out.read_text() == "\n".join(" ".join("%.18e" % v for v in row) for row in g["X"]) + "\n"
    with pytest.raises(ValueError, match="header says"):
        p.write_text("5 2\na 1 2\n")
        embio.read_embs(str(p))


def test_edgelist_and_sif_readers(tmp_path):
    from gcn_drug_repurposing_amd import embio
    e = tmp_path / "g.edgelist"
    e.write_text("n1 n0 0.5\nn0 n2 2\n")
    src, dst, w, names = embio.read_edgelist(str(e), ["n0", "n1", "n2"])
    assert src.tolist() == [1, 0] and dst.tolist() == [0, 2] and w.tolist() == [0.5, 2.0]
    with pytest.raises(KeyError):
        embio.read_edgelist(str(e), ["n0", "n1"])
    s = tmp_path / "t.sif"
    s.write_text("A 1 B\nB 1 C\n")
    src, dst, w, names = embio.read_edgelist(str(s))
    assert names == ["A", "B", "C"] and sorted(zip(src.tolist(), dst.tolist())) == [(0, 1), (1, 0), (1, 2), (2, 1)]


def test_sampler_reproduces_reference_batches_for_same_seed():
    """the reference draws batches from torch's global RNG after the model init consumed part of it
    (train.py:74-76,111-131); the trainer follows the same sequence."""
    torch = pytest.importorskip("torch")
    from torch.utils.data import DataLoader

    from gcn_drug_repurposing_amd import trainer
    from gcn_drug_repurposing_amd.model import ResidualGraphConvolutionalNetwork
    g = load_golden("train_py_n200_d16")
    if bytes(g["torch_version"]).decode() != torch.__version__:
        pytest.skip("fixture recorded with another torch version")
    seed = int(g["seed"])
    torch.manual_seed(seed)
    np.random.seed(seed)
    ResidualGraphConvolutionalNetwork(64, 200, 2, 16, 1e-5, 0.3)
    loader = DataLoader(trainer._IndexDataset(200), batch_size=64, shuffle=True, num_workers=0)
    got = np.concatenate([b.numpy() for _ in range(3) for b in trainer.epoch_batches(loader)])
    assert np.array_equal(got, g["batches"])


def test_cli_argument_errors_match_reference():
    from gcn_drug_repurposing_amd import trainer
    with pytest.raises(Exception, match="can not be used at the same time"):
        trainer.main(["--emb-file", "x", "--beta", "0.1", "--beta-percentile", "98"])
    with pytest.raises(Exception, match="At least one of beta and beta_percentile"):
        trainer.main(["--emb-file", "x"])
    args = trainer.build_parser().parse_args(["--emb-file", "x", "--dataset", "rparis6k", "--report-hard", "--kq", "7"])
    assert args.epochs == 200 and args.lr == 0.0001 and args.layer_decay == 0.3 and args.hidden_units == 128


def test_bench_self_launch_builds_a_torchrun_command(monkeypatch):
    """python bench.py --gpus N outside a torch.distributed job: N ranks are started as a CHILD torch.distributed.run (never an
    exec of a process that touched the GPU), the child's JSON line and exit code are relayed"""
    import importlib.util
    import os
    import subprocess
    import sys
    import types
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}

    def fake_run(cmd, env=None, stdout=None, text=None):
        seen["cmd"], seen["env"] = cmd, env
        return types.SimpleNamespace(returncode=0, stdout='noise\n{"metric": "gcn_spmm_edges_per_s", "n_gpus": 4}\n')

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7", "--warmup", "2"])
    with pytest.raises(SystemExit) as e:
        bench.self_launch(types.SimpleNamespace(gpus=4))
    assert e.value.code == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-6:] == ["--gpus", "4", "--steps", "7", "--warmup", "2"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_native_writer_is_byte_identical_to_np_savetxt(tmp_path):
    """gss_write_embs_text / gss_format_e18 (train.py:193's np.savetxt): exact decimal expansion of widened float32s, round
    half to even -- every value byte for byte what Python's '%.18e' prints, over random bit patterns, denormals, powers of
    two and their neighbours, infinities and NaNs"""
    import ctypes as C
    from gcn_drug_repurposing_amd import _lib, embio
    lib = _lib.load()
    rng = np.random.RandomState(5)
    bits = np.concatenate([rng.randint(0, 2 ** 32, size=60000, dtype=np.uint64).astype(np.uint32),
                           np.array([0, 0x80000000, 1, 2, 3, 0x007fffff, 0x00800000, 0x7f7fffff, 0x3f800000, 0x7f800000, 0xff800000, 0x7fc00000,
                                     0xffc00000, 0x3effffff, 0x4b000000, 0x4b7fffff, 0x4b800000], dtype=np.uint32),
                           np.arange(1, 255, dtype=np.uint32) << 23, (np.arange(1, 255, dtype=np.uint32) << 23) | 1,
                           (np.arange(1, 255, dtype=np.uint32) << 23) | 0x7fffff, np.arange(1, 600, dtype=np.uint32)])
    buf = C.create_string_buffer(32)
    for v in bits.view(np.float32):
        n = lib.gss_format_e18(C.c_float(float(v)), buf)
        want = "%.18e" % float(v)
        assert buf.value.decode() == want and n == len(want), (v, buf.value, want)
    x = (rng.randn(777, 48) / 9).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    x[5, 7], x[6, 0] = 0.0, -0.0
    a, b, c = tmp_path / "a.txt", tmp_path / "b.txt", tmp_path / "c.txt"
    np.savetxt(a, x)
    embio.write_graph_embs(str(b), x)                    # all cores
    embio.write_graph_embs(str(c), x, threads=3)
    assert a.read_bytes() == b.read_bytes() == c.read_bytes()
    np.testing.assert_array_equal(np.loadtxt(b).astype(np.float32), x)       # and it reads back to the same floats
    x64 = rng.randn(5, 4)                                # not float32: numpy's own path, still np.savetxt's bytes
    embio.write_graph_embs(str(b), x64)
    np.savetxt(a, x64)
    assert a.read_bytes() == b.read_bytes()


def test_native_embs_reader_matches_python_float(tmp_path):
    """gss_embs_open (train.py:79-80): names in order, values the doubles float() parses, header check, blank lines skipped;
    malformed files fall through to the strict Python parser"""
    from gcn_drug_repurposing_amd import embio
    rng = np.random.RandomState(6)
    n, d = 1500, 24
    x = rng.randn(n, d) * np.exp(rng.randn(n, 1) * 8)      # wide range of magnitudes, 17 significant digits each
    names = [f"GO:{i:07d}" if i % 2 else str(100000 - i) for i in range(n)]
    p = tmp_path / "e.embs.txt"
    embio.write_embs(str(p), names, x)
    text = p.read_text().splitlines()
    text.insert(7, "")                                    # a blank line in the middle, trailing spaces on another
    text[20] = text[20] + "  "
    p.write_text("\n".join(text) + "\n")
    for threads in (1, 4, 0):
        got_names, got = embio.read_embs(str(p), threads=threads)
        assert got_names == names and np.array_equal(got, x)
    p.write_text(f"{n + 1} {d}\n" + "\n".join(text[1:]) + "\n")
    with pytest.raises(ValueError, match="header says"):
        embio.read_embs(str(p))
    bad = tmp_path / "bad.embs.txt"
    bad.write_text("2 3\na 1 2 3\nb 1 2\n")                   # ragged: the native reader refuses, the Python parser raises
    with pytest.raises(Exception):
        embio.read_embs(str(bad))


def test_native_edgelist_reader_matches_the_python_parser(tmp_path):
    """gss_edgelist_open (the --adj-file reader for 'u v w' lines, predict_drug.py:224-226): same (src, dst, w) as the Python loop,
    comments / blank lines / missing weights handled, unknown node ids reported by the strict parser"""
    from gcn_drug_repurposing_amd import embio
    rng = np.random.RandomState(8)
    names = [f"GO:{i:07d}" if i % 3 == 0 else (f"DB{i:05d}" if i % 3 == 1 else str(i)) for i in range(700)]
    m = 20000
    u, v = rng.randint(0, 700, m), rng.randint(0, 700, m)
    w = rng.rand(m) * np.exp(rng.randn(m) * 3)
    p = tmp_path / "g.edgelist"
    with open(p, "w") as f:
        f.write("# a comment\n\n")
        for k in range(m):
            if k % 997 == 0:
                f.write(f"{names[u[k]]}\t{names[v[k]]}\n")               # no weight -> 1.0, tab separated
                w[k] = 1.0
            else:
                f.write(f"{names[u[k]]} {names[v[k]]} {float(w[k])!r}\n")
    src, dst, ww, nm = embio.read_edgelist(str(p), names)
    assert nm == names and np.array_equal(src, u) and np.array_equal(dst, v) and np.array_equal(ww, w)
    with open(p, "a") as f:
        f.write("not_a_node GO:0000000 1.0\n")
    with pytest.raises(KeyError, match="not_a_node"):
        embio.read_edgelist(str(p), names)


def test_native_edgelist_reader_rejects_what_the_python_parser_would_not_read_the_same_way(tmp_path):
    """the native fast path has to be equivalent to the Python loop it replaces.  strtod reads C hex floats, 'inf' and
    'nan' and ignores what follows the weight; such lines are now reported as bad (gss_edgelist_bad_line) so that the file goes through
    the Python parser -- which raises on a hex weight and reads the others exactly as it always did."""
    import ctypes as C
    import gcn_drug_repurposing_amd as pkg
    from gcn_drug_repurposing_amd import embio
    lib = pkg.load()
    names = ["a", "b", "c"]
    blob = "\n".join(names).encode()

    def native_bad_line(text):
        p = tmp_path / "e.edgelist"
        p.write_text(text)
        h = C.c_void_p()
        assert lib.gss_edgelist_open(C.byref(h), str(p).encode(), blob, len(blob), len(names), 2) == 0
        try:
            return lib.gss_edgelist_bad_line(h), p
        finally:
            lib.gss_edgelist_close(h)

    assert native_bad_line("a b 1.5\nb c 2e-3\nc a\n")[0] == -1
    for bad in ("a b 0x1p3\n", "a b inf\n", "a b nan\n", "a b 1.0 extra\n", "a b 1_0\n", "a b 1.5abc\n", "a b -Infinity\n"):
        line, _ = native_bad_line("a b 1.0\n" + bad)
        assert line == 1, bad
    # what the caller sees: hex raises (Python's float()), inf and trailing tokens read as the Python loop reads them
    _, p = native_bad_line("a b 0x1p3\n")
    with pytest.raises(ValueError):
        embio.read_edgelist(str(p), names)
    _, p = native_bad_line("a b inf\nb c 2.0 trailing tokens\n")
    src, dst, w, _ = embio.read_edgelist(str(p), names)
    assert src.tolist() == [0, 1] and dst.tolist() == [1, 2] and w[0] == np.inf and w[1] == 2.0
    # a name with a newline in it (any position, not just the first) keeps the file away from the native name table
    p.write_text("a b 1.0\n")
    src, dst, w, _ = embio.read_edgelist(str(p), ["a", "b", "c\nd"])
    assert src.tolist() == [0] and dst.tolist() == [1]


def test_native_embs_reader_rejects_non_finite_and_glued_tokens(tmp_path):
    import ctypes as C
    import gcn_drug_repurposing_amd as pkg
    lib = pkg.load()
    for body, ok in (("n0 1.0 2.0\nn1 3.0 4.0\n", True), ("n0 1.0 inf\nn1 3.0 4.0\n", False), ("n0 1.0 2.0x\nn1 3 4\n", False),
                     ("n0 1.0 0x10\nn1 3 4\n", False), ("n0 nan 2.0\nn1 3 4\n", False)):
        p = tmp_path / "x.embs.txt"
        p.write_text("2 2\n" + body)
        h = C.c_void_p()
        rc = lib.gss_embs_open(C.byref(h), str(p).encode(), 1)
        assert (rc == 0) == ok, body
        if rc == 0:
            lib.gss_embs_close(h)


def test_gss_options_environment_is_checked(tmp_path):
    """GSS_OPTIONS=name=value,... is applied when the library is loaded; a bad entry is an error, not ignored"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = "import gcn_drug_repurposing_amd as p; p.load(); print('ok')"
    ok = subprocess.run([sys.executable, "-c", code], cwd=root, env=dict(os.environ, GSS_OPTIONS="lazy_halo=1, spmm_slices=2"), capture_output=True, text=True)
    assert ok.returncode == 0 and "ok" in ok.stdout, ok.stderr[-2000:]
    for bad in ("lazy_halo=7", "no_such_knob=1", "lazy_halo"):
        r = subprocess.run([sys.executable, "-c", code], cwd=root, env=dict(os.environ, GSS_OPTIONS=bad), capture_output=True, text=True)
        assert r.returncode != 0 and "GSS_OPTIONS" in r.stderr, (bad, r.stderr[-500:])


def test_partition_row_weight_follows_width_and_depth():
    """shards.row_weight_for: the cost of a row in stored entries that the node-range partition balances with -- the measured 18 at
    d = 128, L = 2 (tools/scaling_forecast.py), doubling with the width (dense work ~ d^2, SpMM work ~ d), near-constant in the depth"""
    from gcn_drug_repurposing_amd.shards import ROW_WEIGHT, row_weight_for
    assert row_weight_for(128, 2) == 18 and row_weight_for(256, 2) == 36 and row_weight_for(64, 2) == 9
    assert 30 <= row_weight_for(256, 3) <= 40 and row_weight_for(16, 2) >= 4 and ROW_WEIGHT == 12
