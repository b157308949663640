"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/gssgcn.h declares
(no compute calls -- there is no GPU here)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import gcn_drug_repurposing_amd as pkg
    if not os.path.exists(pkg._lib.LIB_PATH):
        pkg.build()
    return pkg.load()


def header_functions():
    text = open(os.path.join(ROOT, "include", "gssgcn.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gss_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound(lib):
    import gcn_drug_repurposing_amd as pkg
    names = header_functions()
    assert len(names) >= 30
    out = subprocess.run(["nm", "-D", "--defined-only", pkg._lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (gss_[a-z0-9_]+)", out))
    assert set(names) <= exported, sorted(set(names) - exported)
    assert set(names) == set(pkg._lib.SIGNATURES), sorted(set(names) ^ set(pkg._lib.SIGNATURES))
    for n in names:
        assert getattr(lib, n) is not None


def test_abi_version_and_error_text(lib):
    import gcn_drug_repurposing_amd as pkg
    header = open(os.path.join(ROOT, "include", "gssgcn.h")).read()
    declared = int(re.search(r"#define GSS_ABI_VERSION (\d+)", header).group(1))
    assert lib.gss_abi_version() == declared == pkg._lib.ABI_VERSION      # header, library and binding agree
    assert isinstance(lib.gss_last_error(), bytes)


def test_code_object_is_gfx950_only():
    import gcn_drug_repurposing_amd as pkg
    blob = open(pkg._lib.LIB_PATH, "rb").read()
    assert b"gfx950" in blob
    for other in (b"gfx942", b"gfx90a", b"sm_90"):
        assert other not in blob


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    import gcn_drug_repurposing_amd as pkg
    monkeypatch.setattr(pkg._lib, "_lib", None)
    monkeypatch.setattr(pkg._lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(pkg.GssError, match="no CPU fallback"):
        pkg._lib.load()


def test_product_package_never_imports_the_oracle():
    pkg_dir = os.path.join(ROOT, "gcn-drug-repurposing_amd")
    for dirpath, _, files in os.walk(pkg_dir):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", text, flags=re.M), f
    for f in ("train.py",):
        assert "oracle" not in open(os.path.join(ROOT, f)).read()


def test_workspace_bounds_cover_every_shorter_batch(lib):
    """Host arithmetic only.  The loss workspace is not monotone in the batch size (fewer row tiles get more column slabs) and neither is
    the weight-gradient slice count: a plan sized for max_batch alone would be too small for the shorter last batch
    of an epoch.  gss_loss_workspace_bytes_max is what plans carve; it covers every batch of 1..b_max rows."""
    for d in (64, 128, 256):
        for b_max in (2048, 1034, 333, 17):
            cap = lib.gss_loss_workspace_bytes_max(b_max, d)
            sizes = [lib.gss_loss_workspace_bytes(b, d) for b in range(1, b_max + 1)]
            assert cap == max(sizes)
    # the case that motivated it: the tail batch of N = 29,960 at B = 2048, and one just below B
    assert lib.gss_loss_workspace_bytes(2032, 128) > lib.gss_loss_workspace_bytes(2048, 128)


def test_library_carries_the_hashes_of_the_sources_in_the_tree(lib):
    """gss_source_hash: the library a test run (and the GPU box) loads was built from exactly the sources in this tree -- a stale .so with
    fresh timestamps is caught here and rebuilt by __graft_entry__.build(); evidence files record these hashes (bench.py compares)"""
    import gcn_drug_repurposing_amd as pkg
    want = pkg._lib.source_hashes()
    assert {"spmm.hip", "dense.hip", "plan.hip", "common.h", "gssgcn.h", "*"} <= set(want)
    for name, h in want.items():
        got = lib.gss_source_hash(name.encode())
        assert got is not None and got.decode() == h, f"{name}: the library was built from another version of this file; run __graft_entry__.build()"
    assert lib.gss_source_hash(None).decode() == want["*"]
    assert lib.gss_source_hash(b"no_such_file.hip") is None
    assert pkg._lib.library_is_current()


def test_removed_knobs_are_refused_by_name(lib):
    """round 6 pruned the access-shape knobs: 17 remain, the removed names are errors, not silent no-ops"""
    for name in ("gemm_ws_mode", "gemm_ws_stagger", "gemm_ws_wgs", "wgrad_variant", "spmm_fly", "gemm_small_nt", "gemm_nt_cap",
                 "gemm_lines", "gemm_hoist", "gemm_rows_split", "wgrad_deep", "spmm_pair", "xcd_remap"):
        assert lib.gss_debug_set_option(name.encode(), 1) != 0, name
        assert name in lib.gss_last_error().decode()
    text = open(os.path.join(ROOT, "gcn-drug-repurposing_amd", "csrc", "common.h")).read()
    body = text[text.index("struct Knobs {"):text.index("};", text.index("struct Knobs {"))]
    assert len(re.findall(r"^  int \w+ = ", body, flags=re.M)) <= 20
