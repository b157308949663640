"""Diffusion-profile baseline: the oracle against the reference's own output
(tests/golden/diffusion_msi_small.npz, made by tests/golden/make_diffusion_fixture.py), the host-side index lists
against the oracle (CPU, through a numpy statement of the device algorithm), and the HIP path against both (-m gpu)."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

from conftest import GOLDEN

SMALL = os.path.join(GOLDEN, "msi_small")
FILES = {k: os.path.join(SMALL, k + ".tsv") for k in ("drug_to_protein", "indication_to_protein", "protein_to_protein",
                                                      "protein_to_functional_pathway", "functional_pathway_to_functional_pathway")}


@pytest.fixture(scope="module")
def gold():
    z = np.load(os.path.join(GOLDEN, "diffusion_msi_small.npz"))
    g = {k: z[k] for k in z.files}
    nodes = [str(v) for v in g["nodelist"]]
    idx = {n: i for i, n in enumerate(nodes)}
    g["m0"] = sp.csr_matrix((g["m_data"], g["m_indices"], g["m_indptr"]), shape=(len(nodes),) * 2)
    g["idx"] = idx
    g["start_idx"] = np.array([idx[str(s)] for s in g["starts"]])
    g["prot"] = {idx[str(s)]: [idx[p] for p in str(ps).split()] for s, ps in zip(g["starts"], g["proteins_of"])}
    g["hp"] = (float(g["alpha"]), int(g["max_iter"]), float(g["tol"]))
    return g


def edge_case_graph():
    """drug 0 -- protein 4 (4's only neighbour: its row becomes empty when 0 is the start node); drug 1 -- proteins 5, 6 and,
    unusually, pathway 8 (an in-edge of a start node that is not cut); indication 2 -- protein 5; node 3: a drug with
    no edges at all (empty row, and a start node whose own row is empty); proteins 5-6-7 chained, 7 -- pathway 8"""
    e = [(0, 4, 1.5), (1, 5, 0.7), (1, 6, 0.4), (1, 8, 0.9), (2, 5, 2.0), (5, 6, 1.1), (6, 7, 0.3), (7, 8, 1.3)]
    r = [a for a, b, w in e] + [b for a, b, w in e]
    c = [b for a, b, w in e] + [a for a, b, w in e]
    w = [w for a, b, w in e] + [w * 1.7 for a, b, w in e]
    m0 = sp.csr_matrix((w, (r, c)), shape=(9, 9))
    return m0, {0: [4], 1: [5, 6], 2: [5], 3: []}


def oracle_profiles(m0, starts, prot, hp):
    from oracle import diffusion_oracle as O
    res = [O.diffusion_profile(m0, int(s), prot, *hp) for s in starts]
    return np.stack([r[0] for r in res]), np.array([r[1] for r in res])


def test_oracle_matches_reference_profiles(gold):
    x, it = oracle_profiles(gold["m0"], gold["start_idx"], gold["prot"], gold["hp"])
    assert np.abs(x - gold["profiles"]).max() < 1e-15
    assert abs(x.sum(1) - 1).max() < 1e-12 and it.min() > 5


def test_loader_reproduces_the_reference_protein_sets_and_matrix(gold):
    from gcn_drug_repurposing_amd.msi import MsiGraph
    g = MsiGraph().load(FILES)
    names = g.names
    assert names == [str(v) for v in gold["nodelist"]]
    assert sorted(g.drugs_in_graph + g.indications_in_graph) == [str(s) for s in gold["starts"]]
    for s, ps in zip(gold["starts"], gold["proteins_of"]):
        assert sorted(g.drug_or_indication2proteins[str(s)]) == str(ps).split()
    from gcn_drug_repurposing_amd.msi import COVID_WEIGHTS
    m0, _, _ = g.weight_graph(COVID_WEIGHTS).to_csr()
    assert (m0 != gold["m0"]).nnz == 0    # nx.to_scipy_sparse_matrix of the weighted graph, entry for entry


@pytest.mark.parametrize("case", ["msi_small", "edge"])
def test_index_lists_reproduce_the_oracle_on_cpu(gold, case):
    from cpu_ops import emulate_ppr
    from gcn_drug_repurposing_amd.diffusion import PprProblem
    if case == "msi_small":
        m0, starts, prot, hp = gold["m0"], gold["start_idx"], gold["prot"], gold["hp"]
    else:
        (m0, prot), starts, hp = edge_case_graph(), np.array([0, 1, 2, 3]), (0.85, 500, 1e-9)
    prob = PprProblem(m0, starts, prot)
    x, it = emulate_ppr(prob, *[hp[0], hp[2], hp[1]])
    ref, ref_it = oracle_profiles(m0, starts, prot, hp)
    assert np.abs(x.T - ref).max() < 1e-14
    assert (it == ref_it).all()
    if case == "edge":
        assert len(prob.zero_ovr) == 1 and len(prob.keep_row) == 1 and prob.start_dangling.tolist() == [0, 0, 0, 1]
        assert 1 not in prob.z_rows and 3 in prob.z_rows   # drug 1 keeps its pathway edge when it is not the start node


def test_file_names_and_loader_follow_the_reference(gold, tmp_path):
    from gcn_drug_repurposing_amd.diffusion import DiffusionProfiles
    dp = DiffusionProfiles(None, None, None, None, None, str(tmp_path))
    for s, v in zip(gold["starts"], gold["profiles"]):
        dp.save_diffusion_profile(v, str(s))
    assert sorted(os.listdir(tmp_path)) == [str(f) for f in gold["file_names"]]
    dp.load_diffusion_profiles([str(s) for s in gold["starts"]])
    assert all((dp.drug_or_indication2diffusion_profile[str(s)] == v).all() for s, v in zip(gold["starts"], gold["profiles"]))


def test_product_path_has_no_cpu_fallback(gold):
    from gcn_drug_repurposing_amd import _lib
    from gcn_drug_repurposing_amd.diffusion import PprEngine, PprProblem
    with pytest.raises(_lib.GssError):
        PprEngine(PprProblem(gold["m0"], gold["start_idx"][:2], gold["prot"]), device="cpu")


# ---------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("case", ["msi_small", "edge"])
def test_gpu_profiles_match_reference_and_oracle(gold, case):
    from gcn_drug_repurposing_amd.diffusion import diffusion_profiles
    if case == "msi_small":
        m0, starts, prot, hp = gold["m0"], gold["start_idx"], gold["prot"], gold["hp"]
    else:
        (m0, prot), starts, hp = edge_case_graph(), np.array([0, 1, 2, 3]), (0.85, 500, 1e-9)
    x, it = diffusion_profiles(m0, starts, prot, hp[0], hp[1], hp[2])
    ref, ref_it = oracle_profiles(m0, starts, prot, hp)
    assert np.abs(x - ref).max() < 1e-13           # fp64; only the summation order differs
    assert (it == ref_it).all()
    if case == "msi_small":
        assert np.abs(x - gold["profiles"]).max() < 1e-13   # the reference's own output
    x2, _ = diffusion_profiles(m0, starts, prot, hp[0], hp[1], hp[2])
    assert (x == x2).all()                          # bitwise reproducible


@pytest.mark.gpu
def test_gpu_fp64_spmm_hubs_empty_rows_and_padding():
    import torch
    from gcn_drug_repurposing_amd.diffusion import PprEngine, PprProblem
    rng = np.random.RandomState(3)
    n = 700
    m = sp.random(n, n, density=0.01, random_state=rng, format="lil")
    m[5, :] = rng.rand(n)            # a hub row of M'^T is a column here: make both
    m[:, 9] = rng.rand(n, 1)
    m[17, :] = 0; m[:, 17] = 0       # isolated node
    m0 = sp.csr_matrix(m)
    prob = PprProblem(m0, np.array([1, 2, 3]), {1: [], 2: [], 3: []})
    eng = PprEngine(prob)
    assert prob.kpad == 64
    x = torch.from_numpy(rng.randn(n, prob.kpad)).cuda()
    y = torch.full((n, prob.kpad), float("nan"), dtype=torch.float64, device="cuda")
    eng.spmm(x, y)
    ref = prob.mt @ x.cpu().numpy()
    assert np.abs(y.cpu().numpy() - ref).max() < 1e-13 * max(1.0, np.abs(ref).max())
    y2 = torch.empty_like(y)
    eng.spmm(x, y2)
    assert torch.equal(y, y2)


@pytest.mark.gpu
def test_gpu_standin_scale_sample_against_oracle():
    """1/8-scale whole-graph stand-in: 312 start nodes in one batch (5 column chunks); every column is a probability
    vector, a sample of columns equals the oracle"""
    from gcn_drug_repurposing_amd import synth
    from gcn_drug_repurposing_amd.diffusion import diffusion_profiles
    adj, ntype, _ = synth.whole_graph_standin(seed=1, scale=8)
    m0 = sp.csr_matrix(adj, dtype=np.float64)
    di = np.flatnonzero(ntype <= 1)
    prot = {int(s): m0.indices[m0.indptr[s]:m0.indptr[s + 1]].tolist() for s in di}
    hp = (0.8595436247434408, 1000, 1e-6)
    x, it = diffusion_profiles(m0, di, prot, *hp)
    assert x.shape == (len(di), m0.shape[0]) and abs(x.sum(1) - 1).max() < 1e-10 and x.min() >= 0
    pick = di[:: max(1, len(di) // 12)]
    ref, ref_it = oracle_profiles(m0, pick, prot, hp)
    sel = np.searchsorted(di, pick)
    assert np.abs(x[sel] - ref).max() < 1e-13 and (it[sel] == ref_it).all()


@pytest.mark.gpu
def test_gpu_class_end_to_end_on_the_msi_tables(gold, tmp_path):
    from gcn_drug_repurposing_amd.diffusion import DiffusionProfiles
    from gcn_drug_repurposing_amd.msi import COVID_WEIGHTS, MsiGraph
    a, mi, tol = gold["hp"]
    dp = DiffusionProfiles(alpha=a, max_iter=mi, tol=tol, weights=COVID_WEIGHTS, num_cores=1, save_load_file_path=str(tmp_path / "dp"))
    dp.calculate_diffusion_profiles(MsiGraph().load(FILES))
    names = [str(s) for s in gold["starts"]]
    dp.load_diffusion_profiles(names)
    got = np.stack([dp.drug_or_indication2diffusion_profile[s] for s in names])
    assert np.abs(got - gold["profiles"]).max() < 1e-13
    assert os.path.exists(tmp_path / "dp" / "node2idx.pkl")


@pytest.mark.gpu
def test_gpu_ppr_workspace_guards_over_a_size_sweep():
    """VERDICT round 2, weak spot 9: the plan's buffers had guards, gss_ppr_*'s slab had none.  Random small problems over the sizes
    that decide its carve -- nodes around the 64-row block size, start nodes around the 64-column padding, with and without empty rows
    and surviving in-edges -- each checked against the oracle, each followed by gss_ppr_check_guards."""
    from gcn_drug_repurposing_amd.diffusion import PprEngine, PprProblem
    from oracle import diffusion_oracle as O
    rng = np.random.RandomState(17)
    hp = (0.85, 500, 1e-9)
    for n, n_start in ((5, 1), (63, 3), (64, 64), (65, 65), (129, 2), (200, 70), (333, 129)):
        n_start = min(n_start, n // 2)
        dens = min(1.0, 6.0 / n)
        a = sp.random(n, n, density=dens, random_state=rng, format="csr")
        a = sp.csr_matrix(a + a.T)
        a.setdiag(0)
        a.eliminate_zeros()
        m0 = sp.csr_matrix(a)
        m0.data = rng.rand(m0.nnz) + 0.1
        starts = np.arange(n_start)
        # start nodes touch only non-start nodes ('proteins'), as drugs / indications do in the MSI: drop edges among start nodes
        m0 = sp.lil_matrix(m0)
        m0[:n_start, :n_start] = 0
        m0 = sp.csr_matrix(m0)
        m0.eliminate_zeros()
        prot = {int(s): [int(c) for c in m0.indices[m0.indptr[s]:m0.indptr[s + 1]]] for s in starts}
        eng = PprEngine(PprProblem(m0, starts, prot))
        x, it = eng.run(*[hp[0], hp[2], hp[1]])
        eng.check_guards()
        got = x[:, :n_start].t().contiguous().cpu().numpy()
        for c in (0, n_start - 1):
            ref, ref_it = O.diffusion_profile(m0, int(starts[c]), prot, *hp)
            assert np.abs(got[c] - ref).max() < 1e-13 and it[c] == ref_it, (n, n_start, c)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["msi_small", "edge"])
def test_gpu_fused_update_equals_the_separate_update_pass(gold, case):
    """round 3: the update x <- alpha (y + dangling e_s) + (1 - alpha) e_s and the column L1 errors run in the SpMM's epilogue (double-buffered
    x, the start nodes' "selected" rows as a sparse addend, the in-place override scaling repaired afterwards).  Against the separate
    update pass (knob ppr_fused = 0): the same iteration counts, profiles equal to rounding (the selected row is added before instead
    of after the store of y; the column errors are summed in another order), guards intact."""
    import gcn_drug_repurposing_amd as pkg
    from gcn_drug_repurposing_amd.diffusion import PprEngine, PprProblem
    lib = pkg.load()
    if case == "msi_small":
        m0, starts, prot, hp = gold["m0"], gold["start_idx"], gold["prot"], gold["hp"]
    else:
        (m0, prot), starts, hp = edge_case_graph(), np.array([0, 1, 2, 3]), (0.85, 500, 1e-9)
    res = {}
    for fused in (1, 0):
        assert lib.gss_debug_set_option(b"ppr_fused", fused) == 0
        try:
            eng = PprEngine(PprProblem(m0, np.asarray(starts), prot))
            x, it = eng.run(hp[0], hp[2], hp[1])
            eng.check_guards()
            res[fused] = (x[:, :len(starts)].t().contiguous().cpu().numpy(), it.copy())
            x2, it2 = eng.run(hp[0], hp[2], hp[1])          # a second run on the same handle starts from scratch
            assert (x2[:, :len(starts)].t().cpu().numpy() == res[fused][0]).all() and (it2 == it).all()
        finally:
            lib.gss_debug_set_option(b"ppr_fused", 1)
    assert (res[1][1] == res[0][1]).all()
    assert np.abs(res[1][0] - res[0][0]).max() < 1e-15
    ref, ref_it = oracle_profiles(m0, starts, prot, hp)
    assert np.abs(res[1][0] - ref).max() < 1e-13 and (res[1][1] == ref_it).all()
