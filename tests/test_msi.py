"""CPU: the TSV -> CSR multiscale-interactome loader against a fixture produced by the
reference's own MSI class + networkx (tests/golden/make_msi_fixture.py)."""
import os

import numpy as np

from conftest import GOLDEN

D = os.path.join(GOLDEN, "msi_small")
FILES = {k: os.path.join(D, k + ".tsv") for k in ("drug_to_protein", "indication_to_protein", "protein_to_protein",
                                                  "protein_to_functional_pathway", "functional_pathway_to_functional_pathway")}


def test_node_order_types_and_edgelist_text_match_reference(tmp_path):
    from gcn_drug_repurposing_amd.msi import MsiGraph
    g = MsiGraph().load(FILES).weight_graph()
    exp_nodes = [l.rstrip("\n").split("\t") for l in open(os.path.join(D, "expected_nodes.txt"))]
    assert g.names == [n for n, _ in exp_nodes]
    assert [g.type[n] for n in g.names] == [t for _, t in exp_nodes]
    out = tmp_path / "g.edgelist"
    g.write_weighted_edgelist(str(out))
    assert out.read_text() == open(os.path.join(D, "expected.weighted.edgelist")).read()   # byte identical
    ids = [l.strip() for l in open(os.path.join(D, "pathway_ids.txt")) if l.strip()]
    g.add_covid_pathway_edges(ids)
    g.write_weighted_edgelist(str(out))
    got = sorted(out.read_text().splitlines())
    exp = sorted(open(os.path.join(D, "expected_pathway.weighted.edgelist")).read().splitlines())
    assert got == exp      # same edges and weights (set(ids) iteration order may differ between processes)


def test_csr_roundtrip_through_the_trainer_reader(tmp_path):
    from gcn_drug_repurposing_amd import embio
    from gcn_drug_repurposing_amd.graph import edgelist_adj
    from gcn_drug_repurposing_amd.msi import MsiGraph
    g = MsiGraph().load(FILES).weight_graph()
    adj, names, types = g.to_csr()
    assert adj.shape == (111, 111) and adj.nnz == 672   # 678 once the 3 COVID-pathway edge pairs are added
    # outgoing weights towards one class sum to that class's weight (msi.py:255-262)
    row = names.index("NodeCovid")
    assert abs(adj[row].sum() - 4.396695660380823) < 1e-12          # an indication only points at proteins
    p = tmp_path / "g.edgelist"
    g.write_weighted_edgelist(str(p))
    src, dst, w, _ = embio.read_edgelist(str(p), names)
    adj2 = edgelist_adj(src, dst, w, len(names))
    assert abs(adj - adj2).max() < 1e-15
    # asymmetric weights: this is why the backward pass needs A_hat^T
    assert abs(adj - adj.T).max() > 0.1


def test_whole_graph_standin_equals_the_msi_loader_on_its_own_tables(tmp_path):
    """the stand-in = the four real MSI layers + a synthetic PPI layer.  synth builds it vectorised;
    here the same five tables go through the MsiGraph loader (the networkx-order restatement checked above against
    the reference's MSI class): node order, node types and every edge weight must agree, at full size."""
    from gcn_drug_repurposing_amd import synth
    from gcn_drug_repurposing_amd.msi import MsiGraph
    tables = synth.standin_tables(seed=1)
    files = {}
    for name, rows in tables.items():
        files[name] = str(tmp_path / (name + ".tsv"))
        with open(files[name], "w") as f:
            f.write("node_1\tnode_2\n")
            f.writelines(f"{u}\t{v}\n" for u, v in rows)
    g = MsiGraph().load(files).weight_graph()
    adj_ref, names_ref, types_ref = g.to_csr()
    adj, ntype, names = synth.whole_graph_standin(seed=1)
    assert adj.shape == (29960, 29960) and adj.nnz == 958068          # 479,034 undirected edges
    assert names == names_ref
    code = {"drug": 0, "indication": 1, "protein": 2, "functional_pathway": 3}
    assert [code[t] for t in types_ref] == ntype.tolist()
    assert np.bincount(ntype).tolist() == [1661, 841, 17660, 9798]    # 840 indications + NodeCovid
    assert (adj != adj_ref).nnz == 0 or abs(adj - adj_ref).max() < 1e-15
    # config 3: + 324 NodeCovid <-> pathway edge pairs of weight 3/353 (predict_drug.py:182-196)
    R = synth.real_layers()
    g.add_covid_pathway_edges([R["names_pathway"][i] for i in R["covid_pathway_idx"]] + [f"absent{i}" for i in range(353 - 324)])
    adj3_ref, _, _ = g.to_csr()
    adj3, _, _ = synth.whole_graph_standin(seed=1, pathway_edges=True)
    assert adj3.nnz == 958068 + 2 * 324 and abs(adj3 - adj3_ref).max() < 1e-15
    covid = names.index("NodeCovid")
    assert abs(adj3[covid].sum() - (4.396695660380823 + 324 * 3.0 / 353.0)) < 1e-12


def test_real_layer_fixture_is_the_reference_tables_verbatim():
    """build container only: data/msi_real_layers.npz holds exactly the node_1/node_2 columns of the shipped tables"""
    import csv
    import pytest
    ref = "/root/reference/data"
    if not os.path.isdir(ref):
        pytest.skip("reference tree not present (GPU box)")
    from gcn_drug_repurposing_amd import synth
    R = synth.real_layers()
    kinds = {"drug_to_protein": ("drug", "protein"), "indication_to_protein": ("indication", "protein"),
             "covid_to_protein": ("indication", "protein"), "protein_to_functional_pathway": ("protein", "pathway"),
             "functional_pathway_to_functional_pathway": ("pathway", "pathway")}
    for stem, (t1, t2) in kinds.items():
        with open(os.path.join(ref, stem + ".tsv"), newline="") as f:
            rows = list(csv.reader(f, delimiter="\t"))[1:]
        got = [(R["names_" + t1][a], R["names_" + t2][b]) for a, b in zip(R[stem + "_u"], R[stem + "_v"])]
        assert got == [(r[0], r[1]) for r in rows], stem
    assert int(R["covid_pathway_total"]) == 353 and len(R["covid_pathway_idx"]) == 324
    with open(os.path.join(ref, "drug_indication_df.tsv"), newline="") as f:
        rows = list(csv.reader(f, delimiter="\t"))[1:]
    want = {}
    for r in rows:
        want.setdefault(r[2], set()).add(r[0])
    assert synth.standin_drug_indications() == want and sum(len(v) for v in want.values()) == 5926
