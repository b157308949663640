#!/usr/bin/env python3
"""Golden fixture for the diffusion-profile baseline: run the REFERENCE's DiffusionProfiles
(multiscale/diff_prof/diffusion_profiles.py) on the small MSI tables of tests/golden/msi_small/ and record, for every
drug and indication, the converged personalised-PageRank vector.  Run in the build container only:

    python tests/golden/make_diffusion_fixture.py

The reference targets networkx 2.x / scipy < 1.8: `nx.to_scipy_sparse_matrix` and the numpy aliases `scipy.array`,
`scipy.repeat`, `scipy.where`, `scipy.absolute` it calls are gone from the installed libraries, so this script puts
those library names back (thin aliases of the functions that replaced them) before importing it.  No reference code is
changed or copied; the output is data only."""
import os
import sys
import tempfile

import numpy as np

REF = os.environ.get("GSS_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
SMALL = os.path.join(HERE, "msi_small")
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(REF, "multiscale"))
sys.path.insert(0, REF)

ALPHA, MAX_ITER, TOL = 0.8595436247434408, 1000, 1e-06                     # evaluate_auc.py:80-83
WEIGHTS = {'down_functional_pathway': 4.4863053901688685, 'indication': 3.541889556309463,
           'functional_pathway': 6.583155399238509, 'up_functional_pathway': 2.09685000906964,
           'protein': 4.396695660380823, 'drug': 3.2071696595616364}        # evaluate_auc.py:84-91


def main():
    import networkx as nx
    import scipy
    import scipy.sparse as sp
    if not hasattr(nx, "to_scipy_sparse_matrix"):   # networkx < 3 API: a csr_matrix ('*' is the matrix product)
        nx.to_scipy_sparse_matrix = lambda g, nodelist=None, weight="weight", dtype=None: sp.csr_matrix(
            nx.to_scipy_sparse_array(g, nodelist=nodelist, weight=weight, dtype=dtype, format="csr"))
    for name in ("array", "repeat", "where", "absolute"):   # scipy < 1.8 re-exported numpy's
        if not hasattr(scipy, name):
            setattr(scipy, name, getattr(np, name))
    from msi.msi import MSI                                      # the reference
    from diff_prof.diffusion_profiles import DiffusionProfiles   # the reference

    p = lambda n: os.path.join(SMALL, n + ".tsv")  # noqa: E731
    msi = MSI(drug2protein_file_path=p("drug_to_protein"), indication2protein_file_path=p("indication_to_protein"),
              protein2protein_file_path=p("protein_to_protein"), protein2functional_pathway_file_path=p("protein_to_functional_pathway"),
              functional_pathway2functional_pathway_file_path=p("functional_pathway_to_functional_pathway"))
    msi.load()
    with tempfile.TemporaryDirectory() as tmp:
        dp = DiffusionProfiles(alpha=ALPHA, max_iter=MAX_ITER, tol=TOL, weights=WEIGHTS, num_cores=1, save_load_file_path=tmp)
        # calculate_diffusion_profiles (diffusion_profiles.py:125-156) without its process pool
        msi.weight_graph(dp.weights)
        dp.get_initial_M(msi)
        starts = sorted(msi.drugs_in_graph + msi.indications_in_graph)
        for s in starts:
            dp.calculate_diffusion_profile(msi, [s])
        dp.load_diffusion_profiles(starts)
        prof = np.stack([dp.drug_or_indication2diffusion_profile[s] for s in starts])
        files = sorted(os.listdir(tmp))
    m0 = dp.initial_M.tocsr()
    m0.sort_indices()
    out = os.path.join(HERE, "diffusion_msi_small.npz")
    np.savez_compressed(out, nodelist=np.array(msi.nodelist), starts=np.array(starts), profiles=prof,
                        m_indptr=m0.indptr, m_indices=m0.indices, m_data=m0.data,
                        proteins_of=np.array([" ".join(sorted(msi.drug_or_indication2proteins[s])) for s in starts]),
                        alpha=ALPHA, tol=TOL, max_iter=MAX_ITER, file_names=np.array(files))
    print("nodes", len(msi.nodelist), "start nodes", len(starts), "profile sums", prof.sum(1).min(), prof.sum(1).max(), "->", out)


if __name__ == "__main__":
    main()
