#!/usr/bin/env python3
"""How far the device's multi-step trajectories are from the reference's (GPU box only): the fixtures' recorded batches through the
single-GPU plan, deviations of losses / last embeddings / final weights from the reference-generated fixture, in the units the tests'
tolerances use (tests/tolerances.py), beside the spread between equally valid CPU runs that tests/test_trajectory_spread.py measures."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import golden_batches, golden_csr, golden_params, load_golden
from gcn_drug_repurposing_amd.engine import GssEngine
from gcn_drug_repurposing_amd.graph import GssGraph

for case in ("toy_sif_d64_L2", "knn_n200_d16_L2", "knn_n2000_d64_L3", "edge_n600_d128_L2"):
    g = load_golden(case)
    n, d, L = (int(v) for v in g["meta"])
    for lazy in (False, True):
        graph = GssGraph(golden_csr(g, "A"))
        params = [torch.from_numpy(g["init_" + k].copy()).cuda() for k in ("W1", "b1", "W2", "b2")]
        eng = GssEngine(graph, torch.from_numpy(g["X"]).cuda(), params, num_layers=L, layer_decay=float(g["decay"]), alpha=float(g["alpha"]),
                        lr=float(g["lr"]), max_batch=n)
        losses = []
        bs = golden_batches(g)
        for i, idx in enumerate(bs):
            t = torch.from_numpy(idx.astype(np.int32)).cuda()
            (eng.step_lazy if (lazy and i < len(bs) - 1) else eng.step)(t, float(g["beta"]))
            losses.append(eng.loss.item())
        emb = eng.emb.cpu().numpy().astype(np.float64)
        dl = np.abs(np.array(losses) - g["losses"]).max() / np.abs(g["losses"]).max()
        de = np.abs(emb - g["emb_last"]).max() / np.abs(g["emb_last"]).max()
        dw = max(np.abs(p.cpu().numpy().astype(np.float64) - g["final_" + k]).max() for k, p in zip(("W1", "b1", "W2", "b2"), params)) / float(g["lr"])
        print(f"{case:20s} lazy={int(lazy)} steps={len(bs)}: loss rel {dl:.2e}  emb rel {de:.2e}  weights {dw:.2e} lr", flush=True)
