#!/usr/bin/env python3
"""run N launches of one SpMM form for rocprofv3
usage: spmm_prof.py <variant> <d> [reps] [workload] [mode]
  workload: whole_graph (default) | whole_graph_pathway | rmat:<nodes>:<edges>
  mode:     plain (y = A x, default) | fwd1 (y = A x, m = y (.) x: the forward SpMM with the fused Hadamard epilogue)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import gcn_drug_repurposing_amd as pkg  # noqa: E402
from gcn_drug_repurposing_amd import _lib, synth  # noqa: E402

lib = pkg.load()
variant, d = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
workload = sys.argv[4] if len(sys.argv) > 4 else "whole_graph"
mode = sys.argv[5] if len(sys.argv) > 5 else "plain"
from gcn_drug_repurposing_amd.dist import local_comms
from gcn_drug_repurposing_amd.shards import RmatSource, ScipySource, build_shard
relabel = os.environ.get("GSS_RELABEL", "auto")
relabel = {"0": False, "1": True}.get(relabel, "auto")
if workload.startswith("rmat"):
    _, n, m = workload.split(":")
    g = build_shard(RmatSource(int(n), int(m), seed=4), local_comms(1)[0], need_transpose=False, relabel=relabel)
else:
    adj, _, _ = synth.whole_graph_standin(1, pathway_edges=workload.endswith("pathway"))
    g = build_shard(ScipySource(adj), local_comms(1)[0], need_transpose=False, relabel=relabel)   # what bench.py times
lib.gss_debug_set_option(b"spmm_variant", variant)
x = torch.randn(g.n, d, device="cuda")
y = torch.empty(g.n, d, device="cuda")
m_out = torch.empty(g.n, d, device="cuda") if mode == "fwd1" else None
for _ in range(reps):
    _lib.check(lib.gss_spmm(g.a.handle, d, x.data_ptr(), y.data_ptr(), x.data_ptr() if m_out is not None else None, _lib.ptr(m_out),
                            _lib.current_stream()))
torch.cuda.synchronize()
print(f"{workload} {mode}: n={g.n} nnz={g.a.nnz} d={d} reps={reps}")
