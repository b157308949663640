#!/usr/bin/env python3
"""gss_spmm with feature slices time-separated (grid.y) vs pinned to XCDs, per slice count (GPU box only)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gcn_drug_repurposing_amd as pkg
from gcn_drug_repurposing_amd import _lib, synth
from gcn_drug_repurposing_amd.graph import GssGraph

lib = pkg.load()
which = sys.argv[1] if len(sys.argv) > 1 else "whole"
if which == "whole":
    adj = synth.whole_graph_standin(1)[0]
elif which == "knn":
    from gcn_drug_repurposing_amd.graph import knn_descriptor_adj_device
    adj = knn_descriptor_adj_device(synth.gaussian_features(29960, 128, 2).astype(np.float64), 5)
else:
    adj = synth.rmat_adj(int(sys.argv[2]), int(sys.argv[3]))
st = _lib.current_stream()
PLAIN = os.environ.get('PLAIN', '1') == '1'
if True:
  g = GssGraph(adj)
  n, nnz = g.n, g.nnz
  print(f"graph {which}: N={n} nnz={nnz}")
  for d in [int(v) for v in (sys.argv[4:] or ["128"])]:
      torch.manual_seed(0)
      x = torch.randn(n, d, device="cuda"); h = torch.randn(n, d, device="cuda")
      y = torch.empty(n, d, device="cuda"); m = torch.empty(n, d, device="cuda")
      ref = None
      for pin in (0, 1):
          for ns in (1, 2, 4, 8):
              if pin and ns == 1: continue
              if d // 4 % ns or d // 4 // ns < 4: continue
              lib.gss_debug_set_option(b"spmm_pin", pin); lib.gss_debug_set_option(b"spmm_slices", ns)
              def go(): _lib.check(lib.gss_spmm(g.a.handle, d, x.data_ptr(), y.data_ptr(), None if PLAIN else h.data_ptr(), None if PLAIN else m.data_ptr(), st))
              for _ in range(5): go()
              torch.cuda.synchronize()
              if ref is None: ref = (y.clone(), m.clone())
              same = bool(torch.equal(ref[0], y) and torch.equal(ref[1], m))
              ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
              ev0.record()
              for _ in range(50): go()
              ev1.record(); torch.cuda.synchronize()
              us = ev0.elapsed_time(ev1) / 50 * 1e3
              print(f"d={d:4d} slices={ns} {'pinned' if pin else 'time  '}: {us:8.1f} us  gather {nnz * d * 4 / us / 1e6:6.2f} TB/s  bitwise_same={same}", flush=True)
lib.gss_debug_set_option(b"spmm_pin", 0); lib.gss_debug_set_option(b"spmm_slices", 0)
