#!/usr/bin/env python3
"""Round 6 experiment: at L >= 3 the weight gradient of a middle layer (fp32 MFMA, 81 us at config 3) does not feed the backward hops that
follow it (input gradient -> A_hat^T hops of the layer below, gather-bound) -- can it hide under them on a second stream?
Runs y = A_hat^T x (d = 256, the whole_graph + pathway stand-in) and gss_dense_bwd_weight alone and concurrently on two streams, with
the SpMM at its normal occupancy (two 1024-thread workgroups per CU: every wave slot) and capped to one workgroup per CU
(GSS_EXP_SPMM_LDS_KB=96 in the environment pads its LDS request), which leaves wave slots for the other kernel.
Result (profiles/r06_corun_wgrad_side_stream.txt): 2 % hidden at config 3, 22-26 % at config 2's sizes; the capped SpMM is 26-28 % slower.  The
environment switch was a one-off build and is not in the product.
usage: corun_bwd_config3.py [d] [workload]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import gcn_drug_repurposing_amd as pkg  # noqa: E402
from gcn_drug_repurposing_amd import _lib, synth  # noqa: E402
from gcn_drug_repurposing_amd.dist import local_comms  # noqa: E402
from gcn_drug_repurposing_amd.shards import ScipySource, build_shard  # noqa: E402

lib = pkg.load()
d = int(sys.argv[1]) if len(sys.argv) > 1 else 256
workload = sys.argv[2] if len(sys.argv) > 2 else "whole_graph_pathway"
adj, _, _ = synth.whole_graph_standin(1, pathway_edges=workload.endswith("pathway"))
g = build_shard(ScipySource(adj), local_comms(1)[0], need_transpose=True)
n = g.n
x, y = torch.randn(n, d, device="cuda"), torch.empty(n, d, device="cuda")
dp, ax, am = (torch.randn(n, d, device="cuda") for _ in range(3))
gw1, gw2, gb = torch.empty(d, d, device="cuda"), torch.empty(d, d, device="cuda"), torch.empty(d, device="cuda")
ws = torch.empty(lib.gss_wgrad_workspace_bytes(n, d), dtype=torch.uint8, device="cuda")
w1t, w2t = (torch.randn(d, d, device="cuda") * 0.05 for _ in range(2))
gax, gam = torch.empty(n, d, device="cuda"), torch.empty(n, d, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def spmm(st):
    _lib.check(lib.gss_spmm(g.at.handle, d, x.data_ptr(), y.data_ptr(), None, None, st.cuda_stream))


def wgrad(st):
    _lib.check(lib.gss_dense_bwd_weight(n, d, dp.data_ptr(), ax.data_ptr(), am.data_ptr(), None, gw1.data_ptr(), gw2.data_ptr(), gb.data_ptr(), 0,
                                        ws.data_ptr(), st.cuda_stream))


def dgrad(st):
    _lib.check(lib.gss_dense_bwd_input(n, d, dp.data_ptr(), w1t.data_ptr(), w2t.data_ptr(), None, gax.data_ptr(), gam.data_ptr(), st.cuda_stream))


def timed(fn, reps=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s1)
    s2.wait_event(e0)
    for _ in range(reps):
        fn()
    e2 = torch.cuda.Event()
    e2.record(s2)
    s1.wait_event(e2)
    e1.record(s1)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for _ in range(6000):
    spmm(s1)
torch.cuda.synchronize()
t_s = min(timed(lambda: spmm(s1)) for _ in range(3))
t_w = min(timed(lambda: wgrad(s1)) for _ in range(3))
t_g = min(timed(lambda: dgrad(s1)) for _ in range(3))
# what the backward pass would do: the chain [dgrad, SpMM, SpMM] on one stream, the weight gradient beside it on the other
t_chain = min(timed(lambda: (dgrad(s1), spmm(s1), spmm(s1))) for _ in range(3))
t_serial = min(timed(lambda: (wgrad(s1), dgrad(s1), spmm(s1), spmm(s1))) for _ in range(3))
t_both = min(timed(lambda: (wgrad(s2), dgrad(s1), spmm(s1), spmm(s1))) for _ in range(3))
t_pair = min(timed(lambda: (spmm(s1), wgrad(s2))) for _ in range(3))
print(f"{workload} d={d} GSS_EXP_SPMM_LDS_KB={os.environ.get('GSS_EXP_SPMM_LDS_KB', '0')}: SpMM {t_s:.1f} us, weight gradient (+ reduce) {t_w:.1f} us, input gradient {t_g:.1f} us; "
      f"SpMM + weight gradient on two streams {t_pair:.1f} us per pair (sum {t_s + t_w:.1f}); "
      f"[wgrad, dgrad, SpMM, SpMM] on one stream {t_serial:.1f} us, chain without wgrad {t_chain:.1f} us, wgrad on the second stream beside the chain {t_both:.1f} us "
      f"({(t_serial - t_both):.1f} us = {(t_serial - t_both) / t_w * 100:.0f} % of the weight gradient hidden)", flush=True)
