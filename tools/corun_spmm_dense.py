#!/usr/bin/env python3
"""Can the dense projection's MFMA work hide under the SpMM's gathers?  Runs the plain forward SpMM (AM = A_hat M) and the
projection kernel (gss_dense_fwd) of config 2 alone and CONCURRENTLY on two HIP streams (independent buffers): if the pair takes
about the sum of the two, the units they contend for (L2 bandwidth, wave slots) leave nothing for a fused kernel to win beyond the
15 MB AM read it would save."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import gcn_drug_repurposing_amd as pkg  # noqa: E402
from gcn_drug_repurposing_amd import _lib, synth  # noqa: E402
from gcn_drug_repurposing_amd.graph import GssGraph  # noqa: E402

lib = pkg.load()
adj, _, _ = synth.whole_graph_standin(1)
g = GssGraph(adj, need_transpose=False)
n, d = g.n, 128
x, y = torch.randn(n, d, device="cuda"), torch.empty(n, d, device="cuda")
ax, am, pp = (torch.randn(n, d, device="cuda") for _ in range(3))
w1, w2 = (torch.randn(d, d, device="cuda") * 0.05 for _ in range(2))
b1, b2 = (torch.zeros(d, device="cuda") for _ in range(2))
p, xn = torch.empty(n, d, device="cuda"), torch.empty(n, d, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def spmm(st):
    _lib.check(lib.gss_spmm(g.a.handle, d, x.data_ptr(), y.data_ptr(), None, None, st.cuda_stream))


def dense(st):
    _lib.check(lib.gss_dense_fwd(n, d, ax.data_ptr(), am.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), pp.data_ptr(), 0.3,
                                 p.data_ptr(), xn.data_ptr(), st.cuda_stream))


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


for _ in range(15000):            # ~0.5 s of work first: the clocks ramp up slowly, a cold measurement reads 2x
    spmm(s1)
torch.cuda.synchronize()
t_s = min(timed(lambda: spmm(s1)) for _ in range(3))
t_d = min(timed(lambda: dense(s1)) for _ in range(3))
t_both = min(timed(lambda: (spmm(s1), dense(s2))) for _ in range(3))
print(f"plain SpMM alone {t_s:.1f} us, projection alone {t_d:.1f} us, sum {t_s + t_d:.1f} us; both at once on two streams {t_both:.1f} us per pair "
      f"({(t_s + t_d - t_both) / (t_s + t_d) * 100:.0f} % of the sum hidden)")

# the same SpMM on torch's default stream (what the plan uses) and on a side stream, for reference
def on_stream(st, reps=200):
    with torch.cuda.stream(st):
        for _ in range(20):
            spmm(st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(reps):
            spmm(st)
        e1.record(st)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3


print(f"SpMM back to back: default stream {on_stream(torch.cuda.default_stream()):.1f} us, side stream {on_stream(s1):.1f} us, "
      f"another side stream {on_stream(torch.cuda.Stream()):.1f} us")
