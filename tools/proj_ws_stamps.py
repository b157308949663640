#!/usr/bin/env python3
"""Round 5: where a launch of the weight-stationary persistent projection spends its time, wave by wave (GPU box only).  100 MHz
wall-clock stamps (gss_debug_set_stamp_buffer; 24 x 8 bytes per wave): start, weights + first tiles in, then per tile the end of its
MFMAs and the issue of its stores.  usage: proj_ws_stamps.py [n] [knob=value ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gcn_drug_repurposing_amd as pkg
from gcn_drug_repurposing_amd import _lib
lib = pkg.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 29960
d = 128
wgs = 512
assert lib.gss_debug_set_option(b"gemm_ws", 1) == 0      # the stamps below are the weight-stationary kernel's (by default it runs from 131,072 rows on)
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    assert lib.gss_debug_set_option(k.encode(), int(v)) == 0, kv
print("knobs:", sys.argv[2:])
ax, am, pp = (torch.randn(n, d, device="cuda") for _ in range(3))
w1, w2 = (torch.randn(d, d, device="cuda") * 0.05 for _ in range(2))
b1, b2 = (torch.randn(d, device="cuda") for _ in range(2))
p = torch.empty(n, d, device="cuda"); xn = torch.empty(n, d, device="cuda")
st = _lib.current_stream()
call = lambda: lib.gss_dense_fwd(n, d, ax.data_ptr(), am.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(),
                                 pp.data_ptr(), 0.3, p.data_ptr(), xn.data_ptr(), st)
for _ in range(300): call()
torch.cuda.synchronize()
nwg = min(wgs, (n + 15) // 16)
buf = torch.zeros(nwg * 4 * 24 + 64, dtype=torch.int64, device="cuda")
for rep in range(2):
    buf.zero_()
    for _ in range(5): call()
    lib.gss_debug_set_stamp_buffer(buf.data_ptr())
    call()
    lib.gss_debug_set_stamp_buffer(None)
    torch.cuda.synchronize()
    z = buf[:nwg * 4 * 24].cpu().numpy().reshape(nwg * 4, 24).astype(np.int64)
    z = z[z[:, 0] > 0]
    t0 = z[:, 0].min()
    lin, tiles = z[:, 20], z[:, 22]
    us = lambda a: (a - t0) / 100.0
    q = lambda a: "min %.2f / median %.2f / p90 %.2f / max %.2f" % (a.min(), np.median(a), np.percentile(a, 90), a.max())
    last = np.array([z[k, 3 + 2 * min(tiles[k], 9) - 2] for k in range(len(z))])
    print(f"run {rep}: {len(z)} waves of {nwg} workgroups; the last store is issued {us(last).max():.2f} us after the first wave's start")
    for name, m in (("workgroups 0..255", lin < 256), ("workgroups 256..", lin >= 256)):
        if not m.any(): continue
        print(f"  {name} ({int(m.sum()) // 4} workgroups, tiles per workgroup {sorted(set(tiles[m].tolist()))[:6]}):")
        print(f"      start                          {q(us(z[m, 0]))}")
        print(f"      L2 warm-up round trip done after {q((z[m, 19] - z[m, 0]) / 100.0)}")
        print(f"      weights + first tiles in after {q((z[m, 1] - z[m, 0]) / 100.0)}")
        for i in range(int(tiles[m].max())):
            mm = m & (tiles > i)
            if i >= 9 or not mm.any(): break
            prev = z[mm, 1] if i == 0 else z[mm, 3 + 2 * (i - 1)]
            print(f"      tile {i}: MFMAs {q((z[mm, 2 + 2 * i] - prev) / 100.0)}; epilogue {q((z[mm, 3 + 2 * i] - z[mm, 2 + 2 * i]) / 100.0)}")
        print(f"      last store issued at           {q(us(last[m]))}")
