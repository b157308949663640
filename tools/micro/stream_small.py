#!/usr/bin/env python3
"""what does the memory system give a plain elementwise kernel at the projection's footprint (N = 29,960, d = 128:
three 15-MB inputs, two 15-MB outputs)?  torch elementwise kernels as the yardstick (GPU box only)"""
import torch
n, d = 29960, 128
ax, am, pp = (torch.randn(n, d, device="cuda") for _ in range(3))
p, xn = torch.empty(n, d, device="cuda"), torch.empty(n, d, device="cuda")
def t(fn, reps=50):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
mb = n * d * 4 / 1e6
us = t(lambda: torch.add(ax, am, out=p)); print(f"p = ax + am            (2 in, 1 out, {3*mb:.0f} MB): {us:5.1f} us  {3*mb/us/1e3*1e3:6.0f} GB/s")
us = t(lambda: p.copy_(ax)); print(f"p = ax                 (1 in, 1 out, {2*mb:.0f} MB): {us:5.1f} us  {2*mb/us/1e3*1e3:6.0f} GB/s")
def five():
    torch.add(ax, am, out=p); torch.add(p, pp, out=xn)
us = t(five); print(f"p = ax + am; xn = p+pp (two launches, 4 in 2 out, {6*mb:.0f} MB): {us:5.1f} us  {6*mb/us/1e3*1e3:6.0f} GB/s")
us = t(lambda: p.fill_(1.0)); print(f"p = 1                  (1 out, {mb:.0f} MB): {us:5.1f} us  {mb/us/1e3*1e3:6.0f} GB/s")
