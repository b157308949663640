#!/usr/bin/env python3
"""Where a projection launch spends its time, wave by wave (GPU box only): 100 MHz wall-clock stamps at a wave's start, at the start
and the end of its K loop and after its last store has left, with the workgroup id and the hardware slot (gss_debug_set_stamp_buffer).
Prints, for the workgroups that were dispatched first (one per CU) and the ones that doubled up: when they started, how long prologue,
loop and epilogue took, when they finished -- relative to the first wave's start, in microseconds; and how many workgroups each CU
hosted.  usage: gemm_stamps.py [n] [d]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gcn_drug_repurposing_amd as pkg
from gcn_drug_repurposing_amd import _lib
lib = pkg.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 29960
d = int(sys.argv[2]) if len(sys.argv) > 2 else 128
for kv in sys.argv[3:]:          # knob=value ...
    k, v = kv.split("=")
    assert lib.gss_debug_set_option(k.encode(), int(v)) == 0, kv
print("knobs:", sys.argv[3:])
ax, am, pp = (torch.randn(n, d, device="cuda") for _ in range(3))
w1, w2 = (torch.randn(d, d, device="cuda") * 0.05 for _ in range(2))
b1, b2 = (torch.randn(d, device="cuda") for _ in range(2))
p = torch.empty(n, d, device="cuda"); xn = torch.empty(n, d, device="cuda")
st = _lib.current_stream()
call = lambda: lib.gss_dense_fwd(n, d, ax.data_ptr(), am.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(),
                                 pp.data_ptr(), 0.3, p.data_ptr(), xn.data_ptr(), st)
for _ in range(300): call()
torch.cuda.synchronize()
nwg = (n + 63) // 64 if d < 256 else (n + 127) // 128
buf = torch.zeros(nwg * 4 * 6 + 64, dtype=torch.int64, device="cuda")
for rep in range(2):
    buf.zero_()
    for _ in range(5): call()
    lib.gss_debug_set_stamp_buffer(buf.data_ptr())
    call()
    lib.gss_debug_set_stamp_buffer(None)
    torch.cuda.synchronize()
    z = buf[:nwg * 4 * 6].cpu().numpy().reshape(nwg * 4, 6).astype(np.int64)
    z = z[z[:, 0] > 0]
    t = (z[:, :4] - z[:, 0].min()) / 100.0          # us
    lin, hw = z[:, 4], z[:, 5]
    cu_id = (hw >> 8) & 0xff | ((hw >> 13) & 0x7) << 8 | ((hw >> 12) & 1) << 11    # cu + se + sh bits of HW_ID (per XCD)
    print(f"run {rep}: {len(z)} waves of {nwg} workgroups; launch spans {t[:, 3].max():.2f} us from the first wave's start to the last store")
    for name, m in (("first 256 workgroups", lin < 256), ("workgroups 256..", lin >= 256)):
        if not m.any(): continue
        q = lambda a: "min %.2f / median %.2f / p90 %.2f / max %.2f" % (a.min(), np.median(a), np.percentile(a, 90), a.max())
        print(f"  {name}: start {q(t[m, 0])}")
        print(f"      prologue (first DMA round trip) {q(t[m, 1] - t[m, 0])}")
        print(f"      K loop {q(t[m, 2] - t[m, 1])}")
        print(f"      epilogue until the last store left {q(t[m, 3] - t[m, 2])}")
        print(f"      finished at {q(t[m, 3])}")
