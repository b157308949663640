import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import gcn_drug_repurposing_amd as pkg
from gcn_drug_repurposing_amd import _lib
lib = pkg.load()
n, d = 2048, 128
dp = torch.randn(n, d, device="cuda"); w1t = torch.randn(d, d, device="cuda"); w2t = torch.randn(d, d, device="cuda")
gax = torch.empty(n, d, device="cuda"); gam = torch.empty(n, d, device="cuda")
st = _lib.current_stream()
x = torch.randn(4096, 4096, device="cuda")
for _ in range(200): x @ x   # spin-up
ref = None
for nt in (0, 8, 4, 2, 1):
    lib.gss_debug_set_option(b"gemm_small_nt", nt)
    for _ in range(50): lib.gss_dense_bwd_input(n, d, dp.data_ptr(), w1t.data_ptr(), w2t.data_ptr(), None, gax.data_ptr(), gam.data_ptr(), st)
    torch.cuda.synchronize()
    if ref is None: ref = (gax.clone(), gam.clone())
    same = torch.equal(ref[0], gax) and torch.equal(ref[1], gam)
    best = 1e9
    for r in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(300): lib.gss_dense_bwd_input(n, d, dp.data_ptr(), w1t.data_ptr(), w2t.data_ptr(), None, gax.data_ptr(), gam.data_ptr(), st)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 300 * 1e3)
    print(f"gemm_small_nt={nt}: {best:.2f} us  bitwise_same={same}")
