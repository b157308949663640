"""Where does the HIP step lose accuracy relative to fp64, next to the reference's fp32 torch-CPU path?
config 2 (whole_graph stand-in, d = 128, L = 2, B = 2048): per-activation and per-gradient max errors."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gcn_drug_repurposing_amd import synth  # noqa: E402
from gcn_drug_repurposing_amd.engine import GssEngine  # noqa: E402
from gcn_drug_repurposing_amd.graph import GssGraph  # noqa: E402
from oracle import gss_oracle as O  # noqa: E402
from oracle.torch_cpu_path import TorchCpuPath  # noqa: E402

adj, _, _ = synth.whole_graph_standin(seed=1)
n, d, L, B = adj.shape[0], 128, 2, 2048
X = synth.gaussian_features(n, d, seed=2)
np.random.seed(7)
p = O.init_layer_weights(d, 1e-2)
idx = np.random.RandomState(0).permutation(n)[:B]
beta = 0.25
a_hat, _ = O.preprocess_graph(adj)
a32 = O.to_fp32_csr(a_hat)
graph = GssGraph.from_normalized(a_hat)
params = [torch.from_numpy(p[k].copy()).cuda() for k in ("W1", "b1", "W2", "b2")]
eng = GssEngine(graph, torch.from_numpy(X).cuda(), params, num_layers=L, layer_decay=0.3, alpha=1.0, lr=3e-4, max_batch=B)
eng.forward()
eng.loss_backward(torch.from_numpy(idx.astype(np.int32)).cuda(), beta)
torch.cuda.synchronize()
emb64, cache = O.forward(X, a32, p, L, 0.3, dtype=np.float64)
g64 = O.backward(cache, O.loss_grad_emb(emb64, beta, idx, 1.0))
cpu = TorchCpuPath(a32, X, p, L, 0.3, 1.0, 3e-4)
e_cpu = cpu.forward()
l_cpu = cpu.loss(e_cpu, beta, idx.astype(np.int64))
cpu.opt.zero_grad()
l_cpu.backward()
print("cache keys", [k for k in cache.keys()] if hasattr(cache, "keys") else type(cache))
print(f"emb: hip-f64 {np.abs(eng.emb.cpu().numpy() - emb64).max():.3e}  cpu-f64 {np.abs(e_cpu.detach().numpy() - emb64).max():.3e}")
for k, g in zip(("W1", "b1", "W2", "b2"), eng.grads):
    s = np.abs(g64[k]).max()
    print(f"grad {k}: scale {s:.3e}  hip-f64 {np.abs(g.cpu().numpy() - g64[k]).max():.3e}  cpu-f64 {np.abs(cpu.p[k].grad.numpy() - g64[k]).max():.3e}  "
          f"hip-cpu {np.abs(g.cpu().numpy() - cpu.p[k].grad.numpy()).max():.3e}")
# the same with the fp64 oracle fed by fp32-rounded forward activations is not available; instead repeat HIP with the step path
eng2 = GssEngine(graph, torch.from_numpy(X).cuda(), [torch.from_numpy(p[k].copy()).cuda() for k in ("W1", "b1", "W2", "b2")], num_layers=L,
                 layer_decay=0.3, alpha=1.0, lr=3e-4, max_batch=B)
lib = eng.lib
from gcn_drug_repurposing_amd import _lib  # noqa: E402
for var in (1, 2):
    _lib.check(lib.gss_debug_set_option(b"spmm_variant", var))
    eng2.forward()
    eng2.loss_backward(torch.from_numpy(idx.astype(np.int32)).cuda(), beta)
    torch.cuda.synchronize()
    print(f"spmm_variant {var}: " + "  ".join(f"{k} {np.abs(g.cpu().numpy() - g64[k]).max():.3e}" for k, g in zip(("W1", "b1", "W2", "b2"), eng2.grads)))

# ---- isolate the loss kernel: same fp32 embeddings in, fp64 oracle on exactly those embeddings
import ctypes as C  # noqa: E402
emb_hip = eng.emb.clone()
idx32 = torch.from_numpy(idx.astype(np.int32)).cuda()
loss_t, de = torch.empty(1, device="cuda"), torch.empty(B, d, device="cuda")
ws = torch.empty(lib.gss_loss_workspace_bytes(B, d), dtype=torch.uint8, device="cuda")
_lib.check(lib.gss_loss_fwd_bwd(n, d, emb_hip.data_ptr(), idx32.data_ptr(), B, beta, 1.0, loss_t.data_ptr(), de.data_ptr(), ws.data_ptr(),
                                _lib.current_stream()))
torch.cuda.synchronize()
e_h = emb_hip.cpu().numpy().astype(np.float64)
de_ref_same = O.loss_grad_emb(e_h, beta, idx, 1.0)[idx]
de_ref_64 = O.loss_grad_emb(emb64, beta, idx, 1.0)[idx]
de_cpu_in = O.loss_grad_emb(e_cpu.detach().numpy().astype(np.float64), beta, idx, 1.0)[idx]
print(f"dE_B scale {np.abs(de_ref_64).max():.3e}: hip kernel vs f64 on the SAME fp32 emb {np.abs(de.cpu().numpy() - de_ref_same).max():.3e};  "
      f"f64(emb_hip) vs f64(emb64) {np.abs(de_ref_same - de_ref_64).max():.3e};  f64(emb_cpu) vs f64(emb64) {np.abs(de_cpu_in - de_ref_64).max():.3e}")
S64 = emb64[idx] @ emb64[idx].T
Sh = e_h[idx] @ e_h[idx].T
Sc = e_cpu.detach().numpy().astype(np.float64)[idx]
Sc = Sc @ Sc.T
print("entries with |S64| < 1e-7:", int((np.abs(S64) < 1e-7).sum()), " sign flips hip-emb vs f64:", int(((Sh > 0) != (S64 > 0)).sum()),
      " cpu-emb vs f64:", int(((Sc > 0) != (S64 > 0)).sum()))
# sign flips inside the fp32 product itself
S32 = (emb_hip[idx32.long()] @ emb_hip[idx32.long()].T).cpu().numpy()
print("sign flips of a fp32 GEMM on emb_hip vs its fp64 product:", int(((S32 > 0) != (Sh > 0)).sum()))
