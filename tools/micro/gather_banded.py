#!/usr/bin/env python3
"""does time-aligned column order help the SpMM gather?  Every workgroup owns a band of rows and walks that band's
entries in column order; all workgroups start together, so at any moment the chip reads one narrow range of the table.
Gathers only (no reduction) -- the ceiling for a band-synchronous SpMM.  Tools only, not shipped."""
import ctypes as C, os, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from gcn_drug_repurposing_amd import synth
import scipy.sparse as sp
SRC = r'''
#include <hip/hip_runtime.h>
template <int UNROLL, int THREADS>
__global__ __launch_bounds__(THREADS) void gather_band(const int* __restrict__ col, const long* __restrict__ ptr, const float* __restrict__ x,
                                                       float* out, int d4) {
  const int lane = threadIdx.x & 63, half = lane >> 5, li = lane & 31;
  const int wave = threadIdx.x >> 6, nw = THREADS / 64;
  const long lo = ptr[blockIdx.x], hi = ptr[blockIdx.x + 1];
  float4 acc = make_float4(0, 0, 0, 0);
  for (long base = lo + wave * 64; base < hi; base += nw * 64) {
    const long e = base + lane;
    const int c = e < hi ? col[e] : col[hi - 1];
    for (int t = 0; t < 64; t += 2 * UNROLL) {
      float4 v[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        const int cc = __shfl(c, t + 2 * u + half, 64);
        v[u] = *reinterpret_cast<const float4*>(x + ((size_t)cc * d4 + li) * 4);
      }
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
  }
  if (acc.x == 12345.f) out[threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}
extern "C" void run(const int* col, const long* ptr, const float* x, float* out, int d4, int bands, int threads, void* st) {
  if (threads == 1024) gather_band<4, 1024><<<bands, 1024, 0, (hipStream_t)st>>>(col, ptr, x, out, d4);
  else if (threads == 512) gather_band<4, 512><<<bands, 512, 0, (hipStream_t)st>>>(col, ptr, x, out, d4);
  else gather_band<4, 256><<<bands, 256, 0, (hipStream_t)st>>>(col, ptr, x, out, d4);
}
'''
tmp = tempfile.mkdtemp()
open(os.path.join(tmp, "g.hip"), "w").write(SRC)
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", os.path.join(tmp, "g.hip"), "-o", os.path.join(tmp, "g.so")])
lib = C.CDLL(os.path.join(tmp, "g.so"))
lib.run.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
adj, _, _ = synth.whole_graph_standin(1)
a = (adj + sp.eye(adj.shape[0], format="csr")).tocsr(); a.sort_indices()
n, nnz, d = a.shape[0], a.nnz, 128
x = torch.randn(n, d, device="cuda"); out = torch.zeros(1024, device="cuda")
rows = np.repeat(np.arange(n), np.diff(a.indptr))
st = torch.cuda.current_stream().cuda_stream
def timeit(fn):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20 * 1e3
for bands in (128, 256, 512, 1024, 2048):
    # nnz-balanced bands of whole rows
    cuts = np.searchsorted(a.indptr, np.linspace(0, nnz, bands + 1)).astype(np.int64); cuts[0] = 0; cuts[-1] = n
    band = np.searchsorted(cuts, rows, side="right") - 1
    ptr = a.indptr[cuts].astype(np.int64)
    for order in ("row", "column", "column, xcd-interleaved bands"):
        if order == "row": cols = a.indices.astype(np.int32)
        else: cols = a.indices[np.lexsort((a.indices, band))].astype(np.int32)
        p = ptr
        cd, pd = torch.from_numpy(cols).cuda(), torch.from_numpy(p).cuda()
        for threads in (256, 1024):
            us = timeit(lambda: lib.run(cd.data_ptr(), pd.data_ptr(), x.data_ptr(), out.data_ptr(), d // 4, bands, threads, st))
            print(f"bands={bands:5d} threads={threads:4d} order={order:32s}: {us:7.1f} us  gather {nnz * d * 4 / us / 1e6:6.2f} TB/s", flush=True)
        if order == "column": break
