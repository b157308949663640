#!/usr/bin/env python3
"""ceiling of an SpMM gather whose feature slices are pinned to XCDs: workgroup b reads only slice b % nslices of every
row it gathers, so each XCD's L2 caches 1/nslices of the table.  Gathers only.  Tools only, not shipped."""
import ctypes as C, os, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from gcn_drug_repurposing_amd import synth
import scipy.sparse as sp
SRC = r'''
#include <hip/hip_runtime.h>
// LPR lanes x float4 read one slice of one row; a wave reads 64/LPR rows per load instruction
template <int LPR, int UNROLL>
__global__ __launch_bounds__(256) void gather_slice(const int* __restrict__ col, long nnz, const float* __restrict__ x, float* out, int d4,
                                                    int nslices) {
  constexpr int RPW = 64 / LPR;
  const int lane = threadIdx.x & 63, sub = lane / LPR, li = lane % LPR;
  const int slice = blockIdx.x % nslices;
  const long wave = (long)(blockIdx.x / nslices) * 4 + (threadIdx.x >> 6);
  const long nwaves = (long)(gridDim.x / nslices) * 4;
  float4 acc = make_float4(0, 0, 0, 0);
  for (long base = wave * 64; base < nnz; base += nwaves * 64) {
    const long e = base + lane;
    const int c = e < nnz ? col[e] : 0;
    for (int t = 0; t < 64; t += RPW * UNROLL) {
      float4 v[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        const int cc = __shfl(c, t + RPW * u + sub, 64);
        v[u] = *reinterpret_cast<const float4*>(x + ((size_t)cc * d4 + slice * LPR + li) * 4);
      }
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
  }
  if (acc.x == 12345.f) out[threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}
extern "C" void run(const int* col, long nnz, const float* x, float* out, int d4, int blocks, int lpr, int unroll, void* st) {
  const int ns = d4 / lpr;
#define GO(L, U) gather_slice<L, U><<<blocks, 256, 0, (hipStream_t)st>>>(col, nnz, x, out, d4, ns)
  if (lpr == 32) { if (unroll == 4) GO(32, 4); else GO(32, 8); }
  else if (lpr == 16) { if (unroll == 4) GO(16, 4); else GO(16, 2); }
  else if (lpr == 8) { if (unroll == 4) GO(8, 4); else GO(8, 2); }
  else { if (unroll == 4) GO(4, 4); else GO(4, 2); }
}
'''
tmp = tempfile.mkdtemp()
open(os.path.join(tmp, "g.hip"), "w").write(SRC)
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", os.path.join(tmp, "g.hip"), "-o", os.path.join(tmp, "g.so")])
lib = C.CDLL(os.path.join(tmp, "g.so"))
lib.run.argtypes = [C.c_void_p, C.c_long, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
adj, _, _ = synth.whole_graph_standin(1)
a = (adj + sp.eye(adj.shape[0], format="csr")).tocsr(); a.sort_indices()
n, nnz, d = a.shape[0], a.nnz, 128
x = torch.randn(n, d, device="cuda"); out = torch.zeros(1024, device="cuda")
cd = torch.from_numpy(a.indices.astype(np.int32)).cuda()
st = torch.cuda.current_stream().cuda_stream
def timeit(fn):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20 * 1e3
for lpr in (32, 16, 8, 4):
    for blocks in (2048, 4096, 8192):
        for unroll in (4, 2 if lpr < 32 else 8):
            us = timeit(lambda: lib.run(cd.data_ptr(), nnz, x.data_ptr(), out.data_ptr(), d // 4, blocks, lpr, unroll, st))
            print(f"slice={lpr*16:4d} B ({d//4//lpr} slices) blocks={blocks:5d} unroll={unroll}: {us:7.1f} us  gather {nnz * d * 4 / us / 1e6:6.2f} TB/s", flush=True)
