#!/usr/bin/env python3
"""Ceiling of an 'LDS hub tile' for the SpMM gather.  The graph's nodes are relabelled hub-first, so the
H most referenced feature rows are rows [0, H) of the operand.  A persistent grid of 1024-thread workgroups, each pinned to one 256-B
feature slice (slice = workgroup id mod 2, as the product kernel pins them to XCD parity), first copies rows [0, H) of its slice into
LDS (H = 320: 80 KB, two workgroups per CU; H = 640: 160 KB, one) and then walks the column stream: 16 lanes fetch one 256-B row slice,
from LDS when col < H, from global memory otherwise; 4 row fetches in flight per lane group, as in spmm_balanced_kernel.  Gathers
only (no values, no reduction, no stores): what the memory system gives for this access stream, with and without the tile.
Prints the share of gathers served by LDS, the time per pass and the gathered bytes per second.  Tools only, not shipped.
usage: gather_hub_lds.py [whole_graph | whole_graph_pathway | rmat:<n>:<m>] [d (default 128; other widths: the bare stream only, H = 0)]"""
import ctypes as C, os, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import scipy.sparse as sp
from gcn_drug_repurposing_amd import synth
SRC = r'''
#include <hip/hip_runtime.h>
// LPR = 16 lanes x float4 = one 256-B slice of one row; a wave has 4 lane groups
template <int FLY, bool BF>
__global__ __launch_bounds__(1024) void gather_hub(const int* __restrict__ col, long nnz, const float* __restrict__ x, float* out, int d4, int H,
                                                   int nslices) {
  extern __shared__ float4 hub[];   // [H][16]
  const int lane = threadIdx.x & 63, sub = lane >> 4, li = lane & 15;
  const int slice = blockIdx.x % nslices;
  const int nwg = gridDim.x / nslices, wg = blockIdx.x / nslices;
  for (int i = threadIdx.x; i < H * 16; i += 1024) hub[i] = *reinterpret_cast<const float4*>(x + ((size_t)(i >> 4) * d4 + slice * 16 + (i & 15)) * 4);
  __syncthreads();
  const long wave = (long)wg * 16 + (threadIdx.x >> 6);
  const long nwaves = (long)nwg * 16;
  float4 acc = make_float4(0, 0, 0, 0);
  for (long base = wave * 64; base < nnz; base += nwaves * 64) {
    const long e = base + lane;
    const int c = e < nnz ? col[e] : 0;
    for (int t = 0; t < 64; t += 4 * FLY) {
      float4 v[FLY];
#pragma unroll
      for (int u = 0; u < FLY; ++u) {
        const int cc = __shfl(c, t + 4 * u + sub, 64);
        if (BF) {
          // branch-free form: the LDS read is always issued (clamped row), the global load is predicated, the value selected afterwards --
          // no wait sits inside a branch, all FLY fetches are in flight together
          float4 gv = make_float4(0, 0, 0, 0);
          if (cc >= H) gv = *reinterpret_cast<const float4*>(x + ((size_t)cc * d4 + slice * 16 + li) * 4);
          const float4 lv = hub[(H > 0 ? min(cc, H - 1) : 0) * 16 + li];
          v[u] = cc < H ? lv : gv;
        } else {
          if (cc < H) v[u] = hub[cc * 16 + li];
          else v[u] = *reinterpret_cast<const float4*>(x + ((size_t)cc * d4 + slice * 16 + li) * 4);
        }
      }
#pragma unroll
      for (int u = 0; u < FLY; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
  }
  if (acc.x == 12345.f) out[threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}
extern "C" int run(const int* col, long nnz, const float* x, float* out, int d4, int blocks, int H, int fly, void* st) {
  const int ns = d4 / 16;
  const size_t lds = (size_t)(H > 0 ? H : 1) * 256;
#define GO(F, B)                                                                                                        \
  do {                                                                                                                  \
    (void)hipFuncSetAttribute((const void*)gather_hub<F, B>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);  \
    gather_hub<F, B><<<blocks, 1024, lds, (hipStream_t)st>>>(col, nnz, x, out, d4, H, ns);                              \
  } while (0)
  if (fly == 4) GO(4, false);
  else if (fly == 8) GO(8, false);
  else if (fly == 14) GO(4, true);     // 1x = branch-free form
  else GO(8, true);
  return (int)hipGetLastError();
}
'''
tmp = tempfile.mkdtemp()
open(os.path.join(tmp, "g.hip"), "w").write(SRC)
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", os.path.join(tmp, "g.hip"), "-o", os.path.join(tmp, "g.so")])
lib = C.CDLL(os.path.join(tmp, "g.so"))
lib.run.argtypes = [C.c_void_p, C.c_long, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
lib.run.restype = C.c_int
what = sys.argv[1] if len(sys.argv) > 1 else "whole_graph"
if what in ("whole_graph", "whole_graph_pathway"):
    adj, _, _ = synth.whole_graph_standin(1, pathway_edges=what.endswith("pathway"))
    a = (adj + sp.eye(adj.shape[0], format="csr")).tocsr()
    a.sort_indices()
    n, nnz = a.shape[0], a.nnz
    deg = np.bincount(a.indices, minlength=n) + np.diff(a.indptr)
    perm = np.argsort(-deg, kind="stable")
    inv = np.empty(n, np.int64); inv[perm] = np.arange(n)
    ar = a[perm]                                   # rows in hub-first order, entries keep their order
    cols = inv[ar.indices].astype(np.int32)
    cd = torch.from_numpy(cols).cuda()
else:
    _, n_, m_ = what.split(":")
    n, m = int(n_), int(m_)
    # RMAT-like skew without the generator: column ids drawn from a Zipf-ish law over hub-first ids (timing only)
    g = torch.Generator(device="cuda"); g.manual_seed(4)
    u = torch.rand(m, generator=g, device="cuda", dtype=torch.float64)
    cd = (n ** u - 1).clamp_(0, n - 1).to(torch.int32)      # log-uniform ids: P(col < k) = log(k+1) / log(n)
    nnz = m
d = int(sys.argv[2]) if len(sys.argv) > 2 else 128
x = torch.randn(n, d, device="cuda"); out = torch.zeros(1024, device="cuda")
st = torch.cuda.current_stream().cuda_stream
def timeit(fn):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20 * 1e3
print(f"{what}: N={n} nnz={nnz} d={d}; gathers per pass {nnz} x {d * 4} B = {nnz * d * 4 / 1e6:.0f} MB", flush=True)
for H, blocks_list in (((0, (512, 1024, 2048)), (160, (512, 1024)), (320, (512,)), (640, (256,))) if d == 128 else ((0, (512, 1024, 2048)),)):
    share = float((cd < H).float().mean()) if H else 0.0
    for blocks in blocks_list:
        for fly in ((4, 8) if H == 0 else (4, 14, 18)):
            def call():
                rc = lib.run(cd.data_ptr(), nnz, x.data_ptr(), out.data_ptr(), d // 4, blocks, H, fly, st)
                assert rc == 0, rc
            us = timeit(call)
            fill = blocks * H * 256 / 1e6
            print(f"H={H:4d} ({H * 256 // 1024:3d} KB LDS, {share * 100:4.1f} % of gathers from LDS, tile fills {fill:5.1f} MB) workgroups={blocks:5d} "
                  f"in flight={fly % 10}{' branch-free' if fly > 10 else ''}: {us:8.1f} us  {nnz * d * 4 / us / 1e6:6.2f} TB/s gathered", flush=True)
