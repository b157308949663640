#!/usr/bin/env python3
"""ceiling of the SpMM gather on this GPU: stream the graph's col[] array and gather 512-B rows, nothing else.
Compiles a tiny HIP kernel with hipcc at run time (tools only, not shipped)."""
import ctypes as C, os, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from gcn_drug_repurposing_amd import synth
import scipy.sparse as sp
SRC = r'''
#include <hip/hip_runtime.h>
typedef float v4f __attribute__((ext_vector_type(4)));
// each half-wave (32 lanes x float4 = 512 B) gathers one row per step; UNROLL rows in flight per lane
template <int UNROLL>
__global__ __launch_bounds__(256) void gather_nt(const int* __restrict__ col, long nnz, const float* __restrict__ x, float* out, int d4, int hot) {
  const int lane = threadIdx.x & 63, half = lane >> 5, li = lane & 31;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long nwaves = (long)gridDim.x * 4;
  float4 acc = make_float4(0, 0, 0, 0);
  for (long base = wave * 64; base < nnz; base += nwaves * 64) {
    const long e = base + lane;
    const int c = e < nnz ? col[e] : 0;
    for (int t = 0; t < 64; t += 2 * UNROLL) {
      float4 v[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        const int cc = __shfl(c, t + 2 * u + half, 64);
        const float4* p = reinterpret_cast<const float4*>(x + ((size_t)cc * d4 + li) * 4);
        if (cc < hot) v[u] = *p; else { const v4f t = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p)); v[u] = make_float4(t[0], t[1], t[2], t[3]); }  // cold rows: do not keep them in L2
      }
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
  }
  if (acc.x == 12345.f) out[threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}
template <int UNROLL>
__global__ __launch_bounds__(256) void gather(const int* __restrict__ col, long nnz, const float* __restrict__ x, float* out, int d4) {
  const int lane = threadIdx.x & 63, half = lane >> 5, li = lane & 31;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long nwaves = (long)gridDim.x * 4;
  float4 acc = make_float4(0, 0, 0, 0);
  for (long base = wave * 64; base < nnz; base += nwaves * 64) {
    const long e = base + lane;
    const int c = e < nnz ? col[e] : 0;
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
extern "C" void run_nt(const int* col, long nnz, const float* x, float* out, int d4, int blocks, int hot, void* st) {
  gather_nt<8><<<blocks, 256, 0, (hipStream_t)st>>>(col, nnz, x, out, d4, hot);
}
extern "C" void run(const int* col, long nnz, const float* x, float* out, int d4, int blocks, int unroll, void* st) {
  if (unroll == 8) gather<8><<<blocks, 256, 0, (hipStream_t)st>>>(col, nnz, x, out, d4);
  else gather<4><<<blocks, 256, 0, (hipStream_t)st>>>(col, nnz, x, out, d4);
}
'''
tmp = tempfile.mkdtemp()
open(os.path.join(tmp, "g.hip"), "w").write(SRC)
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", os.path.join(tmp, "g.hip"), "-o", os.path.join(tmp, "g.so")])
lib = C.CDLL(os.path.join(tmp, "g.so"))
lib.run.argtypes = [C.c_void_p, C.c_long, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
lib.run_nt.argtypes = [C.c_void_p, C.c_long, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
adj, _, _ = synth.whole_graph_standin(1)
a = (adj + sp.eye(adj.shape[0], format="csr")).tocsr(); a.sort_indices()
n, nnz, d = a.shape[0], a.nnz, 128
x = torch.randn(n, d, device="cuda"); out = torch.zeros(256, device="cuda")
rng = np.random.RandomState(0)
streams = {"graph col[] (CSR order)": a.indices.astype(np.int32),
           "uniform random": rng.randint(0, n, nnz).astype(np.int32),
           "graph col[] shuffled": rng.permutation(a.indices).astype(np.int32),
           "sequential": (np.arange(nnz) % n).astype(np.int32),
           # perfect time alignment: concurrently running waves read the same narrow column band
           "graph col[] sorted by column": np.sort(a.indices).astype(np.int32),
           "graph col[] sorted within 16 bands of rows": None}
# rows cut into 16 bands (what one co-resident set of workgroups would own); inside a band the stream is in column order
rb = np.repeat(np.arange(n), np.diff(a.indptr)) * 16 // n
streams["graph col[] sorted within 16 bands of rows"] = a.indices[np.lexsort((a.indices, rb))].astype(np.int32)
if True:
    pass
for name, cols in streams.items():
    cd = torch.from_numpy(cols).cuda()
    for blocks in (2048,):
        for unroll in (4, 8):
            st = torch.cuda.current_stream().cuda_stream
            for _ in range(3): lib.run(cd.data_ptr(), nnz, x.data_ptr(), out.data_ptr(), d // 4, blocks, unroll, st)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): lib.run(cd.data_ptr(), nnz, x.data_ptr(), out.data_ptr(), d // 4, blocks, unroll, st)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 20 * 1e3
            print(f"{name:28s} blocks={blocks} unroll={unroll}: {us:7.1f} us  gather {nnz * d * 4 / us / 1e6:6.2f} TB/s")

# popularity relabelling: node ids sorted by descending in-degree, cold rows loaded non-temporally
indeg = np.bincount(a.indices, minlength=n)
rank = np.empty(n, dtype=np.int64); rank[np.argsort(-indeg, kind="stable")] = np.arange(n)
cols_sorted = rank[a.indices].astype(np.int32)
cd = torch.from_numpy(cols_sorted).cuda()
st = torch.cuda.current_stream().cuda_stream
def timeit(fn):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20 * 1e3
us = timeit(lambda: lib.run(cd.data_ptr(), nnz, x.data_ptr(), out.data_ptr(), d // 4, 2048, 8, st))
print(f"popularity-relabelled, plain loads         : {us:7.1f} us  gather {nnz * d * 4 / us / 1e6:6.2f} TB/s")
for hot in (0, 2048, 4096, 6144, 8192, 12288, n):
    us = timeit(lambda: lib.run_nt(cd.data_ptr(), nnz, x.data_ptr(), out.data_ptr(), d // 4, 2048, hot, st))
    share = indeg[np.argsort(-indeg)][:hot].sum() / nnz
    print(f"popularity-relabelled, nt for cols >= {hot:6d} ({share*100:4.1f}% of refs hot): {us:7.1f} us  gather {nnz * d * 4 / us / 1e6:6.2f} TB/s")
