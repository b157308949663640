// Round 4 micro-benchmark: does the SHAPE of the projection epilogue's accesses matter?  Every wave of the projection owns a 16-node x 128-
// feature tile; lane (r = lane & 15, q = lane >> 4) holds features 16 u + 4 q .. + 3 of node r for u = 0..7, so one store instruction
// writes 16 rows x 64 B (half cache lines), and a row's 512 B leave in 8 instructions.  Here the same bytes (read P_prev, write P and x')
// move with nothing else in the kernel, in three shapes:
//   A  as the epilogue does it             : 16 rows x  64 B per instruction
//   B  lane pairs exchange two u blocks    :  8 rows x 128 B per instruction (whole lines)
//   C  row-contiguous (through LDS, say)   :  2 rows x 512 B per instruction
// build: hipcc --offload-arch=gfx950 -O3 store_pattern.hip -o store_pattern ; usage: store_pattern [nodes]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__device__ __forceinline__ float4 f(float4 a) { return make_float4(a.x * 1.5f + 1.f, a.y * 1.5f + 1.f, a.z * 1.5f + 1.f, a.w * 1.5f + 1.f); }
template <int SHAPE>
__global__ __launch_bounds__(256) void k(int n, const float *__restrict__ pp, float *__restrict__ p, float *__restrict__ xn) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int node0 = blockIdx.x * 64 + 16 * w;
  float4 v[8];
  size_t off[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    int row, col;
    if (SHAPE == 0) {
      row = lane & 15; col = 16 * u + 4 * (lane >> 4);
    } else if (SHAPE == 1) {
      // instruction u: pair m = u >> 1 covers features 32 m .. 32 m + 31; rows of parity u & 1; the lane with r even carries the lower
      // 16 features, its odd neighbour the upper 16
      const int r = lane & 15, q = lane >> 4, m = u >> 1, par = u & 1;
      row = (r & ~1) | par; col = 32 * m + 16 * (r & 1) + 4 * q;
    } else {
      row = 2 * u + (lane >> 5); col = 4 * (lane & 31);
    }
    const int nd = min(n - 1, node0 + row);
    off[u] = (size_t)nd * 128 + col;
  }
#pragma unroll
  for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4 *>(pp + off[u]);
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const float4 o = f(v[u]);
    *reinterpret_cast<float4 *>(p + off[u]) = o;
    *reinterpret_cast<float4 *>(xn + off[u]) = f(o);
  }
}
int main(int argc, char **argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 29960;
  float *pp, *p, *xn;
  hipMalloc(&pp, (size_t)n * 128 * 4); hipMalloc(&p, (size_t)n * 128 * 4); hipMalloc(&xn, (size_t)n * 128 * 4);
  hipMemset(pp, 0, (size_t)n * 128 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = (n + 63) / 64;
  const char *names[3] = {"A 16 rows x 64 B ", "B  8 rows x 128 B", "C  2 rows x 512 B"};
  for (int rnd = 0; rnd < 3; ++rnd)
    for (int s = 0; s < 3; ++s) {
      auto launch = [&]() {
        if (s == 0) k<0><<<blocks, 256>>>(n, pp, p, xn);
        else if (s == 1) k<1><<<blocks, 256>>>(n, pp, p, xn);
        else k<2><<<blocks, 256>>>(n, pp, p, xn);
      };
      for (int i = 0; i < 5; ++i) launch();
      hipDeviceSynchronize();
      const int reps = n < 200000 ? 50 : 5;
      hipEventRecord(e0);
      for (int i = 0; i < reps; ++i) launch();
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double us = ms * 1e3 / reps, mb = 3.0 * n * 512 / 1e6;
      printf("n=%d %s: %8.2f us per launch (%d back to back), %.1f MB -> %.2f TB/s\n", n, names[s], us, reps, mb, mb / us);
    }
  return 0;
}
