// Round 4 micro-benchmark: the projection's OPERAND stream by itself.  Per 16-wide K chunk a workgroup of gemm_nt_lds_kernel moves 64 node
// rows x 64 B and 128 weight rows x 64 B into LDS with global_load_lds_dwordx4 -- 16 rows x 64 B (half cache lines) per instruction, 3
// instructions per wave and chunk, counted waits, one barrier per chunk.  Here that stream runs with no MFMA behind it, in two shapes:
//   A  as the kernel does it                   : 16 rows x  64 B per instruction, 16 chunks
//   B  two chunks at a time, whole cache lines :  8 rows x 128 B per instruction,  8 double chunks (same bytes, same instruction count)
// build: hipcc --offload-arch=gfx950 -O3 dma_shape.hip -o dma_shape ; usage: dma_shape [nodes]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__device__ __forceinline__ void glds16(const float *gsrc, unsigned lds_byte_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_byte_addr) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int SHAPE>
__global__ __launch_bounds__(256) void k(int n, const float *__restrict__ ax, const float *__restrict__ am, const float *__restrict__ w1,
                                         const float *__restrict__ w2, float *__restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int node_base = blockIdx.x * 64;
  constexpr int NBUF = 4, PF = 3, G = 3;
  constexpr int BUF = (64 + 128) * 16 * 4;   // bytes of one 16-wide chunk
  // SHAPE 0: block b (16 rows) of chunk ci: lane (r = lane & 15, q = lane >> 4) reads row 16 b + r, floats 16 ci + 4 q
  // SHAPE 1: double chunk cj (32 floats = 128 B per row): an instruction covers 8 rows: lane (r8 = lane >> 3, p = lane & 7) reads
  //          row 8 b8 + r8, floats 32 cj + 4 p; 24 half-blocks of X+W per double chunk = 6 per wave
  auto row_ptr = [&](int row_in_tile, int kf, bool weights) -> const float * {
    // kf in [0, 256): k < 128 from ax / w1, else am / w2
    if (!weights) {
      const int node = min(n - 1, node_base + row_in_tile);
      return (kf < 128 ? ax + (size_t)node * 128 + kf : am + (size_t)node * 128 + (kf - 128));
    }
    return (kf < 128 ? w1 + (size_t)row_in_tile * 128 + kf : w2 + (size_t)row_in_tile * 128 + (kf - 128));
  };
  float acc = 0.f;
  if (SHAPE == 0) {
    auto stage = [&](int ci) {
      const unsigned buf = lds_base + (unsigned)((ci % NBUF) * BUF);
      const int r = lane & 15, q = lane >> 4;
      glds16(row_ptr(16 * w + r, 16 * ci + 4 * q, false), buf + (unsigned)(w * 1024));
      glds16(row_ptr(16 * w + r, 16 * ci + 4 * q, true), buf + (unsigned)(64 * 64 + w * 1024));
      glds16(row_ptr(16 * (w + 4) + r, 16 * ci + 4 * q, true), buf + (unsigned)(64 * 64 + (w + 4) * 1024));
    };
    for (int c = 0; c < PF; ++c) stage(c);
    for (int ci = 0; ci < 16; ++ci) {
      const int younger = min(PF - 1, 15 - ci);
      if (younger >= 2) wait_vmcnt<2 * G>(); else if (younger == 1) wait_vmcnt<G>(); else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      acc += reinterpret_cast<const float *>(smem)[(ci % NBUF) * (BUF / 4) + threadIdx.x];
      if (ci + PF < 16) stage(ci + PF);
    }
  } else {
    constexpr int DBUF = 2 * BUF;    // a double chunk; ring of 3 (72 KB), 2 in flight
    auto stage = [&](int cj) {
      const unsigned buf = lds_base + (unsigned)((cj % 3) * DBUF);
      const int r8 = lane >> 3, p = lane & 7;
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const int hb = w + 4 * i;                 // half-block 0..23: 0..7 node rows, 8..23 weight rows
        const bool wt = hb >= 8;
        const int row = 8 * (wt ? hb - 8 : hb) + r8;
        glds16(row_ptr(row, 32 * cj + 4 * p, wt), buf + (unsigned)(hb * 1024));
      }
    };
    stage(0); stage(1);
    for (int cj = 0; cj < 8; ++cj) {
      if (cj < 7) wait_vmcnt<6>(); else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      acc += reinterpret_cast<const float *>(smem)[(cj % 3) * (DBUF / 4) + threadIdx.x];
      if (cj + 2 < 8) stage(cj + 2);
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}
int main(int argc, char **argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 29960;
  float *ax, *am, *w1, *w2, *out;
  hipMalloc(&ax, (size_t)n * 512); hipMalloc(&am, (size_t)n * 512); hipMalloc(&w1, 128 * 512); hipMalloc(&w2, 128 * 512);
  const int blocks = (n + 63) / 64;
  hipMalloc(&out, (size_t)blocks * 1024);
  hipMemset(ax, 0, (size_t)n * 512); hipMemset(am, 0, (size_t)n * 512); hipMemset(w1, 0, 128 * 512); hipMemset(w2, 0, 128 * 512);
  hipFuncSetAttribute((const void *)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
  hipFuncSetAttribute((const void *)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char *names[2] = {"A 16 rows x 64 B per instruction, 16 chunks       ", "B  8 rows x 128 B per instruction, 8 double chunks"};
  for (int rnd = 0; rnd < 3; ++rnd)
    for (int s = 0; s < 2; ++s) {
      auto launch = [&]() {
        if (s == 0) k<0><<<blocks, 256, 4 * 12288>>>(n, ax, am, w1, w2, out);
        else k<1><<<blocks, 256, 3 * 24576>>>(n, ax, am, w1, w2, out);
      };
      for (int i = 0; i < 5; ++i) launch();
      hipDeviceSynchronize();
      const int reps = n < 200000 ? 50 : 5;
      hipEventRecord(e0);
      for (int i = 0; i < reps; ++i) launch();
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double us = ms * 1e3 / reps, mb = (double)blocks * (64 + 128) * 1024 / 1e6;
      printf("n=%d %s: %8.2f us per launch, %.1f MB into LDS -> %.2f TB/s (%s)\n", n, names[s], us, mb, mb / us, hipGetErrorString(hipGetLastError()));
    }
  return 0;
}
