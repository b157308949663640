// calibration: v_mfma_f32_16x16x4_f32 fed the way the GEMM feeds it -- 8 ds_read_b128 fragments per 32 MFMAs, the reads
// one chunk ahead -- against the same MFMA stream from fixed registers.  hipcc --offload-arch=gfx950 -O3 mfma_lds.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>  // 0: operands fixed in registers, 1: operands rotate over 32 registers, 2: operands from LDS each chunk
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
  __shared__ float4 lds[16 * 8 * 64];  // 16 chunks x 8 fragment blocks (128 KB)
  for (int i = threadIdx.x; i < 16 * 8 * 64; i += 256) lds[i] = make_float4(seed + i, 1.f, 2.f, 3.f);
  __syncthreads();
  const int lane = threadIdx.x & 63;
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float4 a[8], an[8];
  for (int u = 0; u < 8; ++u) a[u] = lds[u * 64 + lane];
  float4 b = make_float4(seed, seed + 1, seed + 2, seed + 3);
  for (int it = 0; it < iters; ++it) {
    const int kc = (it + 1) & 15;
    if (MODE == 2) {
#pragma unroll
      for (int u = 0; u < 8; ++u) an[u] = lds[(kc * 8 + u) * 64 + lane];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float bv = e == 0 ? b.x : e == 1 ? b.y : e == 2 ? b.z : b.w;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float av = MODE == 0 ? a[0].x : (e == 0 ? a[u].x : e == 1 ? a[u].y : e == 2 ? a[u].z : a[u].w);
        acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[u], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (MODE == 2) {
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] = an[u];
    }
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
// The same wave tile (32 nodes x 128 features of accumulators, 10 ds_read_b128 per 16-wide K chunk) on v_mfma_f32_32x32x2_f32:
// 4 accumulators of 16 registers, 32 MFMAs of 64 cycles per chunk instead of 64 of 32.  Lane l supplies row l % 32 and, from its
// float4, k = 4 (l / 32) + e of an 8-wide K step (a permutation of the reduction order both operands share).
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k32(float* out, int iters, float seed) {
  __shared__ float4 lds[16 * 10 * 64];  // 16 chunks x (2 node + 8 feature fragment blocks)
  for (int i = threadIdx.x; i < 16 * 10 * 64; i += 256) lds[i] = make_float4(seed + i, 1.f, 2.f, 3.f);
  __syncthreads();
  const int lane = threadIdx.x & 63;
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float4 a[8], an[8], b[2], bn[2];
  for (int u = 0; u < 8; ++u) a[u] = lds[u * 64 + lane];
  for (int u = 0; u < 2; ++u) b[u] = lds[(8 + u) * 64 + lane];
  for (int it = 0; it < iters; ++it) {
    const int kc = (it + 1) & 15;
#pragma unroll
    for (int u = 0; u < 8; ++u) an[u] = lds[(kc * 10 + u) * 64 + lane];
#pragma unroll
    for (int u = 0; u < 2; ++u) bn[u] = lds[(kc * 10 + 8 + u) * 64 + lane];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int h = 0; h < 2; ++h) {          // two 8-wide K steps per 16-wide chunk
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float bv = e == 0 ? b[h].x : e == 1 ? b[h].y : e == 2 ? b[h].z : b[h].w;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float4 av4 = a[4 * h + u];
          const float av = e == 0 ? av4.x : e == 1 ? av4.y : e == 2 ? av4.z : av4.w;
          acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[u], 0, 0, 0);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 8; ++u) a[u] = an[u];
#pragma unroll
    for (int u = 0; u < 2; ++u) b[u] = bn[u];
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
// 16x16x4 with the GEMM's wave tile (MT = 2 node tiles x NT = 8 feature tiles: 16 accumulators, 64 MFMAs per chunk, 10 reads)
__global__ __launch_bounds__(256) void k16(float* out, int iters, float seed) {
  __shared__ float4 lds[16 * 10 * 64];
  for (int i = threadIdx.x; i < 16 * 10 * 64; i += 256) lds[i] = make_float4(seed + i, 1.f, 2.f, 3.f);
  __syncthreads();
  const int lane = threadIdx.x & 63;
  f32x4 acc[2][8];
  for (int t = 0; t < 2; ++t) for (int i = 0; i < 8; ++i) acc[t][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float4 a[8], an[8], b[2], bn[2];
  for (int u = 0; u < 8; ++u) a[u] = lds[u * 64 + lane];
  for (int u = 0; u < 2; ++u) b[u] = lds[(8 + u) * 64 + lane];
  for (int it = 0; it < iters; ++it) {
    const int kc = (it + 1) & 15;
#pragma unroll
    for (int u = 0; u < 8; ++u) an[u] = lds[(kc * 10 + u) * 64 + lane];
#pragma unroll
    for (int u = 0; u < 2; ++u) bn[u] = lds[(kc * 10 + 8 + u) * 64 + lane];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const float bv = e == 0 ? b[t].x : e == 1 ? b[t].y : e == 2 ? b[t].z : b[t].w;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const float av = e == 0 ? a[u].x : e == 1 ? a[u].y : e == 2 ? a[u].z : a[u].w;
          acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[t][u], 0, 0, 0);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 8; ++u) a[u] = an[u];
#pragma unroll
    for (int u = 0; u < 2; ++u) b[u] = bn[u];
  }
  float s = 0.f;
  for (int t = 0; t < 2; ++t) for (int i = 0; i < 8; ++i) s += acc[t][i][0] + acc[t][i][1] + acc[t][i][2] + acc[t][i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  float* out; hipMalloc(&out, 4 * 256 * 4096);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int blocks : {256, 512}) {   // one 4-wave workgroup per CU (1 wave per SIMD) / two
    for (int mode : {0, 1, 2}) {
      const int iters = 2048;
      auto launch = [&]() {
        if (mode == 0) k<0><<<blocks, 256>>>(out, iters, 1.f);
        else if (mode == 1) k<1><<<blocks, 256>>>(out, iters, 1.f);
        else k<2><<<blocks, 256>>>(out, iters, 1.f);
      };
      launch(); hipDeviceSynchronize();
      hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double n = (double)blocks * 4 * iters * 32;
      printf("blocks=%4d mode=%d: %.3f ms  %.1f TFLOP/s (%.1f %% of 157.3)\n", blocks, mode, ms, n * 2048.0 / (ms * 1e-3) / 1e12,
             n * 2048.0 / (ms * 1e-3) / 1e12 / 157.3 * 100);
    }
  }
  // equal wave tiles (32 x 128 outputs, 2048 MFMA cycles and 10 ds_read_b128 per 16-wide chunk) on the two fp32 MFMA shapes
  for (int blocks : {256, 512}) {
    for (int shape : {16, 32}) {
      const int iters = 1024;
      auto launch = [&]() {
        if (shape == 16) k16<<<blocks, 256>>>(out, iters, 1.f);
        else k32<<<blocks, 256>>>(out, iters, 1.f);
      };
      launch(); hipDeviceSynchronize();
      hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double fl = (double)blocks * 4 * iters * 2.0 * 32 * 128 * 16;
      printf("wave tile 32x128, LDS-fed, blocks=%4d shape=%s: %.3f ms  %.1f TFLOP/s (%.1f %% of 157.3)\n", blocks,
             shape == 16 ? "16x16x4" : "32x32x2", ms, fl / (ms * 1e-3) / 1e12, fl / (ms * 1e-3) / 1e12 / 157.3 * 100);
    }
  }
  return 0;
}
