// calibration: does VALU work of one wave slow the MFMA stream of another wave on the same SIMD?
// workgroup = 8 waves (2 per SIMD): waves 0-3 run v_mfma_f32_16x16x4_f32 back to back, waves 4-7 run
// (mode 0) nothing, (mode 1) a dependent v_fma_f32 chain, (mode 2) expf, (mode 3) global stores.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, float* sink, int iters, float seed) {
  const int w = threadIdx.x >> 6;
  if (w < 4) {  // waves 0-3 and 4-7 land on SIMDs 0-3 each: every SIMD has one MFMA wave and one other wave
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float av = seed + threadIdx.x, bv = seed;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 512 + threadIdx.x] = s;
  } else if (MODE != 0) {
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = seed * 1e-3f + i + threadIdx.x * 1e-6f;
    for (int it = 0; it < iters; ++it) {
      if (MODE == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
          for (int i = 0; i < 8; ++i) x[i] = fmaf(x[i], 0.999f, 1e-3f);
      } else if (MODE == 2) {
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = expf(x[i] * -0.5f);
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) sink[((size_t)blockIdx.x * 512 + threadIdx.x) * 8 + i + (size_t)(it & 63) * 512 * 8 * 256] = x[i];
      }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
  }
}
int main() {
  float *out, *sink;
  (void)hipMalloc(&out, 4 * 512 * 1024);
  (void)hipMalloc(&sink, (size_t)4 * 512 * 8 * 256 * 64);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int blocks = 256, iters = 2048;
  for (int mode : {0, 1, 2, 3}) {
    auto launch = [&]() {
      if (mode == 0) k<0><<<blocks, 512>>>(out, sink, iters, 1.f);
      else if (mode == 1) k<1><<<blocks, 512>>>(out, sink, iters, 1.f);
      else if (mode == 2) k<2><<<blocks, 512>>>(out, sink, iters, 1.f);
      else k<3><<<blocks, 512>>>(out, sink, iters, 1.f);
    };
    launch(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)blocks * 4 * iters * 32;   // MFMAs: 4 waves per workgroup
    printf("mode=%d: %.3f ms  MFMA waves alone would need %.3f ms; MFMA rate %.1f TFLOP/s\n", mode, ms, n * 32 / 2.4e9 / 1024 * 1e3 * 1.0,
           n * 2048.0 / (ms * 1e-3) / 1e12);
  }
  return 0;
}
