// calibration: issue rate of v_mfma_f32_16x16x4_f32 (1 or 2 waves per SIMD, 8 independent accumulators)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float av = a + threadIdx.x, bv = b;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  float* out; hipMalloc(&out, 4 * 256 * 4096);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int blocks : {256, 512, 1024}) {
    for (int nacc : {2, 8}) {
      const int iters = 4096;
      auto launch = [&]() { if (nacc == 8) k<8><<<blocks, 256>>>(out, iters, 1.f, 2.f); else k<2><<<blocks, 256>>>(out, iters, 1.f, 2.f); };
      launch(); hipDeviceSynchronize();
      hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double mfma_per_simd = (double)blocks * 4 / 1024.0 * iters * nacc;
      printf("blocks=%4d nacc=%d: %.3f ms  -> %.1f ns per MFMA per SIMD = %.1f cycles @2.4GHz, %.1f TFLOP/s\n", blocks, nacc, ms,
             ms * 1e6 / mfma_per_simd, ms * 1e6 / mfma_per_simd * 2.4, (double)blocks * 4 * iters * nacc * 2048.0 / (ms * 1e-3) / 1e12);
    }
  }
  return 0;
}
