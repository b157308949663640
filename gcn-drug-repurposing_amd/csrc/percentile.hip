// percentile.hip -- exact beta selection (K12): np.percentile(E E^T flattened, q), train.py:165-167.
//
// The reference materialises the N x N similarity matrix on the host (3.6 GB at N = 29,960) and
// sorts it.  Here the N^2 inner products are recomputed per pass on fp32 MFMA (upper-triangular
// 64x64 tiles, off-diagonal tiles weighted twice) and the wanted order statistic is found by a
// 3-pass most-significant-digit radix select (11 + 11 + 10 bits of the order-preserving integer image
// of the float), plus one min-reduction pass when the upper neighbour lies outside the final bin.
// One-time setup work; MFMA-bound (2 N^2 d / 2 flops per pass).
#include <vector>

#include "ops.h"

namespace gss {

__device__ __forceinline__ f32x4 mfma16p(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ uint32_t ordered_key(float v) {
  const uint32_t u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
static inline float key_to_float(uint32_t k) {
  const uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

struct PctArgs {
  int n, d;
  const float *e;
  uint32_t mask, prefix;  // count keys with (key & mask) == prefix
  int shift, nbits;       // histogram digit
  unsigned long long *hist;  // [1 << nbits]
  uint32_t floor_key;     // MODE_MIN: smallest key > floor_key
  uint32_t *min_key;
};

// MODE 0: digit histogram of the keys matching the prefix; MODE 1: min key greater than floor_key
template <int MODE>
__global__ __launch_bounds__(256) void pct_kernel(PctArgs g) {
  __shared__ uint32_t lh[2048];
  const int it = blockIdx.y, jt = blockIdx.x;
  if (it > jt) return;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = lane & 15, q = lane >> 4;
  const int n = g.n, d = g.d;
  if (MODE == 0) {
    for (int k = threadIdx.x; k < (1 << g.nbits); k += 256) lh[k] = 0;
    __syncthreads();
  }
  const int i0 = it * 64 + (w >> 1) * 32, j0 = jt * 64 + (w & 1) * 32;
  const float *pi[2], *pj[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    pi[t] = g.e + (size_t)min(n - 1, i0 + 16 * t + c) * d + 4 * q;
    pj[t] = g.e + (size_t)min(n - 1, j0 + 16 * t + c) * d + 4 * q;
  }
  f32x4 s[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) s[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int kc = 0; kc < d; kc += 16) {
    float4 ai[2], bj[2];
    ai[0] = ld4(pi[0] + kc);
    ai[1] = ld4(pi[1] + kc);
    bj[0] = ld4(pj[0] + kc);
    bj[1] = ld4(pj[1] + kc);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        // D[row = j][col = i]: A = E_j, B = E_i
        s[a][b] = mfma16p(bj[b].x, ai[a].x, s[a][b]);
        s[a][b] = mfma16p(bj[b].y, ai[a].y, s[a][b]);
        s[a][b] = mfma16p(bj[b].z, ai[a].z, s[a][b]);
        s[a][b] = mfma16p(bj[b].w, ai[a].w, s[a][b]);
      }
  }
  const uint32_t wgt = (it == jt) ? 1u : 2u;
  uint32_t best = 0xffffffffu;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = i0 + 16 * a + c, j = j0 + 16 * b + 4 * q + r;
        if (i >= n || j >= n) continue;
        const uint32_t key = ordered_key(s[a][b][r]);
        if (MODE == 0) {
          if ((key & g.mask) == g.prefix) atomicAdd(&lh[(key >> g.shift) & ((1u << g.nbits) - 1u)], wgt);
        } else {
          if (key > g.floor_key && key < best) best = key;
        }
      }
  if (MODE == 0) {
    __syncthreads();
    for (int k = threadIdx.x; k < (1 << g.nbits); k += 256)
      if (lh[k]) atomicAdd(&g.hist[k], (unsigned long long)lh[k]);
  } else {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) best = min(best, (uint32_t)__shfl_xor((int)best, o, 64));
    if (lane == 0 && best != 0xffffffffu) atomicMin(g.min_key, best);
  }
}

}  // namespace gss

using namespace gss;

extern "C" int gss_percentile(int32_t n, int32_t d, const float *e, double q, float *h_out, void *stream) {
  if (int rc = check_d(d)) return rc;
  GSS_REQUIRE(n > 0 && e && h_out, "percentile: null operand");
  GSS_REQUIRE(q >= 0.0 && q <= 100.0, "percentile: q=%g outside [0, 100]", q);
  hipStream_t st = as_stream(stream);
  const unsigned long long M = (unsigned long long)n * (unsigned long long)n;
  const double pos = q / 100.0 * (double)(M - 1);
  unsigned long long lo = (unsigned long long)pos;
  if (lo > M - 1) lo = M - 1;
  const double frac = pos - (double)lo;
  const unsigned long long hi = lo + 1 < M ? lo + 1 : M - 1;

  unsigned long long *d_hist = nullptr;
  uint32_t *d_min = nullptr;
  GSS_HIP(hipMalloc((void **)&d_hist, sizeof(unsigned long long) * 2048 + 16));
  d_min = reinterpret_cast<uint32_t *>(d_hist + 2048);
  std::vector<unsigned long long> h_hist(2048);
  const int nt = ceil_div(n, 64);
  dim3 grid(nt, nt), block(256);
  PctArgs g{n, d, e, 0u, 0u, 0, 0, d_hist, 0u, d_min};
  const int shifts[3] = {21, 10, 0}, bits[3] = {11, 11, 10};
  unsigned long long rank = lo, eq_count = 0;
  int rc = GSS_OK;
  for (int pass = 0; pass < 3 && rc == GSS_OK; ++pass) {
    g.shift = shifts[pass];
    g.nbits = bits[pass];
    hipError_t err = hipMemsetAsync(d_hist, 0, sizeof(unsigned long long) * 2048, st);
    if (err == hipSuccess) {
      hipLaunchKernelGGL((pct_kernel<0>), grid, block, 0, st, g);
      err = hipGetLastError();
    }
    if (err == hipSuccess) err = hipMemcpyAsync(h_hist.data(), d_hist, sizeof(unsigned long long) * (1u << g.nbits), hipMemcpyDeviceToHost, st);
    if (err == hipSuccess) err = hipStreamSynchronize(st);
    if (err != hipSuccess) {
      rc = fail(GSS_EHIP, "percentile pass %d: %s", pass, hipGetErrorString(err));
      break;
    }
    unsigned long long cum = 0;
    int chosen = -1;
    for (int b = 0; b < (1 << g.nbits); ++b) {
      if (cum + h_hist[b] > rank) {
        chosen = b;
        break;
      }
      cum += h_hist[b];
    }
    if (chosen < 0) {
      rc = fail(GSS_EHIP, "percentile: rank %llu not found in pass %d (NaN in the embeddings?)", rank, pass);
      break;
    }
    rank -= cum;
    eq_count = h_hist[chosen];
    g.prefix |= (uint32_t)chosen << g.shift;
    g.mask |= ((1u << g.nbits) - 1u) << g.shift;
  }
  if (rc == GSS_OK) {
    const uint32_t key_lo = g.prefix;
    uint32_t key_hi = key_lo;
    // `rank` is now the 0-based position of the wanted element inside its final (single-value) bin
    if (hi != lo && rank + 1 >= eq_count) {
      const uint32_t init = 0xffffffffu;
      hipError_t err = hipMemcpyAsync(d_min, &init, 4, hipMemcpyHostToDevice, st);
      g.floor_key = key_lo;
      if (err == hipSuccess) {
        hipLaunchKernelGGL((pct_kernel<1>), grid, block, 0, st, g);
        err = hipGetLastError();
      }
      uint32_t got = init;
      if (err == hipSuccess) err = hipMemcpyAsync(&got, d_min, 4, hipMemcpyDeviceToHost, st);
      if (err == hipSuccess) err = hipStreamSynchronize(st);
      if (err != hipSuccess)
        rc = fail(GSS_EHIP, "percentile min pass: %s", hipGetErrorString(err));
      else if (got != init)
        key_hi = got;
    }
    if (rc == GSS_OK) {
      const float a = key_to_float(key_lo), b = key_to_float(key_hi);
      *h_out = (float)((double)a + ((double)b - (double)a) * frac);  // numpy 'linear' interpolation
    }
  }
  (void)hipFree(d_hist);
  return rc;
}
