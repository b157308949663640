// knn.hip -- top-k inner-product neighbours for the "descriptor" kNN graph.
//
// Replaces the host side of gen_graph, helpers/helper.py:39-44:  x_sim = X^T X (dense N x N fp64, 7.2 GB at
// N = 29,960) followed by np.argpartition(x_sim, -k, 1)[:, -k:].  Here the similarity tiles are produced on the
// fp64 matrix cores (v_mfma_f64_16x16x4_f64 -- the reference does this product in fp64 and the selection must not
// depend on fp32 rounding) and never leave the CU: a workgroup owns 64 rows, sweeps the columns in 64-wide tiles,
// parks each 64 x 64 tile in LDS, and one lane per row folds its 64 candidates into a running top-k (unordered set
// with a tracked minimum -- the same set argpartition returns; ties are arbitrary there too).
// Output: top_val / top_idx [N][k]; symmetrising and CSR assembly stay in graph.py (index bookkeeping).
//
// Roofline: fp64 MFMA, N^2 d flops... 2 N^2 d = 230 GFLOP at config 2; one-time setup.
#include "ops.h"

namespace gss {

typedef double f64x4 __attribute__((ext_vector_type(4)));

constexpr int kKnnMaxK = 64;

// rows [row_lo, row_hi) against all n columns; top_val / top_idx hold those rows only (a rank of a sharded job computes its own window:
// a row's MFMA chain and its column sweep do not depend on the window, so the table is the same whoever computes it)
__global__ __launch_bounds__(256) void knn_topk_kernel(int n, int d, const double *__restrict__ x, int k, int row_lo, int row_hi,
                                                       double *__restrict__ top_val, int32_t *__restrict__ top_idx) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double *tile = reinterpret_cast<double *>(smem);        // [64][65] similarity tile (padded)
  double *tv = tile + 64 * 65;                            // [64][k]
  int *ti = reinterpret_cast<int *>(tv + 64 * k);         // [64][k]
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = lane & 15, q = lane >> 4;
  const int i0 = row_lo + blockIdx.x * 64;
  // running top-k state of row (i0 + threadIdx.x) for threads 0..63
  double minv = -INFINITY;
  int minp = 0, filled = 0;
  // wave w computes S rows [16 w, 16 w + 16) x 64 columns: A = X_i rows (held), B = X_j rows
  const double *xi = x + (size_t)min(n - 1, i0 + 16 * w + c) * d;
  for (int j0 = 0; j0 < n; j0 += 64) {
    f64x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = (f64x4){0.0, 0.0, 0.0, 0.0};
    const double *xj[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) xj[t] = x + (size_t)min(n - 1, j0 + 16 * t + c) * d;
    for (int kc = 0; kc < d; kc += 8) {
      // lane (c, q) supplies k = kc + 2 q and kc + 2 q + 1 (two MFMA steps per 16-byte load; same order both sides)
      const double2 a2 = *reinterpret_cast<const double2 *>(xi + kc + 2 * q);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const double2 b2 = *reinterpret_cast<const double2 *>(xj[t] + kc + 2 * q);
        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2.x, b2.x, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2.y, b2.y, acc[t], 0, 0, 0);
      }
    }
    // f64 C/D layout: col = lane & 15, row = (lane >> 4) + 4 * reg   (D[row = i][col = j])
    __syncthreads();  // previous tile fully consumed
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) tile[(16 * w + q + 4 * r) * 65 + 16 * t + c] = acc[t][r];
    __syncthreads();
    if (threadIdx.x < 64 && i0 + threadIdx.x < row_hi) {
      const double *row = tile + threadIdx.x * 65;
      double *mv = tv + threadIdx.x * k;
      int *mi = ti + threadIdx.x * k;
      const int jmax = min(64, n - j0);
      for (int j = 0; j < jmax; ++j) {
        const double v = row[j];
        if (filled < k) {
          mv[filled] = v;
          mi[filled] = j0 + j;
          ++filled;
          if (filled == k) {
            minv = mv[0];
            minp = 0;
            for (int p = 1; p < k; ++p)
              if (mv[p] < minv) {
                minv = mv[p];
                minp = p;
              }
          }
        } else if (v > minv) {
          mv[minp] = v;
          mi[minp] = j0 + j;
          minv = mv[0];
          minp = 0;
          for (int p = 1; p < k; ++p)
            if (mv[p] < minv) {
              minv = mv[p];
              minp = p;
            }
        }
      }
    }
  }
  __syncthreads();
  if (threadIdx.x < 64 && i0 + threadIdx.x < row_hi) {
    const size_t o = (size_t)(i0 + threadIdx.x - row_lo) * k;
    for (int p = 0; p < k; ++p) {
      top_val[o + p] = p < filled ? tv[threadIdx.x * k + p] : -INFINITY;
      top_idx[o + p] = p < filled ? ti[threadIdx.x * k + p] : -1;
    }
  }
}

}  // namespace gss

using namespace gss;

static int knn_topk_rows(int32_t n, int32_t d, const double *x, int32_t k, int32_t row_lo, int32_t row_hi, double *top_val, int32_t *top_idx,
                         void *stream) {
  GSS_REQUIRE(n > 0 && x && top_val && top_idx, "knn_topk: null operand");
  GSS_REQUIRE(d >= 8 && d % 8 == 0 && d <= 4096, "knn_topk: d=%d must be a multiple of 8 in [8, 4096]", d);
  GSS_REQUIRE(k >= 1 && k <= kKnnMaxK && k <= n, "knn_topk: k=%d out of [1, min(%d, n)]", k, kKnnMaxK);
  GSS_REQUIRE(row_lo >= 0 && row_lo <= row_hi && row_hi <= n, "knn_topk: rows [%d, %d) out of [0, %d]", row_lo, row_hi, n);
  if (row_hi == row_lo) return GSS_OK;
  const size_t lds = sizeof(double) * 64 * 65 + (sizeof(double) + sizeof(int)) * 64 * (size_t)k;
  hipLaunchKernelGGL(knn_topk_kernel, dim3(ceil_div(row_hi - row_lo, 64)), dim3(256), lds, as_stream(stream), n, d, x, k, row_lo, row_hi, top_val,
                     top_idx);
  GSS_LAUNCH_CHECK("knn_topk_kernel");
  return GSS_OK;
}

extern "C" int gss_knn_topk(int32_t n, int32_t d, const double *x, int32_t k, double *top_val, int32_t *top_idx, void *stream) {
  return knn_topk_rows(n, d, x, k, 0, n, top_val, top_idx, stream);
}

extern "C" int gss_knn_topk_rows(int32_t n, int32_t d, const double *x, int32_t k, int32_t row_lo, int32_t row_hi, double *top_val,
                                 int32_t *top_idx, void *stream) {
  return knn_topk_rows(n, d, x, k, row_lo, row_hi, top_val, top_idx, stream);
}
