// ppr.hip -- diffusion profiles: personalised PageRank from every drug / indication at once.
//
// Replaces multiscale/diff_prof/diffusion_profiles.py:30-90 (one scipy power iteration per start node, a process pool
// over the 2,502 start nodes) by ONE batched iteration on the device: the start nodes are the columns of an fp64
// matrix x[n][kpad], the matrix-vector product becomes an SpMM over a shared transition matrix and everything that
// differs per start node (see include/gssgcn.h) is applied to x around the product.
//
// Per iteration:  column sums over the empty rows (dangling mass) / scale the override entries -> y = M'^T x (SpMM) ->
// restore x -> add the start nodes' own rows -> x_new = alpha (y + dangling e_s) + (1 - alpha) e_s, column L1 errors,
// freeze converged columns.
//
// SpMM: fp64, HBM/cache-bound gather like spmm.hip.  A lane moves one double2 (16 B, the same access width as the fp32
// kernel's float4), 16 lanes cover a 256-B slice of a row, a wave works on 4 segments (<= 32 stored entries each) at a
// time.  The 20-KB rows are walked in 512-B column chunks, time-separated (grid.y is the slow dispatch dimension), each
// chunk as two 256-B slices pinned to XCD parity -- the configuration that won for the 15-MB fp32 table in spmm.hip
// (one chunk of this matrix is exactly that table).  Partial sums meet in fixed order: bitwise reproducible.
#include <algorithm>
#include <utility>
#include <vector>

#include "ops.h"

struct gss_ppr {
  gss_ppr_desc d;
  gss_csr *csr;       // structure handle for the segment schedule
  char *slab;
  size_t slab_bytes;
  std::vector<size_t> guard_off;   // slab offsets of the guards behind the carved buffers
  double *y;          // [n][kpad]
  double *stash_o;    // [n_ovr]
  double *part;       // [max(z chunks, row blocks)][kpad] column partial sums
  double *dsum;       // [kpad]
  double *yself;      // [kpad]
  int32_t *done;      // [kpad]
  int32_t *iters;     // [kpad]
  int32_t *n_active;  // [1]
  int32_t n_rowblocks, n_zchunks;
  // fused update (gss_ppr_desc.ovr_ptr / sel_ptr given): the SpMM's epilogue writes the next iterate and the column errors
  double *ysel;       // [n][kpad], zero except the start nodes' "selected" rows: ysel[sel_row][c] = sel_val x[start[c]][c], rewritten every iteration
  uint32_t *rowflag;  // [n bits]: row has an entry in ysel
  double *part_e;     // [segment blocks][kpad]: per workgroup and column, sum of |x_new - x| over the rows the workgroup wrote
  double *err_corr;   // [kpad]: correction of the column errors for the override entries (x was scaled in place during the product)
  int32_t n_segblocks;
  bool fused;
};

namespace gss {

constexpr int kPprRows = 64;     // rows per block of the column-wise passes
constexpr int kPprThreads = 1024;
constexpr int kPprWaves = kPprThreads / 64;

__device__ __forceinline__ double shfl_f64(double v, int src) {
  const int lo = __shfl(__double2loint(v), src, 64), hi = __shfl(__double2hiint(v), src, 64);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double shfl_xor_f64(double v, int mask) {
  const int lo = __shfl_xor(__double2loint(v), mask, 64), hi = __shfl_xor(__double2hiint(v), mask, 64);
  return __hiloint2double(hi, lo);
}

// y[row][chunk] = sum_e val[e] * x[col[e]][chunk] for the segments of one workgroup; kpad doubles per row
// Fused update (gss_ppr_run with grouped override / start-row lists): instead of y the kernel writes the NEXT iterate
//   x_new = alpha (y + ysel + dangling e_s) + (1 - alpha) e_s      (diffusion_profiles.py:84; the start node's own entry takes yself)
// into the other half of a double buffer -- converged columns are copied across --, and per workgroup and column the sum of
// |x_new - x| over the rows it wrote, reduced in a fixed order (lane groups by shuffles, waves through LDS).
struct PprFuse {
  const double *ysel;
  const uint32_t *rowflag;
  const int32_t *start, *iters;
  const double *dsum, *yself;
  double *part_e;
  double alpha;
  int k, it;
};

__device__ __forceinline__ double2 ppr_fuse_row(const PprFuse &f, int row, double2 acc, const double *__restrict__ x_cur, double *__restrict__ x_nxt,
                                                 size_t off, int sA, int sB, bool dA, bool dB, double dsA, double dsB, double ysA, double ysB) {
  if ((f.rowflag[(unsigned)row >> 5] >> (row & 31)) & 1u) {
    const double2 s = *reinterpret_cast<const double2 *>(f.ysel + off);
    acc.x += s.x;
    acc.y += s.y;
  }
  const double2 xo = *reinterpret_cast<const double2 *>(x_cur + off);
  double2 xn, err = make_double2(0.0, 0.0);
  {
    const double p = row == sA ? 1.0 : 0.0;
    const double yv = row == sA ? ysA : acc.x;
    xn.x = f.alpha * (yv + dsA * p) + (1.0 - f.alpha) * p;
  }
  {
    const double p = row == sB ? 1.0 : 0.0;
    const double yv = row == sB ? ysB : acc.y;
    xn.y = f.alpha * (yv + dsB * p) + (1.0 - f.alpha) * p;
  }
  if (dA) xn.x = xo.x; else err.x = fabs(xn.x - xo.x);
  if (dB) xn.y = xo.y; else err.y = fabs(xn.y - xo.y);
  *reinterpret_cast<double2 *>(x_nxt + off) = xn;
  return err;
}

template <bool NARROW, bool FUSED = false>
__global__ __launch_bounds__(kPprThreads) void ppr_spmm_kernel(const int32_t *__restrict__ col, const double *__restrict__ val,
                                                               const int4 *__restrict__ segs, const double *__restrict__ x,
                                                               double *__restrict__ y, int kpad, const int32_t *__restrict__ done, PprFuse f) {
  __shared__ double2 part[kPprWaves * 16];
  __shared__ double2 errs[kPprWaves * 16];
  constexpr int LPR = 16, LOG = 4, GPW = 4;
  const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
  const int g = lane >> LOG, li = lane & (LPR - 1);
  const int sblk = blockIdx.x >> 1, half = blockIdx.x & 1;
  // gss_ppr_run: columns converge independently and are frozen (ppr_update_kernel leaves them alone); a workgroup whose 32 columns
  // have all converged has nothing left to produce (padding columns count as converged from the start).  Workgroup-uniform.
  if (done) {
    const int c0 = blockIdx.y * 64 + half * 32;
    if (FUSED) {
      // double-buffered x: a converged column is copied across once more in the iteration after it converged (both halves of the
      // buffer then hold its final values); from the iteration after that its group may be skipped (padding columns: iters = -1)
      const int ic = f.iters[c0 + (lane & 31)];
      if (__all(ic != 0 && ic <= f.it - 2)) return;
    } else if (__all(done[c0 + (lane & 31)] != 0)) {
      return;
    }
  }
  const int4 sd = segs[((size_t)sblk * kPprWaves + wib) * GPW + g];  // {row, first entry, entries, flags | log2 p}
  const int row = sd.x;
  const int e0 = sd.y, e1 = sd.y + sd.z;
  const int plog = sd.w & 0xff;
  const bool multiwave = (sd.w & 0x100) != 0;
  const size_t coff = (size_t)blockIdx.y * 64 + half * 32 + li * 2;  // first of this lane's two columns
  const double *xs = x + coff;
  double2 acc = make_double2(0.0, 0.0);
  // the (col, val) pairs of the next LPR entries are requested before the gathers of the current ones
  int c_next = 0;
  double w_next = 0.0;
  if (e0 + li < e1) {
    c_next = col[e0 + li];
    w_next = val[e0 + li];
  }
  for (int base = e0; __any(base < e1); base += LPR) {
    const int c = c_next;
    const double w = w_next;
    const int ne = base + LPR + li;
    c_next = 0;
    w_next = 0.0;
    if (ne < e1) {
      c_next = col[ne];
      w_next = val[ne];
    }
    const int cnt = min(LPR, e1 - base);
    constexpr int kFly = 4;
    for (int t = 0; __any(t < cnt); t += kFly) {
      double2 xv[kFly];
      double wv[kFly];
#pragma unroll
      for (int u = 0; u < kFly; ++u) {
        const int src = (g << LOG) + t + u;  // t + u <= 15: never leaves the group
        const int cc = __shfl(c, src, 64);
        wv[u] = shfl_f64(w, src);
        const bool ok = t + u < cnt;
        if (!ok) wv[u] = 0.0;
        if (NARROW) {  // matrix < 4 GB, < 2^24 rows: 32-bit byte offsets, one 24-bit multiply-add per gather
          const unsigned off = __umul24((unsigned)cc, (unsigned)kpad * 8u);
          xv[u] = ok ? *reinterpret_cast<const double2 *>(reinterpret_cast<const char *>(xs) + off) : make_double2(0.0, 0.0);
        } else {
          xv[u] = ok ? *reinterpret_cast<const double2 *>(xs + (size_t)cc * kpad) : make_double2(0.0, 0.0);
        }
      }
#pragma unroll
      for (int u = 0; u < kFly; ++u) {
        acc.x = fma(wv[u], xv[u].x, acc.x);
        acc.y = fma(wv[u], xv[u].y, acc.y);
      }
    }
  }
  const int pcount = 1 << plog;
#pragma unroll
  for (int o = 1; o < GPW; o <<= 1) {
    const double tx = shfl_xor_f64(acc.x, o * LPR), ty = shfl_xor_f64(acc.y, o * LPR);
    if (pcount > o) {
      acc.x += tx;
      acc.y += ty;
    }
  }
  // this lane's two columns in the fused update
  int sA = -1, sB = -1;
  bool dA = true, dB = true;
  double dsA = 0.0, dsB = 0.0, ysA = 0.0, ysB = 0.0;
  double2 err = make_double2(0.0, 0.0);
  if (FUSED) {
    const int cA = (int)coff, cB = (int)coff + 1;
    dA = done[cA] != 0;
    dB = done[cB] != 0;
    if (cA < f.k) {
      sA = f.start[cA];
      dsA = f.dsum[cA];
      ysA = f.yself[cA];
    }
    if (cB < f.k) {
      sB = f.start[cB];
      dsB = f.dsum[cB];
      ysB = f.yself[cB];
    }
  }
  if (pcount <= GPW && row >= 0 && (g & (pcount - 1)) == 0) {
    const size_t off = (size_t)row * kpad + coff;
    if (FUSED)
      err = ppr_fuse_row(f, row, acc, x, y, off, sA, sB, dA, dB, dsA, dsB, ysA, ysB);
    else
      *reinterpret_cast<double2 *>(y + off) = acc;
  }
  if (!FUSED && !multiwave) return;
  const int nw = pcount > GPW ? pcount / GPW : 1;
  if (multiwave) {   // workgroup-uniform
    if (nw > 1 && g == 0) part[wib * 16 + li] = acc;
    __syncthreads();
    if (nw > 1 && (wib & (nw - 1)) == 0 && g == 0 && row >= 0) {
      double2 t = part[wib * 16 + li];
      for (int k = 1; k < nw; ++k) {
        const double2 q = part[(wib + k) * 16 + li];
        t.x += q.x;
        t.y += q.y;
      }
      const size_t off = (size_t)row * kpad + coff;
      if (FUSED)
        err = ppr_fuse_row(f, row, t, x, y, off, sA, sB, dA, dB, dsA, dsB, ysA, ysB);
      else
        *reinterpret_cast<double2 *>(y + off) = t;
    }
  }
  if (!FUSED) return;
  // column errors of this workgroup: the 4 lane groups of a wave by shuffles, the 16 waves through LDS, always in the same order
#pragma unroll
  for (int o = 1; o < GPW; o <<= 1) {
    err.x += shfl_xor_f64(err.x, o * LPR);
    err.y += shfl_xor_f64(err.y, o * LPR);
  }
  if (g == 0) errs[wib * 16 + li] = err;
  __syncthreads();
  if (wib == 0 && g == 0) {
    double2 t = errs[li];
    for (int k = 1; k < kPprWaves; ++k) {
      t.x += errs[k * 16 + li].x;
      t.y += errs[k * 16 + li].y;
    }
    *reinterpret_cast<double2 *>(f.part_e + (size_t)sblk * kpad + coff) = t;
  }
}

// sum of part[0..m)[c] in index order (the loads of a batch are issued together, the adds stay sequential)
__device__ __forceinline__ double column_sum(const double *__restrict__ part, int m, int kpad, int c) {
  double s = 0.0;
  int j = 0;
  for (; j + 16 <= m; j += 16) {
    double v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = part[(size_t)(j + u) * kpad + c];
#pragma unroll
    for (int u = 0; u < 16; ++u) s += v[u];
  }
  for (; j < m; ++j) s += part[(size_t)j * kpad + c];
  return s;
}

__global__ void ppr_init_kernel(double *x, size_t total, int kpad, int k, double v, int32_t *done, int32_t *iters) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < total) x[i] = (int)(i % kpad) < k ? v : 0.0;
  if (i < (size_t)kpad) {
    done[i] = (int)i < k ? 0 : 1;
    iters[i] = (int)i < k ? 0 : -1;   // padding columns: converged "before the first iteration" (the fused kernel's skip rule)
  }
}

// empty rows of M' (sinks, isolated nodes): what sits there is dangling, except a start node's own entry in its column.
// Column sums per chunk of kPprRows such rows.
__global__ __launch_bounds__(256) void ppr_dangling_kernel(const double *__restrict__ x, const int32_t *__restrict__ z_rows, int n_z,
                                                           const int32_t *__restrict__ start, int k, int kpad, double *__restrict__ part) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= kpad) return;
  const int s = c < k ? start[c] : -1;
  const int t0 = blockIdx.y * kPprRows, t1 = min(n_z, t0 + kPprRows);
  double sum = 0.0;
  for (int t = t0; t < t1; ++t) {
    const int r = z_rows[t];
    const double v = x[(size_t)r * kpad + c];
    if (r != s) sum += v;
  }
  part[(size_t)blockIdx.y * kpad + c] = sum;
}

__global__ void ppr_ovr_scale_kernel(double *__restrict__ x, const int32_t *__restrict__ ovr_col, const int32_t *__restrict__ ovr_row,
                                     const double *__restrict__ ratio, long n_ovr, int kpad, double *__restrict__ stash_o) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_ovr) return;
  const size_t i = (size_t)ovr_row[e] * kpad + ovr_col[e];
  const double v = x[i];
  stash_o[e] = v;
  x[i] = v * ratio[e];
}

// per column: dangling sum (chunks in order, then the zero-ratio overrides, then the start node if its row is empty)
// and the surviving in-edges of the start node
__global__ void ppr_column_kernel(const double *__restrict__ x, const double *__restrict__ part, int n_chunks, const double *__restrict__ stash_o,
                                  const int32_t *__restrict__ zero_ptr, const int32_t *__restrict__ zero_ovr,
                                  const int32_t *__restrict__ start, const int32_t *__restrict__ start_dangling,
                                  const int32_t *__restrict__ keep_ptr, const int32_t *__restrict__ keep_row, const double *__restrict__ keep_val,
                                  int k, int kpad, double *__restrict__ dsum, double *__restrict__ yself) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= k) return;
  double s = column_sum(part, n_chunks, kpad, c);
  for (int e = zero_ptr[c]; e < zero_ptr[c + 1]; ++e) s += stash_o[zero_ovr[e]];
  if (start_dangling[c]) s += x[(size_t)start[c] * kpad + c];
  dsum[c] = s;
  double ys = 0.0;
  for (int e = keep_ptr[c]; e < keep_ptr[c + 1]; ++e) ys += keep_val[e] * x[(size_t)keep_row[e] * kpad + c];
  yself[c] = ys;
}

__global__ void ppr_ovr_restore_kernel(double *__restrict__ x, const int32_t *__restrict__ ovr_col, const int32_t *__restrict__ ovr_row, long n_ovr,
                                       int kpad, const double *__restrict__ stash_o) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < n_ovr) x[(size_t)ovr_row[e] * kpad + ovr_col[e]] = stash_o[e];
}

// the start node's own row in its "selected" form: y[row][c] += val * x[start[c]][c]  ((row, c) pairs are unique)
__global__ void ppr_sel_kernel(double *__restrict__ y, const double *__restrict__ x, const int32_t *__restrict__ sel_col,
                               const int32_t *__restrict__ sel_row, const double *__restrict__ sel_val, long n_sel,
                               const int32_t *__restrict__ start, int kpad) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_sel) return;
  const int c = sel_col[e];
  y[(size_t)sel_row[e] * kpad + c] += sel_val[e] * x[(size_t)start[c] * kpad + c];
}

// x_new = alpha (y + dangling e_s) + (1 - alpha) e_s  (diffusion_profiles.py:84), |x_new - x| summed per column over this
// block's rows; columns that already converged keep their x
__global__ __launch_bounds__(256) void ppr_update_kernel(double *__restrict__ x, const double *__restrict__ y, const int32_t *__restrict__ start,
                                                         const double *__restrict__ dsum, const double *__restrict__ yself,
                                                         const int32_t *__restrict__ done, int n, int k, int kpad, double alpha,
                                                         double *__restrict__ part) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= kpad) return;
  const int j0 = blockIdx.y * kPprRows, j1 = min(n, j0 + kPprRows);
  double err = 0.0;
  if (c < k && !done[c]) {
    const int s = start[c];
    const double ds = dsum[c], ys = yself[c];
    for (int j = j0; j < j1; ++j) {
      const size_t i = (size_t)j * kpad + c;
      const double p = j == s ? 1.0 : 0.0;
      const double yv = j == s ? ys : y[i];
      const double xn = alpha * (yv + ds * p) + (1.0 - alpha) * p;
      err += fabs(xn - x[i]);
      x[i] = xn;
    }
  }
  part[(size_t)blockIdx.y * kpad + c] = err;
}

// fused update: the start nodes' "selected" rows as a sparse addend of the product, ysel[row][c] = val * x[start[c]][c]  (x unscaled:
// this runs before ppr_ovr_scale_kernel)
__global__ void ppr_ysel_kernel(double *__restrict__ ysel, const double *__restrict__ x, const int32_t *__restrict__ sel_col,
                                const int32_t *__restrict__ sel_row, const double *__restrict__ sel_val, long n_sel,
                                const int32_t *__restrict__ start, int kpad) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_sel) return;
  const int c = sel_col[e];
  ysel[(size_t)sel_row[e] * kpad + c] = sel_val[e] * x[(size_t)start[c] * kpad + c];
}
__global__ void ppr_rowflag_kernel(uint32_t *__restrict__ flag, const int32_t *__restrict__ sel_row, long n_sel) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < n_sel) atomicOr(&flag[(unsigned)sel_row[e] >> 5], 1u << (sel_row[e] & 31));
}

// fused update, after the product: the override entries of x were scaled IN PLACE while the fused epilogue read x as "the previous
// iterate".  One wave per column walks the column's overrides (grouped by column: ovr_ptr) and repairs what that touched: a converged
// column's copy takes the true value; for the others the column error gets |x_new - true| - |x_new - scaled|, summed over the
// entries in a fixed order (lanes stride the list, shuffle tree).
__global__ __launch_bounds__(256) void ppr_ovr_fix_kernel(double *__restrict__ x_nxt, const int32_t *__restrict__ ovr_ptr,
                                                          const int32_t *__restrict__ ovr_row, const double *__restrict__ ratio,
                                                          const double *__restrict__ stash_o, const int32_t *__restrict__ done, int k, int kpad,
                                                          double *__restrict__ err_corr) {
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= k) return;
  const bool dn = done[c] != 0;
  double corr = 0.0;
  for (int e = ovr_ptr[c] + lane; e < ovr_ptr[c + 1]; e += 64) {
    const size_t i = (size_t)ovr_row[e] * kpad + c;
    const double tru = stash_o[e], scaled = tru * ratio[e];
    if (dn) {
      x_nxt[i] = tru;
    } else {
      const double xn = x_nxt[i];
      corr += fabs(xn - tru) - fabs(xn - scaled);
    }
  }
  corr = wave_sum_d(corr);
  if (lane == 0) err_corr[c] = corr;
}

__global__ void ppr_finish_fused_kernel(const double *__restrict__ part_e, int n_blocks, const double *__restrict__ err_corr, int k, int kpad,
                                        double thr, int it, int32_t *__restrict__ done, int32_t *__restrict__ iters, int32_t *__restrict__ n_active) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= k || done[c]) return;
  const double err = column_sum(part_e, n_blocks, kpad, c) + err_corr[c];
  if (err < thr) {
    done[c] = 1;
    iters[c] = it;
  } else {
    atomicAdd(n_active, 1);
  }
}

__global__ void ppr_finish_kernel(const double *__restrict__ part, int n_blocks, int k, int kpad, double thr, int it, int32_t *__restrict__ done,
                                  int32_t *__restrict__ iters, int32_t *__restrict__ n_active) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= k || done[c]) return;
  const double err = column_sum(part, n_blocks, kpad, c);
  if (err < thr) {
    done[c] = 1;
    iters[c] = it;
  } else {
    atomicAdd(n_active, 1);
  }
}

static int ppr_spmm_launch(gss_ppr *p, const double *x, double *y, hipStream_t st, const int32_t *done = nullptr, const PprFuse *fuse = nullptr) {
  const int4 *segs = nullptr;
  int nblk = 0;
  if (int rc = csr_segments(p->csr, 2, &segs, &nblk)) return rc;
  if (nblk == 0) return GSS_OK;
  GSS_REQUIRE(!fuse || nblk == p->n_segblocks, "ppr: the segment schedule changed under a fused plan (%d blocks, %d expected)", nblk, p->n_segblocks);
  const bool narrow = (double)p->d.n * p->d.kpad * 8.0 < 4.0e9 && p->d.n < (1 << 24) && (int64_t)p->d.kpad * 8 < (1 << 24);
  const dim3 grid(nblk * 2, p->d.kpad / 64), block(kPprThreads);
  const PprFuse none{};
  if (fuse) {
    if (narrow)
      hipLaunchKernelGGL((ppr_spmm_kernel<true, true>), grid, block, 0, st, p->d.t_col, p->d.t_val, segs, x, y, p->d.kpad, done, *fuse);
    else
      hipLaunchKernelGGL((ppr_spmm_kernel<false, true>), grid, block, 0, st, p->d.t_col, p->d.t_val, segs, x, y, p->d.kpad, done, *fuse);
  } else if (narrow) {
    hipLaunchKernelGGL((ppr_spmm_kernel<true, false>), grid, block, 0, st, p->d.t_col, p->d.t_val, segs, x, y, p->d.kpad, done, none);
  } else {
    hipLaunchKernelGGL((ppr_spmm_kernel<false, false>), grid, block, 0, st, p->d.t_col, p->d.t_val, segs, x, y, p->d.kpad, done, none);
  }
  GSS_LAUNCH_CHECK("ppr_spmm_kernel");
  return GSS_OK;
}

}  // namespace gss

using namespace gss;

extern "C" {

int gss_ppr_create(gss_ppr **out, const gss_ppr_desc *desc) {
  GSS_REQUIRE(out && desc, "ppr_create: null argument");
  const gss_ppr_desc &D = *desc;
  GSS_REQUIRE(D.n > 0 && D.k > 0 && D.kpad >= D.k && D.kpad % 64 == 0, "ppr_create: bad sizes n=%d k=%d kpad=%d (kpad: multiple of 64, >= k)",
              D.n, D.k, D.kpad);
  GSS_REQUIRE(D.h_rowptr && D.t_rowptr && D.t_col && D.t_val && D.start && D.start_dangling && D.zero_ptr && D.keep_ptr,
              "ppr_create: null operand");
  GSS_REQUIRE(D.n_z >= 0 && (D.n_z == 0 || D.z_rows) && D.n_ovr >= 0 && (D.n_ovr == 0 || (D.ovr_col && D.ovr_row && D.ovr_ratio)) &&
                  D.n_sel >= 0 && (D.n_sel == 0 || (D.sel_col && D.sel_row && D.sel_val)),
              "ppr_create: empty-row / override / start-row lists missing");
  GSS_REQUIRE((double)D.n * D.kpad < 9.0e18 / 8, "ppr_create: matrix too large");
  gss_ppr *p = new gss_ppr();
  p->d = D;
  p->csr = nullptr;
  p->slab = nullptr;
  // structure-only CSR handle for the segment schedule (its value pointer is never read as float)
  if (int rc = gss_csr_create(&p->csr, D.n, D.n, D.nnz, D.h_rowptr, D.t_rowptr, D.t_col, reinterpret_cast<const float *>(D.t_val))) {
    delete p;
    return rc;
  }
  p->n_rowblocks = ceil_div(D.n, kPprRows);
  p->n_zchunks = ceil_div(D.n_z, kPprRows);
  // fused update: needs the overrides grouped by column (ovr_ptr); knob "ppr_fused" = 0 keeps the separate update pass (A/B)
  p->fused = K().ppr_fused != 0 && (D.n_ovr == 0 || D.ovr_ptr != nullptr);
  p->n_segblocks = 0;
  p->ysel = p->part_e = p->err_corr = nullptr;
  p->rowflag = nullptr;
  if (p->fused) {
    const int4 *segs = nullptr;
    if (int rc = csr_segments(p->csr, 2, &segs, &p->n_segblocks)) {
      gss_csr_destroy(p->csr);
      delete p;
      return rc;
    }
  }
  const size_t row = (size_t)D.kpad * sizeof(double);
  auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
  const size_t b_y = al((size_t)D.n * row), b_so = al((size_t)D.n_ovr * sizeof(double)),
               b_part = al((size_t)std::max(p->n_rowblocks, std::max(p->n_zchunks, 1)) * row), b_col = al(row),
               b_int = al((size_t)D.kpad * sizeof(int32_t));
  const size_t b_flag = al(((size_t)D.n / 32 + 1) * sizeof(uint32_t)), b_pe = al((size_t)std::max(p->n_segblocks, 1) * row);
  // every carved buffer is followed by a 256-byte guard (as in plan.hip): gss_ppr_check_guards verifies that no kernel wrote one
  constexpr size_t kGuard = 256;
  p->slab_bytes = b_y + b_so + b_part + 2 * b_col + 2 * b_int + 256 + 8 * kGuard + (p->fused ? b_y + b_flag + b_pe + b_col + 4 * kGuard : 0);
  if (hipMalloc((void **)&p->slab, p->slab_bytes) != hipSuccess) {
    gss_csr_destroy(p->csr);
    const size_t want = p->slab_bytes;
    delete p;
    return fail(GSS_ENOMEM, "ppr_create: hipMalloc of %zu bytes failed", want);
  }
  char *q = p->slab;
  auto take = [&](size_t b) {
    char *r = q;
    q += b;
    p->guard_off.push_back((size_t)(q - p->slab));
    q += kGuard;
    return r;
  };
  p->y = (double *)take(b_y);
  p->stash_o = (double *)take(b_so);
  p->part = (double *)take(b_part);
  p->dsum = (double *)take(b_col);
  p->yself = (double *)take(b_col);
  p->done = (int32_t *)take(b_int);
  p->iters = (int32_t *)take(b_int);
  p->n_active = (int32_t *)take(256);
  hipError_t ge = hipSuccess;
  if (p->fused) {
    p->ysel = (double *)take(b_y);
    p->rowflag = (uint32_t *)take(b_flag);
    p->part_e = (double *)take(b_pe);
    p->err_corr = (double *)take(b_col);
    ge = hipMemset(p->ysel, 0, b_y);
    if (ge == hipSuccess) ge = hipMemset(p->rowflag, 0, b_flag);
    if (ge == hipSuccess) ge = hipMemset(p->part_e, 0, b_pe);
    if (ge == hipSuccess) ge = hipMemset(p->err_corr, 0, b_col);
    if (ge == hipSuccess && D.n_sel > 0) {
      hipLaunchKernelGGL(ppr_rowflag_kernel, dim3(ceil_div(D.n_sel, 256)), dim3(256), 0, nullptr, p->rowflag, D.sel_row, (long)D.n_sel);
      ge = hipGetLastError();
    }
  }
  for (size_t g : p->guard_off)
    if (ge == hipSuccess) ge = hipMemset(p->slab + g, 0xA5, kGuard);
  if (ge == hipSuccess) ge = hipDeviceSynchronize();
  if (ge != hipSuccess) {
    gss_csr_destroy(p->csr);
    (void)hipFree(p->slab);
    delete p;
    return fail(GSS_EHIP, "ppr_create: guard fill -> %s", hipGetErrorString(ge));
  }
  *out = p;
  return GSS_OK;
}

void gss_ppr_destroy(gss_ppr *p) {
  if (!p) return;
  if (p->csr) gss_csr_destroy(p->csr);
  if (p->slab) (void)hipFree(p->slab);
  delete p;
}

size_t gss_ppr_device_bytes(const gss_ppr *p) { return p ? p->slab_bytes : 0; }

int gss_ppr_check_guards(gss_ppr *p) {
  GSS_REQUIRE(p, "ppr_check_guards: null handle");
  GSS_HIP(hipDeviceSynchronize());
  unsigned char host[256];
  for (size_t k = 0; k < p->guard_off.size(); ++k) {
    GSS_HIP(hipMemcpy(host, p->slab + p->guard_off[k], sizeof(host), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < sizeof(host); ++i)
      if (host[i] != 0xA5)
        return fail(GSS_EINVAL, "ppr_check_guards: the guard behind carved buffer %zu (slab offset %zu) was overwritten at byte %zu", k,
                    p->guard_off[k], i);
  }
  return GSS_OK;
}

int gss_ppr_spmm(gss_ppr *p, const double *x, double *y, void *stream) {
  GSS_REQUIRE(p && x && y, "ppr_spmm: null argument");
  return ppr_spmm_launch(p, x, y, as_stream(stream));
}

int gss_ppr_run(gss_ppr *p, double alpha, double tol, int32_t max_iter, double *x, int32_t *iters_out, void *stream) {
  GSS_REQUIRE(p && x && iters_out, "ppr_run: null argument");
  GSS_REQUIRE(max_iter >= 1 && alpha >= 0.0 && alpha <= 1.0 && tol > 0.0, "ppr_run: bad alpha=%g tol=%g max_iter=%d", alpha, tol, max_iter);
  const gss_ppr_desc &D = p->d;
  hipStream_t st = as_stream(stream);
  const size_t total = (size_t)D.n * D.kpad;
  hipLaunchKernelGGL(ppr_init_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, x, total, D.kpad, D.k, 1.0 / D.n, p->done, p->iters);
  GSS_LAUNCH_CHECK("ppr_init_kernel");
  const dim3 col_tiles(ceil_div(D.kpad, 256));
  const double thr = (double)D.n * tol;  // diffusion_profiles.py:87
  int32_t active = D.k;
  int it = 0;
  // fused: x is double-buffered (the caller's buffer and p->y swap roles every iteration); `cur` holds the current iterate
  double *cur = x, *nxt = p->y;
  while (active > 0 && it < max_iter) {
    ++it;
    if (D.n_z > 0) {
      hipLaunchKernelGGL(ppr_dangling_kernel, dim3(col_tiles.x, p->n_zchunks), dim3(256), 0, st, cur, D.z_rows, D.n_z, D.start, D.k, D.kpad, p->part);
      GSS_LAUNCH_CHECK("ppr_dangling_kernel");
    }
    if (p->fused && D.n_sel > 0) {   // before the overrides are scaled: it reads the start nodes' own entries
      hipLaunchKernelGGL(ppr_ysel_kernel, dim3(ceil_div(D.n_sel, 256)), dim3(256), 0, st, p->ysel, cur, D.sel_col, D.sel_row, D.sel_val, (long)D.n_sel,
                         D.start, D.kpad);
      GSS_LAUNCH_CHECK("ppr_ysel_kernel");
    }
    if (D.n_ovr > 0) {
      hipLaunchKernelGGL(ppr_ovr_scale_kernel, dim3(ceil_div(D.n_ovr, 256)), dim3(256), 0, st, cur, D.ovr_col, D.ovr_row, D.ovr_ratio, (long)D.n_ovr,
                         D.kpad, p->stash_o);
      GSS_LAUNCH_CHECK("ppr_ovr_scale_kernel");
    }
    hipLaunchKernelGGL(ppr_column_kernel, dim3(ceil_div(D.k, 64)), dim3(64), 0, st, cur, p->part, D.n_z > 0 ? p->n_zchunks : 0, p->stash_o,
                       D.zero_ptr, D.zero_ovr, D.start, D.start_dangling, D.keep_ptr, D.keep_row, D.keep_val, D.k, D.kpad, p->dsum, p->yself);
    GSS_LAUNCH_CHECK("ppr_column_kernel");
    if (p->fused) {
      const PprFuse f{p->ysel, p->rowflag, D.start, p->iters, p->dsum, p->yself, p->part_e, alpha, D.k, it};
      if (int rc = ppr_spmm_launch(p, cur, nxt, st, p->done, &f)) return rc;   // writes the next iterate and the column errors
    } else {
      if (int rc = ppr_spmm_launch(p, cur, p->y, st, p->done)) return rc;   // converged 32-column groups are skipped
    }
    if (D.n_ovr > 0) {
      hipLaunchKernelGGL(ppr_ovr_restore_kernel, dim3(ceil_div(D.n_ovr, 256)), dim3(256), 0, st, cur, D.ovr_col, D.ovr_row, (long)D.n_ovr, D.kpad,
                         p->stash_o);
      GSS_LAUNCH_CHECK("ppr_ovr_restore_kernel");
    }
    GSS_HIP(hipMemsetAsync(p->n_active, 0, sizeof(int32_t), st));
    if (p->fused) {
      if (D.n_ovr > 0) {
        hipLaunchKernelGGL(ppr_ovr_fix_kernel, dim3(ceil_div(D.k, 4)), dim3(256), 0, st, nxt, D.ovr_ptr, D.ovr_row, D.ovr_ratio, p->stash_o, p->done, D.k,
                           D.kpad, p->err_corr);
        GSS_LAUNCH_CHECK("ppr_ovr_fix_kernel");
      }
      hipLaunchKernelGGL(ppr_finish_fused_kernel, dim3(ceil_div(D.k, 64)), dim3(64), 0, st, p->part_e, p->n_segblocks, p->err_corr, D.k, D.kpad, thr, it,
                         p->done, p->iters, p->n_active);
      GSS_LAUNCH_CHECK("ppr_finish_fused_kernel");
      std::swap(cur, nxt);
    } else {
      if (D.n_sel > 0) {
        hipLaunchKernelGGL(ppr_sel_kernel, dim3(ceil_div(D.n_sel, 256)), dim3(256), 0, st, p->y, cur, D.sel_col, D.sel_row, D.sel_val, (long)D.n_sel,
                           D.start, D.kpad);
        GSS_LAUNCH_CHECK("ppr_sel_kernel");
      }
      hipLaunchKernelGGL(ppr_update_kernel, dim3(col_tiles.x, p->n_rowblocks), dim3(256), 0, st, cur, p->y, D.start, p->dsum, p->yself, p->done, D.n, D.k,
                         D.kpad, alpha, p->part);
      GSS_LAUNCH_CHECK("ppr_update_kernel");
      hipLaunchKernelGGL(ppr_finish_kernel, dim3(ceil_div(D.k, 64)), dim3(64), 0, st, p->part, p->n_rowblocks, D.k, D.kpad, thr, it, p->done, p->iters,
                         p->n_active);
      GSS_LAUNCH_CHECK("ppr_finish_kernel");
    }
    GSS_HIP(hipMemcpyAsync(&active, p->n_active, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    GSS_HIP(hipStreamSynchronize(st));
  }
  // the last iterate, in the caller's buffer: the k real columns only -- a group made of padding columns alone is skipped from the first
  // iteration on, so the handle's buffer never wrote them; the caller's padding keeps the zeros ppr_init put there
  if (cur != x)
    GSS_HIP(hipMemcpy2DAsync(x, (size_t)D.kpad * sizeof(double), cur, (size_t)D.kpad * sizeof(double), (size_t)D.k * sizeof(double), (size_t)D.n,
                             hipMemcpyDeviceToDevice, st));
  GSS_HIP(hipMemcpyAsync(iters_out, p->iters, (size_t)D.k * sizeof(int32_t), hipMemcpyDeviceToHost, st));
  GSS_HIP(hipStreamSynchronize(st));
  if (active > 0) return fail(GSS_ENOTCONV, "ppr_run: %d of %d columns did not converge in %d iterations", active, D.k, max_iter);
  return GSS_OK;
}

}  // extern "C"
