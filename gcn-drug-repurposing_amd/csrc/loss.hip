// loss.hip -- GSS similarity loss, forward and backward fused (K6/K7 of SURVEY.md section 2b).
//
// Replaces GSS_loss.gss_loss (modules/model.py:214-221) and its autograd:
//   E_B = e[idx];  S = E_B E_B^T;  loss = mean(-alpha/2 (relu(S) - beta)^2)
//   dE_B[i] = sum_j (G_ij + G_ji) E_B[j] = 2 sum_j G_ij E_B[j],   G = -alpha/B^2 (relu(S)-beta) 1[S>0]
// (S is symmetric bit for bit: S_ij and S_ji are the same fmaf chain over k.)
//
// Flash-style: the B x B matrices S and G are never written.  A wave owns a 16-row i-tile and walks
// j-tiles: S'^T tile = E_j E_i^T on v_mfma_f32_16x16x4_f32 (d/4 MFMAs), G elementwise in registers,
// then dE_i += G E_j (d/4 MFMAs).  The accumulator layout of the first product (lane = i, register =
// j & 3, lane quarter = j >> 2) is exactly the B-operand layout of the second, so G never leaves the
// registers.  A workgroup's 4 waves split the j range, several workgroups (grid.y) split it further;
// partial dE tiles are tree-reduced in LDS and the grid.y partials are summed by loss_finish_kernel.
//
// Roofline: MFMA fp32.  flops = 2 B^2 d (S) + 2 B^2 d (G E) = 4 B^2 d, S tiles computed once.
#include <algorithm>

#include "ops.h"

namespace gss {

__device__ __forceinline__ f32x4 mfma16l(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

constexpr int kLossWaves = 4;

struct LossArgs {
  int d, b, js;  // js = grid.y
  const float *e;  // contiguous E_B [b][d] (gathered by gather_rows_kernel)
  float beta, alpha;
  float *de_part;     // [js][b][d]
  double *loss_part;  // [grid.x * grid.y]
  int tile0, tile_stride;   // workgroup x of the grid takes i tile tile0 + x * tile_stride (a rank's row slab of the sweep; 0 / 1: all tiles)
};

// E_B = e[idx] (modules/model.py:216-217) into a contiguous buffer, so the MFMA loop has no index indirection
// keep != NULL (a shard of a node-range sharded plan): rows this shard does not own are written as zeros, so that
// the all-reduce over the shards assembles E_B
__global__ __launch_bounds__(256) void gather_rows_kernel(int b, int d4, const float *__restrict__ e, const int32_t *__restrict__ idx,
                                                          const float *__restrict__ keep, float *__restrict__ out) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)b * d4) return;
  const int r = (int)(i / d4), f4 = (int)(i % d4);
  const bool mine = !keep || keep[r] != 0.f;
  st4(out + i * 4, mine ? ld4(e + ((size_t)idx[r] * d4 + f4) * 4) : make_float4(0.f, 0.f, 0.f, 0.f));
}

// the same gather for a relabelled graph and/or one node-range shard, with the batch-id translation folded in
// (it used to be a launch of its own): node id -> row (node_map), row -> local row of
// this shard (lo, nl), the position-map id (gid2op, or the local row); the thread of a row's first float4 stores the translated ids
struct BatchMap {
  const int32_t *idx, *node_map, *gid2op;
  int lo, nl;
  int32_t *pid, *rloc;
  float *keep;  // NULL on one GPU
};
__global__ __launch_bounds__(256) void gather_rows_mapped_kernel(int b, int d4, const float *__restrict__ e, BatchMap m, float *__restrict__ out) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)b * d4) return;
  const int r = (int)(i / d4), f4 = (int)(i % d4);
  const int id = m.node_map ? m.node_map[m.idx[r]] : m.idx[r];
  const int rel = id - m.lo;
  const bool mine = rel >= 0 && rel < m.nl;
  if (f4 == 0) {
    m.rloc[r] = min(max(rel, 0), max(m.nl - 1, 0));
    if (m.keep) m.keep[r] = mine ? 1.f : 0.f;
    m.pid[r] = m.gid2op ? m.gid2op[id] : (mine ? rel : -1);
  }
  st4(out + i * 4, mine ? ld4(e + ((size_t)rel * d4 + f4) * 4) : make_float4(0.f, 0.f, 0.f, 0.f));
}

// A shard's contribution to the ONE batch collective of a step (round 4): [E_B | P_B | inv_B] -- the embedding row, the top layer's
// pre-activation row and 1 / ||x|| of every member this shard owns, zeros for the others -- so that after one all-reduce (every element
// has exactly one non-zero contributor: the sum is the owner's value, bit for bit) every rank holds what the finish needs for EVERY
// member and computes the batch rows' input gradient locally: the second all-reduce of a step (the [2 B][d] input gradients) is gone.
// m.idx != NULL: the id translation of gather_rows_mapped_kernel rides along; else rows / keep are the prepared ones.
__global__ __launch_bounds__(256) void gather_batch_kernel(int b, int d4, const float *__restrict__ e, const float *__restrict__ p,
                                                           const float *__restrict__ inv_den, BatchMap m, const int32_t *__restrict__ rows,
                                                           const float *__restrict__ keep, float *__restrict__ out) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)b * d4) return;
  const int r = (int)(i / d4), f4 = (int)(i % d4);
  int rel;
  bool mine;
  if (m.idx) {
    const int id = m.node_map ? m.node_map[m.idx[r]] : m.idx[r];
    rel = id - m.lo;
    mine = rel >= 0 && rel < m.nl;
    if (f4 == 0) {
      m.rloc[r] = min(max(rel, 0), max(m.nl - 1, 0));
      if (m.keep) m.keep[r] = mine ? 1.f : 0.f;
      m.pid[r] = m.gid2op ? m.gid2op[id] : (mine ? rel : -1);
    }
  } else {
    rel = rows[r];
    mine = !keep || keep[r] != 0.f;
  }
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  const size_t src = ((size_t)rel * d4 + f4) * 4;
  st4(out + i * 4, mine ? ld4(e + src) : z);
  st4(out + ((size_t)b * d4 + i) * 4, mine ? ld4(p + src) : z);
  if (f4 == 0) out[(size_t)2 * b * d4 * 4 + r] = mine ? inv_den[rel] : 0.f;
}

// One batch row of the finish (shared by loss_finish_bwd_kernel and loss_finish_dgrad_kernel -- the same lanes in the same order, so the two
// forms give the same bits): lane li of the row's lane group (lpr lanes) holds float4s li + 64 k.
//   de = 2 sum_js de_part; dot = e . de; dx = (de - e dot) * inv; dp = c * dx (.) elu'(p)
// dp_lds (nullable): the row of dP the input-gradient MFMAs of loss_finish_dgrad_kernel read (unmasked when dgrad_all)
template <int VPL>
__device__ __forceinline__ void finish_row(bool ok, int r, int li, int lpr, int d4, int js, int b, const float *__restrict__ de_part,
                                           const float *__restrict__ erow, float inv_u, float keepv, const float *__restrict__ prow, float c,
                                           float *__restrict__ dx_b, float *__restrict__ dp_b, float *dp_lds, bool dgrad_all) {
  float4 g[VPL], ev[VPL], pv[VPL];
  float dot = 0.f;
#pragma unroll
  for (int k = 0; k < VPL; ++k) {
    const int f4 = li + k * 64;
    const bool in = ok && f4 < d4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (in) {
      const size_t off = ((size_t)r * d4 + f4) * 4;
      s = ld4(de_part + off);
      for (int t = 1; t < js; ++t) s = add4(s, ld4(de_part + (size_t)t * b * d4 * 4 + off));
      s = scale4(2.f, s);
    }
    g[k] = s;
    ev[k] = in ? ld4(erow + (size_t)f4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    pv[k] = in ? ld4(prow + (size_t)f4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);   // (with the other operands: one round trip, not two)
    dot += g[k].x * ev[k].x + g[k].y * ev[k].y + g[k].z * ev[k].z + g[k].w * ev[k].w;
  }
  for (int o = 1; o < lpr; o <<= 1) dot += __shfl_xor(dot, o, 64);
  if (!ok) return;
  const float inv = keepv != 0.f ? inv_u : 0.f;
#pragma unroll
  for (int k = 0; k < VPL; ++k) {
    const int f4 = li + k * 64;
    if (f4 >= d4) continue;
    float4 dx;
    dx.x = (g[k].x - ev[k].x * dot) * inv;
    dx.y = (g[k].y - ev[k].y * dot) * inv;
    dx.z = (g[k].z - ev[k].z * dot) * inv;
    dx.w = (g[k].w - ev[k].w * dot) * inv;
    st4(dx_b + ((size_t)r * d4 + f4) * 4, dx);
    const float4 pg = elu_grad4(pv[k]);
    const float4 dp = scale4(c, mul4(dx, pg));
    st4(dp_b + ((size_t)r * d4 + f4) * 4, dp);
    if (dp_lds) {
      float4 dl = dp;
      if (dgrad_all && keepv == 0.f) {   // a member another shard owns: its gradient row all the same (this shard holds its p / inv_den)
        float4 du;
        du.x = (g[k].x - ev[k].x * dot) * inv_u;
        du.y = (g[k].y - ev[k].y * dot) * inv_u;
        du.z = (g[k].z - ev[k].z * dot) * inv_u;
        du.w = (g[k].w - ev[k].w * dot) * inv_u;
        dl = scale4(c, mul4(du, pg));
      }
      *reinterpret_cast<float4 *>(dp_lds + (size_t)f4 * 4) = dl;
    }
  }
}

// EXACT: d == 64 NG, no feature masking anywhere.  From d = 256 on the operand fragments + accumulators need more than the 256
// registers two workgroups per CU leave a lane (372 B/lane of scratch at NG = 4): one workgroup per CU with the full 512
// (accumulators in AGPRs), and at NG = 4 no register prefetch of the next j tile, measured at B = 2048, d = 256:
// 123 us -> 55 us for gather + sweep + finish (tools/loss_prof.py 256).
// Measured dead ends at d = 128, B = 2048 (gather + sweep + finish 31.6 us, tools/loss_prof.py): 8 waves per workgroup 33.2 us; two j
// tiles per trip with two operand register sets that swap roles (no copies, every wait a full tile behind its load in the ISA) 34.7 us;
// the second product's operand loads removed altogether (wrong results, timing only) 30.6 us -- the sweep is bound by neither its
// loads nor their waits; 1024 MFMAs per SIMD are 13.7 us at 2.4 GHz and ~16.5 us at the ~2.0 GHz an MFMA-saturated loop sustains
// (tools/micro/mfma_lds.hip: 126-131 of 157 TFLOP/s), the rest is the fixed start (i-tile fragments) and end (wave tree, partial store).
template <int NG, bool EXACT, bool PF = (NG <= 2), int OCC = (NG >= 4 ? 1 : 2)>
__global__ __launch_bounds__(64 * kLossWaves, OCC) void loss_fused_kernel(LossArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float4 *red = reinterpret_cast<float4 *>(smem);  // [2 slots][NG*4 tiles][64 lanes]
  __shared__ double lsum[kLossWaves];
  const int lane = threadIdx.x & 63;
  const int w = threadIdx.x >> 6;
  const int c = lane & 15, q = lane >> 4;
  const int d = g.d, B = g.b;
  const int fbase = blockIdx.z * (64 * NG);
  const int i0 = (g.tile0 + (int)blockIdx.x * g.tile_stride) * 16;
  const int nj = (B + 15) / 16;
  const int nslots = kLossWaves * g.js;
  const int slot = blockIdx.y * kLossWaves + w;
  const float coef = -g.alpha / ((float)B * (float)B);
  const float beta = g.beta;

  const float *ei = g.e + (size_t)min(B - 1, i0 + c) * d + 4 * q;
  const bool i_ok = (i0 + c) < B;
  // the i-tile's operand fragments stay in registers for the whole j sweep when d <= 256
  constexpr bool HOLD_I = NG <= 4;
  constexpr int NCH = NG * 4;  // 16-wide k chunks covered by the template (d <= 64 NG)
  float4 bi[HOLD_I ? NCH : 1];
  if (HOLD_I) {
#pragma unroll
    for (int k = 0; k < NCH; ++k) bi[k] = (EXACT || 16 * k < d) ? ld4(ei + 16 * k) : make_float4(0.f, 0.f, 0.f, 0.f);
  }

  f32x4 acc[NG][4];
#pragma unroll
  for (int G = 0; G < NG; ++G)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[G][e] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float lacc = 0.f;

  float4 aj[HOLD_I ? NCH : 1];
  if (HOLD_I && PF && slot < nj) {
    const float *ej0 = g.e + (size_t)min(B - 1, slot * 16 + c) * d + 4 * q;
#pragma unroll
    for (int k = 0; k < NCH; ++k) aj[k] = (EXACT || 16 * k < d) ? ld4(ej0 + 16 * k) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int jt = slot; jt < nj; jt += nslots) {
    const int j0 = jt * 16;
    // ---- S'[j][i] = E_j . E_i  (A = E_j rows, B = E_i rows), k order kc + 4 q + e; two accumulators
    // break the dependent MFMA chain; the next j-tile's operands are requested before this tile's MFMAs
    const float *ej = g.e + (size_t)min(B - 1, j0 + c) * d + 4 * q;
    // operands of the second product (rows j0 + 4 q + r of E_B, features fbase + 64 G + 4 c ..): requested now,
    // branch-free, so their latency hides under the 4 NCH MFMAs of the first product
    float4 p2[4][NG];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float *er = g.e + (size_t)min(B - 1, j0 + 4 * q + r) * d;
#pragma unroll
      for (int G = 0; G < NG; ++G) {
        const int f = fbase + 64 * G + 4 * c;
        if (EXACT) {
          p2[r][G] = ld4(er + f);
        } else {
          const bool in = f < d;
          const float4 v = ld4(er + (in ? f : 0));
          p2[r][G] = in ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
    }
    f32x4 s0 = (f32x4){0.f, 0.f, 0.f, 0.f}, s1 = s0;
    if (HOLD_I && !PF) {
#pragma unroll
      for (int k = 0; k < NCH; ++k) aj[k] = (EXACT || 16 * k < d) ? ld4(ej + 16 * k) : make_float4(0.f, 0.f, 0.f, 0.f);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < NCH; ++k) {
        s0 = mfma16l(aj[k].x, bi[k].x, s0);
        s1 = mfma16l(aj[k].y, bi[k].y, s1);
        s0 = mfma16l(aj[k].z, bi[k].z, s0);
        s1 = mfma16l(aj[k].w, bi[k].w, s1);
      }
    } else if (HOLD_I) {
      float4 an[NCH];
      const int jn = jt + nslots;
      const float *ejn = g.e + (size_t)min(B - 1, (jn < nj ? jn : jt) * 16 + c) * d + 4 * q;
#pragma unroll
      for (int k = 0; k < NCH; ++k) an[k] = (EXACT || 16 * k < d) ? ld4(ejn + 16 * k) : make_float4(0.f, 0.f, 0.f, 0.f);
      __builtin_amdgcn_sched_barrier(0);  // hipcc otherwise sinks every load next to its first use (one exposed
                                          // round trip per 4 MFMAs); issue them all here, consume them later
#pragma unroll
      for (int k = 0; k < NCH; ++k) {
        s0 = mfma16l(aj[k].x, bi[k].x, s0);
        s1 = mfma16l(aj[k].y, bi[k].y, s1);
        s0 = mfma16l(aj[k].z, bi[k].z, s0);
        s1 = mfma16l(aj[k].w, bi[k].w, s1);
      }
#pragma unroll
      for (int k = 0; k < NCH; ++k) aj[k] = an[k];
    } else {
      for (int kc = 0; kc < d; kc += 16) {
        const float4 a4 = ld4(ej + kc);
        const float4 b4 = ld4(ei + kc);
        s0 = mfma16l(a4.x, b4.x, s0);
        s1 = mfma16l(a4.y, b4.y, s1);
        s0 = mfma16l(a4.z, b4.z, s0);
        s1 = mfma16l(a4.w, b4.w, s1);
      }
    }
    const f32x4 s = s0 + s1;
    // lane (c, q), reg r: S[i0 + c][j0 + 4 q + r]
    float gv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool ok = i_ok && (j0 + 4 * q + r) < B;
      const float sv = s[r];
      const float t = fmaxf(sv, 0.f) - beta;
      lacc += ok ? t * t : 0.f;
      gv[r] = (ok && sv > 0.f) ? coef * t : 0.f;
    }
    // ---- dE_i[f] += sum_j G[i][j] E_j[f]:  A = E_j^T (feature rows), B = G, k slot q at step r is j0+4q+r
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int G = 0; G < NG; ++G) {
        const float4 a4 = p2[r][G];
        acc[G][0] = mfma16l(a4.x, gv[r], acc[G][0]);
        acc[G][1] = mfma16l(a4.y, gv[r], acc[G][1]);
        acc[G][2] = mfma16l(a4.z, gv[r], acc[G][2]);
        acc[G][3] = mfma16l(a4.w, gv[r], acc[G][3]);
      }
    }
  }

  // ---- tree-reduce the partial dE tiles of the 4 waves (fixed order)
  for (int half = kLossWaves / 2; half >= 1; half >>= 1) {
    if (w >= half && w < 2 * half) {
      float4 *dst = red + (size_t)(w - half) * NG * 4 * 64;
#pragma unroll
      for (int G = 0; G < NG; ++G)
#pragma unroll
        for (int e = 0; e < 4; ++e) dst[(G * 4 + e) * 64 + lane] = make_float4(acc[G][e][0], acc[G][e][1], acc[G][e][2], acc[G][e][3]);
    }
    __syncthreads();
    if (w < half) {
      const float4 *src = red + (size_t)w * NG * 4 * 64;
#pragma unroll
      for (int G = 0; G < NG; ++G)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float4 v = src[(G * 4 + e) * 64 + lane];
          acc[G][e][0] += v.x;
          acc[G][e][1] += v.y;
          acc[G][e][2] += v.z;
          acc[G][e][3] += v.w;
        }
    }
    __syncthreads();
  }
  // tile (G, e), lane (c, q), reg r: node i0 + c, feature fbase + 64 G + 4 (4 q + r) + e
  if (w == 0 && i_ok) {
    float *out = g.de_part + ((size_t)blockIdx.y * B + (i0 + c)) * d;
#pragma unroll
    for (int G = 0; G < NG; ++G)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int f = fbase + 64 * G + 16 * q + 4 * r;
        if (EXACT || f < d) st4(out + f, make_float4(acc[G][0][r], acc[G][1][r], acc[G][2][r], acc[G][3][r]));
      }
  }
  // ---- loss partial (only the z == 0 slab counts it)
  const double ws = wave_sum_d((double)lacc);
  if (lane == 0) lsum[w] = ws;
  __syncthreads();
  if (threadIdx.x == 0 && blockIdx.z == 0) {
    double t = 0.0;
    for (int k = 0; k < kLossWaves; ++k) t += lsum[k];
    g.loss_part[blockIdx.y * gridDim.x + blockIdx.x] = t;
  }
}

// de_b = 2 * sum_js de_part[js] ; loss = -alpha/2 / B^2 * sum(loss_part)
__global__ __launch_bounds__(256) void loss_finish_kernel(int b, int d, int js, int nloss, const float *__restrict__ de_part,
                                                           const double *__restrict__ loss_part, float alpha,
                                                           float *__restrict__ de_b, float *__restrict__ loss_out) {
  const size_t n4 = (size_t)b * d / 4;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n4) {
    float4 s = ld4(de_part + i * 4);
    for (int k = 1; k < js; ++k) s = add4(s, ld4(de_part + ((size_t)k * b * d) + i * 4));
    st4(de_b + i * 4, scale4(2.f, s));
  }
  if (blockIdx.x == 0 && threadIdx.x < 64) {
    double t = 0.0;
    for (int k = threadIdx.x; k < nloss; k += 64) t += loss_part[k];
    t = wave_sum_d(t);
    if (threadIdx.x == 0) loss_out[0] = (float)(-0.5 * (double)alpha * t / ((double)b * (double)b));
  }
}

// Row-slab sweep (a shard took the i tiles tile0, tile0 + stride, ...): de_x = sum_js de_part on the rows of those tiles, zeros on the
// others, and this rank's share of the loss behind the last row -- the buffer the ranks then sum (every element has one non-zero
// contributor, the loss P of them).  The factor 2 of dE = 2 G E is applied by the finish, as in the replicated form.
__global__ __launch_bounds__(256) void loss_slab_sum_kernel(int b, int d, int js, int nloss, int tile0, int tile_stride,
                                                             const float *__restrict__ de_part, const double *__restrict__ loss_part,
                                                             float alpha, float *__restrict__ de_x) {
  const size_t n4 = (size_t)b * d / 4;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n4) {
    const int tile = (int)(i * 4 / d) / 16;
    const bool mine = tile >= tile0 && (tile - tile0) % tile_stride == 0;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (mine) {
      s = ld4(de_part + i * 4);
      for (int k = 1; k < js; ++k) s = add4(s, ld4(de_part + ((size_t)k * b * d) + i * 4));
    }
    st4(de_x + i * 4, s);
  }
  if (blockIdx.x == 0 && threadIdx.x < 64) {
    double t = 0.0;
    for (int k = threadIdx.x; k < nloss; k += 64) t += loss_part[k];
    t = wave_sum_d(t);
    if (threadIdx.x == 0) de_x[(size_t)b * d] = (float)(-0.5 * (double)alpha * t / ((double)b * (double)b));
  }
}

// plan path: loss_finish + the backward of F.normalize and F.elu on the batch rows in one launch (the widths
// loss_finish_dgrad_kernel does not cover).
// One lane group per batch row: de = 2 sum_js de_part (the row of dLoss/dE_B), then
// dx = (de - e (e . de)) * inv_den[node], dp = c * dx (.) elu'(p[node]); e rows come from the gathered E_B, or (erows != NULL) from
// the embedding matrix through the row list.
template <int VPL>
__global__ __launch_bounds__(256) void loss_finish_bwd_kernel(int b, int d4, int lpr_log2, int js, int nloss,
                                                              const float *__restrict__ de_part, const double *__restrict__ loss_part,
                                                              float alpha, const float *__restrict__ e_b, const int32_t *__restrict__ erows,
                                                              const int32_t *__restrict__ idx,
                                                              const int32_t *__restrict__ pos_ids, const float *__restrict__ keep,
                                                              const float *__restrict__ inv_den, const float *__restrict__ p, float c,
                                                              float *__restrict__ dx_b, float *__restrict__ dp_b,
                                                              int32_t *__restrict__ pos_set, float *__restrict__ loss_out) {
  const int lane = threadIdx.x & 63;
  const int lpr = 1 << lpr_log2;
  const int rpw = 64 >> lpr_log2;
  const int li = lane & (lpr - 1);
  const int r = (blockIdx.x * 4 + (threadIdx.x >> 6)) * rpw + (lane >> lpr_log2);
  const bool ok = r < b;
  const int rc = ok ? r : 0;
  const int node = idx ? idx[rc] : rc;   // idx == NULL: inv_den / p are per member (a shard's gathered batch)
  if (nloss > 0 && blockIdx.x == 0 && threadIdx.x < 64) {   // (nloss == 0: the loss came with the summed de, row-slab sweep)
    double t = 0.0;
    for (int k = threadIdx.x; k < nloss; k += 64) t += loss_part[k];
    t = wave_sum_d(t);
    if (threadIdx.x == 0) loss_out[0] = (float)(-0.5 * (double)alpha * t / ((double)b * (double)b));
  }
  // idx: row of inv_den / p (this shard's local row); pos_ids: the id the batch-position map is keyed by (the same array on
  // one GPU, the padded global id on a shard); keep[r] == 0: another shard owns the row -> zero gradient here
  if (ok && pos_set && pos_ids && li == 0 && pos_ids[r] >= 0) pos_set[pos_ids[r]] = r;
  const float *erow = e_b + (size_t)(erows ? erows[rc] : rc) * d4 * 4;
  finish_row<VPL>(ok, r, li, lpr, d4, js, b, de_part, erow, inv_den[node], keep ? keep[rc] : 1.f, p + (size_t)node * d4 * 4, c, dx_b, dp_b, nullptr,
                  false);
}

// finish + the batch rows' input gradient in ONE launch (round 4; d in {64, 128, 256}): what used to be loss_finish_bwd_kernel followed
// by the batch-row launch of gemm_nt_lds_kernel<.., EPI_SPLIT> -- two ~5 us launches, each a short chain of dependent global round
// trips (~1 us apiece on this machine) around < 1 us of work.  One workgroup per 16 batch rows: the 4 waves finish the rows (finish_row:
// loss_finish_bwd_kernel's lanes and order), park the 16 x d tile of dP in LDS, and multiply it with [W1 ; W2] (pre-transposed): wave w
// takes output features [w d / 2, (w + 1) d / 2) of the 2 d, its weight fragments requested in the SAME round trip as the rows'
// operands (NG <= 2: into registers).  The k order (chunk by chunk, e = 0..3 inside) is gemm_nt_lds_kernel's, so [g_ax | g_am] keeps
// the bits of the stand-alone launch.  (Folding all of this into the sweep itself -- the last-arriving workgroup of an i tile as
// finisher -- was built and measured slower: profiles/r04_loss_fold_ab.txt.)
struct FinishDgrad {
  int b, d, js, nloss;
  const float *de_part;
  const double *loss_part;
  float alpha;
  const float *e_b;          // [b][d] gathered batch rows
  const int32_t *rows;       // row of inv_den / p per member (NULL: the member's position -- a shard's gathered batch)
  const int32_t *pos_ids;    // key of the batch-position map per member (nullable with pos_set)
  int32_t *pos_set;          // nullable
  const float *keep;         // nullable: 0 where another shard owns the member (its dx_b / dp_b rows are written as zeros)
  const float *inv_den, *p;
  float c;
  float *dx_b, *dp_b, *loss_out;
  const float *w1t, *w2t;
  float *gax_b, *gam_b;
  int dgrad_all;             // the input gradient of EVERY member (a shard that holds p / inv_den of the whole batch), else of the kept ones
};

template <int NG>
__global__ __launch_bounds__(256) void loss_finish_dgrad_kernel(FinishDgrad T) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float *dp_tile = reinterpret_cast<float *>(smem);   // [16][d + 4]: rows 16 B apart in the banks (the 16 rows of a fragment read do not collide)
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = lane & 15, q = lane >> 4;
  const int d = T.d, B = T.b, d4 = d / 4, ds = d + 4;
  const int i0 = blockIdx.x * 16;
  int lg = 2;
  while ((1 << lg) < d4 && lg < 6) ++lg;
  const int lpr = 1 << lg, rpw = 64 >> lg;
  const int li = lane & (lpr - 1);
  constexpr int RP = 4;                              // rows of the tile per wave at most (4 waves, rpw >= 1 rows per pass)
  constexpr int NTW = 2 * NG;                        // 16-feature blocks per wave: (2 d / 4) / 16 with d = 64 NG
  constexpr int NCHK = 4 * NG;                       // 16-wide k chunks
  constexpr bool WREG = NG <= 2;                     // a wave's weight fragments in registers (32 float4 at d = 128)
  const int jw0 = w * 32 * NG;
  const bool hi = jw0 >= d;
  const float *wt = (hi ? T.w2t : T.w1t) + (size_t)((hi ? jw0 - d : jw0) + c) * d + 4 * q;
  float4 wf[WREG ? NCHK : 1][WREG ? NTW : 1];
  if (WREG) {
#pragma unroll
    for (int k = 0; k < NCHK; ++k)
#pragma unroll
      for (int u = 0; u < NTW; ++u) wf[k][u] = ld4(wt + (size_t)(16 * u) * d + 16 * k);
  }
  if (T.nloss > 0 && blockIdx.x == 0 && threadIdx.x < 64) {   // (nloss == 0: the loss came with the summed de, row-slab sweep)
    double t = 0.0;
    for (int k = threadIdx.x; k < T.nloss; k += 64) t += T.loss_part[k];
    t = wave_sum_d(t);
    if (threadIdx.x == 0) T.loss_out[0] = (float)(-0.5 * (double)T.alpha * t / ((double)B * (double)B));
  }
  // The rows of this wave (RP at most), in three phases so that every load of the launch is in flight before the first store (a store
  // between two loads orders them for the compiler; a dependent round trip costs ~1 us here): (1) the members' rows / keep flags,
  // (2) all operands of all rows, (3) arithmetic -- finish_row's, in its order -- and stores.
  int rrs[RP], nodes[RP];
  float keeps[RP];
#pragma unroll
  for (int k = 0; k < RP; ++k) {
    rrs[k] = w * rpw + (lane >> lg) + k * 4 * rpw;   // (the lanes of a row's group share rr: the shuffles below stay whole)
    const int rc = min(i0 + min(rrs[k], 15), B - 1);
    nodes[k] = T.rows ? T.rows[rc] : rc;
    keeps[k] = T.keep ? T.keep[rc] : 1.f;
  }
  float4 g[RP], ev[RP], pv[RP];
  float invs[RP];
  const bool col = li < d4;                          // (d4 <= 64 here: one float4 of a row per lane)
#pragma unroll
  for (int k = 0; k < RP; ++k) {
    const int r = i0 + rrs[k];
    const bool in = rrs[k] < 16 && r < B && col;
    const int rc = min(i0 + min(rrs[k], 15), B - 1);
    float4 sgm = make_float4(0.f, 0.f, 0.f, 0.f);
    if (in) {
      const size_t off = ((size_t)r * d4 + li) * 4;
      sgm = ld4(T.de_part + off);
      for (int t = 1; t < T.js; ++t) sgm = add4(sgm, ld4(T.de_part + (size_t)t * B * d + off));
      sgm = scale4(2.f, sgm);
    }
    g[k] = sgm;
    ev[k] = in ? ld4(T.e_b + (size_t)rc * d + (size_t)li * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    pv[k] = in ? ld4(T.p + (size_t)nodes[k] * d + (size_t)li * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    invs[k] = T.inv_den[nodes[k]];
  }
#pragma unroll
  for (int k = 0; k < RP; ++k) {
    const int rr = rrs[k];
    if (rr < 16) {
      const int r = i0 + rr;
      const bool ok = r < B;
      float dot = g[k].x * ev[k].x + g[k].y * ev[k].y + g[k].z * ev[k].z + g[k].w * ev[k].w;
      for (int o = 1; o < lpr; o <<= 1) dot += __shfl_xor(dot, o, 64);
      float *dl = dp_tile + (size_t)rr * ds + (size_t)li * 4;
      if (ok && col) {
        if (T.pos_set && li == 0) {
          const int key = T.pos_ids ? T.pos_ids[r] : nodes[k];
          if (key >= 0) T.pos_set[key] = r;
        }
        const float inv = keeps[k] != 0.f ? invs[k] : 0.f;
        float4 dx;
        dx.x = (g[k].x - ev[k].x * dot) * inv;
        dx.y = (g[k].y - ev[k].y * dot) * inv;
        dx.z = (g[k].z - ev[k].z * dot) * inv;
        dx.w = (g[k].w - ev[k].w * dot) * inv;
        st4(T.dx_b + ((size_t)r * d4 + li) * 4, dx);
        const float4 pg = elu_grad4(pv[k]);
        const float4 dp = scale4(T.c, mul4(dx, pg));
        st4(T.dp_b + ((size_t)r * d4 + li) * 4, dp);
        float4 dlv = dp;
        if (T.dgrad_all && keeps[k] == 0.f) {   // a member another shard owns: its gradient row all the same (this shard holds its p / inv_den)
          float4 du;
          du.x = (g[k].x - ev[k].x * dot) * invs[k];
          du.y = (g[k].y - ev[k].y * dot) * invs[k];
          du.z = (g[k].z - ev[k].z * dot) * invs[k];
          du.w = (g[k].w - ev[k].w * dot) * invs[k];
          dlv = scale4(T.c, mul4(du, pg));
        }
        *reinterpret_cast<float4 *>(dl) = dlv;
      } else if (col) {
        *reinterpret_cast<float4 *>(dl) = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  }
  __syncthreads();
  const float *brow = dp_tile + (size_t)c * ds + 4 * q;
  f32x4 o[NTW];
#pragma unroll
  for (int u = 0; u < NTW; ++u) o[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < NCHK; ++k) {
    const float4 bv = *reinterpret_cast<const float4 *>(brow + 16 * k);
    float4 av[NTW];
#pragma unroll
    for (int u = 0; u < NTW; ++u) av[u] = WREG ? wf[k][u] : ld4(wt + (size_t)(16 * u) * d + 16 * k);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float bs = e == 0 ? bv.x : e == 1 ? bv.y : e == 2 ? bv.z : bv.w;
#pragma unroll
      for (int u = 0; u < NTW; ++u) {
        const float as = e == 0 ? av[u].x : e == 1 ? av[u].y : e == 2 ? av[u].z : av[u].w;
        o[u] = mfma16l(as, bs, o[u]);
      }
    }
  }
  if (i0 + c < B) {
    float *out = (hi ? T.gam_b : T.gax_b) + (size_t)(i0 + c) * d + (hi ? jw0 - d : jw0) + 4 * q;
#pragma unroll
    for (int u = 0; u < NTW; ++u) st4(out + 16 * u, make_float4(o[u][0], o[u][1], o[u][2], o[u][3]));
  }
}

// debug knob "loss_wgs": workgroups the (i-tile x j-split) grid aims at.  One workgroup per CU wins at B = 2048 (gather + sweep + finish,
// d = 128: 128 / 256 / 384 / 512 / 768 / 1024 workgroups -> 49.4 / 31.4 / 38.3 / 34.6 / 39.8 / 43.9 us; d = 256: 124.0 at 256 vs 126.3 at 512)

// parts > 1: the i tiles are dealt to `parts` ranks (row-slab sweep); js then follows the tiles ONE rank sweeps -- the same on every rank
static void loss_geometry(int32_t b, int32_t d, int &ni, int &js, int &nz, int &ng, int parts = 1) {
  ni = ceil_div(b, 16);
  const int nj = ni;
  js = ceil_div(K().loss_wgs, ceil_div(ni, parts > 1 ? parts : 1));
  const int js_max = ceil_div(nj, kLossWaves);
  if (js > js_max) js = js_max;
  if (js > 16) js = 16;
  if (js < 1) js = 1;
  const int groups = ceil_div(d, 64);
  ng = groups <= 1 ? 1 : groups <= 2 ? 2 : groups <= 4 ? 4 : 8;
  nz = ceil_div(groups, ng);
}

size_t loss_workspace_bytes(int32_t b, int32_t d, int parts) {
  int ni, js, nz, ng;
  loss_geometry(b, d, ni, js, nz, ng, parts);
  size_t de = sizeof(float) * (size_t)js * b * d;
  de = (de + 15) / 16 * 16;
  return de + (sizeof(double) * (size_t)ni * js + 15) / 16 * 16 + sizeof(float) * (size_t)b * d;
}

// The workspace is NOT monotone in b: the number of j slabs grows as the i tiles get fewer (B = 2048 -> 128 tiles x 2 slabs, 3 B d floats;
// B = 2032 -> 127 x 3, 4 B d), so a plan sized for max_batch alone is too small for the shorter last batch of an epoch.  Upper bound over
// every b in [1, b_max]: within one tile count the largest b needs the most, so one candidate per tile count.
size_t loss_workspace_bytes_max(int32_t b_max, int32_t d, int parts) {
  size_t worst = 0;
  for (int ni = 1; ni <= ceil_div(b_max, 16); ++ni) {
    const int b = std::min(16 * ni, (int)b_max);
    worst = std::max(worst, loss_workspace_bytes(b, d, 1));
    if (parts > 1) worst = std::max(worst, loss_workspace_bytes(b, d, parts));
  }
  return worst;
}

struct LossLaunch {
  int ni, js, nz, ng;
  float *de_part;
  double *loss_part;
  float *e_b;
};

static void loss_layout(int32_t d, int32_t b, void *ws, LossLaunch &L, int parts = 1) {
  loss_geometry(b, d, L.ni, L.js, L.nz, L.ng, parts);
  size_t de_bytes = sizeof(float) * (size_t)L.js * b * d;
  de_bytes = (de_bytes + 15) / 16 * 16;
  const size_t lp_bytes = (sizeof(double) * (size_t)L.ni * L.js + 15) / 16 * 16;
  L.de_part = (float *)ws;
  L.loss_part = (double *)((char *)ws + de_bytes);
  L.e_b = (float *)((char *)ws + de_bytes + lp_bytes);
}

float *loss_workspace_e_b(int32_t d, int32_t b, void *ws) {
  LossLaunch L;
  loss_layout(d, b, ws, L);
  return L.e_b;
}

// stage 1: E_B = e[idx] (zero where keep == 0) into the workspace; returns where it is
static int loss_gather(int32_t d, const float *e, const int32_t *idx, const float *keep, int32_t b, void *ws, hipStream_t st, LossLaunch &L) {
  loss_layout(d, b, ws, L);
  hipLaunchKernelGGL(gather_rows_kernel, dim3(ceil_div((int64_t)b * d / 4, 256)), dim3(256), 0, st, b, d / 4, e, idx, keep, L.e_b);
  GSS_LAUNCH_CHECK("gather_rows_kernel");
  return GSS_OK;
}

// stage 2: the fused S / G / dE sweep over the gathered rows `e_b`
static int loss_sweep(int32_t d, int32_t b, float beta, float alpha, const float *e_b, hipStream_t st, LossLaunch &L, int tile0 = 0,
                      int tile_stride = 1) {
  LossArgs g{d, b, L.js, e_b, beta, alpha, L.de_part, L.loss_part, tile0, tile_stride};
  const int ni_mine = tile0 < L.ni ? ceil_div(L.ni - tile0, tile_stride) : 0;
  if (ni_mine == 0) return GSS_OK;
  dim3 grid(ni_mine, L.js, L.nz), block(64 * kLossWaves);
  const size_t lds = (size_t)2 * L.ng * 4 * 64 * sizeof(float4);
  const bool exact = (d == 64 * L.ng) && L.nz == 1;
#define GSS_LOSS_CASE(NGV)                                                              \
  case NGV:                                                                             \
    if (exact)                                                                          \
      hipLaunchKernelGGL((loss_fused_kernel<NGV, true>), grid, block, lds_request(loss_fused_kernel<NGV, true>, lds), st, g);      \
    else                                                                                \
      hipLaunchKernelGGL((loss_fused_kernel<NGV, false>), grid, block, lds_request(loss_fused_kernel<NGV, false>, lds), st, g);     \
    break;
  switch (L.ng) {
    GSS_LOSS_CASE(1)
    GSS_LOSS_CASE(2)
    GSS_LOSS_CASE(4)
    default:
      GSS_LOSS_CASE(8)
  }
#undef GSS_LOSS_CASE
  GSS_LAUNCH_CHECK("loss_fused_kernel");
  return GSS_OK;
}

int loss_fwd_bwd(int32_t n, int32_t d, const float *e, const int32_t *idx, int32_t b, float beta, float alpha,
                 float *loss_out, float *de_b, void *ws, void *stream) {
  if (int rc = check_d(d)) return rc;
  GSS_REQUIRE(n > 0 && b > 0 && e && idx && loss_out && de_b && ws, "loss_fwd_bwd: null operand or empty batch");
  hipStream_t st = as_stream(stream);
  LossLaunch L;
  if (int rc = loss_gather(d, e, idx, nullptr, b, ws, st, L)) return rc;
  if (int rc = loss_sweep(d, b, beta, alpha, L.e_b, st, L)) return rc;
  const int nb = ceil_div((int64_t)b * d / 4, 256);
  hipLaunchKernelGGL(loss_finish_kernel, dim3(nb > 0 ? nb : 1), dim3(256), 0, st, b, d, L.js, L.ni * L.js, L.de_part, L.loss_part,
                     alpha, de_b, loss_out);
  GSS_LAUNCH_CHECK("loss_finish_kernel");
  return GSS_OK;
}

// loss + the backward of F.normalize / F.elu on the batch rows (what gss_plan_loss_backward needs next), 3 launches
int loss_fwd_bwd_fused(int32_t n, int32_t d, const float *e, const int32_t *idx, int32_t b, float beta, float alpha, float *loss_out,
                       const float *inv_den, const float *p, float c, float *dx_b, float *dp_b, int32_t *pos_set, void *ws,
                       void *stream) {
  float *e_b = nullptr;
  if (int rc = loss_gather_rows(d, e, idx, nullptr, b, ws, &e_b, stream)) return rc;
  return loss_fused_gathered(d, b, beta, alpha, loss_out, idx, idx, nullptr, inv_den, p, c, dx_b, dp_b, pos_set, ws, stream);
}

// the two halves of loss_fwd_bwd_fused, for a plan that gathers the rows itself
int loss_gather_rows(int32_t d, const float *e, const int32_t *rows, const float *keep, int32_t b, void *ws, float **e_b_out, void *stream) {
  if (int rc = check_d(d)) return rc;
  GSS_REQUIRE(b > 0 && (e || keep) && rows && ws && e_b_out, "loss_gather_rows: null operand or empty batch");  // e may be null on an empty shard
  LossLaunch L;
  if (int rc = loss_gather(d, e, rows, keep, b, ws, as_stream(stream), L)) return rc;
  *e_b_out = L.e_b;
  return GSS_OK;
}

// loss_gather_rows with the batch-id translation in the same launch (see gather_rows_mapped_kernel)
int loss_gather_rows_mapped(int32_t d, const float *e, const int32_t *idx, const int32_t *node_map, int32_t lo, int32_t nl, const int32_t *gid2op,
                            int32_t *pid, int32_t *rloc, float *keep, int32_t b, void *ws, float **e_b_out, void *stream) {
  if (int rc = check_d(d)) return rc;
  GSS_REQUIRE(b > 0 && (e || nl == 0) && idx && pid && rloc && ws && e_b_out && nl >= 0, "loss_gather_rows_mapped: null operand or empty batch");
  LossLaunch L;
  loss_layout(d, b, ws, L);
  BatchMap m{idx, node_map, gid2op, lo, nl, pid, rloc, keep};
  hipLaunchKernelGGL(gather_rows_mapped_kernel, dim3(ceil_div((int64_t)b * d / 4, 256)), dim3(256), 0, as_stream(stream), b, d / 4, e, m, L.e_b);
  GSS_LAUNCH_CHECK("gather_rows_mapped_kernel");
  *e_b_out = L.e_b;
  return GSS_OK;
}

// [E_B | P_B | inv_B] of the members this shard owns into out ([b (2 d + 1)] floats), see gather_batch_kernel
int loss_gather_batch(int32_t d, const float *e, const float *p, const float *inv_den, const int32_t *idx, const int32_t *node_map, int32_t lo,
                      int32_t nl, const int32_t *gid2op, int32_t *pid, int32_t *rloc, float *keep, const int32_t *rows, int32_t b, float *out,
                      void *stream) {
  if (int rc = check_d(d)) return rc;
  GSS_REQUIRE(b > 0 && ((e && p && inv_den) || nl == 0) && out && nl >= 0, "loss_gather_batch: null operand or empty batch");
  GSS_REQUIRE(idx ? (pid && rloc) : (rows != nullptr), "loss_gather_batch: neither a batch to translate nor prepared rows");
  BatchMap m{idx, node_map, gid2op, lo, nl, pid, rloc, idx ? keep : nullptr};
  hipLaunchKernelGGL(gather_batch_kernel, dim3(ceil_div((int64_t)b * d / 4, 256)), dim3(256), 0, as_stream(stream), b, d / 4, e, p, inv_den, m, rows,
                     keep, out);
  GSS_LAUNCH_CHECK("gather_batch_kernel");
  return GSS_OK;
}

int loss_fused_gathered(int32_t d, int32_t b, float beta, float alpha, float *loss_out, const int32_t *idx, const int32_t *pos_ids,
                        const float *keep, const float *inv_den, const float *p, float c, float *dx_b, float *dp_b, int32_t *pos_set,
                        void *ws, void *stream) {
  LossStep s{};
  s.d = d;
  s.b = b;
  s.beta = beta;
  s.alpha = alpha;
  s.loss_out = loss_out;
  s.rows = idx;
  s.pos_ids = pos_ids;
  s.keep = keep;
  s.inv_den = inv_den;
  s.p = p;
  s.c = c;
  s.dx_b = dx_b;
  s.dp_b = dp_b;
  s.pos_set = pos_set;
  bool dgrad_done = false;
  return loss_step(s, ws, stream, &dgrad_done);
}

// debug knob "loss_dgrad": finish + the batch rows' input gradient in one launch (loss_finish_dgrad_kernel): -1 (default) = on a shard
// only -- there it is what spares the second batch all-reduce; on one GPU the two stand-alone launches are 1.9 us faster (in-process
// A/B, tools/ab_inproc.py: 0.3043 vs 0.3063 ms per step) --, 0 = never, 1 = always
bool loss_dgrad_available(int32_t d, int32_t b, bool sharded) {
  int ni, js, nz, ng;
  loss_geometry(b, d, ni, js, nz, ng);
  const int knob = K().loss_dgrad;
  return (knob == 1 || (knob < 0 && sharded)) && d == 64 * ng && nz == 1 && ng <= 4;
}

// Row-slab form of the sweep for a shard (LossStep.slab_parts > 1), first half: this rank's i tiles (rank, rank + parts, ...), then
// de_x = its rows of sum_js de_part (zeros elsewhere) + its share of the loss behind them ([b d + 1] floats) -- what the ranks sum.
int loss_step_slab_sweep(const LossStep &s, void *ws, float *de_x, void *stream) {
  if (int rc = check_d(s.d)) return rc;
  GSS_REQUIRE(s.b > 0 && s.e_b && de_x && ws && s.slab_parts > 1 && s.slab_rank >= 0 && s.slab_rank < s.slab_parts, "loss_step_slab_sweep: bad argument");
  hipStream_t st = as_stream(stream);
  LossLaunch L;
  loss_layout(s.d, s.b, ws, L, s.slab_parts);
  if (int rc = loss_sweep(s.d, s.b, s.beta, s.alpha, s.e_b, st, L, s.slab_rank, s.slab_parts)) return rc;
  const int ni_mine = s.slab_rank < L.ni ? ceil_div(L.ni - s.slab_rank, s.slab_parts) : 0;
  const int nb = ceil_div((int64_t)s.b * s.d / 4, 256);
  hipLaunchKernelGGL(loss_slab_sum_kernel, dim3(nb > 0 ? nb : 1), dim3(256), 0, st, s.b, s.d, L.js, ni_mine * L.js, s.slab_rank, s.slab_parts,
                     L.de_part, L.loss_part, s.alpha, de_x);
  GSS_LAUNCH_CHECK("loss_slab_sum_kernel");
  return GSS_OK;
}

// The loss of one step on the plan's path, over rows gathered beforehand: the sweep, then the finish -- with the batch rows' input
// gradient in the same launch when s.w1t is given and the width allows.  *dgrad_done says whether gax_b / gam_b were written.
int loss_step(const LossStep &s, void *ws, void *stream, bool *dgrad_done) {
  if (int rc = check_d(s.d)) return rc;
  GSS_REQUIRE(s.b > 0 && s.loss_out && s.inv_den && s.p && s.dx_b && s.dp_b && ws && dgrad_done, "loss_step: null operand");
  GSS_REQUIRE(s.pos_ids || s.rows || !s.pos_set, "loss_step: a batch-position map needs its keys");
  GSS_REQUIRE(!s.w1t || (s.w2t && s.gax_b && s.gam_b), "loss_step: incomplete input-gradient operands");
  hipStream_t st = as_stream(stream);
  const int d = s.d, b = s.b;
  LossLaunch L;
  loss_layout(d, b, ws, L);
  const float *e_b = s.e_b ? s.e_b : L.e_b;
  if (s.de_x) {
    // second half of the row-slab form: the ranks' sums are in de_x ([b][d], the loss behind it was copied out by the caller) -- one
    // "slab" for the finish, no loss partials
    L.js = 1;
    L.de_part = const_cast<float *>(s.de_x);
  } else if (int rc = loss_sweep(d, b, s.beta, s.alpha, e_b, st, L)) {
    return rc;
  }
  const int32_t *pos_ids = s.pos_ids ? s.pos_ids : s.rows;
  const int nloss = s.de_x ? 0 : L.ni * L.js;      // loss partials for the finish to sum (none: the loss came with de_x)
  *dgrad_done = false;
  if (s.w1t) {   // (the caller asked loss_dgrad_available before it handed the weights over)
    FinishDgrad T{b, d, L.js, nloss, L.de_part, L.loss_part, s.alpha, e_b, s.rows, pos_ids, s.pos_set, s.keep, s.inv_den, s.p, s.c,
                  s.dx_b, s.dp_b, s.loss_out, s.w1t, s.w2t, s.gax_b, s.gam_b, s.dgrad_all ? 1 : 0};
    dim3 grid(L.ni), block(256);
    const size_t lds = sizeof(float) * 16 * (size_t)(d + 4);
    if (L.ng == 1)
      hipLaunchKernelGGL((loss_finish_dgrad_kernel<1>), grid, block, lds, st, T);
    else if (L.ng == 2)
      hipLaunchKernelGGL((loss_finish_dgrad_kernel<2>), grid, block, lds, st, T);
    else
      hipLaunchKernelGGL((loss_finish_dgrad_kernel<4>), grid, block, lds, st, T);
    GSS_LAUNCH_CHECK("loss_finish_dgrad_kernel");
    *dgrad_done = true;
    return GSS_OK;
  }
  const int d4 = d / 4;
  int lg = 2;
  while ((1 << lg) < d4 && lg < 6) ++lg;
  const int vpl = d4 <= 64 ? 1 : d4 <= 128 ? 2 : 4;
  dim3 grid(ceil_div(b, 4 * (64 >> lg))), block(256);
#define GSS_FIN(V)                                                                                                              \
  hipLaunchKernelGGL((loss_finish_bwd_kernel<V>), grid, block, 0, st, b, d4, lg, L.js, nloss, L.de_part, L.loss_part, s.alpha, \
                     e_b, (const int32_t *)nullptr, s.rows, pos_ids, s.keep, s.inv_den, s.p, s.c, s.dx_b, s.dp_b, s.pos_set, s.loss_out)
  if (vpl == 1)
    GSS_FIN(1);
  else if (vpl == 2)
    GSS_FIN(2);
  else
    GSS_FIN(4);
#undef GSS_FIN
  GSS_LAUNCH_CHECK("loss_finish_bwd_kernel");
  return GSS_OK;
}

}  // namespace gss

using namespace gss;
extern "C" {
size_t gss_loss_workspace_bytes(int32_t b, int32_t d) { return loss_workspace_bytes(b, d, 1); }
size_t gss_loss_workspace_bytes_max(int32_t b_max, int32_t d) { return b_max > 0 ? loss_workspace_bytes_max(b_max, d, 1) : 0; }
int gss_loss_fwd_bwd(int32_t n, int32_t d, const float *e, const int32_t *idx, int32_t b, float beta, float alpha,
                     float *loss_out, float *de_b, void *ws, void *stream) {
  return loss_fwd_bwd(n, d, e, idx, b, beta, alpha, loss_out, de_b, ws, stream);
}
}
