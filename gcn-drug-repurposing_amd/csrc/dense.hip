// dense.hip -- the dense half of the GSS layer on fp32 MFMA (K3/K4/K8 of SURVEY.md section 2b).
//
//   gemm_nt_lds_kernel / proj_ws_kernel   OUT[n][j] = sum_k IN[n][k] * W[j][k]      (both operands K-contiguous)
//       forward  : P = [AX|AM] . [W1|W2]^T + b1 + b2, O = ELU(P), x_next = P_prev + decay O
//                  (nn.Linear x2 + add + F.elu + residual, modules/model.py:165,170-173,201-203)
//       backward : [g_ax | g_am] = dP . [W1 ; W2]  using pre-transposed weights
//   wgrad_tn_kernel  dW[f][k] = sum_n dP[n][f] * Z[n][k]        (reduction over the node dimension)
//
// MFMA: v_mfma_f32_16x16x4_f32 (exact fp32, 256 FLOP/clk/CU).  Operands reach the MFMAs through LDS (LDS-DMA staged tiles; round 1's
// L1/L2-fed form moved six times the L2->L1 traffic and is gone).  A float4's four elements
// feed four consecutive k-steps: a k-chunk of 16 is consumed in the order k = kc + 4*(lane>>4) + e,
// a permutation of the reduction order that both operands share.
// The operands are swapped (A = weights, B = node rows) so that each lane ends up with 4 consecutive
// output features of one node row -> float4 epilogue loads/stores.
//
// Roofline: MFMA fp32 (157 TFLOP/s).  flops fwd = 2 N (2d) d; bwd-input = 2 N d (2d); wgrad = 2 N d (2d).
#include <algorithm>

#include "ops.h"

namespace gss {

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

enum GemmEpi { EPI_FWD = 0, EPI_SPLIT = 1, EPI_FWD_NORM = 2 };

struct GemmArgs {
  int n;              // node rows
  int K, ksplit;      // reduction length; k < ksplit reads in0 / w[.][0], else in1 / w[.][1]
  int J, jsplit;      // output features; j < jsplit -> out0 / w[0][.], else out1 / w[1][.]
  const float *in0, *in1;
  int ld_in0, ld_in1;
  const float *w[2][2];
  int ld_w;
  float *out0, *out1;
  int ld_out0, ld_out1;
  // EPI_FWD
  const float *b1, *b2, *p_prev;
  float *x_next;
  float decay;
  // EPI_SPLIT
  const int32_t *rows;  // scatter map for the output row (nullable)
  float *inv_den;       // EPI_FWD_NORM: 1 / max(||x_row||, eps) per node (x_next then receives the unit-norm rows)
  const int32_t *rows_out_pos;   // EPI_FWD_NORM without a row list, with rows_out: node row r's unit-norm row also goes to
                                 // rows_out[rows_out_pos[r]] where that is >= 0 (the batch-position map: E_B of a FULL step)
  float *rows_out;      // EPI_FWD_NORM over a row list (nullable): the unit-norm row of tile row t ALSO goes to rows_out[t] -- the lazy step's
                        // tile rows are the batch positions, so this IS E_B = emb[idx] (model.py:216-217) without a gather launch
  unsigned long long *stamps;  // diagnostic (gss_debug_set_stamp_buffer, NULL in production): per wave {start, loop begin, loop end, end} in
                               // 100 MHz wall-clock ticks + {linear workgroup id, HW_ID}; tools/gemm_stamps.py reads it
};

// Epilogue of the forward projection for one 16-node x 16 NT tile row: lane (r, q) holds OUT[nd][j0 + 16 u + 4 q + 0..3]
// in acc[u].  All loads (biases, previous layer's P) are issued before the first store: the stores may alias them as
// far as the compiler knows, so a load placed after a store waits for that store (vmcnt counts both) -- written the
// naive way this was 3 NT serialised memory round trips per tile and most of the kernel's time.
// What the epilogue reads besides the accumulators -- biases, the previous layer's P, the batch-position map --, fetched AHEAD of the K
// loop (round 4): none of it depends on the loop, and at its end it used to be one more dependent round trip (~1 us with the L2s cold)
// in front of the exp / store phase that every workgroup of the launch enters at the same time.  The K loop's counted vmcnt waits stay
// sound: these loads are older than every LDS-DMA load and loads return in order.
template <int NT>
struct FwdPre {
  float4 bb[NT], pp[NT];
  int ndc, out_row;
  bool live;
  // whole-line form (GemmArgs::lines): the lane serves TWO rows (ndc, ndc1), NT / 2 column blocks each -- bb[m] their biases,
  // pp[m] / pp[NT / 2 + m] the previous layer's P of the first / second row
  int ndc1, out_row1;
  bool live1;
};

// x^2 + y^2 + z^2 + w^2 in ONE stated order (no contraction left to the compiler): both forms of the epilogue add a row's squares
// through it, block by block, so that F.normalize gives the same bits whichever form wrote the row
__device__ __forceinline__ float sumsq4(const float4 &o) { return __fmaf_rn(o.w, o.w, __fmaf_rn(o.z, o.z, __fmaf_rn(o.y, o.y, __fmul_rn(o.x, o.x)))); }
__device__ __forceinline__ float swap_neighbour(float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true)); }

// lanes 2k and 2k + 1 exchange a float4 (DPP quad_perm [1, 0, 3, 2]: no LDS)
__device__ __forceinline__ float4 swap_neighbour(const float4 &v) {
  float4 o;
  o.x = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v.x), 0xB1, 0xF, 0xF, true));
  o.y = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v.y), 0xB1, 0xF, 0xF, true));
  o.z = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v.z), 0xB1, 0xF, 0xF, true));
  o.w = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v.w), 0xB1, 0xF, 0xF, true));
  return o;
}

// Whole-line form of the epilogue (round 4).  In the MFMA's layout lane (r, q) holds features 16 u + 4 q .. + 3 of node r, so a load or
// store instruction touches 16 rows x 64 B: half cache lines, and a row's line is completed by the NEXT instruction.  The same bytes in
// whole lines move 16 % faster at this size (tools/micro/store_pattern.hip: 9.4 -> 7.85 us for the epilogue's 46 MB alone).  So the
// lanes of an even / odd node pair swap one block of every block pair (u, u + 1): afterwards the even lane holds columns 32 m + 4 q of
// BOTH rows of the pair and the odd lane columns 32 m + 16 + 4 q of both -- 8 lanes cover one row's 128-B line, an instruction 8 rows x
// 128 B.  Biases, P_prev and outputs are addressed in that layout; the row norm adds the pair's halves with one more shuffle.
template <int NT, int EPI>
__device__ __forceinline__ void fwd_prefetch_lines(const GemmArgs &g, int nd, int j0, int q, FwdPre<NT> &f) {
  const int odd = (int)(threadIdx.x & 1);
  const int row0 = nd - odd, row1 = row0 + 1;
  f.live = row0 < g.n;
  f.live1 = row1 < g.n;
  f.ndc = min(row0, g.n - 1);
  f.ndc1 = min(row1, g.n - 1);
  if (g.rows) {
    f.ndc = g.rows[f.ndc];
    f.ndc1 = g.rows[f.ndc1];
    if (f.ndc < 0) {
      f.live = false;
      f.ndc = 0;
    }
    if (f.ndc1 < 0) {
      f.live1 = false;
      f.ndc1 = 0;
    }
  }
#pragma unroll
  for (int m = 0; m < NT / 2; ++m) {
    const int j = j0 + 32 * m + 16 * odd + 4 * q;
    f.bb[m] = add4(ld4(g.b1 + j), ld4(g.b2 + j));
    f.pp[m] = g.p_prev ? ld4(g.p_prev + (size_t)f.ndc * g.ld_out0 + j) : make_float4(0.f, 0.f, 0.f, 0.f);
    f.pp[NT / 2 + m] = g.p_prev ? ld4(g.p_prev + (size_t)f.ndc1 * g.ld_out0 + j) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  f.out_row = f.out_row1 = -1;
  if (EPI == EPI_FWD_NORM && g.rows_out) {
    f.out_row = g.rows_out_pos ? g.rows_out_pos[f.ndc] : row0;
    f.out_row1 = g.rows_out_pos ? g.rows_out_pos[f.ndc1] : row1;
  }
}

template <int NT, int EPI>
__device__ __forceinline__ void fwd_epilogue_lines(const GemmArgs &g, const f32x4 (&acc)[NT], const FwdPre<NT> &f, int j0, int q) {
  const bool odd = (threadIdx.x & 1) != 0;
  float4 pv0[NT / 2], pv1[NT / 2], o0[NT / 2], o1[NT / 2];
#pragma unroll
  for (int m = 0; m < NT / 2; ++m) {
    const float4 lo = make_float4(acc[2 * m][0], acc[2 * m][1], acc[2 * m][2], acc[2 * m][3]);
    const float4 hi = make_float4(acc[2 * m + 1][0], acc[2 * m + 1][1], acc[2 * m + 1][2], acc[2 * m + 1][3]);
    const float4 got = swap_neighbour(odd ? lo : hi);     // even lane: the odd row's lower block; odd lane: the even row's upper block
    pv0[m] = add4(odd ? got : lo, f.bb[m]);
    pv1[m] = add4(odd ? hi : got, f.bb[m]);
    o0[m] = make_float4(elu1(pv0[m].x), elu1(pv0[m].y), elu1(pv0[m].z), elu1(pv0[m].w));
    o1[m] = make_float4(elu1(pv1[m].x), elu1(pv1[m].y), elu1(pv1[m].z), elu1(pv1[m].w));
    if (g.p_prev) {
      o0[m] = add4(f.pp[m], scale4(g.decay, o0[m]));
      o1[m] = add4(f.pp[NT / 2 + m], scale4(g.decay, o1[m]));
    }
  }
  float den0 = 1.f, den1 = 1.f;
  if (EPI == EPI_FWD_NORM) {
    // F.normalize (modules/model.py:205).  The even lane collects the squares of the pair's first row, the odd lane those of the second
    // (its neighbour's come by one swap per block), added in block order u = 0 .. NT - 1 like the MFMA-layout form does; then the same
    // two shuffles over q
    float ss = 0.f;
#pragma unroll
    for (int m = 0; m < NT / 2; ++m) {
      const float t0 = sumsq4(o0[m]), t1 = sumsq4(o1[m]);
      const float got = swap_neighbour(odd ? t0 : t1);
      ss = __fadd_rn(ss, odd ? got : t0);      // block 2 m     of the lane's row
      ss = __fadd_rn(ss, odd ? t1 : got);      // block 2 m + 1
    }
    ss += __shfl_xor(ss, 16, 64);
    ss += __shfl_xor(ss, 32, 64);
    const float mine = fmaxf(sqrtf(ss), 1e-12f), other = swap_neighbour(mine);
    den0 = odd ? other : mine;
    den1 = odd ? mine : other;
  }
#pragma unroll
  for (int m = 0; m < NT / 2; ++m) {
    const int j = j0 + 32 * m + 16 * (odd ? 1 : 0) + 4 * q;
    if (f.live) {
      const size_t off = (size_t)f.ndc * g.ld_out0 + j;
      st4(g.out0 + off, pv0[m]);
      st4(g.x_next + off, EPI == EPI_FWD_NORM ? unit4(den0, o0[m]) : o0[m]);
      if (EPI == EPI_FWD_NORM && g.rows_out && f.out_row >= 0) st4(g.rows_out + (size_t)f.out_row * g.ld_out0 + j, unit4(den0, o0[m]));
    }
    if (f.live1) {
      const size_t off = (size_t)f.ndc1 * g.ld_out0 + j;
      st4(g.out0 + off, pv1[m]);
      st4(g.x_next + off, EPI == EPI_FWD_NORM ? unit4(den1, o1[m]) : o1[m]);
      if (EPI == EPI_FWD_NORM && g.rows_out && f.out_row1 >= 0) st4(g.rows_out + (size_t)f.out_row1 * g.ld_out0 + j, unit4(den1, o1[m]));
    }
  }
  if (EPI == EPI_FWD_NORM && q == 0) {
    if (!odd && f.live) g.inv_den[f.ndc] = 1.f / den0;
    if (odd && f.live1) g.inv_den[f.ndc1] = 1.f / den1;
  }
}
template <int NT, int EPI, bool LINES = false>
__device__ __forceinline__ void fwd_prefetch(const GemmArgs &g, int nd, int j0, int q, FwdPre<NT> &f) {
  if constexpr (LINES && NT % 2 == 0) {
    fwd_prefetch_lines<NT, EPI>(g, nd, j0, q, f);
    return;
  }
  f.live = nd < g.n;
  f.ndc = min(nd, g.n - 1);
  if (g.rows) {   // forward over a row list (gss_plan_step_lazy): tile row -> node row, inputs and outputs alike
    f.ndc = g.rows[f.ndc];
    if (f.ndc < 0) {   // a batch member another shard owns: computed from row 0's operands, never stored
      f.live = false;
      f.ndc = 0;
    }
  }
#pragma unroll
  for (int u = 0; u < NT; ++u) {
    const int j = j0 + 16 * u + 4 * q;
    f.bb[u] = add4(ld4(g.b1 + j), ld4(g.b2 + j));
    f.pp[u] = g.p_prev ? ld4(g.p_prev + (size_t)f.ndc * g.ld_out0 + j) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  // E_B (EPI_FWD_NORM, rows_out): with a row list the tile rows are the batch positions; without one the position comes from the map
  f.out_row = -1;
  if (EPI == EPI_FWD_NORM && g.rows_out) f.out_row = g.rows_out_pos ? g.rows_out_pos[f.ndc] : nd;
}

template <int NT, int EPI, bool LINES = false>
__device__ __forceinline__ void fwd_epilogue(const GemmArgs &g, const f32x4 (&acc)[NT], const FwdPre<NT> &f, int j0, int q) {
  if constexpr (LINES && NT % 2 == 0) {
    fwd_epilogue_lines<NT, EPI>(g, acc, f, j0, q);
    return;
  }
  const int ndc = f.ndc;
  float4 pv[NT], o[NT];
  float ss = 0.f;
#pragma unroll
  for (int u = 0; u < NT; ++u) {
    pv[u] = add4(make_float4(acc[u][0], acc[u][1], acc[u][2], acc[u][3]), f.bb[u]);
    o[u] = make_float4(elu1(pv[u].x), elu1(pv[u].y), elu1(pv[u].z), elu1(pv[u].w));
    if (g.p_prev) o[u] = add4(f.pp[u], scale4(g.decay, o[u]));
    ss = __fadd_rn(ss, sumsq4(o[u]));
  }
  float den = 1.f;
  if (EPI == EPI_FWD_NORM) {
    // last layer: F.normalize (modules/model.py:205).  The tile row spans all d features, so the 4 lanes q = 0..3 of a
    // node hold its whole row; the sum of squares is reduced with two xor-shuffles.
    ss += __shfl_xor(ss, 16, 64);
    ss += __shfl_xor(ss, 32, 64);
    den = fmaxf(sqrtf(ss), 1e-12f);
  }
  if (!f.live) return;
  const int out_row = f.out_row;
#pragma unroll
  for (int u = 0; u < NT; ++u) {
    const size_t off = (size_t)ndc * g.ld_out0 + j0 + 16 * u + 4 * q;   // ndc == nd for a live row without a row list
    st4(g.out0 + off, pv[u]);
    st4(g.x_next + off, EPI == EPI_FWD_NORM ? unit4(den, o[u]) : o[u]);
    if (EPI == EPI_FWD_NORM && g.rows_out && out_row >= 0) st4(g.rows_out + (size_t)out_row * g.ld_out0 + j0 + 16 * u + 4 * q, unit4(den, o[u]));
  }
  if (EPI == EPI_FWD_NORM && q == 0) g.inv_den[ndc] = 1.f / den;
}

// (the stand-alone form: everything fetched right in front of the arithmetic -- kernels that do not hoist it)
template <int NT, int EPI, bool LINES = false>
__device__ __forceinline__ void fwd_epilogue(const GemmArgs &g, const f32x4 (&acc)[NT], int nd, int j0, int q) {
  FwdPre<NT> f;
  fwd_prefetch<NT, EPI, LINES>(g, nd, j0, q, f);
  fwd_epilogue<NT, EPI, LINES>(g, acc, f, j0, q);
}

// ---- LDS-staged variant (default) -----------------------------------------------------------------------
// Workgroup tile = 128 nodes x BN features (BN = 16 NT), 4 waves of 32 nodes x BN.  Per 16-wide K chunk the
// operands arrive by LDS-DMA (global_load_lds_dwordx4): one wave instruction moves one MFMA fragment block
// -- 16 rows x 64 B = 1 KB -- and writes it lane-linear (lane l's 16 B at base + 16 l).  The consuming lane
// has the same (row = l & 15, k = 4 (l >> 4)) role as the staging lane, so a fragment is read back with one
// conflict-free ds_read_b128 at base + 16 l.  NBUF = 4 LDS buffers; the DMA runs 3 chunks ahead of the
// MFMAs (counted vmcnt); one barrier per chunk.  Compared with fetching fragments from L1/L2 per wave (the
// kernel above, kept for A/B) this cuts L2->L1 traffic from ~48 to ~8 B/clk/CU.
// LDS-DMA in inline asm: hipcc must not see these loads, otherwise it drains them (vmcnt(0)) before every
// ds_read / barrier and the multi-chunk pipeline collapses (cdna guide section 5 'Pipelining across barriers').
// M0 (the LDS destination base) is written in the same statement that uses it and restored afterwards.
// Measured dead end: letting a workgroup walk 2-4 node tiles in sequence (next tile's DMA in flight under the epilogue)
// to spread the epilogue stores over the launch: 36.8 / 52.8 / 67.8 us vs 30.6 us at d = 128 -- a tile costs ~18 us of
// wall time per workgroup whatever surrounds it (rocprofv3: per wave 16k cycles of MFMA, 19k of s_waitcnt, 2 waves per
// SIMD), so the co-resident workgroups of the one-tile-per-workgroup grid are what hides the stalls.
__device__ __forceinline__ void glds16(const float *gsrc, unsigned lds_byte_addr) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_byte_addr)
      : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}


// Workgroups go to the 8 XCDs round-robin by their linear id, and each XCD has its own L2.  Workgroups that read the SAME operand
// rows (the column tiles of one node tile in the projections, the output tiles of one node slice in the weight gradient) are
// therefore renumbered so that they land on one XCD, next to each other in dispatch order: `spread` indexes the groups (its
// neighbours go to different XCDs), `share` the members of a group.  Linear id L -> (spread, share); the last (n_spread mod 8)
// groups keep the plain order.  (The plain numbering measured 1-3 % slower in three live sweeps; the knob went in round 6.)
struct XcdIds {
  int spread, share;
};
__device__ __forceinline__ XcdIds xcd_ids(int L, int n_spread, int n_share, bool on) {
  const int full = (n_spread / 8) * 8;
  XcdIds r;
  if (!on || n_share == 1) {
    r.spread = L % n_spread;
    r.share = L / n_spread;
  } else if (L < full * n_share) {
    r.share = (L / 8) % n_share;
    r.spread = (L % 8) + 8 * (L / (8 * n_share));
  } else {
    const int rem = L - full * n_share, tail = n_spread - full;
    r.spread = full + rem % tail;
    r.share = rem / tail;
  }
  return r;
}

// WAVES = 1 (forward over a short row list, gss_plan_step_lazy's 2048 batch rows): one wave and 16 MT nodes per workgroup, so that the
// few rows spread over 4 x as many CUs; a row's MFMA chain is the same, its result has the same bits
// STAMP: the diagnostic instantiation of tools/gemm_stamps.py (per-wave wall-clock stamps); production launches use STAMP = false,
// whose code carries none of it
template <int NT, int MT, int EPI, int WAVES = 4, bool STAMP = false, bool LINES = false, int MINW = 1>
__global__ __launch_bounds__(64 * WAVES, MINW) void gemm_nt_lds_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BN = 16 * NT;
  constexpr int BM = 16 * WAVES * MT;      // nodes per workgroup (MT 16-node tiles per wave)
  constexpr int XT = BM * 16;              // floats per X chunk tile
  constexpr int BUF = XT + BN * 16;        // floats per buffer
  constexpr int WG = (NT + WAVES - 1) / WAVES;  // W fragment blocks staged per wave
  constexpr int G = MT + WG;               // DMA instructions per wave and chunk (uniform over waves)
  constexpr int NBUF = 4, PF = 3;          // chunks ci+1 .. ci+PF-1 stay in flight while chunk ci is consumed
  float *lds = reinterpret_cast<float *>(smem);
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 15, q = lane >> 4;
  // grid = (node tiles, column tiles): the column tiles of a node tile read the same input rows
  const int linear = (int)(blockIdx.x + gridDim.x * blockIdx.y);
  unsigned long long *stamp = (STAMP && g.stamps) ? g.stamps + ((size_t)linear * WAVES + (threadIdx.x >> 6)) * 6 : nullptr;
  if (STAMP && stamp && (threadIdx.x & 63) == 0) {
    stamp[0] = wall_clock64();
    stamp[4] = (unsigned long long)linear;
    stamp[5] = (unsigned long long)__builtin_amdgcn_s_getreg((4) | (0 << 6) | ((16 - 1) << 11));
  }
  const XcdIds id = xcd_ids(linear, (int)gridDim.x, (int)gridDim.y, true);
  const int node_base = id.spread * BM;
  const int j0 = id.share * BN;
  const int jh = j0 >= g.jsplit ? 1 : 0;
  const int jrow0 = j0 - (jh ? g.jsplit : 0);

  // staging sources of this lane: X fragment blocks w and w + 4, W fragment blocks (w + 4 i) mod NT
  // (when NT < 4 some waves restage a block another wave also stages: same bytes, same place, and it keeps
  // the per-wave DMA count G uniform so one counted vmcnt fits all waves)
  const float *xsrc[2][MT];
  const float *wsrc[2][WG];
  int wblk[WG];
#pragma unroll
  for (int i = 0; i < WG; ++i) wblk[i] = (w + WAVES * i) % NT;
#pragma unroll
  for (int kh = 0; kh < 2; ++kh) {
    const float *in = kh ? g.in1 : g.in0;
    const int ld = kh ? g.ld_in1 : g.ld_in0;
    const float *wp = g.w[jh][kh];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      int node = min(g.n - 1, node_base + 16 * (w + WAVES * i) + r);
      if (EPI != EPI_SPLIT && g.rows) node = max(g.rows[node], 0);   // a skipped entry (< 0) stages row 0; fwd_epilogue drops the result
      xsrc[kh][i] = in ? in + (size_t)node * ld + 4 * q : nullptr;
    }
#pragma unroll
    for (int i = 0; i < WG; ++i) wsrc[kh][i] = wp ? wp + (size_t)(jrow0 + 16 * wblk[i] + r) * g.ld_w + 4 * q : nullptr;
  }
  const int nchunk = g.K / 16;
  const int csplit = g.ksplit / 16;
  auto stage = [&](int ci) {
    const int kh = ci >= csplit ? 1 : 0;
    const int off = (ci - (kh ? csplit : 0)) * 16;
    const unsigned buf = lds_base + (unsigned)((ci % NBUF) * BUF) * 4u;
#pragma unroll
    for (int i = 0; i < MT; ++i) glds16(xsrc[kh][i] + off, buf + (unsigned)((w + WAVES * i) * 1024));
#pragma unroll
    for (int i = 0; i < WG; ++i) glds16(wsrc[kh][i] + off, buf + (unsigned)(XT * 4 + wblk[i] * 1024));
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int u = 0; u < NT; ++u) acc[t][u] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // the epilogue's own operands, requested before the first chunk (one 16-node tile per wave; with two the registers are better spent)
  constexpr bool HOIST = EPI != EPI_SPLIT && MT == 1 && WAVES <= 4 && NT <= 8;   // (a 256-feature tile row: 32 more float4 would not fit)
  FwdPre<NT> pre;
  if (HOIST) fwd_prefetch<NT, EPI, LINES>(g, node_base + 16 * (MT * w) + r, j0, q, pre);

  for (int c = 0; c < PF && c < nchunk; ++c) stage(c);
  if (STAMP && stamp && lane == 0) stamp[1] = wall_clock64();
  for (int ci = 0; ci < nchunk; ++ci) {
    // chunk ci has landed once at most min(PF-1, nchunk-1-ci) younger chunks of this wave are outstanding
    const int younger = min(PF - 1, nchunk - 1 - ci);
    if (younger >= 2)
      wait_vmcnt<2 * G>();
    else if (younger == 1)
      wait_vmcnt<G>();
    else
      wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();  // every wave's share of chunk ci is in LDS; buffer (ci-1)%NBUF is free again
    const float *cur = lds + (ci % NBUF) * BUF;
    float4 b[MT], a[NT];
#pragma unroll
    for (int t = 0; t < MT; ++t) b[t] = *reinterpret_cast<const float4 *>(cur + (MT * w + t) * 256 + lane * 4);
#pragma unroll
    for (int u = 0; u < NT; ++u) a[u] = *reinterpret_cast<const float4 *>(cur + XT + u * 256 + lane * 4);
    if (ci + PF < nchunk) stage(ci + PF);  // into buffer (ci-1)%NBUF, last read before the barrier above
#pragma unroll
    for (int e = 0; e < 4; ++e) {
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        const float bv = e == 0 ? b[t].x : e == 1 ? b[t].y : e == 2 ? b[t].z : b[t].w;
#pragma unroll
        for (int u = 0; u < NT; ++u) {
          const float av = e == 0 ? a[u].x : e == 1 ? a[u].y : e == 2 ? a[u].z : a[u].w;
          acc[t][u] = mfma16(av, bv, acc[t][u]);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);  // keep the MFMA cluster inside its iteration
  }

  if (STAMP && stamp && lane == 0) stamp[2] = wall_clock64();
  // epilogue: lane (r, q) holds OUT[node_base + 16 (MT w + t) + r][j0 + 16 u + 4 q + 0..3]
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    const int nd = node_base + 16 * (MT * w + t) + r;
    if (EPI == EPI_SPLIT && LINES && NT % 2 == 0) {
      // whole 128-B lines, as fwd_epilogue_lines: the lanes of an even / odd node pair swap one block of every block pair
      const bool odd = (lane & 1) != 0;
      const int row0 = nd - (odd ? 1 : 0), row1 = row0 + 1;
      float *dst = jh == 0 ? g.out0 : g.out1;
      const int ldo = jh == 0 ? g.ld_out0 : g.ld_out1;
      const int jb = j0 - (jh ? g.jsplit : 0);
#pragma unroll
      for (int m = 0; m < NT / 2; ++m) {
        const float4 lo = make_float4(acc[t][2 * m][0], acc[t][2 * m][1], acc[t][2 * m][2], acc[t][2 * m][3]);
        const float4 hi = make_float4(acc[t][2 * m + 1][0], acc[t][2 * m + 1][1], acc[t][2 * m + 1][2], acc[t][2 * m + 1][3]);
        const float4 got = swap_neighbour(odd ? lo : hi);
        const int j = jb + 32 * m + (odd ? 16 : 0) + 4 * q;
        if (row0 < g.n) st4(dst + (size_t)row0 * ldo + j, odd ? got : lo);
        if (row1 < g.n) st4(dst + (size_t)row1 * ldo + j, odd ? hi : got);
      }
    } else if (EPI == EPI_SPLIT) {
      if (nd >= g.n) continue;
      const int orow = g.rows ? g.rows[nd] : nd;
#pragma unroll
      for (int u = 0; u < NT; ++u) {
        const int j = j0 + 16 * u + 4 * q;
        const float4 v = make_float4(acc[t][u][0], acc[t][u][1], acc[t][u][2], acc[t][u][3]);
        if (jh == 0)
          st4(g.out0 + (size_t)orow * g.ld_out0 + j, v);
        else
          st4(g.out1 + (size_t)orow * g.ld_out1 + (j - g.jsplit), v);
      }
    } else if (HOIST) {
      fwd_epilogue<NT, EPI, LINES>(g, acc[t], pre, j0, q);
    } else {
      fwd_epilogue<NT, EPI, LINES>(g, acc[t], nd, j0, q);
    }
  }
  if (STAMP && stamp) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the stores of this wave have left
    if (lane == 0) stamp[3] = wall_clock64();
  }
}

// Forward projection over a SHORT ROW LIST (gss_plan_step_lazy: the top layer on the batch rows, 2048 of them): 16 listed rows per
// workgroup, whose 4 waves split the FEATURES (16 NTW each) instead of the nodes.  The one-wave form of gemm_nt_lds_kernel gives such a
// tile to a single wave -- 512 dependent-latency MFMAs, 16 us for 134 MFLOP at B = 2048, d = 128 --; here every wave issues a quarter
// of them on its own SIMD.  Every output element accumulates its K steps in the same order as in gemm_nt_lds_kernel (chunk by chunk,
// e = 0..3 inside), and the epilogue is the SAME code run by wave 0 on all 4 NTW accumulator blocks of a row (handed over through LDS
// in the register layout the whole-row kernel has): results are bit-identical to the full pass, which the lazy step's contract needs.
template <int NTW, int EPI>
__global__ __launch_bounds__(256) void gemm_rows_split_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NT = 4 * NTW;              // feature blocks of a whole row (J = 16 NT)
  constexpr int XT = 16 * 16;              // floats of the X chunk tile (16 nodes x 16 k)
  constexpr int BUF = XT + NT * 256;       // floats per ring buffer
  constexpr int G = 1 + NTW;               // DMA instructions per wave and chunk: the X block (every wave, same bytes) + its W blocks
  constexpr int NBUF = 4, PF = 3;
  float *lds = reinterpret_cast<float *>(smem);
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 15, q = lane >> 4;
  const int node_base = (int)blockIdx.x * 16;
  const float *xsrc[2];
  const float *wsrc[2][NTW];
#pragma unroll
  for (int kh = 0; kh < 2; ++kh) {
    const float *in = kh ? g.in1 : g.in0;
    const int ld = kh ? g.ld_in1 : g.ld_in0;
    int node = min(g.n - 1, node_base + r);
    node = max(g.rows[node], 0);            // a skipped entry (< 0) stages row 0; fwd_epilogue drops the result
    xsrc[kh] = in + (size_t)node * ld + 4 * q;
#pragma unroll
    for (int i = 0; i < NTW; ++i) wsrc[kh][i] = g.w[0][kh] + (size_t)(16 * (NTW * w + i) + r) * g.ld_w + 4 * q;
  }
  const int nchunk = g.K / 16;
  const int csplit = g.ksplit / 16;
  auto stage = [&](int ci) {
    const int kh = ci >= csplit ? 1 : 0;
    const int off = (ci - (kh ? csplit : 0)) * 16;
    const unsigned buf = lds_base + (unsigned)((ci % NBUF) * BUF) * 4u;
    glds16(xsrc[kh] + off, buf);
#pragma unroll
    for (int i = 0; i < NTW; ++i) glds16(wsrc[kh][i] + off, buf + (unsigned)(XT * 4 + (NTW * w + i) * 1024));
  };
  f32x4 acc[NTW];
#pragma unroll
  for (int u = 0; u < NTW; ++u) acc[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // wave 0 runs the whole row's epilogue: its operands are requested now (see FwdPre)
  FwdPre<NT> pre;
  if (w == 0) fwd_prefetch<NT, EPI>(g, node_base + r, 0, q, pre);
  for (int c = 0; c < PF && c < nchunk; ++c) stage(c);
  for (int ci = 0; ci < nchunk; ++ci) {
    const int younger = min(PF - 1, nchunk - 1 - ci);
    if (younger >= 2)
      wait_vmcnt<2 * G>();
    else if (younger == 1)
      wait_vmcnt<G>();
    else
      wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    const float *cur = lds + (ci % NBUF) * BUF;
    const float4 b = *reinterpret_cast<const float4 *>(cur + lane * 4);
    float4 a[NTW];
#pragma unroll
    for (int u = 0; u < NTW; ++u) a[u] = *reinterpret_cast<const float4 *>(cur + XT + (NTW * w + u) * 256 + lane * 4);
    if (ci + PF < nchunk) stage(ci + PF);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float bv = e == 0 ? b.x : e == 1 ? b.y : e == 2 ? b.z : b.w;
#pragma unroll
      for (int u = 0; u < NTW; ++u) {
        const float av = e == 0 ? a[u].x : e == 1 ? a[u].y : e == 2 ? a[u].z : a[u].w;
        acc[u] = mfma16(av, bv, acc[u]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  // hand the accumulator blocks to wave 0 in the whole-row kernel's layout: block u of the row, lane (r, q)
  __builtin_amdgcn_s_barrier();            // every wave is done with the ring: reuse its first bytes
  float4 *dump = reinterpret_cast<float4 *>(smem);
#pragma unroll
  for (int u = 0; u < NTW; ++u) dump[(NTW * w + u) * 64 + lane] = make_float4(acc[u][0], acc[u][1], acc[u][2], acc[u][3]);
  __syncthreads();
  if (w != 0) return;
  f32x4 row[NT];
#pragma unroll
  for (int u = 0; u < NT; ++u) {
    const float4 v = dump[u * 64 + lane];
    row[u] = (f32x4){v.x, v.y, v.z, v.w};
  }
  fwd_epilogue<NT, EPI>(g, row, pre, 0, q);
}

// ---- weight-stationary persistent projection, d = 128 (round 5) --------------------------------------------------------------
// The one-tile-per-workgroup kernel above runs a launch as one burst of input, one of MFMA and one of output (12.5 us of
// matrix-pipe time + ~15 us of memory phases, serialised).  Here the WEIGHTS stay put and the node rows stream past them:
//   * a workgroup is 4 waves; wave w owns output features [32 w, 32 w + 32) and keeps its K x 32 slice of [W1 | W2] in registers for
//     the whole launch (256 x 32 floats = 128 registers per lane), in the fragment order the staged kernel reads from LDS
//   * a persistent workgroup walks 16-row tiles b, b + G, b + 2 G, ...; a tile's [AX | AM] rows (16 KB) arrive by LDS-DMA in a ring
//     of three slots, two tiles ahead of the MFMAs that read them; every wave reads the same fragments (one ds_read_b128 per 16-wide
//     K chunk) and issues 128 MFMAs per tile
//   * the epilogue is fwd_epilogue_lines' for one block pair: the wave's 32 features are one 128-B line per row; biases are loop
//     invariants, the previous layer's P rows and the batch-position map are requested before the tile's MFMAs
//   * F.normalize needs a row's squares from all four waves: per (row, q) the waves leave their blocks' partial sums in LDS and every
//     wave adds the eight blocks in block order -- the order of the one-wave-per-row forms, so the embeddings keep their bits
//   * two such workgroups share a CU (<= 256 registers); the second generation starts half a tile late (`stagger`), so that one's
//     epilogue and barrier bubbles fall under the other's MFMAs -- with one tile per workgroup a delay only shifted work (round 3),
//     here it is paid once for ~4 tiles
// Every output accumulates chunk by chunk, e = 0..3 inside, as in gemm_nt_lds_kernel: bit-identical results (tests/test_gpu_ops.py).
// vmcnt bookkeeping: a wave's vector-memory operations complete in issue order (loads, stores and LDS-DMA alike), so "all but the 4
// youngest" at the end of a tile's MFMAs is "everything except the DMA pieces of the tile after next", whatever stores are in flight.
constexpr int kWsWorkgroups = 512;        // its persistent workgroups: two per CU (256 / 768 measured slower, profiles/r05_proj_ws_bench.txt)
constexpr int kWsStagger = 4;             // its second generation of workgroups (linear id >= 256) starts this many x 512 cycles late (0 / 2 / 6 / 8 measured slower)
struct WsStamp {
  unsigned long long t[24];   // [0] start, [1] weights + first tile ready, [2 + 2 i] MFMAs of tile i done, [3 + 2 i] its stores issued (i < 9),
};                            // [20] linear id, [21] HW_ID, [22] tiles, [23] XCC_ID

template <int EPI, bool STAMP>
__global__ __launch_bounds__(256, 1) void proj_ws_kernel(GemmArgs g, WsStamp *stamps) {
  constexpr int stagger = kWsStagger;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NCH = 16;             // 16-wide chunks of K = 2 d = 256
  constexpr int SLOT = NCH * 256;     // floats of one tile: 16 rows x 256 k as 16 fragment blocks of 1 KB
  constexpr int NSLOT = 3;
  float *lds = reinterpret_cast<float *>(smem);
  float *part = lds + NSLOT * SLOT;   // [2][16 rows][4 q][8 blocks] partial sums of squares (EPI_FWD_NORM)
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 15, q = lane >> 4;
  const bool odd = (lane & 1) != 0;
  const int ntiles = (g.n + 15) >> 4;
  const int G = (int)gridDim.x, b = (int)blockIdx.x;
  const int my_tiles = b < ntiles ? (ntiles - b + G - 1) / G : 0;
  WsStamp *stamp = (STAMP && stamps) ? stamps + ((size_t)b * 4 + w) : nullptr;
  if (STAMP && stamp && lane == 0) {
    stamp->t[0] = wall_clock64();
    stamp->t[20] = (unsigned long long)b;
    stamp->t[21] = (unsigned long long)__builtin_amdgcn_s_getreg((4) | (0 << 6) | ((32 - 1) << 11));
    stamp->t[22] = (unsigned long long)my_tiles;
    stamp->t[23] = (unsigned long long)__builtin_amdgcn_s_getreg((20) | (0 << 6) | ((4 - 1) << 11));
  }
  if (my_tiles == 0) return;

  // a tile's DMA: wave w stages fragment blocks w, w + 4 (AX: k < 128) and w + 8, w + 12 (AM)
  auto stage = [&](int i) {
    const int node = min(g.n - 1, 16 * (b + i * G) + r);
    const float *s0 = g.in0 + (size_t)node * g.ld_in0 + 4 * q + 16 * w;
    const float *s1 = g.in1 + (size_t)node * g.ld_in1 + 4 * q + 16 * w;
    const unsigned dst = lds_base + (unsigned)(((i % NSLOT) * SLOT + w * 256) * 4);
    glds16(s0, dst);
    glds16(s0 + 64, dst + 4 * 1024);
    glds16(s1, dst + 8 * 1024);
    glds16(s1 + 64, dst + 12 * 1024);
  };
  auto stage_piece = [&](int i, int k) {   // piece k of stage(i): fragment block w + 4 k
    const int node = min(g.n - 1, 16 * (b + i * G) + r);
    const float *s = (k < 2 ? g.in0 + (size_t)node * g.ld_in0 : g.in1 + (size_t)node * g.ld_in1) + 4 * q + 16 * w + 64 * (k & 1);
    glds16(s, lds_base + (unsigned)(((i % NSLOT) * SLOT + (w + 4 * k) * 256) * 4));
  };
  stage(0);
  if (my_tiles > 1) stage(1);

  // L2 warm-up.  A kernel starts with cold L2s, and a CU fills from beyond its L2 at ~11 B/clk (its outstanding-miss budget over the
  // Infinity Cache's latency): every CU pulling the same 128 KB of weights that way took 5 us, with 8 workgroups on the chip as with
  // 256.  So each wave first touches ONE KB of [W1 | W2] -- the 32 CUs x 4 waves of an XCD (workgroups b, b + 8, ... share one) a
  // different KB each -- and waits: one round trip later the whole matrix sits in the XCD's L2 and the real loads below are L2 hits.
  {
    const int slice = ((b >> 3) & 31) * 4 + w;   // 0 .. 127: KB of W1 (< 64) or W2
    const float *wp = g.w[0][slice >> 6] + (size_t)(slice & 63) * 256 + 4 * lane;
    f32x4 sink;
    asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(sink) : "v"(wp) : "memory");
  }
  if (STAMP && stamp && lane == 0) stamp->t[19] = wall_clock64();
  // the wave's slice of [W1 | W2]: block u (features 32 w + 16 u + r), chunk c (k = 16 c + 4 q .. + 3; c >= 8 is W2)
  float4 wr[2][NCH];
  // (fetching them in whole 128-B lines, the halves of a 16-lane row traded by DPP -- round 5's gemm_ws_mode bit 1 -- measured no faster and
  //  was removed in round 6)
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int c = 0; c < NCH; ++c)
      wr[u][c] = ld4(g.w[0][c >> 3] + (size_t)(32 * w + 16 * u + r) * g.ld_w + 16 * (c & 7) + 4 * q);
  const int j = 32 * w + (odd ? 16 : 0) + 4 * q;   // the lane's columns in the whole-line layout (both rows of its pair)
  const float4 bb = add4(ld4(g.b1 + j), ld4(g.b2 + j));
  // second-generation workgroups (those that double up on the CUs, by dispatch order: speed only) start late
  if (b >= 256)
    for (int k = 0; k < stagger; ++k) __builtin_amdgcn_s_sleep(8);
  // everything requested so far is in: both tiles and the weights.  The BUILTIN form, so that hipcc's own wait-count pass sees the
  // weight loads complete here -- with an asm wait it would put its wait for them in front of the loop's first MFMA, where it also
  // covers the previous tile's stores on every trip
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), expcnt / lgkmcnt untouched
  __builtin_amdgcn_s_barrier();
  if (STAMP && stamp && lane == 0) stamp->t[1] = wall_clock64();

  for (int i = 0; i < my_tiles; ++i) {
    const int T = b + i * G;
    // the epilogue's per-tile operands, requested ahead of the tile's MFMAs
    const int row0 = 16 * T + (r & ~1), row1 = row0 + 1;
    const bool live0 = row0 < g.n, live1 = row1 < g.n;
    const int nd0 = min(row0, g.n - 1), nd1 = min(row1, g.n - 1);
    float4 pp0 = make_float4(0.f, 0.f, 0.f, 0.f), pp1 = pp0;
    if (g.p_prev) {
      pp0 = ld4(g.p_prev + (size_t)nd0 * g.ld_out0 + j);
      pp1 = ld4(g.p_prev + (size_t)nd1 * g.ld_out0 + j);
    }
    int orow0 = -1, orow1 = -1;
    if (EPI == EPI_FWD_NORM && g.rows_out) {
      orow0 = g.rows_out_pos[nd0];
      orow1 = g.rows_out_pos[nd1];
    }
    const bool ahead = i + 2 < my_tiles;   // tile i + 2 goes into the slot tile i - 1 was read from (every wave is past the barrier behind its MFMAs)

    const float *cur = lds + (i % NSLOT) * SLOT + lane * 4;
    f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
    // the row fragments are read two chunks (16 MFMAs = 512 pipe cycles) ahead of the MFMAs that use them
    float4 bf[4];
    bf[0] = *reinterpret_cast<const float4 *>(cur);
    bf[1] = *reinterpret_cast<const float4 *>(cur + 256);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      if (c + 2 < NCH) bf[(c + 2) & 3] = *reinterpret_cast<const float4 *>(cur + (c + 2) * 256);
      // the next-but-one tile's four DMA pieces go out one at a time in the shadow of the MFMAs (issued together in front of them
      // they cost the wave ~0.3 us per tile)
      if (c < 8 && (c & 1) && ahead) stage_piece(i + 2, c >> 1);
      __builtin_amdgcn_sched_barrier(0);
      const float4 bc = bf[c & 3];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float bv = e == 0 ? bc.x : e == 1 ? bc.y : e == 2 ? bc.z : bc.w;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const float av = e == 0 ? wr[u][c].x : e == 1 ? wr[u][c].y : e == 2 ? wr[u][c].z : wr[u][c].w;
          acc[u] = mfma16(av, bv, acc[u]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // tile i + 1 (requested a whole tile ago), this tile's operands and the previous tile's stores: all but the pieces just requested
    if (ahead)
      wait_vmcnt<4>();
    else
      wait_vmcnt<0>();
    if (STAMP && stamp && lane == 0 && i < 9) stamp->t[2 + 2 * i] = wall_clock64();

    // epilogue (fwd_epilogue_lines for the one block pair of this wave)
    const float4 lo = make_float4(acc[0][0], acc[0][1], acc[0][2], acc[0][3]);
    const float4 hi = make_float4(acc[1][0], acc[1][1], acc[1][2], acc[1][3]);
    const float4 got = swap_neighbour(odd ? lo : hi);
    const float4 pv0 = add4(odd ? got : lo, bb);
    const float4 pv1 = add4(odd ? hi : got, bb);
    float4 o0 = make_float4(elu1(pv0.x), elu1(pv0.y), elu1(pv0.z), elu1(pv0.w));
    float4 o1 = make_float4(elu1(pv1.x), elu1(pv1.y), elu1(pv1.z), elu1(pv1.w));
    if (g.p_prev) {
      o0 = add4(pp0, scale4(g.decay, o0));
      o1 = add4(pp1, scale4(g.decay, o1));
    }
    float den0 = 1.f, den1 = 1.f;
    if (EPI == EPI_FWD_NORM) {
      // F.normalize (modules/model.py:205): this lane's two partial sums -- rows (r & ~1) and (r | 1), block 2 w + odd -- go to LDS;
      // behind the barrier the lane adds its own row's eight blocks in block order, then the two shuffles over q of the other forms
      float *pt = part + (i & 1) * 512;
      pt[((r & ~1) * 4 + q) * 8 + 2 * w + (odd ? 1 : 0)] = sumsq4(o0);
      pt[((r | 1) * 4 + q) * 8 + 2 * w + (odd ? 1 : 0)] = sumsq4(o1);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      const float4 s0 = *reinterpret_cast<const float4 *>(pt + (r * 4 + q) * 8);
      const float4 s1 = *reinterpret_cast<const float4 *>(pt + (r * 4 + q) * 8 + 4);
      float ss = 0.f;
      ss = __fadd_rn(ss, s0.x);
      ss = __fadd_rn(ss, s0.y);
      ss = __fadd_rn(ss, s0.z);
      ss = __fadd_rn(ss, s0.w);
      ss = __fadd_rn(ss, s1.x);
      ss = __fadd_rn(ss, s1.y);
      ss = __fadd_rn(ss, s1.z);
      ss = __fadd_rn(ss, s1.w);
      ss += __shfl_xor(ss, 16, 64);
      ss += __shfl_xor(ss, 32, 64);
      const float mine = fmaxf(sqrtf(ss), 1e-12f), other = swap_neighbour(mine);
      den0 = odd ? other : mine;
      den1 = odd ? mine : other;
    } else {
      __builtin_amdgcn_s_barrier();   // every wave has read this tile's slot and has the next tile's pieces in
    }
    if (live0) {
      const size_t off = (size_t)nd0 * g.ld_out0 + j;
      st4(g.out0 + off, pv0);
      st4(g.x_next + off, EPI == EPI_FWD_NORM ? unit4(den0, o0) : o0);
      if (EPI == EPI_FWD_NORM && orow0 >= 0) st4(g.rows_out + (size_t)orow0 * g.ld_out0 + j, unit4(den0, o0));
    }
    if (live1) {
      const size_t off = (size_t)nd1 * g.ld_out0 + j;
      st4(g.out0 + off, pv1);
      st4(g.x_next + off, EPI == EPI_FWD_NORM ? unit4(den1, o1) : o1);
      if (EPI == EPI_FWD_NORM && orow1 >= 0) st4(g.rows_out + (size_t)orow1 * g.ld_out0 + j, unit4(den1, o1));
    }
    if (EPI == EPI_FWD_NORM && w == 0 && q == 0) {
      if (!odd && live0) g.inv_den[nd0] = 1.f / den0;
      if (odd && live1) g.inv_den[nd1] = 1.f / den1;
    }
    if (STAMP && stamp && lane == 0 && i < 9) stamp->t[3 + 2 * i] = wall_clock64();
  }
}

// Measured dead end, for the record: a weights-resident variant for d <= 128 ([W1|W2] = 128 KB DMA'd into LDS once per
// workgroup, 8 independent waves per CU, each holding its 16-node [AX|AM] block in registers, fragment reads one chunk
// ahead, no barrier in the loop): 32.9 us vs 31.4 us at N = 29,960, 202 vs 210 us at N = 240k.  Per-wave timestamps
// (wall_clock64) at N = 29,960: 6.5 us until the weights are in LDS (every CU pulls the same 128 KB through the same L2
// channels at the same time, in 64-B pieces), 9-17 us of MFMA (two waves per SIMD), 5.5 us of epilogue, all waves in
// lockstep -- the launch is one burst of input, one of MFMA, one of output, whatever the tiling.  A bare
// ds_read_b128-fed MFMA loop reaches 126-131 TFLOP/s (tools/micro/mfma_lds.hip), so the loop is not the limit.
// Also measured on gemm_nt_lds_kernel at N = 29,960, d = 128 (31-33 us by box): a separate, 7-deep ring for the streamed
// node-row chunks (neutral), delaying the second workgroup of every CU by 3-14 us (neutral to worse), two column halves
// per node tile so that half the stores leave early (35.8 us).  A plain elementwise kernel moves the projection's 76 MB in
// 12.5 us (tools/micro/stream_small.py) and the MFMAs need 13.7 us; the launch takes their sum plus start-up because
// all workgroups reach the store phase together.

// debug knob "wgrad_wgs": workgroups of a full-size weight-gradient launch (one per CU)   [knob wgrad_wgs, common.h Knobs]

unsigned long long *g_gemm_stamps = nullptr;   // gss_debug_set_stamp_buffer: diagnostic only, process-wide, not a knob

constexpr int kWsMinRows = 256 * 128 + 1;   // gemm_ws = -1: the weight-stationary projection from this many rows on -- more 128-node tiles than CUs

template <int EPI>
static int launch_gemm(const GemmArgs &g_in, int d, hipStream_t st) {
  if (g_in.n <= 0) return GSS_OK;
  GemmArgs g = g_in;
  g.stamps = g_gemm_stamps;
  {
    int nt = (d % 128 == 0) ? 8 : (d % 64 == 0) ? 4 : (d % 32 == 0) ? 2 : 1;
    if (EPI == EPI_FWD_NORM && d == 256) nt = 16;   // the fused row norm needs a row's features in one tile
    // few node rows (the top layer's batch-row input gradient: 2048 rows): narrower feature tiles so that the grid covers the chip
    // (64-node x 128-feature tiles give 64 workgroups at B = 2048, d = 128; 32-feature tiles 256)
    if (EPI == EPI_SPLIT)
      while (nt > 2 && (int64_t)ceil_div(g.n, 64) * (g.J / (16 * nt)) < 256) nt >>= 1;
    // 64-node tiles give 2-3 co-resident workgroups per CU (epilogue traffic overlaps MFMA); at d >= 256 the
    // W-staging redundancy of small tiles costs more than that buys (measured, tools/gemm_bench.py), and so it does once
    // the grid is many waves of workgroups deep (d = 128: N = 1M 791 -> 753 us, N = 4M 3061 -> 2913 us with 128-node tiles)
    const int mt = K().gemm_variant == 3 ? 2 : ((d >= 256 || g.n >= 262144) ? 2 : 1);
    if (EPI != EPI_SPLIT && g.rows && (int64_t)ceil_div(g.n, 64) * (g.J / (16 * nt)) < 256) {
      // forward over a short row list: 16 listed rows per workgroup.  Whole rows of 64 / 128 / 256 features: 4 waves that split the
      // features (gemm_rows_split_kernel, same bits); other widths: one wave per 16 x 16 nt tile
      if (EPI != EPI_SPLIT && g.J == d && g.jsplit == d && (d == 64 || d == 128 || d == 256)) {
        dim3 gridr(ceil_div(g.n, 16));
        const size_t ldsr = 4 * (size_t)(16 * 16 + d * 16) * sizeof(float);
        if (d == 64)
          hipLaunchKernelGGL((gemm_rows_split_kernel<1, EPI == EPI_SPLIT ? EPI_FWD : EPI>), gridr, dim3(256), ldsr, st, g);
        else if (d == 128)
          hipLaunchKernelGGL((gemm_rows_split_kernel<2, EPI == EPI_SPLIT ? EPI_FWD : EPI>), gridr, dim3(256), ldsr, st, g);
        else
          hipLaunchKernelGGL((gemm_rows_split_kernel<4, EPI == EPI_SPLIT ? EPI_FWD : EPI>), gridr, dim3(256), ldsr, st, g);
        GSS_LAUNCH_CHECK("gemm_rows_split_kernel");
        return GSS_OK;
      }
      dim3 grid1(ceil_div(g.n, 16), g.J / (16 * nt));
      const size_t lds1 = 4 * (size_t)(16 * 16 + 16 * nt * 16) * sizeof(float);
      switch (nt) {
        case 8: hipLaunchKernelGGL((gemm_nt_lds_kernel<8, 1, EPI, 1>), grid1, dim3(64), lds1, st, g); break;
        case 4: hipLaunchKernelGGL((gemm_nt_lds_kernel<4, 1, EPI, 1>), grid1, dim3(64), lds1, st, g); break;
        case 2: hipLaunchKernelGGL((gemm_nt_lds_kernel<2, 1, EPI, 1>), grid1, dim3(64), lds1, st, g); break;
        default: hipLaunchKernelGGL((gemm_nt_lds_kernel<1, 1, EPI, 1>), grid1, dim3(64), lds1, st, g); break;
      }
      GSS_LAUNCH_CHECK("gemm_nt_lds_kernel (one wave)");
      return GSS_OK;
    }
    if constexpr (EPI != EPI_SPLIT) {
      // d = 128, all rows: the weight-stationary persistent kernel (round 5; proj_ws_kernel above).  Same bits as the staged tiles.  It
      // spends ~5 us per workgroup pulling its 128 KB of weights before the first MFMA and earns that back tile by tile: slower than the
      // staged tiles while those fit the chip in one round (<= 256 tiles of 128 nodes: 31.3 vs 28.5 us at N = 32,768), faster from the
      // first tile beyond (N = 36,000: 33.2 vs 46.7 us -- the staged tiles' second round), 8-19 % faster from 45k rows on, 0.63-0.64 of
      // the fp32-MFMA peak against 0.57-0.58 at 250k - 1M rows (profiles/r05_proj_ws_bench.txt, r05_proj_ws_crossover.txt).
      const bool ws = K().gemm_ws == 1 || (K().gemm_ws < 0 && g.n >= kWsMinRows);
      if (ws && K().gemm_variant == 2 && d == 128 && g.K == 256 && g.J == 128 && !g.rows && g.in1 && (!g.rows_out || g.rows_out_pos)) {
        const int ntiles = ceil_div(g.n, 16);
        const int wgs = std::min(ntiles, kWsWorkgroups);
        const size_t ldsw = (size_t)(3 * 16 * 256 + 2 * 512) * sizeof(float);
        WsStamp *sp = reinterpret_cast<WsStamp *>(g.stamps);   // diagnostic (gss_debug_set_stamp_buffer; tools/proj_ws_stamps.py): 24 x 8 bytes per wave
        if (sp)
          hipLaunchKernelGGL((proj_ws_kernel<EPI, true>), dim3(wgs), dim3(256), ldsw, st, g, sp);
        else
          hipLaunchKernelGGL((proj_ws_kernel<EPI, false>), dim3(wgs), dim3(256), ldsw, st, g, sp);
        GSS_LAUNCH_CHECK("proj_ws_kernel");
        return GSS_OK;
      }
    }
    // (Measured, round 6: the N-row input gradient at d = 256 as 128 x 256 tiles of eight waves -- the shape that pays for the fused-norm
    //  projection below -- is 1.0 us per step SLOWER than the 128 x 128 tiles of four waves: profiles/r06_ab_live_prefetch_and_norm8.txt.)
    dim3 grid(ceil_div(g.n, 64 * mt), g.J / (16 * nt));
    const size_t lds = 4 * (size_t)(64 * mt * 16 + 16 * nt * 16) * sizeof(float);
    if (g.stamps && EPI == EPI_FWD && nt == 8 && mt == 1) {   // diagnostic twin (tools/gemm_stamps.py): the d = 128 projection only
      hipLaunchKernelGGL((gemm_nt_lds_kernel<8, 1, EPI_FWD, 4, true>), grid, dim3(256), lds, st, g);
      GSS_LAUNCH_CHECK("gemm_nt_lds_kernel (stamped)");
      return GSS_OK;
    }
    if ((K().gemm_variant == 5 || (K().gemm_variant == 2 && d == 128)) && nt == 8 && !g.rows && EPI != EPI_SPLIT) {
      // 128-node tiles as EIGHT waves of 16 nodes (round 4): the weights are staged once per 128 nodes as with MT = 2, but a wave keeps
      // the 16-node tile's 104-110 registers (no hoisted operands, the whole-line epilogue), so two such workgroups share a CU: 16
      // waves per CU instead of 8.  Same MFMA order per output, same bits.  d = 128 (tools/gemm_w8_ab.py): 28.0 vs 30.2 us (64-node
      // tiles) at N = 29,960, 188 vs 197 us (128-node, 4 waves) at 250k, equal from 1M on; in the step -1.75 us (full), -1.1 us (the
      // trainer's).  gemm_variant 5 forces it for other widths, 3 forces the 4-wave 128-node tiles.
      dim3 grid8(ceil_div(g.n, 128), g.J / (16 * nt));
      const size_t lds8 = 4 * (size_t)(128 * 16 + 16 * nt * 16) * sizeof(float);
      hipLaunchKernelGGL((gemm_nt_lds_kernel<8, 1, EPI, 8, false, true, 4>), grid8, dim3(512), lds8, st, g);
      GSS_LAUNCH_CHECK("gemm_nt_lds_kernel (8 waves)");
      return GSS_OK;
    }
    if constexpr (EPI == EPI_FWD_NORM) {
      if (nt == 16 && !g.rows) {
        // d = 256, last layer (round 5): 128-node tiles of ALL 256 features, so that F.normalize (and E_B) come out of the epilogue and the
        // stand-alone row normalisation -- a 12 us launch that re-reads and re-writes the [N][256] result -- and the E_B gather go away.
        // Config 3: 21 -> 19 launches, 1.2396 -> 1.2312 ms per step.  Round 6: the tile runs as EIGHT waves of 16 nodes (206 registers, two
        // waves per SIMD) instead of four waves of 32 (256 + 128 accumulation registers, one wave per SIMD): -8.2 us per step on one live
        // plan (profiles/r06_ab_live_prefetch_and_norm8.txt), same MFMA order per output, same bits.  gemm_variant 3 forces the four-wave form.
        dim3 gridn(ceil_div(g.n, 128), 1);
        const size_t ldsn = 4 * (size_t)(128 * 16 + 256 * 16) * sizeof(float);
        if (K().gemm_variant != 3) {
          hipLaunchKernelGGL((gemm_nt_lds_kernel<16, 1, EPI_FWD_NORM, 8, false, true, 2>), gridn, dim3(512), lds_request(gemm_nt_lds_kernel<16, 1, EPI_FWD_NORM, 8, false, true, 2>, ldsn), st, g);
          GSS_LAUNCH_CHECK("gemm_nt_lds_kernel (256 features, fused row norm, 8 waves)");
          return GSS_OK;
        }
        hipLaunchKernelGGL((gemm_nt_lds_kernel<16, 2, EPI_FWD_NORM, 4, false, true>), gridn, dim3(256), lds_request(gemm_nt_lds_kernel<16, 2, EPI_FWD_NORM, 4, false, true>, ldsn), st, g);
        GSS_LAUNCH_CHECK("gemm_nt_lds_kernel (256 features, fused row norm)");
        return GSS_OK;
      }
    }
    // passes without a row list write whole 128-B lines (fwd_epilogue_lines; a row list keeps the MFMA layout: its rows are
    // scattered, and the lazy step's contract -- the bits of the full pass -- holds because both forms add a row's squares in one order)
    if (!g.rows && nt >= 2) {
      constexpr int E = EPI;
#define GSS_GEMM_LINES(NTV)                                                                             \
  case NTV:                                                                                             \
    if (mt == 2)                                                                                        \
      hipLaunchKernelGGL((gemm_nt_lds_kernel<NTV, 2, E, 4, false, true>), grid, dim3(256), lds, st, g); \
    else                                                                                                \
      hipLaunchKernelGGL((gemm_nt_lds_kernel<NTV, 1, E, 4, false, true>), grid, dim3(256), lds, st, g); \
    break;
      switch (nt) {
        GSS_GEMM_LINES(8)
        GSS_GEMM_LINES(4)
        default:
          GSS_GEMM_LINES(2)
      }
#undef GSS_GEMM_LINES
      GSS_LAUNCH_CHECK("gemm_nt_lds_kernel (whole lines)");
      return GSS_OK;
    }
#define GSS_GEMM_CASE(NTV)                                                                              \
  case NTV:                                                                                             \
    if (mt == 2)                                                                                        \
      hipLaunchKernelGGL((gemm_nt_lds_kernel<NTV, 2, EPI>), grid, dim3(256), lds_request(gemm_nt_lds_kernel<NTV, 2, EPI>, lds), st, g); \
    else                                                                                                \
      hipLaunchKernelGGL((gemm_nt_lds_kernel<NTV, 1, EPI>), grid, dim3(256), lds_request(gemm_nt_lds_kernel<NTV, 1, EPI>, lds), st, g); \
    break;
    switch (nt) {
      GSS_GEMM_CASE(8)
      GSS_GEMM_CASE(4)
      GSS_GEMM_CASE(2)
      default:
        GSS_GEMM_CASE(1)
    }
#undef GSS_GEMM_CASE
    GSS_LAUNCH_CHECK("gemm_nt_lds_kernel");
    return GSS_OK;
  }
}

int dense_fwd(int32_t n, int32_t d, const float *ax, const float *am, const float *w1, const float *b1, const float *w2,
              const float *b2, const float *p_prev, float decay, float *p, float *x_next, void *stream, const int32_t *row_list) {
  if (int rc = check_d(d)) return rc;
  GSS_REQUIRE(n >= 0 && ax && am && w1 && b1 && w2 && b2 && p && x_next, "dense_fwd: null operand");
  GemmArgs g{};
  g.n = n;
  g.K = 2 * d;
  g.ksplit = d;
  g.J = d;
  g.jsplit = d;
  g.in0 = ax;
  g.in1 = am;
  g.ld_in0 = g.ld_in1 = d;
  g.w[0][0] = w1;
  g.w[0][1] = w2;
  g.w[1][0] = g.w[1][1] = nullptr;
  g.ld_w = d;
  g.out0 = p;
  g.out1 = nullptr;
  g.ld_out0 = g.ld_out1 = d;
  g.b1 = b1;
  g.b2 = b2;
  g.p_prev = p_prev;
  g.x_next = x_next;
  g.decay = decay;
  g.rows = row_list;
  return launch_gemm<EPI_FWD>(g, d, as_stream(stream));
}

// last layer: P as usual, but the residual mix is row-normalised on the fly: e = normalize(p_prev + decay elu(p))
bool dense_fwd_norm_available(int32_t d) { return d == 256 || d == 128 || d == 64 || d == 32 || d == 16; }

int dense_fwd_norm(int32_t n, int32_t d, const float *ax, const float *am, const float *w1, const float *b1, const float *w2,
                   const float *b2, const float *p_prev, float decay, float *p, float *e, float *inv_den, void *stream, const int32_t *row_list,
                   float *rows_out, const int32_t *rows_out_pos) {
  if (int rc = check_d(d)) return rc;
  GSS_REQUIRE(n >= 0 && ax && am && w1 && b1 && w2 && b2 && p && e && inv_den, "dense_fwd_norm: null operand");
  GSS_REQUIRE(dense_fwd_norm_available(d), "dense_fwd_norm: needs d in {16, 32, 64, 128, 256}");
  GemmArgs g{};
  g.n = n;
  g.K = 2 * d;
  g.ksplit = d;
  g.J = d;
  g.jsplit = d;
  g.in0 = ax;
  g.in1 = am;
  g.ld_in0 = g.ld_in1 = d;
  g.w[0][0] = w1;
  g.w[0][1] = w2;
  g.ld_w = d;
  g.out0 = p;
  g.ld_out0 = g.ld_out1 = d;
  g.b1 = b1;
  g.b2 = b2;
  g.p_prev = p_prev;
  g.x_next = e;
  g.decay = decay;
  g.inv_den = inv_den;
  g.rows = row_list;
  GSS_REQUIRE(!rows_out || (row_list != nullptr) != (rows_out_pos != nullptr), "dense_fwd_norm: rows_out goes with a row list or with a position map");
  g.rows_out = rows_out;
  g.rows_out_pos = rows_out_pos;
  return launch_gemm<EPI_FWD_NORM>(g, d, as_stream(stream));
}

int dense_bwd_input(int32_t n, int32_t d, const float *dp, const float *w1t, const float *w2t, const int32_t *rows,
                    float *g_ax, float *g_am, void *stream) {
  if (int rc = check_d(d)) return rc;
  GSS_REQUIRE(n >= 0 && dp && w1t && w2t && g_ax && g_am, "dense_bwd_input: null operand");
  GemmArgs g{};
  g.n = n;
  g.K = d;
  g.ksplit = d;
  g.J = 2 * d;
  g.jsplit = d;
  g.in0 = dp;
  g.in1 = nullptr;
  g.ld_in0 = g.ld_in1 = d;
  g.w[0][0] = w1t;
  g.w[1][0] = w2t;
  g.w[0][1] = g.w[1][1] = nullptr;
  g.ld_w = d;
  g.out0 = g_ax;
  g.out1 = g_am;
  g.ld_out0 = g.ld_out1 = d;
  g.rows = rows;
  return launch_gemm<EPI_SPLIT>(g, d, as_stream(stream));
}

// ---- weight gradients -----------------------------------------------------------------------------
//
// dWcat[f][k] = sum_n dP[n][f] * Z[n][k],  Z = [AX | AM]  (d x 2d output, reduction over nodes).
// One workgroup = 8 waves = one 64x64 output tile over one slice of the node range; wave w takes rows
// 4*(8 it + w) + (lane>>4).  A lane's float4 of dP holds 4 consecutive features: element e is the A
// operand of the MFMA whose output rows are features 64 G + 4 fi + e (fi = lane & 15) -- a row
// permutation undone when storing.  Same for Z on the column side.  Partial tiles are tree-reduced
// through LDS in a fixed order, written per slice and summed in slice order by wgrad_reduce_kernel.
// Measured at N = 29,960, d = 128 (partial + reduce): 256 workgroups 29.7 us; 128 / 384 / 512 / 1024 workgroups
// 44 / 35 / 31.5 / 35.6 us; 128 x 64 tiles (every dP row read half as often, 196 VGPRs) 31.5 us -- neither more slices
// nor less operand traffic helps, the one-workgroup-per-CU geometry stays.  Round 2: explicit register double-buffering of the
// operand loads (trip i + 1 requested before the MFMAs of trip i): 33.8 vs 31.0 us inside the step -- slower, the second operand
// set costs more in registers than the exposed latency it covers (the SIMD's partner wave already covers it).

constexpr int kWgWaves = 8;

struct WgradArgs {
  int n, d;
  const float *dp, *ax, *am;
  const int32_t *rows;
  float *part_w;  // [nslices][d][2d]
  float *part_b;  // [nslices][d]
  int rows_per_slice;
};

// Two problems may share one launch (grid.y = ns0 + slices of the second): the top layer's batch-row gradient is 64
// latency-bound workgroups on its own (10 us) and rides along with a full-N launch for free.
// (An LDS-DMA ring for the operand rows, one trip ahead -- round 3's wgrad_variant 2 -- measured 2-3 % slower in three live sweeps and was
// removed in round 6: profiles/r03_wgrad_variant_ab.txt, r04_knob_sweep_live.txt, r05_knob_sweep_live_config3.txt.)
__global__ __launch_bounds__(64 * kWgWaves) void wgrad_tn_kernel(WgradArgs g0, WgradArgs g1, int ns0) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // grid = (output tiles, node slices): the output tiles of a slice read the same dP / Z rows
  const XcdIds id = xcd_ids((int)(blockIdx.x + gridDim.x * blockIdx.y), (int)gridDim.y, (int)gridDim.x, true);
  const int bx = id.share, by = id.spread;
  const bool second = by >= ns0;
  const WgradArgs &g = second ? g1 : g0;
  // the two waves of a SIMD run the same program in lockstep; a static priority for one half (MI355X guide, item 4)
  float4 *red = reinterpret_cast<float4 *>(smem);  // [4 slots][16 tiles][64 lanes] float4
  const int lane = threadIdx.x & 63;
  const int w = threadIdx.x >> 6;
  const int fi = lane & 15, q = lane >> 4;
  const int tiles_k = (2 * g.d) / 64;
  const int G = bx / tiles_k, H = bx % tiles_k;
  const int slice = second ? by - ns0 : by;
  const int r0 = slice * g.rows_per_slice;
  const int r1 = min(g.n, r0 + g.rows_per_slice);
  const int dh = g.d / 64;
  const float *z = (H < dh) ? g.ax : g.am;
  const int zc = (H < dh ? H : H - dh) * 64 + 4 * fi;
  const int ac = G * 64 + 4 * fi;

  f32x4 acc[4][4];
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int e2 = 0; e2 < 4; ++e2) acc[e][e2] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);


  // four row-steps per trip: all 8 operand loads are issued first (branch-free: rows beyond the slice read a
  // valid row and are multiplied by 0), then the 64 MFMAs run while the next trip's loads are in flight
  constexpr int U = 4;
  for (int base = r0 + 4 * w; base < r1; base += 4 * kWgWaves * U) {
    float4 a4[U], b4[U];
    float msk[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int row = base + u * 4 * kWgWaves + q;
      const bool ok = row < r1;
      const int rc = ok ? row : r1 - 1;
      msk[u] = ok ? 1.f : 0.f;
      a4[u] = ld4(g.dp + (size_t)rc * g.d + ac);
      const int zr = g.rows ? g.rows[rc] : rc;
      b4[u] = ld4(z + (size_t)zr * g.d + zc);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float4 am = scale4(msk[u], a4[u]);
      cs = add4(cs, am);
      const float av[4] = {am.x, am.y, am.z, am.w};
      const float bv[4] = {b4[u].x, b4[u].y, b4[u].z, b4[u].w};
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2) acc[e][e2] = mfma16(av[e], bv[e2], acc[e][e2]);
    }
  }

  // tree reduction over the 8 waves (fixed order): 4..7 -> 0..3, 2..3 -> 0..1, 1 -> 0
  for (int half = kWgWaves / 2; half >= 1; half >>= 1) {
    if (w >= half && w < 2 * half) {
      float4 *dst = red + (size_t)(w - half) * 16 * 64;
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2)
          dst[(e * 4 + e2) * 64 + lane] = make_float4(acc[e][e2][0], acc[e][e2][1], acc[e][e2][2], acc[e][e2][3]);
    }
    __syncthreads();
    if (w < half) {
      const float4 *src = red + (size_t)w * 16 * 64;
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2) {
          const float4 v = src[(e * 4 + e2) * 64 + lane];
          acc[e][e2][0] += v.x;
          acc[e][e2][1] += v.y;
          acc[e][e2][2] += v.z;
          acc[e][e2][3] += v.w;
        }
    }
    __syncthreads();
  }

  // column sums of dP (bias gradient) from the H == 0 workgroups
  if (H == 0) {
    cs.x += __shfl_xor(cs.x, 16, 64); cs.y += __shfl_xor(cs.y, 16, 64); cs.z += __shfl_xor(cs.z, 16, 64); cs.w += __shfl_xor(cs.w, 16, 64);
    cs.x += __shfl_xor(cs.x, 32, 64); cs.y += __shfl_xor(cs.y, 32, 64); cs.z += __shfl_xor(cs.z, 32, 64); cs.w += __shfl_xor(cs.w, 32, 64);
    if (q == 0) red[w * 16 + fi] = cs;
    __syncthreads();
    if (w == 0 && q == 0) {
      float4 s = red[fi];
      for (int ww = 1; ww < kWgWaves; ++ww) s = add4(s, red[ww * 16 + fi]);
      st4(g.part_b + (size_t)slice * g.d + ac, s);
    }
  }

  if (w == 0) {
    // tile (e, e2), lane (c = fi, q), reg: f = 64 G + 4 (4 q + reg) + e ; k = 64 H + 4 c + e2
    float *pw = g.part_w + (size_t)slice * g.d * (2 * g.d);
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int f = 64 * G + 4 * (4 * q + reg) + e;
        const int k = 64 * H + 4 * fi;
        st4(pw + (size_t)f * (2 * g.d) + k, make_float4(acc[e][0][reg], acc[e][1][reg], acc[e][2][reg], acc[e][3][reg]));
      }
  }
}

// generic VALU fallback for d not a multiple of 64 (test sizes): one thread per output element
__global__ __launch_bounds__(256) void wgrad_simple_kernel(WgradArgs g) {
  const int slice = blockIdx.y;
  const int r0 = slice * g.rows_per_slice, r1 = min(g.n, r0 + g.rows_per_slice);
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const int d = g.d;
  if (idx < d * 2 * d) {
    const int f = idx / (2 * d), k = idx % (2 * d);
    const float *z = k < d ? g.ax : g.am;
    const int kk = k < d ? k : k - d;
    float s = 0.f;
    for (int row = r0; row < r1; ++row) {
      const int zr = g.rows ? g.rows[row] : row;
      s = fmaf(g.dp[(size_t)row * d + f], z[(size_t)zr * d + kk], s);
    }
    g.part_w[(size_t)slice * d * 2 * d + idx] = s;
  } else if (idx < d * 2 * d + d) {
    const int f = idx - d * 2 * d;
    float s = 0.f;
    for (int row = r0; row < r1; ++row) s += g.dp[(size_t)row * d + f];
    g.part_b[(size_t)slice * d + f] = s;
  }
}

// Adam state for the fused reduce + optimizer step (gss_plan_step): parameter order w1, b1, w2, b2
struct FusedAdam {
  float *param[4], *m[4], *v[4];
  float *w1t, *w2t;               // transposed copies of the updated square weights (nullable)
  float lr_over_bc1, inv_sqrt_bc2, beta1, beta2, eps;
  int32_t *pos_clear;             // batch-position map to reset for the next step (nullable)
  const int32_t *idx;
  int b;
  int enabled;
};

__device__ __forceinline__ float adam_update(float p, float g, float &m, float &v, const FusedAdam &a) {
  m = m + (g - m) * (1.f - a.beta1);
  v = v * a.beta2 + (1.f - a.beta2) * g * g;
  return p - a.lr_over_bc1 * (m / (sqrtf(v) * a.inv_sqrt_bc2 + a.eps));
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(int d, int nslices, const float *__restrict__ part_w,
                                                            const float *__restrict__ part_b, float *__restrict__ gw1,
                                                            float *__restrict__ gw2, float *__restrict__ gb, float *__restrict__ gb2,
                                                            int accumulate, FusedAdam ad) {
  constexpr int deep = 2;   // Adam's state / the batch ids requested first, slabs fetched sixteen at a time (the shallower forms of rounds 3-4 went in round 6)
  // 64 consecutive outputs per workgroup; wave w sums slices w, w+4, ... (4 loads in flight), LDS adds the 4
  // wave sums in wave order -> fixed summation order
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int idx = blockIdx.x * 64 + lane;
  const int nw = d * 2 * d;
  const int total = nw + d;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  // (round 4) the launch is a chain of dependent round trips around 33 MB of partial slabs: Adam's state of this output is requested
  // first (it does not depend on the sums), and the slabs are fetched eight at a time instead of four -- added in the same order
  float am0 = 0.f, av0 = 0.f, ap0 = 0.f;
  int clear_id = -1;
  if (deep && ad.enabled && ad.pos_clear && blockIdx.x * 256 + threadIdx.x < ad.b) clear_id = ad.idx[blockIdx.x * 256 + threadIdx.x];
  const bool adam_w = deep && ad.enabled && w == 0 && idx < nw;
  if (adam_w) {
    const int f = idx / (2 * d), k = idx % (2 * d);
    const int which = k < d ? 0 : 2;
    const size_t e = (size_t)f * d + (k < d ? k : k - d);
    am0 = ad.m[which][e];
    av0 = ad.v[which][e];
    ap0 = ad.param[which][e];
  }
  if (idx < total) {
    const float *src = idx < nw ? part_w + idx : part_b + (idx - nw);
    const size_t stride = idx < nw ? (size_t)nw : (size_t)d;
    int sl = w;
    for (; deep >= 2 && sl + 60 < nslices; sl += 64) {   // sixteen at a time: one round trip for the 64 slabs of config 2
      float a[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) a[j] = src[(size_t)(sl + 4 * j) * stride];
#pragma unroll
      for (int j = 0; j < 16; j += 4) {
        s0 += a[j];
        s1 += a[j + 1];
        s2 += a[j + 2];
        s3 += a[j + 3];
      }
    }
    for (; deep && sl + 28 < nslices; sl += 32) {
      const float a0 = src[(size_t)sl * stride], a1 = src[(size_t)(sl + 4) * stride], a2 = src[(size_t)(sl + 8) * stride],
                  a3 = src[(size_t)(sl + 12) * stride], a4 = src[(size_t)(sl + 16) * stride], a5 = src[(size_t)(sl + 20) * stride],
                  a6 = src[(size_t)(sl + 24) * stride], a7 = src[(size_t)(sl + 28) * stride];
      s0 += a0;
      s1 += a1;
      s2 += a2;
      s3 += a3;
      s0 += a4;
      s1 += a5;
      s2 += a6;
      s3 += a7;
    }
    for (; sl + 12 < nslices; sl += 16) {
      s0 += src[(size_t)sl * stride];
      s1 += src[(size_t)(sl + 4) * stride];
      s2 += src[(size_t)(sl + 8) * stride];
      s3 += src[(size_t)(sl + 12) * stride];
    }
    for (; sl < nslices; sl += 4) s0 += src[(size_t)sl * stride];
  }
  red[w][lane] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (w == 0 && idx < total) {
    const float s = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    if (idx < nw) {
      const int f = idx / (2 * d), k = idx % (2 * d);
      const int which = k < d ? 0 : 2;
      const size_t e = (size_t)f * d + (k < d ? k : k - d);
      float *dst = (k < d ? gw1 : gw2) + e;
      const float gval = accumulate ? *dst + s : s;
      *dst = gval;
      if (ad.enabled) {  // torch.optim.Adam on this element, in the same launch (train.py:184)
        float m = deep ? am0 : ad.m[which][e], v = deep ? av0 : ad.v[which][e];
        const float pn = adam_update(deep ? ap0 : ad.param[which][e], gval, m, v, ad);
        ad.m[which][e] = m;
        ad.v[which][e] = v;
        ad.param[which][e] = pn;
        float *wt = which == 0 ? ad.w1t : ad.w2t;
        if (wt) wt[(size_t)(k < d ? k : k - d) * d + f] = pn;
      }
    } else {
      const int f = idx - nw;
      const float gval = accumulate ? gb[f] + s : s;
      gb[f] = gval;
      if (gb2) gb2[f] = gval;
      if (ad.enabled) {
#pragma unroll
        for (int which = 1; which <= 3; which += 2) {  // b1 and b2 share the gradient but have their own Adam state
          float m = ad.m[which][f], v = ad.v[which][f];
          ad.param[which][f] = adam_update(ad.param[which][f], gval, m, v, ad);
          ad.m[which][f] = m;
          ad.v[which][f] = v;
        }
      }
    }
  }
  // reset the batch-position map for the next step (it was last read by the backward SpMM before this kernel)
  if (clear_id >= 0) ad.pos_clear[clear_id] = -1;
  else if (!deep && ad.enabled && ad.pos_clear) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < ad.b) ad.pos_clear[ad.idx[i]] = -1;
  }
}

static void wgrad_geometry(int32_t n, int32_t d, int &nslices, int &rows_per_slice) {
  const int tiles = (d % 64 == 0) ? (d / 64) * (2 * d / 64) : ceil_div((int64_t)d * 2 * d + d, 256);
  int want = ceil_div(K().wgrad_wgs, tiles);
  const int max_slices = n > 0 ? ceil_div(n, 32) : 1;
  if (want > max_slices) want = max_slices;
  if (want < 1) want = 1;
  rows_per_slice = ceil_div(ceil_div(n > 0 ? n : 1, want), 32) * 32;
  nslices = ceil_div(n > 0 ? n : 1, rows_per_slice);
}

size_t wgrad_workspace_bytes(int32_t n, int32_t d) {
  int ns, rps;
  wgrad_geometry(n, d, ns, rps);
  return sizeof(float) * (size_t)ns * ((size_t)d * 2 * d + d);
}

// stage 1: partial slabs of one (layer's) weight gradient into slices [slice0, slice0 + returned count) of ws
int wgrad_partial(int32_t n, int32_t d, const float *dp, const float *ax, const float *am, const int32_t *rows, void *ws,
                  int total_slices, int slice0, int *nslices_out, void *stream) {
  if (int rc = check_d(d)) return rc;
  GSS_REQUIRE(n >= 0 && dp && ax && am && ws && nslices_out, "wgrad_partial: null operand");
  hipStream_t st = as_stream(stream);
  int ns, rps;
  wgrad_geometry(n, d, ns, rps);
  GSS_REQUIRE(slice0 >= 0 && slice0 + ns <= total_slices, "wgrad_partial: slices [%d, %d) exceed %d", slice0, slice0 + ns, total_slices);
  *nslices_out = ns;
  float *pw = (float *)ws + (size_t)slice0 * d * 2 * d;
  float *pb = (float *)ws + (size_t)total_slices * d * 2 * d + (size_t)slice0 * d;
  WgradArgs g{n, d, dp, ax, am, rows, pw, pb, rps};
  if (n == 0) {  // empty shard: its slices must still read as zero
    GSS_HIP(hipMemsetAsync(pw, 0, sizeof(float) * (size_t)ns * d * 2 * d, st));
    GSS_HIP(hipMemsetAsync(pb, 0, sizeof(float) * (size_t)ns * d, st));
    return GSS_OK;
  }
  if (d % 64 == 0) {
    const int tiles = (d / 64) * (2 * d / 64);
    hipLaunchKernelGGL(wgrad_tn_kernel, dim3(tiles, ns), dim3(64 * kWgWaves), lds_request(wgrad_tn_kernel, 4 * 16 * 64 * sizeof(float4)), st, g, g, ns);
    GSS_LAUNCH_CHECK("wgrad_tn_kernel");
  } else {
    hipLaunchKernelGGL(wgrad_simple_kernel, dim3(ceil_div((int64_t)d * 2 * d + d, 256), ns), dim3(256), 0, st, g);
    GSS_LAUNCH_CHECK("wgrad_simple_kernel");
  }
  return GSS_OK;
}

// the same for two problems in ONE launch: problem 0 into slices [slice0_0, ...), problem 1 into [slice0_1, ...).
// Falls back to two launches when the MFMA kernel does not apply.
int wgrad_partial_pair(int32_t d, int32_t n0, const float *dp0, const float *ax0, const float *am0, const int32_t *rows0, int slice0_0,
                       int32_t n1, const float *dp1, const float *ax1, const float *am1, const int32_t *rows1, int slice0_1, void *ws,
                       int total_slices, int *ns0_out, int *ns1_out, void *stream) {
  if (d % 64 != 0 || n0 <= 0 || n1 <= 0) {
    if (int rc = wgrad_partial(n0, d, dp0, ax0, am0, rows0, ws, total_slices, slice0_0, ns0_out, stream)) return rc;
    return wgrad_partial(n1, d, dp1, ax1, am1, rows1, ws, total_slices, slice0_1, ns1_out, stream);
  }
  GSS_REQUIRE(dp0 && ax0 && am0 && dp1 && ax1 && am1 && ws && ns0_out && ns1_out, "wgrad_partial_pair: null operand");
  int ns0, rps0, ns1, rps1;
  wgrad_geometry(n0, d, ns0, rps0);
  wgrad_geometry(n1, d, ns1, rps1);
  GSS_REQUIRE(slice0_0 >= 0 && slice0_0 + ns0 <= total_slices && slice0_1 >= 0 && slice0_1 + ns1 <= total_slices,
              "wgrad_partial_pair: slices exceed %d", total_slices);
  *ns0_out = ns0;
  *ns1_out = ns1;
  float *base_w = (float *)ws, *base_b = (float *)ws + (size_t)total_slices * d * 2 * d;
  WgradArgs g0{n0, d, dp0, ax0, am0, rows0, base_w + (size_t)slice0_0 * d * 2 * d, base_b + (size_t)slice0_0 * d, rps0};
  WgradArgs g1{n1, d, dp1, ax1, am1, rows1, base_w + (size_t)slice0_1 * d * 2 * d, base_b + (size_t)slice0_1 * d, rps1};
  const int tiles = (d / 64) * (2 * d / 64);
  hipLaunchKernelGGL(wgrad_tn_kernel, dim3(tiles, ns0 + ns1), dim3(64 * kWgWaves), lds_request(wgrad_tn_kernel, 4 * 16 * 64 * sizeof(float4)),
                     as_stream(stream), g0, g1, ns0);
  GSS_LAUNCH_CHECK("wgrad_tn_kernel");
  return GSS_OK;
}

// stage 2: fixed-order sum of slices [0, nslices) -> gw1, gw2, gb (and gb2)
static int wgrad_reduce_launch(int32_t d, void *ws, int total_slices, int nslices, float *gw1, float *gw2, float *gb, float *gb2,
                               int accumulate, const FusedAdam &ad, void *stream) {
  GSS_REQUIRE(ws && gw1 && gw2 && gb && nslices >= 0 && nslices <= total_slices, "wgrad_reduce: bad argument");
  const float *pw = (const float *)ws;
  const float *pb = (const float *)ws + (size_t)total_slices * d * 2 * d;
  int nblk = ceil_div((int64_t)d * 2 * d + d, 64);
  if (ad.enabled && ad.pos_clear) nblk = std::max(nblk, ceil_div(ad.b, 256));
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(nblk), dim3(256), 0, as_stream(stream), d, nslices, pw, pb, gw1, gw2, gb, gb2,
                     accumulate, ad);
  GSS_LAUNCH_CHECK("wgrad_reduce_kernel");
  return GSS_OK;
}

int wgrad_reduce(int32_t d, void *ws, int total_slices, int nslices, float *gw1, float *gw2, float *gb, float *gb2, int accumulate,
                 void *stream) {
  FusedAdam none{};
  none.enabled = 0;
  return wgrad_reduce_launch(d, ws, total_slices, nslices, gw1, gw2, gb, gb2, accumulate, none, stream);
}

// fixed-order reduce of the weight-gradient slabs AND the Adam step on all four tensors in one launch
int wgrad_reduce_adam(int32_t d, void *ws, int total_slices, int nslices, float *const grad[4], float *const param[4],
                      float *const m[4], float *const v[4], int32_t step, float lr, float beta1, float beta2, float eps, float *w1t,
                      float *w2t, int32_t *pos_clear, const int32_t *idx, int32_t b, void *stream) {
  GSS_REQUIRE(step >= 1, "adam: step is 1-based");
  FusedAdam ad{};
  for (int k = 0; k < 4; ++k) {
    GSS_REQUIRE(grad[k] && param[k] && m[k] && v[k], "wgrad_reduce_adam: null tensor %d", k);
    ad.param[k] = param[k];
    ad.m[k] = m[k];
    ad.v[k] = v[k];
  }
  ad.w1t = w1t;
  ad.w2t = w2t;
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  ad.lr_over_bc1 = (float)((double)lr / bc1);
  ad.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  ad.beta1 = beta1;
  ad.beta2 = beta2;
  ad.eps = eps;
  ad.pos_clear = pos_clear;
  ad.idx = idx;
  ad.b = b;
  ad.enabled = 1;
  return wgrad_reduce_launch(d, ws, total_slices, nslices, grad[0], grad[2], grad[1], grad[3], 0, ad, stream);
}

int wgrad_slices(int32_t n, int32_t d) {
  int ns, rps;
  wgrad_geometry(n, d, ns, rps);
  return ns;
}

// wgrad_geometry is not monotone in n (rows_per_slice is rounded up to 32, so a slightly smaller n can need more
// slices: d = 128, n = 1034 -> 17, n = 1008 -> 32).  Upper bound over every n in [1, n_max]: nslices <= want <=
// min(ceil(wgrad_wgs / tiles), ceil(n_max / 32)).  Plans size the batch-row region with this so that the short last
// batch of an epoch always fits.
int wgrad_slices_max(int32_t n_max, int32_t d) {
  const int tiles = (d % 64 == 0) ? (d / 64) * (2 * d / 64) : ceil_div((int64_t)d * 2 * d + d, 256);
  const int cap = n_max > 0 ? ceil_div(n_max, 32) : 1;
  const int want = ceil_div(K().wgrad_wgs, tiles);
  return want < cap ? (want < 1 ? 1 : want) : cap;
}

int dense_bwd_weight(int32_t n, int32_t d, const float *dp, const float *ax, const float *am, const int32_t *rows,
                     float *gw1, float *gw2, float *gb, float *gb2, int accumulate, void *ws, void *stream) {
  GSS_REQUIRE(gw1 && gw2 && gb, "dense_bwd_weight: null operand");
  const int total = wgrad_slices(n, d);
  int ns = 0;
  if (int rc = wgrad_partial(n, d, dp, ax, am, rows, ws, total, 0, &ns, stream)) return rc;
  return wgrad_reduce(d, ws, total, ns, gw1, gw2, gb, gb2, accumulate, stream);
}

}  // namespace gss

using namespace gss;

extern "C" {
int gss_dense_fwd(int32_t n, int32_t d, const float *ax, const float *am, const float *w1, const float *b1, const float *w2,
                  const float *b2, const float *p_prev, float decay, float *p, float *x_next, void *stream) {
  return dense_fwd(n, d, ax, am, w1, b1, w2, b2, p_prev, decay, p, x_next, stream);
}
int gss_dense_bwd_input(int32_t n, int32_t d, const float *dp, const float *w1t, const float *w2t, const int32_t *rows,
                        float *g_ax, float *g_am, void *stream) {
  return dense_bwd_input(n, d, dp, w1t, w2t, rows, g_ax, g_am, stream);
}
size_t gss_wgrad_workspace_bytes(int32_t n, int32_t d) { return wgrad_workspace_bytes(n, d); }
int gss_dense_bwd_weight(int32_t n, int32_t d, const float *dp, const float *ax, const float *am, const int32_t *rows,
                         float *gw1, float *gw2, float *gb, int accumulate, void *ws, void *stream) {
  return dense_bwd_weight(n, d, dp, ax, am, rows, gw1, gw2, gb, nullptr, accumulate, ws, stream);
}
}
