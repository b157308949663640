// spmm.hip -- CSR SpMM  Y = A X  for the GSS layer (K1/K2/K9 of SURVEY.md section 2b).
//
// Replaces torch.sparse.mm at modules/model.py:163,169 (and its autograd for layers >= 2).
//
// Mapping (wave64): one wave owns one output row.  A feature row of d floats is d/4 float4s; a row
// vector is covered by LPR = min(64, pow2ceil(d/4)) lanes, so a wave gathers EP = 64/LPR neighbour
// rows per load instruction (d=128: two 512-B rows per global_load_dwordx4).  The wave first loads 64
// (col,val) pairs with one coalesced load each, then broadcasts them lane-group-wise with
// ds_bpermute while the gathers of an unrolled group are in flight.  Edge-slot partial sums are
// combined with xor-shuffles at the end (fixed order -> bitwise reproducible).  Rows longer than
// kLongRow edges are skipped here and handled by spmm_long_kernel with a whole 1024-thread
// workgroup per row (LDS reduction in wave order).  The Hadamard / backward elementwise work that
// the reference does in separate torch ops is fused into the row epilogue.
//
// Roofline: HBM-bound.  Algorithmic bytes per launch = 8 nnz + 4 (N+1) + 4 N d (X once) + 4 N d (Y),
// plus 4 N d per extra epilogue operand/output.
#include <algorithm>
#include <vector>

#include "ops.h"
#include "segments.h"

namespace gss {

constexpr int kLongRow = 512;   // rows with more stored entries go to the long-row kernel
constexpr int kRowsPerBlock = 16;
constexpr int kLongThreads = 1024;

enum SpmmMode { SPMM_PLAIN = 0, SPMM_FWD1 = 1, SPMM_BWD1 = 2, SPMM_BWD2 = 3, SPMM_BWD1S = 4, SPMM_BWD2S = 5 };

struct SpmmEpi {
  const float *a0, *a1, *a2;
  float *o0, *o1;
  float c;
  const int32_t *pos;      // SPMM_BWD1S: column id -> row of the compact operands, -1 if not a batch row
  const int32_t *pos_row;  // SPMM_BWD1S: output row -> row of the compact g_ax, -1 if not a batch row
  const uint32_t *posbits; // SPMM_BWD1S, optional: bit c set <=> pos[c] >= 0.  For huge operands: the bitmap (N / 8 bytes) stays in
                           // L2 where the int32 map (4 N bytes) does not, and only the rare hits go on to read pos[].
                           // SPMM_BWD2S, optional: bit c clear => row c of the gathered operand (u) is all zeros and row c of a0 (t)
                           // as well: such neighbours are skipped, such output rows do not read t (nor p when their sum is zero too)
  uint32_t *nzbits_out;    // SPMM_BWD1S, optional: bit r is set for every output row r whose u may be non-zero (the row had a batch-row
                           // neighbour or is a batch row); the caller clears the map beforehand.  The next hop's posbits.
  int skip_zero_rows;      // SPMM_BWD1S with nzbits_out: rows whose bit stays clear are not written at all (their u and t are zero and
                           // every reader consults the bitmap first); only when no peer reads the rows either (single shard)
  const float *y_in;       // second pass of a two-pass product (a shard's hop overlapped with its halo exchange): [n_rows][d] partial sums
                           // over the entries of another CSR of the same rows; added to this pass's sums before the epilogue.
                           // May alias an output (every element is read, then written, by the same lane)
  const uint32_t *gather_bits;  // SPMM_PLAIN, optional: bit c clear => row c of the operand is all zeros, the neighbour is skipped (the
                                // first pass of a two-pass SPMM_BWD2S, whose second pass carries the same bitmap as posbits)
  const uint32_t *rowbits;      // SPMM_PLAIN, optional: only rows whose bit is set are computed (the first pass of a two-pass SPMM_FWD1
                                // under a row bitmap, whose second pass carries the same bitmap as posbits)
                                // SPMM_BWD1S, optional (round 6): bit r set <=> row r is a batch row or a neighbour of one (the set the lazy
                                // step's forward already marks).  A row whose bit is clear has no entry that can hit: its segments are not
                                // walked (at RMAT 10M the hop streamed 1.68 GB of index to find 45 k entries); it is written as zeros, or
                                // not at all under skip_zero_rows -- exactly what the walk would have produced
  BatchPrep prep;               // SPMM_FWD1, prep.idx != NULL: the launch has one workgroup more than segment blocks -- its first --, which
  int prep_block;               // prepares the batch (see BatchPrep) instead of multiplying
  int pos_row_limit;            // SPMM_BWD2S / SPMM_BWD2, > 0: t, the residual and pos_row are defined for output rows below it only (a shard's
                                // own rows; the rows behind them -- the boundary rows of the in-place transposed A_hat -- have none)
  const int32_t *row_alias;     // SPMM_PLAIN, optional (the chunk pass of GiantRows): the row filters (pos, rowbits) are looked up at
                                // row_alias[row] -- the row a chunk belongs to -- instead of row
};

struct CsrView {
  const int32_t *rowptr, *col;
  const float *val;
  int32_t n_rows;
};

// accumulate sum_e val[e] * X[col[e]] over edges [e0, e1) for this wave.  On return every lane with
// li < d4 (VPL chunks) holds the full sum for its float4 column(s).
template <int LPR_LOG2, int VPL>
__device__ __forceinline__ void row_accumulate(const CsrView &a, const float *__restrict__ x, int d4, int e0, int e1,
                                               int lane, float4 (&acc)[VPL]) {
  constexpr int LPR = 1 << LPR_LOG2;
  constexpr int EP = 64 / LPR;
  constexpr int UNROLL = (EP >= 16) ? 4 : (64 / EP >= 8 ? 8 : 64 / EP);  // EP*UNROLL divides 64
  const int g = lane >> LPR_LOG2;
  const int li = lane & (LPR - 1);
  const bool col_ok = (VPL > 1) || (li < d4);
#pragma unroll
  for (int v = 0; v < VPL; ++v) acc[v] = make_float4(0.f, 0.f, 0.f, 0.f);
  const size_t rowstride = (size_t)d4 * 4;
  for (int base = e0; base < e1; base += 64) {
    const int ce = base + lane;
    int c = 0;
    float w = 0.f;
    if (ce < e1) {
      c = a.col[ce];
      w = a.val[ce];
    }
    const int cnt = min(64, e1 - base);
    for (int t = 0; t < cnt; t += EP * UNROLL) {
      float4 xv[UNROLL][VPL];
      float wv[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        const int src = t + u * EP + g;
        const int cc = __shfl(c, src, 64);
        wv[u] = __shfl(w, src, 64);
        const bool ok = (src < cnt) && col_ok;
        if (!ok) wv[u] = 0.f;
        const float *xr = x + (size_t)cc * rowstride;
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
          const int f4 = li + v * 64;
          xv[u][v] = (ok && (VPL == 1 || f4 < d4)) ? ld4(xr + (size_t)f4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
#pragma unroll
      for (int u = 0; u < UNROLL; ++u)
#pragma unroll
        for (int v = 0; v < VPL; ++v) acc[v] = fma4(wv[u], xv[u][v], acc[v]);
    }
  }
  // combine the EP edge slots
#pragma unroll
  for (int o = LPR; o < 64; o <<= 1) {
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      acc[v].x += __shfl_xor(acc[v].x, o, 64);
      acc[v].y += __shfl_xor(acc[v].y, o, 64);
      acc[v].z += __shfl_xor(acc[v].z, o, 64);
      acc[v].w += __shfl_xor(acc[v].w, o, 64);
    }
  }
}

__device__ __forceinline__ bool is_zero4(const float4 &v) { return v.x == 0.f && v.y == 0.f && v.z == 0.f && v.w == 0.f; }

template <int MODE>
__device__ __forceinline__ void row_epilogue(const SpmmEpi &ep, size_t off, float4 acc, long coff = -1, bool t_zero = false) {
  if (MODE == SPMM_PLAIN) {
    st4(ep.o0 + off, acc);
  } else if (MODE == SPMM_FWD1) {
    st4(ep.o0 + off, acc);
    st4(ep.o1 + off, mul4(acc, ld4(ep.a0 + off)));
  } else if (MODE == SPMM_BWD1) {
    // u = g_ax + dm (.) x_in ; t = dm (.) ax
    st4(ep.o0 + off, add4(ld4(ep.a0 + off), mul4(acc, ld4(ep.a1 + off))));
    st4(ep.o1 + off, mul4(acc, ld4(ep.a2 + off)));
  } else if (MODE == SPMM_BWD1S) {
    // same with a row-sparse g_ax held compactly: coff addresses its row (or is < 0)
    // a row without a batch-row neighbour (nearly all of them when B << N) has acc == 0: u = g_ax (or 0), t = 0 without reading x / ax.
    // t_zero here: the WHOLE row is zero and not a batch row (row_mark_nonzero) -- with skip_zero_rows it is not written at all, its
    // clear bit in nzbits_out tells every reader.  The decision is per row, never per piece: a live row writes every piece, zeros
    // included, so that no piece of it keeps an earlier step's values
    if (t_zero && ep.skip_zero_rows) return;
    const bool z = is_zero4(acc);
    float4 u = z ? make_float4(0.f, 0.f, 0.f, 0.f) : mul4(acc, ld4(ep.a1 + off));
    if (coff >= 0) u = add4(u, ld4(ep.a0 + coff));
    st4(ep.o0 + off, u);
    st4(ep.o1 + off, z ? make_float4(0.f, 0.f, 0.f, 0.f) : mul4(acc, ld4(ep.a2 + off)));
  } else if (MODE == SPMM_BWD2S) {
    // as SPMM_BWD2 with the residual gradient held compactly on the batch rows (coff < 0: not a batch row)
    // t_zero: the caller knows t's row is zero (posbits); with a zero sum as well the row's gradient is the residual alone
    const float4 gx = t_zero ? acc : add4(ld4(ep.a0 + off), acc);
    float4 dp = (t_zero && is_zero4(acc)) ? make_float4(0.f, 0.f, 0.f, 0.f) : scale4(ep.c, mul4(gx, elu_grad4(ld4(ep.a1 + off))));
    if (coff >= 0) dp = add4(dp, ld4(ep.a2 + coff));
    st4(ep.o0 + off, dp);
    if (ep.o1) st4(ep.o1 + off, gx);
  } else {
    // gx = t + A u ; dp = c * gx (.) elu'(p) (+ res)      (t_zero: a row behind pos_row_limit -- neither t nor a residual there)
    const float4 gx = t_zero ? acc : add4(ld4(ep.a0 + off), acc);
    float4 dp = scale4(ep.c, mul4(gx, elu_grad4(ld4(ep.a1 + off))));
    if (ep.a2 && !t_zero) dp = add4(dp, ld4(ep.a2 + off));
    st4(ep.o0 + off, dp);
    if (ep.o1) st4(ep.o1 + off, gx);
  }
}

template <int MODE, int LPR_LOG2, int VPL>
__global__ __launch_bounds__(256) void spmm_rows_kernel(CsrView a, int d4, const float *__restrict__ x, SpmmEpi ep, int nblk) {
  const int lane = threadIdx.x & 63;
  const int wib = threadIdx.x >> 6;
  const int bid = xcd_remap(blockIdx.x, nblk);
  constexpr int LPR = 1 << LPR_LOG2;
  const int g = lane >> LPR_LOG2;
  const int li = lane & (LPR - 1);
  const int row_end = min(a.n_rows, (bid + 1) * kRowsPerBlock);
  for (int row = bid * kRowsPerBlock + wib; row < row_end; row += 4) {
    const int e0 = a.rowptr[row], e1 = a.rowptr[row + 1];
    if (e1 - e0 > kLongRow) continue;  // owned by spmm_long_kernel
    float4 acc[VPL];
    row_accumulate<LPR_LOG2, VPL>(a, x, d4, e0, e1, lane, acc);
    if (g == 0) {
#pragma unroll
      for (int v = 0; v < VPL; ++v) {
        const int f4 = li + v * 64;
        if (f4 < d4) row_epilogue<MODE>(ep, ((size_t)row * d4 + f4) * 4, acc[v]);
      }
    }
  }
}

// one workgroup (16 waves) per long row: waves take 64-aligned slices of the edge range, partial
// sums meet in LDS and are added in wave order.
template <int MODE, int LPR_LOG2, int VPL>
__global__ __launch_bounds__(kLongThreads) void spmm_long_kernel(CsrView a, const int32_t *__restrict__ long_rows, int d4,
                                                                  const float *__restrict__ x, SpmmEpi ep) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float4 *part = reinterpret_cast<float4 *>(smem);  // [16][d4]
  const int lane = threadIdx.x & 63;
  const int wib = threadIdx.x >> 6;
  constexpr int NW = kLongThreads / 64;
  constexpr int LPR = 1 << LPR_LOG2;
  const int g = lane >> LPR_LOG2;
  const int li = lane & (LPR - 1);
  const int row = long_rows[blockIdx.x];
  const int e0 = a.rowptr[row], e1 = a.rowptr[row + 1];
  const int chunks = (e1 - e0 + 63) / 64;
  const int per = (chunks + NW - 1) / NW;
  const int s0 = min(e1, e0 + wib * per * 64), s1 = min(e1, e0 + (wib + 1) * per * 64);
  float4 acc[VPL];
  row_accumulate<LPR_LOG2, VPL>(a, x, d4, s0, s1, lane, acc);
  if (g == 0) {
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      const int f4 = li + v * 64;
      if (f4 < d4) part[wib * d4 + f4] = acc[v];
    }
  }
  __syncthreads();
  for (int f4 = threadIdx.x; f4 < d4; f4 += kLongThreads) {
    float4 s = part[f4];
    for (int w = 1; w < NW; ++w) s = add4(s, part[w * d4 + f4]);
    row_epilogue<MODE>(ep, ((size_t)row * d4 + f4) * 4, s);
  }
}


// ---- variant 2 (default): nnz-balanced segments, whole-row gathers -------------------------------------
//
// Degrees are heavy-tailed (at config 2, 9 % of the rows hold 47 % of the entries and hubs reach thousands),
// so the row-per-wave mapping above ends on a few overloaded waves: rocprofv3 shows its waves alive for 46k
// cycles on average in a 190k-cycle kernel.  Measured dead end, for the record: cutting the feature dimension
// into 64-byte slices pinned to one XCD's L2 balances perfectly but triples the L2 request count (64-B
// requests) and ran slower (111 us vs 90 us at d = 128) -- gathers must stay whole 128-B lines.  256-byte slices
// pinned to XCDs are what pays (launch_balanced below).
//
// Here the host cuts every row into segments of <= kSegEdges entries (build_segments, once per feature width)
// and packs them into a descriptor list, one descriptor per lane group (kSegEdges = 32 by default); a lane group is the LPR lanes that
// cover one feature row (32 lanes x float4 = 512 B at d = 128).  A row with s segments occupies an aligned
// block of p = pow2ceil(s) consecutive groups of one 1024-thread workgroup (16 waves); rows too long for that
// take the whole workgroup with longer segments.  Each group loads its segment's (col,val) pairs with one
// coalesced access, broadcasts them with ds_bpermute and keeps 4 row gathers in flight.  Partial sums are
// combined with xor-shuffles inside a wave and through LDS across the waves of a row's block, always in the
// same order -> bitwise reproducible, no atomics, no second pass.  Workgroups run in the sorted order (hub rows
// first, leaf rows last); interleaving heavy and light workgroups in dispatch order measured 10-25 % slower.
// A persistent launch (512-1024 resident workgroups looping over segment blocks, the next block's descriptor and first
// (col, val) chunk requested under the current block's epilogue) measured 36.9 us vs 30.8 us at d = 128: the loop costs
// more in the per-block code than the hardware dispatcher costs between workgroups.
// debug knob "spmm_seg_edges" (applies to CSR handles created afterwards)   [knob seg_edges, common.h Knobs]
constexpr int kBalThreads = 1024;  // 512-thread workgroups measured 5-30 % slower (hub rows get half the groups)
constexpr int kBalWaves = kBalThreads / 64;
constexpr int kListWorkgroups = 1024;   // persistent workgroups of a list launch (two rounds of the 512 that fit the chip)

template <int MODE>
__device__ __forceinline__ long compact_off(const SpmmEpi &ep, int row, int d4, int f4) {
  if (MODE != SPMM_BWD1S && MODE != SPMM_BWD2S) return -1;
  if (MODE == SPMM_BWD2S && ep.pos_row_limit > 0 && row >= ep.pos_row_limit) return -1;
  const int pr = ep.pos_row[row];
  return pr >= 0 ? ((long)pr * d4 + f4) * 4 : -1;
}

// SPMM_BWD2S with a bitmap: a clear bit says row `row` of t is zero
template <int MODE>
__device__ __forceinline__ bool row_t_zero(const SpmmEpi &ep, int row) {
  if ((MODE == SPMM_BWD2S || MODE == SPMM_BWD2) && ep.pos_row_limit > 0 && row >= ep.pos_row_limit) return true;   // a boundary row: t (and the residual) exist on own rows only
  if (MODE != SPMM_BWD2S || !ep.posbits) return false;
  return ((ep.posbits[(unsigned)row >> 5] >> (row & 31)) & 1u) == 0u;
}
// SPMM_BWD1S with nzbits_out: u = acc (.) x + g_ax can only be non-zero where the sum is, or on a batch row.  gmask = the lanes of
// this lane's group (they hold the pieces of one row); `piece_live` = some piece this lane holds is non-zero or belongs to a batch row.
// Called by every lane of the group that reached the row's epilogue; its first lane sets the row's bit when any lane is live.
// Returns true when the whole row is dead (all pieces of all lanes zero, not a batch row).
template <int MODE>
__device__ __forceinline__ bool row_mark_nonzero(const SpmmEpi &ep, int row, bool piece_live, unsigned long long gmask) {
  if (MODE != SPMM_BWD1S || !ep.nzbits_out) return false;
  const unsigned long long m = __ballot(piece_live);
  const bool live = (m & gmask) != 0ull;
  if (live && (int)(threadIdx.x & 63) == __builtin_ctzll(gmask)) atomicOr(&ep.nzbits_out[(unsigned)row >> 5], 1u << (row & 31));
  return !live;
}

// one workgroup's share of a balanced product: the segments of block `sblk`, feature slice `sidx` (the body of spmm_balanced_kernel and of
// its list form below)
template <int MODE, int LPR_LOG2, int VPL, bool NARROW>
__device__ __forceinline__ void spmm_balanced_block(const CsrView &a, const int4 *__restrict__ segs, int d4, const float *__restrict__ x, const SpmmEpi &ep,
                                                    int rowstride_f, int sblk, int sidx, int3 hot, float4 *part) {
  constexpr int LPR = 1 << LPR_LOG2;
  constexpr int GPW = 64 / LPR;
  const int lane = threadIdx.x & 63;
  const int wib = threadIdx.x >> 6;
  const int g = lane >> LPR_LOG2;
  const int li = lane & (LPR - 1);
  const int4 sd = segs[((size_t)sblk * kBalWaves + wib) * GPW + g];  // {row, first edge, edge count, flags | log2 p}
  int row = sd.x;
  const int e0 = sd.y;
  int e1 = sd.y + sd.z;
  // SPMM_PLAIN with a row map (ep.pos): only rows with pos[row] >= 0 are computed -- the others' segments gather nothing and store
  // nothing (their output rows keep whatever they held).  A computed row is summed exactly as in the full launch (same segments,
  // same tree): gss_plan_step_lazy's top layer needs A_hat M on the batch rows only
  const int frow = (MODE == SPMM_PLAIN && ep.row_alias && row >= 0) ? ep.row_alias[row] : row;   // (the chunk pass: filters speak of the chunk's row)
  if (MODE == SPMM_PLAIN && ep.pos && row >= 0 && ep.pos[frow] < 0) {
    row = -1;
    e1 = e0;
  }
  // SPMM_FWD1 with a row bitmap (ep.posbits): the same for AX / M -- gss_plan_step_lazy on huge graphs needs them on the batch rows
  // and their neighbours only
  if (MODE == SPMM_FWD1 && ep.posbits && row >= 0 && !((ep.posbits[(unsigned)row >> 5] >> (row & 31)) & 1u)) {
    row = -1;
    e1 = e0;
  }
  if (MODE == SPMM_PLAIN && ep.rowbits && row >= 0 && !((ep.rowbits[(unsigned)frow >> 5] >> (frow & 31)) & 1u)) {
    row = -1;
    e1 = e0;
  }
  if (MODE == SPMM_BWD1S && ep.rowbits && row >= 0 && !((ep.rowbits[(unsigned)row >> 5] >> (row & 31)) & 1u)) {
    if (ep.skip_zero_rows && ep.nzbits_out) row = -1;   // (a single shard does not even write its dead rows: their clear bit in nzbits_out says so)
    e1 = e0;
  }
  const int plog = sd.w & 0xff;
  const bool multiwave = (sd.w & 0x100) != 0;  // workgroup-uniform: some row of this workgroup spans several waves
  const bool col_ok = (VPL > 1) || (li < d4);
  constexpr unsigned long long kGroupLow = LPR >= 64 ? ~0ull : ((1ull << (LPR & 63)) - 1ull);   // LPR ones
  const unsigned long long gmask = kGroupLow << ((g << LPR_LOG2) & 63);                          // the lanes of this lane's group
  // feature slicing: this workgroup walks one d4-wide slice of rows that are rowstride_f floats long.  Either
  // time-separated (grid.y, the slow dispatch dimension: slice s+1 starts when slice s drains) or pinned to XCDs
  // (pin_ns, above); launch_balanced picks
  const size_t rowstride = (size_t)rowstride_f;
  const int rs4 = rowstride_f >> 2;
  const int slice_f4 = sidx * d4;
  x += (size_t)slice_f4 * 4;
  float4 acc[VPL];
#pragma unroll
  for (int v = 0; v < VPL; ++v) acc[v] = make_float4(0.f, 0.f, 0.f, 0.f);
  // (Measured dead end, round 6: requesting the epilogue's streamed operands -- the Hadamard operand of FWD1, t and p of the second backward hop
  //  -- HERE, ahead of the row's gathers, so that they are not the last dependent round trip of the row's chain: +0.2 us per step at config 2,
  //  -1.1 us of 1,231 at config 3 on one live plan (profiles/r06_ab_live_prefetch_and_norm8.txt).  The kernel is bound by the rate at which the
  //  L2s serve gathers, not by the length of a row's chain.)
  // one row gather of this lane group: slice of row cc of the operand, zeros when !ok
  auto gather_row = [&](int cc, bool ok, float4 (&dst)[VPL]) __attribute__((always_inline)) {
    if (NARROW) {
      // operand < 4 GB, < 2^24 rows: 32-bit byte offsets from the (uniform) base, one full-rate 24-bit multiply-add per gather
      const unsigned off = __umul24((unsigned)cc, (unsigned)(rowstride_f * 4)) + (unsigned)li * 16u;
#pragma unroll
      for (int v = 0; v < VPL; ++v) {
        const int f4 = li + v * 64;
        dst[v] = (ok && (VPL == 1 || f4 < d4)) ? *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(x) + (off + v * 1024u))
                                             : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      return;
    }
    const float *xr = x + (size_t)cc * rowstride;
    // a CSR with a declared hot set (gss_csr_set_hot: nodes relabelled hub-first; operand rows [0, hot.x) and
    // [hot.y, hot.z) belong to the hubs): every other row is fetched with the non-temporal policy, so that the
    // once-read cold rows do not evict the hubs' rows from L2 / the Infinity Cache
    const bool cold = hot.x >= 0 && !(cc < hot.x || (cc >= hot.y && cc < hot.z));
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      const int f4 = li + v * 64;
      const float *src = xr + (size_t)f4 * 4;
      float4 got = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ok && (VPL == 1 || f4 < d4)) {
        if (cold) {
          const f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(src));
          got = make_float4(t[0], t[1], t[2], t[3]);
        } else {
          got = ld4(src);
        }
      }
      dst[v] = got;
    }
  };
  // (round 4) narrow lane groups (LPR <= 16: the XCD-pinned 256-B slices of config 2) took their (col, val) pairs 16 at a time: 64 B per
  // group and load -- half cache lines, 6 % of the launch's L2 requests in a kernel that sits at the L2s' request ceiling.  Here a lane
  // takes TWO consecutive pairs with one 8-byte load each (dword-aligned: a segment starts anywhere), 2 LPR entries per trip; entry
  // t of the trip lives in lane t >> 1, component t & 1.  The entries are walked in the same order: same sums, same bits.
  typedef int int2u __attribute__((ext_vector_type(2), aligned(4)));
  typedef float float2u __attribute__((ext_vector_type(2), aligned(4)));
  const bool sparse_walk = MODE == SPMM_BWD1S || (MODE == SPMM_BWD2S && ep.posbits) || (MODE == SPMM_PLAIN && ep.gather_bits);
  if (LPR <= 16 && !sparse_walk) {
    auto load2 = [&](int idx, int &c0, int &c1, float &w0, float &w1) __attribute__((always_inline)) {
      c0 = c1 = 0;
      w0 = w1 = 0.f;
      if (idx + 1 < e1) {
        const int2u cv = *reinterpret_cast<const int2u *>(a.col + idx);
        const float2u wv = *reinterpret_cast<const float2u *>(a.val + idx);
        c0 = cv[0];
        c1 = cv[1];
        w0 = wv[0];
        w1 = wv[1];
      } else if (idx < e1) {
        c0 = a.col[idx];
        w0 = a.val[idx];
      }
    };
    int cn0, cn1;
    float wn0, wn1;
    load2(e0 + 2 * li, cn0, cn1, wn0, wn1);
    for (int base = e0; __any(base < e1); base += 2 * LPR) {
      const int c0 = cn0, c1 = cn1;
      const float w0 = wn0, w1 = wn1;
      load2(base + 2 * LPR + 2 * li, cn0, cn1, wn0, wn1);
      const int cnt = min(2 * LPR, e1 - base);  // <= 0 once this group is done
      for (int t = 0; __any(t < cnt); t += 4) {
        float4 xv[4][VPL];
        float wv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int src = (g << LPR_LOG2) + (((t + u) >> 1) & (LPR - 1));
          const int cc = __shfl((u & 1) ? c1 : c0, src, 64);      // t is a multiple of 4: entry t + u sits in component u & 1
          wv[u] = __shfl((u & 1) ? w1 : w0, src, 64);
          const bool ok = (t + u < cnt) && col_ok;
          if (!ok) wv[u] = 0.f;
          gather_row(cc, ok, xv[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int v = 0; v < VPL; ++v) acc[v] = fma4(wv[u], xv[u][v], acc[v]);
      }
    }
  } else {
  // the (col, val) pairs of the next LPR entries are requested before the gathers of the current ones
  int c_next = 0;
  float w_next = 0.f;
  if (e0 + li < e1) {
    c_next = a.col[e0 + li];
    w_next = a.val[e0 + li];
  }
  for (int base = e0; __any(base < e1); base += LPR) {
    const int c = c_next;
    const float w = w_next;
    const int ne = base + LPR + li;
    c_next = 0;
    w_next = 0.f;
    if (ne < e1) {
      c_next = a.col[ne];
      w_next = a.val[ne];
    }
    const int ce = base + li;
    const int cnt = min(LPR, e1 - base);  // <= 0 once this group is done
    constexpr int kFly = 4;   // (8 in flight measured 3-9 % slower at RMAT 10M and no faster at config 2 / 3 in three sweeps; the knob went in round 6)
    if (MODE == SPMM_BWD1S || (MODE == SPMM_BWD2S && ep.posbits) || (MODE == SPMM_PLAIN && ep.gather_bits)) {
      // row-sparse operand.  BWD1S: only neighbours that are batch rows contribute (about B/N of the entries): look the neighbour up in
      // the node -> compact-row map (behind the bitmap when there is one).  BWD2S with a bitmap: only neighbours whose row of u may be
      // non-zero.  The hits of a group are walked in entry order, kFly gathers in flight.
      int cp = -1;
      if (ce < e1) {
        const uint32_t *filter = MODE == SPMM_PLAIN ? ep.gather_bits : ep.posbits;
        const bool member = !filter || ((filter[(unsigned)c >> 5] >> (c & 31)) & 1u);
        if (member) cp = MODE == SPMM_BWD1S ? ep.pos[c] : c;
      }
      const unsigned long long hits = __ballot(cp >= 0);
      unsigned long long gm = (hits >> ((g << LPR_LOG2) & 63)) & kGroupLow;
      while (__any(gm != 0ull)) {
        float4 xv[kFly][VPL];
        float wv[kFly];
#pragma unroll
        for (int u = 0; u < kFly; ++u) {
          const bool ok = gm != 0ull;
          const int srcl = ok ? __builtin_ctzll(gm) : 0;
          gm &= gm - 1ull;
          const int src = (g << LPR_LOG2) + srcl;
          const int cc = __shfl(cp, src, 64);
          const float ww = __shfl(w, src, 64);
          wv[u] = ok ? ww : 0.f;
          gather_row(ok ? cc : 0, ok && col_ok, xv[u]);
        }
#pragma unroll
        for (int u = 0; u < kFly; ++u)
#pragma unroll
          for (int v = 0; v < VPL; ++v) acc[v] = fma4(wv[u], xv[u][v], acc[v]);
      }
      continue;
    }
    // Measured dead end (round 2): broadcasting the entries of a 16-lane group with DPP (v_mov_b32_dpp row_newbcast:k, loop fully
    // unrolled) instead of ds_bpermute: 32.3 vs 32.9 us in a kernel that carried both paths, 31.4 us for this one alone -- the LDS
    // crossbar is not what the gather loop waits for.
    // kFly row gathers in flight per lane group.  Measured at config 2 (d = 128): 16 -> 44.1 us, 8 -> 39.5 us,
    // 4 -> 36.2 us, 2 -> 37.7 us.  Deeper queues only add L2 thrash; at 4 the kernel needs 48 VGPRs, so two
    // 1024-thread workgroups share a CU (32 waves) instead of one.
    for (int t = 0; __any(t < cnt); t += kFly) {
      float4 xv[kFly][VPL];
      float wv[kFly];
#pragma unroll
      for (int u = 0; u < kFly; ++u) {
        const int src = (g << LPR_LOG2) + ((t + u) & (LPR - 1));
        const int cc = __shfl(c, src, 64);
        wv[u] = __shfl(w, src, 64);
        const bool ok = (t + u < cnt) && col_ok;
        if (!ok) wv[u] = 0.f;
        gather_row(cc, ok, xv[u]);
      }
#pragma unroll
      for (int u = 0; u < kFly; ++u)
#pragma unroll
        for (int v = 0; v < VPL; ++v) acc[v] = fma4(wv[u], xv[u][v], acc[v]);
    }
  }
  }   // (one pair per lane and trip)
  // combine the groups of a row inside the wave (aligned power-of-two block of groups)
  const int pcount = 1 << plog;
#pragma unroll
  for (int o = 1; o < GPW; o <<= 1) {
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      const float tx = __shfl_xor(acc[v].x, o * LPR, 64), ty = __shfl_xor(acc[v].y, o * LPR, 64);
      const float tz = __shfl_xor(acc[v].z, o * LPR, 64), tw = __shfl_xor(acc[v].w, o * LPR, 64);
      if (pcount > o) {
        acc[v].x += tx;
        acc[v].y += ty;
        acc[v].z += tz;
        acc[v].w += tw;
      }
    }
  }
  if (pcount <= GPW) {
    if (row >= 0 && (g & (pcount - 1)) == 0) {
      bool tz = row_t_zero<MODE>(ep, row);
      if (ep.y_in) {   // the other pass's partial sums of this row (two-pass product)
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
          const int f4 = li + v * 64;
          if (f4 < d4) acc[v] = add4(ld4(ep.y_in + ((size_t)row * rs4 + slice_f4 + f4) * 4), acc[v]);
        }
      }
      if (MODE == SPMM_BWD1S) {
        bool piece_live = ep.pos_row[row] >= 0;
#pragma unroll
        for (int v = 0; v < VPL; ++v) piece_live |= (li + v * 64 < d4) && !is_zero4(acc[v]);
        tz = row_mark_nonzero<MODE>(ep, row, piece_live, gmask);
      }
#pragma unroll
      for (int v = 0; v < VPL; ++v) {
        const int f4 = li + v * 64;
        if (f4 < d4) {
          const long coff = compact_off<MODE>(ep, row, rs4, slice_f4 + f4);
          row_epilogue<MODE>(ep, ((size_t)row * rs4 + slice_f4 + f4) * 4, acc[v], coff, tz);
        }
      }
    }
  }
  if (!multiwave) return;
  // rows spanning nw = p / GPW waves: wave sums meet in LDS, the row's first wave adds them in wave order
  const int nw = pcount > GPW ? pcount / GPW : 1;
  if (nw > 1 && g == 0) {
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      const int f4 = li + v * 64;
      if (f4 < d4) part[wib * d4 + f4] = acc[v];
    }
  }
  __syncthreads();
  if (nw > 1 && (wib & (nw - 1)) == 0 && g == 0 && row >= 0) {
    float4 t[VPL];
    bool piece_live = MODE == SPMM_BWD1S && ep.pos_row[row] >= 0;
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      const int f4 = li + v * 64;
      t[v] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (f4 >= d4) continue;
      t[v] = part[wib * d4 + f4];
      for (int k = 1; k < nw; ++k) t[v] = add4(t[v], part[(wib + k) * d4 + f4]);
      if (ep.y_in) t[v] = add4(ld4(ep.y_in + ((size_t)row * rs4 + slice_f4 + f4) * 4), t[v]);
      piece_live |= !is_zero4(t[v]);
    }
    bool tz = row_t_zero<MODE>(ep, row);
    if (MODE == SPMM_BWD1S) tz = row_mark_nonzero<MODE>(ep, row, piece_live, gmask);
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      const int f4 = li + v * 64;
      if (f4 >= d4) continue;
      const long coff = compact_off<MODE>(ep, row, rs4, slice_f4 + f4);
      row_epilogue<MODE>(ep, ((size_t)row * rs4 + slice_f4 + f4) * 4, t[v], coff, tz);
    }
  }
}


template <int MODE, int LPR_LOG2, int VPL, bool NARROW>
__global__ __launch_bounds__(kBalThreads) void spmm_balanced_kernel(CsrView a, const int4 *__restrict__ segs, int d4,
                                                                   const float *__restrict__ x, SpmmEpi ep, int rowstride_f, int pin_ns, int3 hot) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float4 *part = reinterpret_cast<float4 *>(smem);  // [16 waves][d4]
  // pin_ns > 0: 1-D grid, slice = blockIdx.x % pin_ns.  Workgroups go to the XCDs round-robin by linear id, so XCD k
  // only ever gathers slice k % pin_ns and its L2 holds 1/pin_ns of the operand
  // (a launch that carries the batch preparation has ONE workgroup more, dispatched FIRST: as the last one it would start when the
  //  others are about to finish and lengthen the launch by its own chain of loads -- measured +2.9 us on the 31.9 us kernel)
  const bool has_prep = MODE == SPMM_FWD1 && ep.prep.idx != nullptr;
  const int bx = (int)blockIdx.x - (has_prep ? 1 : 0);
  const int sblk = pin_ns > 0 ? bx / pin_ns : bx;
  const int sidx = pin_ns > 0 ? bx % pin_ns : (int)blockIdx.y;
  if (has_prep && blockIdx.x == 0) {
    // the side job (workgroup-uniform branch): batch_prepare_kernel's body for the whole batch
    if (blockIdx.y == 0) {
      const BatchPrep &q = ep.prep;
      for (int i = threadIdx.x; i < q.b; i += kBalThreads) {
        const int id = q.node_map ? q.node_map[q.idx[i]] : q.idx[i];
        const int rel = id - q.lo;
        const bool mine = rel >= 0 && rel < q.nl;
        const int op = q.gid2op ? q.gid2op[id] : (mine ? rel : -1);
        if (q.rloc) q.rloc[i] = min(max(rel, 0), max(q.nl - 1, 0));
        if (q.pid) q.pid[i] = op;
        if (q.keep) q.keep[i] = mine ? 1.f : 0.f;
        if (q.rlist) q.rlist[i] = mine ? rel : -1;
        if (op >= 0) q.pos[op] = i;
      }
    }
    return;
  }
  spmm_balanced_block<MODE, LPR_LOG2, VPL, NARROW>(a, segs, d4, x, ep, rowstride_f, sblk, sidx, hot, part);
}

// ---- row-filtered products over MANY workgroups (round 6) ----------------------------------------------------------------------
// A lazy step's top-layer products compute a few thousand rows of millions (the batch rows; the batch rows and their neighbours), and so
// does the batch-sparse backward hop.  The filters above make the other rows' segments gather nothing -- but every workgroup is still
// dispatched, and at RMAT 10M that IS the kernel: 1.8 ms for 440 k workgroups of which 0.4 % hold a row that passes (measured: the
// batch-sparse hop after its row filter went in, profiles/r06_bench_rmat_10M_200M.json).  So: live_blocks_kernel lists the workgroups
// (segment blocks) with at least one row the filter lets through -- one pass over the descriptors, any order: every row is still summed
// by its own lane groups in its own order, the bits do not depend on the list's order -- and spmm_balanced_list_kernel, a fixed grid of
// persistent workgroups, walks that list.  Used from 2,048 workgroups on (knob spmm_list_blocks) where the caller expects few rows to pass (LiveHint).
static thread_local int64_t t_live_hint = 0;
}  // namespace gss
gss::LiveHint::LiveHint(int64_t rows) : prev(gss::t_live_hint) { gss::t_live_hint = rows; }
gss::LiveHint::~LiveHint() { gss::t_live_hint = prev; }
namespace gss {
struct LiveFilter {
  const int32_t *pos;        // live <=> pos[frow] >= 0            (NULL: no row map)
  const uint32_t *bits;      // live <=> bit frow set               (NULL: no bitmap)
  const int32_t *row_alias;  // frow = row_alias[row]               (NULL: frow = row)
};
__global__ __launch_bounds__(256) void live_blocks_kernel(const int4 *__restrict__ segs, int nblk, int groups_per_block, LiveFilter f,
                                                          int32_t *__restrict__ out) {   // out[0] = count (zeroed by the caller), out[1 ..] = blocks
  const int lane = threadIdx.x & 63;
  const int blk = blockIdx.x * 4 + (threadIdx.x >> 6);   // one wave per segment block
  if (blk >= nblk) return;
  bool live = false;
  for (int k = lane; k < groups_per_block; k += 64) {
    const int row = segs[(size_t)blk * groups_per_block + k].x;
    if (row < 0) continue;
    const int frow = f.row_alias ? f.row_alias[row] : row;
    bool ok = true;
    if (f.pos) ok = f.pos[frow] >= 0;
    if (ok && f.bits) ok = ((f.bits[(unsigned)frow >> 5] >> (frow & 31)) & 1u) != 0u;
    live |= ok;
  }
  if (__any(live) && lane == 0) out[1 + atomicAdd(&out[0], 1)] = blk;
}

template <int MODE, int LPR_LOG2, int VPL, bool NARROW>
__global__ __launch_bounds__(kBalThreads) void spmm_balanced_list_kernel(CsrView a, const int4 *__restrict__ segs, int d4, const float *__restrict__ x,
                                                                        SpmmEpi ep, int rowstride_f, int pin_ns, int3 hot,
                                                                        const int32_t *__restrict__ live) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float4 *part = reinterpret_cast<float4 *>(smem);
  const int items = live[0] * (pin_ns > 0 ? pin_ns : 1);   // pinned slices: item i = (block i / ns, slice i % ns); gridDim.x is a multiple of ns, so a
  for (int it = (int)blockIdx.x; it < items; it += (int)gridDim.x) {   // workgroup -- and with it its XCD -- keeps its slice
    const int sblk = live[1 + (pin_ns > 0 ? it / pin_ns : it)];
    const int sidx = pin_ns > 0 ? it % pin_ns : (int)blockIdx.y;
    spmm_balanced_block<MODE, LPR_LOG2, VPL, NARROW>(a, segs, d4, x, ep, rowstride_f, sblk, sidx, hot, part);
    __syncthreads();   // the next block reuses `part`
  }
}

template <int MODE, int LPR_LOG2, int VPL>
static int launch_spmm_t(const gss_csr *a, int d4, const float *x, const SpmmEpi &ep, hipStream_t st);

}  // namespace gss


namespace gss {

template <int MODE, int LPR_LOG2, int VPL>
static int launch_spmm_t(const gss_csr *a, int d4, const float *x, const SpmmEpi &ep, hipStream_t st) {
  CsrView v{a->rowptr, a->col, a->val, a->n_rows};
  const int nblk = ceil_div(a->n_rows, kRowsPerBlock);
  if (nblk > 0) {
    hipLaunchKernelGGL((spmm_rows_kernel<MODE, LPR_LOG2, VPL>), dim3(nblk), dim3(256), 0, st, v, d4, x, ep, nblk);
    GSS_LAUNCH_CHECK("spmm_rows_kernel");
  }
  if (a->n_long > 0) {
    const size_t lds = (size_t)(kLongThreads / 64) * d4 * sizeof(float4);
    hipLaunchKernelGGL((spmm_long_kernel<MODE, LPR_LOG2, VPL>), dim3(a->n_long), dim3(kLongThreads), lds, st, v,
                       a->d_long_rows, d4, x, ep);
    GSS_LAUNCH_CHECK("spmm_long_kernel");
  }
  return GSS_OK;
}


// Segment descriptors for the balanced kernel, built on first use for a given groups-per-wave count.
int csr_segments(const gss_csr *a, int gpw_log2, const int4 **out, int *n_blocks) {
  gss_csr *m = const_cast<gss_csr *>(a);  // lazily filled cache; a gss_csr is used from one host thread
  if (m->d_segs[gpw_log2]) {
    *out = (const int4 *)m->d_segs[gpw_log2];
    *n_blocks = m->n_seg_blocks[gpw_log2];
    return GSS_OK;
  }
  std::vector<int32_t> segs;
  int nblk = 0;
  if (m->by_items) {
    std::vector<SegItem> items(m->item_row.size());
    for (size_t i = 0; i < items.size(); ++i) items[i] = SegItem{m->item_row[i], m->item_first[i], m->item_len[i]};
    nblk = build_segments_items(items.data(), items.size(), kBalWaves, gpw_log2, K().seg_edges, segs);
  } else {
    nblk = build_segments(m->h_rowptr.data(), a->n_rows, kBalWaves, gpw_log2, K().seg_edges, segs);
  }
  const size_t bytes = segs.size() * sizeof(int32_t);
  if (bytes) {
    GSS_HIP(hipMalloc((void **)&m->d_segs[gpw_log2], bytes));
    GSS_HIP(hipMemcpy(m->d_segs[gpw_log2], segs.data(), bytes, hipMemcpyHostToDevice));
  }
  m->n_seg_blocks[gpw_log2] = nblk;
  *out = (const int4 *)m->d_segs[gpw_log2];
  *n_blocks = nblk;
  return GSS_OK;
}

// debug knob "spmm_hot_rows": overrides every CSR's hot set with rows [0, value) (-1 = use the CSR's own, 0 = none)   [knob spmm_hot, common.h Knobs]
// 0 = automatic (see launch_balanced)   [knob spmm_slices, common.h Knobs]
// with a manual "spmm_slices": slices pinned to XCDs (1) or time-separated (0)   [knob spmm_pin, common.h Knobs]

template <int MODE, int LPR_LOG2, int VPL>
static int launch_balanced_t(const gss_csr *a, int d4_slice, int nslices, bool pin, const float *x, const SpmmEpi &ep_in, hipStream_t st) {
  const int4 *segs = nullptr;
  int nblk = 0;
  if (int rc = csr_segments(a, 6 - LPR_LOG2, &segs, &nblk)) return rc;
  if (nblk == 0 && !(MODE == SPMM_FWD1 && ep_in.prep.idx)) return GSS_OK;
  CsrView v{a->rowptr, a->col, a->val, a->n_rows};
  const size_t lds = (size_t)kBalWaves * d4_slice * sizeof(float4);
  SpmmEpi ep = ep_in;
  const bool prep = MODE == SPMM_FWD1 && ep.prep.idx != nullptr;
  ep.prep_block = 0;
  const int extra = prep ? 1 : 0;             // the batch preparation rides as one more workgroup, the first of the launch
  int3 hot = make_int3(a->hot_own, a->hot_halo0, a->hot_halo1);
  if (K().spmm_hot >= 0) hot = K().spmm_hot > 0 ? make_int3(K().spmm_hot, 0, 0) : make_int3(-1, 0, 0);
  // the hot / cold split pays where the table is far beyond the caches; below that every row is "hot"
  if ((double)a->n_cols * d4_slice * nslices * 16.0 < 256.0 * 1024 * 1024 && K().spmm_hot < 0) hot = make_int3(-1, 0, 0);
  const bool narrow = (double)a->n_cols * d4_slice * nslices * 16.0 < 4.0e9 && a->n_cols < (1 << 24) && hot.x < 0;
  // a row filter over many workgroups: list the workgroups that hold a row it lets through and walk the list with persistent workgroups
  // (see live_blocks_kernel).  BWD1S: only when its dead rows are not written at all (else every row's zeros are this launch's to write).
  if constexpr (MODE == SPMM_PLAIN || MODE == SPMM_FWD1 || MODE == SPMM_BWD1S) {
    LiveFilter f{nullptr, nullptr, nullptr};
    if (MODE == SPMM_PLAIN) f = LiveFilter{ep.pos, ep.rowbits, ep.row_alias};
    if (MODE == SPMM_FWD1) f = LiveFilter{nullptr, ep.posbits, nullptr};
    if (MODE == SPMM_BWD1S && ep.skip_zero_rows && ep.nzbits_out) f = LiveFilter{nullptr, ep.rowbits, nullptr};
    const int list_min = K().spmm_list_blocks;
    const bool sparse_enough = list_min == 1 || t_live_hint <= 0 || t_live_hint * 4 < (int64_t)a->n_rows;   // (LiveHint: mostly-live launches keep the hardware's dispatch; knob value 1 = always: tests)
    if ((f.pos || f.bits) && !prep && list_min > 0 && nblk >= list_min && sparse_enough) {
      gss_csr *m = const_cast<gss_csr *>(a);   // lazily allocated scratch of the handle (one stream at a time, like the handle's other caches)
      const int k = 6 - LPR_LOG2;
      if (!m->d_live[k]) GSS_HIP(hipMalloc((void **)&m->d_live[k], sizeof(int32_t) * ((size_t)nblk + 1)));
      GSS_HIP(hipMemsetAsync(m->d_live[k], 0, sizeof(int32_t), st));
      hipLaunchKernelGGL(live_blocks_kernel, dim3(ceil_div(nblk, 4)), dim3(256), 0, st, segs, nblk, kBalWaves * (64 >> LPR_LOG2), f, m->d_live[k]);
      GSS_LAUNCH_CHECK("live_blocks_kernel");
      int gx = std::min(nblk, kListWorkgroups);
      if (pin) gx = std::max(nslices, gx / nslices * nslices);   // a multiple of the slice count: a workgroup keeps its slice
      const dim3 grid = pin ? dim3(gx) : dim3(gx, nslices);
      if (narrow)
        hipLaunchKernelGGL((spmm_balanced_list_kernel<MODE, LPR_LOG2, VPL, true>), grid, dim3(kBalThreads), lds, st, v, segs, d4_slice, x, ep,
                           d4_slice * nslices * 4, pin ? nslices : 0, make_int3(-1, 0, 0), (const int32_t *)m->d_live[k]);
      else
        hipLaunchKernelGGL((spmm_balanced_list_kernel<MODE, LPR_LOG2, VPL, false>), grid, dim3(kBalThreads), lds, st, v, segs, d4_slice, x, ep,
                           d4_slice * nslices * 4, pin ? nslices : 0, hot, (const int32_t *)m->d_live[k]);
      GSS_LAUNCH_CHECK("spmm_balanced_list_kernel");
      return GSS_OK;
    }
  }
  if (narrow)
    hipLaunchKernelGGL((spmm_balanced_kernel<MODE, LPR_LOG2, VPL, true>), pin ? dim3(nblk * nslices + extra) : dim3(nblk + extra, nslices), dim3(kBalThreads),
                       lds, st, v, segs, d4_slice, x, ep, d4_slice * nslices * 4, pin ? nslices : 0, make_int3(-1, 0, 0));
  else
    hipLaunchKernelGGL((spmm_balanced_kernel<MODE, LPR_LOG2, VPL, false>), pin ? dim3(nblk * nslices + extra) : dim3(nblk + extra, nslices), dim3(kBalThreads),
                       lds, st, v, segs, d4_slice, x, ep, d4_slice * nslices * 4, pin ? nslices : 0, hot);
  GSS_LAUNCH_CHECK("spmm_balanced_kernel");
  return GSS_OK;
}

template <int MODE>
static int launch_balanced(const gss_csr *a, int d4, const float *x, const SpmmEpi &ep, hipStream_t st) {
  int ns = K().spmm_slices;
  bool pin = K().spmm_pin != 0;
  if (ns == 0) {
    // automatic: cut the features into 256-B slices when the gathered operand is well beyond one XCD's 4 MB L2.
    //  * small operands (<= 64 MB, the Infinity Cache holds them whole): slices pinned to XCDs -- slice = workgroup
    //    id % ns and workgroups go to the XCDs round-robin, so every L2 caches 1/ns of the table.  Measured at
    //    N = 29,960: d = 128 36.0 -> 31.9 us, d = 256 90 -> 63 us, d = 512 212 -> 122 us (time-separated slices:
    //    37.6 / 68 / 129 us).
    //  * larger operands (RMAT scale): time-separated slices -- grid.y is the slow dispatch dimension, so the caches
    //    hold one slice of the table at a time.  N = 250k: d = 128 319 -> 260 us, d = 256 768 -> 568 us; N = 1M:
    //    d = 128 1.74 -> 1.50 ms, d = 256 3.09 -> 2.76 ms; pinning is 2-7 % behind there.
    const double table = 16.0 * d4 * (double)a->n_cols;
    ns = 1;
    pin = false;
    if (d4 >= 32 && d4 % 16 == 0 && table > 8.0 * 1024 * 1024) {
      ns = d4 / 16 > 8 ? 8 : d4 / 16;
      while (d4 % ns != 0) --ns;
      pin = table <= 64.0 * 1024 * 1024;
    }
  }
  if (ns < 1 || d4 % ns != 0 || (d4 / ns) < 4 || MODE == SPMM_BWD1S) ns = 1;  // the sparse mode gathers few rows anyway
  pin = pin && ns > 1;
  const int ds = d4 / ns;
  if (ds <= 4) return launch_balanced_t<MODE, 2, 1>(a, ds, ns, pin, x, ep, st);
  if (ds <= 8) return launch_balanced_t<MODE, 3, 1>(a, ds, ns, pin, x, ep, st);
  if (ds <= 16) return launch_balanced_t<MODE, 4, 1>(a, ds, ns, pin, x, ep, st);
  if (ds <= 32) return launch_balanced_t<MODE, 5, 1>(a, ds, ns, pin, x, ep, st);
  if (ds <= 64) return launch_balanced_t<MODE, 6, 1>(a, ds, ns, pin, x, ep, st);
  if (ds <= 128) return launch_balanced_t<MODE, 6, 2>(a, ds, ns, pin, x, ep, st);
  return launch_balanced_t<MODE, 6, 4>(a, ds, ns, pin, x, ep, st);
}

}  // namespace gss

// ---- giant rows --------------------------------------------------------------------------------------------------------------
// A row with more segments than a workgroup has lane groups is ONE workgroup's job (longer segments, segments.h): fine for the hubs of a
// protein graph (thousands of entries), a tail for the hubs of a scale-free graph with 10^7 nodes -- the top rows of RMAT 10M / 200M hold
// ~270 k entries each, and the workgroup that walks one gathers for milliseconds while the rest of the launch has drained (on ONE GPU it
// starts first and hides under a 30 ms launch; on the shard that owns the hubs it doubled the launch: 5.0 ms against 2.6-3.0 ms on the
// shards next to it with as many entries, profiles/r05_scaling_forecast_c5.json).  Rows above `spmm_giant` entries (default 32,768)
// are therefore cut into CHUNKS of a quarter of that, and a dense-mode product runs as three launches of the same kernel:
//   1. chunks:  a view whose rows are the chunks -- entry ranges of the ORIGINAL col / val arrays -- PLAIN product into a scratch
//               [chunks][d]; the caller's row filters are looked up at the chunk's row (SpmmEpi::row_alias);
//   2. short:   the caller's product and epilogue over a schedule WITHOUT the giant rows;
//   3. finish:  the caller's product and epilogue over a [rows] x [chunks] matrix of ones -- a giant row = the sum of its chunks, in
//               chunk order -- gathered from the scratch.
// Every row is still written by exactly one lane group, every sum has one fixed order (inside a chunk as inside a row; chunks in
// order): bit-reproducible, and the same on one GPU and on a shard (a row's length does not depend on who owns it).  A giant row's sum
// is grouped differently from an unchunked one (rounding-level).  The sparse walks (the batch-sparse backward hops, the gather-filtered
// first pass) keep the single schedule: they test entries, they do not gather for most of them.
struct gss_giant_rows {
  int32_t n_giant = 0, n_chunks = 0;
  gss_csr chunks{}, shortv{}, finish{};
  int32_t *d_chunk_row = nullptr, *d_fin_col = nullptr;
  float *d_fin_val = nullptr;
  // the chunks' partial sums [n_chunks][d]: one buffer per stream the handle is used on (a plan's pipelined / overlapped hops run products
  // of one matrix on two streams), grown to the widest d seen -- nothing is allocated per product
  struct Scratch {
    hipStream_t st;
    float *buf;
    size_t floats;
  };
  std::vector<Scratch> scratch;
  int scratch_for(hipStream_t st, size_t floats, float **out) {
    for (Scratch &s : scratch)
      if (s.st == st) {
        if (s.floats < floats) {
          GSS_HIP(hipStreamSynchronize(st));   // (a product of a narrower d may still read it)
          GSS_HIP(hipFree(s.buf));
          s.buf = nullptr;
          s.floats = 0;
          GSS_HIP(hipMalloc((void **)&s.buf, sizeof(float) * floats));
          s.floats = floats;
        }
        *out = s.buf;
        return GSS_OK;
      }
    float *b = nullptr;
    GSS_HIP(hipMalloc((void **)&b, sizeof(float) * floats));
    scratch.push_back(Scratch{st, b, floats});
    *out = b;
    return GSS_OK;
  }
  size_t scratch_bytes() const {
    size_t t = 0;
    for (const Scratch &s : scratch) t += sizeof(float) * s.floats;
    return t;
  }
  ~gss_giant_rows() {
    for (Scratch &s : scratch)
      if (s.buf) (void)hipFree(s.buf);
    for (gss_csr *v : {&chunks, &shortv, &finish})
      for (int k = 0; k < 5; ++k) {
        if (v->d_segs[k]) (void)hipFree(v->d_segs[k]);
        if (v->d_live[k]) (void)hipFree(v->d_live[k]);
      }
    if (d_chunk_row) (void)hipFree(d_chunk_row);
    if (d_fin_col) (void)hipFree(d_fin_col);
    if (d_fin_val) (void)hipFree(d_fin_val);
  }
};

namespace gss {

static void giant_view_init(gss_csr &v, const gss_csr *a, int32_t n_rows, int32_t n_cols) {
  v.n_rows = n_rows;
  v.n_cols = n_cols;
  v.nnz = 0;
  v.rowptr = a->rowptr;   // (the balanced kernel reads descriptors, not rowptr)
  v.col = a->col;
  v.val = a->val;
  v.n_long = 0;
  v.d_long_rows = nullptr;
  v.max_row = 0;
  v.hot_own = a->hot_own;
  v.hot_halo0 = a->hot_halo0;
  v.hot_halo1 = a->hot_halo1;
  for (int k = 0; k < 5; ++k) {
    v.d_segs[k] = nullptr;
    v.n_seg_blocks[k] = 0;
  }
  v.by_items = true;
}

// the chunked views of `a`, or NULL when it has no giant row (looked at once per handle, with the knob as it stands then)
static int csr_giant_rows(const gss_csr *a, gss_giant_rows **out) {
  gss_csr *m = const_cast<gss_csr *>(a);   // lazily filled cache; a gss_csr is used from one host thread
  *out = nullptr;
  if (m->by_items) return GSS_OK;          // a view itself
  const int thr = K().spmm_giant;
  if (m->giant_threshold == thr) {
    *out = m->giant;
    return GSS_OK;
  }
  delete m->giant;
  m->giant = nullptr;
  m->giant_threshold = thr;
  if (thr <= 0 || m->h_rowptr.empty()) return GSS_OK;
  const int32_t *rp = m->h_rowptr.data();
  int64_t n_giant = 0;
  for (int32_t r = 0; r < a->n_rows; ++r) n_giant += (rp[r + 1] - rp[r]) > thr;
  if (n_giant == 0) return GSS_OK;
  GiantItems items;
  (void)giant_items(rp, a->n_rows, thr, items);     // segments.h: host-only, checked under ASan / UBSan by tests/native/segments_check.cpp
  gss_giant_rows *g = new gss_giant_rows();
  giant_view_init(g->chunks, a, 0, a->n_cols);
  giant_view_init(g->shortv, a, a->n_rows, a->n_cols);
  giant_view_init(g->finish, a, a->n_rows, 0);
  auto fill = [](gss_csr &v, const std::vector<SegItem> &src) {
    v.item_row.reserve(src.size());
    v.item_first.reserve(src.size());
    v.item_len.reserve(src.size());
    for (const SegItem &it : src) {
      v.item_row.push_back(it.row);
      v.item_first.push_back(it.first);
      v.item_len.push_back(it.len);
    }
  };
  fill(g->shortv, items.shortv);
  fill(g->chunks, items.chunks);
  fill(g->finish, items.finish);
  const std::vector<int32_t> &chunk_row = items.chunk_row;
  g->n_giant = (int32_t)n_giant;
  g->n_chunks = (int32_t)chunk_row.size();
  g->chunks.n_rows = g->n_chunks;
  g->finish.n_cols = g->n_chunks;
  g->finish.hot_own = -1;                  // the scratch is small: no hot / cold split
  std::vector<int32_t> fin_col((size_t)g->n_chunks);
  std::vector<float> fin_val((size_t)g->n_chunks, 1.0f);
  for (int32_t k = 0; k < g->n_chunks; ++k) fin_col[(size_t)k] = k;
  hipError_t e = hipMalloc((void **)&g->d_chunk_row, sizeof(int32_t) * (size_t)g->n_chunks);
  if (e == hipSuccess) e = hipMalloc((void **)&g->d_fin_col, sizeof(int32_t) * (size_t)g->n_chunks);
  if (e == hipSuccess) e = hipMalloc((void **)&g->d_fin_val, sizeof(float) * (size_t)g->n_chunks);
  if (e == hipSuccess) e = hipMemcpy(g->d_chunk_row, chunk_row.data(), sizeof(int32_t) * (size_t)g->n_chunks, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(g->d_fin_col, fin_col.data(), sizeof(int32_t) * (size_t)g->n_chunks, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(g->d_fin_val, fin_val.data(), sizeof(float) * (size_t)g->n_chunks, hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    delete g;
    return fail(GSS_EHIP, "spmm: giant-row tables -> %s", hipGetErrorString(e));
  }
  g->finish.col = g->d_fin_col;
  g->finish.val = g->d_fin_val;
  m->giant = g;
  *out = g;
  return GSS_OK;
}

template <int MODE>
static int launch_giant(const gss_csr *a, gss_giant_rows *g, int d4, const float *x, const SpmmEpi &ep, hipStream_t st) {
  (void)a;
  float *scratch = nullptr;
  int rc = g->scratch_for(st, 4 * (size_t)d4 * (size_t)g->n_chunks, &scratch);
  if (rc != GSS_OK) return rc;
  {
    // 1. the chunks' partial sums; only chunks of rows the caller's product computes at all
    SpmmEpi pe{};
    pe.o0 = scratch;
    pe.row_alias = g->d_chunk_row;
    if (MODE == SPMM_PLAIN) {
      pe.pos = ep.pos;
      pe.rowbits = ep.rowbits;
    }
    if (MODE == SPMM_FWD1) pe.rowbits = ep.posbits;   // (FWD1 keeps its row bitmap in posbits)
    rc = launch_balanced<SPMM_PLAIN>(&g->chunks, d4, x, pe, st);
  }
  // 2. every other row: the caller's product as it is (the batch-preparation side job of a FWD1 launch rides here)
  if (rc == GSS_OK) rc = launch_balanced<MODE>(&g->shortv, d4, x, ep, st);
  // 3. the giant rows: the sum of their chunks + the caller's epilogue
  if (rc == GSS_OK) {
    SpmmEpi fe = ep;
    fe.prep = BatchPrep{};
    rc = launch_balanced<MODE>(&g->finish, d4, scratch, fe, st);
  }
  return rc;
}

template <int MODE>
static int launch_spmm(const gss_csr *a, int32_t d, const float *x, const SpmmEpi &ep, void *stream) {
  if (int rc = check_d(d)) return rc;
  GSS_REQUIRE(a && x, "spmm: null operand");
  hipStream_t st = as_stream(stream);
  const int d4 = d / 4;
  if (K().spmm_variant == 2) {
    constexpr bool kDense = MODE == SPMM_FWD1 || MODE == SPMM_PLAIN || MODE == SPMM_BWD1 || MODE == SPMM_BWD2;
    if (kDense && !(MODE == SPMM_PLAIN && ep.gather_bits)) {
      gss_giant_rows *g = nullptr;
      if (int rc = csr_giant_rows(a, &g)) return rc;
      if (g) return launch_giant<MODE>(a, g, d4, x, ep, st);
    }
    return launch_balanced<MODE>(a, d4, x, ep, st);
  }
  if (d4 <= 4) return launch_spmm_t<MODE, 2, 1>(a, d4, x, ep, st);
  if (d4 <= 8) return launch_spmm_t<MODE, 3, 1>(a, d4, x, ep, st);
  if (d4 <= 16) return launch_spmm_t<MODE, 4, 1>(a, d4, x, ep, st);
  if (d4 <= 32) return launch_spmm_t<MODE, 5, 1>(a, d4, x, ep, st);
  if (d4 <= 64) return launch_spmm_t<MODE, 6, 1>(a, d4, x, ep, st);
  if (d4 <= 128) return launch_spmm_t<MODE, 6, 2>(a, d4, x, ep, st);
  return launch_spmm_t<MODE, 6, 4>(a, d4, x, ep, st);
}

// internal entry points shared with plan.hip
int spmm_fwd(const gss_csr *a, int32_t d, const float *x, float *y, const float *h, float *m, void *stream, const int32_t *row_pos,
             const uint32_t *row_bits, const float *y_in, const uint32_t *gather_bits, const BatchPrep *prep) {
  GSS_REQUIRE(y, "spmm: y is null");
  GSS_REQUIRE(!prep || (m && K().spmm_variant == 2 && prep->idx && prep->pos && prep->b > 0), "spmm: the batch-preparation side job goes with the Hadamard-fused product of the balanced SpMM");
  GSS_REQUIRE(!row_pos || (!m && K().spmm_variant == 2), "spmm: a row map goes with the plain product of the balanced SpMM only");
  GSS_REQUIRE(!row_bits || K().spmm_variant == 2, "spmm: a row bitmap needs the balanced SpMM (spmm_variant 2)");
  GSS_REQUIRE((!y_in && !gather_bits) || K().spmm_variant == 2, "spmm: a two-pass product needs the balanced SpMM (spmm_variant 2)");
  GSS_REQUIRE(!gather_bits || !m, "spmm: a gather filter goes with the plain product only");
  if (m) {
    GSS_REQUIRE(h, "spmm: m given without h");
    SpmmEpi ep{h, nullptr, nullptr, y, m, 0.f, nullptr, nullptr, row_bits};
    ep.y_in = y_in;
    if (prep) ep.prep = *prep;
    return launch_spmm<SPMM_FWD1>(a, d, x, ep, stream);
  }
  SpmmEpi ep{nullptr, nullptr, nullptr, y, nullptr, 0.f, row_pos, nullptr, nullptr};
  ep.y_in = y_in;
  ep.gather_bits = gather_bits;
  ep.rowbits = row_bits;
  return launch_spmm<SPMM_PLAIN>(a, d, x, ep, stream);
}

int spmm_bwd1(const gss_csr *at, int32_t d, const float *g_am, const float *g_ax, const float *x_in, const float *ax,
              float *u, float *t, void *stream, const float *y_in) {
  GSS_REQUIRE(g_am && g_ax && x_in && ax && u && t, "spmm_bwd1: null operand");
  GSS_REQUIRE(!y_in || K().spmm_variant == 2, "spmm_bwd1: a two-pass product needs the balanced SpMM (spmm_variant 2)");
  SpmmEpi ep{g_ax, x_in, ax, u, t, 0.f, nullptr, nullptr, nullptr, nullptr};
  ep.y_in = y_in;
  return launch_spmm<SPMM_BWD1>(at, d, g_am, ep, stream);
}

int spmm_bwd1_sparse(const gss_csr *at, int32_t d, const float *g_am_b, const float *g_ax_b, const int32_t *pos,
                     const int32_t *pos_row, const float *x_in, const float *ax, float *u, float *t, void *stream, const uint32_t *posbits,
                     uint32_t *nzbits_out, int skip_zero_rows, const uint32_t *live_rows) {
  GSS_REQUIRE(g_am_b && g_ax_b && pos && pos_row && x_in && ax && u && t, "spmm_bwd1_sparse: null operand");
  GSS_REQUIRE(K().spmm_variant == 2, "spmm_bwd1_sparse needs the balanced SpMM (spmm_variant 2)");
  GSS_REQUIRE(!skip_zero_rows || nzbits_out, "spmm_bwd1_sparse: skipping the zero rows needs the bitmap that records them");
  SpmmEpi ep{g_ax_b, x_in, ax, u, t, 0.f, pos, pos_row, posbits, nzbits_out, skip_zero_rows};
  ep.rowbits = live_rows;
  return launch_spmm<SPMM_BWD1S>(at, d, g_am_b, ep, stream);
}

int spmm_bwd2_sparse_res(const gss_csr *at, int32_t d, const float *u, const float *t, const float *p, float c, const float *res_b,
                         const int32_t *pos_row, float *dp, float *gx_out, void *stream, const uint32_t *nzbits, const float *y_in,
                         int32_t pos_row_limit) {
  GSS_REQUIRE(u && t && p && res_b && pos_row && dp, "spmm_bwd2_sparse_res: null operand");
  GSS_REQUIRE(K().spmm_variant == 2, "spmm_bwd2_sparse_res needs the balanced SpMM (spmm_variant 2)");
  SpmmEpi ep{t, p, res_b, dp, gx_out, c, nullptr, pos_row, nzbits, nullptr, 0};
  ep.y_in = y_in;
  ep.pos_row_limit = pos_row_limit;
  return launch_spmm<SPMM_BWD2S>(at, d, u, ep, stream);
}

bool spmm_sparse_available() { return K().spmm_variant == 2; }

// bits[r] := 1 for every listed row r and every column of the listed rows of `a` (one wave per listed row)
__global__ __launch_bounds__(256) void mark_rows_and_neighbours_kernel(CsrView a, const int32_t *__restrict__ rows, int b, uint32_t *__restrict__ bits) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= b) return;
  const int r = rows[i];
  if (r < 0) return;   // a row-list entry another shard owns
  if (lane == 0) atomicOr(&bits[(unsigned)r >> 5], 1u << (r & 31));
  for (int e = a.rowptr[r] + lane; e < a.rowptr[r + 1]; e += 64) {
    const int c = a.col[e];
    atomicOr(&bits[(unsigned)c >> 5], 1u << (c & 31));
  }
}

int mark_rows_and_neighbours(const gss_csr *a, const int32_t *rows, int32_t b, uint32_t *bits, void *stream) {
  GSS_REQUIRE(a && rows && bits && b >= 0, "mark_rows_and_neighbours: null operand");
  if (b == 0) return GSS_OK;
  CsrView v{a->rowptr, a->col, a->val, a->n_rows};
  hipLaunchKernelGGL(mark_rows_and_neighbours_kernel, dim3(ceil_div(b, 4)), dim3(256), 0, as_stream(stream), v, rows, b, bits);
  GSS_LAUNCH_CHECK("mark_rows_and_neighbours_kernel");
  return GSS_OK;
}

int spmm_bwd2(const gss_csr *at, int32_t d, const float *u, const float *t, const float *p, float c, const float *res,
              float *dp, float *gx_out, void *stream, const float *y_in, int32_t own_row_limit) {
  GSS_REQUIRE(u && t && p && dp, "spmm_bwd2: null operand");
  GSS_REQUIRE(!y_in || K().spmm_variant == 2, "spmm_bwd2: a two-pass product needs the balanced SpMM (spmm_variant 2)");
  SpmmEpi ep{t, p, res, dp, gx_out, c, nullptr, nullptr, nullptr, nullptr};
  ep.y_in = y_in;
  ep.pos_row_limit = own_row_limit;
  GSS_REQUIRE(own_row_limit == 0 || K().spmm_variant == 2, "spmm_bwd2: a row limit needs the balanced SpMM (spmm_variant 2)");
  return launch_spmm<SPMM_BWD2>(at, d, u, ep, stream);
}

// ---- K11 normalize_adj ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rowsum_kernel(int n, const int32_t *__restrict__ rowptr, const double *__restrict__ val,
                                                     double *__restrict__ dinv, double *__restrict__ rowsum) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  const int e0 = rowptr[row], e1 = rowptr[row + 1];
  double s = 0.0;
  for (int e = e0 + lane; e < e1; e += 64) s += val[e];
  s = wave_sum_d(s);
  if (lane == 0) {
    dinv[row] = pow(s, -0.5);  // np.power(rowsum, -0.5), helper.py:85
    if (rowsum) rowsum[row] = s;
  }
}

// rows whose D_ii is not > 0 (negative, zero or NaN): their D_ii^-1/2 is NaN / inf, and so is every entry of their row and column
__global__ __launch_bounds__(256) void rowsum_check_kernel(int n, const double *__restrict__ rowsum, unsigned long long *__restrict__ count,
                                                           int *__restrict__ first) {
  const int row = blockIdx.x * 256 + threadIdx.x;
  const bool bad = row < n && !(rowsum[row] > 0.0);
  const unsigned long long m = __ballot(bad);
  if (m == 0) return;
  if ((threadIdx.x & 63) == __ffsll((long long)m) - 1) {   // the wave's first bad lane reports for the wave
    atomicAdd(count, (unsigned long long)__popcll(m));
    atomicMin(first, row);
  }
}

__global__ __launch_bounds__(256) void scale_kernel(int n, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                    const double *__restrict__ val, const double *__restrict__ dinv,
                                                    float *__restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  const int e0 = rowptr[row], e1 = rowptr[row + 1];
  const double di = dinv[row];
  // (A_ D^-1/2)^T D^-1/2 ^T : column scale first, then row scale (helper.py:86-87)
  for (int e = e0 + lane; e < e1; e += 64) out[e] = (float)((val[e] * dinv[col[e]]) * di);
}

// sharded form of scale_kernel: the shard holds rows [row0, row0 + n) of A + I (or of its transpose) with GLOBAL column ids;
// dinv is the all-gathered D^-1/2 of every node.  The rounding sequence is the single-GPU kernel's: (val * dinv[column of
// A]) * dinv[row of A]; for a transposed shard the entry (i, j) is A[j][i], so the roles of the shard's row and column swap
// and the result is bit-identical to the matching entry of A_hat.
__global__ __launch_bounds__(256) void scale_shard_kernel(int n, int row0, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                          const double *__restrict__ val, const double *__restrict__ dinv, int transposed,
                                                          float *__restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  const int e0 = rowptr[row], e1 = rowptr[row + 1];
  const double di = dinv[row0 + row];
  if (transposed) {
    for (int e = e0 + lane; e < e1; e += 64) out[e] = (float)((val[e] * di) * dinv[col[e]]);
  } else {
    for (int e = e0 + lane; e < e1; e += 64) out[e] = (float)((val[e] * dinv[col[e]]) * di);
  }
}

}  // namespace gss

using namespace gss;

extern "C" {

int gss_rowsum_dinv(int32_t n, const int32_t *rowptr, const double *val, double *dinv_out, double *rowsum_out, void *stream) {
  if (n == 0) return GSS_OK;  // an empty shard
  GSS_REQUIRE(n > 0 && rowptr && val && dinv_out, "rowsum_dinv: null operand");
  hipLaunchKernelGGL(rowsum_kernel, dim3(ceil_div(n, 4)), dim3(256), 0, as_stream(stream), n, rowptr, val, dinv_out, rowsum_out);
  GSS_LAUNCH_CHECK("rowsum_kernel");
  return GSS_OK;
}

int gss_rowsum_check(int32_t n, const double *rowsum, int64_t *h_count_out, int32_t *h_first_out, void *stream) {
  GSS_REQUIRE(n >= 0 && (n == 0 || rowsum) && h_count_out, "rowsum_check: null operand");
  *h_count_out = 0;
  if (h_first_out) *h_first_out = -1;
  if (n == 0) return GSS_OK;
  hipStream_t st = as_stream(stream);
  struct Res {
    unsigned long long count;
    int first;
  } h{0ull, INT32_MAX};
  Res *dres = nullptr;
  GSS_HIP(hipMallocAsync((void **)&dres, sizeof(Res), st));
  GSS_HIP(hipMemcpyAsync(dres, &h, sizeof(Res), hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(rowsum_check_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, st, n, rowsum, &dres->count, &dres->first);
  GSS_LAUNCH_CHECK("rowsum_check_kernel");
  GSS_HIP(hipMemcpyAsync(&h, dres, sizeof(Res), hipMemcpyDeviceToHost, st));
  GSS_HIP(hipFreeAsync(dres, st));
  GSS_HIP(hipStreamSynchronize(st));   // a setup-time check: the answer is a host value
  *h_count_out = (int64_t)h.count;
  if (h_first_out) *h_first_out = h.count ? h.first : -1;
  return GSS_OK;
}

int gss_scale_adj_shard(int32_t n, int32_t row0, const int32_t *rowptr, const int32_t *col, const double *val, const double *dinv_global,
                        int32_t transposed, float *val_out, void *stream) {
  if (n == 0) return GSS_OK;  // an empty shard
  GSS_REQUIRE(n > 0 && row0 >= 0 && rowptr && dinv_global && val_out && col && val, "scale_adj_shard: null operand");
  hipLaunchKernelGGL(scale_shard_kernel, dim3(ceil_div(n, 4)), dim3(256), 0, as_stream(stream), n, row0, rowptr, col, val, dinv_global,
                     transposed, val_out);
  GSS_LAUNCH_CHECK("scale_shard_kernel");
  return GSS_OK;
}

int gss_normalize_adj(int32_t n, const int32_t *rowptr, const int32_t *col, const double *val, float *val_out,
                      double *rowsum_out, void *stream) {
  GSS_REQUIRE(n >= 0 && rowptr && col && val && val_out, "normalize_adj: null operand");
  if (n == 0) return GSS_OK;
  hipStream_t st = as_stream(stream);
  double *dinv = nullptr;
  GSS_HIP(hipMallocAsync((void **)&dinv, sizeof(double) * (size_t)n, st));
  const int nblk = ceil_div(n, 4);
  hipLaunchKernelGGL(rowsum_kernel, dim3(nblk), dim3(256), 0, st, n, rowptr, val, dinv, rowsum_out);
  GSS_LAUNCH_CHECK("rowsum_kernel");
  hipLaunchKernelGGL(scale_kernel, dim3(nblk), dim3(256), 0, st, n, rowptr, col, val, dinv, val_out);
  GSS_LAUNCH_CHECK("scale_kernel");
  GSS_HIP(hipFreeAsync(dinv, st));
  return GSS_OK;
}

int gss_csr_create(gss_csr **out, int32_t n_rows, int32_t n_cols, int64_t nnz, const int32_t *h_rowptr,
                   const int32_t *d_rowptr, const int32_t *d_col, const float *d_val) {
  GSS_REQUIRE(out && h_rowptr && d_rowptr && (nnz == 0 || (d_col && d_val)), "csr_create: null operand");
  GSS_REQUIRE(n_rows >= 0 && n_cols >= 0 && nnz >= 0 && nnz < (int64_t)INT32_MAX, "csr_create: bad sizes n_rows=%d n_cols=%d nnz=%lld",
              n_rows, n_cols, (long long)nnz);
  GSS_REQUIRE(h_rowptr[0] == 0 && (int64_t)h_rowptr[n_rows] == nnz, "csr_create: rowptr[0]=%d rowptr[n]=%d do not match nnz=%lld",
              h_rowptr[0], h_rowptr[n_rows], (long long)nnz);
  gss_csr *a = new gss_csr();
  a->n_rows = n_rows;
  a->n_cols = n_cols;
  a->nnz = nnz;
  a->rowptr = d_rowptr;
  a->col = d_col;
  a->val = d_val;
  a->n_long = 0;
  a->hot_own = -1;
  a->hot_halo0 = a->hot_halo1 = 0;
  a->d_long_rows = nullptr;
  a->max_row = 0;
  for (int k = 0; k < 5; ++k) {
    a->d_segs[k] = nullptr;
    a->n_seg_blocks[k] = 0;
  }
  int32_t *h_long = nullptr;
  int n_long = 0;
  for (int pass = 0; pass < 2; ++pass) {
    n_long = 0;
    for (int32_t r = 0; r < n_rows; ++r) {
      const int32_t len = h_rowptr[r + 1] - h_rowptr[r];
      if (len < 0) {
        delete[] h_long;
        delete a;
        return fail(GSS_EINVAL, "csr_create: rowptr not monotone at row %d", r);
      }
      if (len > a->max_row) a->max_row = len;
      if (len > kLongRow) {
        if (pass == 1) h_long[n_long] = r;
        ++n_long;
      }
    }
    if (pass == 0) {
      if (n_long == 0) break;
      h_long = new int32_t[n_long];
    }
  }
  if (n_long > 0) {
    hipError_t e = hipMalloc((void **)&a->d_long_rows, sizeof(int32_t) * (size_t)n_long);
    if (e == hipSuccess) e = hipMemcpy(a->d_long_rows, h_long, sizeof(int32_t) * (size_t)n_long, hipMemcpyHostToDevice);
    delete[] h_long;
    if (e != hipSuccess) {
      if (a->d_long_rows) (void)hipFree(a->d_long_rows);
      delete a;
      return fail(GSS_ENOMEM, "csr_create: long-row list: %s", hipGetErrorString(e));
    }
    a->n_long = n_long;
  }
  a->h_rowptr.assign(h_rowptr, h_rowptr + (size_t)n_rows + 1);
  *out = a;
  return GSS_OK;
}

int gss_csr_set_hot(gss_csr *a, int32_t own_hot, int32_t halo_begin, int32_t halo_end) {
  GSS_REQUIRE(a, "csr_set_hot: null handle");
  GSS_REQUIRE(own_hot >= -1 && halo_begin >= 0 && halo_end >= halo_begin && halo_end <= a->n_cols, "csr_set_hot: bad ranges %d [%d, %d)", own_hot,
              halo_begin, halo_end);
  a->hot_own = own_hot;
  a->hot_halo0 = halo_begin;
  a->hot_halo1 = halo_end;
  if (a->giant)   // the views that gather from the same operand follow
    for (gss_csr *v : {&a->giant->chunks, &a->giant->shortv}) {
      v->hot_own = own_hot;
      v->hot_halo0 = halo_begin;
      v->hot_halo1 = halo_end;
    }
  return GSS_OK;
}

int gss_csr_giant_rows(const gss_csr *a, int32_t *n_rows_out, int32_t *n_chunks_out) {
  GSS_REQUIRE(a && n_rows_out && n_chunks_out, "csr_giant_rows: null argument");
  KnobScope none(nullptr);
  gss_giant_rows *g = nullptr;
  if (int rc = csr_giant_rows(a, &g)) return rc;
  *n_rows_out = g ? g->n_giant : 0;
  *n_chunks_out = g ? g->n_chunks : 0;
  return GSS_OK;
}

void gss_csr_destroy(gss_csr *a) {
  if (!a) return;
  delete a->giant;
  if (a->d_long_rows) (void)hipFree(a->d_long_rows);
  for (int k = 0; k < 5; ++k) {
    if (a->d_segs[k]) (void)hipFree(a->d_segs[k]);
    if (a->d_live[k]) (void)hipFree(a->d_live[k]);
  }
  delete a;
}


int gss_spmm(const gss_csr *a, int32_t d, const float *x, float *y, const float *h, float *m, void *stream) {
  return spmm_fwd(a, d, x, y, h, m, stream);
}
int gss_spmm_add(const gss_csr *a, int32_t d, const float *x, const float *y_in, float *y, const float *h, float *m, void *stream) {
  GSS_REQUIRE(y_in, "spmm_add: y_in is null");
  return spmm_fwd(a, d, x, y, h, m, stream, nullptr, nullptr, y_in);
}
int gss_spmm_bwd1(const gss_csr *at, int32_t d, const float *g_am, const float *g_ax, const float *x_in, const float *ax,
                  float *u, float *t, void *stream) {
  return spmm_bwd1(at, d, g_am, g_ax, x_in, ax, u, t, stream);
}
int gss_spmm_bwd1_sparse(const gss_csr *at, int32_t d, const float *g_am_b, const float *g_ax_b, const int32_t *pos_col,
                         const int32_t *pos_row, const float *x_in, const float *ax, float *u, float *t, void *stream) {
  return spmm_bwd1_sparse(at, d, g_am_b, g_ax_b, pos_col, pos_row, x_in, ax, u, t, stream);
}
int gss_spmm_bwd2(const gss_csr *at, int32_t d, const float *u, const float *t, const float *p, float c, const float *res,
                  float *dp, float *gx_out, void *stream) {
  return spmm_bwd2(at, d, u, t, p, c, res, dp, gx_out, stream);
}
}
