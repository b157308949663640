// elementwise.hip -- row normalisation (K5), its backward fused with ELU' on the batch rows,
// row scatter-add, Adam (K10) and weight transposes.  All HBM/L2-bound streaming kernels.
#include "ops.h"

namespace gss {

// ---- K5  F.normalize(x, dim=1), modules/model.py:205 -----------------------------------------------
// A row of d floats is covered by LPR = min(64, pow2ceil(d/4)) lanes holding VPL float4 each, so one
// wave normalises 64/LPR rows; the sum of squares is reduced with xor-shuffles inside the lane group.
template <int VPL>
__global__ __launch_bounds__(256) void rownorm_fwd_kernel(int n, int d4, int lpr_log2, const float *__restrict__ x,
                                                          float *__restrict__ e, float *__restrict__ inv_den, const int32_t *__restrict__ rows) {
  const int lane = threadIdx.x & 63;
  const int lpr = 1 << lpr_log2;
  const int rpw = 64 >> lpr_log2;
  const int li = lane & (lpr - 1);
  const int ri = (blockIdx.x * 4 + (threadIdx.x >> 6)) * rpw + (lane >> lpr_log2);
  // a row list: the n listed rows of x / e / inv_den (gss_plan_step_lazy); a negative entry = a batch member another shard owns: skipped
  const int listed = (ri < n && rows) ? rows[ri] : ri;
  const bool ok = ri < n && listed >= 0;
  const int row = ok ? listed : 0;
  float4 v[VPL];
  float ss = 0.f;
#pragma unroll
  for (int k = 0; k < VPL; ++k) {
    const int f4 = li + k * 64;
    v[k] = (ok && f4 < d4) ? ld4(x + ((size_t)row * d4 + f4) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    ss += v[k].x * v[k].x + v[k].y * v[k].y + v[k].z * v[k].z + v[k].w * v[k].w;
  }
  for (int o = 1; o < lpr; o <<= 1) ss += __shfl_xor(ss, o, 64);
  const float den = fmaxf(sqrtf(ss), 1e-12f);
  if (!ok) return;
#pragma unroll
  for (int k = 0; k < VPL; ++k) {
    const int f4 = li + k * 64;
    if (f4 < d4) st4(e + ((size_t)row * d4 + f4) * 4, unit4(den, v[k]));
  }
  if (li == 0) inv_den[row] = 1.f / den;
}

// backward of F.normalize and of F.elu / the residual mix on the batch rows (autograd of
// model.py:173,201-205): dx = (de - e (e . de)) * inv_den ; dp = c * dx (.) elu'(p)
template <int VPL>
__global__ __launch_bounds__(256) void rownorm_elu_bwd_kernel(int b, int d4, int lpr_log2, const float *__restrict__ de_b,
                                                              const int32_t *__restrict__ idx, const float *__restrict__ e,
                                                              const float *__restrict__ inv_den, const float *__restrict__ p,
                                                              float c, float *__restrict__ dx_b, float *__restrict__ dp_b,
                                                              int32_t *__restrict__ pos_set) {
  const int lane = threadIdx.x & 63;
  const int lpr = 1 << lpr_log2;
  const int rpw = 64 >> lpr_log2;
  const int li = lane & (lpr - 1);
  const int r = (blockIdx.x * 4 + (threadIdx.x >> 6)) * rpw + (lane >> lpr_log2);
  const bool ok = r < b;
  const int node = ok ? idx[r] : 0;
  float4 g[VPL], ev[VPL];
  float dot = 0.f;
#pragma unroll
  for (int k = 0; k < VPL; ++k) {
    const int f4 = li + k * 64;
    const bool in = ok && f4 < d4;
    g[k] = in ? ld4(de_b + ((size_t)r * d4 + f4) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    ev[k] = in ? ld4(e + ((size_t)node * d4 + f4) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    dot += g[k].x * ev[k].x + g[k].y * ev[k].y + g[k].z * ev[k].z + g[k].w * ev[k].w;
  }
  for (int o = 1; o < lpr; o <<= 1) dot += __shfl_xor(dot, o, 64);
  if (!ok) return;
  if (pos_set && li == 0) pos_set[node] = r;  // node -> batch position map for the sparsity-aware backward SpMM
  const float inv = inv_den[node];
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
    const float4 pg = elu_grad4(ld4(p + ((size_t)node * d4 + f4) * 4));
    st4(dp_b + ((size_t)r * d4 + f4) * 4, scale4(c, mul4(dx, pg)));
  }
}

__global__ __launch_bounds__(256) void scatter_add_rows_kernel(int b, int d4, const float *__restrict__ src,
                                                               const int32_t *__restrict__ rows, const float *__restrict__ keep,
                                                               float *__restrict__ dst, int32_t *__restrict__ pos_clear,
                                                               const int32_t *__restrict__ pos_ids) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)b * d4) return;
  const int r = (int)(i / d4), f4 = (int)(i % d4);
  if (pos_clear && f4 == 0 && pos_ids[r] >= 0) pos_clear[pos_ids[r]] = -1;
  if (rows[r] < 0 || (keep && keep[r] == 0.f)) return;  // not a row of this shard
  float *p = dst + ((size_t)rows[r] * d4 + f4) * 4;
  st4(p, add4(ld4(p), ld4(src + i * 4)));
}

// bitmap companion of the batch-position map (spmm.hip SPMM_BWD1S): set the members' bits / zero their words
__global__ void batch_bits_kernel(const int32_t *__restrict__ ids, int b, uint32_t *__restrict__ bits, int set) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= b) return;
  const int id = ids[i];
  if (id < 0) return;
  if (set)
    atomicOr(&bits[(unsigned)id >> 5], 1u << (id & 31));
  else
    bits[(unsigned)id >> 5] = 0u;  // every set bit of the word belongs to a member of this batch
}

// gss_plan_step_lazy, before the forward pass: node id -> row (node_map, relabelled graphs) -> local row of this shard (lo, nl) /
// position-map id (gid2op, or the local row) -- the translation loss.hip's gather_rows_mapped_kernel does in the full step -- and the
// batch-position map pos[id] = position (on own rows it is the row mask of the top layer, later the key of the sparse backward hop)
__global__ void batch_prepare_kernel(const int32_t *__restrict__ idx, int b, const int32_t *__restrict__ node_map, int lo, int nl,
                                     const int32_t *__restrict__ gid2op, int32_t *__restrict__ rloc, int32_t *__restrict__ pid,
                                     float *__restrict__ keep, int32_t *__restrict__ pos, int32_t *__restrict__ rlist) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= b) return;
  const int id = node_map ? node_map[idx[i]] : idx[i];
  const int rel = id - lo;
  const bool mine = rel >= 0 && rel < nl;
  const int op = gid2op ? gid2op[id] : (mine ? rel : -1);
  if (rloc) rloc[i] = min(max(rel, 0), max(nl - 1, 0));
  if (pid) pid[i] = op;
  if (keep) keep[i] = mine ? 1.f : 0.f;
  if (rlist) rlist[i] = mine ? rel : -1;   // the row list of the lazy top layer: the members this shard owns, -1 for the others
  if (op >= 0) pos[op] = i;
}

__global__ void bits_fill_kernel(uint32_t *__restrict__ bits, long long first, long long last) {
  const long long w = first / 32 + (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (w > (last - 1) / 32) return;
  const long long b0 = w * 32;
  const int lo = (int)(first > b0 ? first - b0 : 0), hi = (int)(last < b0 + 32 ? last - b0 : 32);
  const uint32_t mask = (hi >= 32 ? 0xffffffffu : ((1u << hi) - 1u)) & ~((1u << lo) - 1u);
  bits[w] |= mask;   // one thread per word
}

// halo exchange, sender side: out[k] = src[rows[k]] -- the rows the peers reference, packed in peer order
__global__ __launch_bounds__(256) void pack_rows_kernel(int64_t n, int d4, const float *__restrict__ src, const int32_t *__restrict__ rows,
                                                        float *__restrict__ out) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)n * d4) return;
  const size_t r = i / d4, f4 = i % d4;
  st4(out + i * 4, ld4(src + ((size_t)rows[r] * d4 + f4) * 4));
}

// halo exchange, receiver side of a compacted exchange: dst[rows[k]] = src[k]
__global__ __launch_bounds__(256) void unpack_rows_kernel(int64_t n, int d4, const float *__restrict__ src, const int32_t *__restrict__ rows,
                                                          float *__restrict__ dst) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)n * d4) return;
  const size_t r = i / d4, f4 = i % d4;
  st4(dst + ((size_t)rows[r] * d4 + f4) * 4, ld4(src + i * 4));
}

// Lazy halo (plan.hip plan_halo_needed): which boundary rows do the listed rows of this shard read?  One wave per listed row; an entry
// whose column is halo slot h of owner q sets bit (h - recv_off[q]) of q's word range [wrecv_off[q], wrecv_off[q + 1]).
__global__ __launch_bounds__(256) void halo_need_mark_kernel(const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                             const int32_t *__restrict__ rows, int b, int n, int P,
                                                             const int64_t *__restrict__ recv_off, const int64_t *__restrict__ wrecv_off,
                                                             uint32_t *__restrict__ needw) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= b) return;
  const int r = rows[i];
  if (r < 0) return;   // a member another shard owns
  for (int e = rowptr[r] + lane; e < rowptr[r + 1]; e += 64) {
    const int c = col[e];
    if (c < n) continue;
    const int64_t h = c - n;
    int q = 0;
    while (q + 1 < P && recv_off[q + 1] <= h) ++q;
    const int64_t bit = wrecv_off[q] * 32 + (h - recv_off[q]);
    atomicOr(&needw[bit >> 5], 1u << (bit & 31));
  }
}

// Sender-driven form: word w of peer q's range [wsend_off[q], wsend_off[q + 1]) collects bits[send_rows[s]] of its 32 send slots
__global__ __launch_bounds__(256) void send_slot_bits_kernel(const uint32_t *__restrict__ bits, const int32_t *__restrict__ send_rows, int P,
                                                             const int64_t *__restrict__ send_off, const int64_t *__restrict__ wsend_off,
                                                             int64_t n_words, uint32_t *__restrict__ out) {
  const int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (w >= n_words) return;
  int q = 0;
  while (q + 1 < P && wsend_off[q + 1] <= w) ++q;
  const int64_t s0 = send_off[q] + (w - wsend_off[q]) * 32, s1 = send_off[q + 1];
  uint32_t word = 0;
  for (int j = 0; j < 32 && s0 + j < s1; ++j) {
    const int r = send_rows[s0 + j];
    word |= ((bits[(unsigned)r >> 5] >> (r & 31)) & 1u) << j;
  }
  out[w] = word;
}

// bits [first, last) := 0, other bits untouched (atomics: the edge words may be shared with concurrent writers of other bits)
__global__ __launch_bounds__(256) void bits_clear_kernel(uint32_t *__restrict__ bits, long long first, long long last) {
  const long long w = first / 32 + (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (w > (last - 1) / 32) return;
  const long long b0 = w * 32;
  const int lo = (int)(first > b0 ? first - b0 : 0), hi = (int)(last < b0 + 32 ? last - b0 : 32);
  const uint32_t mask = (hi >= 32 ? 0xffffffffu : ((1u << hi) - 1u)) & ~((1u << lo) - 1u);
  if (mask == 0xffffffffu)
    bits[w] = 0u;
  else
    atomicAnd(&bits[w], ~mask);
}

__global__ __launch_bounds__(256) void bits_set_list_kernel(uint32_t *__restrict__ bits, const int32_t *__restrict__ list, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int r = list[i];
  atomicOr(&bits[(unsigned)r >> 5], 1u << (r & 31));
}

// The set bits of P word ranges, in ascending order, as a list: bit j of range q is slot slot_off[q] + j; the list holds map[slot]
// (map != NULL) or slot + add.  out_off[q + 1] = entries up to and including range q.  Both sides of an exchange run it over the same
// bits, so their counts and orders agree by construction.  Three launches (round 5; one workgroup walking the whole bitmap took 1-2 ms at
// RMAT 10M, where a rank's halo is 10^5 words -- as long as everything the request phase is meant to hide under): the ranges are cut into
// blocks of 1024 words (never across a range), (1) every block counts its bits, (2) one workgroup turns the counts into block bases and
// the ranges' offsets, (3) every block lists its bits behind its base.
constexpr int kCompactWords = 1024;
// block `blk` of the concatenated block sequence -> its range q and first word (P is small: a linear walk)
__device__ __forceinline__ void compact_block(int blk, int P, const int64_t *__restrict__ woff, int &q_out, int64_t &w_first, int64_t &w_end) {
  int q = 0, first_blk = 0;
  for (; q < P; ++q) {
    const int nb = (int)((woff[q + 1] - woff[q] + kCompactWords - 1) / kCompactWords);
    if (blk < first_blk + nb) break;
    first_blk += nb;
  }
  q_out = q;
  w_first = woff[q] + (int64_t)(blk - first_blk) * kCompactWords;
  w_end = woff[q + 1];
}

__global__ __launch_bounds__(1024) void bits_count_kernel(const uint32_t *__restrict__ words, int P, const int64_t *__restrict__ woff,
                                                          int32_t *__restrict__ block_tot) {
  __shared__ int wave_tot[16];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  int q;
  int64_t w0, w1;
  compact_block((int)blockIdx.x, P, woff, q, w0, w1);
  const int64_t w = w0 + tid;
  int c = w < w1 ? __popc(words[w]) : 0;
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
  if (lane == 0) wave_tot[wv] = c;
  __syncthreads();
  if (tid == 0) {
    int t = 0;
    for (int k = 0; k < 16; ++k) t += wave_tot[k];
    block_tot[blockIdx.x] = t;
  }
}

// one workgroup: exclusive scan of the block counts (in place: block_tot[blk] becomes the block's base), the ranges' offsets
__global__ __launch_bounds__(1024) void bits_scan_kernel(int P, const int64_t *__restrict__ woff, int nblk, int32_t *__restrict__ block_tot,
                                                         int64_t *__restrict__ base_out, int64_t *__restrict__ out_off) {
  __shared__ long long wave_tot[16];
  __shared__ long long run;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (tid == 0) run = 0;
  __syncthreads();
  for (int b0 = 0; b0 < nblk; b0 += 1024) {
    const int blk = b0 + tid;
    const long long c = blk < nblk ? block_tot[blk] : 0;
    long long incl = c;
    for (int o = 1; o < 64; o <<= 1) {
      const long long v = __shfl_up(incl, o);
      if (lane >= o) incl += v;
    }
    if (lane == 63) wave_tot[wv] = incl;
    __syncthreads();
    long long pos = run + incl - c;
    for (int k = 0; k < wv; ++k) pos += wave_tot[k];
    if (blk < nblk) base_out[blk] = pos;
    __syncthreads();
    if (tid == 0) {
      long long t = 0;
      for (int k = 0; k < 16; ++k) t += wave_tot[k];
      run += t;
    }
    __syncthreads();
  }
  if (tid == 0) {
    base_out[nblk] = run;
    out_off[0] = 0;
    int first_blk = 0;
    for (int q = 0; q < P; ++q) {   // a range ends where its last block does
      first_blk += (int)((woff[q + 1] - woff[q] + kCompactWords - 1) / kCompactWords);
      out_off[q + 1] = base_out[first_blk];
    }
  }
}

__global__ __launch_bounds__(1024) void bits_write_kernel(const uint32_t *__restrict__ words, int P, const int64_t *__restrict__ woff,
                                                          const int64_t *__restrict__ slot_off, const int32_t *__restrict__ map, int32_t add,
                                                          const int64_t *__restrict__ base, int32_t *__restrict__ out) {
  __shared__ int wave_tot[16];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  int q;
  int64_t w0, w1;
  compact_block((int)blockIdx.x, P, woff, q, w0, w1);
  const int64_t w = w0 + tid;
  uint32_t bits = w < w1 ? words[w] : 0u;
  const int c = __popc(bits);
  int incl = c;
  for (int o = 1; o < 64; o <<= 1) {
    const int v = __shfl_up(incl, o);
    if (lane >= o) incl += v;
  }
  if (lane == 63) wave_tot[wv] = incl;
  __syncthreads();
  long long pos = base[blockIdx.x] + incl - c;
  for (int k = 0; k < wv; ++k) pos += wave_tot[k];
  const int64_t slot_base = slot_off[q] + (w - woff[q]) * 32;
  while (bits) {
    const int bpos = __ffs(bits) - 1;
    bits &= bits - 1;
    const int64_t slot = slot_base + bpos;
    out[pos++] = map ? map[slot] : (int32_t)(slot + add);
  }
}

// ---- K10  torch.optim.Adam (single-tensor form of torch/optim/adam.py; train.py:139-141,184) -------
__global__ __launch_bounds__(256) void adam_kernel(int64_t count, float *__restrict__ param, const float *__restrict__ grad,
                                                   float *__restrict__ m, float *__restrict__ v, float lr_over_bc1,
                                                   float inv_sqrt_bc2, float beta1, float beta2, float eps,
                                                   float *__restrict__ wt, int dim) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= count) return;
  const float g = grad[i];
  const float mi = m[i] + (g - m[i]) * (1.f - beta1);      // exp_avg.lerp_(grad, 1 - beta1)
  const float vi = v[i] * beta2 + (1.f - beta2) * g * g;  // exp_avg_sq.mul_(beta2).addcmul_(g, g, 1 - beta2)
  m[i] = mi;
  v[i] = vi;
  const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;     // (sqrt(v) / sqrt(bc2)) + eps
  const float pn = param[i] - lr_over_bc1 * (mi / denom);
  param[i] = pn;
  if (wt) {
    const int r = (int)(i / dim), cidx = (int)(i % dim);
    wt[(size_t)cidx * dim + r] = pn;
  }
}

struct Adam4 {
  AdamTensor t[4];
  int64_t start[5];
  float *wt[4];  // optional transposed copies of the square tensors (the backward GEMM wants [in][out])
  int dim;
  int32_t *pos_clear;  // optional: reset the batch-position map entries pos_clear[ids[0..b)] for the next step
  const int32_t *ids;
  int b;
};

// all four parameter tensors in one launch (the reference's optimizer.step() is one foreach call)
__global__ __launch_bounds__(256) void adam4_kernel(Adam4 a, float lr_over_bc1, float inv_sqrt_bc2, float beta1, float beta2,
                                                    float eps) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (a.pos_clear && i < a.b && a.ids[i] >= 0) a.pos_clear[a.ids[i]] = -1;
  if (i >= a.start[4]) return;
  const int k = i < a.start[1] ? 0 : i < a.start[2] ? 1 : i < a.start[3] ? 2 : 3;
  const int64_t j = i - a.start[k];
  const float g = a.t[k].grad[j];
  const float mi = a.t[k].m[j] + (g - a.t[k].m[j]) * (1.f - beta1);
  const float vi = a.t[k].v[j] * beta2 + (1.f - beta2) * g * g;
  a.t[k].m[j] = mi;
  a.t[k].v[j] = vi;
  const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
  const float pn = a.t[k].param[j] - lr_over_bc1 * (mi / denom);
  a.t[k].param[j] = pn;
  if (a.wt[k]) {
    const int r = (int)(j / a.dim), cidx = (int)(j % a.dim);
    a.wt[k][(size_t)cidx * a.dim + r] = pn;
  }
}

__global__ __launch_bounds__(256) void transpose2_kernel(int dim, const float *__restrict__ a, const float *__restrict__ b,
                                                         float *__restrict__ at, float *__restrict__ bt) {
  __shared__ float tile[2][32][33];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int k = ty; k < 32; k += 8) {
    const int r = by + k, cidx = bx + tx;
    if (r < dim && cidx < dim) {
      tile[0][k][tx] = a[(size_t)r * dim + cidx];
      tile[1][k][tx] = b[(size_t)r * dim + cidx];
    }
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {
    const int r = bx + k, cidx = by + tx;
    if (r < dim && cidx < dim) {
      at[(size_t)r * dim + cidx] = tile[0][tx][k];
      bt[(size_t)r * dim + cidx] = tile[1][tx][k];
    }
  }
}

static void row_geometry(int d, int &d4, int &lpr_log2, int &vpl) {
  d4 = d / 4;
  lpr_log2 = 2;
  while ((1 << lpr_log2) < d4 && lpr_log2 < 6) ++lpr_log2;
  vpl = d4 <= 64 ? 1 : d4 <= 128 ? 2 : 4;
}

int rownorm_fwd(int32_t n, int32_t d, const float *x, float *e, float *inv_den, void *stream, const int32_t *rows) {
  if (int rc = check_d(d)) return rc;
  GSS_REQUIRE(n >= 0 && x && e && inv_den, "rownorm_fwd: null operand");
  if (n == 0) return GSS_OK;
  int d4, lg, vpl;
  row_geometry(d, d4, lg, vpl);
  const int rows_per_block = 4 * (64 >> lg);
  dim3 grid(ceil_div(n, rows_per_block)), block(256);
  hipStream_t st = as_stream(stream);
  if (vpl == 1)
    hipLaunchKernelGGL((rownorm_fwd_kernel<1>), grid, block, 0, st, n, d4, lg, x, e, inv_den, rows);
  else if (vpl == 2)
    hipLaunchKernelGGL((rownorm_fwd_kernel<2>), grid, block, 0, st, n, d4, lg, x, e, inv_den, rows);
  else
    hipLaunchKernelGGL((rownorm_fwd_kernel<4>), grid, block, 0, st, n, d4, lg, x, e, inv_den, rows);
  GSS_LAUNCH_CHECK("rownorm_fwd_kernel");
  return GSS_OK;
}

int rownorm_elu_bwd(int32_t d, const float *de_b, const int32_t *idx, int32_t b, const float *e, const float *inv_den,
                    const float *p, float c, float *dx_b, float *dp_b, int32_t *pos_set, void *stream) {
  if (int rc = check_d(d)) return rc;
  GSS_REQUIRE(b >= 0 && de_b && idx && e && inv_den && p && dx_b && dp_b, "rownorm_elu_bwd: null operand");
  if (b == 0) return GSS_OK;
  int d4, lg, vpl;
  row_geometry(d, d4, lg, vpl);
  const int rows_per_block = 4 * (64 >> lg);
  dim3 grid(ceil_div(b, rows_per_block)), block(256);
  hipStream_t st = as_stream(stream);
  if (vpl == 1)
    hipLaunchKernelGGL((rownorm_elu_bwd_kernel<1>), grid, block, 0, st, b, d4, lg, de_b, idx, e, inv_den, p, c, dx_b, dp_b, pos_set);
  else if (vpl == 2)
    hipLaunchKernelGGL((rownorm_elu_bwd_kernel<2>), grid, block, 0, st, b, d4, lg, de_b, idx, e, inv_den, p, c, dx_b, dp_b, pos_set);
  else
    hipLaunchKernelGGL((rownorm_elu_bwd_kernel<4>), grid, block, 0, st, b, d4, lg, de_b, idx, e, inv_den, p, c, dx_b, dp_b, pos_set);
  GSS_LAUNCH_CHECK("rownorm_elu_bwd_kernel");
  return GSS_OK;
}

int scatter_add_rows(int32_t d, const float *src, const int32_t *rows, const float *keep, int32_t b, float *dst, int32_t *pos_clear,
                     const int32_t *pos_ids, void *stream) {
  if (int rc = check_d(d)) return rc;
  GSS_REQUIRE(b >= 0 && src && rows && dst && (!pos_clear || pos_ids), "scatter_add_rows: null operand");
  if (b == 0) return GSS_OK;
  hipLaunchKernelGGL(scatter_add_rows_kernel, dim3(ceil_div((int64_t)b * d / 4, 256)), dim3(256), 0, as_stream(stream), b,
                     d / 4, src, rows, keep, dst, pos_clear, pos_ids);
  GSS_LAUNCH_CHECK("scatter_add_rows_kernel");
  return GSS_OK;
}

int adam_step(int64_t count, float *param, const float *grad, float *m, float *v, int32_t step, float lr, float beta1,
              float beta2, float eps, float *wt, int32_t dim, void *stream) {
  GSS_REQUIRE(count >= 0 && param && grad && m && v && step >= 1, "adam_step: bad argument (step is 1-based)");
  GSS_REQUIRE(!wt || (int64_t)dim * dim == count, "adam_step: transposed copy needs count == dim*dim");
  if (count == 0) return GSS_OK;
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  hipLaunchKernelGGL(adam_kernel, dim3(ceil_div(count, 256)), dim3(256), 0, as_stream(stream), count, param, grad, m, v,
                     (float)((double)lr / bc1), (float)(1.0 / sqrt(bc2)), beta1, beta2, eps, wt, dim > 0 ? dim : 1);
  GSS_LAUNCH_CHECK("adam_kernel");
  return GSS_OK;
}

int batch_bits(const int32_t *ids, int32_t b, uint32_t *bits, int set, void *stream) {
  GSS_REQUIRE(ids && bits && b >= 0, "batch_bits: null operand");
  if (b == 0) return GSS_OK;
  hipLaunchKernelGGL(batch_bits_kernel, dim3(ceil_div(b, 256)), dim3(256), 0, as_stream(stream), ids, b, bits, set);
  GSS_LAUNCH_CHECK("batch_bits_kernel");
  return GSS_OK;
}

int batch_prepare(const int32_t *idx, int32_t b, const int32_t *node_map, int32_t lo, int32_t nl, const int32_t *gid2op, int32_t *rloc,
                  int32_t *pid, float *keep, int32_t *pos, void *stream, int32_t *rlist) {
  GSS_REQUIRE(idx && pos && b >= 0 && nl >= 0, "batch_prepare: null operand");
  if (b == 0) return GSS_OK;
  hipLaunchKernelGGL(batch_prepare_kernel, dim3(ceil_div(b, 256)), dim3(256), 0, as_stream(stream), idx, b, node_map, lo, nl, gid2op, rloc, pid,
                     keep, pos, rlist);
  GSS_LAUNCH_CHECK("batch_prepare_kernel");
  return GSS_OK;
}

int bits_fill(uint32_t *bits, int64_t first, int64_t last, void *stream) {
  GSS_REQUIRE(bits && first >= 0, "bits_fill: bad argument");
  if (last <= first) return GSS_OK;
  const int64_t words = (last - 1) / 32 - first / 32 + 1;
  hipLaunchKernelGGL(bits_fill_kernel, dim3(ceil_div(words, 256)), dim3(256), 0, as_stream(stream), bits, (long long)first, (long long)last);
  GSS_LAUNCH_CHECK("bits_fill_kernel");
  return GSS_OK;
}

int pack_rows(int32_t d, const float *src, const int32_t *rows, int64_t n, float *out, void *stream) {
  if (int rc = check_d(d)) return rc;
  GSS_REQUIRE(n >= 0 && (n == 0 || (src && rows && out)), "pack_rows: null operand");
  if (n == 0) return GSS_OK;
  hipLaunchKernelGGL(pack_rows_kernel, dim3(ceil_div(n * (d / 4), 256)), dim3(256), 0, as_stream(stream), n, d / 4, src, rows, out);
  GSS_LAUNCH_CHECK("pack_rows_kernel");
  return GSS_OK;
}

int unpack_rows(int32_t d, const float *src, const int32_t *rows, int64_t n, float *dst, void *stream) {
  if (int rc = check_d(d)) return rc;
  GSS_REQUIRE(n >= 0 && (n == 0 || (src && rows && dst)), "unpack_rows: null operand");
  if (n == 0) return GSS_OK;
  hipLaunchKernelGGL(unpack_rows_kernel, dim3(ceil_div(n * (d / 4), 256)), dim3(256), 0, as_stream(stream), n, d / 4, src, rows, dst);
  GSS_LAUNCH_CHECK("unpack_rows_kernel");
  return GSS_OK;
}

int halo_need_mark(const gss_csr *a, const int32_t *rows, int32_t b, int32_t n, int P, const int64_t *d_recv_off, const int64_t *d_wrecv_off,
                   uint32_t *needw, void *stream) {
  GSS_REQUIRE(a && rows && d_recv_off && d_wrecv_off && needw && b >= 0 && P >= 1, "halo_need_mark: bad argument");
  if (b == 0 || a->n_rows == 0) return GSS_OK;
  hipLaunchKernelGGL(halo_need_mark_kernel, dim3(ceil_div(b, 4)), dim3(256), 0, as_stream(stream), a->rowptr, a->col, rows, b, n, P, d_recv_off,
                     d_wrecv_off, needw);
  GSS_LAUNCH_CHECK("halo_need_mark_kernel");
  return GSS_OK;
}

int send_slot_bits(const uint32_t *bits, const int32_t *send_rows, int P, const int64_t *d_send_off, const int64_t *d_wsend_off, int64_t n_words,
                   uint32_t *out, void *stream) {
  GSS_REQUIRE(bits && d_send_off && d_wsend_off && out && n_words >= 0 && (n_words == 0 || send_rows), "send_slot_bits: bad argument");
  if (n_words == 0) return GSS_OK;
  hipLaunchKernelGGL(send_slot_bits_kernel, dim3(ceil_div(n_words, 256)), dim3(256), 0, as_stream(stream), bits, send_rows, P, d_send_off, d_wsend_off,
                     n_words, out);
  GSS_LAUNCH_CHECK("send_slot_bits_kernel");
  return GSS_OK;
}

int bits_clear(uint32_t *bits, int64_t first, int64_t last, void *stream) {
  GSS_REQUIRE(bits && first >= 0, "bits_clear: bad argument");
  if (last <= first) return GSS_OK;
  const int64_t words = (last - 1) / 32 - first / 32 + 1;
  hipLaunchKernelGGL(bits_clear_kernel, dim3(ceil_div(words, 256)), dim3(256), 0, as_stream(stream), bits, (long long)first, (long long)last);
  GSS_LAUNCH_CHECK("bits_clear_kernel");
  return GSS_OK;
}

int bits_set_list(uint32_t *bits, const int32_t *list, int64_t n, void *stream) {
  GSS_REQUIRE(bits && n >= 0 && (n == 0 || list), "bits_set_list: bad argument");
  if (n == 0) return GSS_OK;
  hipLaunchKernelGGL(bits_set_list_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, as_stream(stream), bits, list, n);
  GSS_LAUNCH_CHECK("bits_set_list_kernel");
  return GSS_OK;
}

size_t bits_compact_scratch_bytes(int P, const int64_t *h_woff) {
  int64_t nblk = 0;
  for (int q = 0; q < P; ++q) nblk += (h_woff[q + 1] - h_woff[q] + kCompactWords - 1) / kCompactWords;
  return ((size_t)nblk * sizeof(int32_t) + 15) / 16 * 16 + sizeof(int64_t) * ((size_t)nblk + 1);
}

int bits_compact(const uint32_t *words, int P, const int64_t *d_woff, const int64_t *h_woff, const int64_t *d_slot_off, const int32_t *map, int32_t add,
                 int32_t *out, int64_t *d_out_off, void *stream, void *scratch, size_t scratch_bytes) {
  GSS_REQUIRE(words && d_woff && h_woff && d_slot_off && out && d_out_off && P >= 1, "bits_compact: bad argument");
  hipStream_t st = as_stream(stream);
  int64_t nblk = 0;
  for (int q = 0; q < P; ++q) nblk += (h_woff[q + 1] - h_woff[q] + kCompactWords - 1) / kCompactWords;
  GSS_REQUIRE(nblk < (int64_t)INT32_MAX, "bits_compact: %lld blocks", (long long)nblk);
  // counts (int32) and bases (int64, one more than blocks) of the blocks: the caller's scratch (a plan carves it from its slab: nothing is
  // allocated inside a step), or stream-ordered scratch for a caller without one
  const size_t tot_bytes = ((size_t)nblk * sizeof(int32_t) + 15) / 16 * 16;
  const size_t need = tot_bytes + sizeof(int64_t) * ((size_t)nblk + 1);
  char *tmp = static_cast<char *>(scratch);
  const bool own = !tmp || scratch_bytes < need;
  if (own) GSS_HIP(hipMallocAsync((void **)&tmp, need, st));
  int32_t *block_tot = reinterpret_cast<int32_t *>(tmp);
  int64_t *base = reinterpret_cast<int64_t *>(tmp + tot_bytes);
  hipError_t e = hipSuccess;
  const char *what = "";
  if (nblk > 0) {
    hipLaunchKernelGGL(bits_count_kernel, dim3((unsigned)nblk), dim3(1024), 0, st, words, P, d_woff, block_tot);
    e = hipGetLastError();
    what = "bits_count_kernel";
  }
  if (e == hipSuccess) {
    hipLaunchKernelGGL(bits_scan_kernel, dim3(1), dim3(1024), 0, st, P, d_woff, (int)nblk, block_tot, base, d_out_off);
    e = hipGetLastError();
    what = "bits_scan_kernel";
  }
  if (e == hipSuccess && nblk > 0) {
    hipLaunchKernelGGL(bits_write_kernel, dim3((unsigned)nblk), dim3(1024), 0, st, words, P, d_woff, d_slot_off, map, add, base, out);
    e = hipGetLastError();
    what = "bits_write_kernel";
  }
  if (own) (void)hipFreeAsync(tmp, st);     // on the error paths too
  if (e != hipSuccess) return fail(GSS_EHIP, "launch %s -> %s", what, hipGetErrorString(e));
  return GSS_OK;
}

int adam_step4(const AdamTensor (&t)[4], int32_t step, float lr, float beta1, float beta2, float eps, float *w1t, float *w2t,
               int32_t dim, void *stream, int32_t *pos_clear, const int32_t *ids, int32_t b) {
  GSS_REQUIRE(step >= 1, "adam: step is 1-based");
  Adam4 a;
  a.pos_clear = pos_clear;
  a.ids = ids;
  a.b = pos_clear ? b : 0;
  a.wt[0] = w1t;
  a.wt[1] = nullptr;
  a.wt[2] = w2t;
  a.wt[3] = nullptr;
  a.dim = dim > 0 ? dim : 1;
  a.start[0] = 0;
  for (int k = 0; k < 4; ++k) {
    GSS_REQUIRE(t[k].param && t[k].grad && t[k].m && t[k].v && t[k].count >= 0, "adam: null tensor %d", k);
    a.t[k] = t[k];
    a.start[k + 1] = a.start[k] + t[k].count;
  }
  if (a.start[4] == 0) return GSS_OK;
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  hipLaunchKernelGGL(adam4_kernel, dim3(ceil_div(a.start[4] > a.b ? a.start[4] : (int64_t)a.b, 256)), dim3(256), 0, as_stream(stream), a, (float)((double)lr / bc1),
                     (float)(1.0 / sqrt(bc2)), beta1, beta2, eps);
  GSS_LAUNCH_CHECK("adam4_kernel");
  return GSS_OK;
}

int transpose2(int32_t dim, const float *a, const float *b, float *at, float *bt, void *stream) {
  const int nb = ceil_div(dim, 32);
  hipLaunchKernelGGL(transpose2_kernel, dim3(nb, nb), dim3(256), 0, as_stream(stream), dim, a, b, at, bt);
  GSS_LAUNCH_CHECK("transpose2_kernel");
  return GSS_OK;
}

}  // namespace gss

using namespace gss;
extern "C" {
int gss_rownorm_fwd(int32_t n, int32_t d, const float *x, float *e, float *inv_den, void *stream) {
  return rownorm_fwd(n, d, x, e, inv_den, stream);
}
int gss_rownorm_elu_bwd(int32_t d, const float *de_b, const int32_t *idx, int32_t b, const float *e, const float *inv_den,
                        const float *p, float c, float *dx_b, float *dp_b, void *stream) {
  return rownorm_elu_bwd(d, de_b, idx, b, e, inv_den, p, c, dx_b, dp_b, nullptr, stream);
}
int gss_scatter_add_rows(int32_t d, const float *src, const int32_t *rows, int32_t b, float *dst, void *stream) {
  return scatter_add_rows(d, src, rows, nullptr, b, dst, nullptr, nullptr, stream);
}
int gss_adam_step(int64_t count, float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int32_t step, float lr,
                  float beta1, float beta2, float eps, float *wt, int32_t dim, void *stream) {
  return adam_step(count, param, grad, exp_avg, exp_avg_sq, step, lr, beta1, beta2, eps, wt, dim, stream);
}
}
