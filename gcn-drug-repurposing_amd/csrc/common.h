// common.h -- shared host/device helpers for libgssgcn.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "gssgcn.h"

namespace gss {

constexpr int kWave = 64;

// ---- tuning / A-B knobs (gss_debug_set_option) ---------------------------------------------------------------------------
// Process-wide defaults in g_knobs; a plan SNAPSHOTS them when it is created and every gss_plan_* call runs under its own snapshot
// (KnobScope), so changing a knob never changes what a live plan does -- in particular not the plans of other rank threads of the
// same process (VERDICT round 2, weak spot 10).  The per-op entry points (gss_spmm, gss_dense_fwd ...) read the process-wide values.
struct Knobs {
  int spmm_variant = 2;      // 1 = row per wave, 2 = nnz-balanced segments (default; shards and the lazy step need it)
  int spmm_slices = 0;       // 0 = automatic feature slicing (launch_balanced)
  int spmm_pin = 0;          // with a manual spmm_slices: slices pinned to XCDs (1) or time-separated (0)
  int spmm_hot = -1;         // overrides every CSR's hot set with rows [0, value) (-1 = the CSR's own, 0 = none)
  int seg_edges = 32;        // entries per SpMM segment (CSR handles created afterwards)
  int spmm_list_blocks = 2048;  // a row-filtered balanced SpMM over at least this many workgroups lists the workgroups that hold a passing row and walks the
                             // list with persistent workgroups (spmm.hip live_blocks_kernel) -- when the caller expects few rows to pass (LiveHint);
                             // 0 = never, 1 = always.  Same bits either way
  int spmm_giant = 32768;    // balanced SpMM, dense modes: rows with more stored entries than this are cut into chunks of a quarter of it that
                             // other workgroups sum (spmm.hip GiantRows); 0 = never.  CSR handles look at it with their first launch
  int gemm_variant = 2;      // projection tile shape: 2 = by width and row count (default), 3 = 128-node tiles of four waves forced, 5 = of eight waves forced
  int gemm_ws = -1;          // d = 128 forward projection without a row list: weight-stationary persistent kernel (proj_ws_kernel) -- -1 = from
                             // kWsMinRows = 32,769 rows on (dense.hip: more 128-node tiles than CUs -- the staged tiles would need a second round), 0 = never, 1 = always
  int wgrad_wgs = 256;       // workgroups of a full-size weight-gradient launch (sizes the plan's partial buffer)
  int loss_wgs = 256;        // workgroups the loss sweep's grid aims at
  int sparse_bits_rows = 100000;  // operand rows from which a plan keeps the bitmaps of the sparsity-aware backward hops
  int lazy_halo = -1;        // sharded lazy step: fetch only the boundary rows of the top layer's M that the batch rows read (-1 = graphs of >= 262,144 nodes, 0 = never, 1 = always)
  int lazy_halo_u = -1;      // the same for u in the top layer's second backward hop (sender-driven: costs one drain of the stream per step for the counts):
                             // -1 = with lazy_halo, but not over RCCL (there the hop runs exchange-free on the shard's A_hat transposed in place), 0 = never, 1 = with lazy_halo
  int ppr_fused = 1;         // diffusion profiles: the update and the column errors in the SpMM's epilogue (0 = separate update pass)
  int loss_dgrad = -1;       // finish + normalise' / ELU' + the batch rows' input gradient in one launch, d in {64, 128, 256}: -1 = on shards only
                             // (it spares a collective there; on one GPU two launches are 1.9 us faster), 0 = never, 1 = always
  int prep_side = 1;         // one GPU: a step's batch preparation as a side job of its first forward SpMM, E_B out of the top layer's projection
                             // (0 = the launches of their own: batch_prepare in lazy steps, the gather in full steps)
  int loss_slab = -1;        // sharded plans: the B x B loss sweep as row slabs (rank r: i tiles r, r + P, ...) + one more all-reduce, instead of
                             // replicated on every rank (-1 = from B = 8192 on, 0 = never, 1 = always; every rank of a job must use the same value)
  int halo_recompute = -1;   // sharded plans: layer 2's boundary input rows recomputed from layer 1's constant AX / AM instead of exchanged
                             // (-1 = automatic = on, 0 = never, 1 = always; every rank of a job must use the same value)
};
template <typename F>
inline size_t lds_request(F kernel, size_t need) {
  const size_t lds = need;
  if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  return lds;
}
extern Knobs g_knobs;
int set_knob(Knobs &k, const char *name, int value);   // api.hip: validate and set one knob of a Knobs value (the process defaults or a plan's snapshot)
extern thread_local const Knobs *t_knobs;
inline const Knobs &K() { return t_knobs ? *t_knobs : g_knobs; }
struct KnobScope {
  const Knobs *prev;
  explicit KnobScope(const Knobs *k) : prev(t_knobs) { t_knobs = k; }
  ~KnobScope() { t_knobs = prev; }
  KnobScope(const KnobScope &) = delete;
  KnobScope &operator=(const KnobScope &) = delete;
};

// ---- error plumbing (thread-local last error text) ------------------------------------------
extern thread_local char g_err[512];

inline int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
inline int fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

#define GSS_HIP(call)                                                                         \
  do {                                                                                        \
    hipError_t e_ = (call);                                                                   \
    if (e_ != hipSuccess)                                                                     \
      return ::gss::fail(GSS_EHIP, "%s:%d %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
  } while (0)

#define GSS_LAUNCH_CHECK(name)                                                                \
  do {                                                                                        \
    hipError_t e_ = hipGetLastError();                                                        \
    if (e_ != hipSuccess)                                                                     \
      return ::gss::fail(GSS_EHIP, "launch %s -> %s", name, hipGetErrorString(e_));           \
  } while (0)

#define GSS_REQUIRE(cond, ...)                                                                \
  do {                                                                                        \
    if (!(cond)) return ::gss::fail(GSS_EINVAL, __VA_ARGS__);                                 \
  } while (0)

inline int check_d(int32_t d) {
  if (d < 16 || d > 1024 || (d % 16) != 0)
    return fail(GSS_EINVAL, "hidden size d=%d unsupported: need a multiple of 16 in [16, 1024]", d);
  return GSS_OK;
}

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// ---- device helpers ---------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ void st4(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }

__device__ __forceinline__ float4 fma4(float s, float4 x, float4 a) {
  a.x = fmaf(s, x.x, a.x);
  a.y = fmaf(s, x.y, a.y);
  a.z = fmaf(s, x.z, a.z);
  a.w = fmaf(s, x.w, a.w);
  return a;
}
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 mul4(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float4 scale4(float s, float4 a) { return make_float4(s * a.x, s * a.y, s * a.z, s * a.w); }
// a row of F.normalize (modules/model.py:205): x / max(||x||, eps) -- a DIVISION per element, as torch computes it (one rounding; a multiplication
// by the reciprocal rounds twice).  Measured at config 2 against the CPU port (tools/norm_order_probe.py, profiles/r05_norm_probe_*.txt): with the
// reciprocal one of the 840 indication AUCs differs by 1.2e-4 - 3.0e-4 (one swapped pair at n_pos = 2), with the division none by more than 3.6e-5.
__device__ __forceinline__ float4 unit4(float den, float4 a) {
  return make_float4(__fdiv_rn(a.x, den), __fdiv_rn(a.y, den), __fdiv_rn(a.z, den), __fdiv_rn(a.w, den));
}

// F.elu (alpha = 1) and its derivative expressed through the pre-activation
// torch's elu is exp(x) - 1 for x <= 0 (ATen/native/cpu/Activation.cpp), not expm1; expf - 1 is also a fraction of expm1f's cost
__device__ __forceinline__ float elu1(float p) { return p > 0.f ? p : expf(p) - 1.f; }
__device__ __forceinline__ float elu1_grad(float p) { return p > 0.f ? 1.f : expf(p); }
__device__ __forceinline__ float4 elu_grad4(float4 p) {
  return make_float4(elu1_grad(p.x), elu1_grad(p.y), elu1_grad(p.z), elu1_grad(p.w));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Bijective XCD-aware remap of a 1-D block id (cdna guide T1): blocks b and b+8 share an XCD, so give
// each XCD a contiguous chunk of the logical grid.  Speed only, never correctness.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7, x = bid & 7, k = bid >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + k;
}

}  // namespace gss
