// api.hip -- ABI version, error text and the tuning / A-B knobs.
#include "common.h"

namespace gss {
thread_local char g_err[512] = "";
Knobs g_knobs;
thread_local const Knobs *t_knobs = nullptr;
}  // namespace gss

namespace gss {
extern unsigned long long *g_gemm_stamps;   // dense.hip
}
using namespace gss;

extern "C" {
// diagnostic: a device buffer of 6 x 8 bytes per wave of the next projection launches (NULL switches it off); tools/gemm_stamps.py
int gss_debug_set_stamp_buffer(void *device_buffer) {
  g_gemm_stamps = static_cast<unsigned long long *>(device_buffer);
  return GSS_OK;
}
int gss_abi_version(void) { return GSS_ABI_VERSION; }
const char *gss_last_error(void) { return gss::g_err; }
}   // extern "C"

namespace gss {
// Every knob selects between implementations that all produce correct results (the tests run them all).  The values are process-wide
// DEFAULTS: a plan snapshots them at creation (common.h Knobs / KnobScope) and keeps running under its snapshot, so a knob changed here
// reaches per-op calls and plans created afterwards, never a live plan -- not this thread's, not another rank thread's.
int set_knob(Knobs &k, const char *name, int value) {
  GSS_REQUIRE(name, "debug_set_option: null name");
  if (strcmp(name, "spmm_variant") == 0) {
    GSS_REQUIRE(value == 1 || value == 2, "spmm_variant must be 1 (row per wave) or 2 (nnz-balanced segments)");
    k.spmm_variant = value;
    return GSS_OK;
  }
  if (strcmp(name, "spmm_pair") == 0) {
    k.spmm_pair = value ? 1 : 0;
    return GSS_OK;
  }
  if (strcmp(name, "spmm_pin") == 0) {
    GSS_REQUIRE(value == 0 || value == 1, "spmm_pin must be 0 or 1");
    k.spmm_pin = value;
    return GSS_OK;
  }
  if (strcmp(name, "spmm_fly") == 0) {
    GSS_REQUIRE(value == 4 || value == 8, "spmm_fly must be 4 or 8");
    k.spmm_fly = value;
    return GSS_OK;
  }
  if (strcmp(name, "spmm_hot_rows") == 0) {
    GSS_REQUIRE(value >= -1, "spmm_hot_rows must be >= -1");
    k.spmm_hot = value;
    return GSS_OK;
  }
  if (strcmp(name, "spmm_slices") == 0) {
    GSS_REQUIRE(value >= 0 && value <= 8, "spmm_slices must be in [0, 8] (0 = automatic)");
    k.spmm_slices = value;
    return GSS_OK;
  }
  if (strcmp(name, "spmm_seg_edges") == 0) {
    GSS_REQUIRE(value >= 4 && value <= 1024, "spmm_seg_edges must be in [4, 1024]");
    k.seg_edges = value;
    return GSS_OK;
  }
  if (strcmp(name, "spmm_giant") == 0) {
    GSS_REQUIRE(value == 0 || (value >= 64 && value % 4 == 0), "spmm_giant must be 0 (off) or a multiple of 4 >= 64 (stored entries)");
    k.spmm_giant = value;
    return GSS_OK;
  }
  if (strcmp(name, "loss_wgs") == 0) {
    GSS_REQUIRE(value >= 64 && value <= 4096, "loss_wgs must be in [64, 4096]");
    k.loss_wgs = value;
    return GSS_OK;
  }
  if (strcmp(name, "sparse_bits_rows") == 0) {
    GSS_REQUIRE(value >= 1, "sparse_bits_rows must be >= 1");
    k.sparse_bits_rows = value;
    return GSS_OK;
  }
  if (strcmp(name, "gemm_nt_cap") == 0) {
    GSS_REQUIRE(value == 0 || value == 1 || value == 2 || value == 4 || value == 8, "gemm_nt_cap must be 0, 1, 2, 4 or 8");
    k.gemm_nt_cap = value;
    return GSS_OK;
  }
  if (strcmp(name, "wgrad_wgs") == 0) {
    GSS_REQUIRE(value >= 8 && value <= 4096, "wgrad_wgs must be in [8, 4096]");
    k.wgrad_wgs = value;   // before any plan is created: the plan sizes its partial buffer with it
    return GSS_OK;
  }
  if (strcmp(name, "xcd_remap") == 0) {
    k.xcd_remap = value ? 1 : 0;
    return GSS_OK;
  }
  if (strcmp(name, "gemm_small_nt") == 0) {
    GSS_REQUIRE(value == 0 || value == 1 || value == 2 || value == 4 || value == 8, "gemm_small_nt must be 0, 1, 2, 4 or 8");
    k.gemm_small_nt = value;
    return GSS_OK;
  }
  if (strcmp(name, "wgrad_variant") == 0) {
    GSS_REQUIRE(value == 1 || value == 2, "wgrad_variant must be 1 (direct loads) or 2 (LDS-DMA ring)");
    k.wgrad_variant = value;
    return GSS_OK;
  }
  if (strcmp(name, "wgrad_deep") == 0) {
    k.wgrad_deep = value < 0 ? 0 : (value > 2 ? 2 : value);
    return GSS_OK;
  }
  if (strcmp(name, "gemm_lines") == 0) {
    k.gemm_lines = value ? 1 : 0;
    return GSS_OK;
  }
  if (strcmp(name, "gemm_hoist") == 0) {
    k.gemm_hoist = value ? 1 : 0;
    return GSS_OK;
  }
  if (strcmp(name, "gemm_rows_split") == 0) {
    k.gemm_rows_split = value ? 1 : 0;
    return GSS_OK;
  }
  if (strcmp(name, "lazy_halo") == 0) {
    GSS_REQUIRE(value >= -1 && value <= 1, "lazy_halo must be -1 (by graph size), 0 or 1");
    k.lazy_halo = value;   // sharded plans created afterwards; every rank of a job must use the same value
    return GSS_OK;
  }
  if (strcmp(name, "lazy_halo_u") == 0) {
    GSS_REQUIRE(value >= -1 && value <= 1, "lazy_halo_u must be -1 (automatic), 0 or 1");
    k.lazy_halo_u = value;   // sharded plans created afterwards; every rank of a job must use the same value
    return GSS_OK;
  }
  if (strcmp(name, "ppr_fused") == 0) {
    k.ppr_fused = value ? 1 : 0;   // handles created afterwards
    return GSS_OK;
  }
  if (strcmp(name, "loss_dgrad") == 0) {
    GSS_REQUIRE(value >= -1 && value <= 1, "loss_dgrad must be -1 (shards only), 0 or 1");
    k.loss_dgrad = value;
    return GSS_OK;
  }
  if (strcmp(name, "prep_side") == 0) {
    k.prep_side = value ? 1 : 0;
    return GSS_OK;
  }
  if (strcmp(name, "loss_slab") == 0) {
    GSS_REQUIRE(value >= -1 && value <= 1, "loss_slab must be -1 (by batch size), 0 or 1");
    k.loss_slab = value;   // read at every step of a sharded plan created afterwards; every rank of a job must use the same value
    return GSS_OK;
  }
  if (strcmp(name, "halo_recompute") == 0) {
    GSS_REQUIRE(value >= -1 && value <= 1, "halo_recompute must be -1 (automatic: on), 0 or 1");
    k.halo_recompute = value;   // sharded plans created afterwards; every rank of a job must use the same value
    return GSS_OK;
  }
  if (strcmp(name, "gemm_variant") == 0) {
    GSS_REQUIRE(value == 2 || value == 3 || value == 5, "gemm_variant must be 2 (automatic), 3 (128-node tiles, four waves) or 5 (128-node tiles, eight waves)");
    k.gemm_variant = value;
    return GSS_OK;
  }
  if (strcmp(name, "gemm_ws") == 0) {
    GSS_REQUIRE(value >= -1 && value <= 1, "gemm_ws must be -1 (by row count), 0 or 1");
    k.gemm_ws = value;
    return GSS_OK;
  }
  if (strcmp(name, "gemm_ws_wgs") == 0) {
    GSS_REQUIRE(value >= 1 && value <= 4096, "gemm_ws_wgs must be in [1, 4096]");
    k.gemm_ws_wgs = value;
    return GSS_OK;
  }
  if (strcmp(name, "gemm_ws_mode") == 0) {
    GSS_REQUIRE(value >= 0 && value <= 3, "gemm_ws_mode must be in [0, 3]");
    k.gemm_ws_mode = value;
    return GSS_OK;
  }
  if (strcmp(name, "gemm_ws_stagger") == 0) {
    GSS_REQUIRE(value >= 0 && value <= 64, "gemm_ws_stagger must be in [0, 64] (units of 512 cycles)");
    k.gemm_ws_stagger = value;
    return GSS_OK;
  }
  return fail(GSS_EINVAL, "unknown option %s", name);
}
}  // namespace gss

extern "C" {
// process-wide defaults: per-op entry points and plans created afterwards (a live plan keeps its snapshot)
int gss_debug_set_option(const char *name, int value) { return gss::set_knob(gss::g_knobs, name, value); }
}
