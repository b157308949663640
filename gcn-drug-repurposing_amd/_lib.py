"""ctypes binding of libgssgcn.so (include/gssgcn.h).  No CPU fallback: if the library is missing or a
call fails this raises -- the product path never routes around the HIP kernels."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgssgcn.so")
CSRC = os.path.join(_HERE, "csrc")
ABI_VERSION = 6

_lib = None


class GssError(RuntimeError):
    pass


class PlanDesc(C.Structure):
    _fields_ = [("n", C.c_int32), ("d", C.c_int32), ("num_layers", C.c_int32), ("max_batch", C.c_int32),
                ("layer_decay", C.c_float), ("alpha", C.c_float), ("lr", C.c_float), ("beta1", C.c_float),
                ("beta2", C.c_float), ("eps", C.c_float), ("cache_layer1", C.c_int32), ("pipeline_layer1", C.c_int32),
                ("node_map", C.c_void_p)]


class PprDesc(C.Structure):
    _fields_ = ([("n", C.c_int32), ("k", C.c_int32), ("kpad", C.c_int32), ("nnz", C.c_int64)]
                + [(k, C.c_void_p) for k in ("h_rowptr", "t_rowptr", "t_col", "t_val", "start", "start_dangling")]
                + [("n_z", C.c_int32), ("z_rows", C.c_void_p), ("n_ovr", C.c_int64)]
                + [(k, C.c_void_p) for k in ("ovr_col", "ovr_row", "ovr_ratio", "zero_ptr", "zero_ovr")]
                + [("n_sel", C.c_int64)]
                + [(k, C.c_void_p) for k in ("sel_col", "sel_row", "sel_val", "keep_ptr", "keep_row", "keep_val", "ovr_ptr")])


class HaloDesc(C.Structure):
    _fields_ = [("h_recv_off", C.c_void_p), ("h_send_off", C.c_void_p), ("d_send_rows", C.c_void_p)]


class ShardDesc(C.Structure):
    _fields_ = [("world", C.c_int32), ("rank", C.c_int32), ("h_bounds", C.c_void_p), ("halo_a", HaloDesc), ("halo_at", HaloDesc),
                ("d_gid2op_t", C.c_void_p), ("a_own", C.c_void_p), ("a_halo", C.c_void_p), ("at_own", C.c_void_p), ("at_halo", C.c_void_p),
                ("a_loc_t", C.c_void_p)]


class PlanIO(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("x", "w1", "b1", "w2", "b2", "emb", "loss", "gw1", "gb1", "gw2", "gb2")]


_P, _I32, _I64, _F, _D, _SZ = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_double, C.c_size_t

# name -> (restype, argtypes); mirrors include/gssgcn.h one to one
SIGNATURES = {
    "gss_abi_version": (C.c_int, []),
    "gss_last_error": (C.c_char_p, []),
    "gss_source_hash": (C.c_char_p, [C.c_char_p]),
    "gss_normalize_adj": (C.c_int, [_I32, _P, _P, _P, _P, _P, _P]),
    "gss_rowsum_dinv": (C.c_int, [_I32, _P, _P, _P, _P, _P]),
    "gss_rowsum_check": (C.c_int, [_I32, _P, _P, _P, _P]),
    "gss_csr_giant_rows": (C.c_int, [_P, C.POINTER(_I32), C.POINTER(_I32)]),
    "gss_scale_adj_shard": (C.c_int, [_I32, _I32, _P, _P, _P, _P, _I32, _P, _P]),
    "gss_csr_create": (C.c_int, [C.POINTER(_P), _I32, _I32, _I64, _P, _P, _P, _P]),
    "gss_csr_destroy": (None, [_P]),
    "gss_csr_set_hot": (C.c_int, [_P, _I32, _I32, _I32]),
    "gss_spmm": (C.c_int, [_P, _I32, _P, _P, _P, _P, _P]),
    "gss_spmm_add": (C.c_int, [_P, _I32, _P, _P, _P, _P, _P, _P]),
    "gss_spmm_bwd1": (C.c_int, [_P, _I32, _P, _P, _P, _P, _P, _P, _P]),
    "gss_spmm_bwd2": (C.c_int, [_P, _I32, _P, _P, _P, _F, _P, _P, _P, _P]),
    "gss_spmm_bwd1_sparse": (C.c_int, [_P, _I32, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "gss_dense_fwd": (C.c_int, [_I32, _I32, _P, _P, _P, _P, _P, _P, _P, _F, _P, _P, _P]),
    "gss_dense_bwd_input": (C.c_int, [_I32, _I32, _P, _P, _P, _P, _P, _P, _P]),
    "gss_wgrad_workspace_bytes": (_SZ, [_I32, _I32]),
    "gss_dense_bwd_weight": (C.c_int, [_I32, _I32, _P, _P, _P, _P, _P, _P, _P, C.c_int, _P, _P]),
    "gss_rownorm_fwd": (C.c_int, [_I32, _I32, _P, _P, _P, _P]),
    "gss_loss_workspace_bytes": (_SZ, [_I32, _I32]),
    "gss_loss_workspace_bytes_max": (_SZ, [_I32, _I32]),
    "gss_loss_fwd_bwd": (C.c_int, [_I32, _I32, _P, _P, _I32, _F, _F, _P, _P, _P, _P]),
    "gss_rownorm_elu_bwd": (C.c_int, [_I32, _P, _P, _I32, _P, _P, _P, _F, _P, _P, _P]),
    "gss_scatter_add_rows": (C.c_int, [_I32, _P, _P, _I32, _P, _P]),
    "gss_comm_unique_id": (C.c_int, [_P]),
    "gss_comm_create_rccl": (C.c_int, [C.POINTER(_P), _I32, _I32, _P]),
    "gss_comm_create_local": (C.c_int, [C.POINTER(_P), _I32]),
    "gss_comm_create_host": (C.c_int, [C.POINTER(_P), _I32, _I32, _P, _P]),
    "gss_comm_destroy": (None, [_P]),
    "gss_comm_abort": (None, [_P]),
    "gss_comm_check": (C.c_int, [_P]),
    "gss_comm_count": (C.c_int, [_P, C.POINTER(_I32)]),
    "gss_comm_sync": (C.c_int, [_P, _P, _D]),
    "gss_comm_world": (_I32, [_P]),
    "gss_comm_rank": (_I32, [_P]),
    "gss_allgather_rows": (C.c_int, [_P, _I32, _I32, _P, _P, _P]),
    "gss_allgather_bytes": (C.c_int, [_P, _P, _P, _SZ, _P]),
    "gss_allreduce_sum": (C.c_int, [_P, _P, _I64, _P]),
    "gss_exchange_rows": (C.c_int, [_P, _I32, _P, _P, _P, _P, _P]),
    "gss_adam_step": (C.c_int, [_I64, _P, _P, _P, _P, _I32, _F, _F, _F, _F, _P, _I32, _P]),
    "gss_percentile": (C.c_int, [_I32, _I32, _P, _D, C.POINTER(_F), _P]),
    "gss_knn_topk": (C.c_int, [_I32, _I32, _P, _I32, _P, _P, _P]),
    "gss_knn_topk_rows": (C.c_int, [_I32, _I32, _P, _I32, _I32, _I32, _P, _P, _P]),
    "gss_write_embs_text": (C.c_int, [C.c_char_p, _P, _I64, _I32, _I32]),
    "gss_format_e18": (C.c_int, [_F, C.c_char_p]),
    "gss_embs_open": (C.c_int, [C.POINTER(_P), C.c_char_p, _I32]),
    "gss_embs_rows": (_I64, [_P]),
    "gss_embs_cols": (_I32, [_P]),
    "gss_embs_names_bytes": (_I64, [_P]),
    "gss_embs_copy": (C.c_int, [_P, _P, C.c_char_p, _I64, C.POINTER(_I64)]),
    "gss_embs_close": (None, [_P]),
    "gss_edgelist_open": (C.c_int, [C.POINTER(_P), C.c_char_p, C.c_char_p, _I64, _I64, _I32]),
    "gss_edgelist_edges": (_I64, [_P]),
    "gss_edgelist_bad_line": (_I64, [_P]),
    "gss_edgelist_copy": (C.c_int, [_P, _P, _P, _P]),
    "gss_edgelist_close": (None, [_P]),
    "gss_plan_create": (C.c_int, [C.POINTER(_P), C.POINTER(PlanDesc), _P, _P, C.POINTER(PlanIO)]),
    "gss_plan_create_sharded": (C.c_int, [C.POINTER(_P), C.POINTER(PlanDesc), C.POINTER(ShardDesc), _P, _P, _P, C.POINTER(PlanIO)]),
    "gss_plan_gather_embeddings": (C.c_int, [_P, _P, _P]),
    "gss_plan_destroy": (None, [_P]),
    "gss_plan_forward": (C.c_int, [_P, _P]),
    "gss_plan_loss_backward": (C.c_int, [_P, _P, _I32, _F, _P]),
    "gss_plan_backward": (C.c_int, [_P, _P, _I32, _P, _P]),
    "gss_plan_adam": (C.c_int, [_P, _P]),
    "gss_plan_step": (C.c_int, [_P, _P, _I32, _F, _P]),
    "gss_plan_step_lazy": (C.c_int, [_P, _P, _I32, _F, _P]),
    "gss_plan_activation": (_P, [_P, C.c_int, C.c_int]),
    "gss_plan_device_bytes": (_SZ, [_P]),
    "gss_plan_check_guards": (C.c_int, [_P]),
    "gss_plan_lazy_halo_rows": (C.c_int, [_P, C.POINTER(_I64)]),
    "gss_plan_comm_stats": (C.c_int, [_P, C.POINTER(_I64)]),
    "gss_plan_sync_stats": (C.c_int, [_P, C.POINTER(_I64)]),
    "gss_comm_local_mode": (C.c_int, [_P, _I32]),
    "gss_comm_local_log": (C.c_int, [_P, C.POINTER(_I64), _I32, C.POINTER(_I32)]),
    "gss_plan_set_step": (None, [_P, _I32]),
    "gss_plan_adam_buffer": (_P, [_P, _I32, _I32]),
    "gss_plan_get_step": (_I32, [_P]),
    "gss_plan_profile": (C.c_int, [_P, C.c_int]),
    "gss_plan_profile_read": (C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_int64), _P]),
    "gss_debug_set_option": (C.c_int, [C.c_char_p, C.c_int]),
    "gss_debug_set_stamp_buffer": (C.c_int, [_P]),
    "gss_plan_debug_set_option": (C.c_int, [_P, C.c_char_p, C.c_int]),
    "gss_memcpy_d2d": (C.c_int, [_P, _P, _SZ, _P]),
    "gss_ppr_create": (C.c_int, [C.POINTER(_P), C.POINTER(PprDesc)]),
    "gss_ppr_destroy": (None, [_P]),
    "gss_ppr_device_bytes": (_SZ, [_P]),
    "gss_ppr_check_guards": (C.c_int, [_P]),
    "gss_ppr_run": (C.c_int, [_P, _D, _D, _I32, _P, _P, _P]),
    "gss_ppr_spmm": (C.c_int, [_P, _P, _P, _P]),
}


PROF_CLASSES = ("spmm_fwd_hadamard", "spmm_fwd", "spmm_bwd1", "spmm_bwd2", "dense_fwd", "dgrad", "wgrad", "wgrad_batch",
                "loss", "rownorm", "elementwise", "adam", "comm", "comm_batch", "comm_grads", "spmm_bwd1_dense", "spmm_bwd2_dense")


def source_hashes():
    """sha256 of every source the Makefile hashes into the library (csrc/*.hip, csrc/*.h, include/gssgcn.h), by file name, + "*" = all of them
    concatenated in the Makefile's order -- what gss_source_hash must return for a library built from this tree"""
    import glob
    import hashlib
    files = sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h")))
    files.append(os.path.join(os.path.dirname(_HERE), "include", "gssgcn.h"))
    out, every = {}, hashlib.sha256()
    for f in files:
        data = open(f, "rb").read()
        out[os.path.basename(f)] = hashlib.sha256(data).hexdigest()
        every.update(data)
    out["*"] = every.hexdigest()
    return out


def library_is_current():
    """does the built library carry the hashes of the sources in this tree?  (dlopen only: no GPU needed)"""
    if not os.path.exists(LIB_PATH):
        return False
    import torch  # noqa: F401  (torch's HIP runtime has to be in the process before this library is: see load())
    try:
        lib = C.CDLL(LIB_PATH)
        fn = lib.gss_source_hash
    except (OSError, AttributeError):
        return False
    fn.restype, fn.argtypes = C.c_char_p, [C.c_char_p]
    for name, want in source_hashes().items():
        got = fn(name.encode())
        if got is None or got.decode() != want:
            return False
    return True


def build(verbose: bool = False) -> str:
    """Compile libgssgcn.so for gfx950 with hipcc (cross-compiles without a GPU).  An incremental `make` trusts timestamps, and the built
    library travels to the GPU box while being git-ignored: a stale .so with fresh timestamps would be reused silently.  So the library
    carries the sha256 of its sources (gss_source_hash) and this function checks them against the tree after the incremental build; a
    mismatch -- or GSS_REBUILD=1 -- forces `make -B`.  Says which it did."""
    jobs = str(min(8, os.cpu_count() or 1))

    def make(*extra):
        res = subprocess.run(["make", "-C", CSRC, "-j", jobs, *extra], capture_output=True, text=True)
        if verbose or res.returncode != 0:
            print(res.stdout[-4000:])
            print(res.stderr[-4000:])
        if res.returncode != 0:
            raise GssError("building libgssgcn.so failed (hipcc --offload-arch=gfx950); see output above")

    if os.environ.get("GSS_REBUILD") == "1":
        make("-B")
        how = "full rebuild (GSS_REBUILD=1)"
    else:
        make()
        how = "incremental make"
        if not library_is_current():
            make("-B")
            how = "full rebuild (the incrementally built library's source hashes did not match the tree)"
    if not library_is_current():
        raise GssError("libgssgcn.so does not carry the hashes of the sources in this tree even after a full rebuild")
    print(f"libgssgcn.so: {how}; source hashes match the tree")
    return LIB_PATH


def load():
    """dlopen the library and declare every prototype.  Raises GssError when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GssError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                       f"or `make -C {CSRC}` -- there is no CPU fallback for the GSS-GCN kernels")
    # torch ships its own HIP runtime (torch/lib/libamdhip64.so): it has to be in the process before this library is,
    # otherwise the library binds to /opt/rocm's copy and the two runtimes do not see each other's device state
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here means header and library disagree
        fn.restype = res
        fn.argtypes = args
    if lib.gss_abi_version() != ABI_VERSION:
        raise GssError(f"libgssgcn.so ABI {lib.gss_abi_version()} != binding ABI {ABI_VERSION}; rebuild")
    # GSS_OPTIONS="name=value,name=value": tuning knobs (gss_debug_set_option) for entry points without a flag for them -- every rank of
    # a torch.distributed job inherits the same environment, which is what the job-wide knobs (lazy_halo) need
    for kv in filter(None, (t.strip() for t in os.environ.get("GSS_OPTIONS", "").split(","))):
        name, _, val = kv.partition("=")
        try:
            rc = lib.gss_debug_set_option(name.strip().encode(), int(val))
        except ValueError:
            raise GssError(f"GSS_OPTIONS: {kv!r} is not name=integer") from None
        if rc != 0:
            raise GssError(f"GSS_OPTIONS: {kv!r}: {lib.gss_last_error().decode(errors='replace')}")
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().gss_last_error().decode(errors="replace")
        raise GssError(f"{what or 'libgssgcn'} failed with code {rc}: {msg}")


def ptr(t) -> int:
    """device pointer of a torch tensor (None -> NULL)"""
    return 0 if t is None else t.data_ptr()


def current_stream() -> int:
    import torch
    return torch.cuda.current_stream().cuda_stream
