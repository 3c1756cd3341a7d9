"""ctypes declarations for include/vb2_abi.h (the C-ABI of libvb2.so).

This is the same binding a maintainer of another host language would write (see
INTEGRATION.md); the Python layer adds nothing but marshalling.  If libvb2.so is
missing the import fails loudly -- there is no Python or CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VB2_LIB_PATH") or os.path.join(_PKG, "libvb2.so")      # (VB2_LIB_PATH: A/B builds)

VB2_MAX_PC = 64
VB2_OK = 0
VB2_ERR_INVALID, VB2_ERR_NO_DEVICE, VB2_ERR_HIP, VB2_ERR_IO, VB2_ERR_NOMEM, VB2_ERR_SANITY = \
    -1, -2, -3, -4, -5, -6


class Input(C.Structure):
    _fields_ = [
        ("num_marker", C.c_int32), ("num_pc", C.c_int32),
        ("ud", C.c_void_p), ("means", C.c_void_p), ("read_off", C.c_void_p),
        ("bases", C.c_void_p), ("quals", C.c_void_p), ("alt_base", C.c_void_p),
        ("known_af", C.c_void_p),
        ("avg_depth", C.c_double), ("sd_depth", C.c_double),
        ("sanity_disabled", C.c_int32), ("reserved", C.c_int32),
    ]


class Options(C.Structure):
    _fields_ = [("device", C.c_int32), ("flags", C.c_int32), ("stream", C.c_void_p)]


class Info(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("device", C.c_int32), ("num_marker", C.c_int32),
        ("num_pc", C.c_int32), ("num_active_marker", C.c_int64), ("num_read", C.c_int64),
        ("num_read_other", C.c_int64), ("num_code", C.c_int32), ("num_tile", C.c_int32),
        ("device_bytes", C.c_int64), ("algorithmic_bytes_per_eval", C.c_int64),
        ("device_name", C.c_char * 64), ("arch", C.c_char * 32), ("cohort_step_bytes", C.c_int64),
        ("layout", C.c_int32), ("num_table_row", C.c_int32), ("num_step", C.c_int64),
    ]


class Model(C.Structure):
    _fields_ = [
        ("is_heter", C.c_int32), ("is_pc_fixed", C.c_int32), ("is_alpha_fixed", C.c_int32),
        ("is_af_known", C.c_int32), ("fix_alpha", C.c_double), ("fix_pc", C.c_void_p),
        ("epsilon", C.c_double), ("verbose", C.c_int32), ("notices", C.c_int32),
    ]


class SearchOpts(C.Structure):
    _fields_ = [
        ("num_start", C.c_int32), ("seed", C.c_uint32), ("start_sd", C.c_double),
        ("line_search", C.c_int32), ("reserved", C.c_int32),
    ]


class Estimate(C.Structure):
    _fields_ = [
        ("alpha", C.c_double), ("llk1", C.c_double), ("llk0", C.c_double),
        ("pc", C.c_double * VB2_MAX_PC), ("pc2", C.c_double * VB2_MAX_PC),
        ("num_eval", C.c_int64), ("num_launch_point", C.c_int64),
        ("converged", C.c_int32), ("reserved", C.c_int32),
    ]


class Trace(C.Structure):
    _fields_ = [
        ("capacity", C.c_int64), ("count", C.c_int64),
        ("alpha", C.c_void_p), ("pc1", C.c_void_p), ("pc2", C.c_void_p), ("llk", C.c_void_p),
    ]


class MpileupOpts(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("given", "min_bq", "min_mq", "adjust_mq", "max_depth", "no_orphans",
                                          "incl_flags", "excl_flags")]


class RunArgs(C.Structure):
    _fields_ = [
        ("ud_path", C.c_char_p), ("mean_path", C.c_char_p), ("bed_path", C.c_char_p),
        ("pileup_path", C.c_char_p), ("known_af_path", C.c_char_p), ("output_prefix", C.c_char_p),
        ("num_pc", C.c_int32), ("disable_sanity", C.c_int32), ("output_pileup", C.c_int32),
        ("device", C.c_int32), ("model", Model),
        ("devices", C.POINTER(C.c_int32)), ("num_device", C.c_int32), ("reserved", C.c_int32),
        ("bam_path", C.c_char_p), ("reference_path", C.c_char_p), ("search", SearchOpts),
        ("mpileup", MpileupOpts),
    ]


class RunResult(C.Structure):
    _fields_ = [
        ("est", Estimate), ("num_marker", C.c_int32), ("num_site", C.c_int32),
        ("num_bases", C.c_int64), ("avg_depth", C.c_double), ("sd_depth", C.c_double),
        ("seconds_load", C.c_double), ("seconds_optimize", C.c_double),
    ]


class CohortArgs(C.Structure):
    _fields_ = [
        ("base", RunArgs), ("num_sample", C.c_int32), ("pileup_paths", C.POINTER(C.c_char_p)),
        ("output_prefixes", C.POINTER(C.c_char_p)), ("group_size", C.c_int32), ("num_host_thread", C.c_int32),
    ]


class ShardInfo(C.Structure):
    _fields_ = [
        ("num_shard", C.c_int32), ("nranks", C.c_int32), ("rank", C.c_int32), ("uses_rccl", C.c_int32),
        ("num_allreduce", C.c_int64), ("marker_lo", C.c_int32 * 64), ("marker_hi", C.c_int32 * 64),
        ("num_read", C.c_int64 * 64), ("partial_sums", C.c_int32), ("rccl_stub", C.c_int32),
    ]


EVAL_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double),
                      C.POINTER(C.c_double), C.POINTER(C.c_double))

# every symbol include/vb2_abi.h declares
SYMBOLS = [
    "vb2_ctx_create", "vb2_ctx_destroy", "vb2_ctx_info", "vb2_llk_eval_batch",
    "vb2_llk_eval_batch_device", "vb2_ctx_search_begin", "vb2_ctx_search_end", "vb2_optimize_llk", "vb2_ctx_optimize_llk", "vb2_ctx_optimize_llk_ex", "vb2_run", "vb2_cohort_run",
    "vb2_flat_load", "vb2_flat_input", "vb2_flat_stats", "vb2_flat_free", "vb2_last_error",
    "vb2_abi_version", "vb2_device_count",
    "vb2_batch_create", "vb2_batch_destroy", "vb2_batch_eval", "vb2_batch_optimize_llk",
    "vb2_shard_group_create", "vb2_rccl_unique_id", "vb2_shard_group_create_rank", "vb2_shard_group_eval",
    "vb2_shard_group_optimize_llk", "vb2_shard_group_info", "vb2_shard_group_destroy", "vb2_shard_range",
]

_lib = None


def lib():
    """Load libvb2.so (once).  Raises if it has not been built -- by design."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "verifybamid_amd: %s is missing; build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` or `make -C verifybamid_amd/csrc`. There is no CPU fallback." % LIB_PATH)
    # PyTorch-ROCm wheels bundle their own libamdhip64/libhsa-runtime64.  Two HIP runtimes
    # in one process cannot both own the device, so when torch is installed it must be
    # loaded FIRST: libvb2.so's NEEDED libamdhip64.so.7 then binds to the copy that is
    # already mapped, and torch streams/events/tensors interoperate with our launches.
    # (The C++ command line never loads torch and uses /opt/rocm's runtime.)
    if os.environ.get("VB2_NO_TORCH", "") != "1":
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    L = C.CDLL(LIB_PATH)
    L.vb2_last_error.restype = C.c_char_p
    L.vb2_abi_version.restype = C.c_int
    L.vb2_device_count.restype = C.c_int
    L.vb2_ctx_create.argtypes = [C.POINTER(Input), C.POINTER(Options), C.POINTER(C.c_void_p)]
    L.vb2_ctx_destroy.argtypes = [C.c_void_p]
    L.vb2_ctx_destroy.restype = None
    L.vb2_ctx_info.argtypes = [C.c_void_p, C.POINTER(Info)]
    L.vb2_llk_eval_batch.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p]
    L.vb2_llk_eval_batch_device.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                            C.c_void_p]
    L.vb2_cohort_run.argtypes = [C.POINTER(CohortArgs), C.POINTER(RunResult), C.POINTER(C.c_int32)]
    L.vb2_ctx_search_begin.argtypes = [C.c_void_p]
    L.vb2_ctx_search_end.argtypes = [C.c_void_p]
    L.vb2_ctx_search_end.restype = None
    L.vb2_optimize_llk.argtypes = [EVAL_FN, C.c_void_p, C.c_int32, C.POINTER(Model),
                                   C.POINTER(Estimate), C.POINTER(Trace)]
    L.vb2_ctx_optimize_llk_ex.argtypes = [C.c_void_p, C.POINTER(Model), C.POINTER(SearchOpts), C.POINTER(Estimate),
                                          C.c_void_p]
    L.vb2_ctx_optimize_llk.argtypes = [C.c_void_p, C.POINTER(Model), C.POINTER(Estimate),
                                       C.POINTER(Trace)]
    L.vb2_run.argtypes = [C.POINTER(RunArgs), C.POINTER(RunResult)]
    L.vb2_flat_load.argtypes = [C.POINTER(RunArgs), C.POINTER(C.c_void_p)]
    L.vb2_flat_input.argtypes = [C.c_void_p]
    L.vb2_flat_input.restype = C.POINTER(Input)
    L.vb2_flat_stats.argtypes = [C.c_void_p, C.POINTER(RunResult)]
    L.vb2_flat_free.argtypes = [C.c_void_p]
    L.vb2_flat_free.restype = None
    L.vb2_batch_create.argtypes = [C.POINTER(C.c_void_p), C.c_int32, C.POINTER(C.c_void_p)]
    L.vb2_batch_destroy.argtypes = [C.c_void_p]
    L.vb2_batch_destroy.restype = None
    L.vb2_batch_eval.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.vb2_batch_optimize_llk.argtypes = [C.c_void_p, C.POINTER(Model), C.c_int32, C.POINTER(Estimate)]
    L.vb2_shard_group_create.argtypes = [C.POINTER(Input), C.POINTER(C.c_int32), C.c_int32, C.POINTER(C.c_void_p)]
    L.vb2_rccl_unique_id.argtypes = [C.c_void_p]
    L.vb2_shard_group_create_rank.argtypes = [C.POINTER(Input), C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                              C.POINTER(C.c_void_p)]
    L.vb2_shard_group_eval.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.vb2_shard_group_optimize_llk.argtypes = [C.c_void_p, C.POINTER(Model), C.POINTER(Estimate), C.POINTER(Trace)]
    L.vb2_shard_group_info.argtypes = [C.c_void_p, C.POINTER(ShardInfo)]
    L.vb2_shard_range.argtypes = [C.POINTER(Input), C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.vb2_shard_group_destroy.argtypes = [C.c_void_p]
    L.vb2_shard_group_destroy.restype = None
    L.vb2_debug_set_tunable.argtypes = [C.c_char_p, C.c_int]
    L.vb2_debug_get_tunable.argtypes = [C.c_char_p, C.POINTER(C.c_int)]
    _lib = L
    return L


def set_tunable(name, value):
    """A run-time switch of the library by name (csrc/tunables.h): tests and measurement scripts only."""
    if lib().vb2_debug_set_tunable(name.encode(), int(value)) != VB2_OK:
        raise KeyError("libvb2 has no tunable %r" % name)


def get_tunable(name):
    v = C.c_int(0)
    if lib().vb2_debug_get_tunable(name.encode(), C.byref(v)) != VB2_OK:
        raise KeyError("libvb2 has no tunable %r" % name)
    return v.value


class Vb2Error(RuntimeError):
    def __init__(self, code, where):
        msg = lib().vb2_last_error()
        super().__init__("%s failed (%d): %s" % (where, code, msg.decode() if msg else ""))
        self.code = code


def check(code, where):
    if code != VB2_OK:
        raise Vb2Error(code, where)


def kernel_source_hash():
    """sha256 (16 hex digits) over the sources of the evaluation kernels: what ties a PMC-derived figure kept under profiles/
    (instruction counts, HBM-side bytes) to the code it was measured on -- bench.py drops figures whose hash is another."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
    for name in ("llk_kernels.hip", "resident_kernel.inc", "llk_kernels.h", "kernel_debug.h", "log_table.inc", "Makefile"):
        with open(os.path.join(d, name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]
