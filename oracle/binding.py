"""binding.py -- TEST INFRASTRUCTURE ONLY: ctypes access to oracle/liboracle.so
(the plain-C restatement) and, when present, oracle/_ref/libvb2ref.so (the
reference's own AmoebaMinimizer compiled in place).  Never imported by the
product package."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")
_REF = os.path.join(_HERE, "_ref", "libvb2ref.so")


class _Data(C.Structure):
    _fields_ = [
        ("num_marker", C.c_int32), ("num_pc", C.c_int32),
        ("ud", C.c_void_p), ("means", C.c_void_p),
        ("base_info_index", C.c_void_p), ("alt_base", C.c_void_p),
        ("known_af", C.c_void_p),
        ("num_site", C.c_int32), ("site_off", C.c_void_p),
        ("bases", C.c_void_p), ("quals", C.c_void_p),
        ("avg_depth", C.c_double), ("sd_depth", C.c_double),
        ("sanity_disabled", C.c_int32), ("af_known", C.c_int32),
    ]


class _Options(C.Structure):
    _fields_ = [
        ("is_heter", C.c_int32), ("is_pc_fixed", C.c_int32), ("is_alpha_fixed", C.c_int32),
        ("fix_alpha", C.c_double), ("fix_pc", C.c_void_p), ("epsilon", C.c_double),
        ("num_thread", C.c_int32), ("verbose", C.c_int32),
    ]


class _Trace(C.Structure):
    _fields_ = [
        ("capacity", C.c_int64), ("count", C.c_int64),
        ("alpha", C.c_void_p), ("pc1", C.c_void_p), ("pc2", C.c_void_p), ("llk", C.c_void_p),
    ]


class _Result(C.Structure):
    _fields_ = [
        ("alpha", C.c_double), ("llk1", C.c_double), ("llk0", C.c_double),
        ("num_eval", C.c_int64), ("converged", C.c_int32),
    ]


_OBJECTIVE = C.CFUNCTYPE(C.c_double, C.c_void_p, C.POINTER(C.c_double), C.c_int)


def build(force=False):
    """Compile the oracle (and oracle/_ref when /root/reference exists)."""
    if force or not os.path.exists(_LIB) or \
            os.path.getmtime(_LIB) < os.path.getmtime(os.path.join(_HERE, "vb2_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "all"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference") and (force or not os.path.exists(_REF)):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB)
        _lib.vb2o_compute_mix_llks.restype = C.c_double
        _lib.vb2o_compute_mix_llks.argtypes = [C.POINTER(_Data), C.c_void_p, C.c_void_p,
                                               C.c_double, C.c_int]
        _lib.vb2o_optimize_llk.restype = C.c_int
        _lib.vb2o_optimize_llk.argtypes = [C.POINTER(_Data), C.POINTER(_Options), C.c_void_p,
                                           C.POINTER(_Trace), C.c_void_p, C.c_void_p,
                                           C.POINTER(_Result)]
        _lib.vb2o_amoeba_minimize.restype = C.c_double
        _lib.vb2o_amoeba_minimize.argtypes = [_OBJECTIVE, C.c_void_p, C.c_int,
                                              C.POINTER(C.c_double), C.c_double]
    return _lib


def ref_lib():
    """The reference's AmoebaMinimizer, or None if oracle/_ref was never built."""
    global _ref
    if _ref is None:
        build()
        if not os.path.exists(_REF):
            return None
        _ref = C.CDLL(_REF)
        _ref.vb2ref_amoeba_minimize.restype = C.c_double
        _ref.vb2ref_amoeba_minimize.argtypes = [_OBJECTIVE, C.c_void_p, C.c_int,
                                                C.POINTER(C.c_double), C.c_double]
    return _ref


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class OracleData:
    """Keeps the numpy buffers alive next to the C struct that points into them."""

    def __init__(self, flat):
        self.flat = flat
        self._keep = dict(
            ud=np.ascontiguousarray(flat.ud, dtype=np.float64),
            means=np.ascontiguousarray(flat.means, dtype=np.float64),
            idx=np.ascontiguousarray(flat.base_info_index, dtype=np.int32),
            alt=np.ascontiguousarray(flat.alt_base, dtype=np.uint8),
            kaf=None if flat.known_af is None else np.ascontiguousarray(flat.known_af, dtype=np.float64),
            off=np.ascontiguousarray(flat.site_off, dtype=np.int64),
            bases=np.ascontiguousarray(flat.bases, dtype=np.uint8),
            quals=np.ascontiguousarray(flat.quals, dtype=np.uint8),
        )
        k = self._keep
        self.c = _Data(int(flat.num_marker), int(flat.num_pc), _ptr(k["ud"]), _ptr(k["means"]),
                       _ptr(k["idx"]), _ptr(k["alt"]), _ptr(k["kaf"]),
                       int(len(k["off"]) - 1), _ptr(k["off"]), _ptr(k["bases"]), _ptr(k["quals"]),
                       float(flat.avg_depth), float(flat.sd_depth),
                       int(bool(flat.sanity_disabled)), int(bool(flat.af_known)))

    def llk(self, pc1, pc2, alpha, num_thread=1):
        pc1 = np.ascontiguousarray(pc1, dtype=np.float64)
        pc2 = np.ascontiguousarray(pc2, dtype=np.float64)
        assert pc1.size == self.flat.num_pc and pc2.size == self.flat.num_pc
        return lib().vb2o_compute_mix_llks(C.byref(self.c), _ptr(pc1), _ptr(pc2),
                                           float(alpha), int(num_thread))

    def optimize(self, *, within_ancestry=False, fix_pc=None, fix_alpha=None, epsilon=1e-8,
                 num_thread=1, minimizer="oracle", trace_capacity=0, verbose=False):
        """Full OptimizeLLK.  minimizer: 'oracle' (C restatement) or 'reference'
        (oracle/_ref AmoebaMinimizer)."""
        k = self.flat.num_pc
        fpc = None if fix_pc is None else np.ascontiguousarray(fix_pc, dtype=np.float64)
        opt = _Options(int(not within_ancestry), int(fix_pc is not None),
                       int(fix_pc is None and fix_alpha is not None),
                       float(fix_alpha if fix_alpha is not None else 0.0), _ptr(fpc),
                       float(epsilon), int(num_thread), int(bool(verbose)))
        mini = None
        if minimizer == "reference":
            r = ref_lib()
            if r is None:
                raise RuntimeError("oracle/_ref/libvb2ref.so not built")
            mini = C.cast(r.vb2ref_amoeba_minimize, C.c_void_p)
        tr, bufs = None, None
        if trace_capacity:
            bufs = dict(alpha=np.zeros(trace_capacity), pc1=np.zeros((trace_capacity, k)),
                        pc2=np.zeros((trace_capacity, k)), llk=np.zeros(trace_capacity))
            tr = _Trace(trace_capacity, 0, _ptr(bufs["alpha"]), _ptr(bufs["pc1"]),
                        _ptr(bufs["pc2"]), _ptr(bufs["llk"]))
        pc, pc2 = np.zeros(k), np.zeros(k)
        res = _Result()
        rc = lib().vb2o_optimize_llk(C.byref(self.c), C.byref(opt), mini,
                                     C.byref(tr) if tr is not None else None,
                                     _ptr(pc), _ptr(pc2), C.byref(res))
        if rc != 0:
            raise RuntimeError("vb2o_optimize_llk failed: %d" % rc)
        out = dict(alpha=res.alpha, llk1=res.llk1, llk0=res.llk0, num_eval=int(res.num_eval),
                   converged=bool(res.converged), pc=pc, pc2=pc2)
        if tr is not None:
            n = int(min(tr.count, trace_capacity))
            out["trace"] = {key: val[:n] for key, val in bufs.items()}
            out["trace_count"] = int(tr.count)
        return out


def amoeba(fn, start, ftol=1e-8, which="oracle"):
    """Minimise a Python callable with either Nelder-Mead implementation."""
    n = len(start)
    cb = _OBJECTIVE(lambda _u, v, nn: float(fn(np.ctypeslib.as_array(v, shape=(nn,)).copy())))
    point = (C.c_double * n)(*[float(x) for x in start])
    if which == "oracle":
        ret = lib().vb2o_amoeba_minimize(cb, None, n, point, float(ftol))
    else:
        ret = ref_lib().vb2ref_amoeba_minimize(cb, None, n, point, float(ftol))
    return ret, np.array(list(point))


_SCALAR = C.CFUNCTYPE(C.c_double, C.c_void_p, C.c_double)


def reference_scalar_minimize(fn, lo, hi, tol=1e-6):
    """The reference's own ScalarMinimizer (MathGold.cpp: Bracket + Brent, compiled in place in
    oracle/_ref) on a Python callable.  Returns dict(min, fmin, a, b, c)."""
    r = ref_lib()
    r.vb2ref_scalar_minimize.restype = None
    r.vb2ref_scalar_minimize.argtypes = [_SCALAR, C.c_void_p, C.c_double, C.c_double, C.c_double, C.POINTER(C.c_double)]
    out = (C.c_double * 5)()
    cb = _SCALAR(lambda _u, x: float(fn(x)))
    r.vb2ref_scalar_minimize(cb, None, float(lo), float(hi), float(tol), out)
    return dict(zip(("min", "fmin", "a", "b", "c"), list(out)))


class _AdapterIO(C.Structure):
    _fields_ = [
        ("num_pc", C.c_int32), ("is_heter", C.c_int32), ("is_pc_fixed", C.c_int32), ("is_alpha_fixed", C.c_int32),
        ("fix_alpha", C.c_double), ("epsilon", C.c_double), ("fix_pc", C.c_void_p),
        ("alpha", C.c_double), ("llk1", C.c_double), ("llk0", C.c_double),
        ("pc", C.c_void_p), ("pc2", C.c_void_p), ("num_eval", C.c_int64),
        ("trace", C.c_void_p), ("trace_capacity", C.c_int64), ("trace_count", C.c_int64),
        ("error", C.c_int32),
    ]


def reference_optimiser_on_gpu(product_lib, ctx_handle, num_pc, within_ancestry=False, fix_pc=None,
                               fix_alpha=None, epsilon=1e-8, trace_capacity=8192, bracket=True):
    """INTEGRATION.md section A, executed (oracle/ref_adapter.cpp): the REFERENCE's compiled
    AmoebaMinimizer drives the product's GPU likelihood through the C-ABI -- a VectorFunc subclass
    whose ComputeMixLLKs is vb2_llk_eval_batch, inside vb2_ctx_search_begin/end.  `product_lib` is
    the loaded libvb2.so (ctypes), `ctx_handle` a vb2_ctx*.  Returns the estimate and the trace of
    every Evaluate the reference made."""
    r = ref_lib()
    if r is None:
        raise RuntimeError("oracle/_ref/libvb2ref.so was never built (needs /root/reference once)")
    k = int(num_pc)
    fpc = None if fix_pc is None else np.ascontiguousarray(fix_pc, dtype=np.float64)
    pc, pc2 = np.zeros(k), np.zeros(k)
    trace = np.zeros((trace_capacity, 2 * k + 2))
    io = _AdapterIO(k, int(not within_ancestry), int(fix_pc is not None),
                    int(fix_pc is None and fix_alpha is not None),
                    float(fix_alpha if fix_alpha is not None else 0.0), float(epsilon),
                    None if fpc is None else fpc.ctypes.data,
                    0.0, 0.0, 0.0, pc.ctypes.data, pc2.ctypes.data, 0,
                    trace.ctypes.data, trace_capacity, 0, 0)
    addr = lambda f: C.cast(f, C.c_void_p)
    r.vb2ref_adapter_optimize.restype = C.c_int
    r.vb2ref_adapter_optimize.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(_AdapterIO)]
    rc = r.vb2ref_adapter_optimize(addr(product_lib.vb2_llk_eval_batch),
                                   addr(product_lib.vb2_ctx_search_begin) if bracket else None,
                                   addr(product_lib.vb2_ctx_search_end) if bracket else None,
                                   ctx_handle, C.byref(io))
    if rc:
        raise RuntimeError("evaluator failed inside the reference's optimiser: %d" % rc)
    n = int(min(io.trace_count, trace_capacity))
    return dict(alpha=io.alpha, llk1=io.llk1, llk0=io.llk0, pc=pc, pc2=pc2, num_eval=int(io.num_eval),
                trace=dict(alpha=trace[:n, 0].copy(), llk=trace[:n, 1].copy(), pc1=trace[:n, 2:2 + k].copy(),
                           pc2=trace[:n, 2 + k:].copy()), trace_count=int(io.trace_count))
