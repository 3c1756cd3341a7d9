"""Thin Python marshalling over the C-ABI (include/vb2_abi.h).

Names follow the reference's domain: a *panel* (UD, mu, bed markers), a *pileup*
(bases/quals per marker), a likelihood *context* (the data ComputeMixLLKs sees),
an *estimate* (alpha, PCs, llk1, llk0).  All compute happens in libvb2.so on the
GPU; nothing here evaluates a likelihood.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import _abi


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


@dataclass
class PileupData:
    """Host arrays in the shape of vb2_input: panel-ordered markers, marker i owning
    bases/quals[read_off[i]:read_off[i+1]] (empty = absent from the pileup)."""
    num_pc: int
    ud: np.ndarray                  # [M, k] f64
    means: np.ndarray               # [M] f64
    read_off: np.ndarray            # [M+1] i64
    bases: np.ndarray               # [R] u8
    quals: np.ndarray               # [R] u8
    alt_base: np.ndarray            # [M] u8
    known_af: np.ndarray | None = None
    avg_depth: float = 0.0
    sd_depth: float = 0.0
    sanity_disabled: bool = True
    meta: dict = field(default_factory=dict)

    def __post_init__(self):
        self.ud = np.ascontiguousarray(self.ud, dtype=np.float64).reshape(-1, self.num_pc)
        self.means = np.ascontiguousarray(self.means, dtype=np.float64)
        self.read_off = np.ascontiguousarray(self.read_off, dtype=np.int64)
        self.bases = np.ascontiguousarray(self.bases, dtype=np.uint8)
        self.quals = np.ascontiguousarray(self.quals, dtype=np.uint8)
        self.alt_base = np.ascontiguousarray(self.alt_base, dtype=np.uint8)
        if self.known_af is not None:
            self.known_af = np.ascontiguousarray(self.known_af, dtype=np.float64)

    @property
    def num_marker(self):
        return int(self.read_off.shape[0] - 1)

    @property
    def num_read(self):
        return int(self.read_off[-1] - self.read_off[0])

    def as_input(self):
        return _abi.Input(self.num_marker, self.num_pc, _p(self.ud), _p(self.means),
                          _p(self.read_off), _p(self.bases), _p(self.quals), _p(self.alt_base),
                          _p(self.known_af), float(self.avg_depth), float(self.sd_depth),
                          int(bool(self.sanity_disabled)), 0)

    def shard_range(self, rank, world):
        """Markers [lo, hi) of shard `rank` of `world` -- the library's own partition (vb2_shard_range)."""
        inp = self.as_input()
        lo, hi = C.c_int32(), C.c_int32()
        _abi.check(_abi.lib().vb2_shard_range(C.byref(inp), int(rank), int(world), C.byref(lo), C.byref(hi)),
                   "vb2_shard_range")
        return lo.value, hi.value

    def shard(self, rank, world):
        """Contiguous marker range holding ~1/world of the READS (balance on R, not M:
        SURVEY 8e).  The depth filter statistics stay global."""
        if world <= 1:
            return self
        lo, hi = self.shard_range(rank, world)
        b, e = int(self.read_off[lo]), int(self.read_off[hi])
        return PileupData(self.num_pc, self.ud[lo:hi], self.means[lo:hi],
                          self.read_off[lo:hi + 1] - b, self.bases[b:e], self.quals[b:e],
                          self.alt_base[lo:hi],
                          None if self.known_af is None else self.known_af[lo:hi],
                          self.avg_depth, self.sd_depth, self.sanity_disabled,
                          dict(self.meta, shard=(rank, world), marker_range=(lo, hi)))

    @staticmethod
    def from_files(svd_prefix, pileup_path, num_pc=2, disable_sanity=True, known_af_path=None):
        """Panel + pileup readers of the library (vb2_flat_load), copied into numpy."""
        L = _abi.lib()
        args, keep = _run_args(svd_prefix, pileup_path, num_pc, disable_sanity, known_af_path, None)
        h = C.c_void_p()
        rc = L.vb2_flat_load(C.byref(args), C.byref(h))
        if rc not in (_abi.VB2_OK, _abi.VB2_ERR_SANITY) or not h:
            _abi.check(rc, "vb2_flat_load")
        try:
            inp = L.vb2_flat_input(h).contents
            M, k = inp.num_marker, inp.num_pc
            off = np.ctypeslib.as_array(C.cast(inp.read_off, C.POINTER(C.c_int64)), (M + 1,)).copy()
            R = int(off[-1])

            def arr(ptr, ctype, n, dt):
                if not ptr or n == 0:
                    return np.zeros(n, dtype=dt)
                return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), (n,)).astype(dt, copy=True)
            st = _abi.RunResult()
            L.vb2_flat_stats(h, C.byref(st))
            return PileupData(
                k, arr(inp.ud, C.c_double, M * k, np.float64), arr(inp.means, C.c_double, M, np.float64),
                off, arr(inp.bases, C.c_uint8, R, np.uint8), arr(inp.quals, C.c_uint8, R, np.uint8),
                arr(inp.alt_base, C.c_uint8, M, np.uint8),
                arr(inp.known_af, C.c_double, M, np.float64) if inp.known_af else None,
                inp.avg_depth, inp.sd_depth, bool(inp.sanity_disabled),
                dict(num_site=st.num_site, num_bases=st.num_bases, sanity_ok=(rc == _abi.VB2_OK)))
        finally:
            L.vb2_flat_free(h)


def _model(within_ancestry=False, fix_pc=None, fix_alpha=None, known_af=False, epsilon=1e-8,
           verbose=False):
    fpc = None if fix_pc is None else np.ascontiguousarray(fix_pc, dtype=np.float64)
    m = _abi.Model(int(not within_ancestry), int(fix_pc is not None),
                   int(fix_pc is None and fix_alpha is not None), int(bool(known_af)),
                   float(fix_alpha if fix_alpha is not None else 0.0), _p(fpc), float(epsilon),
                   int(bool(verbose)), 0)
    return m, fpc


def _run_args(svd_prefix, pileup_path, num_pc, disable_sanity, known_af_path, output_prefix,
              device=-1, output_pileup=False, devices=None, num_start=1, seed=0, line_search=False, **model_kw):
    m, keep = _model(known_af=known_af_path is not None, **model_kw)
    enc = lambda s: None if s is None else str(s).encode()
    devs = None
    if devices is not None and len(devices) > 0:
        devs = (C.c_int32 * len(devices))(*[int(d) for d in devices])
    args = _abi.RunArgs(enc(svd_prefix + ".UD"), enc(svd_prefix + ".mu"), enc(svd_prefix + ".bed"),
                        enc(pileup_path), enc(known_af_path), enc(output_prefix), int(num_pc),
                        int(bool(disable_sanity)), int(bool(output_pileup)), int(device), m,
                        devs, 0 if devs is None else len(devices), 0, None, None,
                        _abi.SearchOpts(int(num_start), int(seed), 0.0, 1 if line_search else 0, 0))
    return args, (keep, devs)


def _estimate_dict(est, k):
    return dict(alpha=est.alpha, llk1=est.llk1, llk0=est.llk0,
                pc=np.array(est.pc[:k]), pc2=np.array(est.pc2[:k]),
                num_eval=int(est.num_eval), num_launch_point=int(est.num_launch_point),
                converged=bool(est.converged))


class _TraceBuf:
    def __init__(self, capacity, k):
        self.bufs = dict(alpha=np.zeros(capacity), pc1=np.zeros((capacity, k)),
                         pc2=np.zeros((capacity, k)), llk=np.zeros(capacity))
        self.c = _abi.Trace(capacity, 0, _p(self.bufs["alpha"]), _p(self.bufs["pc1"]),
                            _p(self.bufs["pc2"]), _p(self.bufs["llk"]))
        self.capacity = capacity

    def result(self):
        n = int(min(self.c.count, self.capacity))
        return {k: v[:n] for k, v in self.bufs.items()}, int(self.c.count)


class LikelihoodContext:
    """vb2_ctx: the pileup + panel resident in HBM, ready for batched evaluation."""

    def __init__(self, data: PileupData, device=-1, stream=None, cohort_layout=False):
        self._lib = _abi.lib()
        self.data = data
        self.num_pc = data.num_pc
        inp = data.as_input()
        opt = _abi.Options(int(device), 1 if cohort_layout else 0,      # VB2_OPT_COHORT_LAYOUT
                           C.c_void_p(stream) if stream else None)
        h = C.c_void_p()
        _abi.check(self._lib.vb2_ctx_create(C.byref(inp), C.byref(opt), C.byref(h)), "vb2_ctx_create")
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._lib.vb2_ctx_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def info(self):
        i = _abi.Info()
        _abi.check(self._lib.vb2_ctx_info(self._h, C.byref(i)), "vb2_ctx_info")
        d = {name: getattr(i, name) for name, _ in _abi.Info._fields_}
        d["device_name"] = i.device_name.decode()
        d["arch"] = i.arch.decode()
        return d

    def llk(self, pc1, pc2, alpha):
        """+LLK for B points (ComputeMixLLKs, batched).  pc1/pc2: [B,k] (or [k]); alpha: [B]."""
        pc1 = np.ascontiguousarray(np.atleast_2d(np.asarray(pc1, dtype=np.float64)))
        pc2 = np.ascontiguousarray(np.atleast_2d(np.asarray(pc2, dtype=np.float64)))
        alpha = np.ascontiguousarray(np.atleast_1d(np.asarray(alpha, dtype=np.float64)))
        B = alpha.shape[0]
        assert pc1.shape == (B, self.num_pc) and pc2.shape == (B, self.num_pc)
        out = np.zeros(B)
        _abi.check(self._lib.vb2_llk_eval_batch(self._h, B, _p(pc1), _p(pc2), _p(alpha), _p(out)),
                   "vb2_llk_eval_batch")
        return out

    def llk_device(self, points_ptr, out_ptr, num_point, stream=None):
        """Enqueue an evaluation on device pointers (rows = pc1|pc2|alpha); no host sync."""
        _abi.check(self._lib.vb2_llk_eval_batch_device(self._h, int(num_point), C.c_void_p(points_ptr),
                                                       C.c_void_p(out_ptr),
                                                       C.c_void_p(stream) if stream else None),
                   "vb2_llk_eval_batch_device")

    def search(self):
        """Context manager around a series of dependent llk() calls (a search driven by the
        caller): vb2_ctx_search_begin / vb2_ctx_search_end."""
        ctx = self

        class _Search:
            def __enter__(self_inner):
                _abi.check(ctx._lib.vb2_ctx_search_begin(ctx._h), "vb2_ctx_search_begin")
                return ctx

            def __exit__(self_inner, *exc):
                ctx._lib.vb2_ctx_search_end(ctx._h)
                return False

        return _Search()

    def optimize(self, trace_capacity=0, **model_kw):
        """OptimizeLLK on this context.  model_kw: within_ancestry, fix_pc, fix_alpha, epsilon..."""
        m, keep = _model(known_af=self.data.known_af is not None, **model_kw)
        est = _abi.Estimate()
        tb = _TraceBuf(trace_capacity, self.num_pc) if trace_capacity else None
        _abi.check(self._lib.vb2_ctx_optimize_llk(self._h, C.byref(m), C.byref(est),
                                                  C.byref(tb.c) if tb else None),
                   "vb2_ctx_optimize_llk")
        out = _estimate_dict(est, self.num_pc)
        if tb:
            out["trace"], out["trace_count"] = tb.result()
        return out


    def optimize_ex(self, num_start=1, seed=0, start_sd=0.0, line_search=False, **model_kw):
        """Optimiser variants (vb2_ctx_optimize_llk_ex): num_start searches from seeded starting
        points in lock-step (start 0 = the reference's), and / or Brent's line search for the
        one-parameter models.  Returns (best, all): best has "start" = index of the winning run."""
        m, keep = _model(known_af=self.data.known_af is not None, **model_kw)
        n = max(1, int(num_start))
        opts = _abi.SearchOpts(n, int(seed), float(start_sd), 1 if line_search else 0, 0)
        best = _abi.Estimate()
        every = (_abi.Estimate * n)()
        _abi.check(self._lib.vb2_ctx_optimize_llk_ex(self._h, C.byref(m), C.byref(opts), C.byref(best),
                                                     C.cast(every, C.c_void_p)),
                   "vb2_ctx_optimize_llk_ex")
        out = _estimate_dict(best, self.num_pc)
        out["start"] = int(best.reserved)
        return out, [_estimate_dict(every[i], self.num_pc) for i in range(n)]


class CohortBatch:
    """vb2_batch: several LikelihoodContexts (one device, same --NumPC) evaluated and
    optimised in lock-step, one kernel launch per step for the whole cohort."""
    SLOTS = 8

    def __init__(self, contexts):
        self._lib = _abi.lib()
        self.contexts = list(contexts)
        self.num_pc = self.contexts[0].num_pc
        n = len(self.contexts)
        arr = (C.c_void_p * n)(*[c._h for c in self.contexts])
        h = C.c_void_p()
        _abi.check(self._lib.vb2_batch_create(arr, n, C.byref(h)), "vb2_batch_create")
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._lib.vb2_batch_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def eval(self, num_point, pc1, pc2, alpha):
        """num_point: [S] ints (0..8); pc1/pc2: [S,8,k]; alpha: [S,8] -> llk [S,8]."""
        S, k = len(self.contexts), self.num_pc
        npt = np.ascontiguousarray(num_point, dtype=np.int32)
        pc1 = np.ascontiguousarray(pc1, dtype=np.float64).reshape(S, self.SLOTS, k)
        pc2 = np.ascontiguousarray(pc2, dtype=np.float64).reshape(S, self.SLOTS, k)
        alpha = np.ascontiguousarray(alpha, dtype=np.float64).reshape(S, self.SLOTS)
        out = np.zeros((S, self.SLOTS))
        _abi.check(self._lib.vb2_batch_eval(self._h, _p(npt), _p(pc1), _p(pc2), _p(alpha), _p(out)),
                   "vb2_batch_eval")
        return out

    def prepared_eval(self, num_point, pc1, pc2, alpha):
        """The same call with the argument marshalling done once: returns (step, out) where step() makes one
        vb2_batch_eval call on the captured arrays (bench.py times steps, not numpy conversions)."""
        S, k = len(self.contexts), self.num_pc
        npt = np.ascontiguousarray(num_point, dtype=np.int32)
        pc1 = np.ascontiguousarray(pc1, dtype=np.float64).reshape(S, self.SLOTS, k)
        pc2 = np.ascontiguousarray(pc2, dtype=np.float64).reshape(S, self.SLOTS, k)
        alpha = np.ascontiguousarray(alpha, dtype=np.float64).reshape(S, self.SLOTS)
        out = np.zeros((S, self.SLOTS))
        fn, h, a = self._lib.vb2_batch_eval, self._h, (_p(npt), _p(pc1), _p(pc2), _p(alpha), _p(out))
        keep = (npt, pc1, pc2, alpha)

        def step(_keep=keep):
            rc = fn(h, *a)
            if rc:
                _abi.check(rc, "vb2_batch_eval")
        return step, out

    def optimize(self, **model_kw):
        S = len(self.contexts)
        m, keep = _model(**model_kw)
        est = (_abi.Estimate * S)()
        _abi.check(self._lib.vb2_batch_optimize_llk(self._h, C.byref(m), 1, est),
                   "vb2_batch_optimize_llk")
        return [_estimate_dict(est[s], self.num_pc) for s in range(S)]


class ShardGroup:
    """vb2_shard_group: ONE sample's markers sharded over several GPUs, partial LLKs met in one
    RCCL all-reduce per batch (BASELINE.json configs[3]).

    ShardGroup(data, devices=[0, 1, ...])            one process drives all devices
    ShardGroup(data, device=d, rank=r, nranks=n, unique_id=b)   one process per GPU; rank 0 gets the
        128-byte id from ShardGroup.unique_id() and the caller broadcasts it (torch.distributed,
        MPI, a file ...).  unique_id=ShardGroup.PARTIAL_SUMS: no communicator, llk() returns this rank's
        partial sums (the caller reduces them); unique_id=None is accepted with nranks == 1 only."""

    PARTIAL_SUMS = "partial-sums"

    def __init__(self, data: PileupData, devices=None, device=0, rank=0, nranks=1, unique_id=None):
        self._lib = _abi.lib()
        self.data = data
        self.num_pc = data.num_pc
        inp = data.as_input()
        h = C.c_void_p()
        if devices is not None:
            arr = (C.c_int32 * len(devices))(*[int(d) for d in devices])
            _abi.check(self._lib.vb2_shard_group_create(C.byref(inp), arr, len(devices), C.byref(h)),
                       "vb2_shard_group_create")
        else:
            if isinstance(unique_id, str) and unique_id == self.PARTIAL_SUMS:
                idbuf = C.c_void_p(1)                      # VB2_SHARD_PARTIAL_SUMS
            else:
                idbuf = None if unique_id is None else C.create_string_buffer(bytes(unique_id), 128)
            _abi.check(self._lib.vb2_shard_group_create_rank(C.byref(inp), int(device), int(rank), int(nranks),
                                                             idbuf, C.byref(h)),
                       "vb2_shard_group_create_rank")
        self._h = h

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(128)
        _abi.check(_abi.lib().vb2_rccl_unique_id(buf), "vb2_rccl_unique_id")
        return buf.raw

    def close(self):
        if getattr(self, "_h", None):
            self._lib.vb2_shard_group_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def info(self):
        i = _abi.ShardInfo()
        _abi.check(self._lib.vb2_shard_group_info(self._h, C.byref(i)), "vb2_shard_group_info")
        n = i.num_shard
        return dict(num_shard=n, nranks=i.nranks, rank=i.rank, uses_rccl=bool(i.uses_rccl),
                    partial_sums=bool(i.partial_sums), rccl_stub=bool(i.rccl_stub),
                    num_allreduce=int(i.num_allreduce), marker_lo=list(i.marker_lo[:n]),
                    marker_hi=list(i.marker_hi[:n]), num_read=[int(x) for x in i.num_read[:n]])

    def llk(self, pc1, pc2, alpha):
        pc1 = np.ascontiguousarray(np.atleast_2d(np.asarray(pc1, dtype=np.float64)))
        pc2 = np.ascontiguousarray(np.atleast_2d(np.asarray(pc2, dtype=np.float64)))
        alpha = np.ascontiguousarray(np.atleast_1d(np.asarray(alpha, dtype=np.float64)))
        B = alpha.shape[0]
        assert pc1.shape == (B, self.num_pc) and pc2.shape == (B, self.num_pc)
        out = np.zeros(B)
        _abi.check(self._lib.vb2_shard_group_eval(self._h, B, _p(pc1), _p(pc2), _p(alpha), _p(out)),
                   "vb2_shard_group_eval")
        return out

    def prepared_llk(self, pc1, pc2, alpha):
        """llk() with the argument marshalling done once: returns (step, out); step() makes one
        vb2_shard_group_eval call on the captured arrays (bench.py times steps, not numpy conversions)."""
        pc1 = np.ascontiguousarray(np.atleast_2d(np.asarray(pc1, dtype=np.float64)))
        pc2 = np.ascontiguousarray(np.atleast_2d(np.asarray(pc2, dtype=np.float64)))
        alpha = np.ascontiguousarray(np.atleast_1d(np.asarray(alpha, dtype=np.float64)))
        B = alpha.shape[0]
        assert pc1.shape == (B, self.num_pc) and pc2.shape == (B, self.num_pc)
        out = np.zeros(B)
        fn, h, a = self._lib.vb2_shard_group_eval, self._h, (_p(pc1), _p(pc2), _p(alpha), _p(out))

        def step(_keep=(pc1, pc2, alpha)):
            rc = fn(h, B, *a)
            if rc:
                _abi.check(rc, "vb2_shard_group_eval")
        return step, out

    def optimize(self, trace_capacity=0, **model_kw):
        m, keep = _model(known_af=self.data.known_af is not None, **model_kw)
        est = _abi.Estimate()
        tb = _TraceBuf(trace_capacity, self.num_pc) if trace_capacity else None
        _abi.check(self._lib.vb2_shard_group_optimize_llk(self._h, C.byref(m), C.byref(est),
                                                          C.byref(tb.c) if tb else None),
                   "vb2_shard_group_optimize_llk")
        out = _estimate_dict(est, self.num_pc)
        if tb:
            out["trace"], out["trace_count"] = tb.result()
        return out


def optimize_with_evaluator(evaluate, num_pc, trace_capacity=0, known_af=False, **model_kw):
    """OptimizeLLK over an arbitrary batched evaluator
        evaluate(pc1[B,k], pc2[B,k], alpha[B]) -> llk[B]
    (e.g. marker shards + an RCCL all-reduce).  The optimiser is the library's."""
    L = _abi.lib()
    k = int(num_pc)
    err = []

    def cb(_user, n, p1, p2, a, out):
        try:
            pc1 = np.ctypeslib.as_array(p1, (n, k)).copy()
            pc2 = np.ctypeslib.as_array(p2, (n, k)).copy()
            al = np.ctypeslib.as_array(a, (n,)).copy()
            res = np.asarray(evaluate(pc1, pc2, al), dtype=np.float64).reshape(n)
            np.ctypeslib.as_array(out, (n,))[:] = res
            return 0
        except Exception as exc:   # never let an exception cross the C boundary
            err.append(exc)
            return _abi.VB2_ERR_INVALID
    fn = _abi.EVAL_FN(cb)
    m, keep = _model(known_af=known_af, **model_kw)
    est = _abi.Estimate()
    tb = _TraceBuf(trace_capacity, k) if trace_capacity else None
    rc = L.vb2_optimize_llk(fn, None, k, C.byref(m), C.byref(est), C.byref(tb.c) if tb else None)
    if err:
        raise err[0]
    _abi.check(rc, "vb2_optimize_llk")
    out = _estimate_dict(est, k)
    if tb:
        out["trace"], out["trace_count"] = tb.result()
    return out


def run_files(svd_prefix, pileup_path, output_prefix=None, num_pc=2, disable_sanity=False,
              known_af_path=None, device=-1, output_pileup=False, devices=None, **model_kw):
    """The --SVDPrefix/--PileupFile flow of execute() (vb2_run); devices=[a, b, ...] shards the
    sample's markers over those GPUs."""
    args, keep = _run_args(svd_prefix, pileup_path, num_pc, disable_sanity, known_af_path,
                           output_prefix, device, output_pileup, devices=devices, **model_kw)
    res = _abi.RunResult()
    _abi.check(_abi.lib().vb2_run(C.byref(args), C.byref(res)), "vb2_run")
    out = _estimate_dict(res.est, num_pc)
    out.update(num_marker=res.num_marker, num_site=res.num_site, num_bases=int(res.num_bases),
               avg_depth=res.avg_depth, sd_depth=res.sd_depth, seconds_load=res.seconds_load,
               seconds_optimize=res.seconds_optimize)
    return out


def run_cohort_files(svd_prefix, pileup_paths, output_prefixes=None, num_pc=2, disable_sanity=False,
                     known_af_path=None, device=-1, output_pileup=False, group_size=0, num_host_thread=0,
                     devices=None, **model_kw):
    """Many pileups against one panel (vb2_cohort_run): the panel is read once, the pileups are read
    and flattened by host threads while the device searches the previous group in lock-step.
    Returns one dict per sample (with its own "status" code)."""
    S = len(pileup_paths)
    args, keep = _run_args(svd_prefix, pileup_paths[0], num_pc, disable_sanity, known_af_path, None,
                           device, output_pileup, devices=devices, **model_kw)
    ca = _abi.CohortArgs()
    ca.base = args
    ca.num_sample = S
    piles = (C.c_char_p * S)(*[str(p).encode() for p in pileup_paths])
    ca.pileup_paths = piles
    prefs = None
    if output_prefixes is not None:
        prefs = (C.c_char_p * S)(*[str(p).encode() for p in output_prefixes])
        ca.output_prefixes = prefs
    ca.group_size = int(group_size)
    ca.num_host_thread = int(num_host_thread)
    res = (_abi.RunResult * S)()
    status = (C.c_int32 * S)()
    _abi.check(_abi.lib().vb2_cohort_run(C.byref(ca), res, status), "vb2_cohort_run")
    out = []
    for s in range(S):
        d = _estimate_dict(res[s].est, num_pc)
        d.update(status=int(status[s]), num_marker=res[s].num_marker, num_site=res[s].num_site,
                 num_bases=int(res[s].num_bases), avg_depth=res[s].avg_depth, sd_depth=res[s].sd_depth,
                 seconds_load=res[s].seconds_load, seconds_optimize=res[s].seconds_optimize)
        out.append(d)
    return out
