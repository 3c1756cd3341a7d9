"""One-off hunt, not a test: random shapes and models, the search done three ways on the same
context -- simplex on the device, host optimiser through the resident kernel, host optimiser with
plain launches -- and as a 2-sample cohort; everything that must agree bit for bit is compared."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import verifybamid_amd as vb
from verifybamid_amd import _abi
lib = _abi.lib()
lib.vb2_debug_set_device_simplex.argtypes = [C.c_void_p, C.c_int]
lib.vb2_debug_set_resident.argtypes = [C.c_void_p, C.c_int]
rng = np.random.default_rng(int(os.environ.get("VB2_FUZZ_SEED", 1)))
bad = 0
N = int(os.environ.get("VB2_FUZZ_N", 40))
for it in range(N):
    M = int(rng.integers(40, 6000)); depth = float(rng.choice([3, 8, 30, 120, 600])); k = int(rng.integers(1, 7))
    qlo = int(rng.integers(0, 35)); qhi = int(min(93, qlo + rng.integers(0, 40)))
    d = vb.synth.make_pileup(M, depth, k, alpha_true=float(rng.uniform(0, 0.5)), seed=int(rng.integers(1, 10**6)), q_lo=qlo, q_hi=qhi)
    fix = list(rng.normal(0, 0.02, size=k))
    models = [{}, {"within_ancestry": True}, {"fix_alpha": float(rng.uniform(0.01, 0.4))}, {"fix_pc": fix},
              {"within_ancestry": True, "fix_pc": fix}, {"within_ancestry": True, "fix_alpha": 0.2}]
    kw = models[int(rng.integers(0, 6))]
    with vb.LikelihoodContext(d) as ctx:
        dev = ctx.optimize(trace_capacity=1 << 14, **kw)
        lib.vb2_debug_set_device_simplex(ctx._h, 0)
        host = ctx.optimize(trace_capacity=1 << 14, **kw)
        lib.vb2_debug_set_resident(ctx._h, 0)
        plain = ctx.optimize(trace_capacity=1 << 14, **kw)
        lib.vb2_debug_set_resident(ctx._h, 1); lib.vb2_debug_set_device_simplex(ctx._h, 1)
    ok = True
    for other, name in ((host, "host/resident"), (plain, "host/launches")):
        for key in ("alpha", "llk1", "llk0", "num_eval"):
            if dev[key] != other[key]:
                ok = False
                print("MISMATCH it=%d M=%d depth=%g k=%d q=%d..%d %s: %s dev %r vs %s %r" % (it, M, depth, k, qlo, qhi, kw, key, dev[key], name, other[key]))
        n = min(dev["trace_count"], other["trace_count"], 1 << 14)
        if dev["trace_count"] != other["trace_count"] or not np.array_equal(dev["trace"]["llk"][:n], other["trace"]["llk"][:n]):
            ok = False
            print("TRACE MISMATCH it=%d %s vs %s" % (it, kw, name))
    bad += 0 if ok else 1
print("search fuzz: %d of %d cases disagree" % (bad, N))
