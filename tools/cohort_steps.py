"""Cohort step A/B on the GPU box: 32 C3-shaped samples in lock-step, time per synchronous step with 1 / 2 / 4 points per
sample and the whole cohort search (parity: tests/test_gpu_parity.py -k cohort).  One JSON line.   VB2_LIB_PATH=<other build> for a same-box A/B.

    python tools/cohort_steps.py [samples] [markers]
"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import verifybamid_amd as vb

S = int(sys.argv[1]) if len(sys.argv) > 1 else 32
M = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
k = 4
distinct = [vb.synth.make_pileup(M, 30, k, alpha_true=0.02 + 0.02 * s, seed=1000 + s) for s in range(min(S, 4))]
ctxs = [vb.LikelihoodContext(distinct[s % len(distinct)], device=0) for s in range(S)]
rng = np.random.default_rng(123)
p1 = rng.normal(0, 0.03, size=(S, 8, k)); p2 = rng.normal(0, 0.03, size=(S, 8, k)); al = rng.uniform(0.01, 0.3, size=(S, 8))
out = {"samples": S, "markers": M, "lib": os.environ.get("VB2_LIB_PATH", "default")}
with vb.CohortBatch(ctxs) as batch:
    def time_steps(n, reps=300):
        npt = np.full(S, n, dtype=np.int32)
        step, _ = batch.prepared_eval(npt, p1, p2, al)
        for _ in range(60):
            step()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(reps):
                step()
            best = min(best, (time.perf_counter() - t0) / reps)
        return 1e6 * best
    for n in (1, 2, 4, 8):
        out["step_us_%d" % n] = round(time_steps(n), 2)
    if os.environ.get("VB2_STEPS_ONLY"):
        print(json.dumps(out), flush=True)
        os._exit(0)
    batch.optimize()
    t0 = time.perf_counter(); est = batch.optimize(); dt = time.perf_counter() - t0
    out["search_ms_per_sample"] = round(1e3 * dt / S, 4)
    out["samples_per_s"] = round(S / dt, 1)
    out["alpha0"] = est[0]["alpha"]
for c in ctxs:
    c.close()
print(json.dumps(out), flush=True)
