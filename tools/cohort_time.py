"""Cohort (multi-sample lock-step) throughput vs one-sample-at-a-time, C3-shaped samples."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import verifybamid_amd as vb
S = int(os.environ.get("VB2_S", 16)); M = int(os.environ.get("VB2_M", 100000)); k = 4
datas = [vb.synth.make_pileup(M, 30, k, alpha_true=0.01 * (1 + s % 20), seed=1000 + s) for s in range(S)]
t0 = time.perf_counter(); ctxs = [vb.LikelihoodContext(d) for d in datas]; t1 = time.perf_counter()
print("created %d contexts in %.2f s" % (S, t1 - t0))
ctxs[0].optimize()
t0 = time.perf_counter(); one = [c.optimize() for c in ctxs]; t_seq = time.perf_counter() - t0
with vb.CohortBatch(ctxs) as b:
    b.optimize()
    t0 = time.perf_counter(); res = b.optimize(); t_bat = time.perf_counter() - t0
    # raw evaluation throughput: S samples x 8 points per launch
    rng = np.random.default_rng(1)
    npt = np.full(S, 8, dtype=np.int32)
    pc1 = rng.normal(0, 0.03, size=(S, 8, k)); pc2 = rng.normal(0, 0.03, size=(S, 8, k)); al = rng.uniform(0.01, 0.3, size=(S, 8))
    def rate(n):                                               # best of 3 runs of 100 launches, past the clock ramp
        for _ in range(100): b.eval(n, pc1, pc2, al)
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(100): b.eval(n, pc1, pc2, al)
            best = min(best, (time.perf_counter() - t0) / 100)
        return best
    dt8 = rate(npt)
    npt4 = np.full(S, 4, dtype=np.int32)
    dt4 = rate(npt4)
da = max(abs(r["alpha"] - o["alpha"]) for r, o in zip(res, one))
print("S=%d M=%d: sequential %.1f ms (%.2f ms/sample), lock-step cohort %.1f ms (%.2f ms/sample), max |dalpha| %.1e"
      % (S, M, 1e3 * t_seq, 1e3 * t_seq / S, 1e3 * t_bat, 1e3 * t_bat / S, da))
print("cohort eval: 8 pts/sample %.1f us/launch = %.2f us/eval (%.0f evals/s); 4 pts/sample %.1f us/launch = %.2f us/eval"
      % (1e6 * dt8, 1e6 * dt8 / (8 * S), 8 * S / dt8, 1e6 * dt4, 1e6 * dt4 / (4 * S)))
for c in ctxs: c.close()
