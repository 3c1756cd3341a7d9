"""Wall-clock of OptimizeLLK on C2/C3 shapes (host-driven search, GPU evaluations)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import verifybamid_amd as vb
for (M, k, seed) in [(10000, 2, 1), (100000, 4, 2)]:
    d = vb.synth.make_pileup(M, 30, k, 0.05, seed)
    with vb.LikelihoodContext(d) as ctx:
        ctx.optimize()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); r = ctx.optimize(); ts.append(time.perf_counter() - t0)
        print("M=%d k=%d optimize: best %.2f ms median %.2f ms  alpha %.7f evals %d points %d"
              % (M, k, 1e3 * min(ts), 1e3 * sorted(ts)[2], r["alpha"], r["num_eval"], r["num_launch_point"]))
