import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import verifybamid_amd as vb
for M in (4096, 16384, 100000):
    d = vb.synth.make_pileup(M, 30, 4, 0.05, 2)
    with vb.LikelihoodContext(d) as ctx:
        ctx.optimize()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); r = ctx.optimize(); ts.append(time.perf_counter() - t0)
        rounds = r["num_launch_point"] / 4.0
        print("M=%d: %.2f ms, %d evals, %d points -> %.1f us per 4-point round" % (M, 1e3 * min(ts), r["num_eval"], r["num_launch_point"], 1e6 * min(ts) / rounds))
