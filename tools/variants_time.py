"""Optimiser variants (vb2_ctx_optimize_llk_ex) at C3 size: multi-start searches in lock-step on one
context (a step's points of all restarts in ONE launch) against the same number of plain runs one
after the other; Brent's line search against the two-vertex simplex on the one-parameter model."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import verifybamid_amd as vb
d = vb.synth.make_pileup(100000, 30, 4, 0.05, 2)
def best_of(f, n=3):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); r = f(); ts.append(time.perf_counter() - t0)
    return 1e3 * min(ts), r
with vb.LikelihoodContext(d) as ctx:
    ctx.optimize()
    t1, plain = best_of(ctx.optimize)
    print("plain search: %.2f ms (%d evaluations)" % (t1, plain["num_eval"]))
    for n in (2, 4, 8, 12, 24):
        t, (best, every) = best_of(lambda: ctx.optimize_ex(num_start=n, seed=1))
        pts = sum(e["num_launch_point"] for e in every)
        print("%2d starts in lock-step: %.2f ms = %.2f ms per start (%.1fx one plain search each), %d points launched, "
              "alpha spread %.2e, best start %d" % (n, t, t / n, n * t1 / t, pts,
                                                   max(e["alpha"] for e in every) - min(e["alpha"] for e in every), best["start"]))
    fix = [0.01, -0.02, 0.005, 0.0]
    ts, simplex = best_of(lambda: ctx.optimize(fix_pc=fix, within_ancestry=True))
    tb, (brent, _) = best_of(lambda: ctx.optimize_ex(line_search=True, fix_pc=fix, within_ancestry=True))
    print("one-parameter model: simplex %.2f ms (%d evaluations, alpha %.7f), Brent %.2f ms (%d evaluations, alpha %.7f)"
          % (ts, simplex["num_eval"], simplex["alpha"], tb, brent["num_eval"], brent["alpha"]))
