"""One evaluation per call (INTEGRATION.md option A: the reference's optimiser drives): plain
launches vs the vb2_ctx_search_begin/end bracket, C3-sized sample."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import verifybamid_amd as vb
d = vb.synth.make_pileup(100000, 30, 4, 0.05, 2)
rng = np.random.default_rng(3)
pts = [(rng.normal(0, 0.03, size=(1, 4)), rng.normal(0, 0.03, size=(1, 4)), rng.uniform(0.01, 0.3, size=1)) for _ in range(2000)]
with vb.LikelihoodContext(d) as ctx:
    for p in pts[:500]: ctx.llk(*p)
    t0 = time.perf_counter(); a = [ctx.llk(*p)[0] for p in pts]; t_plain = time.perf_counter() - t0
    with ctx.search():
        for p in pts[:500]: ctx.llk(*p)
        t0 = time.perf_counter(); b = [ctx.llk(*p)[0] for p in pts]; t_br = time.perf_counter() - t0
    assert a == b
    print("one point per call, 100k markers: %.1f us per call plain, %.1f us inside the search bracket (python call overhead included)"
          % (1e6 * t_plain / len(pts), 1e6 * t_br / len(pts)))
