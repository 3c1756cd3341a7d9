"""Where a marker-shard step's time goes with ONE rank (no peers): the synchronous single-context call,
the 1-rank RCCL group (launch + ncclAllReduce + publish), both publish variants.  48 and 4 points."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import verifybamid_amd as vb
d = vb.synth.make_pileup(100000, 30, 4, alpha_true=0.05, seed=2)
rng = np.random.default_rng(1)
def pts(B): return rng.normal(0, .03, (B, 4)), rng.normal(0, .03, (B, 4)), rng.uniform(.01, .3, B)
def rate(fn, n=400):
    for _ in range(50): fn()
    t = time.perf_counter()
    for _ in range(n): fn()
    return 1e6 * (time.perf_counter() - t) / n
with vb.LikelihoodContext(d) as ctx:
    p48 = pts(48)
    t_end = time.perf_counter() + 0.3           # an idle MI355X needs a few hundred ms of load to reach its clocks
    while time.perf_counter() < t_end: ctx.llk(*p48)
    for B in (48, 4):
        p = pts(B)
        print("single context, synchronous host call, %2d points: %.1f us" % (B, rate(lambda: ctx.llk(*p))))
uid = vb.ShardGroup.unique_id()
with vb.ShardGroup(d, device=0, rank=0, nranks=1, unique_id=uid) as g:
    for B in (48, 4):
        p = pts(B)
        print("1-rank RCCL group , %2d points: %.1f us" % (B, rate(lambda: g.llk(*p))))
    t = time.perf_counter(); e = g.optimize(); print("optimize %.2f ms" % (1e3 * (time.perf_counter() - t)), e["alpha"])
with vb.ShardGroup(d, devices=[0] * 8) as g:
    for B in (48, 4):
        p = pts(B)
        print("8 virtual shards on one device (host sum), %2d points: %.1f us" % (B, rate(lambda: g.llk(*p), 100)))
