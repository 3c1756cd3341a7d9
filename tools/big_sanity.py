"""Sanity at sizes beyond the bench: 1M markers, deep targeted data, wide-quality cohort."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import verifybamid_amd as vb
from oracle.bridge import oracle_data
rng = np.random.default_rng(3)
def check(d, k, B=5, tag=""):
    od = oracle_data(d)
    pc1 = rng.normal(0, 0.03, size=(B, k)); pc2 = rng.normal(0, 0.03, size=(B, k)); al = rng.uniform(0, 0.4, size=B)
    t0 = time.perf_counter(); ctx = vb.LikelihoodContext(d); t1 = time.perf_counter()
    got = ctx.llk(pc1, pc2, al)
    want = np.array([od.llk(pc1[i], pc2[i], al[i], num_thread=16) for i in range(B)])
    t2 = time.perf_counter(); [ctx.llk(pc1, pc2, al) for _ in range(10)]; t3 = time.perf_counter()
    info = ctx.info()
    print("%s: create %.2fs, rel err %.2e, %.1f us per %d-point call, codes %d, tiles %d, HBM %.1f MB"
          % (tag, t1 - t0, np.max(np.abs(got - want) / np.abs(want)), 1e5 * (t3 - t2), B, info["num_code"], info["num_tile"], info["device_bytes"] / 1e6))
    return ctx
c = check(vb.synth.make_pileup(1000000, 30, 4, seed=5), 4, tag="1M markers x 30"); c.close()
c = check(vb.synth.make_pileup(5000, 1000, 4, seed=6), 4, tag="5k markers x depth 1000"); c.close()
c = check(vb.synth.make_pileup(20000, 40, 10, seed=7, q_lo=0, q_hi=93), 10, tag="20k markers, q 0..93, k=10")
# parameter points far outside the plausible range (allele frequencies beyond [0, 1], alpha 0 / 0.5 / 1)
for (M, depth, k, seed) in ((10000, 30, 2, 1), (3000, 200, 4, 9), (2000, 5, 3, 4)):
    d = vb.synth.make_pileup(M, depth, k, seed=seed)
    od = oracle_data(d)
    for scale in (0.3, 2.0):
        r2 = np.random.default_rng(11)
        B = 16
        pc1 = r2.normal(0, scale, size=(B, k)); pc2 = r2.normal(0, scale, size=(B, k)); al = r2.uniform(0, 1.0, size=B)
        al[0] = 0.0; al[1] = 1.0; al[2] = 0.5
        with vb.LikelihoodContext(d) as cx:
            got = cx.llk(pc1, pc2, al)
        want = np.array([od.llk(pc1[i], pc2[i], al[i], num_thread=8) for i in range(B)])
        err = np.max(np.abs(got - want) / np.abs(want))
        assert err < 1e-12, (M, depth, k, scale, err)
print("extreme parameter points ok")
d2 = [vb.synth.make_pileup(3000 + 500 * i, 20 + i, 10, seed=20 + i, q_lo=0, q_hi=93) for i in range(5)]
ctxs = [vb.LikelihoodContext(d) for d in d2]
with vb.CohortBatch(ctxs) as b:
    ests = b.optimize()
for cx, e in zip(ctxs, ests):
    one = cx.optimize()
    assert abs(one["alpha"] - e["alpha"]) < 1e-6, (one["alpha"], e["alpha"])
print("wide-dictionary cohort ok:", [round(e["alpha"], 4) for e in ests])
