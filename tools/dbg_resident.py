import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import verifybamid_amd as vb
from oracle.bridge import oracle_data
d = vb.synth.make_pileup(10000, 30, 2, 0.05, 1)
od = oracle_data(d)
ref = od.optimize(trace_capacity=4096)
for rep in range(3):
    with vb.LikelihoodContext(d) as ctx:
        est = ctx.optimize(trace_capacity=4096)
        n = min(len(est["trace"]["llk"]), len(ref["trace"]["llk"]))
        bad = [i for i in range(n) if abs(est["trace"]["llk"][i] - ref["trace"]["llk"][i]) > 1e-9 * abs(ref["trace"]["llk"][i])]
        print("rep", rep, "n", n, "bad", bad[:10], "alpha", est["alpha"], ref["alpha"])
        for i in bad[:3]:
            p1, p2, a = est["trace"]["pc1"][i], est["trace"]["pc2"][i], est["trace"]["alpha"][i]
            direct = ctx.llk(p1[None], p2[None], np.array([a]))[0]
            print("  i", i, "trace", est["trace"]["llk"][i], "ref", ref["trace"]["llk"][i], "direct", direct,
                  "same point", np.array_equal(p1, ref["trace"]["pc1"][i]))
