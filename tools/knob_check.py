"""Every A/B knob still gives the same results: LLKs against the oracle, wave-shape independence, one search.
for kv in NONE=1 VB2_SINGLE_LAUNCH=0 VB2_REDUCE=1 VB2_REDUCE=2 VB2_DYN_TILES=0 VB2_RESIDENT=0 VB2_SCHED=0 VB2_COOP=0 \
          VB2_DEVICE_SIMPLEX=0 VB2_LDS_CACHE=0 VB2_PASSES=0; do env KNOB=$kv $kv python tools/knob_check.py; done"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import verifybamid_amd as vb
from oracle.bridge import oracle_data
d = vb.synth.make_pileup(20000, 20, 4, 0.03, 5)
od = oracle_data(d)
rng = np.random.default_rng(2)
B = 13
pc1 = rng.normal(0, 0.03, (B, 4)); pc2 = rng.normal(0, 0.03, (B, 4)); al = rng.uniform(0.01, 0.4, B)
want = np.array([od.llk(pc1[i], pc2[i], al[i], num_thread=8) for i in range(B)])
with vb.LikelihoodContext(d) as c:
    got = c.llk(pc1, pc2, al)
    one = np.array([c.llk(pc1[i:i+1], pc2[i:i+1], al[i:i+1])[0] for i in range(3)])
    four = c.llk(pc1[:4], pc2[:4], al[:4])
    est = c.optimize()
print(os.environ.get("KNOB"), "max rel err %.2e" % np.max(np.abs(got - want) / np.abs(want)), "shape-independent", bool(np.array_equal(one, got[:3]) and np.array_equal(four, got[:4])), "alpha %.6f evals %d" % (est["alpha"], est["num_eval"]))
