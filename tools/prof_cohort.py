"""Lock-step cohort steps of 32 C3-shaped samples with 1 and 2 points per sample (llk_eval_multi_kernel<4,.>, <5,.>:
the steps of a cohort search), to be run under rocprofv3 (--stats or --pmc)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import verifybamid_amd as vb

k, S = 4, int(os.environ.get("VB2_COHORT", "32"))
N = int(os.environ.get("VB2_STEPS", "200"))
distinct = [vb.synth.make_pileup(100000, 30, k, alpha_true=0.05, seed=2 + s) for s in range(4)]
ctxs = [vb.LikelihoodContext(distinct[s % 4], device=0) for s in range(S)]
rng = np.random.default_rng(123)
p1 = rng.normal(0, 0.03, size=(S, 8, k)); p2 = rng.normal(0, 0.03, size=(S, 8, k)); al = rng.uniform(0.01, 0.3, size=(S, 8))
with vb.CohortBatch(ctxs) as batch:
    for n in (1, 2):
        step, _ = batch.prepared_eval(np.full(S, n, dtype=np.int32), p1, p2, al)
        for _ in range(N):
            step()
for c in ctxs:
    c.close()
