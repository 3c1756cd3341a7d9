"""Launches of the other wave shapes, to be run under `rocprofv3 --kernel-trace --stats`:
4-point and 1-point launches of one context (llk_eval_kernel<3,.>, <4,.>: the operating points of a
Nelder-Mead search) and lock-step cohort steps of 32 samples with 4, 8, 2 and 1 points per sample
(llk_eval_multi_kernel<3,.>, <2,.>, <5,.>, <4,.>: the last two are the steps of a cohort search, which
speculates on {R, C_R} only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import verifybamid_amd as vb

k, S = 4, int(os.environ.get("VB2_COHORT", "32"))
distinct = [vb.synth.make_pileup(100000, 30, k, alpha_true=0.05, seed=2 + s) for s in range(4)]
ctxs = [vb.LikelihoodContext(distinct[s % 4], device=0) for s in range(S)]
rng = np.random.default_rng(123)
pts_h = np.concatenate([rng.normal(0, 0.03, size=(8, 2 * k)), rng.uniform(0.01, 0.3, size=(8, 1))], axis=1)
pts = torch.tensor(pts_h, dtype=torch.float64, device="cuda")
out = torch.zeros(8, dtype=torch.float64, device="cuda")
for _ in range(600):                       # bring the clocks up
    ctxs[0].llk_device(pts.data_ptr(), out.data_ptr(), 8)
torch.cuda.synchronize()
for nb in (4, 1):
    for _ in range(2000):
        ctxs[0].llk_device(pts.data_ptr(), out.data_ptr(), nb)
    torch.cuda.synchronize()
with vb.CohortBatch(ctxs) as batch:
    npt = np.full(S, 4, dtype=np.int32)
    p1 = np.zeros((S, 8, k)); p2 = np.zeros((S, 8, k)); al = np.full((S, 8), 0.1)
    p1[:, :4] = pts_h[:4, :k]; p2[:, :4] = pts_h[:4, k:2 * k]; al[:, :4] = pts_h[:4, 2 * k]
    for _ in range(300):
        batch.eval(npt, p1, p2, al)
    npt[:] = 8
    p1[:] = pts_h[:8, :k]; p2[:] = pts_h[:8, k:2 * k]; al[:] = pts_h[:8, 2 * k]
    for _ in range(150):
        batch.eval(npt, p1, p2, al)
    for n in (2, 1):
        npt[:] = n
        for _ in range(300):
            batch.eval(npt, p1, p2, al)
for c in ctxs:
    c.close()
