"""One-off hunt, not a test: random shapes (incl. deep markers at the underflow boundary, wide and
narrow quality alphabets, many PCs), random parameter points, the HIP path against the oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import verifybamid_amd as vb
from oracle.bridge import oracle_data
rng = np.random.default_rng(int(os.environ.get("VB2_FUZZ_SEED", 1)))
N = int(os.environ.get("VB2_FUZZ_N", 40))
worst, bad = 0.0, 0
for it in range(N):
    depth = float(rng.choice([2, 10, 30, 60, 150, 400, 700, 820, 880, 1000, 1500]))      # (probability domain up to ~850 reads)
    M = int(rng.integers(30, 3000 if depth > 500 else 20000)); k = int(rng.integers(1, 11))
    qlo = int(rng.integers(0, 40)); qhi = int(min(93, qlo + rng.integers(0, 50)))
    d = vb.synth.make_pileup(M, depth, k, alpha_true=float(rng.uniform(0, 0.5)), seed=int(rng.integers(1, 10**6)), q_lo=qlo, q_hi=qhi,
                             missing_frac=float(rng.choice([0.0, 0.0, 0.3])))
    # quality profiles: as drawn (uniform), binned to a few values (deep runs: many steps per window of the probability
    # domain's dictionary), or one dominant quality
    prof = int(rng.integers(0, 3))
    if prof == 1:
        bins = np.sort(rng.choice(np.arange(qlo, qhi + 1), size=min(qhi - qlo + 1, int(rng.integers(1, 6))), replace=False)).astype(np.uint8)
        d.quals[:] = bins[(d.quals - 33) % len(bins)] + 33
    elif prof == 2:
        d.quals[rng.random(d.quals.size) < rng.uniform(0.5, 0.95)] = int(rng.integers(qlo, qhi + 1)) + 33
    od = oracle_data(d)
    B = int(rng.choice([rng.integers(1, 20), rng.integers(20, 60)]))                         # (beyond 24 points: split launches)
    scale = float(rng.choice([0.01, 0.05, 0.5]))
    pc1 = rng.normal(0, scale, size=(B, k)); pc2 = rng.normal(0, scale, size=(B, k)); al = rng.uniform(0, 1, size=B)
    al[0] = float(rng.choice([al[0], 0.0, 1.0, 1e-300]))
    with vb.LikelihoodContext(d) as ctx:
        got = ctx.llk(pc1, pc2, al)
        layout = ctx.info()["layout"]
    want = np.array([od.llk(pc1[i], pc2[i], al[i], num_thread=8) for i in range(B)])
    err = np.abs(got - want) / np.maximum(1.0, np.abs(want))
    if float(err.max()) > worst:
        worst_case = "it=%d M=%d depth=%g k=%d q=%d..%d B=%d profile %d layout %d alpha[argmax] %.3g" % (it, M, depth, k, qlo, qhi, B, prof, layout, al[int(err.argmax())])
    worst = max(worst, float(err.max()))
    if err.max() > 1e-11:
        bad += 1
        print("MISMATCH it=%d M=%d depth=%g k=%d q=%d..%d B=%d layout %d: max abs diff %.3g (rel %.2e)" % (it, M, depth, k, qlo, qhi, B, layout, np.abs(got - want).max(), err.max()))
    nlay = locals().get("nlay", [0, 0]); nlay[layout] += 1
print("worst case:", locals().get("worst_case"))
print("llk fuzz: %d of %d cases off, worst relative error %.2e; layouts: %d run words, %d probability domain" % (bad, N, worst, nlay[0], nlay[1]))
