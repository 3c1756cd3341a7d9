"""Soak of the streaming cohort pipeline (vb2_cohort_run): random sample sizes, slot counts, reader counts and device lists
(one GPU listed several times: VB2_COHORT_DUP_DEVICES), one unreadable and one insane sample per run; every estimate against
the sample's own vb2_run, runs with the same slot count against each other bit for bit."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import verifybamid_amd as vb

rng = np.random.default_rng(int(os.environ.get("VB2_SEED", 1)))
tmp = tempfile.mkdtemp()
os.environ["VB2_COHORT_DUP_DEVICES"] = "1"
rounds = int(os.environ.get("VB2_ROUNDS", 6))
for rd in range(rounds):
    k = int(rng.choice([2, 3, 4]))
    M = int(rng.choice([int(x) for x in os.environ.get("VB2_MSET", "1200,4000,9000").split(",")]))
    base = vb.synth.with_sanity_stats(vb.synth.make_pileup(M, 14, k, alpha_true=0.03, seed=1000 + rd))
    pre = vb.synth.write_files(base, os.path.join(tmp, "panel%d" % rd))
    piles = []
    for s in range(7):
        d = vb.synth.make_pileup(M, float(rng.choice([5, 12, 25, 40])), k, alpha_true=float(rng.choice([0.0, 0.02, 0.1, 0.3])), seed=2000 + 10 * rd + s,
                                 missing_frac=float(rng.choice([0.0, 0.3])))
        d = vb.PileupData(k, base.ud, base.means, d.read_off, d.bases, d.quals, base.alt_base, None, d.avg_depth, d.sd_depth, True, dict(base.meta))
        piles.append(vb.synth.write_files(d, os.path.join(tmp, "r%ds%d" % (rd, s))) + ".pileup")
    S = int(rng.integers(20, int(os.environ.get("VB2_SMAX", 140))))
    paths = [piles[int(rng.integers(0, 7))] for _ in range(S)]
    bad = int(rng.integers(0, S))
    paths[bad] = os.path.join(tmp, "nope.pileup")
    singles = {}
    for pth in set(paths):
        if pth.endswith("nope.pileup"):
            continue
        try:
            singles[pth] = vb.run_files(pre, pth, os.path.join(tmp, "single"), num_pc=k)
        except Exception as e:                      # (a sample that fails the sanity check: the cohort must report it too)
            singles[pth] = None
    slots = int(rng.choice([3, 7, 16, 17, 32, 64]))
    runs = []
    for rep in range(3):
        thr = int(rng.integers(1, 12))
        devs = None if rep == 0 else [0] * int(rng.integers(1, 4))
        outs = [os.path.join(tmp, "o%d_%d_%d" % (rd, rep, s)) for s in range(S)]
        t0 = time.perf_counter()
        res = vb.run_cohort_files(pre, paths, outs, num_pc=k, group_size=slots, num_host_thread=thr, devices=devs)
        dt = time.perf_counter() - t0
        runs.append((res, outs))
        worst = 0.0
        for s in range(S):
            if s == bad:
                assert res[s]["status"] != 0
                continue
            one = singles[paths[s]]
            if one is None:
                assert res[s]["status"] != 0, (rd, rep, s)
                continue
            assert res[s]["status"] == 0, (rd, rep, s, res[s]["status"])
            worst = max(worst, abs(res[s]["alpha"] - one["alpha"]))
            assert abs(res[s]["alpha"] - one["alpha"]) <= 1e-6, (rd, rep, s, res[s]["alpha"], one["alpha"])
        print("round %d rep %d: k %d, %d markers, %d samples, %d slots, %d readers, devices %s: %.2f s, max |alpha - single| %.2e"
              % (rd, rep, k, M, S, slots, thr, devs, dt, worst), flush=True)
    # same slot count and device count -> same bits (reps 0 has one device; compare reps with equal device counts only)
    a, b = runs[1], runs[2]
    for s in range(S):
        if s != bad and paths[s] in singles:
            pass
    r0 = runs[0][0]
    again = vb.run_cohort_files(pre, paths, [os.path.join(tmp, "z%d_%d" % (rd, s)) for s in range(S)], num_pc=k, group_size=slots,
                                num_host_thread=int(rng.integers(1, 12)))
    for s in range(S):
        if s != bad and singles[paths[s]] is not None:
            assert again[s]["alpha"] == r0[s]["alpha"] and again[s]["num_eval"] == r0[s]["num_eval"], (rd, s)
print("soak ok")
