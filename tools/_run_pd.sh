#!/bin/bash
# same-box A/B of the probability-domain timing emulation (round 6): headline, alphabets, 4-point search rounds, cohort steps
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for rep in 1 2; do
  for v in $VARIANTS; do
    echo "== $v"
    export VB2_LIB_PATH=$PWD/build_variants/$v/libvb2.so
    python bench.py --steps 1500 --warmup 200 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('  headline %.1f k evals/s  %.2f us' % (d['value']/1e3, d['ms_per_step']*1e3))"
    python tools/quality_profile_time.py 2>&1 | grep "codes" | head -2
    python - <<'PY'
import sys, time, numpy as np
sys.path.insert(0, ".")
import verifybamid_amd as vb
d = vb.synth.make_pileup(100000, 30, 4, 0.05, 2)
rng = np.random.default_rng(3)
with vb.LikelihoodContext(d) as ctx:
    for n in (4, 1):
        pts = [(rng.normal(0, 0.03, size=(n, 4)), rng.normal(0, 0.03, size=(n, 4)), rng.uniform(0.01, 0.3, size=n)) for _ in range(1500)]
        with ctx.search():
            for p in pts[:300]: ctx.llk(*p)
            t0 = time.perf_counter()
            for p in pts: ctx.llk(*p)
            dt = time.perf_counter() - t0
        print("  %d points per call in the search bracket: %.2f us" % (n, 1e6 * dt / len(pts)))
PY
    VB2_STEPS_ONLY=1 python tools/cohort_steps.py 2>&1 | grep samples
  done
done
