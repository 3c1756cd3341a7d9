"""File-level cohort throughput (vb2_cohort_run): S C3-sized pileups against one panel, panel read
once, pileups read/flattened by host threads while the device searches the previous group."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import verifybamid_amd as vb
S = int(os.environ.get("VB2_S", 32)); M = int(os.environ.get("VB2_M", 100000)); k = 4
tmp = tempfile.mkdtemp()
base = vb.synth.with_sanity_stats(vb.synth.make_pileup(M, 30, k, 0.05, 2))
pre = vb.synth.write_files(base, os.path.join(tmp, "panel"))
t0 = time.perf_counter()
piles = []
DISTINCT = int(os.environ.get("VB2_DISTINCT", S))      # fewer distinct files than samples: reuse them cyclically
for s in range(min(S, DISTINCT)):
    d = vb.synth.make_pileup(M, 30, k, alpha_true=0.01 * (1 + s % 20), seed=1000 + s)
    d = vb.PileupData(k, base.ud, base.means, d.read_off, d.bases, d.quals, base.alt_base, None,
                      d.avg_depth, d.sd_depth, True, dict(base.meta))
    piles.append(vb.synth.write_files(d, os.path.join(tmp, "s%d" % s)) + ".pileup")
print("wrote %d pileups in %.1f s" % (len(piles), time.perf_counter() - t0))
piles = [piles[s % len(piles)] for s in range(S)]
outs = [os.path.join(tmp, "out%d" % s) for s in range(S)]
t0 = time.perf_counter(); one = vb.run_files(pre, piles[0], os.path.join(tmp, "single"), num_pc=k); t_one = time.perf_counter() - t0
for threads in [int(x) for x in os.environ.get("VB2_THREADS", "4,16").split(",")]:
    for group in [int(x) for x in os.environ.get("VB2_GROUPS", "8,32").split(",")]:
        t0 = time.perf_counter()
        res = vb.run_cohort_files(pre, piles, outs, num_pc=k, group_size=group, num_host_thread=threads)
        dt = time.perf_counter() - t0
        assert all(r["status"] == 0 for r in res)
        print("host threads %2d, group %2d: %d samples in %.2f s = %.1f ms/sample (%.1f samples/s); one sample alone (vb2_run): %.0f ms"
              % (threads, group, S, dt, 1e3 * dt / S, S / dt, 1e3 * t_one))
print("alpha[0] cohort %.7f single %.7f" % (res[0]["alpha"], one["alpha"]))
