"""vb2_cohort_run on 256 C3-shaped text pileups (8 distinct, like bench.py's from_text leg), repeated, with the flatten
on the device and on the host -- same box, same files.   VB2_CPUS=8 python tools/cohort_from_text.py [files] [repeats]"""
import os, sys, time, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import verifybamid_amd as vb
from verifybamid_amd import _abi
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rep = int(sys.argv[2]) if len(sys.argv) > 2 else 3
M, k, depth = 100000, 4, 30
tmp = tempfile.mkdtemp(prefix="vb2_ft_")
try:
    data = vb.synth.make_pileup(M, depth, k, alpha_true=0.05, seed=2)
    base = vb.synth.with_sanity_stats(data)
    pre = vb.synth.write_files(base, os.path.join(tmp, "panel"))
    piles = []
    for s_ in range(8):
        off_, ch_, qu_ = vb.synth.reads_on_panel(base.means, base.meta["ref_base"], base.alt_base, depth,
                                                 alpha_true=0.01 * (1 + s_), seed=2000 + s_)
        dep_ = np.diff(off_).astype(np.float64)
        dd = vb.PileupData(k, base.ud, base.means, off_, ch_, qu_, base.alt_base, None, float(dep_[dep_ > 0].mean()), 0.0,
                           True, dict(base.meta))
        piles.append(vb.synth.write_files(dd, os.path.join(tmp, "s%d" % s_)) + ".pileup")
    paths = [piles[i % 8] for i in range(nf)]
    outs = [os.path.join(tmp, "out%d" % i) for i in range(nf)]
    devnull = open(os.devnull, "w")
    saved = os.dup(2)
    group = int(os.environ.get("VB2_GROUP", "0"))
    readers = int(os.environ.get("VB2_READERS", "0"))
    for mode in os.environ.get("VB2_MODES", "device,host,device,host").split(","):
        _abi.set_tunable("host_flatten", 0 if mode == "device" else 1)
        rates = []
        for _ in range(rep):
            if os.environ.get('VB2_DEBUG_LOCKSTEP', '') != '1': os.dup2(devnull.fileno(), 2)
            t1 = time.perf_counter(); p1 = time.process_time()
            res = vb.run_cohort_files(pre, paths, outs, num_pc=k, group_size=group, num_host_thread=readers)
            dt = time.perf_counter() - t1; cpu = time.process_time() - p1
            os.dup2(saved, 2)
            assert all(r["status"] == 0 for r in res)
            rates.append((nf / dt, 1e3 * cpu / nf))
        print("flatten on the %-6s: %s samples/s; process CPU per sample %s ms; alpha[0] %.6f"
              % (mode, " ".join("%.0f" % r[0] for r in rates), " ".join("%.2f" % r[1] for r in rates), res[0]["alpha"]), flush=True)
finally:
    shutil.rmtree(tmp, ignore_errors=True)
