"""End-to-end wall-clock of the command line on a C3-sized sample (files on local disk):
file parsing + flatten + search + writers, next to the search alone."""
import os, sys, time, subprocess, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import verifybamid_amd as vb
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
M = int(os.environ.get("VB2_M", 100000))
d = vb.synth.with_sanity_stats(vb.synth.make_pileup(M, 30, 4, 0.05, 2))
tmp = tempfile.mkdtemp()
prefix = vb.synth.write_files(d, os.path.join(tmp, "c3"))
exe = os.path.join(ROOT, "verifybamid_amd", "bin", "VerifyBamID")
cmd = [exe, "--SVDPrefix", prefix, "--PileupFile", prefix + ".pileup", "--Reference", "x.fa", "--NumPC", "4",
       "--Output", os.path.join(tmp, "out")]
env = dict(os.environ, VB2_DEBUG_TIMING="1")
for rep in range(3):
    t0 = time.perf_counter()
    r = subprocess.run(cmd, capture_output=True, text=True, env=env)
    dt = time.perf_counter() - t0
    print("rep %d: rc=%d wall %.1f ms" % (rep, r.returncode, 1e3 * dt))
print(r.stderr[-1500:])
print(r.stdout[-400:])
print("file sizes: pileup %.1f MB, UD %.1f MB" % (os.path.getsize(prefix + ".pileup") / 1e6, os.path.getsize(prefix + ".UD") / 1e6))
