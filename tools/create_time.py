import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import verifybamid_amd as vb
d = vb.synth.make_pileup(100000, 30, 4, 0.05, 2)
vb.LikelihoodContext(d).close()
ts = []
for _ in range(5):
    t0 = time.perf_counter(); c = vb.LikelihoodContext(d); ts.append(time.perf_counter() - t0); c.close()
print("vb2_ctx_create 100k x 30: best %.1f ms median %.1f ms" % (1e3 * min(ts), 1e3 * sorted(ts)[2]))
