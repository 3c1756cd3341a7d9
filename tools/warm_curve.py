"""Per-launch device time of the 48-point launch as a function of time since the GPU went busy."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import verifybamid_amd as vb
d = vb.synth.make_pileup(100000, 30, 4, 0.05, 2)
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
ctx = vb.LikelihoodContext(d, device=0, stream=stream.cuda_stream)
rng = np.random.default_rng(1)
B = 48
pts = torch.tensor(np.concatenate([rng.normal(0, 0.03, size=(B, 8)), rng.uniform(0.01, 0.3, size=(B, 1))], axis=1), device="cuda")
out = torch.zeros(B, dtype=torch.float64, device="cuda")
torch.cuda.synchronize(); time.sleep(float(os.environ.get("IDLE", "2")))
evs = [torch.cuda.Event(enable_timing=True) for _ in range(61)]
evs[0].record(stream)
for i in range(60):
    for _ in range(100):
        ctx.llk_device(pts.data_ptr(), out.data_ptr(), B, stream.cuda_stream)
    evs[i + 1].record(stream)
torch.cuda.synchronize()
t = 0.0
for i in range(60):
    dt = evs[i].elapsed_time(evs[i + 1])
    t += dt
    if i < 12 or i % 6 == 5:
        print("launches %4d-%4d  (t=%6.1f ms): %.2f us per launch" % (100 * i, 100 * i + 99, t, 10 * dt))
