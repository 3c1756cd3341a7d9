"""Per-launch device time: one API call per 48-point launch vs several launches inside one call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import verifybamid_amd as vb
d = vb.synth.make_pileup(100000, 30, 4, 0.05, 2)
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
ctx = vb.LikelihoodContext(d, device=0, stream=stream.cuda_stream)
rng = np.random.default_rng(1)
def run(B, calls, steps=100):
    pts = torch.tensor(np.concatenate([rng.normal(0, 0.03, size=(B, 8)), rng.uniform(0.01, 0.3, size=(B, 1))], axis=1), device="cuda")
    out = torch.zeros(B, dtype=torch.float64, device="cuda")
    per = B // calls
    def step():
        for c in range(calls):
            ctx.llk_device(pts[c * per:].data_ptr(), out[c * per:].data_ptr(), per, stream.cuda_stream)
    for _ in range(10): step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(steps): step()
    e1.record(stream); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / steps / (B / 48)
for B, calls in ((48, 1), (96, 2), (96, 1), (480, 10), (480, 1), (48, 1)):
    print("B=%d in %d call(s): %.2f us per 48-point launch" % (B, calls, run(B, calls)))
