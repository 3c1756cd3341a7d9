"""Experiment: the one-point-per-lane wave shape held to 64 registers (8 waves per SIMD when TWO
workgroups share a CU).  Two contexts of the same sample, each evaluating 24 points (6 groups of 4) per launch on its
own stream, against one context evaluating 48 points (6 groups of 8, two points per lane) per launch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import verifybamid_amd as vb
k = 4
d = vb.synth.make_pileup(100000, 30, k, 0.05, 2)
rng = np.random.default_rng(3)
B = 48
pts = np.concatenate([rng.normal(0, 0.03, (B, k)), rng.normal(0, 0.03, (B, k)), rng.uniform(0.01, 0.4, (B, 1))], axis=1)
def run(ctxs, streams, pt_list, reps=400):
    outs = [torch.zeros(p.shape[0], dtype=torch.float64, device="cuda") for p in pt_list]
    for _ in range(30):
        for c, s, p, o in zip(ctxs, streams, pt_list, outs):
            c.llk_device(p.data_ptr(), o.data_ptr(), p.shape[0], s.cuda_stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        for c, s, p, o in zip(ctxs, streams, pt_list, outs):
            c.llk_device(p.data_ptr(), o.data_ptr(), p.shape[0], s.cuda_stream)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps, [o.cpu().numpy() for o in outs]
tp = torch.tensor(pts, device="cuda")
s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
mode = os.environ.get("VB2_PAIR_MODE", "pair")
if mode == "single48":
    c = vb.LikelihoodContext(d, device=0, stream=s0.cuda_stream)
    dt, o = run([c], [s0], [tp])
    print("one context, 48 points per launch: %.2f us per 48 points" % (1e6 * dt)); ref = o[0]
else:
    a = vb.LikelihoodContext(d, device=0, stream=s0.cuda_stream)
    b = vb.LikelihoodContext(d, device=0, stream=s1.cuda_stream)
    dt1, o1 = run([a], [s0], [tp[:24].contiguous()])
    dt, o = run([a, b], [s0, s1], [tp[:24].contiguous(), tp[24:].contiguous()])
    print("row_bytes narrow=%s one_point=%s: one context 24 points alone %.2f us; two contexts x 24 points concurrently: %.2f us per 48 points"
          % (os.environ.get("VB2_FORCE_NARROW"), os.environ.get("VB2_ONE_POINT"), 1e6 * dt1, 1e6 * dt))
    ref = np.concatenate(o)
np.save("/tmp/one_point_%s.npy" % mode, ref)
