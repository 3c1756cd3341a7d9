"""Experiment (round 4, second session): the 8-point shape as TWO workgroups per CU, each with three of a 48-point call's six
point groups -- so each builds only its own three tables and 80 KB of LDS hold them -- at 10 waves and <= 96 registers
(5 waves per SIMD instead of 4; build: tools/build_variant.sh g2w10 -DVB2_G2_WAVES=10 -DVB2_G2_WPS=5).  Emulated with two
contexts of the same sample launching 24 points each on two streams, against one context launching 48."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import verifybamid_amd as vb
k = 4
d = vb.synth.make_pileup(100000, 30, k, 0.05, 2)
rng = np.random.default_rng(3)
B = 48
pts = np.concatenate([rng.normal(0, 0.03, (B, k)), rng.normal(0, 0.03, (B, k)), rng.uniform(0.01, 0.4, (B, 1))], axis=1)
def run(ctxs, streams, pt_list, reps=1500):
    outs = [torch.zeros(p.shape[0], dtype=torch.float64, device="cuda") for p in pt_list]
    for _ in range(300):
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
tagname = os.environ.get("VB2_PAIR_TAG", mode)
if mode == "single48":
    c = vb.LikelihoodContext(d, device=0, stream=s0.cuda_stream)
    dt, o = run([c], [s0], [tp])
    print("%s: one context, 48 points per launch: %.2f us per 48 points" % (tagname, 1e6 * dt)); ref = o[0]
else:
    a = vb.LikelihoodContext(d, device=0, stream=s0.cuda_stream)
    b = vb.LikelihoodContext(d, device=0, stream=s1.cuda_stream)
    dt1, o1 = run([a], [s0], [tp[:24].contiguous()])
    dt, o = run([a, b], [s0, s1], [tp[:24].contiguous(), tp[24:].contiguous()])
    print("%s (GEOM2=%s): one context 24 points alone %.2f us; two contexts x 24 points concurrently: %.2f us per 48 points"
          % (tagname, os.environ.get("VB2_GEOM2"), 1e6 * dt1, 1e6 * dt))
    ref = np.concatenate(o)
np.save("/tmp/two_wg_%s.npy" % tagname, ref)
if os.path.exists("/tmp/two_wg_base.npy") and tagname != "base":
    base = np.load("/tmp/two_wg_base.npy")
    print("   max |diff| vs the 48-point launch: %.3e  (bit-equal: %s)" % (np.abs(base - ref).max(), bool((base == ref).all())))
