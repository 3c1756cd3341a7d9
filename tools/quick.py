"""Kernel-iteration check on the GPU box: parity vs the oracle on the C3-shaped sample (a few
points), then device time per launch at several batch sizes (HIP events, sustained clocks) and
the OptimizeLLK wall-clock.  One line of JSON per measurement.

    python tools/quick.py [--markers 100000] [--batches 48,4,1] [--codes wide]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import verifybamid_amd as vb
from oracle.bridge import oracle_data

ap = argparse.ArgumentParser()
ap.add_argument("--markers", type=int, default=100000)
ap.add_argument("--num-pc", type=int, default=4)
ap.add_argument("--batches", default="48,32,8,4,1")
ap.add_argument("--q-lo", type=int, default=20)
ap.add_argument("--q-hi", type=int, default=40)
ap.add_argument("--steps", type=int, default=1500)
ap.add_argument("--no-parity", action="store_true")
ap.add_argument("--no-optimize", action="store_true")
a = ap.parse_args()

k = a.num_pc
d = vb.synth.make_pileup(a.markers, 30, k, alpha_true=0.05, seed=2, q_lo=a.q_lo, q_hi=a.q_hi)
stream = torch.cuda.Stream()
torch.cuda.set_stream(stream)
ctx = vb.LikelihoodContext(d, device=0, stream=stream.cuda_stream)
info = ctx.info()
rng = np.random.default_rng(123)
B0 = 48
pts_h = np.concatenate([rng.normal(0, 0.03, size=(B0, 2 * k)), rng.uniform(0.01, 0.3, size=(B0, 1))], axis=1)
out = {"markers": a.markers, "codes": int(info["num_code"]), "reads": int(info["num_read"])}
if not a.no_parity:
    od = oracle_data(d)
    got = ctx.llk(pts_h[:, :k], pts_h[:, k:2 * k], pts_h[:, 2 * k])
    nchk = 6
    want = np.array([od.llk(pts_h[i, :k], pts_h[i, k:2 * k], pts_h[i, 2 * k], num_thread=os.cpu_count() or 1)
                     for i in range(nchk)])
    out["parity_max_rel"] = float(np.max(np.abs(got[:nchk] - want) / np.abs(want)))
    g1 = ctx.llk(pts_h[:1, :k], pts_h[:1, k:2 * k], pts_h[:1, 2 * k])
    g4 = ctx.llk(pts_h[:4, :k], pts_h[:4, k:2 * k], pts_h[:4, 2 * k])
    out["slot_independent"] = bool(g1[0] == got[0] and np.array_equal(g4, got[:4]))
print(json.dumps(out), flush=True)

pts = torch.tensor(pts_h, dtype=torch.float64, device="cuda")
res = torch.zeros(B0, dtype=torch.float64, device="cuda")
t_pw = time.perf_counter()
while time.perf_counter() - t_pw < 0.15:
    for _ in range(50):
        ctx.llk_device(pts.data_ptr(), res.data_ptr(), B0, stream.cuda_stream)
    torch.cuda.synchronize()
for B in [int(x) for x in a.batches.split(",")]:
    for _ in range(100):
        ctx.llk_device(pts.data_ptr(), res.data_ptr(), B, stream.cuda_stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(a.steps):
        ctx.llk_device(pts.data_ptr(), res.data_ptr(), B, stream.cuda_stream)
    e1.record(stream)
    torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / a.steps
    print(json.dumps({"batch": B, "us_per_launch": round(us, 2), "us_per_eval": round(us / B, 3),
                      "evals_per_s": round(B / us * 1e6)}), flush=True)
if not a.no_optimize:
    ctx.optimize()
    ts = []
    for _ in range(5):
        t1 = time.perf_counter()
        est = ctx.optimize()
        ts.append(time.perf_counter() - t1)
    print(json.dumps({"optimize_ms": round(1e3 * min(ts), 3), "alpha": est["alpha"], "num_eval": est["num_eval"],
                      "num_launch_point": est["num_launch_point"]}), flush=True)
ctx.close()
