"""Data dependence of the kernel: the work per marker is its number of distinct (class, quality)
pairs (run-length coding).  Same 100k x 30 sample with 21 / 8 / 4 / 1 distinct quality values."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import verifybamid_amd as vb
rng = np.random.default_rng(1)
B = 48
for (lo, hi, label) in ((2, 60, "59 values (BAQ-like wide alphabet, narrow table rows)"), (20, 40, "21 values (uniform 20..40, the bench workload)"), (30, 37, "8 values"),
                        (34, 37, "4 values (binned-quality instruments)"), (37, 37, "1 value")):
    d = vb.synth.make_pileup(100000, 30, 4, 0.05, 2, q_lo=lo, q_hi=hi)
    stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
    ctx = vb.LikelihoodContext(d, device=0, stream=stream.cuda_stream)
    info = ctx.info()
    pts = torch.tensor(np.concatenate([rng.normal(0, 0.03, size=(B, 8)), rng.uniform(0.01, 0.3, size=(B, 1))], axis=1), device="cuda")
    out = torch.zeros(B, dtype=torch.float64, device="cuda")
    for _ in range(1500): ctx.llk_device(pts.data_ptr(), out.data_ptr(), B, stream.cuda_stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(1000): ctx.llk_device(pts.data_ptr(), out.data_ptr(), B, stream.cuda_stream)
    e1.record(stream); torch.cuda.synchronize()
    us = e0.elapsed_time(e1)
    print("%-48s codes %3d: %6.1f us per 48-point launch = %4.0f k evals/s" % (label, info["num_code"], us, B / us * 1e3))
    ctx.close()
