"""Round 6: layout a sample takes, parity of a batch against the oracle, launch times (48 / 4 / 1 points)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import verifybamid_amd as vb
from oracle.bridge import oracle_data
rng = np.random.default_rng(1)
for (M, depth, k, lo, hi) in [(10000, 30, 2, 20, 40), (100000, 30, 4, 20, 40), (100000, 30, 4, 10, 45), (100000, 30, 4, 2, 60), (20000, 60, 4, 20, 40)]:
    d = vb.synth.make_pileup(M, depth, k, 0.05, 2, q_lo=lo, q_hi=hi)
    stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
    ctx = vb.LikelihoodContext(d, device=0, stream=stream.cuda_stream)
    info = ctx.info()
    B = 48
    pc1 = rng.normal(0, 0.03, size=(B, k)); pc2 = rng.normal(0, 0.03, size=(B, k)); al = rng.uniform(0.01, 0.3, size=B)
    got = ctx.llk(pc1, pc2, al)
    od = oracle_data(d)
    want = np.array([od.llk(pc1[i], pc2[i], al[i], num_thread=8) for i in range(0, B, 7)])
    err = np.max(np.abs(got[::7] - want) / np.abs(want))
    one = np.array([ctx.llk(pc1[i:i+1], pc2[i:i+1], al[i:i+1])[0] for i in range(0, B, 7)])
    four = ctx.llk(pc1[:4], pc2[:4], al[:4])
    pts = torch.tensor(np.concatenate([pc1, pc2, al[:, None]], axis=1), device="cuda")
    out = torch.zeros(B, dtype=torch.float64, device="cuda")
    for _ in range(300): ctx.llk_device(pts.data_ptr(), out.data_ptr(), B, stream.cuda_stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(500): ctx.llk_device(pts.data_ptr(), out.data_ptr(), B, stream.cuda_stream)
    e1.record(stream); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 500 * 1e3
    t0 = time.perf_counter(); r = ctx.optimize(); r = ctx.optimize(); dt = (time.perf_counter() - t0) / 2
    print("M=%d depth=%d q%d..%d: layout %d rows %d codes %d tiles %d | rel err %.2e | shapes equal %s %s | 48-pt %.1f us = %.0f k evals/s | optimize %.2f ms (%d evals) alpha %.6f"
          % (M, depth, lo, hi, info["layout"], info["num_table_row"], info["num_code"], info["num_tile"], err,
             bool(np.array_equal(one, got[::7])), bool(np.array_equal(four, got[:4])), us, B / us * 1e3, dt * 1e3, r["num_eval"], r["alpha"]), flush=True)
    ctx.close()
