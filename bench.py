#!/usr/bin/env python
"""bench.py -- LLK evaluations per second on the BASELINE.json workload.

    python bench.py --gpus N --steps K --warmup W
    (N > 1:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (config.workload): synthetic pileup of 100 000 markers x depth 30, --NumPC 4
(BASELINE.json configs[2], the shape the metric is quoted on), resident in HBM before the
timed region.  One STEP = one pass of the hot path over that pileup for a batch of
`--batch` parameter points (default 48, the most one launch carries): one
`vb2_llk_eval_batch_device` call = one launch of the dominant kernel
(llk_eval_kernel<2,true>: 8 points per group, 4 groups).  One "eval" = one
(pc1, pc2, alpha) point = one call of the reference's ComputeMixLLKs.

N > 1 (default `--mode sample`): every rank owns a different sample of the same shape
(sample-parallel, BASELINE.json configs[4]); no data-path collective, weak scaling.
`--mode marker` shards one sample's markers over the ranks and all-reduces the partial
LLKs over RCCL each step (configs[3]; strong scaling, latency-bound -- reported in
DESIGN.md, not the default).

Prints ONE JSON line on rank 0.  The oracle (oracle/) is used only as the checker of a
small parity probe and as the cpu_baseline leg; it is never the thing timed as `value`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--prewarm-ms", type=float, default=100.0,
                    help="keep the GPU busy this long before the warmup steps: an idle MI355X needs ~40 ms of "
                         "load to reach its sustained clocks (tools/warm_curve.py: 113 -> 87 us per launch)")
    ap.add_argument("--batch", type=int, default=48, help="parameter points per step (one launch carries up to 48)")
    ap.add_argument("--markers", type=int, default=100000)
    ap.add_argument("--depth", type=float, default=30.0)
    ap.add_argument("--num-pc", type=int, default=4)
    ap.add_argument("--mode", choices=["sample", "marker"], default="sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-optimize", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    import verifybamid_amd as vb

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with %d ranks" % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    k, B = args.num_pc, args.batch
    # ---- inputs (synthetic, seeded), flattened into HBM before timing ----
    if args.mode == "sample":
        data = vb.synth.make_pileup(args.markers, args.depth, k, alpha_true=0.05, seed=2 + rank)
        shard = data
    else:
        data = vb.synth.make_pileup(args.markers, args.depth, k, alpha_true=0.05, seed=2)
        shard = data.shard(rank, world)
    # an explicit (non-null) stream: the kernels, the HIP events and the RCCL collective
    # all go on it, so the events bracket exactly the timed launches
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    ctx = vb.LikelihoodContext(shard, device=local_rank, stream=stream.cuda_stream)
    info = ctx.info()

    rng = np.random.default_rng(123)
    stride = 2 * k + 1
    pts_h = np.concatenate([rng.normal(0, 0.03, size=(B, 2 * k)), rng.uniform(0.01, 0.3, size=(B, 1))], axis=1)
    pts = torch.tensor(pts_h, dtype=torch.float64, device="cuda")
    out = torch.zeros(B, dtype=torch.float64, device="cuda")
    assert pts.is_contiguous() and pts.shape == (B, stride)

    def step():
        ctx.llk_device(pts.data_ptr(), out.data_ptr(), B, stream.cuda_stream)
        if dist is not None and args.mode == "marker":
            dist.all_reduce(out)      # sum of per-shard partial LLKs over RCCL

    # untimed: bring the clocks up (same work as a step), then the W warmup steps
    t_pw = time.perf_counter()
    while 1e3 * (time.perf_counter() - t_pw) < args.prewarm_ms:
        for _ in range(50):
            step()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(stream)
    for _ in range(args.steps):
        step()
    ev1.record(stream)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)
    if dist is not None:
        t = torch.tensor([wall, dev_ms], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall, dev_ms = float(t[0]), float(t[1])

    evals_per_step = B * (world if args.mode == "sample" else 1)
    value = evals_per_step * args.steps / wall
    llk_dev = out.cpu().numpy().copy()

    result = {
        "metric": "llk_evals_per_sec", "value": value, "unit": "evals/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * wall / args.steps,
        "higher_is_better": True, "scaling": "weak" if args.mode == "sample" else "strong",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {
            "workload": "synthetic pileup %d markers x depth %g, --NumPC %d (BASELINE.json configs[2] shape)"
                        % (args.markers, args.depth, k),
            "batch_points_per_step": B, "prewarm_ms": args.prewarm_ms,
            "parallelism": ("1 GPU" if world == 1 else
                            ("sample-parallel x%d (one sample per GPU, no collective)" % world
                             if args.mode == "sample" else
                             "marker-sharded x%d + RCCL all-reduce of %d doubles per step" % (world, B))),
            "reads": int(info["num_read"]), "active_markers": int(info["num_active_marker"]),
            "distinct_codes": int(info["num_code"]), "device": info["arch"],
        },
    }

    if rank == 0:
        # roofline of the dominant kernel: algorithmic bytes per launch / device time
        bytes_per_launch = info["algorithmic_bytes_per_eval"] * B
        step_us = 1e3 * dev_ms / args.steps
        achieved = bytes_per_launch / (step_us * 1e-6) / 1e9
        # HBM-side bytes per launch cannot be read live (PMC needs rocprofv3): take the value
        # measured by the committed PMC passes of this same command when the shape matches
        traffic, traffic_src = None, None
        for rnd in sorted(os.listdir(os.path.join(ROOT, "profiles")), reverse=True) \
                if os.path.isdir(os.path.join(ROOT, "profiles")) else []:
            f = os.path.join(ROOT, "profiles", rnd, "traffic_b%d.json" % B)
            if os.path.exists(f):
                tj = json.load(open(f))
                if tj.get("markers") == args.markers and tj.get("num_pc") == k:
                    traffic, traffic_src = tj["traffic_bytes_per_launch"], os.path.relpath(f, ROOT)
                    break
        result["roofline"] = {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_src,
            "kernel": "llk_eval_kernel<%d,true>" % (2 if B > 4 else 3),
            "launches_per_step": (B + 47) // 48,
            "algorithmic_bytes_per_launch": int(bytes_per_launch),
            "device_us_per_launch": step_us,
            "note": "device time = HIP events on the launch stream over the timed region / steps; "
                    "the pileup is L2/MALL resident and the kernel is latency/VALU-bound, see DESIGN.md",
        }
        # small parity probe against the oracle (checker only)
        from oracle.bridge import oracle_data
        od = oracle_data(data)
        if args.mode == "sample" or world == 1:
            want = np.array([od.llk(pts_h[i, :k], pts_h[i, k:2 * k], pts_h[i, 2 * k],
                                    num_thread=os.cpu_count() or 1) for i in range(min(B, 2))])
            rel = float(np.max(np.abs(llk_dev[:len(want)] - want) / np.abs(want)))
            result["parity_probe_max_rel_err"] = rel
        if world == 1 and not args.no_optimize:
            # second half of the metric: wall-clock of OptimizeLLK (Initialize + Homo + Heter +
            # LLK0), best of 3; measured before the CPU leg so no OpenMP threads are around
            ctx.optimize()
            t_opt = []
            for _ in range(3):
                t1 = time.perf_counter()
                est = ctx.optimize()
                t_opt.append(time.perf_counter() - t1)
            result["optimize"] = {
                "wall_ms_to_converged_alpha": 1e3 * min(t_opt),
                "alpha": est["alpha"], "alpha_true": 0.05, "num_eval": est["num_eval"],
                "num_launch_point": est["num_launch_point"],
            }
        if world == 1 and not args.no_cpu_baseline:
            # bounded sample (~10 s wall): the C oracle on the SAME pileup, OpenMP over
            # markers like the reference; thread counts 1, 4 (the reference's default
            # --NumThread) and powers of two up to the cores this process may use; the
            # best one is reported.
            try:
                navail = len(os.sched_getaffinity(0))
            except AttributeError:
                navail = os.cpu_count() or 1
            sweep = sorted({1, 4} | {t for t in (8, 16, 32, 64, 128, 256) if t <= navail})
            rates = {}
            for nt in sweep:
                n_cpu, tc = 0, time.perf_counter()
                while time.perf_counter() - tc < 10.0 / len(sweep) or n_cpu < 2:
                    od.llk(pts_h[n_cpu % B, :k], pts_h[n_cpu % B, k:2 * k], pts_h[n_cpu % B, 2 * k],
                           num_thread=nt)
                    n_cpu += 1
                rates[nt] = n_cpu / (time.perf_counter() - tc)
            best = max(rates, key=rates.get)
            result["cpu_baseline"] = {
                "value": rates[best], "unit": "evals/s", "cores": best, "kind": "port",
                "sample": "C oracle (oracle/vb2_oracle.c, OpenMP over markers like the reference) on the "
                          "same %d-marker pileup, ~%.1f s per thread count; evals/s by threads: %s; "
                          "%d cores available" % (args.markers, 10.0 / len(sweep),
                                                  {t: round(r, 1) for t, r in rates.items()}, navail),
            }
        print(json.dumps(result))
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
