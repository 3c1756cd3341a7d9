"""Multi-GPU drivers: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over
xGMI on ROCm; "gloo" for the CPU tests).  SURVEY.md 8e / DESIGN.md 5.

* marker sharding: LLK = sum over markers, so every rank evaluates its contiguous,
  read-balanced marker range and ONE all-reduce of `num_point` doubles finishes an
  evaluation batch; the (host) optimiser runs identically on every rank.
* sample parallelism: samples are independent searches; ranks take samples round-robin,
  results are gathered at the end -- no collective on the data path.
"""
from __future__ import annotations

import numpy as np

from . import api


def make_sharded_evaluator(local_eval, device=None, group=None):
    """local_eval(pc1[B,k], pc2[B,k], alpha[B]) -> partial llk[B] of this rank's markers.
    Returns an evaluator of the same signature that yields the all-reduced sums."""
    import torch
    import torch.distributed as dist

    def evaluate(pc1, pc2, alpha):
        part = np.asarray(local_eval(pc1, pc2, alpha), dtype=np.float64)
        t = torch.from_numpy(part.copy())
        if device is not None:
            t = t.to(device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        return t.cpu().numpy()
    return evaluate


def make_gpu_sharded_evaluator(ctx, num_pc, group=None, max_points=64):
    """Marker-sharded evaluator on a LikelihoodContext: parameter rows go to the device,
    `vb2_llk_eval_batch_device` writes partial LLKs into a device tensor on torch's current
    stream, RCCL all-reduces it in place; one host synchronisation per batch."""
    import torch
    import torch.distributed as dist
    stride = 2 * num_pc + 1
    pts = torch.zeros((max_points, stride), dtype=torch.float64, device="cuda")
    out = torch.zeros(max_points, dtype=torch.float64, device="cuda")

    def evaluate(pc1, pc2, alpha):
        B = len(alpha)
        assert B <= max_points
        rows = np.concatenate([np.asarray(pc1).reshape(B, num_pc), np.asarray(pc2).reshape(B, num_pc),
                               np.asarray(alpha).reshape(B, 1)], axis=1)
        pts[:B].copy_(torch.from_numpy(rows), non_blocking=False)
        stream = torch.cuda.current_stream().cuda_stream
        ctx.llk_device(pts.data_ptr(), out.data_ptr(), B, stream)
        dist.all_reduce(out[:B], op=dist.ReduceOp.SUM, group=group)
        return out[:B].cpu().numpy()
    return evaluate


def optimize_marker_sharded(evaluate_all_reduced, num_pc, **model_kw):
    """OptimizeLLK over an all-reducing evaluator; every rank gets the same estimate."""
    return api.optimize_with_evaluator(evaluate_all_reduced, num_pc, **model_kw)


def optimize_sample_parallel(samples, run_one, rank, world, group=None):
    """samples: list of work items; run_one(item) -> dict.  Rank r handles items r, r+world, ...
    Returns the full result list (in input order) on every rank."""
    import torch.distributed as dist
    mine = {i: run_one(samples[i]) for i in range(rank, len(samples), world)}
    gathered = [None] * world
    dist.all_gather_object(gathered, mine, group=group)
    out = [None] * len(samples)
    for part in gathered:
        for i, r in part.items():
            out[i] = r
    return out
