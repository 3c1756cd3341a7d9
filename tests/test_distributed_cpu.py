"""N > 1 paths on CPU: world_size 2 over gloo (no GPU here, so the per-rank likelihood is the
oracle on the rank's shard -- what is under test is the sharding, the all-reduce plumbing,
the library's optimiser running in lock-step on every rank, and the sample-parallel gather).
The same drivers run with backend "nccl" (= RCCL) and a LikelihoodContext on the GPU box
(`bench.py --mode marker`, tests/test_gpu_parity.py::test_marker_sharded_single_process)."""
import os
import socket

import numpy as np
import pytest

import verifybamid_amd as vb
from verifybamid_amd import distributed as vbd
from oracle.bridge import oracle_data

torch = pytest.importorskip("torch")
import torch.multiprocessing as mp  # noqa: E402


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _oracle_eval(od):
    return lambda p1, p2, a: [od.llk(p1[i], p2[i], a[i]) for i in range(len(a))]


def _worker(rank, world, port, mode, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        if mode == "marker":
            d = vb.synth.make_pileup(1500, 12, 2, alpha_true=0.1, seed=31, missing_frac=0.05)
            shard = d.shard(rank, world)
            ev = vbd.make_sharded_evaluator(_oracle_eval(oracle_data(shard)))
            est = vbd.optimize_marker_sharded(ev, 2, trace_capacity=8192)
            q.put((rank, est["alpha"], est["llk1"], est["num_eval"], est["pc"].tolist(),
                   shard.num_marker, shard.num_read))
        else:
            seeds = [41, 42, 43, 44, 45]

            def run_one(seed):
                d = vb.synth.make_pileup(400, 10, 2, alpha_true=0.02 * (seed - 40), seed=seed)
                est = vb.optimize_with_evaluator(_oracle_eval(oracle_data(d)), 2)
                return dict(seed=seed, alpha=est["alpha"], llk1=est["llk1"])
            res = vbd.optimize_sample_parallel(seeds, run_one, rank, world)
            q.put((rank, res))
    finally:
        dist.destroy_process_group()


def _spawn(mode, world=2):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, mode, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return sorted(out, key=lambda x: x[0])


def test_marker_sharded_world2_matches_single_process():
    out = _spawn("marker")
    (r0, a0, l0, n0, pc0, m0, rd0), (r1, a1, l1, n1, pc1, m1, rd1) = out
    d = vb.synth.make_pileup(1500, 12, 2, alpha_true=0.1, seed=31, missing_frac=0.05)
    assert m0 + m1 == d.num_marker and rd0 + rd1 == d.num_read
    assert abs(rd0 - rd1) < 200                       # balanced on reads
    # every rank ran the same search on the same all-reduced values
    assert (a0, l0, n0, pc0) == (a1, l1, n1, pc1)
    ref = oracle_data(d).optimize()
    assert abs(a0 - ref["alpha"]) <= 1e-6             # north star: 1e-4
    assert abs(l0 - ref["llk1"]) <= 1e-9 * abs(ref["llk1"])


def test_sample_parallel_world2_gathers_all_results():
    out = _spawn("sample")
    res0, res1 = out[0][1], out[1][1]
    assert res0 == res1 and [r["seed"] for r in res0] == [41, 42, 43, 44, 45]
    for r in res0:
        d = vb.synth.make_pileup(400, 10, 2, alpha_true=0.02 * (r["seed"] - 40), seed=r["seed"])
        ref = oracle_data(d).optimize()
        assert r["alpha"] == ref["alpha"] and r["llk1"] == ref["llk1"]
