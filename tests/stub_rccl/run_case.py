"""Runs ONE N > 1 case of vb2_shard_group_* against the in-process librccl stand-in and prints a JSON line.

Started by tests/test_gpu_parity.py in a fresh process (the run-time binding of the collective library is
decided once per process) with VB2_RCCL_LIB=tests/stub_rccl/librccl_stub.so.

    python tests/stub_rccl/run_case.py group <c2|c3> <nshard>   one process, nshard shards on device 0:
                                                                 ncclCommInitAll + the grouped all-reduce loop
    python tests/stub_rccl/run_case.py ranks <c2|c3> <nranks>   nranks threads, each a rank-mode group
                                                                 (ncclCommInitRank with nranks > 1)
"""
import json
import os
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np          # noqa: E402
import verifybamid_amd as vb  # noqa: E402
from verifybamid_amd import _abi  # noqa: E402


def fixture(size):
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "synthetic_%s.json" % size)))
    g = fx["generator"]
    d = vb.synth.make_pileup(g["markers"], g["mean_depth"], g["num_pc"], alpha_true=g["alpha_true"], seed=g["seed"])
    P = fx["points"]
    want = np.array([float.fromhex(x) for x in fx["llk_hex"]])
    m = fx["models"]["heter"]
    return d, np.array(P["pc1"]), np.array(P["pc2"]), np.array(P["alpha"]), want, m


def est_summary(e):
    return {"alpha_hex": float(e["alpha"]).hex(), "llk1_hex": float(e["llk1"]).hex(), "num_eval": int(e["num_eval"])}


def main():
    mode, size, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
    d, pc1, pc2, al, want, m = fixture(size)
    out = {"mode": mode, "size": size, "n": n, "want_alpha_hex": m["alpha_hex"], "want_llk1_hex": m["llk1_hex"],
           "want_num_eval": m["num_eval"]}
    rel = lambda a, b: float(np.max(np.abs(np.asarray(a) - np.asarray(b)) / np.abs(np.asarray(b))))   # noqa: E731
    if mode == "group":
        with vb.ShardGroup(d, devices=[0] * n) as g:
            info = g.info()
            out["info"] = {k: info[k] for k in ("num_shard", "nranks", "uses_rccl", "rccl_stub", "partial_sums")}
            got = g.llk(pc1, pc2, al)
            big = g.llk(np.tile(pc1, (7, 1)), np.tile(pc2, (7, 1)), np.tile(al, 7))     # more than one launch per call
            out["allreduces_after_eval"] = g.info()["num_allreduce"]
            est = g.optimize()
            out["allreduces_after_search"] = g.info()["num_allreduce"]
        _abi.set_tunable("shard_reduce_host", 1)                 # the same shards, summed on the host in shard order
        with vb.ShardGroup(d, devices=[0] * n) as gh:
            assert not gh.info()["uses_rccl"]
            host = gh.llk(pc1, pc2, al)
            est_h = gh.optimize()
        _abi.set_tunable("shard_reduce_host", 0)
        out.update(rel_vs_fixture=rel(got, want), equals_host_sum=bool(np.array_equal(got, host)),
                   big_equals_tiled=bool(np.array_equal(big, np.tile(got, 7))), est=est_summary(est),
                   est_host=est_summary(est_h))
    elif mode == "ranks":
        uid = vb.ShardGroup.unique_id()
        res = [None] * n
        err = [None] * n

        def rank_main(r):
            try:
                with vb.ShardGroup(d, device=0, rank=r, nranks=n, unique_id=uid) as g:
                    info = g.info()
                    got = g.llk(pc1, pc2, al)
                    est = g.optimize()
                    res[r] = dict(info={k: info[k] for k in ("num_shard", "nranks", "rank", "uses_rccl", "rccl_stub",
                                                              "partial_sums")},
                                  got=got, est=est_summary(est), allreduces=g.info()["num_allreduce"])
            except Exception as exc:            # noqa: BLE001 -- reported in the JSON
                err[r] = "%s: %s" % (type(exc).__name__, exc)
        ts = [threading.Thread(target=rank_main, args=(r,)) for r in range(n)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        out["errors"] = err
        if not any(err):
            out["ranks"] = [dict(info=r["info"], est=r["est"], allreduces=r["allreduces"],
                                 rel_vs_fixture=rel(r["got"], want)) for r in res]
            out["all_ranks_equal"] = bool(all(np.array_equal(res[0]["got"], r["got"]) for r in res[1:]))
            # the shards of a rank-mode group are the shards of the one-process group: same partial sums, same order
            _abi.set_tunable("shard_reduce_host", 1)
            with vb.ShardGroup(d, devices=[0] * n) as gh:
                host = gh.llk(pc1, pc2, al)
            out["equals_host_sum"] = bool(np.array_equal(res[0]["got"], host))
    else:
        raise SystemExit("unknown mode")
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
