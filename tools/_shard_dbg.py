import os, sys, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests/stub_rccl")
import verifybamid_amd as vb
from verifybamid_amd import _abi
from run_case import fixture
d, pc1, pc2, al, want, m = fixture("c3")
n = 4
for rep in range(3):
    with vb.ShardGroup(d, devices=[0] * n) as g:
        got = g.llk(pc1, pc2, al)
        big = g.llk(np.tile(pc1, (7, 1)), np.tile(pc2, (7, 1)), np.tile(al, 7))
        t = np.tile(got, 7)
        bad = np.nonzero(big != t)[0]
        print("uses_rccl", g.info()["uses_rccl"], "points", len(al), "mismatches at", bad.tolist(), [(big[i] - t[i]) / t[i] for i in bad[:5]])
