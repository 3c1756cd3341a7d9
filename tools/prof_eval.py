"""Small driver for rocprofv3: C3-shaped context, a few evals at each batch size."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import verifybamid_amd as vb
M = int(os.environ.get('VB2_M', 100000)); k = int(os.environ.get('VB2_K', 4))
d = vb.synth.make_pileup(M, 30, k, 0.05, 2)
rng = np.random.default_rng(5)
ctx = vb.LikelihoodContext(d)
for B in (1, 2, 4, 8):
    pc1 = rng.normal(0,0.03,size=(B,k)); pc2 = rng.normal(0,0.03,size=(B,k)); al = rng.uniform(0,0.5,size=B)
    for _ in range(20): ctx.llk(pc1,pc2,al)
ctx.close()
