"""rocprofv3 driver: C3-shaped context, 30 launches of the 8-point kernel (device API path via host)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import verifybamid_amd as vb
M = int(os.environ.get('VB2_M', 100000)); k = int(os.environ.get('VB2_K', 4)); B = int(os.environ.get('VB2_B', 8))
d = vb.synth.make_pileup(M, 30, k, 0.05, 2)
rng = np.random.default_rng(5)
ctx = vb.LikelihoodContext(d)
pc1 = rng.normal(0,0.03,size=(B,k)); pc2 = rng.normal(0,0.03,size=(B,k)); al = rng.uniform(0,0.5,size=B)
for _ in range(int(os.environ.get('VB2_N', 30))): ctx.llk(pc1,pc2,al)
ctx.close()
