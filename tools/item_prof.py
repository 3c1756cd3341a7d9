"""Where a wave's time per work item goes in the 48-point launch (build: tools/build_variant.sh itemprof -DVB2_WITH_STAMPS
-DVB2_ITEM_PROF; run with VB2_LIB_PATH=build_variants/itemprof/libvb2.so): per item, in shader-clock cycles summed over the
waves of every workgroup -- waiting for the tile record, for the first rows, the read loop, waiting for the per-marker
constants, the epilogue, the product exchange + slot write, the next draw."""
import os, sys, ctypes as C
os.environ["VB2_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import verifybamid_amd as vb
from verifybamid_amd import _abi
B = int(os.environ.get("VB2_B", 48)); M = int(os.environ.get("VB2_M", 100000))
d = vb.synth.make_pileup(M, 30, 4, 0.05, 2, q_lo=int(os.environ.get("VB2_QLO", 20)), q_hi=int(os.environ.get("VB2_QHI", 40)))
rng = np.random.default_rng(5)
ctx = vb.LikelihoodContext(d)
pc1 = rng.normal(0, 0.03, size=(B, 4)); pc2 = rng.normal(0, 0.03, size=(B, 4)); al = rng.uniform(0, 0.5, size=B)
lib = _abi.lib()
lib.vb2_debug_read_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
def read():
    buf = (C.c_ulonglong * (8 * 512))()
    nb = lib.vb2_debug_read_stamps(ctx._h, buf, 512)
    return np.array(buf[:8 * nb], dtype=np.float64).reshape(nb, 8)
for _ in range(20): ctx.llk(pc1, pc2, al)
a = read()
N = 50
for _ in range(N): ctx.llk(pc1, pc2, al)
s = read() - a
keep = [i for i in range(len(s)) if i not in (20, 21, 22) and s[i, 7] > 0]
s = s[keep]
items = s[:, 7].sum()
names = ["tile record wait", "first rows wait", "read loop", "constants wait", "epilogue", "exchange + slot", "next draw"]
tot = s[:, :7].sum()
print("%d workgroups, %.0f items per launch, %.0f cycles per item (per wave)" % (len(s), items / N, tot / items))
for i, n in enumerate(names):
    print("  %-18s %8.0f cycles  %5.1f %%" % (n, s[:, i].sum() / items, 100 * s[:, i].sum() / tot))
ctx.close()
