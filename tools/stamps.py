"""In-kernel timeline of llk_eval_kernel from per-workgroup wall-clock stamps (VB2_STAMPS=1)."""
import os, sys, ctypes as C
os.environ["VB2_STAMPS"] = "1"
_stamps_lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "verifybamid_amd", "libvb2_stamps.so")
if os.path.exists(_stamps_lib):
    os.environ.setdefault("VB2_LIB_PATH", _stamps_lib)      # (the stamps are compiled out of libvb2.so)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import verifybamid_amd as vb
from verifybamid_amd import _abi
B = int(os.environ.get("VB2_B", 8)); M = int(os.environ.get("VB2_M", 100000))
d = vb.synth.make_pileup(M, 30, 4, 0.05, 2)
rng = np.random.default_rng(5)
ctx = vb.LikelihoodContext(d)
pc1 = rng.normal(0, 0.03, size=(B, 4)); pc2 = rng.normal(0, 0.03, size=(B, 4)); al = rng.uniform(0, 0.5, size=B)
for _ in range(5): ctx.llk(pc1, pc2, al)
lib = _abi.lib()
buf = (C.c_ulonglong * (8 * 512))()
lib.vb2_debug_read_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
nb = lib.vb2_debug_read_stamps(ctx._h, buf, 512)
s = np.array(buf[:8 * nb], dtype=np.float64).reshape(nb, 8)
s = s[s[:, 0] > 0]
t0 = s[:, 0].min()
us = (s - t0) / 100.0          # 100 MHz -> us
names = ["entry", "points in LDS", "table built", "wave0 tiles done", "last wave tiles done", "block reduced", "finalized (last block)"]
print("blocks with stamps:", len(s), "B =", B)
for i, n in enumerate(names):
    col = us[:, i][s[:, i] > 0]
    if len(col): print("%-24s min %6.2f  median %6.2f  max %6.2f us" % (n, col.min(), np.median(col), col.max()))
if os.environ.get("VB2_STAMPS_DETAIL"):
    t4 = us[:, 4]
    print("last wave done, median by workgroup index mod 8 (XCD): " + " ".join("%.2f" % np.median(t4[x::8]) for x in range(8)))
    print("   by index mod 16: " + " ".join("%.1f" % np.median(t4[x::16]) for x in range(16)))
    print("   by index // 32:  " + " ".join("%.2f" % np.median(t4[32 * x:32 * x + 32]) for x in range(8)))
    order = np.argsort(-t4)
    print("   slowest: " + ", ".join("%d: %.1f" % (i, t4[i]) for i in order[:16]))
    print("   fastest: " + ", ".join("%d: %.1f" % (i, t4[i]) for i in order[-16:]))
ctx.close()
