"""In-kernel timeline of one command of the resident search kernel (VB2_STAMPS=1): the stamps
left behind are those of the last evaluation of the search."""
import os, sys, ctypes as C
os.environ["VB2_STAMPS"] = "1"
# VB2_STAMPS_ROUND=1: the build whose per-workgroup stamps are those of round 200 (make stamps_round), times from
# workgroup 0's entry into that round's evaluation
_round = os.environ.get("VB2_STAMPS_ROUND", "") == "1"
_stamps_lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "verifybamid_amd",
                           "libvb2_stamps_r.so" if _round else "libvb2_stamps.so")
if os.path.exists(_stamps_lib):
    os.environ.setdefault("VB2_LIB_PATH", _stamps_lib)      # (the stamps are compiled out of libvb2.so)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import verifybamid_amd as vb
from verifybamid_amd import _abi
M = int(os.environ.get("VB2_M", 100000))
d = vb.synth.make_pileup(M, 30, 4, 0.05, 2)
ctx = vb.LikelihoodContext(d)
ctx.optimize()
lib = _abi.lib()
buf = (C.c_ulonglong * (8 * 512))()
lib.vb2_debug_read_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
nb = lib.vb2_debug_read_stamps(ctx._h, buf, 512)
s = np.array(buf[:8 * nb], dtype=np.float64).reshape(nb, 8)
t0 = s[0, 0] if _round else s[0, 7]
if nb > 4 and s[4, 7] > 0:
    n = s[4, 7]
    print("device-side search, %d rounds, per round: control step %.2f us, relay hop (control done -> workgroup 0 has the "
          "round) %.2f us, evaluation (has the round -> next round's top) %.2f us"
          % (n, s[1, 7] / n / 100.0, s[2, 7] / n / 100.0, s[3, 7] / n / 100.0))
if nb > 16 and s[4, 7] > 0:
    n = s[4, 7]
    names2 = ["consume + commit + accept", "rank + candidates + convergence", "unpack (InvLogit) + rows to the relay",
              "state load", "save + publish (incl. store drain)"]
    print("control wave, per round (long way + replay): " + ", ".join("%s %.2f us" % (nm, s[5 + i, 7] / n / 100.0) for i, nm in enumerate(names2)))
    print("rounds that took the short way (post_decide + one relay word): %d of %d; control wave's tile-phase work "
          "(replay + speculate) %.2f us per round" % (s[10, 7], n, s[11, 7] / n / 100.0))
    print("  of which: " + ", ".join("%s %.2f" % (nm, s[12 + i, 7] / n / 100.0) for i, nm in enumerate(
        ["state load", "apply_outcome", "-", "speculate", "save + drain"])) + " us (each incl. one clock read, ~0.4 us)")
if nb > 22 and s[4, 7] > 0:
    n = s[4, 7]
    print("workgroup 0, averaged over the rounds, since it has the round: own block reduced %.2f us, all sets in + summed %.2f us"
          % (s[20, 7] / n / 100.0, s[21, 7] / n / 100.0))
s = s[s[:, 0] > 0]
us = (s - t0) / 100.0
names = ["entry", "points in LDS", "table built", "wave 1 out of its read loop", "slowest wave done", "block reduced", "finalized (last block)"]
print("blocks:", len(s), " t=0: " + ("workgroup 0 enters the evaluation of round 200" if _round else "round begun (loop top of the resident kernel)"))
for i, n in enumerate(names):
    col = us[:, i][s[:, i] > 0]
    if len(col): print("%-24s min %6.2f  median %6.2f  max %6.2f us" % (n, col.min(), np.median(col), col.max()))
if os.environ.get("VB2_STAMPS_DETAIL"):
    print("workgroup 0's own stamps (us): " + ", ".join("%s %.2f" % (nm, us[0, i]) for i, nm in enumerate(names)))
    print("workgroup 1's:                 " + ", ".join("%s %.2f" % (nm, us[1, i]) for i, nm in enumerate(names)))
    print("workgroup 200's:               " + ", ".join("%s %.2f" % (nm, us[200, i]) for i, nm in enumerate(names)))
    br = us[:, 5]
    order = np.argsort(-br)
    print("slowest workgroups by 'block reduced' (us): " + ", ".join("%d: %.2f" % (i, br[i]) for i in order[:12]))
    print("fastest: " + ", ".join("%d: %.2f" % (i, br[i]) for i in order[-8:]))
    by_xcd = [np.median(br[x::8]) for x in range(8)]
    print("median by workgroup index mod 8 (XCD): " + " ".join("%.2f" % v for v in by_xcd))
    t2 = us[:, 2]
    print("table built, by index mod 8: " + " ".join("%.2f" % np.median(t2[x::8]) for x in range(8)))
ctx.close()
