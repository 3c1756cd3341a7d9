"""bridge.py -- TEST INFRASTRUCTURE ONLY: present a vb2_input-shaped object (anything
with num_pc, ud, means, read_off, bases, quals, alt_base, known_af, avg_depth,
sd_depth, sanity_disabled -- e.g. verifybamid_amd.PileupData) to the C oracle in the
reference's own shape (markers -> baseInfoIndex -> per-site vectors)."""
import numpy as np

from . import binding, refio


def flat_from_input(d):
    off = np.asarray(d.read_off, dtype=np.int64)
    depth = np.diff(off)
    M = depth.shape[0]
    present = depth > 0
    idx = np.full(M, -1, dtype=np.int32)
    idx[present] = np.arange(int(present.sum()), dtype=np.int32)
    site_off = np.zeros(int(present.sum()) + 1, dtype=np.int64)
    np.cumsum(depth[present], out=site_off[1:])
    b0, b1 = int(off[0]), int(off[-1])
    return refio.FlatInput(int(d.num_pc), np.asarray(d.ud, dtype=np.float64).reshape(M, -1),
                           np.asarray(d.means, dtype=np.float64), idx,
                           np.asarray(d.alt_base, dtype=np.uint8),
                           None if d.known_af is None else np.asarray(d.known_af, dtype=np.float64),
                           site_off, np.asarray(d.bases, dtype=np.uint8)[b0:b1],
                           np.asarray(d.quals, dtype=np.uint8)[b0:b1], float(d.avg_depth),
                           float(d.sd_depth), bool(d.sanity_disabled), d.known_af is not None)


def oracle_data(d):
    return binding.OracleData(flat_from_input(d))
