"""Seeded synthetic pileups of the shapes BASELINE.json names (recipe: SURVEY.md 8d).

Panel: UD ~ N(0, sigma_k) with column s.d. (7, 5, 3, 1.5, ...), mu ~ 2*Beta(0.8, 0.8)
clipped to [0.02, 1.98].  Per marker: AF = clamp(mu/2); intended genotype
g2 ~ Binom(2, AF), contaminant g1 ~ Binom(2, AF); depth ~ Poisson(mean_depth);
each read comes from the contaminant with probability alpha_true, carries the alt
allele with probability g/2, has Phred q ~ UniformInt[q_lo, q_hi], and is replaced by
one of the three other bases with probability 10^(-q/10); strand 50/50 ('.'/',' for
ref, upper/lower-case letter otherwise); quality character = q + 33.

Pure numpy, deterministic for a given seed; the same arrays feed the HIP path and
the oracle.  `write_files` emits a samtools-style pileup + .UD/.mu/.bed so the CLI and
file readers can be exercised on the same data.
"""
from __future__ import annotations

import numpy as np

from .api import PileupData

_SD = (7.0, 5.0, 3.0, 1.5, 1.0, 0.8, 0.6, 0.5, 0.4, 0.3)
_BASES = np.frombuffer(b"ACGT", dtype=np.uint8)
_LOWER = np.frombuffer(b"acgt", dtype=np.uint8)


def make_pileup(num_marker, mean_depth=30.0, num_pc=4, alpha_true=0.05, seed=1, q_lo=20, q_hi=40,
                missing_frac=0.0):
    rng = np.random.default_rng(seed)
    M, k = int(num_marker), int(num_pc)
    sd = np.array([_SD[i] if i < len(_SD) else 0.3 for i in range(k)])
    ud = rng.normal(0.0, 1.0, size=(M, k)) * sd
    mu = np.clip(2.0 * rng.beta(0.8, 0.8, size=M), 0.02, 1.98)
    af = np.clip(mu / 2.0, 0.00005, 0.99995)
    g2 = rng.binomial(2, af)
    g1 = rng.binomial(2, af)
    depth = rng.poisson(mean_depth, size=M).astype(np.int64)
    if missing_frac > 0:
        depth[rng.random(M) < missing_frac] = 0
    ref_i = rng.integers(0, 4, size=M)
    alt_i = (ref_i + rng.integers(1, 4, size=M)) % 4
    read_off = np.zeros(M + 1, dtype=np.int64)
    np.cumsum(depth, out=read_off[1:])
    R = int(read_off[-1])
    mk = np.repeat(np.arange(M), depth)
    from_contam = rng.random(R) < alpha_true
    g = np.where(from_contam, g1[mk], g2[mk])
    is_alt = rng.random(R) < (g / 2.0)
    q = rng.integers(q_lo, q_hi + 1, size=R)
    err = rng.random(R) < np.power(10.0, -q / 10.0)
    true_base = np.where(is_alt, alt_i[mk], ref_i[mk])
    obs_base = np.where(err, (true_base + rng.integers(1, 4, size=R)) % 4, true_base)
    fwd = rng.random(R) < 0.5
    is_ref = obs_base == ref_i[mk]
    ch = np.where(fwd, _BASES[obs_base], _LOWER[obs_base])
    ch = np.where(is_ref, np.where(fwd, ord("."), ord(",")), ch).astype(np.uint8)
    quals = (q + 33).astype(np.uint8)
    alt_base = _BASES[alt_i]
    nsite = int((depth > 0).sum())
    avg = float(R) / nsite if nsite else float("nan")
    d = PileupData(k, ud, mu, read_off, ch, quals, alt_base, None, avg, 0.0, True,
                   dict(seed=seed, alpha_true=alpha_true, mean_depth=mean_depth,
                        ref_base=_BASES[ref_i]))
    return d


def with_sanity_stats(d: PileupData):
    """avgDepth/sdDepth as IsSanityCheckOK computes them (ContaminationEstimator.cpp:543-587),
    enabling the +-3sd depth filter."""
    depth = np.diff(d.read_off).astype(np.float64)
    present = depth > 0
    n = int(present.sum())
    avg = depth.sum() / n
    sd = float(np.sqrt((depth[present] ** 2).sum() / n - avg * avg))
    return PileupData(d.num_pc, d.ud, d.means, d.read_off, d.bases, d.quals, d.alt_base, d.known_af,
                      float(avg), sd, False, dict(d.meta))


def write_files(d: PileupData, prefix):
    """Write <prefix>.UD/.mu/.bed and <prefix>.pileup (6-column samtools pileup)."""
    M = d.num_marker
    ref = d.meta["ref_base"]
    with open(prefix + ".UD", "w") as f:
        for i in range(M):
            f.write("\t".join(repr(float(x)) for x in d.ud[i]) + "\n")
    with open(prefix + ".mu", "w") as f:
        for i in range(M):
            f.write("1:%d_%s/%s_rs%d\t%r\n" % (1000 + 10 * i, chr(ref[i]), chr(d.alt_base[i]), i,
                                                float(d.means[i])))
    with open(prefix + ".bed", "w") as f:
        for i in range(M):
            p = 1000 + 10 * i
            f.write("1\t%d\t%d\t%s\t%s\n" % (p - 1, p, chr(ref[i]), chr(d.alt_base[i])))
    bases = d.bases.tobytes().decode("latin-1")
    quals = d.quals.tobytes().decode("latin-1")
    with open(prefix + ".pileup", "w") as f:
        for i in range(M):
            b, e = int(d.read_off[i]), int(d.read_off[i + 1])
            if e == b:
                continue
            f.write("1\t%d\t%s\t%d\t%s\t%s\n" % (1000 + 10 * i, chr(ref[i]), e - b, bases[b:e], quals[b:e]))
    return prefix


def reads_on_panel(means, ref_char, alt_char, mean_depth=30.0, alpha_true=0.05, seed=1, q_lo=20, q_hi=40):
    """The read-drawing half of the recipe on a GIVEN panel (e.g. the reference's bundled
    1000g.phase3 panels: real mean genotypes, real ref/alt alleles -- first character of the
    .bed columns, like the reference reads them).  Genotypes ~ Binom(2, clamp(mu/2)) for both
    samples.  Returns (read_off, bases, quals): one pileup entry per panel row, in row order."""
    rng = np.random.default_rng(seed)
    mu = np.asarray(means, dtype=np.float64)
    M = mu.shape[0]
    code = {ord("A"): 0, ord("C"): 1, ord("G"): 2, ord("T"): 3}
    ref_i = np.array([code.get(int(c), 0) for c in np.asarray(ref_char, dtype=np.uint8)])
    alt_i = np.array([code.get(int(c), 1) for c in np.asarray(alt_char, dtype=np.uint8)])
    af = np.clip(mu / 2.0, 0.00005, 0.99995)
    g2 = rng.binomial(2, af)
    g1 = rng.binomial(2, af)
    depth = rng.poisson(mean_depth, size=M).astype(np.int64)
    read_off = np.zeros(M + 1, dtype=np.int64)
    np.cumsum(depth, out=read_off[1:])
    R = int(read_off[-1])
    mk = np.repeat(np.arange(M), depth)
    g = np.where(rng.random(R) < alpha_true, g1[mk], g2[mk])
    is_alt = rng.random(R) < (g / 2.0)
    q = rng.integers(q_lo, q_hi + 1, size=R)
    err = rng.random(R) < np.power(10.0, -q / 10.0)
    true_base = np.where(is_alt, alt_i[mk], ref_i[mk])
    obs_base = np.where(err, (true_base + rng.integers(1, 4, size=R)) % 4, true_base)
    fwd = rng.random(R) < 0.5
    ch = np.where(fwd, _BASES[obs_base], _LOWER[obs_base])
    ch = np.where(obs_base == ref_i[mk], np.where(fwd, ord("."), ord(",")), ch).astype(np.uint8)
    return read_off, ch, (q + 33).astype(np.uint8)


def read_bed_rows(bed_path):
    """(chr, pos, first char of ref, first char of alt) per row of a .bed (plain or gzip'd)."""
    import gzip
    raw = open(bed_path, "rb").read()
    if raw[:2] == b"\x1f\x8b":
        raw = gzip.decompress(raw)
    chrs, poss, refs, alts = [], [], [], []
    for line in raw.decode("latin-1").split("\n"):
        tok = line.split()
        if len(tok) < 5:
            continue
        chrs.append(tok[0]); poss.append(int(tok[2])); refs.append(ord(tok[3][0])); alts.append(ord(tok[4][0]))
    return chrs, np.array(poss), np.array(refs, dtype=np.uint8), np.array(alts, dtype=np.uint8)


def read_mu_column(mu_path):
    import gzip
    raw = open(mu_path, "rb").read()
    if raw[:2] == b"\x1f\x8b":
        raw = gzip.decompress(raw)
    return np.array([float(line.split()[1]) for line in raw.decode("latin-1").split("\n") if line.strip()])


def write_pileup_text(path, chrs, poss, ref_char, read_off, bases, quals):
    """6-column samtools pileup for the drawn reads, one line per covered panel row."""
    b = np.asarray(bases, dtype=np.uint8).tobytes().decode("latin-1")
    q = np.asarray(quals, dtype=np.uint8).tobytes().decode("latin-1")
    with open(path, "w") as f:
        for i in range(len(chrs)):
            lo, hi = int(read_off[i]), int(read_off[i + 1])
            if hi > lo:
                f.write("%s\t%d\t%s\t%d\t%s\t%s\n" % (chrs[i], int(poss[i]), chr(int(ref_char[i])), hi - lo, b[lo:hi], q[lo:hi]))
    return path


def real_panel_sample(svd_prefix, pileup_path, mean_depth=30.0, alpha_true=0.05, seed=1):
    """Draws reads on the panel behind `svd_prefix` (.mu/.bed) and writes them as a text pileup."""
    chrs, poss, refs, alts = read_bed_rows(svd_prefix + ".bed")
    mu = read_mu_column(svd_prefix + ".mu")
    off, b, q = reads_on_panel(mu, refs, alts, mean_depth, alpha_true, seed)
    return write_pileup_text(pileup_path, chrs, poss, refs, off, b, q)


def write_known_af(svd_prefix, path, seed=1):
    """A --KnownAF file for the panel behind `svd_prefix` (.bed/.mu): one `chr start end ref alt af` row per panel
    row, af = clamp(mu / 2 + N(0, 0.02)) -- seeded, so a fixture recipe and its tests write the same file."""
    chrs, poss, refs, alts = read_bed_rows(svd_prefix + ".bed")
    mu = read_mu_column(svd_prefix + ".mu")
    rng = np.random.default_rng(seed)
    af = np.clip(mu / 2.0 + rng.normal(0.0, 0.02, size=mu.shape[0]), 0.001, 0.999)
    with open(path, "w") as f:
        for i in range(len(chrs)):
            f.write("%s\t%d\t%d\t%s\t%s\t%r\n" % (chrs[i], int(poss[i]) - 1, int(poss[i]), chr(int(refs[i])),
                                                  chr(int(alts[i])), float(af[i])))
    return path
