"""refio.py -- TEST INFRASTRUCTURE ONLY.

Python restatement of the reference's *input side* of the contamination path, used
to feed the C oracle (vb2_oracle.c) with exactly the arrays the reference's
ComputeMixLLKs would see, and to cross-check the product's own C++ readers.
Nothing under verifybamid_amd/ imports this.

Restated (file:line relative to /root/reference):
  * ContaminationEstimator::ReadChooseBed    ContaminationEstimator.cpp:413-438
  * ContaminationEstimator::ReadMatrixUD     ContaminationEstimator.cpp:342-373
  * ContaminationEstimator::ReadMean         ContaminationEstimator.cpp:440-459
  * ContaminationEstimator::ReadAF           ContaminationEstimator.cpp:461-487
  * SimplePileupViewer::ReadPileup           SimplePileupViewer.cpp:748-833
  * ParsePileupSeqBasesOnly                  SimplePileupViewer.cpp:711-746
  * ContaminationEstimator::BuildResolvedMarkers  ContaminationEstimator.cpp:67-86
  * ContaminationEstimator::IsSanityCheckOK  ContaminationEstimator.cpp:543-587
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np


def _newline_terminated_lines(path):
    """statgen InputFile::readLine (statgen/InputFile.cpp:112-130) returns -1 on
    EOF, so a final line without '\\n' is silently dropped (SURVEY quirk vi)."""
    with open(path, "rb") as fh:
        data = fh.read()
    if data[:2] == b"\x1f\x8b":          # InputFile opens gzip'd files by their magic bytes (GzipFileType)
        import gzip
        data = gzip.decompress(data)
    parts = data.split(b"\n")
    return [p.decode("latin-1") for p in parts[:-1]]


@dataclass
class Panel:
    num_pc: int
    ud: np.ndarray          # [M, k] float64
    means: np.ndarray       # [M]    float64
    pos: list               # [(chr, pos)] panel order (PosVec)
    bed: dict               # ChooseBed: chr -> {pos: (ref, alt)}  single chars
    known_af: dict | None = None

    @property
    def num_marker(self):
        return self.ud.shape[0]


def read_bed(path):
    pos, bed = [], {}
    chrom, p, ref, alt = "", 0, "\0", "\0"
    for line in _newline_terminated_lines(path):
        tok = line.split()
        # `ss >> chr >> pos >> pos; ss >> ref >> alt` with ref/alt of type char:
        # only the FIRST character of each allele column survives (quirk iv).
        if len(tok) >= 3:
            chrom, p = tok[0], int(tok[2])
        # operator>>(char&) skips whitespace and then takes ONE character, twice:
        # ref/alt are the first two non-blank characters after the third column.
        rest = "".join(tok[3:])
        if len(rest) >= 1:
            ref = rest[0]
        if len(rest) >= 2:
            alt = rest[1]
        pos.append((chrom, p))
        bed.setdefault(chrom, {})[p] = (ref, alt)
    return pos, bed


def read_ud(path, num_pc):
    rows = []
    for line in _newline_terminated_lines(path):
        tok = line.split()
        if len(tok) < num_pc:
            raise SystemExit("--NumPC larger than the number of PCs in the .UD file")
        rows.append([float(x) for x in tok[:num_pc]])
    return np.asarray(rows, dtype=np.float64).reshape(len(rows), num_pc)


def read_mu(path):
    vals = []
    for line in _newline_terminated_lines(path):
        tok = line.split()
        vals.append(float(tok[1]))
    return np.asarray(vals, dtype=np.float64)


def read_known_af(path):
    af = {}
    with open(path) as fh:
        for line in fh:  # std::getline: the last unterminated line IS kept here
            tok = line.split()
            if len(tok) < 6:
                continue
            af.setdefault(tok[0], {})[int(tok[2])] = float(tok[5])
    return af


def read_panel(prefix, num_pc, known_af_path=None):
    pos, bed = read_bed(prefix + ".bed")
    ud = read_ud(prefix + ".UD", num_pc)
    mu = read_mu(prefix + ".mu")
    kaf = read_known_af(known_af_path) if known_af_path else None
    return Panel(num_pc, ud, mu, pos, bed, kaf)


def parse_pileup_seq(seq, qual):
    """ParsePileupSeqBasesOnly (SimplePileupViewer.cpp:711-746)."""
    pseq, pqual = [], []
    i, iq, n = 0, 0, len(seq)
    while i < n:
        c = seq[i]
        if c in "+-":
            j = i + 1
            while j < n and seq[j].isdigit():
                j += 1
            digit_len = j - (i + 1)
            clip = int(seq[i + 1:j])
            i += digit_len + clip
        elif c == "^":
            i += 1
        elif c in ".,ACGTNacgtn":
            pseq.append(c)
            pqual.append(qual[iq])
            iq += 1
        elif c in "*#":
            iq += 1
        i += 1
    return "".join(pseq), "".join(pqual)


@dataclass
class Viewer:
    base_info: list = field(default_factory=list)   # list[str]
    qual_info: list = field(default_factory=list)
    pos_index: dict = field(default_factory=dict)   # chr -> {pos: idx}
    num_bases: int = 0
    effective_num_site: int = 0
    avg_depth: float = 0.0
    sd_depth: float = 0.0


def read_pileup(path, bed):
    """SimplePileupViewer::ReadPileup (SimplePileupViewer.cpp:748-833)."""
    v = Viewer()
    with open(path) as fh:
        for line in fh.read().split("\n"):
            if line == "" :
                # std::getline yields a last empty "line" only if the file does not
                # end in '\n'; an empty line re-uses stale fields in the reference.
                # Fixture files have neither; skip.
                continue
            tok = line.split()
            chrom, p, ref = tok[0], int(tok[1]), tok[2]
            seq = tok[4] if len(tok) > 4 else ""
            qual = tok[5] if len(tok) > 5 else ""
            if ("." in seq or "," in seq) and ref == ".":
                raise SystemExit("Pileup format error: cannot find ref allele")
            pseq, pqual = parse_pileup_seq(seq, qual)
            depth = len(pqual)
            if chrom not in bed or p not in bed[chrom]:
                continue
            existed = chrom in v.pos_index and p in v.pos_index[chrom]
            if not existed:
                v.pos_index.setdefault(chrom, {})[p] = len(v.base_info)
                v.base_info.append(pseq)
                v.qual_info.append(pqual)
            # duplicate line: merged copy is built but never stored (quirk vii)
            v.num_bases += depth
            v.effective_num_site += 1
    v.avg_depth = v.num_bases / v.effective_num_site if v.effective_num_site else float("nan")
    return v


def sanity_check(panel, v):
    """IsSanityCheckOK (ContaminationEstimator.cpp:543-587). Mutates v like the reference."""
    acc = v.sd_depth
    for chrom, p in panel.pos:
        if chrom not in v.pos_index or p not in v.pos_index[chrom]:
            continue
        d = len(v.base_info[v.pos_index[chrom][p]])
        acc += d * d
    v.sd_depth = math.sqrt(acc / v.effective_num_site - v.avg_depth * v.avg_depth)
    v.effective_num_site = 0
    for chrom, p in panel.pos:
        if chrom not in v.pos_index or p not in v.pos_index[chrom]:
            continue
        d = len(v.base_info[v.pos_index[chrom][p]])
        if d == 0 or d < (v.avg_depth - 3 * v.sd_depth) or d > (v.avg_depth + 3 * v.sd_depth):
            continue
        v.effective_num_site += 1
    return v.effective_num_site > 1000 and v.effective_num_site > panel.num_marker * 0.1


@dataclass
class FlatInput:
    """What ComputeMixLLKs reaches through `ptr` -- the argument of the C oracle."""
    num_pc: int
    ud: np.ndarray                 # [M,k] f64
    means: np.ndarray              # [M] f64
    base_info_index: np.ndarray    # [M] i32
    alt_base: np.ndarray           # [M] u8 (S1)
    known_af: np.ndarray | None    # [M] f64
    site_off: np.ndarray           # [S+1] i64
    bases: np.ndarray              # [R] u8
    quals: np.ndarray              # [R] u8
    avg_depth: float
    sd_depth: float
    sanity_disabled: bool
    af_known: bool

    @property
    def num_marker(self):
        return int(self.ud.shape[0])


def build_resolved(panel, v, sanity_disabled=True):
    """BuildResolvedMarkers (ContaminationEstimator.cpp:67-86) + flattening."""
    M = panel.num_marker
    idx = np.full(M, -1, dtype=np.int32)
    alt = np.zeros(M, dtype=np.uint8)
    kaf = np.zeros(M, dtype=np.float64) if panel.known_af is not None else None
    for i, (chrom, p) in enumerate(panel.pos[:M]):
        ci = v.pos_index.get(chrom)
        if ci is None or p not in ci:
            continue
        idx[i] = ci[p]
        alt[i] = ord(panel.bed[chrom][p][1])
        if kaf is not None:
            # unordered_map operator[] default-constructs 0.0 for a missing key
            kaf[i] = panel.known_af.get(chrom, {}).get(p, 0.0)
    off = np.zeros(len(v.base_info) + 1, dtype=np.int64)
    for s, b in enumerate(v.base_info):
        off[s + 1] = off[s] + len(b)
    bases = np.frombuffer("".join(v.base_info).encode("latin-1"), dtype=np.uint8).copy()
    quals = np.frombuffer("".join(v.qual_info).encode("latin-1"), dtype=np.uint8).copy()
    return FlatInput(panel.num_pc, np.ascontiguousarray(panel.ud), np.ascontiguousarray(panel.means[:M]),
                     idx, alt, kaf, off, bases, quals, float(v.avg_depth), float(v.sd_depth),
                     bool(sanity_disabled), panel.known_af is not None)


def load_flat(svd_prefix, pileup_path, num_pc, sanity_disabled=True, known_af_path=None):
    panel = read_panel(svd_prefix, num_pc, known_af_path)
    v = read_pileup(pileup_path, panel.bed)
    if not sanity_disabled:
        sanity_check(panel, v)
    return build_resolved(panel, v, sanity_disabled), panel, v
