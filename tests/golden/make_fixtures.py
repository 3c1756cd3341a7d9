#!/usr/bin/env python
"""Regenerates the committed golden vectors of tests/golden/ -- the recipe behind the numbers.

    python tests/golden/make_fixtures.py            # rewrite kat.json's LLK values + synthetic_c2/c3.json
    python tests/golden/make_fixtures.py --check    # recompute and compare with what is committed

Run in the BUILD container (it uses the oracle and, for the search half, oracle/_ref: the
reference's own AmoebaMinimizer compiled in place from /root/reference).  Nothing here runs on
the GPU box; the tests read the JSON files.

What pins what:
  * tests/golden/expected/* and the two pileups are byte-identical copies of the reference's own
    CTest fixtures (resource/test/expected/*, CMakeLists.txt:86-147).  tests/test_oracle_golden.py
    holds the ORACLE (oracle/vb2_oracle.c) to them: six .Ancestry files byte for byte, both
    .selfSM files, 604 evaluations -- and holds its simplex search to the reference's compiled
    AmoebaMinimizer (oracle/_ref) bit for bit.
  * kat.json `llk_hex`: +LLK of ComputeMixLLKs at five fixed points on the two bundled inputs.
    The values were first captured from a build of the reference's ContaminationEstimator.h
    (SURVEY.md 8c); this script RECOMPUTES them with the pinned oracle and refuses to write if
    they do not reproduce bit for bit -- the committed numbers are a function of committed code
    and committed data, not a transcription.
  * synthetic_c2.json / synthetic_c3.json (BASELINE.json configs[1] and [2] shapes): seeded
    synthetic pileups (verifybamid_amd/synth.py), sha256 of the generated arrays, +LLK hex-floats
    of the oracle (1 thread: a fixed summation order) at seeded points, and the OptimizeLLK result
    -- alpha, llk1, llk0, evaluation count, a digest of the whole evaluation trace -- computed
    twice: with the oracle's minimiser and with the reference's own (oracle/_ref); the two must
    agree exactly or nothing is written.
"""
import argparse
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HAPMAP = os.path.join(HERE, "hapmap", "hapmap_3.3.b37.dat")
SHAPES = {
    "synthetic_c2.json": dict(markers=10000, depth=30, num_pc=2, alpha_true=0.05, seed=1, points=12),
    "synthetic_c3.json": dict(markers=100000, depth=30, num_pc=4, alpha_true=0.05, seed=2, points=8),
}


def sha(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def trace_digest(tr):
    return sha(tr["alpha"], tr["pc1"], tr["pc2"], tr["llk"])


def synthetic_fixture(spec):
    import verifybamid_amd as vb
    from oracle.bridge import oracle_data
    from oracle import binding
    k = spec["num_pc"]
    d = vb.synth.make_pileup(spec["markers"], spec["depth"], k, alpha_true=spec["alpha_true"], seed=spec["seed"])
    od = oracle_data(d)
    rng = np.random.default_rng(1000 + spec["seed"])
    B = spec["points"]
    pc1 = rng.normal(0, 0.03, size=(B, k))
    pc2 = rng.normal(0, 0.03, size=(B, k))
    alpha = rng.uniform(0.0, 0.5, size=B)
    alpha[0] = 0.0
    llk = [od.llk(pc1[i], pc2[i], alpha[i], num_thread=1) for i in range(B)]
    out = dict(
        _what="seeded synthetic pileup (verifybamid_amd.synth.make_pileup) + oracle results; "
              "regenerate with tests/golden/make_fixtures.py",
        generator=dict(markers=spec["markers"], mean_depth=spec["depth"], num_pc=k,
                       alpha_true=spec["alpha_true"], seed=spec["seed"]),
        input_sha256=sha(d.ud, d.means, d.read_off, d.bases, d.quals, d.alt_base),
        reads=int(d.num_read),
        points=dict(pc1=pc1.tolist(), pc2=pc2.tolist(), alpha=alpha.tolist()),
        llk_hex=[float(x).hex() for x in llk],
        llk_threads=1,
        models={},
    )
    models = {"heter": {}, "within": {"within_ancestry": True}} if spec["markers"] <= 20000 else {"heter": {}}
    for name, kw in models.items():
        a = od.optimize(num_thread=1, minimizer="oracle", trace_capacity=1 << 14, **kw)
        have_ref = binding.ref_lib() is not None
        if have_ref:
            b = od.optimize(num_thread=1, minimizer="reference", trace_capacity=1 << 14, **kw)
            same = (a["alpha"] == b["alpha"] and a["llk1"] == b["llk1"] and a["llk0"] == b["llk0"] and
                    a["num_eval"] == b["num_eval"] and trace_digest(a["trace"]) == trace_digest(b["trace"]))
            if not same:
                raise SystemExit("oracle minimiser and the reference's AmoebaMinimizer disagree on %s" % name)
        out["models"][name] = dict(
            args=kw, alpha_hex=float(a["alpha"]).hex(), llk1_hex=float(a["llk1"]).hex(),
            llk0_hex=float(a["llk0"]).hex(), num_eval=a["num_eval"],
            pc_hex=[float(x).hex() for x in a["pc"]], pc2_hex=[float(x).hex() for x in a["pc2"]],
            trace_sha256=trace_digest(a["trace"]),
            trace_head_llk_hex=[float(x).hex() for x in a["trace"]["llk"][:6]],
            cross_checked_with_reference_minimiser=have_ref)
    return out


PANELS = os.path.join(HERE, "panels")
REAL = {
    # BASELINE.json configs[0] names this panel; configs[2] is the shape of its 100k sibling
    "10k_k2": dict(panel="1000g.phase3.10k.b37.vcf.gz.dat", num_pc=2, depth=30, alpha_true=0.05, seed=11, points=8),
    "10k_k4": dict(panel="1000g.phase3.10k.b37.vcf.gz.dat", num_pc=4, depth=30, alpha_true=0.08, seed=12, points=8),
    # (round 4: the other model branches of ContaminationEstimator.cpp:98-150 at this size too -- --WithinAncestry on the
    # 100k panel, and --KnownAF, which main.cpp:314-319 turns into a one-parameter search over alpha)
    "100k_k4": dict(panel="1000g.phase3.100k.b37.vcf.gz.dat", num_pc=4, depth=30, alpha_true=0.03, seed=13, points=3,
                    extra_models=("within", "known_af")),
}
KNOWN_AF_SEED = 77


def file_sha(path):
    """sha256 of a panel file's CONTENT (gzip'd copies are inflated first: the 100k panel is committed gzip'd)."""
    import gzip
    raw = open(path, "rb").read()
    if raw[:2] == b"\x1f\x8b":
        raw = gzip.decompress(raw)
    return hashlib.sha256(raw).hexdigest()


def real_panel_fixture(spec, tmpdir):
    """Reads drawn on the REFERENCE'S OWN bundled panel (real U.D spectra, real mean genotypes, real
    alleles incl. the multi-allelic `A,G` rows of which the reference keeps the first character,
    ContaminationEstimator.cpp:417,428-429), written as a text pileup and read back through the file
    readers with the sanity check ON (the +-3 sd depth filter of h:246-249 is active)."""
    import verifybamid_amd as vb
    from oracle import binding, refio
    from oracle.bridge import oracle_data
    k = spec["num_pc"]
    prefix = os.path.join(PANELS, spec["panel"])
    ref_prefix = os.path.join("/root/reference/resource", spec["panel"])
    shas = {}
    for ext in ("UD", "mu", "bed"):
        shas[ext] = file_sha(prefix + "." + ext)
        if os.path.exists(ref_prefix + "." + ext) and file_sha(ref_prefix + "." + ext) != shas[ext]:
            raise SystemExit("%s.%s is not the reference's file" % (prefix, ext))
    pile = vb.synth.real_panel_sample(prefix, os.path.join(tmpdir, "real.pileup"), spec["depth"], spec["alpha_true"],
                                      spec["seed"])
    flat, panel, viewer = refio.load_flat(prefix, pile, k, sanity_disabled=False)     # the reference's readers, restated
    od = binding.OracleData(flat)
    d = vb.PileupData.from_files(prefix, pile, k, disable_sanity=False)               # the product's readers (host only)
    od2 = oracle_data(d)
    rng = np.random.default_rng(2000 + spec["seed"])
    B = spec["points"]
    pc1 = rng.normal(0, 0.01, size=(B, k))
    pc2 = rng.normal(0, 0.01, size=(B, k))
    alpha = rng.uniform(0.0, 0.5, size=B)
    alpha[0] = 0.0
    llk = [od.llk(pc1[i], pc2[i], alpha[i], num_thread=1) for i in range(B)]
    llk2 = [od2.llk(pc1[i], pc2[i], alpha[i], num_thread=1) for i in range(B)]
    if llk != llk2:
        raise SystemExit("the product's file readers and the restated reference readers disagree on %s" % spec["panel"])
    multi = sum(1 for line in refio._newline_terminated_lines(prefix + ".bed") if "," in line.split()[4])
    out = dict(
        panel=spec["panel"], panel_sha256=shas, multi_allelic_alt_rows=multi,
        generator=dict(mean_depth=spec["depth"], num_pc=k, alpha_true=spec["alpha_true"], seed=spec["seed"]),
        input_sha256=sha(d.ud, d.means, d.read_off, d.bases, d.quals, d.alt_base),
        reads=int(d.num_read), avg_depth_hex=float(d.avg_depth).hex(), sd_depth_hex=float(d.sd_depth).hex(),
        points=dict(pc1=pc1.tolist(), pc2=pc2.tolist(), alpha=alpha.tolist()),
        llk_hex=[float(x).hex() for x in llk], llk_threads=1, models={})
    a = od.optimize(num_thread=1, minimizer="oracle", trace_capacity=1 << 14)
    have_ref = binding.ref_lib() is not None
    if have_ref:
        b = od.optimize(num_thread=1, minimizer="reference", trace_capacity=1 << 14)
        if not (a["alpha"] == b["alpha"] and a["llk1"] == b["llk1"] and a["llk0"] == b["llk0"] and
                a["num_eval"] == b["num_eval"] and trace_digest(a["trace"]) == trace_digest(b["trace"])):
            raise SystemExit("oracle minimiser and the reference's AmoebaMinimizer disagree on %s" % spec["panel"])
    out["models"]["heter"] = dict(
        args={}, alpha_hex=float(a["alpha"]).hex(), llk1_hex=float(a["llk1"]).hex(), llk0_hex=float(a["llk0"]).hex(),
        num_eval=a["num_eval"], pc_hex=[float(x).hex() for x in a["pc"]], pc2_hex=[float(x).hex() for x in a["pc2"]],
        trace_sha256=trace_digest(a["trace"]), trace_head_llk_hex=[float(x).hex() for x in a["trace"]["llk"][:6]],
        cross_checked_with_reference_minimiser=have_ref)
    def model_record(od_, args, label):
        a_ = od_.optimize(num_thread=1, minimizer="oracle", trace_capacity=1 << 14, **args)
        if have_ref:
            b_ = od_.optimize(num_thread=1, minimizer="reference", trace_capacity=1 << 14, **args)
            if not (a_["alpha"] == b_["alpha"] and a_["llk1"] == b_["llk1"] and a_["llk0"] == b_["llk0"] and
                    a_["num_eval"] == b_["num_eval"] and trace_digest(a_["trace"]) == trace_digest(b_["trace"])):
                raise SystemExit("oracle minimiser and the reference's AmoebaMinimizer disagree on %s (%s)" % (spec["panel"], label))
        return dict(args=args, alpha_hex=float(a_["alpha"]).hex(), llk1_hex=float(a_["llk1"]).hex(),
                    llk0_hex=float(a_["llk0"]).hex(), num_eval=a_["num_eval"], pc_hex=[float(x).hex() for x in a_["pc"]],
                    pc2_hex=[float(x).hex() for x in a_["pc2"]], trace_sha256=trace_digest(a_["trace"]),
                    trace_head_llk_hex=[float(x).hex() for x in a_["trace"]["llk"][:6]],
                    cross_checked_with_reference_minimiser=have_ref)
    if "within" in spec.get("extra_models", ()):
        out["models"]["within"] = model_record(od, {"within_ancestry": True}, "within")
    if "known_af" in spec.get("extra_models", ()):
        afp = vb.synth.write_known_af(prefix, os.path.join(tmpdir, "real.af"), KNOWN_AF_SEED)
        flat_k, _, _ = refio.load_flat(prefix, pile, k, sanity_disabled=False, known_af_path=afp)
        if not flat_k.af_known:
            raise SystemExit("the known-AF file was not taken up")
        od_k = binding.OracleData(flat_k)
        rec = model_record(od_k, {}, "known_af")
        rec["known_af_seed"] = KNOWN_AF_SEED
        rec["known_af_sha256"] = hashlib.sha256(open(afp, "rb").read()).hexdigest()
        rec["llk_hex"] = [float(od_k.llk(pc1[i], pc2[i], alpha[i], num_thread=1)).hex() for i in range(B)]
        out["models"]["known_af"] = rec
    return out


def kat_llk():
    """The five known-answer points on the two bundled inputs, recomputed with the oracle."""
    from oracle import binding, refio
    kat = json.load(open(os.path.join(HERE, "kat.json")))
    got = {}
    for name, spec in kat["inputs"].items():
        flat, _, _ = refio.load_flat(HAPMAP, os.path.join(HERE, spec["pileup"]), 2)
        od = binding.OracleData(flat)
        got[name] = [float(od.llk(p["pc1"], p["pc2"], p["alpha"], num_thread=1)).hex() for p in kat["points"]]
    return kat, got


def norm_hex(h):
    return float.fromhex(h).hex()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    args = ap.parse_args()
    bad = 0
    kat, got = kat_llk()
    for name, vals in got.items():
        want = [norm_hex(h) for h in kat["inputs"][name]["llk_hex"]]
        if vals != want:
            print("kat.json %s: oracle gives %s, file holds %s" % (name, vals, want))
            bad += 1
    if bad:
        raise SystemExit("kat.json does not reproduce from the oracle: not writing anything")
    if not args.check:
        kat["_source"] = ("Known-answer +LLK values of the reference's FullLLKFunc::ComputeMixLLKs "
                          "(ContaminationEstimator.h:194-314): hapmap_3.3.b37.dat panel, --NumPC 2, sanity check "
                          "disabled; first captured from the reference (SURVEY.md 8c), reproduced bit for bit by "
                          "tests/golden/make_fixtures.py from the oracle that tests/test_oracle_golden.py pins to "
                          "the reference's own fixtures. Hex-floats are the exact IEEE-754 doubles.")
        json.dump(kat, open(os.path.join(HERE, "kat.json"), "w"), indent=1)
    for fname, spec in SHAPES.items():
        fx = synthetic_fixture(spec)
        path = os.path.join(HERE, fname)
        if args.check:
            old = json.load(open(path))
            if old != json.loads(json.dumps(fx)):
                print("%s differs from what the recipe produces now" % fname)
                bad += 1
        else:
            json.dump(fx, open(path, "w"), indent=1)
            print("wrote %s (%d points, models %s)" % (fname, len(fx["llk_hex"]), sorted(fx["models"])))
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        real = dict(_what="reads drawn (verifybamid_amd.synth.real_panel_sample) on the reference's bundled 1000g.phase3 "
                          "panels (tests/golden/panels/: byte-identical data files, the 100k ones gzip'd), read back "
                          "through the file readers with the sanity check on; oracle results; regenerate with "
                          "tests/golden/make_fixtures.py",
                    cases={name: real_panel_fixture(spec, tmp) for name, spec in REAL.items()})
    path = os.path.join(HERE, "real_panel.json")
    if args.check:
        if json.load(open(path)) != json.loads(json.dumps(real)):
            print("real_panel.json differs from what the recipe produces now")
            bad += 1
    else:
        json.dump(real, open(path, "w"), indent=1)
        print("wrote real_panel.json (%s)" % ", ".join("%s: alpha %.6f, %d evals" % (
            n, float.fromhex(c["models"]["heter"]["alpha_hex"]), c["models"]["heter"]["num_eval"])
            for n, c in real["cases"].items()))
    if bad:
        raise SystemExit(1)
    print("fixtures %s" % ("reproduce" if args.check else "written"))


if __name__ == "__main__":
    main()
