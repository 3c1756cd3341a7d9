"""Copies the summaries tools/collect_profiles.sh left under gpurun_out/<round>/ into
profiles/<round>/ (tracked): bench line, kernel stats (batch launches, the other wave shapes, a
search), PMC summary, batch sweep, and the two derived files bench.py reads back
(traffic_b48.json: HBM-side bytes per launch; valu_b48.json: VALU / LDS pipe occupancy and lane
instructions per marker x point).
Usage: python tools/update_profiles.py r02"""
import csv, json, os, shutil, subprocess, sys
R = sys.argv[1] if len(sys.argv) > 1 else "r06"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from verifybamid_amd._abi import kernel_source_hash
# every derived file carries the hash of the kernel sources it was measured on (bench.py: a figure of another hash is dropped)
KHASH = kernel_source_hash()
_dump = json.dump


def _dump_with_hash(obj, fp, **kw):
    if isinstance(obj, dict):
        obj = dict(obj, kernel_src_hash=KHASH)
    return _dump(obj, fp, **kw)


json.dump = _dump_with_hash
EVAL_KERNELS = ("llk_eval_kernel", "llk_eval_split_kernel")
src, dst = os.path.join(ROOT, "gpurun_out", R), os.path.join(ROOT, "profiles", R)
os.makedirs(dst, exist_ok=True)
with open(os.path.join(src, "bench.json")) as f:
    line = [l for l in f if l.startswith("{")][-1]
bench = json.loads(line)
B = bench["config"]["batch_points_per_step"]
open(os.path.join(dst, "bench_b%d.json" % B), "w").write(line)
shutil.copy(os.path.join(src, "trace", "b_kernel_stats.csv"), os.path.join(dst, "bench_b%d_kernel_stats.csv" % B))
shutil.copy(os.path.join(src, "bench_batch_sweep.jsonl"), os.path.join(dst, "bench_batch_sweep.jsonl"))
for sub, pre, name in (("trace_modes", "m", "modes_kernel_stats.csv"), ("trace_opt", "o", "optimize_kernel_stats.csv")):
    p = os.path.join(src, sub, pre + "_kernel_stats.csv")
    if os.path.exists(p):
        shutil.copy(p, os.path.join(dst, name))
passes = [d for d in ("pmc_fetch", "pmc_write", "pmc_sq1", "pmc_sq2", "pmc_grbm") if os.path.isdir(os.path.join(src, d))]
out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py")] +
                     [os.path.join(src, d) for d in passes], capture_output=True, text=True).stdout
keep, on = [], False
for l in out.splitlines():
    if l.startswith("##"):
        on = any(kn in l for kn in EVAL_KERNELS)
    if on:
        keep.append(l)
open(os.path.join(dst, "bench_b%d_pmc_summary.txt" % B), "w").write(
    "# rocprofv3 --kernel-trace --pmc <counters> -- python bench.py --no-cpu-baseline --no-optimize --no-extras "
    "--steps 50 --warmup 20 --prewarm-ms 0  (one pass per counter group; tools/collect_profiles.sh)\n" +
    "\n".join(keep) + "\n")


def avg(d, name):
    vals = []
    p = [os.path.join(src, d, f) for f in os.listdir(os.path.join(src, d)) if f.endswith("counter_collection.csv")][0]
    for row in csv.DictReader(open(p)):
        if any(kn in row["Kernel_Name"] for kn in EVAL_KERNELS) and row["Counter_Name"] == name:
            vals.append(float(row["Counter_Value"]))
    return sum(vals) / len(vals)


markers, k = bench["config"]["active_markers"], 4
fetch, write = avg("pmc_fetch", "FETCH_SIZE"), avg("pmc_write", "WRITE_SIZE")
t = {"_what": "HBM-side traffic of llk_eval_kernel<2,true> per launch (%d points = %d groups of 8): rocprofv3 --pmc " % (B, B // 8) +
              "FETCH_SIZE and --pmc WRITE_SIZE, separate passes of `python bench.py --no-cpu-baseline "
              "--no-optimize --no-extras --steps 50` (bench_b%d_pmc_summary.txt; tools/collect_profiles.sh)" % B,
     "markers": markers, "batch": B, "num_pc": k,
     "FETCH_SIZE_KB": round(fetch, 1), "WRITE_SIZE_KB": round(write, 1),
     "correction": "gfx950: FETCH_SIZE tallies 64 B per 128-B request -> x2 (MI355X_MICROARCH.md, HBM); WRITE_SIZE uncorrected",
     "traffic_bytes_per_launch": int((2 * fetch + write) * 1024)}
json.dump(t, open(os.path.join(dst, "traffic_b%d.json" % B), "w"), indent=1)

insts, active = avg("pmc_sq1", "SQ_INSTS_VALU"), avg("pmc_sq1", "SQ_ACTIVE_INST_VALU")
lds_active, lds_conf = avg("pmc_sq1", "SQ_LDS_IDX_ACTIVE"), avg("pmc_sq1", "SQ_LDS_BANK_CONFLICT")
# rocprofv3 sums GRBM_GUI_ACTIVE over the 8 XCDs of the device: cycles of one launch = value / 8
# (197 k cycles for a ~90 us profiled launch = 2.2 GHz, a plausible shader clock; the raw sum is not)
cycles = avg("pmc_grbm", "GRBM_GUI_ACTIVE") / 8.0 if "pmc_grbm" in passes else None
v = {"_what": "VALU / LDS pipe occupancy of llk_eval_kernel<2,true> per launch of %d points, from the SQ and GRBM "
              "passes of bench_b%d_pmc_summary.txt: lane instructions = SQ_INSTS_VALU x 64 / (markers x points); "
              "VALU busy = SQ_ACTIVE_INST_VALU (quad-cycles) x 4 / (1024 SIMDs x GRBM_GUI_ACTIVE cycles of the "
              "launch, GRBM_GUI_ACTIVE being summed over the 8 XCDs -> / 8); LDS busy = SQ_LDS_IDX_ACTIVE / (256 CUs x cycles). "
              "Profiled launches run ~15 %% slower than un-profiled ones, so both fractions read low by about that much" % (B, B),
     "markers": markers, "batch": B, "num_pc": k,
     "SQ_INSTS_VALU": insts, "SQ_ACTIVE_INST_VALU": active, "SQ_LDS_IDX_ACTIVE": lds_active,
     "SQ_LDS_BANK_CONFLICT": lds_conf, "cycles_per_launch_profiled": cycles,
     "lane_instr_per_marker_point": round(insts * 64 / (markers * B), 1),
     "valu_busy_frac": round(active * 4 / (1024 * cycles), 3) if cycles else None,
     "lds_busy_frac": round(lds_active / (256 * cycles), 3) if cycles else None}
json.dump(v, open(os.path.join(dst, "valu_b%d.json" % B), "w"), indent=1)

# FP64 vector instructions of the launch by class (wave-instructions; x 64 lanes; an FMA = 2 flops)
if os.path.isdir(os.path.join(src, "pmc_f64")):
    try:
        fa, fm = avg("pmc_f64", "SQ_INSTS_VALU_ADD_F64"), avg("pmc_f64", "SQ_INSTS_VALU_MUL_F64")
        ff, ft = avg("pmc_f64", "SQ_INSTS_VALU_FMA_F64"), avg("pmc_f64", "SQ_INSTS_VALU_TRANS_F64")
        fl = {"_what": "FP64 vector instructions of llk_eval_kernel<2,true> per launch of %d points by class: rocprofv3 --pmc "
                       "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 (wave-instructions; "
                       "flops = (add + mul + trans + 2 x fma) x 64 lanes) -- the numerator of bench.py's roofline.fp64" % B,
              "markers": markers, "batch": B, "num_pc": k,
              "SQ_INSTS_VALU_ADD_F64": fa, "SQ_INSTS_VALU_MUL_F64": fm, "SQ_INSTS_VALU_FMA_F64": ff, "SQ_INSTS_VALU_TRANS_F64": ft,
              "fp64_instr_per_marker_point": round((fa + fm + ff + ft) * 64 / (markers * B), 1),
              "fp64_flops_per_launch": (fa + fm + ft + 2 * ff) * 64,
              "fp64_flops_per_marker_point": round((fa + fm + ft + 2 * ff) * 64 / (markers * B), 1)}
        json.dump(fl, open(os.path.join(dst, "flops_b%d.json" % B), "w"), indent=1)
        print("FP64 instructions per marker x point: %.1f (add %.1f mul %.1f fma %.1f trans %.1f), %.1f flops" % (
            fl["fp64_instr_per_marker_point"], fa * 64 / (markers * B), fm * 64 / (markers * B), ff * 64 / (markers * B),
            ft * 64 / (markers * B), fl["fp64_flops_per_marker_point"]))
    except Exception as exc:      # noqa: BLE001 -- a counter the box does not have: the other summaries still go out
        print("pmc_f64: %s" % exc)

# cohort steps: FETCH_SIZE per launch of llk_eval_multi_kernel<MODE, ...> (MODE 4: 1 point per sample, 5: 2, 3: 4, 2: 8)
cm = os.path.join(src, "pmc_modes_fetch")
if os.path.isdir(cm):
    import re
    pth = [os.path.join(cm, f) for f in os.listdir(cm) if f.endswith("counter_collection.csv")][0]
    acc = {}
    for row in csv.DictReader(open(pth)):
        mm = re.search(r"llk_eval_multi_kernel<(\d), (true|false)[^>]*>", row["Kernel_Name"])
        if mm and row["Counter_Name"] == "FETCH_SIZE":
            acc.setdefault((int(mm.group(1)), mm.group(2) == "true"), []).append(float(row["Counter_Value"]))
    np_of = {4: 1, 5: 2, 3: 4, 2: 8}
    ct = {"_what": "HBM-side bytes of one lock-step cohort step of 32 C3-shaped samples, by points per sample: rocprofv3 --pmc "
                   "FETCH_SIZE over tools/prof_modes.py, x2 (gfx950: FETCH_SIZE tallies 64 B per 128-B request), per launch of "
                   "llk_eval_multi_kernel<MODE, true, W16> (MODE 4: 1 point, 5: 2, 3: 4, 2: 8; W16: the 16-bit run lists)",
          "samples": int(os.environ.get("VB2_COHORT", "32")), "markers": markers, "traffic_bytes_per_step": {}, "kernels": {}}
    for (mode, w16), vals in sorted(acc.items()):
        b = int(2 * 1024 * sum(vals) / len(vals))
        ct["traffic_bytes_per_step"][str(np_of[mode])] = b
        ct["kernels"][str(np_of[mode])] = "llk_eval_multi_kernel<%d, true, %s>, %d launches" % (mode, "true" if w16 else "false", len(vals))
    json.dump(ct, open(os.path.join(dst, "cohort_traffic.json"), "w"), indent=1)
    print("cohort step traffic (bytes):", ct["traffic_bytes_per_step"])

# ---- round 4: the cohort steps of a search (1 / 2 points per sample), the wide alphabet, the round timeline ----
def first_csv(d, suffix):
    dd = os.path.join(src, d)
    if not os.path.isdir(dd):
        return None
    c = [os.path.join(dd, f) for f in os.listdir(dd) if f.endswith(suffix)]
    return c[0] if c else None


def avg_of(path, kernel_sub, name):
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path))
            if kernel_sub in r["Kernel_Name"] and r["Counter_Name"] == name]
    return sum(vals) / len(vals) if vals else None


ks = first_csv("trace_cohort", "kernel_stats.csv")
if ks:
    shutil.copy(ks, os.path.join(dst, "cohort_kernel_stats.csv"))
    cp = {"_what": "the steps of a cohort search -- 32 C3-shaped samples, 1 and 2 points per sample: llk_eval_multi_kernel<4, true, true, 4, 1> "
                   "and <5, ...> (16-bit run lists, --NumPC 4, static deal with the pipelined item loop) -- under rocprofv3 --pmc "
                   "(tools/prof_cohort.py; one pass per counter group): VALU busy = SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x cycles), "
                   "cycles = GRBM_GUI_ACTIVE / 8; LDS busy = SQ_LDS_IDX_ACTIVE / (256 CUs x cycles); parked = SQ_WAIT_ANY / "
                   "SQ_WAVE_CYCLES (waves in s_waitcnt or at a barrier); traffic = FETCH_SIZE x 2 (gfx950 correction)",
          "samples": 32, "markers": markers, "kernels": {}}
    for mode, npnt in ((4, 1), (5, 2)):
        sub = "llk_eval_multi_kernel<%d," % mode
        sq1, sq2 = first_csv("pmc_cohort_sq1", "counter_collection.csv"), first_csv("pmc_cohort_sq2", "counter_collection.csv")
        fe, gr = first_csv("pmc_cohort_fetch", "counter_collection.csv"), first_csv("pmc_cohort_grbm", "counter_collection.csv")
        if not (sq1 and sq2 and fe and gr):
            continue
        cyc = avg_of(gr, sub, "GRBM_GUI_ACTIVE") / 8.0
        iv, av = avg_of(sq1, sub, "SQ_INSTS_VALU"), avg_of(sq1, sub, "SQ_ACTIVE_INST_VALU")
        cp["kernels"][str(npnt)] = {
            "points_per_sample": npnt, "cycles_per_step_profiled": cyc,
            "lane_instr_per_marker_point": round(iv * 64 / (32.0 * markers * npnt), 1),
            "valu_busy_frac": round(av * 4 / (1024 * cyc), 3),
            "lds_busy_frac": round(avg_of(sq1, sub, "SQ_LDS_IDX_ACTIVE") / (256 * cyc), 3),
            "lds_bank_conflict_frac": round(avg_of(sq1, sub, "SQ_LDS_BANK_CONFLICT") / max(1.0, avg_of(sq1, sub, "SQ_LDS_IDX_ACTIVE")), 4),
            "parked_frac": round(avg_of(sq2, sub, "SQ_WAIT_ANY") / avg_of(sq1, sub, "SQ_WAVE_CYCLES"), 3),
            "issue_stall_frac": round(avg_of(sq2, sub, "SQ_WAIT_INST_ANY") / avg_of(sq1, sub, "SQ_WAVE_CYCLES"), 3),
            "traffic_bytes_per_step": int(2 * 1024 * avg_of(fe, sub, "FETCH_SIZE"))}
    json.dump(cp, open(os.path.join(dst, "cohort_pmc.json"), "w"), indent=1)
    # bench.py reads cohort_traffic.json: refresh the 1- and 2-point entries from this pass
    ctp = os.path.join(dst, "cohort_traffic.json")
    ct2 = json.load(open(ctp)) if os.path.exists(ctp) else {"samples": 32, "markers": markers, "traffic_bytes_per_step": {}, "kernels": {}}
    for npnt, kk in cp["kernels"].items():
        ct2["traffic_bytes_per_step"][npnt] = kk["traffic_bytes_per_step"]
    json.dump(ct2, open(ctp, "w"), indent=1)
    print("cohort steps:", {n: (kk["valu_busy_frac"], kk["lds_busy_frac"], kk["parked_frac"], kk["traffic_bytes_per_step"]) for n, kk in cp["kernels"].items()})

for aname, qlo, qhi in (("wide", 2, 60), ("mid", 10, 45)):
    ws = first_csv("trace_%s" % aname, "kernel_stats.csv")
    if ws:
        shutil.copy(ws, os.path.join(dst, "bench_b%d_%s_kernel_stats.csv" % (B, aname)))
        sq1, gr = first_csv("pmc_%s_sq1" % aname, "counter_collection.csv"), first_csv("pmc_%s_grbm" % aname, "counter_collection.csv")
        if sq1 and gr:
            sub = "llk_eval_passes_kernel"          # (one launch of two passes of 24 points since round 4; VB2_PASSES=0: llk_eval_kernel x 3)
            if avg_of(gr, sub, "GRBM_GUI_ACTIVE") is None:
                sub = "llk_eval_split_kernel"       # (round 6, probability domain: one launch, the groups split between workgroup pairs)
            if avg_of(gr, sub, "GRBM_GUI_ACTIVE") is None:
                sub = "llk_eval_kernel"
            cyc = avg_of(gr, sub, "GRBM_GUI_ACTIVE") / 8.0
            iv = avg_of(sq1, sub, "SQ_INSTS_VALU")
            # (a %d-point call on this alphabet is several launches: per-launch counters, points per launch from the instruction count's
            # ratio is not needed -- lane instructions are per marker x point of the launch's own points)
            launches = len([1 for r in csv.DictReader(open(sq1)) if sub in r["Kernel_Name"] and r["Counter_Name"] == "SQ_INSTS_VALU"])
            calls = 50 + 20
            ppl = B / max(1.0, round(launches / float(calls)))
            wv = {"_what": "the headline call on base qualities %d..%d (a dictionary this wide: three point groups' tables fit the LDS "
                           "beside a compact exp table, a 48-point call is ONE launch of two passes -- llk_eval_passes_kernel): SQ / "
                           "GRBM passes of `bench.py --q-lo %d --q-hi %d --no-extras`, per LAUNCH" % (qlo, qhi, qlo, qhi), "kernel": sub,
                  "markers": markers, "batch": B, "num_pc": k, "points_per_launch": ppl,
                  "lane_instr_per_marker_point": round(iv * 64 / (markers * ppl), 1),
                  "valu_busy_frac": round(avg_of(sq1, sub, "SQ_ACTIVE_INST_VALU") * 4 / (1024 * cyc), 3),
                  "lds_busy_frac": round(avg_of(sq1, sub, "SQ_LDS_IDX_ACTIVE") / (256 * cyc), 3),
                  "lds_bank_conflict_frac": round(avg_of(sq1, sub, "SQ_LDS_BANK_CONFLICT") / max(1.0, avg_of(sq1, sub, "SQ_LDS_IDX_ACTIVE")), 4)}
            json.dump(wv, open(os.path.join(dst, "valu_b%d_%s.json" % (B, aname)), "w"), indent=1)
            print(aname, "alphabet:", wv["points_per_launch"], wv["lane_instr_per_marker_point"], wv["valu_busy_frac"], wv["lds_busy_frac"])
cs = first_csv("trace_create", "kernel_stats.csv")
if cs:
    shutil.copy(cs, os.path.join(dst, "create_kernel_stats.csv"))
    cj = {"_what": "device time of vb2_ctx_create's kernels on a C3 sample (100 000 markers x depth 30), average per call in us: "
                   "rocprofv3 --kernel-trace --stats over tools/create_time.py (contexts of both alphabets -- 42 and 118 codes -- "
                   "in the three flatten modes; pack_sched_kernel runs for the 118-code contexts only)"}
    for r in csv.DictReader(open(cs)):
        nm = r["Name"].split("(")[0].replace("void ", "").replace("vb2::", "").split("<")[0]
        if nm in ("classify_kernel", "pack_layout_kernel", "pack_sched_kernel", "pack_codes16_kernel"):
            cj[nm + "_us"] = round(float(r["AverageNs"]) / 1e3, 1)
    json.dump(cj, open(os.path.join(dst, "create_kernel_stats.json"), "w"), indent=1)
for f in ("create_time.txt", "ubench_lds_fma_mix.txt", "ubench_valu_rates.txt"):
    if os.path.exists(os.path.join(src, f)):
        open(os.path.join(dst, f), "w").write("".join(l for l in open(os.path.join(src, f)) if "amdgpu.ids" not in l))
for f in ("search_round_stamps.txt", "search_round200_stamps.txt"):
    if os.path.exists(os.path.join(src, f)):
        open(os.path.join(dst, f), "w").write("".join(l for l in open(os.path.join(src, f)) if "amdgpu.ids" not in l))

print("value %.0f evals/s, %.2f us/launch, frac %.3f, optimize %.2f ms, traffic %d B/launch, %s lane instr per marker x point, "
      "VALU busy %s, LDS busy %s" % (bench["value"], bench["roofline"]["device_us_per_launch"], bench["roofline"]["frac"] or 0.0,
                                     bench["optimize"]["wall_ms_to_converged_alpha"], t["traffic_bytes_per_launch"],
                                     v["lane_instr_per_marker_point"], v["valu_busy_frac"], v["lds_busy_frac"]))
print(open(os.path.join(dst, "bench_b%d_kernel_stats.csv" % B)).read().splitlines()[1][:200])
for l in open(os.path.join(dst, "bench_batch_sweep.jsonl")):
    r = json.loads(l)
    print(r["config"]["batch_points_per_step"], round(r["roofline"]["device_us_per_launch"], 2), round(r["value"]),
          round(r["roofline"]["frac"] or 0.0, 3))
