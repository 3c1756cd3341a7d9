"""Copies the summaries tools/collect_profiles.sh left under gpurun_out/<round>/ into
profiles/<round>/ (tracked): bench line, kernel stats, PMC summary, batch sweep, traffic file.
Usage: python tools/update_profiles.py r01"""
import csv, collections, io, json, os, shutil, subprocess, sys
R = sys.argv[1] if len(sys.argv) > 1 else "r01"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(ROOT, "gpurun_out", R), os.path.join(ROOT, "profiles", R)
os.makedirs(dst, exist_ok=True)
with open(os.path.join(src, "bench.json")) as f:
    line = [l for l in f if l.startswith("{")][-1]
B = json.loads(line)["config"]["batch_points_per_step"]
open(os.path.join(dst, "bench_b%d.json" % B), "w").write(line)
shutil.copy(os.path.join(src, "trace", "b_kernel_stats.csv"), os.path.join(dst, "bench_b%d_kernel_stats.csv" % B))
shutil.copy(os.path.join(src, "bench_batch_sweep.jsonl"), os.path.join(dst, "bench_batch_sweep.jsonl"))
out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py")] +
                     [os.path.join(src, d) for d in ("pmc_fetch", "pmc_write", "pmc_sq1", "pmc_sq2")],
                     capture_output=True, text=True).stdout
keep, on = [], False
for l in out.splitlines():
    if l.startswith("##"):
        on = "llk_eval_kernel" in l
    if on:
        keep.append(l)
open(os.path.join(dst, "bench_b%d_pmc_summary.txt" % B), "w").write(
    "# rocprofv3 --kernel-trace --pmc <counters> -- python bench.py --no-cpu-baseline --no-optimize --steps 50"
    "  (one pass per counter group; tools/collect_profiles.sh)\n" + "\n".join(keep) + "\n")

def avg(d, name):
    vals = []
    p = [os.path.join(src, d, f) for f in os.listdir(os.path.join(src, d)) if f.endswith("counter_collection.csv")][0]
    for row in csv.DictReader(open(p)):
        if "llk_eval_kernel" in row["Kernel_Name"] and row["Counter_Name"] == name:
            vals.append(float(row["Counter_Value"]))
    return sum(vals) / len(vals)

fetch, write = avg("pmc_fetch", "FETCH_SIZE"), avg("pmc_write", "WRITE_SIZE")
b = json.loads(line)
t = {"_what": "HBM-side traffic of llk_eval_kernel<2,true> per launch (%d points = %d groups of 8): rocprofv3 --pmc " % (B, B // 8) +
              "FETCH_SIZE and --pmc WRITE_SIZE, separate passes of `python bench.py --no-cpu-baseline "
              "--no-optimize --steps 50` (bench_b%d_pmc_summary.txt; tools/collect_profiles.sh)" % B,
     "markers": 100000, "batch": B, "num_pc": 4,
     "FETCH_SIZE_KB": round(fetch, 1), "WRITE_SIZE_KB": round(write, 1),
     "correction": "gfx950: FETCH_SIZE tallies 64 B per 128-B request -> x2 (MI355X_MICROARCH.md, HBM); WRITE_SIZE uncorrected",
     "traffic_bytes_per_launch": int((2 * fetch + write) * 1024)}
json.dump(t, open(os.path.join(dst, "traffic_b%d.json" % B), "w"), indent=1)
print("value %.0f evals/s, %.2f us/launch, frac %.3f, optimize %.2f ms, traffic %d B/launch" % (
    b["value"], b["roofline"]["device_us_per_launch"], b["roofline"]["frac"],
    b["optimize"]["wall_ms_to_converged_alpha"], t["traffic_bytes_per_launch"]))
print(open(os.path.join(dst, "bench_b%d_kernel_stats.csv" % B)).read().splitlines()[1][:200])
for l in open(os.path.join(dst, "bench_batch_sweep.jsonl")):
    r = json.loads(l)
    print(r["config"]["batch_points_per_step"], round(r["roofline"]["device_us_per_launch"], 2), round(r["value"]),
          round(r["roofline"]["frac"], 3))
