"""Turns the rocprofv3 CSVs that a gpurun call left under gpurun_out/ into the tracked summaries in profiles/.
usage: summarize_profiles.py <round-tag>   (e.g. r01)"""
import csv, collections, json, os, sys, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
out = os.path.join(ROOT, "profiles"); os.makedirs(out, exist_ok=True)
ks = os.path.join(ROOT, "gpurun_out", "prof_bench", "bench_kernel_stats.csv")
if os.path.exists(ks):
    rows = list(csv.DictReader(open(ks)))
    with open(os.path.join(out, f"{tag}_bench_kernel_stats.csv"), "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline (1x MI355X)\n")
        w = csv.writer(f); w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows:
            w.writerow([r["Name"][:140], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
def load(p):
    d = collections.OrderedDict()
    if not os.path.exists(p): return d
    for r in csv.DictReader(open(p)):
        d.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    return d
f = load(os.path.join(ROOT, "gpurun_out", "pmc_fetch", "f_counter_collection.csv"))
w = load(os.path.join(ROOT, "gpurun_out", "pmc_write", "w_counter_collection.csv"))
summ = {"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on tools/pmc_probe.py, 1 GiB buffer per launch. "
                "Counter unit = KiB. gfx950 correction (MI355X_MICROARCH.md §HBM): FETCH_SIZE counts 1/2 of wide coalesced reads -> doubled.",
        "kernels": {}}
for k in f:
    if "vkfft" not in k: continue
    fe = sum(f[k]) / len(f[k]) * 1024 * 2  # bytes, corrected x2
    wr = sum(w.get(k, [0])) / max(len(w.get(k, [0])), 1) * 1024
    summ["kernels"][k[:120]] = dict(launches=len(f[k]), fetch_bytes_corrected=fe, write_bytes=wr, hbm_bytes_per_launch=fe + wr,
                                    algorithmic_bytes_per_pass=2.0 * (1 << 30))
dom = [k for k in summ["kernels"] if "pow2_col_kernel" in k and "4, 3, 3, 0>, 16" in k] or [k for k in summ["kernels"] if "pow2_col_kernel" in k]
if dom:
    summ["hbm_bytes_per_launch"] = summ["kernels"][dom[0]]["hbm_bytes_per_launch"]
    summ["dominant_kernel"] = dom[0]
json.dump(summ, open(os.path.join(out, f"{tag}_pmc_traffic.json"), "w"), indent=1)
print(json.dumps(summ, indent=1)[:1500])
