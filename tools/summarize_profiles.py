"""Turns the rocprofv3 CSVs that a gpurun call left under gpurun_out/ into the tracked summaries in profiles/.
usage: summarize_profiles.py <round-tag> <gpurun_out subdir> [name of the kernel-stats summary] [command line it came from]
(e.g. r03 r03/prof_bench bench_kernel_stats "python bench.py --steps 3 --warmup 1 --no-cpu-baseline")"""
import csv, collections, json, os, sys, glob
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
sub = sys.argv[2] if len(sys.argv) > 2 else "r2prof"
ksname = sys.argv[3] if len(sys.argv) > 3 else "bench_kernel_stats"
kscmd = sys.argv[4] if len(sys.argv) > 4 else "python bench.py --steps 3 --warmup 1 --no-cpu-baseline"
sys.path.insert(0, ROOT)
from vkfft_amd import api
base = os.path.join(ROOT, "gpurun_out", sub)
out = os.path.join(ROOT, "profiles"); os.makedirs(out, exist_ok=True)

def find(pattern):
    hits = sorted(glob.glob(os.path.join(base, "**", pattern), recursive=True))
    return hits[0] if hits else None

ks = find("*kernel_stats.csv")
if ks:
    rows = list(csv.DictReader(open(ks)))
    with open(os.path.join(out, f"{tag}_{ksname}.csv"), "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats --output-format csv -- {kscmd} (1x MI355X; sources {api.source_hash()})\n")
        w = csv.writer(f); w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows:
            w.writerow([r["Name"][:150], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])

def load(p):
    d = collections.OrderedDict()
    if not p or not os.path.exists(p): return d
    for r in csv.DictReader(open(p)):
        d.setdefault((r["Kernel_Name"], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    return d

cnt = collections.OrderedDict()
for p in sorted(glob.glob(os.path.join(base, "**", "*counter_collection.csv"), recursive=True)):
    for (kname, cname), vals in load(p).items():
        cnt.setdefault(kname, {})[cname] = sum(vals) / len(vals)
summ = {"note": "rocprofv3 --pmc <one set per pass> on tools/pmc_probe.py, 1 GiB buffer per launch, averages per launch.  FETCH_SIZE / WRITE_SIZE unit = KiB; "
                "gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts 1/2 of wide coalesced reads -> doubled in fetch_bytes_corrected.  "
                "These are requests at the L2<->fabric boundary: hits in the memory-side Infinity Cache are included (no HBM-level counter is exposed; "
                f"see {tag}_mall_evidence.jsonl for the HBM-level argument).",
        "source_hash": api.source_hash(),
        "kernels": {}}
for kname, c in cnt.items():
    if "vkfft" not in kname and "copy" not in kname.lower() and "elementwise" not in kname.lower(): continue
    e = dict(c)
    if "FETCH_SIZE" in c: e["fetch_bytes_corrected"] = c["FETCH_SIZE"] * 1024 * 2
    if "WRITE_SIZE" in c: e["write_bytes"] = c["WRITE_SIZE"] * 1024
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c: e["bytes_per_launch"] = e["fetch_bytes_corrected"] + e["write_bytes"]
    summ["kernels"][kname[:220]] = e
# the instance of every size: log2 N from the template arguments of the kernel's name (row kernels: the bits of their schedule; fused kernels: the bits of
# both factors; tiles of two halves: twice the half's bits + the two split flags) — bench.py reads by_log2N[size the roofline names]
import re
summ["by_log2N"] = {}
for kname, e in summ["kernels"].items():
    if "bytes_per_launch" not in e or "vkfft_mi355x::pow2_" not in kname: continue
    m = re.search(r"pow2_fused_pkh_kernel<float, (?:vkfft_mi355x::)?Pow2Sched<(\d+), (\d+), (\d+), 0>, (\d), (\d)", kname)
    if m: k = 2 * (int(m.group(1)) + int(m.group(2)) + int(m.group(3))) + int(m.group(4)) + int(m.group(5))
    else:
        sch = re.findall(r"Pow2Sched<(\d+), (\d+), (\d+), (\d+)>", kname)
        if not sch or "col" in kname or "blue" in kname: continue
        k = sum(int(b) for t in sch for b in t)
        if "pow2_row_pairs_kernel" in kname: k += 1  # (rows as pairs of samples: the schedule is the half length's)
    summ["by_log2N"][str(k)] = dict(kernel=kname, bytes_per_launch=e["bytes_per_launch"], fetch_bytes_corrected=e["fetch_bytes_corrected"], write_bytes=e["write_bytes"],
                                    algorithmic_bytes_per_transform=2.0 * (1 << 30))
# the fused Four-Step instances of non-power-of-two lengths (kernel_mix_fused.h): keyed by the length = product of the radices of both factors; tools/pmc_probe.py
# launches (2^25 // N) transforms of N points each
summ["by_length"] = {}
for kname, e in summ["kernels"].items():
    if "bytes_per_launch" not in e or "mix_fused_kernel" not in kname: continue
    sch = re.findall(r"MixSched<(\d+), (\d+), (\d+), (\d+), (\d+)>", kname)
    if len(sch) != 2: continue
    n = 1
    for t in sch:
        for b in t: n *= int(b)
    alg = 2.0 * 8.0 * n * ((1 << 25) // n)
    summ["by_length"][str(n)] = dict(kernel=kname, bytes_per_launch=e["bytes_per_launch"], fetch_bytes_corrected=e["fetch_bytes_corrected"], write_bytes=e["write_bytes"],
                                     algorithmic_bytes_per_launch=alg, ratio=round(e["bytes_per_launch"] / alg, 3))
for fam in ("pow2_fused_kernel", "pow2_row_kernel", "pow2_col_kernel"):
    ks_ = [k for k in summ["kernels"] if fam in k and "bytes_per_launch" in summ["kernels"][k]]
    if ks_:
        summ[fam] = {"bytes_per_launch": max(summ["kernels"][k]["bytes_per_launch"] for k in ks_), "algorithmic_bytes_per_transform": 2.0 * (1 << 30)}
hf = find("pmc_source_hash.txt")
if hf: summ["source_hash"] = open(hf).read().strip()   # what the probe itself saw on the GPU box
if cnt:
    json.dump(summ, open(os.path.join(out, f"{tag}_pmc_traffic.json"), "w"), indent=1)
    print(json.dumps(summ, indent=1)[:3000])
