"""Per-kernel sums of a rocprofv3 --pmc run (counter_collection.csv files under a directory): python tools/pmc_table.py <dir> [name filter]"""
import csv, glob, os, sys, collections
d = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.OrderedDict()
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if flt and flt not in k: continue
        key = (k[:100], r.get("Grid_Size"), r.get("LDS_Block_Size"), r.get("VGPR_Count"))
        e = acc.setdefault(key, collections.Counter())
        e[r["Counter_Name"]] += float(r["Counter_Value"]); e["_n_" + r["Counter_Name"]] += 1
for (k, grid, lds, vgpr), e in acc.items():
    names = [n for n in e if not n.startswith("_n_")]
    print(k, "grid", grid, "lds", lds, "vgpr", vgpr)
    wc = e.get("SQ_WAVE_CYCLES", 0) or 1
    print("   ", "  ".join(f"{n}={e[n] / e['_n_' + n]:.3g}" + (f" ({e[n] / e['_n_' + n] / (wc / e['_n_SQ_WAVE_CYCLES']) * 100:.0f}%)" if n.startswith("SQ_") and n != "SQ_WAVE_CYCLES" and "SQ_WAVE_CYCLES" in e else "") for n in names))
