"""Turns the remarks of `make CXXFLAGS='... -Rpass-analysis=kernel-resource-usage'` (one or more build logs, later ones override earlier ones for the kernels they rebuilt)
into a tracked summary: per kernel family the instance count, the VGPR range, the instances with scratch, the occupancy histogram; and the full list of instances with scratch.
usage: kernel_resources.py <out.json> <log> [<log> ...]"""
import collections, json, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vkfft_amd import api
out, logs = sys.argv[1], sys.argv[2:]
kern = collections.OrderedDict()
def grab(pat, b):
    m = re.search(pat, b)
    return int(m.group(1)) if m else None
for log in logs:
    txt = open(log, errors="replace").read()
    for b in re.split(r"remark: Function Name: ", txt)[1:]:
        name = b.split()[0]
        kern[name] = dict(vgpr=grab(r"VGPRs: (\d+)", b), agpr=grab(r"AGPRs: (\d+)", b), sgpr=grab(r"SGPRs: (\d+)", b), scratch=grab(r"ScratchSize \[bytes/lane\]: (\d+)", b),
                          occupancy=grab(r"Occupancy \[waves/SIMD\]: (\d+)", b), lds=grab(r"LDS Size \[bytes/block\]: (\d+)", b), sgpr_spill=grab(r"SGPRs Spill: (\d+)", b), vgpr_spill=grab(r"VGPRs Spill: (\d+)", b))
fam = collections.OrderedDict()
for name, r in kern.items():
    m = re.match(r"_ZN12vkfft_mi355x\d+([a-z0-9_]+?)I", name)
    f = m.group(1) if m else name[:40]
    if f == "mixed_row_kernel":
        f += "<OPS=%s>" % name.split("EEEvNS_10PassParamsE")[0][-1]
    e = fam.setdefault(f, dict(instances=0, vgpr_min=10**9, vgpr_max=0, with_scratch=0, scratch_bytes_max=0, occupancy=collections.Counter()))
    e["instances"] += 1
    if r["vgpr"] is not None:
        e["vgpr_min"] = min(e["vgpr_min"], r["vgpr"]); e["vgpr_max"] = max(e["vgpr_max"], r["vgpr"])
    if r["scratch"]:
        e["with_scratch"] += 1; e["scratch_bytes_max"] = max(e["scratch_bytes_max"], r["scratch"])
    e["occupancy"][str(r["occupancy"])] += 1
for e in fam.values():
    e["occupancy"] = dict(sorted(e["occupancy"].items()))
json.dump({"note": "clang -Rpass-analysis=kernel-resource-usage over the whole library (gfx950); scratch = bytes per lane; occupancy = waves per SIMD by registers and LDS",
           "source_hash": api.source_hash(), "kernels": len(kern), "families": fam,
           "instances_with_scratch": {n: {k: v for k, v in r.items() if v is not None} for n, r in kern.items() if r["scratch"]}}, open(out, "w"), indent=1)
for f, e in fam.items():
    print(f, e["instances"], "vgpr", e["vgpr_min"], "-", e["vgpr_max"], "scratch:", e["with_scratch"], "max", e["scratch_bytes_max"], "occ", e["occupancy"])
