"""Copies what tools/gpurun_scripts/r06_final.sh left under gpurun_out/r06z into the tracked summaries of profiles/ (round 6), then rewrites the round's README section.
usage: python tools/collect_final_r06.py [subdir of gpurun_out, default r06z]"""
import json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sub = sys.argv[1] if len(sys.argv) > 1 else "r06z"
G = os.path.join(ROOT, "gpurun_out", sub); P = os.path.join(ROOT, "profiles")
def cp(src, dst):
    if os.path.exists(os.path.join(G, src)) and os.path.getsize(os.path.join(G, src)) > 0:
        shutil.copy(os.path.join(G, src), os.path.join(P, dst)); print("copied", dst)
    else: print("MISSING", src)
cp("bench.json", "r06_bench.json")
cp("reference_pow2_same_lease.jsonl", "r06_reference_pow2_same_lease.jsonl")
cp("sample1000.jsonl", "r06_sample1000_sampling_with_reference_same_call.jsonl")
cp("composites.jsonl", "r06_rader_stage_final_kernel_with_reference.jsonl")
for f in ("r2c", "dct2", "dct4"):
    cp(f + "_rows.jsonl", "r06_%s_rows_reference_every_length_final.jsonl" % f)
cp("config34.jsonl", "r06_config34_with_reference_same_call.jsonl")
cp("gpu_suite.log", "r06_gpu_suite.log")
cp("mix_fused_with_reference.jsonl", "r06_mix_fused_final_with_reference.jsonl")
cp("mix_fused_separate_passes.jsonl", "r06_mix_fused_final_separate_passes_same_call.jsonl")
cp("long_rows_with_reference.jsonl", "r06_long_rows_one_pass_with_reference.jsonl")
cp("long_rows_two_passes.jsonl", "r06_long_rows_two_passes_same_call.jsonl")
# kernel stats of the bench line + PMC traffic (power-of-two plans by_log2N, fused non-power-of-two plans by_length)
for d in ("pmc_fetch", "pmc_write"):
    pass
tmp = os.path.join(ROOT, "gpurun_out", sub + "_pmc"); shutil.rmtree(tmp, ignore_errors=True); os.makedirs(tmp)
def newest_only(src, dst):
    # (gpurun MERGES a call's files into gpurun_out/: an earlier call's CSVs — other process ids in their names — are still there; keep the newest of each kind)
    import glob, re
    os.makedirs(dst, exist_ok=True)
    kinds = {}
    for f in glob.glob(os.path.join(src, "**", "*.csv"), recursive=True):
        k = re.sub(r"^\d+_", "", os.path.basename(f))
        if k not in kinds or os.path.getmtime(f) > os.path.getmtime(kinds[k]): kinds[k] = f
    for k, f in kinds.items(): shutil.copy(f, os.path.join(dst, os.path.basename(f)))
for d in ("prof_bench", "pmc_fetch", "pmc_write"):
    if os.path.isdir(os.path.join(G, d)): newest_only(os.path.join(G, d), os.path.join(tmp, d))
if os.path.exists(os.path.join(G, "pmc_source_hash.txt")): shutil.copy(os.path.join(G, "pmc_source_hash.txt"), tmp)
subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "summarize_profiles.py"), "r06", sub + "_pmc"], stdout=subprocess.DEVNULL)
tmp2 = os.path.join(ROOT, "gpurun_out", sub + "_mf"); shutil.rmtree(tmp2, ignore_errors=True); os.makedirs(tmp2)
if os.path.isdir(os.path.join(G, "prof_mixfused")):
    newest_only(os.path.join(G, "prof_mixfused"), os.path.join(tmp2, "prof_mixfused"))
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "summarize_profiles.py"), "r06tmp", sub + "_mf", "mix_fused_kernel_stats",
                           "python tools/pmc_mixrad.py 59049 177147 531441 78125 390625 117649 161051 1771561 28561"], stdout=subprocess.DEVNULL)
    if os.path.exists(os.path.join(P, "r06tmp_mix_fused_kernel_stats.csv")): os.replace(os.path.join(P, "r06tmp_mix_fused_kernel_stats.csv"), os.path.join(P, "r06_mix_fused_kernel_stats.csv"))
pj = json.load(open(os.path.join(P, "r06_pmc_traffic.json")))
print("pmc sources", pj.get("source_hash"), "by_log2N", sorted(pj.get("by_log2N", {}), key=int), "by_length", {k: v["ratio"] for k, v in pj.get("by_length", {}).items()})
