#!/usr/bin/env python3
"""Rewrites the "## Round <n>" section of profiles/README.md from the tracked evidence files of that round (ONE generator; rounds 1-5 had one script each, their
sections stay in the README as written).  usage: python tools/gen_profiles_readme.py r06 [previous round tag, default: the one before]
Refuses to run when the bench line, the kernel stats and the PMC summary were not taken on the same sources; the reference column comes ONLY from
`<tag>_reference_pow2_same_lease.jsonl` (the reference's own sweep run in the same gpurun call as the bench line: the boxes of the pool differ by 3-4 %).  The traffic of
the roofline's instance is read from the PMC summary (`by_log2N`), not from the bench line (which may have been written before the counters existed)."""
import collections, glob, json, math, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
prev = sys.argv[2] if len(sys.argv) > 2 else "r%02d" % (int(tag[1:3]) - 1)
rnd = int(tag[1:3])
def geo(v): return math.exp(sum(math.log(x) for x in v) / len(v))
def jl(name): return [json.loads(l) for l in open(os.path.join(P, name)) if l.startswith("{")]
out = ""
# ---- the bench line, the reference in the same lease, kernel stats, PMC traffic
if os.path.exists(f"{P}/{tag}_bench.json"):
    d = json.load(open(f"{P}/{tag}_bench.json")); d0 = json.load(open(f"{P}/{prev}_bench.json"))
    pmc = json.load(open(f"{P}/{tag}_pmc_traffic.json"))
    ks_lines = open(f"{P}/{tag}_bench_kernel_stats.csv").read().splitlines()
    ks_hash = re.search(r"sources ([0-9a-f]{16})", ks_lines[0]).group(1)
    if ks_hash != pmc["source_hash"]:
        sys.exit(f"kernel stats were taken on sources {ks_hash}, the PMC summary on {pmc['source_hash']}: re-collect both on one build")
    ref = {r["log2N"]: r for r in jl(f"{tag}_reference_pow2_same_lease.jsonl")}
    if sorted(ref) != list(range(8, 23)):
        sys.exit(f"{tag}_reference_pow2_same_lease.jsonl does not hold the 15 sizes of the sweep")
    ps, ps0 = d["per_size"], d0["per_size"]
    rows = ""
    for k in range(8, 23):
        v = ps[str(k)]; by = pmc.get("by_log2N", {}).get(str(k), {})
        rows += (f"| {k} | {round(ref[k]['alg_GBps'])} | {round(ps0[str(k)]['alg_GBps'])} | **{round(v['alg_GBps'])}** | {round(v['fwd_only_alg_GBps'])} | {v['alg_GBps'] / ref[k]['alg_GBps']:.2f} | "
                 f"{v['alg_GBps'] / 8000:.2f} | `{v['kernel'].split('<')[0]}` | {by.get('fetch_bytes_corrected', 0) / 2**30:.3f} / {by.get('write_bytes', 0) / 2**30:.3f} |\n")
    rl = d["roofline"]
    inst = pmc.get("by_log2N", {}).get(str(rl["size_log2N"]), {})
    ref_ms = sum(r["pair_ms"] for r in ref.values()); our_ms = sum(v["pair_ms"] for v in ps.values())
    ks = [l.split('","')[0].strip('"') + " | " + l.rsplit('"', 1)[1] for l in ks_lines[2:8]]
    out += f'''
Files `{tag}_*`.  The bench line, the kernel stats and the PMC traffic are of ONE build (sources `{pmc['source_hash']}`, `vkfft_amd.api.source_hash()`), the reference's sweep
(`oracle/_ref/vkfft_ref_bench 8 22 0`) was run in the SAME gpurun call as the bench line.  Bench line: **{d['value']/1000:.2f} TFLOP/s, {d['ms_per_step']:.2f} ms per step** (round {rnd - 1}: {d0['value']/1000:.2f} / {d0['ms_per_step']:.2f});
copy rate of that box {rl['copy_GBps_same_box']/1000:.2f} TB/s (torch) / {(rl.get('copy_GBps_own_float4') or 0)/1000:.2f} TB/s (the library's own 16-byte-per-lane copy); after the timed loop the buffer equals its
initial contents to {d['roundtrip_rel_l2']:.2e} relative L2 over {d['roundtrip_pairs']} transform pairs (limit {d['roundtrip_limit_rel_l2']:.1e}).  Sum of the 15 pair times: reference {ref_ms:.2f} ms, this library {our_ms:.2f} ms.
`roofline`: `{rl['kernel']}` at 2^{rl['size_log2N']}, {rl['launch_ms']} ms per launch (HIP events), {rl['achieved']} GB/s algorithmic = {rl['frac']} of 8 TB/s; traffic of THAT instance (PMC summary, `by_log2N[{rl['size_log2N']}]`)
{(inst.get('bytes_per_launch') or 0)/2**30:.3f} GiB per launch = {(inst.get('bytes_per_launch') or 0) / (2**31 / rl['launches_per_transform']) :.2f} x the algorithmic bytes.

| log2 N | reference, same lease (alg. GB/s, paired) | round {rnd - 1} | **round {rnd}** | round {rnd}, forward only | ratio to the reference | fraction of 8 TB/s | kernel | PMC fetch / write per launch (GiB; 1 GiB data) |
|---|---|---|---|---|---|---|---|---|
{rows}
Dominant kernels (rocprofv3 `--kernel-trace --stats`, `{tag}_bench_kernel_stats.csv`; name | calls, total ns, average ns, share):
''' + "".join(f"* `{k[:110]}`\n" for k in ks)
# ---- sampling of sample 1000
f = glob.glob(f"{P}/{tag}_sample1000_*.jsonl")
if f:
    rows_ = [r for r in jl(os.path.basename(f[0])) if r.get("ref_pair_ms")]
    q = sorted((r["ref_pair_ms"] / r["pair_ms"], r["shape"][0]) for r in rows_)
    out += (f"\nSampling of the reference's sample 1000 (`{os.path.basename(f[0])}`, {len(q)} lengths 2 … 4096, reference in the same process): geometric mean **{geo([x[0] for x in q]):.3f}**, "
            f"below 0.9 ×: {', '.join(f'{n} ({x:.2f})' for x, n in q if x < 0.9) or 'none'}; below 0.7 ×: {sum(1 for x, n in q if x < 0.7)}.\n")
# ---- real rows
rr = ""
for fam, label in (("r2c", "R2C 4 … 400 (step 3)"), ("dct2", "DCT-II 4 … 400 (step 3)"), ("dct4", "DCT-IV 5 … 400 (step 5)")):
    cur = sorted(glob.glob(f"{P}/{tag}_{fam}_rows_reference_every_length_*.jsonl")); old = glob.glob(f"{P}/{prev}_{fam}_rows_reference_every_length_final.jsonl")
    if not cur: continue
    def summ(fn):
        rs = [(r["ref_ms"] / r["ms"], r["N"], r["kernel"]) for r in jl(os.path.basename(fn)) if r.get("ref_ms")]
        by = collections.defaultdict(list)
        for x, n, k in rs: by[k].append(x)
        return rs, by
    rs, by = summ(cur[-1]); o = geo([x[0] for x in summ(old[0])[0]]) if old else float("nan")
    worst = min(rs)
    rr += (f"| {label} | {len(rs)} | {o:.2f} | **{geo([x[0] for x in rs]):.2f}** | {sum(1 for x in rs if x[0] < 0.5)} | {worst[1]} ({worst[0]:.2f}) | "
           + ", ".join(f"`{k}` {len(v)}: {geo(v):.2f}" for k, v in sorted(by.items())) + f" | `{os.path.basename(cur[-1])}` |\n")
if rr:
    out += f"""
Real rows, geometric mean of (reference pair time ÷ our pair time), the reference timed in the same process on EVERY length (`tools/perf_real_sweep_r05.py`; 2^25 reals per launch):

| sweep | lengths | round {rnd - 1} | **round {rnd}** | lengths below 0.5 × | worst length | by kernel (count: geometric mean) | file |
|---|---|---|---|---|---|---|---|
{rr}"""
# ---- two-pass plans replaced this round: fused Four-Step of non-power-of-two lengths, long rows in one pass
def two_col(title, cur, base, label_cur, label_base):
    if not (os.path.exists(f"{P}/{cur}") and os.path.exists(f"{P}/{base}")): return ""
    a = {r["shape"][0]: r for r in jl(cur) if "alg_GBps" in r}; b = {r["shape"][0]: r for r in jl(base) if "alg_GBps" in r}
    t = f"\n{title} (alg. GB/s of a forward + inverse pair, 2^25 points; `{cur}`, `{base}`, same gpurun call):\n\n| N | {label_base} | **{label_cur}** | gain | reference, same process | ratio to the reference | fraction of 8 TB/s |\n|---|---|---|---|---|---|---|\n"
    for n, r in a.items():
        ref = r.get("ref_alg_GBps")
        t += f"| {n} | {round(b[n]['alg_GBps']) if n in b else ''} | **{round(r['alg_GBps'])}** | {r['alg_GBps'] / b[n]['alg_GBps']:.2f} | {round(ref) if ref else ''} | {r['alg_GBps'] / ref:.2f} | {r['alg_GBps'] / 8000:.2f} |\n" if n in b and ref else ""
    return t
out += two_col("Fused Four-Step of non-power-of-two two-factor lengths (`kernel_mix_fused.h`, DESIGN §4.15) against the separate passes", f"{tag}_mix_fused_final_with_reference.jsonl", f"{tag}_mix_fused_final_separate_passes_same_call.jsonl", "one fused launch", "separate passes")
out += two_col("Rows of 8192 … 16807 points in ONE pass (`mixed_table_6.inc`, DESIGN §4.4a) against the two-pass plans", f"{tag}_long_rows_one_pass_with_reference.jsonl", f"{tag}_long_rows_two_passes_same_call.jsonl", "one pass", "two passes")
if os.path.exists(f"{P}/{tag}_pmc_traffic.json"):
    bl = json.load(open(f"{P}/{tag}_pmc_traffic.json")).get("by_length", {})
    if bl:
        out += "\nL2 ↔ fabric traffic of the fused non-power-of-two launches (`" + tag + "_pmc_traffic.json` `by_length`; bytes per launch ÷ algorithmic bytes, Infinity-Cache hits of the ring included): " + ", ".join(f"{n}: {v['ratio']:.2f}" for n, v in sorted(bl.items(), key=lambda kv: int(kv[0]))) + ".\n"
# ---- every other file of the round: its first comment / note, by name
notes = json.load(open(f"{P}/{tag}_files.json")) if os.path.exists(f"{P}/{tag}_files.json") else {}
files = sorted(os.path.basename(x) for x in glob.glob(f"{P}/{tag}_*") if not x.endswith("_files.json"))
out += "\n| file | what |\n|---|---|\n" + "".join(f"| `{n}` | {notes.get(n, '')} |\n" for n in files)
s = open(f"{P}/README.md").read()
head = f"## Round {rnd}"
if head in s:
    s = s[:s.index(head)]
open(f"{P}/README.md", "w").write(s.rstrip("\n") + f"\n\n{head}\n" + out + "\n")
print(f"profiles/README.md: round {rnd} section written ({len(files)} files)")
