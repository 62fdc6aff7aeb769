#!/usr/bin/env python3
"""Rewrites the "## Round 5" section of profiles/README.md from the tracked `r05_*` evidence files.
Refuses to run when the bench line, the kernel stats and the PMC summary were not taken on the same sources, and takes the reference column ONLY from
`r05_reference_pow2_same_lease.jsonl` (the reference's own sweep run in the same gpurun call as the bench line: the boxes of the pool differ by 3-4 %)."""
import json, math, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
d = json.load(open(f"{P}/r05_bench.json")); d0 = json.load(open(f"{P}/r04b_bench.json"))
pmc = json.load(open(f"{P}/r05_pmc_traffic.json"))
ks_head = open(f"{P}/r05_bench_kernel_stats.csv").readline()
ks_hash = re.search(r"sources ([0-9a-f]{16})", ks_head).group(1)
if ks_hash != pmc["source_hash"]:
    sys.exit(f"kernel stats were taken on sources {ks_hash}, the PMC summary on {pmc['source_hash']}: re-collect both on one build")
ref = {json.loads(l)["log2N"]: json.loads(l) for l in open(f"{P}/r05_reference_pow2_same_lease.jsonl") if l.startswith("{")}
if sorted(ref) != list(range(8, 23)):
    sys.exit("r05_reference_pow2_same_lease.jsonl does not hold the 15 sizes of the sweep")
ps, ps0 = d["per_size"], d0["per_size"]
rows = ""
for k in range(8, 23):
    v = ps[str(k)]; by = pmc.get("by_log2N", {}).get(str(k), {})
    rows += (f"| {k} | {round(ref[k]['alg_GBps'])} | {round(ps0[str(k)]['alg_GBps'])} | **{round(v['alg_GBps'])}** | {round(v['fwd_only_alg_GBps'])} | {v['alg_GBps'] / ref[k]['alg_GBps']:.2f} | "
             f"{v['alg_GBps'] / 8000:.2f} | `{v['kernel'].split('<')[0]}` | {by.get('fetch_bytes_corrected', 0) / 2**30:.3f} / {by.get('write_bytes', 0) / 2**30:.3f} |\n")
ref_ms = sum(r["pair_ms"] for r in ref.values()); our_ms = sum(v["pair_ms"] for v in ps.values())
ks = [l.split('","')[0].strip('"') + " | " + l.rsplit('"', 1)[1] for l in open(f"{P}/r05_bench_kernel_stats.csv").read().splitlines()[2:8]]
s = open(f"{P}/README.md").read()
tag = "## Round 5"
if tag in s:
    s = s[:s.index(tag)]
rl = d["roofline"]
s = s.rstrip("\n") + f'''

{tag}

Files `r05_*`.  The bench line, the kernel stats and the PMC traffic are of ONE build (sources `{pmc['source_hash']}`, `vkfft_amd.api.source_hash()`), the reference's sweep
(`oracle/_ref/vkfft_ref_bench 8 22 0`) was run in the SAME gpurun call as the bench line.  Bench line: **{d['value']/1000:.2f} TFLOP/s, {d['ms_per_step']:.2f} ms per step** (round 4: {d0['value']/1000:.2f} / {d0['ms_per_step']:.2f});
copy rate of that box {rl['copy_GBps_same_box']/1000:.2f} TB/s (torch) / {(rl.get('copy_GBps_own_float4') or 0)/1000:.2f} TB/s (the library's own 16-byte-per-lane copy); after the timed loop the buffer equals its
initial contents to {d['roundtrip_rel_l2']:.2e} relative L2 over {d['roundtrip_pairs']} transform pairs (limit {d['roundtrip_limit_rel_l2']:.1e}).  Sum of the 15 pair times: reference {ref_ms:.2f} ms, this library {our_ms:.2f} ms.
`roofline`: `{rl['kernel']}` at 2^{rl['size_log2N']}, {rl['launch_ms']} ms per launch (HIP events), {rl['achieved']} GB/s algorithmic = {rl['frac']} of 8 TB/s; traffic of THAT instance {((rl.get('traffic') or 0))/2**30:.3f} GiB per launch.

| log2 N | reference, same lease (alg. GB/s, paired) | round 4 (r04b) | **round 5** | round 5, forward only | ratio to the reference | fraction of 8 TB/s | kernel | PMC fetch / write per launch (GiB; 1 GiB data) |
|---|---|---|---|---|---|---|---|---|
{rows}
Dominant kernels (rocprofv3 `--kernel-trace --stats`, `r05_bench_kernel_stats.csv`; name | calls, total ns, average ns, share):
''' + "".join(f"* `{k[:110]}`\n" for k in ks) + f'''
| file | what | command |
|---|---|---|
| `r05_bench.json` | bench.py JSON line | `python bench.py --steps 5 --warmup 2` |
| `r05_reference_pow2_same_lease.jsonl` | the reference's HIP backend, sample-0 protocol, same gpurun call as the bench line | `oracle/_ref/vkfft_ref_bench 8 22 0` |
| `r05_bench_kernel_stats.csv` | per-kernel time of the headline benchmark | `cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline` |
| `r05_pmc_traffic.json` | bytes per launch at the L2↔fabric boundary of EVERY plan bench.py launches (`by_log2N`: the instance of each size; FETCH_SIZE × 2 per the gfx950 rule + WRITE_SIZE, separate passes) | `rocprofv3 --pmc FETCH_SIZE … python tools/pmc_probe.py`, same with `WRITE_SIZE`; `python tools/summarize_profiles.py r05 <dir>` |
| `r05_ab_first_packed_build_vs_round4_shapes.jsonl`, `r05_ab_packed_vs_round4_shapes.jsonl` | A/B on one box, packed-pair kernels against the round-4 shapes (`P2V<k>`, `FUV<k>` select them), result check per line (largest element error in ulp of the rms after a forward transform and after a round trip): first build (vector arithmetic halved, times unchanged: the kernels are not VALU-bound) and final kernels | `python tools/ab_r05.py` |
| `r05_fused_phase_profile_packed_kernels.txt` | per-phase cycle sums per ticket of the packed fused kernels, three builds (DESIGN §4.10b: the stall at the issue of stores, the second half requested behind stores, scratch reloads, the twiddle look-ups) | `make dev; VKFFT_MI355X_LIB=… python tools/prof_fused.py <k> <k>` |
| `r05_fused_lag_ring_pairs.jsonl`, `r05_fused_margin_chunk_queue_knobs.jsonl` | lag / ring pairs and the other planner knobs at 2^19 … 2^22 (2^22: lag 4 / ring 8 = 256 MiB 3.04 TB/s against 2.73 for lag 3 / ring 6: the ring budget of 32 MiB transforms) | `python tools/ab_r05.py lags`, `… tune` |
| `r05_fused_stress_unbalanced_queues.jsonl` | 480 launch pairs under unbalanced queues (3, 5, 6, 7 queues, lag 1, ring 4) on the packed kernels, 2^16 … 2^22: 0 wrong | `python tools/ab_r05.py stress` |
| `r05_real_rows_selected_baseline.jsonl` | real rows off the fused-map lists with the reference in the same process BEFORE the table-driven maps ( R2C 169 0.35 ×, DCT-II 169 0.24 ×, R2C 385 0.56 ×, R2C 100 0.94 ×) | `python tools/perf_real_rows.py …` |
| `r05_kernel_resources.json` | registers, scratch and occupancy of every kernel instance of the final sources | `make CXXFLAGS='… -Rpass-analysis=kernel-resource-usage' 2> log; python tools/kernel_resources.py profiles/r05_kernel_resources.json log` |
| `r05_gpu_suite.log` | `pytest -m gpu` on the device, final sources (the hash is the first line of the file): 648 passed, 1 skipped | see the file |
'''
# ---- real rows: the three sweeps of the final build, reference in the same process on every length
import collections
def geo(v): return math.exp(sum(math.log(x) for x in v) / len(v))
rr = ""
for fam, label, old in (("r2c", "R2C 4 … 400 (step 3)", 0.63), ("dct2", "DCT-II 4 … 400 (step 3)", 0.53), ("dct4", "DCT-IV 5 … 400 (step 5)", 0.64)):
    fn = f"{P}/r05_{fam}_rows_reference_every_length_final.jsonl"
    if not os.path.exists(fn): continue
    rows_ = [json.loads(l) for l in open(fn) if l.startswith("{")]
    rs = [(r["ref_ms"] / r["ms"], r["N"], r["kernel"]) for r in rows_ if r.get("ref_ms")]
    by = collections.defaultdict(list)
    for x, n, k in rs: by[k].append(x)
    worst = min(rs)
    rr += (f"| {label} | {len(rs)} | {old:.2f} | **{geo([x[0] for x in rs]):.2f}** | {sum(1 for x in rs if x[0] < 0.5)} | {worst[1]} ({worst[0]:.2f}) | "
           + ", ".join(f"`{k}` {len(v)}: {geo(v):.2f}" for k, v in sorted(by.items())) + " |\n")
if rr:
    s += f"""
Real rows, geometric mean of (reference pair time ÷ our pair time), the reference timed in the same process on EVERY length (`tools/perf_real_sweep_r05.py`; 2^25 reals per launch):

| sweep | lengths | round 4 (r04b) | **round 5** | lengths below 0.5 × | worst length | by kernel (count: geometric mean) |
|---|---|---|---|---|---|---|
{rr}
| file | what |
|---|---|
| `r05_{{r2c,dct2,dct4}}_rows_reference_every_length_final.jsonl` | the three sweeps, sources `88eb3dd2fd1e7cad` — one commit before the bench line's: the same kernels; after it only the registry order of 2^11 / 2^12 and a planner guard changed |
| `r05_*_rows_reference_every_length_step1…4_*.jsonl` | the same sweeps after each step of DESIGN §4.12c: tables in the instance kernels; + two rows per Bluestein transform and the Rader-stage tables; + the staging tile as static LDS (a regression: the even lengths); + as dynamic LDS |
| `r05_real_rows_table_maps_ab.jsonl` | A/B on one box: table-driven against generic maps of the same plans, even lengths on half- against full-length forms, thresholds of threads per row |
| `r05_real_rows_169_counters.txt` | rocprofv3 kernel trace and SQ counters of the 169-point rows: what bounds them (DESIGN §4.12c) |
"""
open(f"{P}/README.md", "w").write(s + "\n")
print("profiles/README.md: round 5 section written;", f"{d['value']/1000:.2f} TFLOP/s", "ratios", [round(ps[str(k)]['alg_GBps'] / ref[k]['alg_GBps'], 2) for k in range(8, 23)])
