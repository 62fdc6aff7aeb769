#!/usr/bin/env python3
"""Rewrites the "## Round 2" section of profiles/README.md from the tracked round-2 evidence files (tables of this library next to the reference)."""
import json, os, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
s = open(os.path.join(P, "README.md")).read()
a = s.index("## Round 2")
d = json.load(open(os.path.join(P, "r02_bench.json")))
ps = d["per_size"]
r1 = {8: 6183, 9: 5972, 10: 5954, 11: 5791, 12: 5656, 13: 5220, 14: 4468, 15: 2700, 16: 2716, 17: 2473, 18: 2392, 19: 2305, 20: 2309, 21: 1868, 22: 1757}
ref = {8: 5437, 9: 5758, 10: 5512, 11: 5572, 12: 5358, 13: 5248, 14: 4477, 15: 2439, 16: 2500, 17: 2475, 18: 2388, 19: 2209, 20: 1568, 21: 1621, 22: 1494}
rows = "".join(f"| {k} | {ref[k]} | {r1[k]} | {round(ps[str(k)]['alg_GBps'])} | {round(ps[str(k)]['fwd_only_alg_GBps'])} | {ps[str(k)]['alg_GBps'] / ref[k]:.2f} |\n" for k in range(8, 23))
kn = {0: "C2C", 1: "R2C", 11: "DCT-I", 12: "DCT-II", 13: "DCT-III", 14: "DCT-IV"}
def table(items):
    return "".join(f"| {kn[c['kind']]} | {'×'.join(map(str, c['shape']))} | {'fp64' if c['dp'] else 'fp32'} | {'+'.join(map(str, c['uploads']))} | {round(c['alg_GBps'])} | {round(c['ref_alg_GBps'])} | {c['alg_GBps'] / c['ref_alg_GBps']:.2f} |\n" for c in items if c.get("ref_alg_GBps"))
L = lambda n: [json.loads(l) for l in open(os.path.join(P, f"r02_{n}_with_reference_same_call.jsonl"))]
cfg, sm, s7, s1000, sr = L("config34"), L("samples_3_6_100"), L("sample7_prime_planes"), L("sample1000_sampling"), L("short_rows")
g = lambda r: statistics.geometric_mean([x["alg_GBps"] / x["ref_alg_GBps"] for x in r if x.get("ref_alg_GBps")])
hdr = "| transform | shape | precision | passes | this library | reference | ratio |\n|---|---|---|---|---|---|---|\n"
new = f'''## Round 2

All files from the final round-2 build unless noted.  "reference in the same process" = the reference VkFFT-HIP (`oracle/_ref`, built by `oracle/build_ref.sh`)
timed right after this library on the same shapes with the same protocol; ratios > 1 mean this library is faster.  (Regenerate this section with `tools/gen_profiles_readme.py`.)

| file | what | command |
|---|---|---|
| `r02_bench.json` | bench.py JSON line ({d['value'] / 1000:.2f} TFLOP/s, {d['ms_per_step']:.2f} ms/step on the box of the last run, whose copy rate was {d['roofline']['copy_GBps_same_box'] / 1000:.2f} TB/s; 16.5–16.6 TFLOP/s on boxes copying at 5.2–5.5) | `python bench.py --steps 5 --warmup 2` |
| `r02_bench_kernel_stats.csv` | per-kernel time of the headline benchmark | `cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline` |
| `r02_pmc_traffic.json` | bytes per launch at the L2↔fabric boundary (FETCH_SIZE ×2 per the gfx950 rule, WRITE_SIZE, TCC_EA0_RDREQ/WRREQ incl. `_DRAM`), one counter set per rocprofv3 pass | `rocprofv3 --pmc <set> --kernel-trace … python tools/pmc_probe.py`; `tools/summarize_profiles.py r02 <dir>` |
| `r02_mall_evidence.jsonl` | HBM-level argument for the fused kernel: the same kernel with rings of 208 MiB … 2 GiB (time rises to that of two passes or beyond once the ring exceeds the Infinity Cache) | `python tools/mall_evidence.py` |
| `r02_probe3_read_write_ceilings.jsonl` | read-only / write-only / copy ceilings by working-set size (1 MiB … 1 GiB) and cache policy | `build/probe3` (`tools/probe3.hip`) |
| `r02_probe2_ticketed_copies.jsonl` | the same copies with one global ticket counter per 32 KiB block: saturates at ≈5 TB/s (≈90 tickets/µs) — why the fused kernel has one queue per XCD | `build/probe2` (`tools/probe2.hip`) |
| `r02_probe_exchange.jsonl` | small row kernels (2^8..2^10) as built vs with the inter-stage exchange compiled out: same time — the exchange mechanism (LDS vs shuffles) cannot matter | `build/probe_exchange`, `build/probe_exchange_nox` (`tools/probe_exchange.hip`, `-DVKFFT_PROBE_NO_EXCHANGE`) |
| `r02_config34_with_reference_same_call.jsonl` | configs 3/4 (non-pow2, Rader/Bluestein, fp64, R2C, DCT, 3D, multi-pass), reference in the same process: geometric mean {g(cfg):.2f} | `python tools/perf_configs.py` |
| `r02_samples_3_6_100_with_reference_same_call.jsonl` | a sampling of the size lists of the reference's multi-dimensional benchmarks (sample 3 C2C, sample 6 R2C, sample 100 DCT-II): geometric mean {g(sm):.2f} | `python tools/perf_samples.py` |
| `r02_sample7_prime_planes_with_reference_same_call.jsonl` | the reference's sample 7 (prime × prime planes, prime cubes): geometric mean {g(s7):.2f} (0.27 at the start of the round) | `python tools/perf_sample7.py` |
| `r02_sample1000_sampling_with_reference_same_call.jsonl` | a sampling of the reference's sample 1000 (1-D C2C of every length 2 … 4096): geometric mean {g(s1000):.2f} | `python tools/perf_sample1000.py 40 1` |
| `r02_short_rows_with_reference_same_call.jsonl` | 1-D rows of 4 … 128 points and small planes / cubes: geometric mean {g(sr):.2f} | `python tools/perf_small_rows.py` |
| `r02_fp64_fused_vs_separate.jsonl` | fp64 2^14…2^21: fused Four-Step against separate passes | `python tools/perf_fused_fp64.py` |
| `r02_convolution_with_reference_same_call.jsonl` | convolution plans (2-D, 1×1 and 3×3, C2C and R2C) with the reference's merged kernels timed in the same process | `python tools/perf_conv.py` |
| `r02_reference_conv_zeropad_probe.txt` | what the reference's HIP backend returns for convolution / zero-padding configurations against the definition (2-D convolution and axis-0 padding agree; small 1-D convolutions die with SIGFPE; 3-D convolution and axis-1 padding do not follow the definition) | `python tools/ref_probe_conv.py; python tools/ref_probe_zeropad.py` |
| `r02_fused_stress_after_barrier_fix.jsonl` | 300 forward+inverse launch pairs per line under unbalanced queues and default settings after the barrier fix of DESIGN §4.10: 0 wrong (before: 2–29 wrong in 300 with 3, 5, 6, 7 queues) | `python tools/stress_fused.py <log2N> 300 KEY=VALUE…` |

Headline sweep, algorithmic GB/s of an FFT+iFFT pair (1 GiB), this library round 1 → round 2 (`r02_bench.json`), reference from round 1's table:

| log2 N | reference VkFFT-HIP | round 1 | round 2 | round 2 forward-only | round 2 / reference |
|---|---|---|---|---|---|
{rows}
Sum of the 15 pair times: reference 22.89 ms, round 1 20.59 ms, round 2 {d['ms_per_step']:.2f} ms per step ({d['value'] / 1000:.2f} TFLOP/s nominal; 18.15–18.6 ms over the boxes of the last day).

Configs 3/4 (`r02_config34_with_reference_same_call.jsonl`):

{hdr}{table(cfg)}
Samplings of the reference's multi-dimensional benchmark lists (`r02_samples_3_6_100_…jsonl`):

{hdr}{table(sm)}
Short rows and small systems (`r02_short_rows_…jsonl`):

{hdr}{table(sr)}
Sample 7, every third system (`r02_sample7_…jsonl`):

{hdr}{table(s7[::3])}
Sample 1000, the sampled lengths (`r02_sample1000_…jsonl`):

{hdr}{table(s1000)}'''
open(os.path.join(P, "README.md"), "w").write(s[:a] + new)
print("profiles/README.md: round-2 section rewritten")
