#!/usr/bin/env python3
"""Rewrites the "## Round 3" section of profiles/README.md from the tracked round-3 evidence files (tables of this library next to the reference).
The sections of earlier rounds stay as they were written."""
import json, os, statistics, glob
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
s = open(os.path.join(P, "README.md")).read()
if "## Round 3" in s:
    s = s[:s.index("## Round 3")]
s = s.rstrip("\n") + "\n\n"
d = json.load(open(os.path.join(P, "r03_bench.json")))
d2 = json.load(open(os.path.join(P, "r02_bench.json")))
ps, ps2 = d["per_size"], d2["per_size"]
ref = {8: 5437, 9: 5758, 10: 5512, 11: 5572, 12: 5358, 13: 5248, 14: 4477, 15: 2439, 16: 2500, 17: 2475, 18: 2388, 19: 2209, 20: 1568, 21: 1621, 22: 1494}
rows = "".join(f"| {k} | {ref[k]} | {round(ps2[str(k)]['alg_GBps'])} | {round(ps[str(k)]['alg_GBps'])} | {round(ps[str(k)]['fwd_only_alg_GBps'])} | {ps[str(k)]['alg_GBps'] / ref[k]:.2f} |\n" for k in range(8, 23))
kn = {0: "C2C", 1: "R2C", 11: "DCT-I", 12: "DCT-II", 13: "DCT-III", 14: "DCT-IV"}
def table(items):
    return "".join(f"| {kn[c['kind']]} | {'×'.join(map(str, c['shape']))} | {'fp64' if c['dp'] else 'fp32'} | {'+'.join(map(str, c['uploads']))} | {round(c['alg_GBps'])} | {round(c['ref_alg_GBps'])} | {c['alg_GBps'] / c['ref_alg_GBps']:.2f} |\n"
                   for c in items if c.get("ref_alg_GBps"))
def L(n):
    f = os.path.join(P, f"r03_{n}_with_reference_same_call.jsonl")
    return [json.loads(l) for l in open(f) if l.strip().startswith("{")] if os.path.exists(f) else []
cfg, sm, s7, s1000 = L("config34"), L("samples_3_6_100"), L("sample7_prime_planes"), L("sample1000_sampling")
g = lambda r: statistics.geometric_mean([x["alg_GBps"] / x["ref_alg_GBps"] for x in r if x.get("ref_alg_GBps")]) if r else float("nan")
hdr = "| transform | shape | precision | passes | this library | reference | ratio |\n|---|---|---|---|---|---|---|\n"
what = {
 "r03_bench.json": f"bench.py JSON line ({d['value'] / 1000:.2f} TFLOP/s, {d['ms_per_step']:.2f} ms/step; copy rate of that box {d['roofline']['copy_GBps_same_box'] / 1000:.2f} TB/s) | `python bench.py`",
 "r03_bench_kernel_stats.csv": "per-kernel time of the headline benchmark | `cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline`",
 "r03_config34_kernel_stats.csv": "per-kernel time of the configs-3/4 sweep (the new families included) | the same around `python tools/perf_configs.py`",
 "r03_pmc_traffic.json": "bytes per launch at the L2↔fabric boundary, keyed to the hash of the sources of the build (`source_hash`; bench.py reports `traffic` only when it matches) | `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (one pass each) `-- python tools/pmc_probe.py`, `tools/summarize_profiles.py r03 <dir>`",
 "r03_mall_evidence.jsonl": "HBM-level argument for the fused kernel, 2^16 … 2^22, queue count fixed per size: rings of 128 MiB … 1 GiB | `python tools/mall_evidence.py`",
 "r03_fused_gen1_per_phase_cycle_profile_2p15_to_2p22.txt": "per-phase cycle profile of the product fused kernel (development build) | `VKFFT_MI355X_LIB=build/libvkfft_mi355x_dev.so python tools/prof_fused.py 15 22`",
 "r03_probe_ring_trip_ceilings.jsonl": "what a ring trip costs with NO arithmetic, tickets or flags: 3.5–3.66 TB/s through the Infinity Cache whatever the structure; 4.7–6.1 through an XCD-private L2 ring | `build/probe_dma` (`tools/probe_dma.hip`)",
 "r03_probe_dma_lds_dma_semantics_and_column_streams.jsonl": "LDS-DMA (`buffer_load … lds`) semantics on gfx950 and column-tile stream rates (5.8–6.0 TB/s; 64-byte segments 2.6) | `build/probe_dma`",
 "r03_fused_gen2_dev_profile_2p16.jsonl, r03_fused_gen2_variants_2p16.jsonl, r03_fused_gen2b_*": "second-generation fused kernel (LDS-DMA double buffering; service-wave form; sliced DMA): bit-identical, not faster than generation 1 | `python tools/exp_fused2.py` on the development library",
 "r03_fused_gen1_small_xcd_local_ring_experiment.jsonl": "generation 1 with plain ring stores and rings of 2–6 MiB per XCD: 1.35–2.33 TB/s (dependency stalls) | `tools/exp_fused2.py`",
 "r03_fused_nt_hint_per_side.jsonl": "the non-temporal hint on both sides (product) / neither / loads only / stores only, 2^15 … 2^22: both is best everywhere | `VKFFT_MI355X_LIB=build/libvkfft_mi355x_dev.so python tools/ab_nt.py 15 22`",
 "r03_mixconv_rows_family_on_vs_off.jsonl, r03_mixconv_columns_family_on_vs_off.jsonl": "the Rader / smooth-Bluestein kernel family forced on vs off (the planner's cost factors come from these) | `python tools/tune_mixconv.py rows` / `cols`",
 "r03_all_lengths_2_320_with_reference_same_call.jsonl, r03_all_lengths_2_320_before_single_buffer_rows.jsonl": "1-D C2C of EVERY length 2 … 320 beside the reference (2^25 points): 0.99× after / 0.90× before the short rows shared one LDS buffer | `python tools/perf_all_short.py 2 320`",
 "r03_r2c_rows_4_400_with_reference_same_call.jsonl, r03_dct2_rows_4_400_with_reference_same_call.jsonl, r03_dct4_rows_5_400_with_reference_same_call.jsonl, r03_r2c_rows_4_400_before_*, r03_dct2_rows_4_400_before_*": "real rows of arbitrary length beside the reference: R2C 0.44× (0.25× before the instance transform ran between the interpreter's maps), DCT-II 0.37× (0.22×), DCT-IV 0.40× | `python tools/perf_all_short.py 4 400 1 3` (12, 14: DCT-II, DCT-IV)",
 "r03_short_real_rows_fused_maps_vs_instance_between_maps.jsonl": "fused-map kernels vs the instance transform between the interpreter's maps on short real rows (the planner's threshold) | `python tools/tune_mixed_ops.py`",
 "r03_short_rows_with_reference_same_call.jsonl": "1-D rows of 4 … 128 points and small planes / cubes | `python tools/perf_small_rows.py`",
 "r03_planes_of_smooth_lengths_outside_the_curated_lists_with_reference_same_call.jsonl": "84², 168², 252², 84³ … C2C 0.94–1.09× after the plain column kernels got an instance for every 13-smooth length ≤ 1024; R2C / DCT planes of such lengths 0.4–0.8× (real rows between the generic maps, strided DCT axes on the interpreter) | `python tools/perf_odd_planes.py`",
 "r03_zero_padding_with_reference_same_call.jsonl": "zero-padded 3-D / 2-D systems, padded vs unpadded, reference in the same process | `python tools/perf_zeropad.py`",
 "r03_convolution_with_reference_same_call.jsonl": "convolution plans with the merged last axis, reference's merged kernels in the same process | `python tools/perf_conv.py`",
 "r03_multi_gpu_cxx_drivers_one_gpu_box.jsonl": "C++ drivers: 4 virtual ranks verified against a single-device plan, one rank over RCCL, batch sharding | `build/vkfft_mi355x_multi …`",
 "r03_cli_1024cube_one_gpu_64bit_column_kernel.txt": "1024³ C2C on ONE GPU after the 64-bit column kernel: 31.3 ms per forward+inverse (273 ms in round 2) | `build/vkfft_mi355x_cli -benchmark_vkfft -X 1024 -Y 1024 -Z 1024 -N 3`",
 "r03_gpu_suite.log": "tail of `pytest -m gpu` on the device | `python -m pytest tests -x -q -m gpu`",
 "r03_config34_with_reference_same_call.jsonl": f"configs 3/4, reference in the same process: geometric mean {g(cfg):.2f} | `python tools/perf_configs.py`",
 "r03_samples_3_6_100_with_reference_same_call.jsonl": f"samplings of the reference's multi-dimensional benchmark lists: geometric mean {g(sm):.2f} | `python tools/perf_samples.py`",
 "r03_sample7_prime_planes_with_reference_same_call.jsonl": f"the reference's sample 7 (prime planes and cubes): geometric mean {g(s7):.2f} (round 2: 0.75) | `python tools/perf_sample7.py`",
 "r03_sample1000_sampling_with_reference_same_call.jsonl": f"a sampling of the reference's sample 1000 (every length 2 … 4096): geometric mean {g(s1000):.2f} | `python tools/perf_sample1000.py 60`",
}
files = "".join(f"| `{k}` | {v.split(' | ')[0]} | {v.split(' | ')[1]} |\n" for k, v in what.items()
                if any(os.path.exists(os.path.join(P, n.strip().replace('*', ''))) or glob.glob(os.path.join(P, n.strip())) for n in k.split(",")))
new = f'''## Round 3

"reference in the same process" = the reference VkFFT-HIP (`oracle/_ref`, built by `oracle/build_ref.sh`) timed right after this library on the same shapes with the same
protocol; ratios > 1 mean this library is faster.  (Regenerate this section with `tools/gen_profiles_readme.py`.)

| file | what | command |
|---|---|---|
{files}
Headline sweep, algorithmic GB/s of an FFT+iFFT pair (1 GiB); reference from round 1's table (same protocol, its own binary):

| log2 N | reference VkFFT-HIP | round 2 | round 3 | round 3 forward-only | round 3 / reference |
|---|---|---|---|---|---|
{rows}
Configs 3/4 (`r03_config34_with_reference_same_call.jsonl`):

{hdr}{table(cfg)}
Sample 7 (`r03_sample7_…jsonl`):

{hdr}{table(s7)}
Sample 1000, the sampled lengths (`r03_sample1000_…jsonl`):

{hdr}{table(s1000)}
Samplings of the reference's multi-dimensional benchmark lists (`r03_samples_3_6_100_…jsonl`):

{hdr}{table(sm)}'''
open(os.path.join(P, "README.md"), "w").write(s + new)
print("profiles/README.md: round-3 section rewritten")
