"""Development tool (round 4): the three real-row sweeps of the round-3 review — R2C and DCT-II rows of 4 ... 400 reals (step 3), DCT-IV rows of 5 ... 400 (step 5) —
in up to three plans per length: the default, VKFFT_MI355X_NO_ROW_PAIRS=1 (one row per transform, round 3's form) and VKFFT_MI355X_PAIR_PREFER=1 (paired rows
between the generic maps also where a fused-map instance exists).  The reference is timed in the same process on every `REF_EVERY`-th length (its plans are
compiled at run time: about a second each); for the others the ratio column uses the round-4 sweep already in profiles/.
python tools/perf_real_sweep.py <r2c|dct2|dct4> [REF_EVERY]"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
which = sys.argv[1]
ref_every = int(sys.argv[2]) if len(sys.argv) > 2 else 4
import perf_configs
from perf_configs import run
kind = {"r2c": 1, "dct2": 12, "dct4": 14}[which]
lengths = list(range(5, 401, 5)) if which == "dct4" else list(range(4, 401, 3))
old = {}
fn = os.path.join(ROOT, "profiles", {"r2c": "r04_r2c_rows_4_400", "dct2": "r04_dct2_rows_4_400", "dct4": "r04_dct4_rows_5_400"}[which] + "_with_reference_same_call.jsonl")
if os.path.exists(fn):
    for l in open(fn):
        if l.startswith("{"):
            r = json.loads(l)
            if "ref_pair_ms" in r:
                old[r["shape"][0]] = r
saved_ref = perf_configs.ref
for i, n in enumerate(lengths):
    rec = {"kind": kind, "N": n}
    modes = (("default", {}), ("one_row_per_transform", {"VKFFT_MI355X_NO_ROW_PAIRS": "1"}), ("pairs_preferred", {"VKFFT_MI355X_PAIR_PREFER": "1"}))
    for tag, env in (modes if n % 2 else modes[:1]):  # (rows of even length run their half-length forms: nothing to pair)
        for k in ("VKFFT_MI355X_NO_ROW_PAIRS", "VKFFT_MI355X_PAIR_PREFER"):
            os.environ.pop(k, None)
        os.environ.update(env)
        perf_configs.ref = saved_ref if (tag == "default" and i % ref_every == 0) else None
        try:
            r = run(kind, (n,), False, total_log2=25)
        except Exception as ex:
            rec[tag + "_error"] = str(ex); continue
        rec[tag + "_ms"] = r["pair_ms"]
        if "ref_pair_ms" in r:
            rec["ref_ms"] = r["ref_pair_ms"]
    if n in old:
        rec["ref_ms_round4_sweep"] = old[n]["ref_pair_ms"]; rec["ours_ms_round4_sweep"] = old[n]["pair_ms"]
    print(json.dumps(rec), flush=True)
