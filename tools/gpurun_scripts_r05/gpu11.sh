# round 5: configs 3 / 4 and the sampling of sample 1000 with the reference in the same process (the kernels of these paths did not change this round: regression record)
export TMPDIR=/tmp; O=gpurun_out/r05n; mkdir -p $O
timeout 400 python tools/perf_configs.py > $O/config34.jsonl 2> $O/config34.err; tail -2 $O/config34.err
timeout 300 python tools/perf_sample1000.py 60 > $O/sample1000.jsonl 2> $O/sample1000.err; tail -2 $O/sample1000.err
python - <<'PY'
import json, math
for f in ("config34", "sample1000"):
    r = [json.loads(l) for l in open(f"gpurun_out/r05n/{f}.jsonl") if l.startswith("{")]
    x = [a["alg_GBps"] / a["ref_alg_GBps"] for a in r if a.get("ref_alg_GBps")]
    print(f, len(x), "geometric mean", round(math.exp(sum(map(math.log, x)) / len(x)), 3), "min", round(min(x), 2), "below 0.7:", sum(1 for v in x if v < 0.7))
PY
