# round 5, call 1: device parity of the packed-pair kernels, A/B against the round-4 shapes, first bench line, the reference's sweep in the same lease
export TMPDIR=/tmp; O=gpurun_out/r05a; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -n 4 -k "fused or register_lean or full_size or zero_padding_never or native_library" > $O/tests.log 2>&1; tail -6 $O/tests.log
timeout 300 python tools/ab_r05.py > $O/ab.jsonl 2> $O/ab.err; tail -3 $O/ab.err
timeout 300 python bench.py --steps 3 --warmup 1 > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err
timeout 200 oracle/_ref/vkfft_ref_bench 8 22 0 > $O/reference_pow2_same_lease.jsonl 2> $O/ref.err
cat $O/ab.jsonl | cut -c1-260
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05a/bench.json"))
print(d["value"], d["ms_per_step"], {k: (v["alg_GBps"], v["fwd_only_alg_GBps"], v["kernel"][:28]) for k, v in d["per_size"].items()})
print(d["roofline"])
PY
tail -20 $O/reference_pow2_same_lease.jsonl | cut -c1-200
