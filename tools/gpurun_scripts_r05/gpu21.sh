#!/bin/bash
# round 5, after the last source change (2^11 / 2^12 default = packed rows; planner guard): bench line, reference sweep in the same lease, rocprofv3 kernel stats, PMC passes,
# and the device tests of the power-of-two rows and the real transforms
export TMPDIR=/tmp; O=gpurun_out/r05y; mkdir -p $O
timeout 300 python bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err
timeout 200 oracle/_ref/vkfft_ref_bench 8 22 0 > $O/reference_pow2_same_lease.jsonl 2> $O/ref.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/prof_bench.log 2>&1
export VKFFT_PMC_HASH_FILE=$GRAFT_REPO_ROOT/$O/pmc_source_hash.txt
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_fetch -- python $GRAFT_REPO_ROOT/tools/pmc_probe.py > $GRAFT_REPO_ROOT/$O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_write -- python $GRAFT_REPO_ROOT/tools/pmc_probe.py > $GRAFT_REPO_ROOT/$O/pmc_write.log 2>&1
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -q -n 8 -k "pow2 or lean or golden or real or r2c or dct or dst or two_rows or r2r or padding or stride" > $O/gpu_subset.log 2>&1; tail -3 $O/gpu_subset.log
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05y/bench.json"))
print(d["value"], d["ms_per_step"], {k: v["alg_GBps"] for k, v in d["per_size"].items()})
print(d["roofline"])
PY
find $O -name "*.db" -delete; du -sh $O
