# round 5, call 8: first full evidence set (bench line, reference sweep in the same lease, rocprofv3 kernel stats, PMC passes) + a baseline of real rows
export TMPDIR=/tmp; O=gpurun_out/r05h; mkdir -p $O
timeout 300 python bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err
timeout 200 oracle/_ref/vkfft_ref_bench 8 22 0 > $O/reference_pow2_same_lease.jsonl 2> $O/ref.err
timeout 120 python tools/perf_real_rows.py 1:169 0:169 12:169 1:385 0:385 1:100 0:100 12:100 1:31 12:16 1:265 1:64 12:64 14:145 > $O/rows.jsonl 2> $O/rows.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/prof_bench.log 2>&1
cd $GRAFT_REPO_ROOT
export VKFFT_PMC_HASH_FILE=$GRAFT_REPO_ROOT/$O/pmc_source_hash.txt
cd /tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_fetch -- python $GRAFT_REPO_ROOT/tools/pmc_probe.py > $GRAFT_REPO_ROOT/$O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_write -- python $GRAFT_REPO_ROOT/tools/pmc_probe.py > $GRAFT_REPO_ROOT/$O/pmc_write.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05h/bench.json"))
print(d["value"], d["ms_per_step"], {k: (v["alg_GBps"], v["fwd_only_alg_GBps"]) for k, v in d["per_size"].items()})
print(d["roofline"])
PY
cut -c1-250 $O/rows.jsonl
find $O -name "*.csv" | head; du -sh $O
