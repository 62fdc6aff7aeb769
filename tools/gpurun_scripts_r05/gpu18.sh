#!/bin/bash
# round 5, final evidence set on the final sources: bench line, reference sweep in the same lease, rocprofv3 kernel stats, PMC passes, the three real-row sweeps, the device suite
export TMPDIR=/tmp; O=gpurun_out/r05z; mkdir -p $O
timeout 300 python bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err
timeout 200 oracle/_ref/vkfft_ref_bench 8 22 0 > $O/reference_pow2_same_lease.jsonl 2> $O/ref.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/prof_bench.log 2>&1
export VKFFT_PMC_HASH_FILE=$GRAFT_REPO_ROOT/$O/pmc_source_hash.txt
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_fetch -- python $GRAFT_REPO_ROOT/tools/pmc_probe.py > $GRAFT_REPO_ROOT/$O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_write -- python $GRAFT_REPO_ROOT/tools/pmc_probe.py > $GRAFT_REPO_ROOT/$O/pmc_write.log 2>&1
cd $GRAFT_REPO_ROOT
for f in r2c dct2 dct4; do timeout 500 python tools/perf_real_sweep_r05.py $f > $O/${f}_rows.jsonl 2> $O/$f.err; done
timeout 1500 python -m pytest tests -m gpu -q -n 8 > $O/gpu_suite.log 2>&1; tail -4 $O/gpu_suite.log
python - <<'PY'
import json, math, collections
d = json.load(open("gpurun_out/r05z/bench.json"))
print(d["value"], d["ms_per_step"], {k: v["alg_GBps"] for k, v in d["per_size"].items()})
print(d["roofline"])
for f in ('r2c','dct2','dct4'):
    rows=[json.loads(l) for l in open(f'gpurun_out/r05z/{f}_rows.jsonl')]
    rs=[(r['ref_ms']/r['ms'],r['N'],r['kernel']) for r in rows if r.get('ref_ms')]
    g=math.exp(sum(math.log(x[0]) for x in rs)/len(rs))
    print(f,len(rs),'geomean %.3f'%g,'below 0.5:',sum(1 for x in rs if x[0]<0.5),'min',min(rs))
PY
find $O -name "*.db" -delete; du -sh $O
