#!/bin/bash
# the three real-row sweeps of the review, default plans, reference in the same process on every length
cd $GRAFT_REPO_ROOT; O=gpurun_out/tm20; mkdir -p $O
for f in r2c dct2 dct4; do timeout 700 python tools/perf_real_sweep_r05.py $f > $O/${f}_rows.jsonl 2> $O/$f.err; done
python - <<'PY'
import json, math, collections
for f in ('r2c','dct2','dct4'):
    rows=[json.loads(l) for l in open(f'gpurun_out/tm20/{f}_rows.jsonl')]
    rs=[(r['ref_ms']/r['ms'],r['N'],r['kernel']) for r in rows if r.get('ref_ms')]
    g=math.exp(sum(math.log(x[0]) for x in rs)/len(rs))
    print(f,len(rs),'geomean %.3f'%g,'below 0.5:',sum(1 for x in rs if x[0]<0.5),'min',min(rs))
    by=collections.defaultdict(list)
    for x,n,k in rs: by[k].append(x)
    for k,v in by.items(): print('   ',k,len(v),'geo %.2f'%math.exp(sum(math.log(x) for x in v)/len(v)),'min %.2f'%min(v))
PY
