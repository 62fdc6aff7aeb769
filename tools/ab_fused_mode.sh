#!/bin/bash
# development: bench.py with each fused-kernel mode given on the command line, per-size pair rates side by side
mkdir -p gpurun_out/ab
for m in "$@"; do
  VKFFT_MI355X_FUSED_MODE=$m timeout 250 python bench.py 2>/dev/null | tail -1 > gpurun_out/ab/bench_m$m.json
  python -c "
import json
d=json.load(open('gpurun_out/ab/bench_m$m.json'))
print('mode $m', d['value'], d['ms_per_step'])
ps=d.get('per_size')
print(ps if not isinstance(ps,dict) else {k:(round(v['alg_GBps']) if isinstance(v,dict) else v) for k,v in ps.items()})
"
done
