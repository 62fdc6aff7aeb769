#!/bin/bash
# the real-transform lines of BASELINE configs 3 / 4 on the final sources (reference in the same process)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05v
timeout 100 python tools/perf_configs.py 18 28 > gpurun_out/r05v/config34_real.jsonl 2> gpurun_out/r05v/err
timeout 60 python tools/perf_configs.py 39 41 >> gpurun_out/r05v/config34_real.jsonl 2>> gpurun_out/r05v/err
cut -c1-230 gpurun_out/r05v/config34_real.jsonl
