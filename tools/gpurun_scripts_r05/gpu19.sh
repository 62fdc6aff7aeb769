#!/bin/bash
# quick check of the lengths the last step touched (compact copy-out, even DCT-IV with few threads back on the generic maps)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/tm21
timeout 300 python tools/perf_real_rows.py 1:31 1:91 1:13 1:55 14:20 14:30 12:31 12:91 1:19 1:85 1:169 12:169 > gpurun_out/tm21/a.jsonl 2> gpurun_out/tm21/a.err
cut -c1-200 gpurun_out/tm21/a.jsonl
