#!/bin/bash
# 2^11 / 2^12: the round-1 row kernel (default) against the packed multi-row form (registry index 1), three interleaved repeats, the reference in the same lease
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ab1112
for i in 1 2 3; do timeout 120 python tools/ab_r05.py 11 12 >> gpurun_out/ab1112/ab.jsonl 2>> gpurun_out/ab1112/err; done
timeout 100 oracle/_ref/vkfft_ref_bench 11 12 0 > gpurun_out/ab1112/ref.jsonl 2>> gpurun_out/ab1112/err
cut -c1-160 gpurun_out/ab1112/ab.jsonl; cut -c1-200 gpurun_out/ab1112/ref.jsonl
