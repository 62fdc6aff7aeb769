export TMPDIR=/tmp; O=gpurun_out/r04c; mkdir -p $O
timeout 170 python tools/perf_configs.py > $O/config34.jsonl 2> $O/config34.err
timeout 150 python tools/perf_sample1000.py 60 > $O/sample1000.jsonl 2> $O/sample1000.err
wc -l $O/*.jsonl
