export TMPDIR=/tmp; mkdir -p gpurun_out/s2b
timeout 100 python tools/perf_real_sweep.py r2c 8 > gpurun_out/s2b/r2c.jsonl 2> gpurun_out/s2b/r2c.err
timeout 100 python tools/perf_real_sweep.py dct4 8 > gpurun_out/s2b/dct4.jsonl 2> gpurun_out/s2b/dct4.err
timeout 100 python tools/perf_real_sweep.py dct2 8 > gpurun_out/s2b/dct2.jsonl 2> gpurun_out/s2b/dct2.err
timeout 40 python tools/perf_real_rows.py 14:1451 14:1125 14:30 14:20 12:28 1:169 12:169 1:4095 > gpurun_out/s2b/rows.jsonl 2> gpurun_out/s2b/rows.err
wc -l gpurun_out/s2b/*.jsonl
