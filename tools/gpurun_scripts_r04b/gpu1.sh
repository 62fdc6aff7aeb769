export TMPDIR=/tmp; mkdir -p gpurun_out/s2a
timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -n 4 -k "two_real_rows or dct4_dst4_of_odd or never_touches or native_library or real_transforms or dct_dst or r2c_c2r or r2r_whose" > gpurun_out/s2a/new_tests.log 2>&1
tail -5 gpurun_out/s2a/new_tests.log
timeout 170 python tools/perf_real_sweep.py r2c 6 > gpurun_out/s2a/r2c.jsonl 2> gpurun_out/s2a/r2c.err
timeout 170 python tools/perf_real_sweep.py dct4 6 > gpurun_out/s2a/dct4.jsonl 2> gpurun_out/s2a/dct4.err
timeout 170 python tools/perf_real_sweep.py dct2 6 > gpurun_out/s2a/dct2.jsonl 2> gpurun_out/s2a/dct2.err
timeout 60 python tools/perf_real_rows.py 14:1451 14:1125 1:235 12:235 14:235 14:30 14:20 12:28 > gpurun_out/s2a/rows.jsonl 2> gpurun_out/s2a/rows.err
wc -l gpurun_out/s2a/*.jsonl
