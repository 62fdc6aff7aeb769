export TMPDIR=/tmp; O=gpurun_out/r04d; mkdir -p $O
timeout 130 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -n 8 -k "(two_real_rows and not 4099) or never_touches or test_bluestein_fp32 or dct4_dst4 or native_library or r2c_c2r" > $O/tests.log 2>&1; tail -3 $O/tests.log
timeout 60 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 200 $O/bench.json
(cd /tmp && timeout 60 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/prof_bench -- python /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /root/repo/$O/prof_bench.log 2>&1)
(cd /tmp && VKFFT_PMC_HASH_FILE=/root/repo/$O/pmc_source_hash.txt timeout 40 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /root/repo/$O/pmc_fetch -- python /root/repo/tools/pmc_probe.py > /root/repo/$O/pmc_fetch.log 2>&1)
(cd /tmp && timeout 40 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /root/repo/$O/pmc_write -- python /root/repo/tools/pmc_probe.py > /root/repo/$O/pmc_write.log 2>&1)
timeout 40 python tools/perf_real_sweep.py r2c 6 > $O/r2c.jsonl 2> $O/r2c.err
timeout 60 python tools/perf_sample1000.py 60 > $O/sample1000.jsonl 2> $O/sample1000.err
timeout 30 python tools/perf_real_sweep.py dct4 6 > $O/dct4.jsonl 2> $O/dct4.err
timeout 40 python tools/perf_real_sweep.py dct2 6 > $O/dct2.jsonl 2> $O/dct2.err
timeout 60 python tools/perf_configs.py > $O/config34.jsonl 2> $O/config34.err
ls $O
