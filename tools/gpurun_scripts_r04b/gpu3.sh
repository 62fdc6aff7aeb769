export TMPDIR=/tmp; O=gpurun_out/r04b; mkdir -p $O
timeout 540 python -m pytest tests -m gpu -q -n 8 > $O/gpu_suite.log 2>&1; tail -4 $O/gpu_suite.log
timeout 170 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 400 $O/bench.json
(cd /tmp && timeout 170 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/prof_bench -- python /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /root/repo/$O/prof_bench.log 2>&1)
(cd /tmp && VKFFT_PMC_HASH_FILE=/root/repo/$O/pmc_source_hash.txt timeout 110 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /root/repo/$O/pmc_fetch -- python /root/repo/tools/pmc_probe.py > /root/repo/$O/pmc_fetch.log 2>&1)
(cd /tmp && timeout 110 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /root/repo/$O/pmc_write -- python /root/repo/tools/pmc_probe.py > /root/repo/$O/pmc_write.log 2>&1)
timeout 95 python tools/perf_real_sweep.py r2c 6 > $O/r2c.jsonl 2> $O/r2c.err
timeout 95 python tools/perf_real_sweep.py dct4 6 > $O/dct4.jsonl 2> $O/dct4.err
timeout 95 python tools/perf_real_sweep.py dct2 6 > $O/dct2.jsonl 2> $O/dct2.err
timeout 60 python tools/perf_real_rows.py 14:1451 14:1125 1:235 12:235 14:235 14:30 14:20 12:28 1:169 12:169 1:4095 1:4096 12:4096 > $O/rows.jsonl 2> $O/rows.err
ls $O; du -sh $O
