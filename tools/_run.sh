R=$GRAFT_REPO_ROOT
cd $R
timeout 1400 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
python bench.py --steps 5 --warmup 2 > gpurun_out/bench_latest.json 2>gpurun_out/bench_latest.err
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_bench $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_bench_stdout.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch -o f -- python $R/tools/pmc_probe.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write -o w -- python $R/tools/pmc_probe.py > /dev/null 2>&1
