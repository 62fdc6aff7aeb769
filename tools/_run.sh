cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_bench gpurun_out/prof_cfg
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 ) 2>&1 | tee gpurun_out/gpu_tests.log
timeout 900 python bench.py 2>gpurun_out/bench_latest.err | tail -1 > gpurun_out/bench_latest.json
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_bench -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_bench/bench.log 2>&1 )
( cd /tmp && NO_REF=1 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_cfg -o cfg -- python $GRAFT_REPO_ROOT/tools/perf_configs.py 0 36 > $GRAFT_REPO_ROOT/gpurun_out/prof_cfg/cfg.log 2>&1 )
find gpurun_out/prof_bench gpurun_out/prof_cfg -name "*kernel_trace*" -delete; find gpurun_out/prof_bench gpurun_out/prof_cfg -name "*agent_info*" -delete
ls gpurun_out/prof_bench gpurun_out/prof_cfg
