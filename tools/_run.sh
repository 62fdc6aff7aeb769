cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 ) 2>&1 | tee gpurun_out/gpu_tests.log
timeout 600 python bench.py 2>&1 | tail -1 | tee gpurun_out/bench.json
NO_REF=1 timeout 600 python tools/perf_configs.py 0 24 2>&1 | grep "^{" | tee gpurun_out/perf_configs.jsonl
