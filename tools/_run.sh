cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 1200 python -m pytest tests -x -q -m gpu -k "bluestein or rader" 2>&1 | tail -4 ) 2>&1 | tee gpurun_out/gpu_tests_blue.log
NO_REF=1 timeout 900 python tools/perf_configs.py 34 36 2>&1 | grep "^{" | tee gpurun_out/perf_configs_blue3.jsonl
