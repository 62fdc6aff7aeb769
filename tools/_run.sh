cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 ) 2>&1 | tee gpurun_out/gpu_tests.log
NO_REF=1 timeout 600 python tools/perf_configs.py 10 14 2>&1 | grep "^{" | tee gpurun_out/perf_configs_b.jsonl
NO_REF=1 timeout 600 python tools/perf_configs.py 18 28 2>&1 | grep "^{" | tee -a gpurun_out/perf_configs_b.jsonl
