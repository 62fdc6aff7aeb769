cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 1200 python -m pytest tests -x -q -m gpu -k "opfft or dct or dst" 2>&1 | tail -4 ) 2>&1 | tee gpurun_out/gpu_tests_opfft.log
NO_REF=1 timeout 600 python tools/perf_configs.py 19 20 2>&1 | grep "^{" | tee gpurun_out/perf_configs_real.jsonl
NO_REF=1 timeout 600 python tools/perf_configs.py 22 28 2>&1 | grep "^{" | tee -a gpurun_out/perf_configs_real.jsonl
