cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 1200 python -m pytest tests -x -q -m gpu -k "opfft or r2c or dct or dst or multidim" 2>&1 | tail -8 ) 2>&1 | tee gpurun_out/gpu_tests_opfft.log
NO_REF=1 timeout 600 python tools/perf_configs.py 18 24 2>&1 | grep "^{" | tee gpurun_out/perf_configs_real.jsonl
