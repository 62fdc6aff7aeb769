cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 1200 python -m pytest tests -x -q -m gpu -k "opfft or multi_pass or fourstep or radix" 2>&1 | tail -4 ) 2>&1 | tee gpurun_out/gpu_tests_opfft.log
NO_REF=1 timeout 900 python tools/perf_configs.py 28 34 2>&1 | grep "^{" | tee gpurun_out/perf_configs_multipass2.jsonl
