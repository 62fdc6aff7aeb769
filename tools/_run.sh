cd $GRAFT_REPO_ROOT
timeout 300 python tools/gpu_check.py > gpurun_out/check9.log 2>&1; tail -1 gpurun_out/check9.log; grep -E "BAD|EXC" gpurun_out/check9.log | head -5
timeout 300 python tools/gpu_check.py real > gpurun_out/check9r.log 2>&1; tail -1 gpurun_out/check9r.log; grep -E "BAD|EXC" gpurun_out/check9r.log | head -5
NO_REF=1 timeout 600 python tools/perf_configs.py 2>&1 | grep "^{"
