cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -x -q -m gpu -k "rader or bluestein or opfft" 2>&1 | tail -3
for b in 1.4 0.5 10; do echo "BIAS=$b"; VKFFT_MI355X_BLUE_BIAS=$b NO_REF=1 timeout 600 python tools/perf_configs.py 10 14 2>&1 | grep "^{"; done | tee gpurun_out/perf_blue_ab.log
