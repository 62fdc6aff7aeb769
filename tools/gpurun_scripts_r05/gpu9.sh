# round 5: the whole device suite on the current sources
export TMPDIR=/tmp; O=gpurun_out/r05s; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -n 8 > $O/gpu_suite.log 2>&1; tail -15 $O/gpu_suite.log
