export TMPDIR=/tmp; O=gpurun_out/r04e; mkdir -p $O
timeout 110 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -n 8 --durations=8 -k "two_real_rows" > $O/tests.log 2>&1; tail -14 $O/tests.log
