export TMPDIR=/tmp; O=gpurun_out/r04f; mkdir -p $O
timeout 70 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -n 4 -k "against_the_reference_live" > $O/tests.log 2>&1; tail -12 $O/tests.log
