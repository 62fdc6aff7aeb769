export TMPDIR=/tmp; O=gpurun_out/r04h; mkdir -p $O
timeout 55 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --durations=5 -k "reference_live or live_reference or chip_full or preferred_over" > $O/tests.log 2>&1; tail -12 $O/tests.log
