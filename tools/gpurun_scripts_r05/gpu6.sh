# round 5, call 7: regenerate the reference-held fixtures (tests/golden/make_golden.py, 43 cases) and run the fixture test against them
export TMPDIR=/tmp; O=gpurun_out/r05g; mkdir -p $O
timeout 300 python tests/golden/make_golden.py $O/ref_vkfft.npz > $O/make_golden.log 2>&1; tail -25 $O/make_golden.log
cp $O/ref_vkfft.npz tests/golden/ref_vkfft.npz
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "golden_reference_fixtures or live_reference" > $O/tests.log 2>&1; tail -5 $O/tests.log
ls -la $O
