# round 5: 2^15 as pairs of samples, persistent and pipelined over the rows, against the one-row kernels
export TMPDIR=/tmp; O=gpurun_out/r05o; mkdir -p $O
timeout 100 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -n 4 -k "register_lean_rows_every_variant_on_device" > $O/tests.log 2>&1; tail -3 $O/tests.log
timeout 200 python tools/ab_r05.py 15 > $O/ab.jsonl 2> $O/ab.err; timeout 200 python tools/ab_r05.py 15 >> $O/ab.jsonl 2>> $O/ab.err
cut -c1-200 $O/ab.jsonl
