# round 5: 2^20 as tiles of two 16-column halves (256-byte segments on both HBM sides) against the 16-column pipelined kernel; 2^21 / 2^22 after the reordering of the second look-up
export TMPDIR=/tmp; O=gpurun_out/r05q; mkdir -p $O
timeout 100 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -n 4 -k "fused_fourstep_every_registered_shape_on_device and (19 or 20)" > $O/tests.log 2>&1; tail -3 $O/tests.log
timeout 200 python tools/ab_r05.py 19 20 > $O/ab.jsonl 2> $O/ab.err; timeout 200 python tools/ab_r05.py 19 >> $O/ab.jsonl 2>> $O/ab.err
cut -c1-200 $O/ab.jsonl
