# round 5, call 5: ring / lag tuning of the packed kernels at 2^19 ... 2^22, stress under unbalanced queues
export TMPDIR=/tmp; O=gpurun_out/r05e; mkdir -p $O
timeout 200 python tools/ab_r05.py 19 20 21 22 > $O/ab.jsonl 2> $O/ab.err
timeout 400 python tools/ab_r05.py tune 20 21 22 > $O/tune.jsonl 2>> $O/ab.err
timeout 300 python tools/ab_r05.py stress > $O/stress.jsonl 2>> $O/ab.err
cut -c1-200 $O/ab.jsonl $O/tune.jsonl $O/stress.jsonl
