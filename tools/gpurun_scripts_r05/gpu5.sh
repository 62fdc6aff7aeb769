# round 5, call 6: lag / ring pairs of the packed kernels at 2^19 ... 2^22 (is the ring budget about the slots between write and read, or about all slots?)
export TMPDIR=/tmp; O=gpurun_out/r05f; mkdir -p $O
timeout 600 python tools/ab_r05.py lags > $O/lags.jsonl 2> $O/lags.err
cut -c1-200 $O/lags.jsonl
