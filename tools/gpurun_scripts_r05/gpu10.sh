# round 5: real rows between the generic maps with one element of look-ahead in the pre-map loop (baseline: profiles/r05_real_rows_selected_baseline.jsonl), parity of that path
export TMPDIR=/tmp; O=gpurun_out/r05l; mkdir -p $O
timeout 200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -n 4 -k "two_real_rows or real_rows or r2c_c2r or dct_dst" > $O/tests.log 2>&1; tail -3 $O/tests.log
timeout 120 python tools/perf_real_rows.py 1:169 0:169 12:169 1:385 0:385 1:100 0:100 12:100 1:31 12:16 1:265 1:64 12:64 14:145 > $O/rows.jsonl 2> $O/rows.err
cut -c1-250 $O/rows.jsonl
