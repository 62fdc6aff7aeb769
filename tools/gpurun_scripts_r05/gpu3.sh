# round 5, call 3: tiles of two halves with the reworked request order, transposition pitch for 32-column tiles
export TMPDIR=/tmp; O=gpurun_out/r05d; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -n 4 -k "fused_fourstep_every_registered_shape_on_device or fused_fourstep_equals" > $O/tests.log 2>&1; tail -4 $O/tests.log
timeout 300 python tools/ab_r05.py 16 17 18 19 20 21 22 > $O/ab.jsonl 2> $O/ab.err; tail -3 $O/ab.err
cut -c1-230 $O/ab.jsonl
export VKFFT_MI355X_LIB=vkfft_amd/lib/libvkfft_mi355x_dev.so
for k in 16 18 20 21 22; do
  echo "== $k packed (FUV$k=0)" >> $O/phase.txt
  timeout 60 python tools/prof_fused.py $k $k ROW15=0 > $O/tmp.log 2>&1; grep "fused profile" $O/tmp.log | head -2 >> $O/phase.txt; grep -i "error\|Traceback" -A3 $O/tmp.log | head -5 >> $O/phase.txt
done
cat $O/phase.txt
