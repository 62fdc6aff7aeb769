# round 5, call 2: per-phase cycle profile of the packed kernels (development library)
export TMPDIR=/tmp; O=gpurun_out/r05b; mkdir -p $O
export VKFFT_MI355X_LIB=vkfft_amd/lib/libvkfft_mi355x_dev.so
for k in 16 18 19 20 21 22; do
  echo "== $k packed (FUV$k=0)" >> $O/phase.txt
  timeout 60 python tools/prof_fused.py $k $k ROW15=0 > $O/tmp.log 2>&1; grep "fused profile" $O/tmp.log | head -2 >> $O/phase.txt; grep -i "error\|Traceback" -A3 $O/tmp.log | head -5 >> $O/phase.txt
done
echo "== 20 round-4 pipelined form has no counters; round-2 form FUV20=2" >> $O/phase.txt
timeout 60 python tools/prof_fused.py 20 20 FUV20=2 > $O/tmp.log 2>&1; grep "fused profile" $O/tmp.log | head -2 >> $O/phase.txt; grep -i "error\|Traceback" -A3 $O/tmp.log | head -5 >> $O/phase.txt
echo "== 22 round-4 form FUV22=1" >> $O/phase.txt
timeout 60 python tools/prof_fused.py 22 22 FUV22=1 > $O/tmp.log 2>&1; grep "fused profile" $O/tmp.log | head -2 >> $O/phase.txt; grep -i "error\|Traceback" -A3 $O/tmp.log | head -5 >> $O/phase.txt
echo "== 21 second orientation FUV21=1" >> $O/phase.txt
timeout 60 python tools/prof_fused.py 21 21 FUV21=1 > $O/tmp.log 2>&1; grep "fused profile" $O/tmp.log | head -2 >> $O/phase.txt; grep -i "error\|Traceback" -A3 $O/tmp.log | head -5 >> $O/phase.txt
cat $O/phase.txt
