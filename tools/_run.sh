cd /tmp && export TMPDIR=/tmp
export VKFFT_MI355X_CHUNK_MIB=0
for k in 12 16 20 22; do
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof$k -o p$k -- python $GRAFT_REPO_ROOT/tools/perf_sweep.py $k $k > /dev/null 2>&1
done
find $GRAFT_REPO_ROOT/gpurun_out/ -name "*kernel_stats.csv" | while read f; do echo $f; cut -d, -f1-8 $f | head -8; done
