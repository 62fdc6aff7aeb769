#!/bin/bash
# kernel durations and SQ / cache counters of the 169-point real rows (table-driven maps) next to the complex rows
export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/tm15; mkdir -p $O; cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $GRAFT_REPO_ROOT/tools/pmc_rows.py 169 > $O/trace.log 2>&1
timeout 60 rocprofv3 --list-avail > $O/avail.txt 2>&1
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR" "FETCH_SIZE WRITE_SIZE TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" "TA_BUSY_avr TA_TA_BUSY_sum TA_BUFFER_WAVEFRONTS_sum TA_BUFFER_READ_WAVEFRONTS_sum TA_BUFFER_WRITE_WAVEFRONTS_sum TA_BUFFER_TOTAL_CYCLES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $C --output-format csv -d $O/pmc$i -- python $GRAFT_REPO_ROOT/tools/pmc_rows.py 169 > $O/pmc$i.log 2>&1; echo "pass $i rc=$?"
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections, os
O='gpurun_out/tm15'
for f in glob.glob(O+'/trace/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        print(r['Name'][:110], r['Calls'], r['AverageNs'])
for i in range(1,7):
    fs=glob.glob(O+f'/pmc{i}/**/*counter_collection.csv', recursive=True)
    if not fs: print('pass',i,'no csv'); os.system(f'tail -3 {O}/pmc{i}.log'); continue
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        acc[r['Kernel_Name'][:90]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in acc.items():
        print(i,k,{c:round(sum(x)/len(x)) for c,x in v.items()}, 'n=',len(next(iter(v.values()))))
PY
rm -rf $O/trace/*/*.db 2>/dev/null; du -sh $O
