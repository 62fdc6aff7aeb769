#!/bin/bash
# real rows with the table-driven maps against the generic maps of the same plans (VKFFT_MI355X_NO_TMAPS=1), against the half-length forms of the even lengths
# (VKFFT_MI355X_EVEN_FULL=0) and the reference in the same process
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/tm
CASES="1:169 12:169 13:169 14:169 1:145 12:145 14:145 1:25 12:25 1:385 12:385 1:100 12:100 13:100 1:91 12:91 14:125 14:65 14:130 1:55 12:55 1:196 12:196 1:364 12:364 1:130 12:130 1:250 12:250 1:157 12:157 1:31 12:31 11:100 11:169"
timeout 600 python tools/perf_real_rows.py $CASES > gpurun_out/tm/tm.jsonl 2> gpurun_out/tm/tm.err
VKFFT_MI355X_NO_TMAPS=1 NO_REF=1 timeout 600 python tools/perf_real_rows.py $CASES > gpurun_out/tm/generic.jsonl 2>> gpurun_out/tm/tm.err
VKFFT_MI355X_EVEN_FULL=0 NO_REF=1 timeout 600 python tools/perf_real_rows.py $CASES > gpurun_out/tm/evenhalf.jsonl 2>> gpurun_out/tm/tm.err
VKFFT_MI355X_EVEN_FULL=2 NO_REF=1 timeout 600 python tools/perf_real_rows.py $CASES > gpurun_out/tm/evenfull2.jsonl 2>> gpurun_out/tm/tm.err
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "real or r2c or dct or dst or two_rows or r2r" > gpurun_out/tm/pytest.log 2>&1
tail -3 gpurun_out/tm/pytest.log
python - <<'PY'
import json
L=lambda f:[json.loads(l) for l in open('gpurun_out/tm/'+f)]
a,b,c,d=L('tm.jsonl'),L('generic.jsonl'),L('evenhalf.jsonl'),L('evenfull2.jsonl')
for x,y,z,w in zip(a,b,c,d):
    print(x['kind'],x['shape'],'tm',x['pair_ms'],'generic',y['pair_ms'],'evenhalf',z['pair_ms'],'evenfull2',w['pair_ms'],'ref',x.get('ref_pair_ms'),'ratio_ref/tm',round(x.get('ref_pair_ms',0)/x['pair_ms'],2))
PY
