#!/bin/bash
# real rows after the occupancy changes (rows per workgroup of the OPS form, lighter two-term pre-map): default, generic maps, table maps from 4 / 1 threads per row on
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/tm16
CASES="1:169 12:169 13:169 14:169 1:145 12:145 14:145 1:25 12:25 14:25 1:385 12:385 1:91 12:91 14:125 14:65 14:130 1:55 12:55 14:55 1:364 12:364 1:130 12:130 1:157 12:157 1:61 12:61 1:31 12:31 1:49 12:49 1:34 12:34 11:100 1:301 12:301"
timeout 600 python tools/perf_real_rows.py $CASES > gpurun_out/tm16/tm.jsonl 2> gpurun_out/tm16/tm.err
VKFFT_MI355X_NO_TMAPS=1 NO_REF=1 timeout 600 python tools/perf_real_rows.py $CASES > gpurun_out/tm16/generic.jsonl 2>> gpurun_out/tm16/tm.err
VKFFT_MI355X_TMAPS_MIN_TPF=4 NO_REF=1 timeout 600 python tools/perf_real_rows.py $CASES > gpurun_out/tm16/tpf4.jsonl 2>> gpurun_out/tm16/tm.err
VKFFT_MI355X_TMAPS_MIN_TPF=2 NO_REF=1 timeout 600 python tools/perf_real_rows.py $CASES > gpurun_out/tm16/tpf2.jsonl 2>> gpurun_out/tm16/tm.err
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "real or r2c or dct or dst or two_rows or r2r" > gpurun_out/tm16/pytest.log 2>&1
tail -3 gpurun_out/tm16/pytest.log
python - <<'PY'
import json
L=lambda f:[json.loads(l) for l in open('gpurun_out/tm16/'+f)]
a,b,c,d=L('tm.jsonl'),L('generic.jsonl'),L('tpf4.jsonl'),L('tpf2.jsonl')
for x,y,z,w in zip(a,b,c,d):
    print(x['kind'],x['shape'],'tm',x['pair_ms'],'generic',y['pair_ms'],'tpf4',z['pair_ms'],'tpf2',w['pair_ms'],'ref',x.get('ref_pair_ms'),'ratio_ref/tm',round(x.get('ref_pair_ms',0)/x['pair_ms'],2))
PY
