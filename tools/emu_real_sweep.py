"""Development tool: every real family (R2C / C2R, DCT / DST I-IV) on every length of a range through the CPU-emulated build of the sources, against the oracle
(python tools/emu_real_sweep.py <first> <last+1> [dp]; fp32: 3 rows, fp64: 1 and 4 rows).  Round 5, final sources: 2 ... 419, both precisions, 11 286 checks, and every 13th length 421 ... 4199 in fp32 (2 619 checks): none failed.  Optional 4th argument: step."""
import sys, os, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from vkfft_amd import api
import parity
from helpers import Runner
import importlib

lib = api.load_test_double(os.path.join(ROOT, 'tests', 'hostemu', '_build', 'libvkfft_hostemu.so'))
from oracle import oracle as O
O.build()
run = Runner(lib, 'emu')
lo, hi = int(sys.argv[1]), int(sys.argv[2]); DP = len(sys.argv) > 3 and sys.argv[3] == "dp"; BATCHES = (1, 4) if DP else (3,)
bad = 0
STEP = int(sys.argv[4]) if len(sys.argv) > 4 else 1
for N in range(lo, hi, STEP):
    for batch in BATCHES:
        try:
            parity.check_r2c(run, O, (N,), batch, DP)
        except Exception as e:
            bad += 1; print('FAIL r2c', N, str(e)[:100], flush=True)
        for type, dst in [(1, False), (2, False), (3, False), (4, False), (1, True), (2, True), (3, True), (4, True)]:
            if type == 1 and not dst and N < 2: continue
            try:
                parity.check_r2r(run, O, (N,), batch, DP, type, dst)
            except Exception as e:
                bad += 1; print('FAIL r2r', N, type, dst, str(e)[:100], flush=True)
print('range', lo, hi, 'bad', bad, flush=True)
