"""Sampled parity scan on the device (development tool): forward against the double truth and the round trip for every `step`-th length up to 8300 of all ten transform
kinds (C2C, R2C, DCT / DST I-IV), fp32, `batch` transforms per plan, through the C-ABI.  python tools/scan_device_parity.py [step=37] [offset=5] [batch=64]
Prints one JSON line per kind: lengths checked, failures (length, message)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from vkfft_amd import api
from helpers import Runner
import parity
from oracle import oracle as O
step = int(sys.argv[1]) if len(sys.argv) > 1 else 37
off = int(sys.argv[2]) if len(sys.argv) > 2 else 5
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 64
O.build()
run = Runner(api.load(), "gpu")
class Orc:
    truth_c2c = staticmethod(O.truth_c2c); truth_r2r = staticmethod(O.truth_r2r); c2c = staticmethod(O.c2c); r2r = staticmethod(O.r2r); r2c_rows = staticmethod(O.r2c_rows)
for kind in ("c2c", "r2c", "dct1", "dct2", "dct3", "dct4", "dst1", "dst2", "dst3", "dst4"):
    bad, n = [], 0
    for N in range(4 + off, 8300, step):
        n += 1
        try:
            if kind == "c2c": parity.check_c2c(run, Orc, (N,), batch, False, kind="bluestein", use_c_oracle=False)
            elif kind == "r2c": parity.check_r2c(run, Orc, (N,), batch, False)
            else: parity.check_r2r(run, Orc, (N,), batch, False, int(kind[3]), kind.startswith("dst"))
        except AssertionError as e: bad.append((N, str(e)[:100]))
        except Exception as e: bad.append((N, "EXC " + str(e)[:100]))
    print(json.dumps(dict(kind=kind, step=step, offset=off, batch=batch, lengths=n, failures=bad, sources=api.source_hash())), flush=True)
