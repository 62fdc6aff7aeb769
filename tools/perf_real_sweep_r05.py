"""Round 5: the three real-row sweeps of the review — R2C and DCT-II rows of 4 ... 400 reals (step 3), DCT-IV rows of 5 ... 400 (step 5) — default plans, the
reference timed in the same process on EVERY length, and the kernel the plan runs on.
python tools/perf_real_sweep_r05.py <r2c|dct2|dct4> [step multiplier]"""
import sys, os, json, re
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vkfft_amd import api
from perf_configs import run
which = sys.argv[1]
mult = int(sys.argv[2]) if len(sys.argv) > 2 else 1
kind = {"r2c": 1, "dct2": 12, "dct4": 14}[which]
lengths = list(range(5, 401, 5 * mult)) if which == "dct4" else list(range(4, 401, 3 * mult))
probe = torch.zeros(4096, dtype=torch.float32, device="cuda")
for n in lengths:
    kw = {"r2c": True} if kind == 1 else {"dct": kind - 10}
    app = api.App([n], (1 << 25) // n, buffer_ptr=probe.data_ptr(), **kw)  # (plan only: which kernel)
    _, name = app.launch_info(); app.delete()
    r = run(kind, (n,), False, total_log2=25)
    print(json.dumps({"kind": kind, "N": n, "ms": r["pair_ms"], "ref_ms": r.get("ref_pair_ms"), "kernel": re.sub(r"<.*", "", name)}), flush=True)
