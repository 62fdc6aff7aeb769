"""Bluestein rows of large primes: the planner's default against a 7-smooth padded length forced through fixMaxRadixBluestein (development tool).
python tools/perf_blue_smooth.py N [N ...]"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vkfft_amd import api
def run(n, **kw):
    B = max(1, (1 << 25) // n)
    t = torch.empty(2 * n * B, dtype=torch.float32, device="cuda").uniform_(-1, 1)
    app = api.App([n], B, buffer_ptr=t.data_ptr(), normalize=True, **kw)
    for _ in range(2): app.forward(); app.inverse()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e30
    for rep in range(3):
        e0.record()
        for _ in range(5): app.forward(); app.inverse()
        e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / 5)
    info = app.launch_info(False); up = app.uploads(); app.delete()
    return dict(N=n, batch=B, kw=kw, launches=info[0], kernel=info[1], uploads=up, pair_ms=round(best, 4), alg_GBps=round(4 * 8 * n * B / (best * 1e-3) / 1e9, 1))
for n in [int(a) for a in sys.argv[1:]]:
    for kw in ({}, {"fixMaxRadixBluestein": 7}, {"fixMaxRadixBluestein": 13}):
        try: print(json.dumps(run(n, **kw)), flush=True)
        except Exception as ex: print(json.dumps(dict(N=n, kw=kw, error=str(ex))), flush=True)
