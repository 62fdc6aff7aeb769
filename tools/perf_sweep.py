"""Throughput sweep through the C-ABI (development tool): batched 1D C2C, 1 GiB buffer (sample-0 protocol)."""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vkfft_amd import api

def bench(k, dp=False, total_log2=27, iters=None, **kw):
    N = 1 << k
    B = max(1, (1 << total_log2) // N)
    es = 16 if dp else 8
    nbytes = N * B * es
    t = torch.empty(nbytes // 4, dtype=torch.float32, device="cuda").uniform_(-1, 1)
    app = api.App([N], B, dp=dp, buffer_ptr=t.data_ptr(), **kw)
    if iters is None:
        iters = max(1, min(1000, (3 * 4096 * 1024 * 1024) // nbytes))
    for _ in range(2):
        app.forward(); app.inverse()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e30
    for rep in range(3):
        e0.record()
        for _ in range(iters):
            app.forward(); app.inverse()
        e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters)
        t.uniform_(-1, 1)
    up = app.uploads()
    app.delete()
    del t
    return dict(log2N=k, N=N, batch=B, dp=int(dp), uploads=up[0], pair_ms=round(best, 5),
                alg_GBps=round(2 * 2 * nbytes / (best * 1e-3) / 1e9, 1), GFLOPs=round(2 * 5 * N * k * B / (best * 1e-3) / 1e9, 1))

if __name__ == "__main__":
    kmin = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    kmax = int(sys.argv[2]) if len(sys.argv) > 2 else 22
    dp = bool(int(sys.argv[3])) if len(sys.argv) > 3 else False
    for k in range(kmin, kmax + 1):
        print(json.dumps(bench(k, dp)), flush=True)
