"""Development tool: A/B of the fused Four-Step kernel's knobs through the C-ABI (1 GiB batched 1D C2C fp32, FFT+iFFT pairs).
usage: python tools/tune_fused.py <kmin> <kmax> [quick]"""
import sys, os, json, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vkfft_amd import api

TOTAL = 27

def run(k, env, iters=6, check=False):
    for key in list(os.environ):
        if key.startswith("VKFFT_MI355X_"):
            del os.environ[key]
    os.environ.update({k2: str(v) for k2, v in env.items()})
    N = 1 << k; B = (1 << TOTAL) // N
    t = torch.empty(2 << TOTAL, dtype=torch.float32, device="cuda").uniform_(-1, 1)
    ref = t.clone() if check else None
    app = api.App([N], B, buffer_ptr=t.data_ptr(), normalize=True)
    app.forward(); app.inverse()
    torch.cuda.synchronize()
    err = None
    if check:
        err = float((t - ref).abs().max().item())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e30
    for rep in range(2):
        e0.record()
        for _ in range(iters):
            app.forward(); app.inverse()
        e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters)
    app.delete()
    return dict(k=k, **{a.replace("VKFFT_MI355X_", ""): b for a, b in env.items()}, pair_ms=round(best, 4),
                alg_GBps=round(4 * (8 << TOTAL) / (best * 1e-3) / 1e9, 1), roundtrip_maxerr=err)

if __name__ == "__main__":
    kmin, kmax = int(sys.argv[1]), int(sys.argv[2])
    specs = [{"FUSED": 0}, {}]
    for arg in sys.argv[3:]:
        specs.append(dict(kv.split("=") for kv in arg.split(",")))
    for k in range(kmin, kmax + 1):
        for sp in specs:
            env = {"VKFFT_MI355X_" + a.replace("KK", str(k)): b for a, b in sp.items()}
            print(json.dumps(run(k, env, check=True)), flush=True)
