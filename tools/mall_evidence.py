"""HBM-level evidence for the fused Four-Step kernel without an HBM counter (rocprofv3 exposes the L2<->fabric requests, which also count
Infinity-Cache hits): the SAME kernel is run with rings of growing size.  While the ring fits the 256 MiB Infinity Cache the intermediate
never reaches HBM and the transform is faster than the two separate passes; once the ring is several times larger than the cache every
ring access becomes an HBM access, the traffic equals that of the separate passes — and so does the time."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vkfft_amd import api

def pair_ms(k, env):
    for key in list(os.environ):
        if key.startswith("VKFFT_MI355X_"):
            del os.environ[key]
    os.environ.update(env)
    N = 1 << k
    t = torch.empty(2 << 27, dtype=torch.float32, device="cuda").uniform_(-1, 1)
    app = api.App([N], (1 << 27) // N, buffer_ptr=t.data_ptr(), normalize=True)
    for _ in range(2):
        app.forward(); app.inverse()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(6):
            app.forward(); app.inverse()
        e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / 6)
    app.delete()
    return best

for k in (16, 18):
    rows = [("separate passes (VKFFT_MI355X_FUSED=0)", {"VKFFT_MI355X_FUSED": "0"}), ("fused, default ring", {})]
    for lag, ring in ((13, 26), (26, 52), (52, 104), (100, 128)):
        rows.append((f"fused, lag {lag} ring {ring} slots/queue", {"VKFFT_MI355X_FUSED_LAG": str(lag), "VKFFT_MI355X_FUSED_RING": str(ring)}))
    for name, env in rows:
        ms = pair_ms(k, env)
        chunk_mib = max(1.0, (8 << k) / 2.0 ** 20)  # the planner's chunk: about 1 MiB, at least one transform
        ring_mib = None if "ring" not in name or "default" in name else 8 * int(env["VKFFT_MI355X_FUSED_RING"]) * chunk_mib
        print(json.dumps(dict(log2N=k, config=name, ring_MiB=ring_mib, pair_ms=round(ms, 4), alg_GBps=round(4 * (8 << 27) / (ms * 1e-3) / 1e9, 1))), flush=True)
