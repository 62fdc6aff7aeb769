"""HBM-level evidence for the fused Four-Step kernel without an HBM counter (rocprofv3 exposes the L2<->fabric requests, which also count
Infinity-Cache hits): the SAME kernel is run with rings of growing size.  While the ring fits the 256 MiB Infinity Cache the intermediate
never reaches HBM and the transform is faster than the two separate passes; once the ring is several times larger than the cache every
ring access becomes an HBM access, the traffic equals that of the separate passes — and so does the time."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vkfft_amd import api

import re, tempfile

def pair_ms(k, env):
    """-> (best pair time in ms, ring size in MiB as the planner reports it under VKFFT_MI355X_PRINT_PLAN, or None for separate passes)"""
    for key in list(os.environ):
        if key.startswith("VKFFT_MI355X_"):
            del os.environ[key]
    os.environ.update(env)
    os.environ["VKFFT_MI355X_PRINT_PLAN"] = "1"
    N = 1 << k
    t = torch.empty(2 << 27, dtype=torch.float32, device="cuda").uniform_(-1, 1)
    with tempfile.TemporaryFile() as tf:           # the library prints the plan on the C stderr
        sys.stderr.flush(); saved = os.dup(2); os.dup2(tf.fileno(), 2)
        try:
            app = api.App([N], (1 << 27) // N, buffer_ptr=t.data_ptr(), normalize=True)
        finally:
            os.dup2(saved, 2); os.close(saved)
        tf.seek(0); m = re.search(r"ring \d+ \(([0-9.]+) MiB\)", tf.read().decode(errors="replace"))
    ring = float(m.group(1)) if m else None
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
    return best, ring

# (queues, lag, ring slots per queue): the queue count is FIXED per size (the planner would otherwise fall back to one queue when the ring
# exceeds its cache budget, which changes the ticket rate as well); the chunk is about 1 MiB or one transform (8 MiB at 2^20, 32 MiB at 2^22),
# ring bytes = queues x slots x chunk.  A longer lag only relaxes dependencies, so what slows the larger rings down is where their bytes live.
SWEEP = {16: ((8, 13, 26), (8, 26, 52), (8, 52, 104), (8, 64, 128)), 18: ((8, 6, 13), (8, 13, 26), (8, 26, 52), (8, 32, 64)),
         20: ((8, 1, 2), (8, 2, 4), (8, 4, 8), (8, 8, 16)), 22: ((2, 1, 2), (2, 2, 4), (2, 4, 8), (2, 8, 16))}
for k in (16, 18, 20, 22):
    rows = [("separate passes (VKFFT_MI355X_FUSED=0)", {"VKFFT_MI355X_FUSED": "0"}), ("fused, default ring", {})]
    for q, lag, ring in SWEEP[k]:
        rows.append((f"fused, {q} queues, lag {lag}, ring {ring} slots/queue",
                     {"VKFFT_MI355X_FUSED_QUEUES": str(q), "VKFFT_MI355X_FUSED_LAG": str(lag), "VKFFT_MI355X_FUSED_RING": str(ring)}))
    for name, env in rows:
        ms, ring_mib = pair_ms(k, env)
        print(json.dumps(dict(log2N=k, config=name, ring_MiB=ring_mib, pair_ms=round(ms, 4), alg_GBps=round(4 * (8 << 27) / (ms * 1e-3) / 1e9, 1))), flush=True)
