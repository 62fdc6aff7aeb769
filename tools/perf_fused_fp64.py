import sys, os, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from vkfft_amd import api
for k in (14, 15, 16, 17, 18, 19, 20, 21):
    for fused in ("0", "1"):
        os.environ["VKFFT_MI355X_FUSED"] = fused
        N = 1 << k; B = (1 << 26) // N
        t = torch.empty(4 * N * B // 2, dtype=torch.float64, device="cuda").uniform_(-1, 1)
        ref = t.clone()
        app = api.App([N], B, dp=True, buffer_ptr=t.data_ptr(), normalize=True)
        app.forward(); app.inverse(); torch.cuda.synchronize()
        err = float((t - ref).abs().max())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(3):
            e0.record()
            for _ in range(6): app.forward(); app.inverse()
            e1.record(); e1.synchronize(); best = min(best, e0.elapsed_time(e1) / 6)
        app.delete()
        print(json.dumps(dict(k=k, fused=fused, pair_ms=round(best, 4), alg_GBps=round(4 * (16 << 26) / (best * 1e-3) / 1e9, 1), err=err)), flush=True)
