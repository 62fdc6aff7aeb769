"""Development tool: the non-temporal hint of the fused Four-Step kernel on both sides (product), on neither, on the loads only, on the stores only.
usage: VKFFT_MI355X_LIB=build/libvkfft_mi355x_dev.so python tools/ab_nt.py 15 22"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vkfft_amd import api
def pair_ms(k, mode):
    os.environ["VKFFT_MI355X_FUSED_MODE"] = str(mode)
    N = 1 << k
    t = torch.empty(2 << 27, dtype=torch.float32, device="cuda").uniform_(-1, 1)
    app = api.App([N], (1 << 27) // N, buffer_ptr=t.data_ptr(), normalize=True)
    for _ in range(2): app.forward(); app.inverse()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(6): app.forward(); app.inverse()
        e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / 6)
    app.delete()
    return best
for k in range(int(sys.argv[1]), int(sys.argv[2]) + 1):
    out = {"log2N": k}
    for name, mode in (("both", 2), ("none", 0), ("loads_only", 66), ("stores_only", 130)):
        ms = pair_ms(k, mode)
        out[name + "_GBps"] = round(4 * (8 << 27) / (ms * 1e-3) / 1e9, 1)
    print(json.dumps(out), flush=True)
