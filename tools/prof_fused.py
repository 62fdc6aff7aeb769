"""Development tool: per-phase cycle profile of the fused Four-Step kernel (VKFFT_MI355X_FUSED_MODE=6 + VKFFT_MI355X_FUSED_PROFILE=1)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vkfft_amd import api
os.environ["VKFFT_MI355X_FUSED_MODE"] = "6"
os.environ["VKFFT_MI355X_FUSED_PROFILE"] = "1"
for kv in sys.argv[3:]:
    a, b = kv.split("=")
    os.environ["VKFFT_MI355X_" + a] = b
for k in range(int(sys.argv[1]), int(sys.argv[2]) + 1):
    N = 1 << k; B = (1 << 27) // N
    t = torch.empty(2 << 27, dtype=torch.float32, device="cuda").uniform_(-1, 1)
    app = api.App([N], B, buffer_ptr=t.data_ptr(), normalize=True)
    print("log2N", k, file=sys.stderr, flush=True)
    for _ in range(3):
        app.forward(); app.inverse()
    torch.cuda.synchronize()
    app.delete()
