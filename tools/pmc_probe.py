"""Workload for the rocprofv3 PMC passes (one counter set per pass): a calibration copy of known size, then a few launches of the
headline kernels on the 1 GiB buffer — single-pass row kernels, the fused Four-Step kernel, and the same sizes with the fusion off."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vkfft_amd import api
if os.environ.get("VKFFT_PMC_HASH_FILE"):
    open(os.environ["VKFFT_PMC_HASH_FILE"], "w").write(api.source_hash() + "\n")
buf = torch.empty(2 << 27, dtype=torch.float32, device="cuda").uniform_(-1, 1)
dst = torch.empty_like(buf)
for _ in range(2):
    dst.copy_(buf)          # calibration: reads 1 GiB, writes 1 GiB
torch.cuda.synchronize()
del dst
for fused in ("1", "0"):
    os.environ["VKFFT_MI355X_FUSED"] = fused
    for k in ((10, 12, 14, 15, 16, 18, 20, 22) if fused == "1" else (16, 20)):
        N = 1 << k
        app = api.App([N], (1 << 27) // N, buffer_ptr=buf.data_ptr(), normalize=True)
        for _ in range(2):
            app.forward(); app.inverse()
        torch.cuda.synchronize()
        app.delete()
