"""Workload for rocprofv3 PMC passes on the real-transform row kernels: DCT-II 4096 staged / direct, R2C 4096, C2C 4096."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vkfft_amd import api
buf = torch.empty(1 << 26, dtype=torch.float32, device="cuda").uniform_(-1, 1)
def go(**kw):
    app = api.App([4096], kw.pop("batch"), buffer_ptr=buf.data_ptr(), normalize=True, **kw)
    for _ in range(2):
        app.forward(); app.inverse()
    torch.cuda.synchronize(); app.delete()
os.environ["VKFFT_MI355X_OPSTG"] = "1"; go(batch=16384, dct=2)
os.environ["VKFFT_MI355X_OPSTG"] = "0"; go(batch=16384, dct=2)
go(batch=8192)
go(batch=16384 * 4096 // 4098, r2c=True)
