"""Workload for the rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE): a calibration copy of known size followed by
single launches of the headline kernels on the 1 GiB buffer."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vkfft_amd import api
buf = torch.empty(2 << 27, dtype=torch.float32, device="cuda").uniform_(-1, 1)
dst = torch.empty_like(buf)
for _ in range(2):
    dst.copy_(buf)          # calibration: reads 1 GiB, writes 1 GiB
torch.cuda.synchronize()
for k in (10, 12, 14, 16, 20, 22):
    N = 1 << k
    app = api.App([N], (1 << 27) // N, buffer_ptr=buf.data_ptr(), normalize=True)
    for _ in range(2):
        app.forward(); app.inverse()
    torch.cuda.synchronize()
    app.delete()
