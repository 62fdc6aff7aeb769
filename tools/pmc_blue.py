"""Workload for rocprofv3 PMC passes on the fused Bluestein kernels (N = 127, 1009, 4093) with C2C 256 / 8192 beside them."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vkfft_amd import api
buf = torch.empty(1 << 27, dtype=torch.float32, device="cuda").uniform_(-1, 1)
for N in (127, 1009, 4093, 256, 8192):
    app = api.App([N], (1 << 26) // N, buffer_ptr=buf.data_ptr(), normalize=True)
    for _ in range(2):
        app.forward(); app.inverse()
    torch.cuda.synchronize(); app.delete()
