"""Workload for rocprofv3 PMC passes on the mixed-radix row kernels (config-3 single-pass lengths) with C2C 2048 beside them."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vkfft_amd import api
buf = torch.empty(1 << 27, dtype=torch.float32, device="cuda").uniform_(-1, 1)
for N in (4000, 2401, 2197, 1331, 2187, 3125, 1080, 3840, 2048):
    app = api.App([N], (1 << 26) // N, buffer_ptr=buf.data_ptr(), normalize=True)
    for _ in range(2):
        app.forward(); app.inverse()
    torch.cuda.synchronize(); app.delete()
