"""Workload for rocprofv3 PMC passes on short rows (256 MiB: the cache-resident regime of the reference's all-lengths benchmark)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vkfft_amd import api
buf = torch.empty(1 << 26, dtype=torch.float32, device="cuda").uniform_(-1, 1)
for N in (100, 128, 120, 75, 175, 64, 45, 169):
    app = api.App([N], (1 << 25) // N, buffer_ptr=buf.data_ptr(), normalize=True)
    for _ in range(3):
        app.forward(); app.inverse()
    torch.cuda.synchronize(); app.delete()
