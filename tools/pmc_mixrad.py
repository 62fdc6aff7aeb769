"""Workload for rocprofv3 PMC passes on the Rader-stage kernel (kernel_mixrad.h): python tools/pmc_mixrad.py N [N ...] — one forward + inverse pair per length (2^25 points)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vkfft_amd import api
buf = torch.empty(1 << 26, dtype=torch.float32, device="cuda").uniform_(-1, 1)
for N in [int(a) for a in sys.argv[1:]]:
    app = api.App([N], (1 << 25) // N, buffer_ptr=buf.data_ptr(), normalize=True)
    for _ in range(2):
        app.forward(); app.inverse()
    torch.cuda.synchronize(); app.delete()
