"""Workload for rocprofv3 passes on real rows of one length next to the complex rows of the same length (python tools/pmc_rows.py N)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vkfft_amd import api
N = int(sys.argv[1]) if len(sys.argv) > 1 else 169
B = (1 << 25) // N
buf = torch.empty(1 << 26, dtype=torch.float32, device="cuda").uniform_(-1, 1)
def go(**kw):
    app = api.App([N], B, buffer_ptr=buf.data_ptr(), normalize=True, **kw)
    for _ in range(3):
        app.forward(); app.inverse()
    torch.cuda.synchronize(); app.delete()
go()
go(r2c=True)
go(dct=2)
go(dct=4)
