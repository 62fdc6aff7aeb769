import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vkfft_amd import api
N, B = int(sys.argv[1]), int(sys.argv[2])
torch.manual_seed(0)
for rep in range(4):
    x = torch.randn(B, N, dtype=torch.complex128, device="cuda")
    buf = x.clone()
    app = api.App([N], B, dp=True, buffer_ptr=buf.data_ptr())
    app.forward(); torch.cuda.synchronize()
    ref = torch.fft.fft(x, dim=1)
    err = (buf - ref).abs()
    badmask = err > 1e-9
    nb = int(badmask.sum().item())
    print("rep", rep, "bad elements", nb, flush=True)
    if nb:
        idx = badmask.nonzero()
        rows = idx[:, 0].unique()
        print("  bad rows", rows[:20].tolist(), "count", len(rows))
        r0 = int(rows[0].item())
        cols = idx[idx[:, 0] == r0][:, 1]
        print("  row", r0, "bad cols", cols[:40].tolist(), "n", len(cols))
        print("  got", buf[r0, cols[:3]].tolist(), "want", ref[r0, cols[:3]].tolist())
    app.delete()
