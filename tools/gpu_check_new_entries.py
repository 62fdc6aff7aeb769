"""One-off device check of the op-FFT instances with L = 8192 (fp32) / 4096 (fp64) against torch's double FFT (no oracle: seconds)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vkfft_amd import api
ok = True
for N, dp in ((16384, False), (8192, True)):
    rt = torch.float64 if dp else torch.float32
    B = 300
    x = torch.empty(B, N + 2, dtype=rt, device="cuda").uniform_(-1, 1); x[:, N:] = 0
    buf = x.clone()
    app = api.App([N], B, dp=dp, r2c=True, buffer_ptr=buf.data_ptr()); app.forward(); torch.cuda.synchronize()
    X = torch.view_as_complex(buf.view(B, N // 2 + 1, 2)); ref = torch.fft.rfft(x[:, :N].double(), dim=1)
    e = (torch.linalg.norm(X.to(torch.complex128) - ref) / torch.linalg.norm(ref)).item()
    app.inverse(); torch.cuda.synchronize()
    e2 = (torch.linalg.norm(buf[:, :N].double() - N * x[:, :N].double()) / torch.linalg.norm(N * x[:, :N].double())).item()
    print("r2c", N, dp, e, e2); ok &= e < (1e-14 if dp else 2e-6) and e2 < (2e-14 if dp else 4e-6); app.delete()
    for t in (2, 4):
        y = torch.empty(B, N, dtype=rt, device="cuda").uniform_(-1, 1); b2 = y.clone()
        app = api.App([N], B, dp=dp, dct=t, buffer_ptr=b2.data_ptr()); app.forward(); app.inverse(); torch.cuda.synchronize()
        e3 = (torch.linalg.norm(b2.double() - 2 * N * y.double()) / torch.linalg.norm(2 * N * y.double())).item()
        print("dct", t, N, dp, e3); ok &= e3 < (3e-14 if dp else 6e-6); app.delete()
print("ALL OK" if ok else "FAILED")
