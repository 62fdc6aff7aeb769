"""One-off device check (seconds, torch's double FFT as truth): generic-kernel paths after the natural-index change, and the
multi-pass real transforms."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vkfft_amd import api
ok = True
def c2c(N, B):
    global ok
    x = torch.empty(B, 2 * N, dtype=torch.float32, device="cuda").uniform_(-1, 1); b = x.clone()
    app = api.App([N], B, buffer_ptr=b.data_ptr()); app.forward(); torch.cuda.synchronize()
    ref = torch.fft.fft(torch.view_as_complex(x.view(B, N, 2)).to(torch.complex128), dim=1)
    e = (torch.linalg.norm(torch.view_as_complex(b.view(B, N, 2)).to(torch.complex128) - ref) / torch.linalg.norm(ref)).item()
    app.inverse(); torch.cuda.synchronize(); e2 = (torch.linalg.norm(b.double() - N * x.double()) / torch.linalg.norm(N * x.double())).item()
    print("c2c", N, e, e2); ok &= e < 3e-6 and e2 < 6e-6; app.delete()
def r2c(N, B):
    global ok
    x = torch.empty(B, 2 * (N // 2 + 1), dtype=torch.float32, device="cuda").uniform_(-1, 1); x[:, N:] = 0; b = x.clone()
    app = api.App([N], B, r2c=True, buffer_ptr=b.data_ptr()); app.forward(); torch.cuda.synchronize()
    ref = torch.fft.rfft(x[:, :N].double(), dim=1)
    e = (torch.linalg.norm(torch.view_as_complex(b.view(B, N // 2 + 1, 2)).to(torch.complex128) - ref) / torch.linalg.norm(ref)).item()
    app.inverse(); torch.cuda.synchronize(); e2 = (torch.linalg.norm(b[:, :N].double() - N * x[:, :N].double()) / torch.linalg.norm(N * x[:, :N].double())).item()
    print("r2c", N, e, e2); ok &= e < 3e-6 and e2 < 6e-6; app.delete()
def dct(N, B, t, scale):
    global ok
    x = torch.empty(B, N, dtype=torch.float32, device="cuda").uniform_(-1, 1); b = x.clone()
    app = api.App([N], B, dct=t, buffer_ptr=b.data_ptr()); app.forward(); app.inverse(); torch.cuda.synchronize()
    e = (torch.linalg.norm(b.double() - scale * x.double()) / torch.linalg.norm(scale * x.double())).item()
    print("dct", t, N, e); ok &= e < 8e-6; app.delete()
c2c(30030, 7); c2c(2 * 3 * 5 * 7 * 11, 33); c2c(1078, 50)      # generic single / multi-pass paths
r2c(1001, 9); r2c(11583, 5); r2c(18375, 3)                     # odd rows: single pass, multi-pass
dct(32768, 4, 2, 2 * 32768); dct(16385, 3, 1, 2 * 16384); dct(32768, 2, 4, 2 * 32768); dct(1200000, 1, 2, 2 * 1200000)
print("ALL OK" if ok else "FAILED")
