"""Quick GPU sanity sweep (development tool; the asserted versions live in tests/)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vkfft_amd import api

def rel_l2(a, b):
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))

def run_c2c(shape, batch, dp, inverse=False, **kw):
    rng = np.random.default_rng(1)
    n = int(np.prod(shape)) * batch
    ct = np.complex128 if dp else np.complex64
    x = (rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n)).astype(ct)
    t = torch.from_numpy(x.view(np.float64 if dp else np.float32)).cuda()
    app = api.App(list(shape), batch, dp=dp, buffer_ptr=t.data_ptr(), **kw)
    app.append(inverse)
    torch.cuda.synchronize()
    y = t.cpu().numpy().view(ct).reshape([batch] + list(shape)[::-1])
    xs = x.astype(np.complex128).reshape([batch] + list(shape)[::-1])
    axes = tuple(range(1, 1 + len(shape)))
    ref = np.fft.ifftn(xs, axes=axes) * np.prod(shape) if inverse else np.fft.fftn(xs, axes=axes)
    up = app.uploads(inverse)
    app.delete()
    return rel_l2(y.astype(np.complex128), ref), up

if __name__ == "__main__":
    cases = []
    for N in [2, 3, 4, 5, 7, 8, 11, 13, 16, 32, 64, 100, 128, 256, 512, 1024, 2048, 4096, 8192, 1080, 243, 343, 121 * 4, 13 * 64, 6561, 3125,
              17, 31, 34 * 16, 127, 1009, 2 ** 14, 2 ** 15, 2 ** 16, 3 ** 10, 2 ** 18, 2 ** 20, 2 ** 22, 5 ** 9]:
        b = max(1, min(64, (1 << 16) // N)) if N < 2 ** 16 else 3
        cases.append(((N,), b))
    cases += [((64, 64), 3), ((512, 512), 2), ((128, 64, 32), 2), ((100, 60), 2), ((32, 32, 32), 1)]
    bad = 0
    for dp in (False, True):
        for shape, b in cases:
            for inv in (False, True):
                try:
                    e, up = run_c2c(shape, b, dp, inv)
                    tol = 2e-15 * 5 if dp else 1e-6
                    flag = "" if e < tol else "  <<<<<< BAD"
                    bad += e >= tol
                    print(f"c2c {'fp64' if dp else 'fp32'} {shape} b={b} inv={int(inv)} uploads={up} relL2={e:.3e}{flag}", flush=True)
                except Exception as ex:
                    bad += 1
                    print(f"c2c {'fp64' if dp else 'fp32'} {shape} b={b} inv={int(inv)} EXC {ex}", flush=True)
    print("BAD", bad)
