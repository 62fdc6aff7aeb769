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

if __name__ == "__main__" and not (len(sys.argv) > 1 and sys.argv[1] == "real"):
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

def run_r2c(shape, batch, dp):
    """in-place padded layout: forward then inverse; returns (err_fwd, err_roundtrip)"""
    rng = np.random.default_rng(2)
    W = shape[0]; Wc = W // 2 + 1
    rest = int(np.prod(shape[1:])) if len(shape) > 1 else 1
    rt = np.float64 if dp else np.float32
    x = rng.uniform(-1, 1, (batch * rest, W)).astype(rt)
    buf = np.zeros((batch * rest, 2 * Wc), dtype=rt); buf[:, :W] = x
    t = torch.from_numpy(buf).cuda()
    app = api.App(list(shape), batch, dp=dp, r2c=True, buffer_ptr=t.data_ptr())
    app.forward(); torch.cuda.synchronize()
    ct = np.complex128 if dp else np.complex64
    y = t.cpu().numpy().view(ct).reshape([batch] + list(shape[1:])[::-1] + [Wc])
    xs = x.astype(np.float64).reshape([batch] + list(shape)[::-1])
    axes = tuple(range(1, 1 + len(shape)))
    ref = np.fft.rfftn(xs, axes=axes)
    e1 = rel_l2(y.astype(np.complex128), ref)
    app.inverse(); torch.cuda.synchronize()
    back = t.cpu().numpy()[:, :W].astype(np.float64)
    e2 = rel_l2(back, x.astype(np.float64) * np.prod(shape))
    app.delete()
    return e1, e2

def run_r2r(shape, batch, dp, type, dst):
    import scipy.fft as sf
    rng = np.random.default_rng(3)
    rt = np.float64 if dp else np.float32
    n = int(np.prod(shape)) * batch
    x = rng.uniform(-1, 1, n).astype(rt)
    t = torch.from_numpy(x.copy()).cuda()
    kw = dict(dst=type) if dst else dict(dct=type)
    app = api.App(list(shape), batch, dp=dp, buffer_ptr=t.data_ptr(), **kw)
    app.forward(); torch.cuda.synchronize()
    y = t.cpu().numpy().astype(np.float64)
    xs = x.astype(np.float64).reshape([batch] + list(shape)[::-1])
    axes = tuple(range(1, 1 + len(shape)))
    f = sf.dstn if dst else sf.dctn
    ref = f(xs, type=type, axes=axes).reshape(-1)
    e1 = rel_l2(y, ref)
    app.inverse(); torch.cuda.synchronize()
    back = t.cpu().numpy().astype(np.float64)
    norm = 1.0
    for s in shape:
        norm *= (2.0 * (s - 1) if not dst else 2.0 * (s + 1)) if type == 1 else 2.0 * s
    e2 = rel_l2(back, x.astype(np.float64) * norm)
    app.delete()
    return e1, e2

def main_real():
    bad = 0
    for dp in (False, True):
        tol = 1e-14 if dp else 2e-6
        for shape, b in [((16,), 4), ((15,), 4), ((256,), 8), ((1000,), 3), ((243,), 3), ((4096,), 2), ((8192,), 2), ((1024, 1024), 1), ((64, 32), 2), ((30, 20, 10), 2), ((33, 8), 2)]:
            try:
                e1, e2 = run_r2c(shape, b, dp)
                flag = "" if max(e1, e2) < tol else "  <<<<<< BAD"; bad += max(e1, e2) >= tol
                print(f"r2c {'fp64' if dp else 'fp32'} {shape} b={b} fwd={e1:.3e} roundtrip={e2:.3e}{flag}", flush=True)
            except Exception as ex:
                bad += 1; print(f"r2c {'fp64' if dp else 'fp32'} {shape} b={b} EXC {ex}", flush=True)
        for dst in (False, True):
            for type in (1, 2, 3, 4):
                for shape, b in [((8,), 3), ((9,), 3), ((64,), 4), ((100,), 3), ((81,), 2), ((1024,), 2), ((1024, 1024), 1), ((32, 24), 2), ((12, 10, 6), 2)]:
                    try:
                        e1, e2 = run_r2r(shape, b, dp, type, dst)
                        flag = "" if max(e1, e2) < tol else "  <<<<<< BAD"; bad += max(e1, e2) >= tol
                        print(f"{'dst' if dst else 'dct'}{type} {'fp64' if dp else 'fp32'} {shape} b={b} fwd={e1:.3e} roundtrip={e2:.3e}{flag}", flush=True)
                    except Exception as ex:
                        bad += 1; print(f"{'dst' if dst else 'dct'}{type} {'fp64' if dp else 'fp32'} {shape} b={b} EXC {ex}", flush=True)
    print("BAD_REAL", bad)

if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "real":
    main_real()
