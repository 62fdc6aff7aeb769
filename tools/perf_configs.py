"""Configs 3/4 throughput (development tool): non-pow2 / Rader / Bluestein / fp64 / R2C / DCT / 3D through the C-ABI,
with the reference VkFFT-HIP (oracle/_ref) timed beside it when present.  Algorithmic bytes: SURVEY §8d."""
import sys, os, json, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vkfft_amd import api
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ref = None
p = os.path.join(ROOT, "oracle", "_ref", "libvkfft_ref.so")
if os.path.exists(p) and not os.environ.get("NO_REF"):
    ref = C.CDLL(p); ref.ref_bench_pair_ms.restype = C.c_double

def run(kind, shape, dp, total_log2=26):
    # kind: 0 c2c, 1 r2c, 12 dct2 ...
    n = 1
    for s in shape: n *= s
    es = (16 if dp else 8) if kind in (0,) else (8 if dp else 4)
    B = max(1, (1 << total_log2) // n)
    if kind == 1:
        W = shape[0]; rows = n // W * B
        nbytes = rows * (W // 2 + 1) * (16 if dp else 8)
        alg = rows * (W * (8 if dp else 4) + (W // 2 + 1) * (16 if dp else 8))
    else:
        nbytes = n * B * es; alg = 2 * nbytes
    t = torch.empty(nbytes // 4, dtype=torch.float32, device="cuda").uniform_(-1, 1)
    kw = {}
    if kind == 1: kw["r2c"] = True
    if kind >= 11: kw["dct"] = kind - 10
    app = api.App(list(shape), B, dp=dp, buffer_ptr=t.data_ptr(), normalize=True, **kw)
    for _ in range(2): app.forward(); app.inverse()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    iters = max(2, min(50, int(3e9 // nbytes)))
    best = 1e30
    for rep in range(3):
        e0.record()
        for _ in range(iters): app.forward(); app.inverse()
        e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters)
    up = app.uploads(); app.delete(); del t
    out = dict(kind=kind, shape=list(shape), dp=int(dp), batch=B, uploads=up, pair_ms=round(best, 4), alg_GBps=round(2 * alg / (best * 1e-3) / 1e9, 1))
    if ref is not None:
        size = (C.c_uint64 * 4)(*shape); upr = (C.c_uint64 * 4)()
        ms = ref.ref_bench_pair_ms(C.c_int(len(shape)), size, C.c_uint64(B), C.c_int(int(dp)), C.c_int(kind), C.c_uint64(nbytes), C.c_int(iters), upr)
        out["ref_pair_ms"] = round(ms, 4); out["ref_alg_GBps"] = round(2 * alg / (ms * 1e-3) / 1e9, 1) if ms > 0 else None
    return out

if __name__ == "__main__":
    cases = [(0, (1080,), False), (0, (2160,), False), (0, (3840,), False), (0, (4000,), False), (0, (7680,), False), (0, (2187,), False), (0, (3125,), False),
             (0, (2401,), False), (0, (1331,), False), (0, (2197,), False), (0, (127,), False), (0, (257,), False), (0, (1009,), False), (0, (4093,), False),
             (0, (1024,), True), (0, (4096,), True), (0, (1080,), True), (0, (65536,), True),
             (1, (1024, 1024), False), (12, (1024, 1024), False), (0, (512, 512, 512), False), (1, (4096,), False), (12, (4096,), False), (13, (4096,), False), (11, (1025,), False), (14, (1024,), False), (12, (1024,), False), (13, (1024,), False),
             (0, (3 ** 10,), False), (0, (5 ** 8,), False), (0, (7 ** 7,), False), (0, (11 ** 5,), False), (0, (13 ** 5,), False), (0, (3 ** 13,), False), (0, (15319,), False), (0, (2000083,), False), (0, (21269,), False), (0, (524309,), False), (0, (8191,), False), (1, (5606,), False), (14, (1451,), False)]
    if len(sys.argv) > 1:
        cases = cases[int(sys.argv[1]):int(sys.argv[2])]
    for k, shape, dp in cases:
        try:
            print(json.dumps(run(k, shape, dp)), flush=True)
        except Exception as ex:
            print(json.dumps(dict(kind=k, shape=list(shape), dp=int(dp), error=str(ex))), flush=True)
