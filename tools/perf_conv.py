"""Convolution throughput next to the reference on the same box (same call): python tools/perf_conv.py  ->  JSON lines.
Algorithmic bytes of one convolution append = read + write of the data systems + one read of the kernel systems."""
import ctypes as C, json, os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import numpy as np, torch
from vkfft_amd import api
ref = None
p = os.path.join(root, "oracle", "_ref", "libvkfft_ref.so")
if os.path.exists(p):
    ref = C.CDLL(p); ref.ref_convolution.restype = C.c_int
for shape, m, r2c in [((4096, 4096), 1, False), ((2048, 2048), 3, False), ((4096, 4096), 1, True), ((1024, 1024), 3, True)]:
    cf = m; ksys = m * m
    elems = (shape[0] // 2 + 1 if r2c else shape[0]) * shape[1]
    kbytes, dbytes = ksys * elems * 8, cf * elems * 8
    kern = torch.randn(kbytes // 4, device="cuda"); data = torch.randn(dbytes // 4, device="cuda")
    ka = api.App(list(shape), 1, buffer_ptr=kern.data_ptr(), coordinateFeatures=ksys, kernelConvolution=1, r2c=r2c, normalize=True)
    ka.forward()
    ca = api.App(list(shape), 1, buffer_ptr=data.data_ptr(), coordinateFeatures=cf, performConvolution=1, matrixConvolution=m, kernel=kern.data_ptr(), r2c=r2c, normalize=True)
    for _ in range(3): ca.forward()
    torch.cuda.synchronize(); t0 = time.perf_counter(); it = 20
    for _ in range(it): ca.forward()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / it * 1e3
    ka.delete(); ca.delete()
    out = dict(shape=shape, matrix=m, r2c=r2c, ms=round(ms, 4), alg_GBps=round((2 * dbytes + kbytes) / ms / 1e6, 1))
    if ref is not None:
        hk = np.random.default_rng(0).uniform(-1, 1, kbytes // 4).astype(np.float32); hd = np.random.default_rng(1).uniform(-1, 1, dbytes // 4).astype(np.float32)
        rms = C.c_double(0)
        rc = ref.ref_convolution(2, (C.c_uint64 * 4)(*shape), int(r2c), 0, C.c_uint64(cf), C.c_uint64(m), C.c_uint64(1), 0, 0, 0, C.c_uint64(ksys), hk.ctypes.data_as(C.c_void_p),
                                 C.c_uint64(kbytes), hd.ctypes.data_as(C.c_void_p), C.c_uint64(dbytes), C.byref(rms), 20)
        out.update(ref_rc=rc, ref_ms=round(rms.value, 4), ratio_vs_ref=round(rms.value / ms, 2) if rc == 0 else None)
    print(json.dumps(out), flush=True)
