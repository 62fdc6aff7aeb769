"""Native zero padding next to the reference on the same box (the reference's samples 4 / 5 pattern: the upper half of every axis is padding):
forward + inverse pair time with and without the padding flags, this library and the reference's HIP backend.  JSON lines."""
import ctypes as C, json, os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import torch
from vkfft_amd import api
ref = None
p = os.path.join(root, "oracle", "_ref", "libvkfft_ref.so")
if os.path.exists(p):
    ref = C.CDLL(p)
    if hasattr(ref, "ref_bench_pair_zeropad_ms"):
        ref.ref_bench_pair_zeropad_ms.restype = C.c_double
    else:
        ref = None


def ours(shape, r2c, pads):
    elems = (shape[0] // 2 + 1 if r2c else shape[0])
    for s in shape[1:]:
        elems *= s
    buf = torch.empty(elems * 2, dtype=torch.float32, device="cuda").uniform_(-1, 1)
    kw = {}
    if pads:
        kw = dict(performZeropadding=[1] * len(shape) + [0] * (4 - len(shape)), fft_zeropad_left=[(s + 1) // 2 for s in shape] + [0] * (4 - len(shape)),
                  fft_zeropad_right=list(shape) + [0] * (4 - len(shape)))
    app = api.App(list(shape), 1, buffer_ptr=buf.data_ptr(), r2c=r2c, normalize=True, **kw)
    for _ in range(3):
        app.forward(); app.inverse()
    torch.cuda.synchronize()
    it = 20; t0 = time.perf_counter()
    for _ in range(it):
        app.forward(); app.inverse()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / it * 1e3
    app.delete()
    return ms, elems * 8


for shape, r2c in [((256, 256, 256), False), ((512, 512, 512), False), ((256, 256, 256), True), ((512, 512, 512), True), ((4096, 4096), False), ((2048, 2048), True)]:
    ms0, nbytes = ours(shape, r2c, False)
    ms1, _ = ours(shape, r2c, True)
    out = dict(shape=shape, r2c=r2c, pair_ms_plain=round(ms0, 4), pair_ms_zero_padded=round(ms1, 4), gain=round(ms0 / ms1, 2))
    if ref is not None:
        nd = len(shape)
        size = (C.c_uint64 * 4)(*(list(shape) + [1] * (4 - nd)))
        fl = (C.c_uint64 * 4)(*([1] * nd + [0] * (4 - nd))); le = (C.c_uint64 * 4)(*([(s + 1) // 2 for s in shape] + [0] * (4 - nd))); ri = (C.c_uint64 * 4)(*(list(shape) + [0] * (4 - nd)))
        r0 = ref.ref_bench_pair_zeropad_ms(nd, size, C.c_uint64(1), 0, 1 if r2c else 0, C.c_uint64(nbytes), 20, None, None, None, None)
        r1 = ref.ref_bench_pair_zeropad_ms(nd, size, C.c_uint64(1), 0, 1 if r2c else 0, C.c_uint64(nbytes), 20, None, fl, le, ri)
        out.update(ref_pair_ms_plain=round(r0, 4), ref_pair_ms_zero_padded=round(r1, 4), ref_gain=round(r0 / r1, 2) if r1 > 0 else None,
                   ratio_vs_ref_zero_padded=round(r1 / ms1, 2) if r1 > 0 else None)
    print(json.dumps(out), flush=True)
