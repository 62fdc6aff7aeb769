"""Shared helpers of the parity tests: run a transform through the C-ABI (real library on the GPU, or the
CPU test double) and fetch the result as numpy."""
import numpy as np
from vkfft_amd import api

# tolerances (the reference asserts none, its published bands are in BASELINE.md)
TOL = {
    ("c2c", False): 1.0e-6, ("c2c", True): 2.0e-15,
    ("bluestein", False): 3.0e-6, ("bluestein", True): 6.0e-15,
    ("real", False): 2.0e-6, ("real", True): 6.0e-15,
}


# Per-element bounds — the reference's own metric (sample_11_precision_VkFFT_single.cpp:289-331: max / average |delta| and |delta|/|ref| per element
# against the double-precision truth; it prints them and asserts nothing).  Stated in ULPs of the output's RMS, ulp = eps(dtype) * rms(ref):
#   max |delta|            <= MAX_ULP ulp   (one wrong element in 2^27 moves the relative L2 by 1e-4 of nothing; it moves this by 10^6 ulp)
#   mean |delta| / |ref|   <= AVG_EPS_ULP * eps   (complex outputs only: for real outputs E[1/|ref|] diverges; elements below 2^-10 of the RMS are left out)
MAX_ULP = {("c2c", False): 64, ("c2c", True): 72, ("bluestein", False): 192, ("bluestein", True): 216, ("real", False): 128, ("real", True): 216}
AVG_EPS_ULP = {("c2c", False): 24, ("c2c", True): 27, ("bluestein", False): 72, ("bluestein", True): 81, ("real", False): 48, ("real", True): 81}


def element_errors(y, ref):
    """(max |delta| and mean |delta|/|ref|) in ulps of the output's RMS resp. in eps — see MAX_ULP"""
    ref = np.asarray(ref).reshape(-1)
    y = np.asarray(y).reshape(-1)
    dp = y.dtype in (np.float64, np.complex128)
    eps = np.finfo(np.float64 if dp else np.float32).eps
    d = np.abs(y.astype(np.complex128 if np.iscomplexobj(ref) else np.float64) - ref.astype(np.complex128 if np.iscomplexobj(ref) else np.float64))
    mag = np.abs(ref.astype(np.complex128 if np.iscomplexobj(ref) else np.float64))
    rms = max(float(np.sqrt(np.mean(mag ** 2))), 1e-300)
    nz = mag > rms * 2.0 ** -10  # (elements 60 dB below the RMS are left out of the mean: with a reference value next to zero the quotient is unbounded — seen once in 1 700 fuzz cases, a 13 x 2 R2C plane of 14 bins: mean 48.8 eps from ONE bin)
    return float(d.max() / (eps * rms)), float(np.mean(d[nz] / mag[nz]) / eps) if nz.any() else 0.0


def assert_elementwise(y, ref, kind, dp, what=""):
    mx, avg = element_errors(y, ref)
    assert mx <= MAX_ULP[(kind, dp)], f"{what}: max |delta| = {mx:.1f} ulp of the output RMS > {MAX_ULP[(kind, dp)]}"
    if np.iscomplexobj(ref):
        assert avg <= AVG_EPS_ULP[(kind, dp)], f"{what}: mean |delta|/|ref| = {avg:.1f} eps > {AVG_EPS_ULP[(kind, dp)]}"
    return mx, avg


def rel_l2(a, b):
    a = np.asarray(a).reshape(-1).astype(np.clongdouble if np.iscomplexobj(b) or np.iscomplexobj(a) else np.longdouble)
    b = np.asarray(b).reshape(-1)
    return float(np.linalg.norm((a - b).astype(np.complex128)) / max(float(np.linalg.norm(b.astype(np.complex128))), 1e-300))


class Runner:
    """device='emu' : host arrays + CPU test double;  device='gpu': torch CUDA tensors + the real library."""

    def __init__(self, lib, device):
        self.lib, self.device = lib, device

    def _alloc(self, host):
        if self.device == "gpu":
            import torch
            t = torch.from_numpy(np.ascontiguousarray(host).view(np.uint8).reshape(-1).copy()).cuda()
            return t, t.data_ptr()
        h = np.ascontiguousarray(host).copy()
        return h, h.ctypes.data

    def _fetch(self, handle, dtype):
        if self.device == "gpu":
            import torch
            torch.cuda.synchronize()
            return handle.cpu().numpy().view(dtype).copy()
        return handle.view(dtype).copy() if handle.dtype != dtype else handle.copy()

    def transform(self, x, shape, batch=1, inverse=False, both=False, **kw):
        """x: flat numpy array in the library's buffer layout.  Returns the buffer after the transform
        (after forward and after inverse when both=True)."""
        dp = x.dtype in (np.float64, np.complex128)
        h, ptr = self._alloc(x)
        app = api.App(list(shape), batch, dp=dp, buffer_ptr=ptr, lib=self.lib, **kw)
        try:
            if both:
                app.forward()
                y = self._fetch(h, x.dtype)
                app.inverse()
                z = self._fetch(h, x.dtype)
                return y, z, app.uploads()
            app.append(inverse)
            return self._fetch(h, x.dtype), app.uploads(inverse)
        finally:
            app.delete()
