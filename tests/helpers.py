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
