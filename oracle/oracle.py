"""TEST INFRASTRUCTURE ONLY: Python face of the CPU oracle (oracle/vkfft_oracle.c) plus the independent
double/long-double ground truth (scipy.fft = the role FFTW plays in the reference's precision samples,
sample_11_precision_VkFFT_single.cpp:116-132) and the reference's error metric (:289-331).
Never imported by the product package."""
import ctypes as C
import os
import subprocess
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None
_mkl = None


def build():
    subprocess.check_call(["make", "-s", "-C", HERE, "_build/liboracle.so", "_build/libfftw_mkl.so"])


def lib():
    global _lib
    if _lib is None:
        p = os.path.join(HERE, "_build", "liboracle.so")
        if not os.path.exists(p):
            build()
        _lib = C.CDLL(p)
    return _lib


def _sizes(shape):
    arr = (C.c_uint64 * len(shape))(*shape)
    return arr


def c2c(x, shape, batch=1, inverse=False, split0=0):
    """x: complex64/complex128 flat array of batch*prod(shape) elements, shape = (W,H,D) with W fastest."""
    x = np.ascontiguousarray(x).copy()
    fn = lib().f64_oracle_c2c if x.dtype == np.complex128 else lib().f32_oracle_c2c
    fn(x.ctypes.data_as(C.c_void_p), C.c_int(len(shape)), _sizes(shape), C.c_uint64(batch), C.c_int(int(inverse)), C.c_uint64(split0))
    return x


def r2c_rows(x, N, rows):
    dp = x.dtype == np.float64
    out = np.zeros(rows * (N // 2 + 1), dtype=np.complex128 if dp else np.complex64)
    fn = lib().f64_oracle_r2c_rows if dp else lib().f32_oracle_r2c_rows
    x = np.ascontiguousarray(x)
    fn(x.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), C.c_uint64(N), C.c_uint64(rows), C.c_uint64(N), C.c_uint64(N // 2 + 1))
    return out


def c2r_rows(X, N, rows):
    dp = X.dtype == np.complex128
    out = np.zeros(rows * N, dtype=np.float64 if dp else np.float32)
    fn = lib().f64_oracle_c2r_rows if dp else lib().f32_oracle_c2r_rows
    X = np.ascontiguousarray(X)
    fn(X.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), C.c_uint64(N), C.c_uint64(rows), C.c_uint64(N // 2 + 1), C.c_uint64(N))
    return out


def r2r(x, shape, batch=1, type=2, dst=False):
    x = np.ascontiguousarray(x).copy()
    fn = lib().f64_oracle_r2r if x.dtype == np.float64 else lib().f32_oracle_r2r
    fn(x.ctypes.data_as(C.c_void_p), C.c_int(len(shape)), _sizes(shape), C.c_uint64(batch), C.c_int(type), C.c_int(int(dst)))
    return x


def rand_sample(n):
    """glibc rand() seed-1 stream mapped to [-1,1] exactly as the reference's samples fill their buffers."""
    out = np.empty(n, dtype=np.float32)
    lib().oracle_fill_rand_sample(out.ctypes.data_as(C.c_void_p), C.c_uint64(n))
    return out


# ---- independent ground truth (double / long double), FFTW conventions -----------------------------------
def truth_c2c(x, shape, batch=1, inverse=False, longdouble=False):
    import scipy.fft as sf
    ct = np.clongdouble if longdouble else np.complex128
    a = np.asarray(x).astype(ct).reshape([batch] + list(shape)[::-1])
    axes = tuple(range(1, 1 + len(shape)))
    if inverse:
        r = sf.ifftn(a, axes=axes) * np.prod(shape)
    else:
        r = sf.fftn(a, axes=axes)
    return r.reshape(-1)


def truth_r2r(x, shape, batch=1, type=2, dst=False, longdouble=False):
    import scipy.fft as sf
    rt = np.longdouble if longdouble else np.float64
    a = np.asarray(x).astype(rt).reshape([batch] + list(shape)[::-1])
    f = sf.dstn if dst else sf.dctn
    axes = tuple(range(1, 1 + len(shape)))
    return f(a, type=type, axes=axes).reshape(-1)


def errors(y, ref):
    """max/avg absolute and relative element errors (the reference's metric) + relative L2 (its published plots)."""
    y = np.asarray(y).astype(np.clongdouble if np.iscomplexobj(ref) else np.longdouble).reshape(-1)
    ref = np.asarray(ref).reshape(-1)
    d = np.abs(y - ref)
    mag = np.abs(ref)
    rel = d / np.where(mag > 0, mag, 1)
    return dict(max_abs=float(d.max()), avg_abs=float(d.mean()), max_rel=float(rel.max()), avg_rel=float(rel.mean()),
                rel_l2=float(np.sqrt((d.astype(np.float64) ** 2).sum()) / max(float(np.sqrt((mag.astype(np.float64) ** 2).sum())), 1e-300)))


# ---- FFTW API through MKL (the reference's CPU path), used for cross-checks and the timed CPU baseline ----
def mkl():
    global _mkl
    if _mkl is None:
        p = os.path.join(HERE, "_build", "libfftw_mkl.so")
        if not os.path.exists(p):
            build()
        l = C.CDLL(p)
        l.fftw_mkl_available.restype = C.c_int
        l.fftw_mkl_c2c.restype = C.c_double
        l.fftw_mkl_c2c.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        _mkl = l
    return _mkl


def fftw_available():
    os.environ.setdefault("MKL_THREADING_LAYER", "GNU")
    return bool(mkl().fftw_mkl_available())


def fftw_set_threads(n):
    """FFTW threading API (fftw_init_threads / fftw_plan_with_nthreads) for plans created afterwards; False if not exported."""
    l = mkl()
    l.fftw_mkl_set_threads.restype = C.c_int
    l.fftw_mkl_set_threads.argtypes = [C.c_int]
    return bool(l.fftw_mkl_set_threads(int(n)))


def fftw_c2c(x, N, batch, inverse=False, reps=1):
    """in-place batched 1D C2C via the FFTW3 API (MKL); returns (result, seconds per execute)."""
    x = np.ascontiguousarray(x).copy()
    dp = x.dtype == np.complex128
    t = mkl().fftw_mkl_c2c(x.ctypes.data_as(C.c_void_p), N, batch, 1 if inverse else -1, int(dp), reps)
    if t < 0:
        raise RuntimeError("FFTW/MKL not available")
    return x, t
