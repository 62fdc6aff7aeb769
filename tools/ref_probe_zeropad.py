"""Development probe: what the reference VkFFT (oracle/_ref, HIP backend) returns for zero-padding configurations, against numpy."""
import ctypes as C, numpy as np, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ref = C.CDLL(os.path.join(root, "oracle", "_ref", "libvkfft_ref.so"))
ref.ref_transform_zeropad.restype = C.c_int; ref.ref_convolution.restype = C.c_int
u4 = lambda v: (C.c_uint64 * 4)(*v)
rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
rng = np.random.default_rng(1)
def zp(shape, pads, inverse=0, freq=0):
    dims = tuple(reversed(shape)); B = 2
    x = (rng.uniform(-1, 1, (B,) + dims) + 1j * rng.uniform(-1, 1, (B,) + dims)).astype(np.complex64)
    flags = [0]*4; left=[0]*4; right=[0]*4
    for a,(l,r) in pads.items(): flags[a],left[a],right[a]=1,l,r
    r_ = x.copy()
    rc = ref.ref_transform_zeropad(0, len(shape), u4(list(shape)+[1]*(4-len(shape))), C.c_uint64(B), 0, inverse, u4(flags), u4(left), u4(right), freq, r_.ctypes.data_as(C.c_void_p), C.c_uint64(r_.nbytes))
    ax = tuple(range(-len(dims), 0))
    def masked(which):
        m = x.astype(np.complex128).copy()
        for a in which:
            l, r = pads[a]; idx=[slice(None)]*m.ndim; idx[m.ndim-1-a]=slice(l,r); m[tuple(idx)] = 0
        return m
    full = np.fft.fftn(masked(pads.keys()), axes=ax)
    print(shape, pads, "rc", rc, "vs fft(masked all):", rel(r_, full))
    # compare only outside padded ranges of the output
    keep = np.ones(x.shape, bool)
    for a,(l,r) in pads.items():
        idx=[slice(None)]*x.ndim; idx[x.ndim-1-a]=slice(l,r); keep[tuple(idx)] = False
    print("   on the non-padded output region only:", rel(r_[keep], full[keep]))
    for a in pads:
        keep1 = np.ones(x.shape, bool); l,r = pads[a]; idx=[slice(None)]*x.ndim; idx[x.ndim-1-a]=slice(l,r); keep1[tuple(idx)] = False
        print("   excluding output range of axis", a, ":", rel(r_[keep1], full[keep1]))
zp((64,32), {0:(32,64)})
zp((64,32), {1:(16,32)})
zp((64,32), {0:(32,64), 1:(16,32)})
zp((16,16,16), {0:(8,16),1:(8,16),2:(8,16)})
print("---- axis-1 hypotheses")
shape=(64,32); dims=(32,64); B=2
x = (rng.uniform(-1, 1, (B,) + dims) + 1j * rng.uniform(-1, 1, (B,) + dims)).astype(np.complex64)
r_ = x.copy()
rc = ref.ref_transform_zeropad(0, 2, u4([64,32,1,1]), C.c_uint64(B), 0, 0, u4([0,1,0,0]), u4([0,16,0,0]), u4([0,32,0,0]), 0, r_.ctypes.data_as(C.c_void_p), C.c_uint64(r_.nbytes))
X = x.astype(np.complex128)
print("unmasked fft2:", rel(r_, np.fft.fft2(X)))
m = X.copy(); m[:, 16:32, :] = 0
print("masked fft2:", rel(r_, np.fft.fft2(m)))
h = np.fft.fft(X, axis=-1); h[:, 16:32, :] = X[:, 16:32, :]
print("x-fft skipped on padded rows, y-fft reads all:", rel(r_, np.fft.fft(h, axis=-2)))
h2 = np.fft.fft(X, axis=-1); print("only x-fft done:", rel(r_, h2))
h3 = np.fft.fft(X, axis=-1); h3[:, 16:32, :] = X[:, 16:32, :]; print("x-fft on rows<16 only, no y-fft:", rel(r_, h3))
print("batch 0 only, masked:", rel(r_[0], np.fft.fft2(m)[0]), " batch 1:", rel(r_[1], np.fft.fft2(m)[1]))
print("---- axis-1 padding, input really zero in the padded range")
xz = x.copy(); xz[:, 16:32, :] = 0; r_ = xz.copy()
rc = ref.ref_transform_zeropad(0, 2, u4([64,32,1,1]), C.c_uint64(B), 0, 0, u4([0,1,0,0]), u4([0,16,0,0]), u4([0,32,0,0]), 0, r_.ctypes.data_as(C.c_void_p), C.c_uint64(r_.nbytes))
print("masked fft2:", rel(r_, np.fft.fft2(xz.astype(np.complex128))))
r2 = xz.copy()
rc = ref.ref_transform_zeropad(0, 2, u4([64,32,1,1]), C.c_uint64(B), 0, 0, u4([0,0,0,0]), u4([0,0,0,0]), u4([0,0,0,0]), 0, r2.ctypes.data_as(C.c_void_p), C.c_uint64(r2.nbytes))
print("reference without the padding flags on the same input:", rel(r2, np.fft.fft2(xz.astype(np.complex128))))
np.set_printoptions(precision=3, linewidth=200)
F = np.fft.fft2(xz.astype(np.complex128))
print("row norms ref/true, batch 0:", np.round(np.linalg.norm(r_[0], axis=1) / np.linalg.norm(F[0], axis=1), 2))
