"""Development probe: what the reference VkFFT (oracle/_ref, HIP backend) returns for convolution configurations, against the definition."""
import ctypes as C, numpy as np, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
ref = C.CDLL(os.path.join(root, "oracle", "_ref", "libvkfft_ref.so"))
ref.ref_convolution.restype = C.c_int
rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
rng = np.random.default_rng(1)
def conv(shape, m, r2c=False):
    dims = tuple(reversed(shape)); ksys = m*m if m>1 else 1; cf = m
    ax = tuple(range(-len(dims), 0))
    if r2c:
        pad = dims[:-1] + (shape[0]+2,)
        k0 = rng.uniform(-1,1,(1,ksys)+dims); d0 = rng.uniform(-1,1,(1,cf)+dims)
        kern = np.zeros((1,ksys)+pad, np.float32); kern[..., :shape[0]] = k0
        data = np.zeros((1,cf)+pad, np.float32); data[..., :shape[0]] = d0
        K = np.fft.rfftn(k0, axes=ax); X = np.fft.rfftn(d0, axes=ax); inv = lambda a: np.fft.irfftn(a, s=dims, axes=ax)
    else:
        k0 = rng.uniform(-1,1,(1,ksys)+dims)+1j*rng.uniform(-1,1,(1,ksys)+dims); d0 = rng.uniform(-1,1,(1,cf)+dims)+1j*rng.uniform(-1,1,(1,cf)+dims)
        kern = k0.astype(np.complex64); data = d0.astype(np.complex64)
        K = np.fft.fftn(k0, axes=ax); X = np.fft.fftn(d0, axes=ax); inv = lambda a: np.fft.ifftn(a, axes=ax)
    size = (C.c_uint64*4)(*shape)
    rc = ref.ref_convolution(len(shape), size, int(r2c), 0, C.c_uint64(cf), C.c_uint64(m), C.c_uint64(1), 0, 0, 0, C.c_uint64(ksys), kern.ctypes.data_as(C.c_void_p), C.c_uint64(kern.nbytes), data.ctypes.data_as(C.c_void_p), C.c_uint64(data.nbytes), None, 0)
    got = data[..., :shape[0]] if r2c else data
    Y = np.zeros_like(X); YT = np.zeros_like(X)
    for j in range(m):
        for l in range(m):
            Y[0,j] += K[0,j*m+l]*X[0,l]; YT[0,j] += K[0,l*m+j]*X[0,l]
    print(shape, "m", m, "r2c", r2c, "rc", rc, "| vs K[j*m+l]:", rel(got, inv(Y)), " vs transposed:", rel(got, inv(YT)))
import ast
if len(sys.argv) > 1:
    a = ast.literal_eval(sys.argv[1]); conv(*a)
else:
    import subprocess
    for a in [((243,12),3), ((64,),2), ((64,),3), ((32,16,8),2,True), ((32,16,8),3,False), ((32,16),3), ((32,16),3,True), ((4096,),1), ((64,64),1)]:
        r = subprocess.run([sys.executable, "-u", __file__, repr(a)], capture_output=True, text=True)
        print(a, "->", (r.stdout.strip().splitlines() or ["(no output)"])[-1], "| exit", r.returncode, flush=True)
