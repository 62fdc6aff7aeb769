export TMPDIR=/tmp; O=gpurun_out/r04g; mkdir -p $O
timeout 40 python - > $O/ours.log 2>&1 <<'PY'
import sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from helpers import Runner
from vkfft_amd import api
import scipy.fft as sf
run = Runner(api.load(), "gpu")
for B in (1, 2, 3):
    x = np.random.default_rng(B).uniform(-1, 1, 1451 * B).astype(np.float32)
    y, _ = run.transform(x, (1451,), B, dct=4)
    t = sf.dct(x.reshape(B, 1451).astype(np.float64), type=4, axis=1).reshape(-1)
    print("ours 1451 x", B, "rel err", float(np.linalg.norm(y - t) / np.linalg.norm(t)), flush=True)
PY
tail -4 $O/ours.log
timeout 40 python - > $O/ref.log 2>&1 <<'PY'
import ctypes as C, numpy as np
ref = C.CDLL("oracle/_ref/libvkfft_ref.so"); ref.ref_transform.restype = C.c_int
for B in (64, 2):
    x = np.random.default_rng(B).uniform(-1, 1, 1451 * B).astype(np.float32)
    print("reference 1451 x", B, "...", flush=True)
    rc = ref.ref_transform(C.c_int(14), C.c_int(1), (C.c_uint64 * 4)(1451), C.c_uint64(B), C.c_int(0), C.c_int(0), C.c_int(0), x.ctypes.data_as(C.c_void_p), C.c_uint64(x.nbytes), None)
    print("rc", rc, flush=True)
PY
echo "exit code of the reference process: $?" >> $O/ref.log; tail -5 $O/ref.log
