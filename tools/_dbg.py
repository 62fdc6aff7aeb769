import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from vkfft_amd import api
from helpers import Runner
lib = api.load()
run = Runner(lib, "gpu")
for N, batch in [(74, 7), (123, 7), (123, 64), (296, 7), (889, 7), (2670, 3), (2670, 40)]:
    rng = np.random.default_rng(N)
    x = (rng.uniform(-1, 1, N * batch) + 1j * rng.uniform(-1, 1, N * batch)).astype(np.complex64)
    y, up = run.transform(x, (N,), batch)
    ref = np.fft.fft(x.astype(np.complex128).reshape(batch, N), axis=1)
    err = np.abs(y.reshape(batch, N) - ref)
    bad = np.argwhere(err > 1e-3 * np.abs(ref).max())
    print(N, batch, "max err", err.max(), "bad count", len(bad), "first bad", bad[:6].tolist(), "rows with bad", sorted(set(bad[:, 0].tolist()))[:10], "cols", sorted(set(bad[:, 1].tolist()))[:12], flush=True)
