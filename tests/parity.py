"""Parity checks shared by the CPU (emulated) and GPU test modules: library (through the C-ABI) vs the oracle
on the same seeded inputs."""
import numpy as np
from helpers import rel_l2, TOL


def seeded_complex(n, dp, seed):
    rng = np.random.default_rng(seed)
    v = rng.uniform(-1, 1, 2 * n).astype(np.float64 if dp else np.float32)
    return v.view(np.complex128 if dp else np.complex64)


def check_c2c(runner, oracle, shape, batch, dp, *, kind="c2c", use_c_oracle=None, seed=1):
    n = int(np.prod(shape)) * batch
    x = seeded_complex(n, dp, seed + n)
    tol = TOL[(kind, dp)]
    y, z, up = runner.transform(x, shape, batch, both=True)
    truth = oracle.truth_c2c(x, shape, batch, longdouble=dp)
    e = rel_l2(y, truth)
    assert e < tol, f"forward {shape} b={batch} dp={dp}: rel-L2 {e:.3e} >= {tol}"
    # forward o inverse = N x (unnormalised), bound: twice the one-way bound
    e2 = rel_l2(z, x.astype(np.complex128) * np.prod(shape))
    assert e2 < 2 * tol, f"roundtrip {shape}: {e2:.3e}"
    if use_c_oracle is None:
        use_c_oracle = n <= (1 << 16)
    if use_c_oracle:  # the C restatement of the reference's algorithm, same precision
        o = oracle.c2c(x, shape, batch)
        assert rel_l2(y, o) < 2 * tol
    return up


def check_r2c(runner, oracle, shape, batch, dp, seed=2):
    rng = np.random.default_rng(seed + int(np.prod(shape)))
    W = shape[0]; Wc = W // 2 + 1
    rest = int(np.prod(shape[1:])) if len(shape) > 1 else 1
    rt = np.float64 if dp else np.float32
    rows = rng.uniform(-1, 1, (batch * rest, W)).astype(rt)
    buf = np.zeros((batch * rest, 2 * Wc), dtype=rt); buf[:, :W] = rows
    tol = TOL[("real", dp)]
    y, z, _ = runner.transform(buf.reshape(-1), shape, batch, both=True, r2c=True)
    Y = y.view(np.complex128 if dp else np.complex64)
    truth = np.fft.rfftn(rows.astype(np.float64).reshape([batch] + list(shape)[::-1]), axes=tuple(range(1, 1 + len(shape)))).reshape(-1)
    assert rel_l2(Y, truth) < tol, f"r2c {shape}"
    back = z.reshape(batch * rest, 2 * Wc)[:, :W]
    assert rel_l2(back, rows.astype(np.float64) * np.prod(shape)) < 2 * tol, f"c2r(r2c) {shape}"
    if len(shape) == 1:
        o = oracle.r2c_rows(rows.reshape(-1), W, batch)
        assert rel_l2(Y, o) < 2 * tol


def check_r2r(runner, oracle, shape, batch, dp, type, dst, seed=3):
    rng = np.random.default_rng(seed + int(np.prod(shape)) + type)
    rt = np.float64 if dp else np.float32
    x = rng.uniform(-1, 1, int(np.prod(shape)) * batch).astype(rt)
    tol = TOL[("real", dp)]
    kw = dict(dst=type) if dst else dict(dct=type)
    y, z, _ = runner.transform(x, shape, batch, both=True, **kw)
    truth = oracle.truth_r2r(x, shape, batch, type=type, dst=dst, longdouble=dp)
    assert rel_l2(y, truth) < tol, f"{'dst' if dst else 'dct'}{type} {shape}"
    norm = 1.0
    for s in shape:
        norm *= (2.0 * (s + 1) if dst else 2.0 * (s - 1)) if type == 1 else 2.0 * s
    assert rel_l2(z, x.astype(np.float64) * norm) < 2 * tol, f"inverse {'dst' if dst else 'dct'}{type} {shape}"
    o = oracle.r2r(x, shape, batch, type=type, dst=dst)
    assert rel_l2(y, o) < 2 * tol
