"""Seeded differential fuzzing of the product sources (CPU SIMT emulation) against the oracle: random 1D-3D C2C / R2C / DCT / DST
plans of smooth, prime and arbitrary lengths in both precisions, and C2C plans on padded strides / out-of-place buffers.  A plan
may be rejected as unsupported (VkFFT error 3002-3005: documented limits) but must never return a wrong result."""
import random

import numpy as np
import pytest

import parity
from helpers import Runner, rel_l2
from vkfft_amd import api

UNSUPPORTED = (3002, 3003, 3004, 3005)


@pytest.fixture(scope="module")
def run(emu_lib):
    return Runner(emu_lib, "emu")


def _smooth(rnd, maxn):
    while True:
        n = 1
        for p, emax in ((2, 12), (3, 5), (5, 4), (7, 3), (11, 2), (13, 2)):
            n *= p ** rnd.randint(0, emax)
        if 2 <= n <= maxn:
            return n


import os as _os
_EXTRA_1D = int(_os.environ.get("VKFFT_FUZZ_EXTRA_SEEDS", "0"))  # (development: more seeds for a longer hunt)


@pytest.mark.parametrize("seed", range(6 + _EXTRA_1D))
def test_random_plans_against_the_oracle(run, oracle, seed):
    rnd = random.Random(1000 + seed)
    unsupported = 0
    cases = 30
    for _ in range(cases):
        kind = rnd.choice(["c2c1", "c2c1", "c2c1p", "c2c2", "c2c3", "r2c1", "r2c1a", "r2c2", "dct1d", "dct1da", "dct2d"])
        dp = rnd.random() < 0.3
        try:
            if kind == "c2c1":
                N = _smooth(rnd, 200000)
                parity.check_c2c(run, oracle, (N,), rnd.randint(1, max(1, min(40, 300000 // N))), dp, use_c_oracle=False)
            elif kind == "c2c1p":
                N = rnd.randint(2, 20000)
                parity.check_c2c(run, oracle, (N,), rnd.randint(1, max(1, min(20, 100000 // N))), dp, kind="bluestein", use_c_oracle=False)
            elif kind == "c2c2":
                parity.check_c2c(run, oracle, (_smooth(rnd, 500), _smooth(rnd, 500)), rnd.randint(1, 3), dp, use_c_oracle=False)
            elif kind == "c2c3":
                parity.check_c2c(run, oracle, (_smooth(rnd, 50), _smooth(rnd, 50), _smooth(rnd, 30)), rnd.randint(1, 2), dp, use_c_oracle=False)
            elif kind == "r2c1":
                N = _smooth(rnd, 16000)
                parity.check_r2c(run, oracle, (N,), rnd.randint(1, max(1, min(30, 100000 // N))), dp)
            elif kind == "r2c1a":
                parity.check_r2c(run, oracle, (rnd.randint(2, 5000),), rnd.randint(1, 10), dp)
            elif kind == "r2c2":
                parity.check_r2c(run, oracle, (2 * _smooth(rnd, 250), _smooth(rnd, 300)), rnd.randint(1, 3), dp)
            elif kind == "dct1d":
                t = rnd.randint(1, 4)
                parity.check_r2r(run, oracle, (max(3, _smooth(rnd, 3000 if t != 1 else 1500)),), rnd.randint(1, 15), dp, t, rnd.random() < 0.4)
            elif kind == "dct1da":
                parity.check_r2r(run, oracle, (rnd.randint(3, 1800),), rnd.randint(1, 10), dp, rnd.randint(1, 4), rnd.random() < 0.5)
            else:
                parity.check_r2r(run, oracle, (max(3, _smooth(rnd, 250)), max(3, _smooth(rnd, 250))), rnd.randint(1, 2), dp, rnd.randint(2, 4), rnd.random() < 0.4)
        except api.VkFFTError as e:
            assert e.code in UNSUPPORTED, e
            unsupported += 1
    assert unsupported <= cases // 4


@pytest.mark.parametrize("seed", range(3))
def test_random_padded_strides_and_out_of_place(emu_lib, seed):
    rnd = random.Random(2000 + seed)
    for it in range(25):
        dp = rnd.random() < 0.3
        ct = np.complex128 if dp else np.complex64
        nd = rnd.choice([1, 1, 2, 3])
        shape = [rnd.choice([_smooth(rnd, 3000 if nd == 1 else 150), rnd.randint(2, 200 if nd == 1 else 50)]) for _ in range(nd)]
        B = rnd.randint(1, 4)
        pitches, acc = [], 1
        for s in shape:
            acc = acc * s + rnd.choice([0, 0, 1, 3, 8])
            pitches.append(acc)
        total = pitches[-1] * B
        rng = np.random.default_rng(seed * 100 + it)
        src = (rng.uniform(-1, 1, total) + 1j * rng.uniform(-1, 1, total)).astype(ct)
        orig = src.copy()
        strides = [pitches[-1]] + [pitches[i - 1] if i > 0 else 1 for i in range(nd - 1, -1, -1)]
        idx = np.indices([B] + shape[::-1]).reshape(nd + 1, -1)
        off = sum(idx[d] * strides[d] for d in range(nd + 1))
        view = lambda a: a[off].reshape([B] + shape[::-1]).astype(np.complex128)
        kw = dict(bufferStride=pitches + [0] * (4 - nd))
        oop = rnd.random() < 0.5
        try:
            if oop:
                dst = np.zeros(total, ct)
                app = api.App(shape, B, dp=dp, buffer_ptr=dst.ctypes.data, isInputFormatted=1, inputBuffer=src.ctypes.data,
                              inputBufferStride=pitches + [0] * (4 - nd), lib=emu_lib, **kw)
                app.forward(); out = dst
            else:
                app = api.App(shape, B, dp=dp, buffer_ptr=src.ctypes.data, lib=emu_lib, **kw)
                app.forward(); out = src
        except api.VkFFTError as e:
            assert e.code in UNSUPPORTED, e
            continue
        e = rel_l2(view(out), np.fft.fftn(view(orig), axes=tuple(range(1, nd + 1))))
        assert e < (3e-14 if dp else 5e-6), (shape, B, dp, pitches, oop, e)
        if oop:
            assert np.array_equal(src, orig)
        app.delete()


def _any_len(rnd, maxn):
    """smooth, prime or arbitrary: the three planner classes"""
    c = rnd.random()
    if c < 0.4:
        return _smooth(rnd, maxn)
    n = rnd.randint(2, maxn)
    if c < 0.7:
        while any(n % q == 0 for q in range(2, int(n ** 0.5) + 1)):
            n -= 1
    return max(2, n)


import os
_EXTRA = int(os.environ.get("VKFFT_FUZZ_EXTRA_SEEDS", "0"))  # (development: more seeds for a longer hunt)


@pytest.mark.parametrize("seed", range(4 + _EXTRA))
def test_random_planes_and_volumes_of_any_length(run, oracle, seed):
    """2-D / 3-D C2C, R2C and DCT systems whose axes are smooth, prime or arbitrary, small batches: the column-tile kernels (merged tiles over two dimensions,
    Rader / Bluestein column tiles, transposed rows) and the real-row forms of round 3 behind every combination the planner can produce"""
    rnd = random.Random(5000 + seed)
    unsupported = 0
    cases = 20
    for _ in range(cases):
        dp = rnd.random() < 0.3
        nd = rnd.choice([2, 2, 3])
        lim = 120 if nd == 2 else 40
        shape = tuple(_any_len(rnd, lim) for _ in range(nd))
        batch = rnd.randint(1, 5)
        kind = rnd.choice(["c2c", "c2c", "r2c", "dct"])
        try:
            if kind == "c2c":
                parity.check_c2c(run, oracle, shape, batch, dp, kind="bluestein", use_c_oracle=False)
            elif kind == "r2c":
                parity.check_r2c(run, oracle, shape, batch, dp)
            else:
                parity.check_r2r(run, oracle, tuple(max(3, s) for s in shape), batch, dp, rnd.randint(1, 4), rnd.random() < 0.4)
        except api.VkFFTError as e:
            assert e.code in UNSUPPORTED, (e, shape, kind)
            unsupported += 1
    assert unsupported <= cases // 4
