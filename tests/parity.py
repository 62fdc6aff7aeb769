"""Parity checks shared by the CPU (emulated) and GPU test modules: library (through the C-ABI) vs the oracle
on the same seeded inputs."""
import numpy as np
from helpers import rel_l2, TOL, assert_elementwise


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
    assert_elementwise(y, truth, kind, dp, f"forward {shape} b={batch} dp={dp}")
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
    assert_elementwise(Y, truth, "real", dp, f"r2c {shape}")
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
    assert_elementwise(y, truth, "real", dp, f"{'dst' if dst else 'dct'}{type} {shape}")
    norm = 1.0
    for s in shape:
        norm *= (2.0 * (s + 1) if dst else 2.0 * (s - 1)) if type == 1 else 2.0 * s
    assert rel_l2(z, x.astype(np.float64) * norm) < 2 * tol, f"inverse {'dst' if dst else 'dct'}{type} {shape}"
    o = oracle.r2r(x, shape, batch, type=type, dst=dst)
    assert rel_l2(y, o) < 2 * tol


# ---- op-FFT kernel family (kernel_opfft.h): one case per generated table entry -----------------------------------
def opfft_cases():
    """[(family, L, col, dp)] parsed from the generated tables vkfft_amd/csrc/opfft_table_*.inc."""
    import os, re
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vkfft_amd", "csrc")
    cases = []
    for dp, tag in ((False, "f32"), (True, "f64")):
        for col in (False, True):
            txt = "".join(open(os.path.join(root, "opfft_table_%s_%s_%d.inc" % (tag, "col" if col else "row", h))).read() for h in (0, 1))
            for fam, L in re.findall(r"// (\w+) L=(\d+)", txt):
                if fam == "dst1" and int(L) == 4:
                    continue  # DST-I of length 1: size-1 axes are omitted (reference InitializeApp.h:1378-1381), nothing to run
                cases.append((fam, int(L), col, dp))
    return cases


def opfft_transform_of(fam, L, col):
    """(shape, kwargs of Runner.transform, real data?) of the smallest transform whose plan uses this table entry."""
    if fam in ("c2c4", "c2cT"):  # Four-Step passes of a 1D transform: L * L (two passes) exercises the first-pass kernel,
        return ((L * L * (L if fam == "c2c4" else 1),), {}, False)  # L^3 (three passes) the middle one
    if fam in ("r2cf", "c2rf"):
        return (L,), dict(r2c=True), True
    n = {"r2c": 2 * L, "c2r": 2 * L, "dct2": L, "dct3": L, "dct2h": 2 * L, "dct3h": 2 * L, "dct4": 2 * L, "dct1": L // 2 + 1, "dct1h": L + 1, "dst1": L // 2 - 1, "c2c": L}[fam]
    kw = {"r2c": dict(r2c=True), "c2r": dict(r2c=True), "dct2": dict(dct=2), "dct3": dict(dct=3), "dct2h": dict(dct=2), "dct3h": dict(dct=3), "dct4": dict(dct=4), "dct1": dict(dct=1), "dct1h": dict(dct=1),
          "dst1": dict(dst=1), "c2c": {}}[fam]
    shape = (24, n) if col else (n,)
    return shape, kw, fam != "c2c"


def check_opfft_case(runner, oracle, fam, L, col, dp, load_elems=0, max_points=1 << 23):
    """Parity of one table entry against the oracle; with load_elems > 0 also a chip-filling batch of the same sequences
    whose output must repeat the small-batch output bit for bit."""
    shape, kw, real = opfft_transform_of(fam, L, col)
    batch = 3
    if fam in ("c2c", "c2c4", "c2cT"):
        if int(np.prod(shape)) > max_points:
            return  # (covered by the smaller factor lengths; a 2^23+ point host reference per table entry is too slow)
        check_c2c(runner, oracle, shape, 1 if fam != "c2c" else batch, dp, use_c_oracle=False)
    elif fam in ("r2c", "c2r", "r2cf", "c2rf"):
        check_r2c(runner, oracle, shape, batch, dp)
    else:
        check_r2r(runner, oracle, shape, batch, dp, int(fam[3]), fam.startswith("dst"))
    if load_elems:
        n = int(np.prod(shape))
        rng = np.random.default_rng(L)
        if fam in ("c2c4", "c2cT"):
            return
        if fam == "c2c":
            x = seeded_complex(n * batch, dp, L)
        elif fam in ("r2c", "c2r", "r2cf", "c2rf"):
            W = shape[0]
            x = np.zeros((batch * n // W, 2 * (W // 2 + 1)), dtype=np.float64 if dp else np.float32)
            x[:, :W] = rng.uniform(-1, 1, (batch * n // W, W))
            x = x.reshape(-1)
        else:
            x = rng.uniform(-1, 1, n * batch).astype(np.float64 if dp else np.float32)
        # the repeated unit holds an EVEN number of rows: where two real rows travel through one complex transform (PassParams::pairRows, the reference's
        # mergeSequencesR2C) a row's last bits depend on the row it is paired with, so "bit for bit" holds for equal PAIRS — with three rows per unit the third
        # row would meet a zero partner in the small batch and the first row of the next unit in the large one (seen on the device: dct3 / r2cf / dct2 of
        # complex length 15, c2rf of 9, the lengths 8 ... 16 that moved between the generic maps in round 4)
        x2 = np.tile(x, 2)
        small = runner.transform(x2, shape, 2 * batch, both=True, **kw)
        reps = max(2, load_elems // (n * 2 * batch))
        big = runner.transform(np.tile(x2, reps), shape, 2 * batch * reps, both=True, **kw)
        for s, b in zip(small[:2], big[:2]):
            bad = np.flatnonzero(np.tile(s, reps).view(np.uint8) != b.view(np.uint8))
            assert bad.size == 0, (fam, L, col, dp, bad[:8])


def mixconv_entries():
    """(dp, rader, col, p or M) of every instance of kernel_mixconv.h (generated tables)"""
    import os, re, glob
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    txt = "".join(open(f).read() for f in sorted(glob.glob(os.path.join(root, "vkfft_amd", "csrc", "mixconv_table_*.inc"))))
    out = []
    for m in re.finditer(r"VKFFT_MC\((float|double), \w+, (\d), (\d),[^\n]*=(\d+)\n", txt):
        out.append((m.group(1) == "double", int(m.group(2)) == 1, int(m.group(3)) == 1, int(m.group(4))))
    return sorted(set(out))


def mixconv_length_for(rader, v):
    """a transform length that the instance serves: the Rader prime itself; for a Bluestein padded length M the largest prime <= (M+1)/2 that has no
    radix-kernel instance (> 31) — None when there is none (the shortest ladder lengths)"""
    if rader:
        return v
    n = (v + 1) // 2
    while n > 31:
        if all(n % q for q in range(2, int(n ** 0.5) + 1)):
            return n
        n -= 1
    return None
