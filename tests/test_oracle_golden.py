"""CPU tests of the oracle itself: it must agree with the independent double-precision ground truth (the role
FFTW plays for the reference) and with the reference's own outputs captured on an MI355X (tests/golden)."""
import numpy as np
import pytest

from helpers import rel_l2


def test_glibc_rand_stream(oracle):
    # the reference fills buffers with unseeded rand(): first values of the glibc seed-1 stream
    r = oracle.rand_sample(4)
    assert np.allclose(r, [0.68037546, -0.21123415, 0.56619847, 0.5968801], atol=1e-7)


@pytest.mark.parametrize("N", [2, 3, 4, 5, 7, 8, 9, 11, 13, 16, 30, 64, 100, 243, 343, 1024, 1080, 17, 127, 1009])
@pytest.mark.parametrize("dp", [False, True])
def test_oracle_c2c_vs_truth(oracle, N, dp):
    rng = np.random.default_rng(N)
    ct = np.complex128 if dp else np.complex64
    x = (rng.uniform(-1, 1, 3 * N) + 1j * rng.uniform(-1, 1, 3 * N)).astype(ct)
    tol = 4e-15 if dp else 1e-6
    for inv in (False, True):
        e = rel_l2(oracle.c2c(x, (N,), 3, inverse=inv), oracle.truth_c2c(x, (N,), 3, inverse=inv, longdouble=dp))
        assert e < tol, (N, dp, inv, e)


def test_oracle_fourstep_and_nd(oracle):
    rng = np.random.default_rng(5)
    x = (rng.uniform(-1, 1, 8192) + 1j * rng.uniform(-1, 1, 8192)).astype(np.complex64)
    assert rel_l2(oracle.c2c(x, (8192,), 1, split0=64), oracle.truth_c2c(x, (8192,))) < 1e-6
    y = (rng.uniform(-1, 1, 2 * 720) + 1j * rng.uniform(-1, 1, 2 * 720)).astype(np.complex128)
    for inv in (False, True):
        assert rel_l2(oracle.c2c(y, (12, 10, 6), 2, inverse=inv), oracle.truth_c2c(y, (12, 10, 6), 2, inverse=inv)) < 4e-15


@pytest.mark.parametrize("N", [2, 15, 16, 100, 243, 256])
def test_oracle_r2c_c2r(oracle, N):
    rng = np.random.default_rng(N)
    x = rng.uniform(-1, 1, 2 * N)
    X = oracle.r2c_rows(x, N, 2)
    assert np.abs(X.reshape(2, -1) - np.fft.rfft(x.reshape(2, N), axis=1)).max() < 1e-12
    assert np.abs(oracle.c2r_rows(X, N, 2) - N * x).max() < 1e-11


@pytest.mark.parametrize("type", [1, 2, 3, 4])
@pytest.mark.parametrize("dst", [False, True])
@pytest.mark.parametrize("shape", [(8,), (9,), (64,), (100,), (12, 10)])
def test_oracle_r2r(oracle, type, dst, shape):
    rng = np.random.default_rng(11)
    x = rng.uniform(-1, 1, 2 * int(np.prod(shape)))
    assert rel_l2(oracle.r2r(x, shape, 2, type, dst), oracle.truth_r2r(x, shape, 2, type, dst)) < 1e-13


@pytest.mark.parametrize("N", [3, 5, 7, 9, 11, 13, 15, 21, 45, 105, 239, 1125, 1451])
@pytest.mark.parametrize("dst", [False, True])
def test_oracle_dct4_of_odd_length_same_length_form(oracle, N, dst):
    """the reference's same-length algorithm for DCT-IV / DST-IV of odd length (vkFFT_R2R.h:414-481, 922-972, 1032-1272) as the oracle restates it, in both
    precisions against the double truth (also inside a plane)"""
    rng = np.random.default_rng(N)
    x = rng.uniform(-1, 1, 3 * N)
    assert rel_l2(oracle.r2r(x, (N,), 3, 4, dst), oracle.truth_r2r(x, (N,), 3, 4, dst)) < 1e-13
    assert rel_l2(oracle.r2r(x.astype(np.float32), (N,), 3, 4, dst), oracle.truth_r2r(x.astype(np.float32), (N,), 3, 4, dst)) < 2e-6
    if N <= 45:
        y = rng.uniform(-1, 1, 2 * N * 6)
        assert rel_l2(oracle.r2r(y, (N, 6), 2, 4, dst), oracle.truth_r2r(y, (N, 6), 2, 4, dst)) < 1e-13


def _truth_for_case(oracle, mod, case, x):
    """double-precision ground truth of a golden case, in the buffer layout the library returns."""
    shape, b = case["shape"], case["batch"]
    if case["kind"] == 0:
        return oracle.truth_c2c(x, shape, b, inverse=bool(case["inverse"]))
    if case["kind"] == 1:
        W = shape[0]
        rows = x.reshape(-1, 2 * (W // 2 + 1))[:, :W].astype(np.float64)
        full = rows.reshape([b] + list(shape)[::-1])
        return np.fft.rfftn(full, axes=tuple(range(1, 1 + len(shape)))).reshape(-1)
    if case["kind"] >= 21:
        return oracle.truth_r2r(x, shape, b, type=case["kind"] - 20, dst=True)
    return oracle.truth_r2r(x, shape, b, type=case["kind"] - 10)


def _oracle_for_case(oracle, case, x):
    shape, b = case["shape"], case["batch"]
    if case["kind"] == 0:
        return oracle.c2c(x, shape, b, inverse=bool(case["inverse"]))
    if case["kind"] == 1:
        W = shape[0]
        rows = np.ascontiguousarray(x.reshape(-1, 2 * (W // 2 + 1))[:, :W])
        X = oracle.r2c_rows(rows.reshape(-1), W, rows.shape[0])
        assert len(shape) == 1
        return X
    if case["kind"] >= 21:
        return oracle.r2r(x, shape, b, case["kind"] - 20, True)
    return oracle.r2r(x, shape, b, type=case["kind"] - 10)


def test_golden_reference_outputs_match_truth_and_oracle(oracle, golden):
    """Pins the oracle: the reference's own results (VkFFT HIP backend on an MI355X) agree with the
    double-precision truth within the fp32/fp64 bands, and the oracle agrees with both."""
    mod, data = golden
    for case in mod.CASES:
        if case["name"] not in data:
            continue
        x = mod.golden_input(case)
        ref_out = data[case["name"]]
        dp = bool(case["dp"])
        if case["kind"] == 1:
            ref_c = ref_out.view(np.complex128 if dp else np.complex64)
        else:
            ref_c = ref_out
        truth = _truth_for_case(oracle, mod, case, x)
        tol = 1e-14 if dp else 3e-6
        if case.get("sample"):  # long results are stored as every n-th bin + the norm of the whole
            assert abs(np.linalg.norm(truth) / data[case["name"] + "__l2"][0] - 1) < 1e-6, case["name"]
            truth = truth[:: case["sample"]]
        assert rel_l2(ref_c, truth) < tol, ("reference vs truth", case["name"])
        if (case["kind"] == 1 and len(case["shape"]) > 1) or case.get("sample") or int(np.prod(case["shape"])) > 40000:
            continue  # (the scalar C restatement is pinned on the short cases; its multi-pass forms have their own tests)
        mine = _oracle_for_case(oracle, case, x)
        assert rel_l2(mine, truth) < tol, ("oracle vs truth", case["name"])
        assert rel_l2(mine, ref_c) < 2 * tol, ("oracle vs reference", case["name"])


def test_fftw_api_cross_check(oracle):
    """The reference's CPU path is FFTW; where MKL's FFTW3 interface is present it must agree with the oracle."""
    if not oracle.fftw_available():
        pytest.skip("no FFTW3 provider (libmkl_rt) on this box")
    rng = np.random.default_rng(3)
    for N, B in [(4096, 1), (1000, 4), (127, 2)]:
        x = (rng.uniform(-1, 1, N * B) + 1j * rng.uniform(-1, 1, N * B)).astype(np.complex64)
        y, _ = oracle.fftw_c2c(x, N, B)
        assert rel_l2(y, oracle.truth_c2c(x, (N,), B)) < 1e-6
        assert rel_l2(oracle.c2c(x, (N,), B), y) < 2e-6
