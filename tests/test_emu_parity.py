"""CPU tests of the host logic (planner, pass descriptors, launch layer, API semantics) and of the kernels'
index maps: the product sources compiled against the CPU SIMT emulation (tests/hostemu) must reproduce the
oracle.  The real-GPU versions of these checks are in test_gpu_parity.py."""
import ctypes as C

import numpy as np
import pytest

import parity
from helpers import Runner, rel_l2
from vkfft_amd import api


@pytest.fixture(scope="module")
def run(emu_lib):
    return Runner(emu_lib, "emu")


@pytest.mark.parametrize("N", [2, 3, 4, 5, 7, 8, 11, 13, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384])
def test_pow2_and_small_radix_rows(run, oracle, N):
    parity.check_c2c(run, oracle, (N,), 3, False)


@pytest.mark.parametrize("N", [6, 9, 10, 12, 14, 15, 30, 100, 243, 343, 121, 169, 1000, 1080, 3125, 2401, 1331, 2197, 6561])
@pytest.mark.parametrize("dp", [False, True])
def test_mixed_radix(run, oracle, N, dp):
    parity.check_c2c(run, oracle, (N,), 2, dp)


def _mixed_table_sizes():
    import os, re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import glob
    txt = "".join(open(f).read() for f in sorted(glob.glob(os.path.join(root, "vkfft_amd", "csrc", "mixed_table_*.inc"))))
    return sorted(set(int(m) for m in re.findall(r"// N=(\d+)", txt)))


def test_every_mixed_radix_table_entry(run, oracle):
    """each ahead-of-time mixed-radix kernel instance (fp32 and fp64) against the double-precision truth"""
    for N in _mixed_table_sizes():
        for dp in (False, True):
            if dp and N > 4096:
                continue
            x = parity.seeded_complex(N * 2, dp, N)
            y, _ = run.transform(x, (N,), 2)
            e = rel_l2(y, oracle.truth_c2c(x, (N,), 2, longdouble=dp))
            assert e < (3e-15 if dp else 1e-6), (N, dp, e)


def _opfft_emu_cases():
    cases = parity.opfft_cases()
    return [c for i, c in enumerate(cases) if c[1] <= 128 or i % 7 == 0]


@pytest.mark.parametrize("chunk", range(8))
def test_opfft_table_entries(run, oracle, chunk):
    """Fused pre/post map kernels (R2C/C2R, DCT/DST, strided C2C): every entry up to L=128 and every 7th of the rest."""
    cases = _opfft_emu_cases()
    for fam, L, col, dp in cases[chunk::8]:
        parity.check_opfft_case(run, oracle, fam, L, col, dp, max_points=1 << 17)


@pytest.mark.parametrize("N", [17, 19, 23, 29, 31, 37, 41, 43, 47, 53, 59, 61, 17 * 16, 31 * 9])
def test_rader_direct_primes(run, oracle, N):
    parity.check_c2c(run, oracle, (N,), 2, False)


@pytest.mark.parametrize("N", [67, 71, 73, 89, 97, 127, 257, 641, 2311, 7681, 67 * 8, 4 * 97 * 3, 67 * 67])
@pytest.mark.parametrize("dp", [False, True])
def test_rader_fft_convolution_primes(run, oracle, N, dp):
    """primes whose P-1 is 13-smooth run as an FFT-convolution Rader stage inside the Stockham pass"""
    up = parity.check_c2c(run, oracle, (N,), 2, dp, kind="bluestein")
    assert up == [1] or (dp and N > 4096)  # fp64 rows longer than the one-pass capacity fall back to multi-pass Bluestein


@pytest.mark.parametrize("N", [83, 107, 251, 1009, 2039, 4093, 15319, 83 * 4])
@pytest.mark.parametrize("dp", [False, True])
def test_bluestein(run, oracle, N, dp):
    parity.check_c2c(run, oracle, (N,), 2, dp, kind="bluestein")


@pytest.mark.parametrize("N,dp,uploads,two_launches", [(4099, False, 1, 1), (8191, False, 1, 1), (4093, True, 1, 1), (8209, False, 3, 1), (15319, False, 3, 1), (21269, True, 3, 0), (524309, False, 5, 1),
                                                       (8209, False, 3, 0), (15319, False, 3, 0), (524309, False, 5, 0)])
def test_bluestein_multi_pass_fused(run, oracle, monkeypatch, N, dp, uploads, two_launches):
    """Rows whose padded length does not fit one pass.  fp32 (round 6): TWO launches of the fused Four-Step kernel with the chirp-z hooks on a registered padded length
    (kernel_mix_fused.h: 8209 -> 2^15, 15319 -> 30720 = 160 x 192, 524309 -> 1049760 = 972 x 1080); fp64 and VKFFT_MI355X_MIXFUSED=0: the 3 / 5 passes of
    pow2_col_blue_kernel on a power of two."""
    if two_launches:
        monkeypatch.setenv("VKFFT_MI355X_MIXFUSED_BLUE", "1")
    up = parity.check_c2c(run, oracle, (N,), 2 if N < 100000 else 1, dp, kind="bluestein", use_c_oracle=False)
    h, ptr = run._alloc(np.zeros(2 * N, np.complex128 if dp else np.complex64))
    app = api.App([N], 2, dp=dp, buffer_ptr=ptr, lib=run.lib)
    n, names = app.launch_info(False)
    app.delete()
    if two_launches and uploads > 1:
        assert up == [2] and n == 2 and names.startswith("mix_fused_kernel"), (up, n, names)
    else:
        assert up == [uploads] and n == uploads, (up, n, names)


def _largest_prime_with_padded_length(M):
    def smooth13(v):
        for q in (2, 3, 5, 7, 11, 13):
            while v % q == 0:
                v //= q
        return v == 1
    n = (M + 1) // 2
    while any(n % d == 0 for d in range(2, int(n ** 0.5) + 1)) or smooth13(n - 1):  # (a prime without a Rader form: p - 1 not 13-smooth)
        n -= 1
    return n


@pytest.mark.parametrize("M", [30720, 43008, 1 << 16, 1 << 17, 1 << 18, 1 << 19, 1 << 20, 1049760])
def test_chirp_z_in_two_fused_launches_every_padded_length(run, oracle, monkeypatch, M):
    """every registered padded length of the two-launch chirp-z plan (kernels_mixfused.hip, the instances with the hooks): the largest prime N with 2N - 1 <= M, forward
    against the double truth and the round trip (chirp and zero padding on the first launch's loads, FFT(chirp) on its stores, second chirp and the write mask on the
    second launch's stores)"""
    monkeypatch.setenv("VKFFT_MI355X_MIXFUSED_BLUE", "1")
    N = _largest_prime_with_padded_length(M)
    batch = 3 if M <= (1 << 17) else 1
    up = parity.check_c2c(run, oracle, (N,), batch, False, kind="bluestein", use_c_oracle=False)
    assert up == [2]
    h, ptr = run._alloc(np.zeros(batch * N, np.complex64))
    app = api.App([N], batch, buffer_ptr=ptr, lib=run.lib)
    n, names = app.launch_info(False)
    split = [int(app.app.localFFTPlan.contents.axisSplit[0][i]) for i in range(2)]
    app.delete()
    assert n == 2 and names.startswith("mix_fused_kernel") and split[0] * split[1] == M, (n, names, split)


@pytest.mark.parametrize("N,passes", [(1 << 15, 1), (1 << 16, 2), (1 << 18, 2), (3 ** 10, 2), (1 << 21, 2), (5 ** 9, 3)])
def test_fourstep(run, oracle, N, passes):
    up = parity.check_c2c(run, oracle, (N,), 1, False, use_c_oracle=N <= (1 << 16))
    assert up == [passes]


@pytest.mark.parametrize("shape,dp,passes", [((32, 32768), False, 2), ((8, 3 ** 10), True, 2), ((4, 4, 32768), False, 2), ((8, 4096), False, 2), ((8, 3000), False, 2), ((8, 2048), False, 1)])
def test_fourstep_along_strided_axis(run, oracle, shape, dp, passes):
    up = parity.check_c2c(run, oracle, shape, 1, dp, use_c_oracle=False)
    assert up[-1] == passes


def test_fourstep_fp64_and_batch(run, oracle):
    assert parity.check_c2c(run, oracle, (1 << 14,), 3, True) == [2]


@pytest.mark.parametrize("shape", [(16, 8), (64, 32), (100, 60), (12, 10, 6), (32, 16, 8), (8, 4, 2, 3)])
@pytest.mark.parametrize("dp", [False, True])
def test_multidim(run, oracle, shape, dp):
    parity.check_c2c(run, oracle, shape, 2, dp)


@pytest.mark.parametrize("shape", [(2,), (16,), (15,), (256,), (1000,), (243,), (64, 32), (30, 20, 10), (33, 8), (65536,), (2 * 3 ** 9, 3)])
@pytest.mark.parametrize("dp", [False, True])
def test_r2c_c2r(run, oracle, shape, dp):
    parity.check_r2c(run, oracle, shape, 2, dp)


@pytest.mark.parametrize("type", [1, 2, 3, 4])
@pytest.mark.parametrize("dst", [False, True])
@pytest.mark.parametrize("shape", [(8,), (9,), (64,), (81,), (32, 24), (12, 10, 6)])
def test_dct_dst(run, oracle, type, dst, shape):
    parity.check_r2r(run, oracle, shape, 2, False, type, dst)
    if shape in ((9,), (32, 24)):
        parity.check_r2r(run, oracle, shape, 2, True, type, dst)


@pytest.mark.parametrize("kind,shape,dp,type,dst", [("r2c", (11583,), False, 0, False), ("r2c", (18375,), True, 0, False), ("r2c", (9555, 4), False, 0, False),
                                                    ("r2r", (32768,), False, 2, False), ("r2r", (40000,), False, 3, False), ("r2r", (16385,), False, 1, False),
                                                    ("r2r", (16383,), True, 1, True), ("r2r", (32768,), False, 4, False), ("r2r", (30000,), True, 2, True)])
def test_real_transforms_longer_than_one_pass(run, oracle, kind, shape, dp, type, dst):
    """Odd R2C rows and DCT/DST whose embedding length exceeds one pass: the full-length form of the real transform through a
    multi-pass complex FFT, pre / post map applied to the row by natural index in the first load / last store."""
    if kind == "r2c":
        parity.check_r2c(run, oracle, shape, 2, dp)
    else:
        parity.check_r2r(run, oracle, shape, 2, dp, type, dst)


@pytest.mark.parametrize("shape,dp", [((5606,), False), ((916,), True), ((1217,), False), ((139, 12), False), ((2 * 2803, 6), False)])
def test_r2c_whose_half_length_needs_bluestein(run, oracle, shape, dp):
    """Real rows whose (half) length has a prime factor outside the radix / Rader stages (5606 = 2 * 2803): the full-length
    R2C / C2R maps around a fused Bluestein transform (kernel_blue_r2r.h)."""
    parity.check_r2c(run, oracle, shape, 3, dp)


@pytest.mark.parametrize("kind,shape,dp,type,dst", [("r2r", (64, 239), False, 2, False), ("r2r", (64, 239), True, 2, False), ("r2r", (10007,), False, 2, False),
                                                    ("r2r", (9001,), False, 4, False), ("r2r", (4999, 2), False, 1, True), ("r2r", (5, 239), False, 3, True),
                                                    ("r2r", (3, 239), True, 4, True), ("r2r", (24, 1451), False, 4, False), ("r2r", (7, 3, 241), False, 1, False),
                                                    ("r2c", (20011,), False, 0, False), ("r2c", (10007,), True, 0, False), ("r2c", (8209, 3), False, 0, False)])
def test_real_transforms_without_a_fused_bluestein_form(run, oracle, kind, shape, dp, type, dst):
    """Real transforms whose embedding length needs Bluestein along a STRIDED axis (DCT-II 64 x 239: the 239-point axis) or on more points than the
    fused Bluestein kernels hold (DCT-II 10007, DCT-IV 9001, DST-I 4999, R2C 20011): pre-map pass, complex plan of the embedding length on dense
    scratch rows, post-map pass (planner.cpp plan_real_by_maps; the reference: vkFFT_Scheduler.h:2271-2280, 2894-2944).  These lengths used to be
    rejected with VKFFT_ERROR_UNSUPPORTED_FFT_LENGTH_R2C / _R2R."""
    if kind == "r2c":
        parity.check_r2c(run, oracle, shape, 2, dp)
    else:
        parity.check_r2r(run, oracle, shape, 2, dp, type, dst)


@pytest.mark.parametrize("N,dp,type,dst", [(240, False, 1, False), (1014, False, 1, False), (478, False, 2, False), (478, False, 3, True), (239, False, 2, True),
                                           (240, True, 1, False), (718, True, 3, False), (1902, True, 4, False), (1451, False, 4, True), (879, False, 4, False)])
def test_r2r_whose_embedding_length_needs_bluestein(run, oracle, N, dp, type, dst):
    """DCT/DST whose embedding FFT length has a prime factor outside the radix / Rader stages (DCT-I of 240: 478 = 2 * 239):
    the real transform's maps around a fused Bluestein transform (kernel_blue_r2r.h)."""
    parity.check_r2r(run, oracle, (N,), 6, dp, type, dst)


@pytest.mark.parametrize("shape", [(53, 53), (41, 43, 5), (37, 37, 37), (8, 947), (33, 83), (16, 2, 257)])
@pytest.mark.parametrize("dp", [False, True])
def test_prime_planes_use_the_column_bluestein_kernel(run, oracle, shape, dp, monkeypatch):
    """strided axes of non-smooth length (the reference's sample-7 systems): one pass of pow2_col_blue_kernel (MODE 5) per axis (the Rader / smooth-length
    family of kernel_mixconv.h, which takes most of these since round 3, is switched off here: it has its own tests)"""
    monkeypatch.setenv("VKFFT_MI355X_MIXCONV", "0")
    up = parity.check_c2c(run, oracle, shape, 2, dp, kind="bluestein", use_c_oracle=False)
    assert up == [1] * len(shape)
    if dp and max(shape[1:]) > 512:
        return  # fp64 column tiles reach 1024 padded points: longer strided axes stay on the generic kernel
    buf = np.zeros(int(np.prod(shape)) * 2, np.complex128 if dp else np.complex64)
    app = api.App(list(shape), 2, dp=dp, buffer_ptr=buf.ctypes.data, lib=run.lib)
    names = C.create_string_buffer(1024)
    run.lib.vkfftMI355XDescribePlan(C.byref(app.app), 0, names, 1024)
    app.delete()
    assert "pow2_col_blue_kernel" in names.value.decode()


@pytest.mark.parametrize("shape,dp", [((8, 1087), False), ((37, 3, 1229), False), ((33, 1031, 2), True), ((20, 2909), False)])
def test_long_strided_axis_of_non_smooth_length_runs_as_transposed_rows(run, oracle, shape, dp):
    """padded length above the column Bluestein kernel's reach: transpose into scratch, fused Bluestein on unit-stride rows, transpose back"""
    up = parity.check_c2c(run, oracle, shape, 2, dp, kind="bluestein", use_c_oracle=False)
    assert up == [1] * len(shape)
    buf = np.zeros(int(np.prod(shape)) * 2, np.complex128 if dp else np.complex64)
    app = api.App(list(shape), 2, dp=dp, buffer_ptr=buf.ctypes.data, lib=run.lib)
    names = C.create_string_buffer(1024)
    run.lib.vkfftMI355XDescribePlan(C.byref(app.app), 0, names, 1024)
    app.delete()
    assert names.value.decode().count("transpose_kernel") == 2


@pytest.mark.parametrize("shape", [(17,), (31,), (17, 17), (23, 19, 5), (29, 29, 29)])
@pytest.mark.parametrize("dp", [False, True])
def test_small_primes_use_direct_butterflies(run, oracle, shape, dp):
    """17 .. 31 are radices of the mixed-radix row kernel and the strided op-FFT kernel (direct symmetric butterfly in registers)"""
    up = parity.check_c2c(run, oracle, shape, 70, dp, use_c_oracle=False)
    assert up == [1] * len(shape)
    buf = np.zeros(int(np.prod(shape)) * 2, np.complex128 if dp else np.complex64)
    app = api.App(list(shape), 2, dp=dp, buffer_ptr=buf.ctypes.data, lib=run.lib)
    names = C.create_string_buffer(1024)
    run.lib.vkfftMI355XDescribePlan(C.byref(app.app), 0, names, 1024)
    app.delete()
    got = names.value.decode()
    assert "blue" not in got and "mixed_row_kernel" in got and (len(shape) == 1 or "opfft_kernel" in got)


def test_golden_reference_fixtures(run, golden):
    """library (emulated) vs the reference's own outputs captured on an MI355X"""
    mod, data = golden
    for case in mod.CASES:
        if case["name"] not in data:
            continue
        if int(np.prod(case["shape"])) > (1 << 20):
            continue  # (2^22 takes the emulator minutes: the device test covers it)
        x = mod.golden_input(case)
        kw = {}
        if case["kind"] == 1:
            kw["r2c"] = True
        elif case["kind"] >= 21:
            kw["dst"] = case["kind"] - 20
        elif case["kind"] >= 11:
            kw["dct"] = case["kind"] - 10
        y, _ = run.transform(x, case["shape"], case["batch"], inverse=bool(case["inverse"]), **kw)
        ref = data[case["name"]]
        tol = 2e-14 if case["dp"] else 4e-6
        if case.get("sample"):  # long results are stored as every n-th bin + the norm of the whole
            assert abs(np.linalg.norm(y.astype(np.complex128)) / data[case["name"] + "__l2"][0] - 1) < 1e-6, case["name"]
            y = y[:: case["sample"]]
        if case["kind"] == 1:  # compare the Hermitian half only (padding lanes are the same memory)
            ct = np.complex128 if case["dp"] else np.complex64
            y, ref = y.view(ct), ref.view(ct)
        assert rel_l2(y, ref) < tol, case["name"]


def test_normalize_and_batch_folding(run, emu_lib):
    N, B = 64, 5
    x = parity.seeded_complex(N * B, False, 9)
    h = x.copy()
    app = api.App([N], B, buffer_ptr=h.ctypes.data, normalize=True, lib=emu_lib)
    # reference folds batches of a 1D plan into dimension 1 (vkFFT_Plan_FFT.h:55-61)
    assert app.app.actualNumBatches == B and app.app.configuration.numberBatches == 1
    assert app.app.localFFTPlan.contents.actualFFTSizePerAxis[0][1] == B
    app.forward(); app.inverse()
    assert rel_l2(h, x) < 1e-6
    app.delete()
    assert bytes(app.app) == bytes(C.sizeof(api.VkFFTApplication))  # deleteVkFFT zeroes the app (DeleteApp.h:322)


def test_direction_selection_and_plan_only_errors(emu_lib):
    x = parity.seeded_complex(32, False, 1); h = x.copy()
    app = api.App([32], 1, buffer_ptr=h.ctypes.data, makeForwardPlanOnly=1, lib=emu_lib)
    with pytest.raises(api.VkFFTError) as e:
        app.inverse()
    assert e.value.code == 1006  # ONLY_FORWARD_FFT_INITIALIZED
    # anything != 1 is forward (vkFFT_RunApp.h:102-111)
    lp = api.VkFFTLaunchParams()
    assert emu_lib.VkFFTAppend(C.byref(app.app), 0, C.byref(lp)) == 0
    assert rel_l2(h, np.fft.fft(x.astype(np.complex128))) < 1e-6
    app.delete()


def test_launch_time_buffer_override_and_missing_buffer(emu_lib):
    x = parity.seeded_complex(128, False, 2); h = x.copy()
    app = api.App([128], 1, buffer_ptr=0, lib=emu_lib)  # no buffer at plan time (UpdateBuffers.h:633-636)
    with pytest.raises(api.VkFFTError) as e:
        app.forward()
    assert e.value.code == 2004  # EMPTY_buffer
    app.forward(buffer_ptr=h.ctypes.data)
    assert rel_l2(h, np.fft.fft(x.astype(np.complex128))) < 1e-6
    app.delete()


def test_out_of_place_formatted_buffers(emu_lib):
    N, B = 100, 3
    x = parity.seeded_complex(N * B, False, 4)
    src = x.copy(); dst = np.zeros_like(x)
    app = api.App([N], B, buffer_ptr=dst.ctypes.data, isInputFormatted=1, inputBuffer=src.ctypes.data, lib=emu_lib)
    app.forward()
    assert np.array_equal(src, x)  # input untouched
    assert rel_l2(dst, np.fft.fft(x.astype(np.complex128).reshape(B, N), axis=1)) < 1e-6
    app.delete()


def test_user_temp_buffer_too_small(emu_lib):
    N = 1 << 16
    buf = np.zeros(N, np.complex64); tmp = np.zeros(16, np.complex64)
    with pytest.raises(api.VkFFTError) as e:
        api.App([N], 1, buffer_ptr=buf.ctypes.data, userTempBuffer=1, tempBuffer=tmp.ctypes.data, tempBufferSize=tmp.nbytes, lib=emu_lib)
    assert e.value.code == 2016  # INVALID_user_tempBuffer_too_small (Scheduler.h:2940-2942)


def test_disable_reorder_four_step_is_reported_back_as_not_applied(run, oracle):
    """the reference leaves a multi-upload result in an unspecified transposed order when disableReorderFourStep is set (vkFFT_InitializeApp.h:1312-1316); this
    library keeps natural order and tells the caller: the application's copy of the configuration reads 0 / reorderFourStep 1"""
    N = 1 << 16
    x = parity.seeded_complex(N, False, 5)
    h, ptr = run._alloc(x)
    app = api.App([N], 1, buffer_ptr=ptr, lib=run.lib, disableReorderFourStep=1)
    assert app.app.configuration.disableReorderFourStep == 0 and app.app.configuration.reorderFourStep == 1
    assert app.uploads() == [2]
    app.forward(); y = run._fetch(h, np.complex64); app.delete()
    assert rel_l2(y, oracle.truth_c2c(x, (N,), 1)) < 1e-6  # natural order


def test_unsupported_features_are_rejected(emu_lib):
    buf = np.zeros(64, np.complex64)
    for kw in (dict(performConvolution=1, matrixConvolution=9), dict(halfPrecision=1), dict(quadDoubleDoublePrecision=1), dict(bufferNum=2),
               dict(performZeropadding=[1, 0, 0, 0], fft_zeropad_left=[10, 0, 0, 0], fft_zeropad_right=[80, 0, 0, 0]),  # range outside the axis
               dict(performConvolution=1, numberKernels=2, numberBatches_override=2)):
        if "numberBatches_override" in kw:
            kw = dict(performConvolution=1, numberKernels=2)
            with pytest.raises(api.VkFFTError):
                api.App([64], 2, buffer_ptr=buf.ctypes.data, lib=emu_lib, **kw)
            continue
        with pytest.raises(api.VkFFTError):
            api.App([64], 1, buffer_ptr=buf.ctypes.data, lib=emu_lib, **kw)


def test_index_permutations_are_bit_exact(run):
    """Four-Step reorder / transposed stores are pure index permutations: a unit impulse at position p must
    produce exactly exp(-2 pi i p k / N) to rounding of the twiddles only, and a DC input must give an exact
    constant N at bin 0 and exact zeros elsewhere is not required - but *positions* must be exact."""
    N = 1 << 15
    for p in (0, 1, 777, N - 1):
        x = np.zeros(N, np.complex64); x[p] = 1.0
        y, _ = run.transform(x, (N,), 1)
        k = np.arange(N)
        ref = np.exp(-2j * np.pi * ((p * k) % N) / N)
        assert np.abs(y - ref).max() < 2e-6  # every output bin at its natural-order position


@pytest.mark.parametrize("N,batch,chunk_kib,lag,ring,queues", [(1 << 15, 7, 256, 2, 3, 1), (1 << 15, 7, 512, 1, 2, 1), (1 << 16, 5, 512, 2, 4, 1), (1 << 15, 9, 512, 3, 4, 1),
                                                               (1 << 17, 3, 1024, 1, 2, 1), (1 << 15, 37, 256, 2, 3, 8), (1 << 15, 21, 512, 1, 2, 4), (1 << 16, 19, 512, 3, 5, 8)])
def test_fused_fourstep_queue(run, oracle, monkeypatch, N, batch, chunk_kib, lag, ring, queues):
    """fused Four-Step (kernel_pow2_fused.h): ticket decode, per-XCD queues and queue helping, chunk ring reuse, partial last chunk and
    the reversed sweep of the inverse (the emulator drains the queues with one workgroup, in ticket order)"""
    monkeypatch.setenv("VKFFT_MI355X_FUSED_CHUNK_KIB", str(chunk_kib))
    monkeypatch.setenv("VKFFT_MI355X_FUSED_LAG", str(lag))
    monkeypatch.setenv("VKFFT_MI355X_FUSED_RING", str(ring))
    monkeypatch.setenv("VKFFT_MI355X_FUSED_QUEUES", str(queues))
    monkeypatch.setenv("VKFFT_MI355X_ROW15", "0")  # (2^15 runs as one pass of the register-lean row kernel since round 4: keep the two-pass plan here)
    x = parity.seeded_complex(N * batch, False, N + batch)
    y, z, up = run.transform(x, (N,), batch, both=True)
    assert up == [2]
    assert rel_l2(y, oracle.truth_c2c(x, (N,), batch)) < 1e-6
    assert rel_l2(z, x.astype(np.complex128) * N) < 2e-6


@pytest.mark.parametrize("N,batch", [(59049, 3), (177147, 2), (531441, 1), (78125, 3), (390625, 1), (117649, 2), (161051, 2), (1771561, 1), (28561, 3)])
def test_fused_fourstep_of_non_power_of_two_lengths(run, oracle, monkeypatch, N, batch):
    """fused Four-Step of two mixed-radix factors (kernel_mix_fused.h): every registered length of BASELINE config 3's powers of 3, 5, 7, 11 and 13 — partial last
    tiles of either phase (243 = 15 x 16 + 3 columns), phases with different tile counts, one launch per direction"""
    monkeypatch.setenv("VKFFT_MI355X_LONGROWS", "0")  # (11^4, 5^6, 7^5 run as ONE pass of the long mixed-radix rows by default: test_long_mixed_radix_rows_in_one_pass)
    x = parity.seeded_complex(N * batch, False, N + batch)
    y, z, up = run.transform(x, (N,), batch, both=True)
    assert up == [2]
    h, ptr = run._alloc(x)
    app = api.App([N], batch, buffer_ptr=ptr, lib=run.lib)
    n, names = app.launch_info(False)
    app.delete()
    assert n == 1 and names.startswith("mix_fused_kernel"), (n, names)
    truth = oracle.truth_c2c(x, (N,), batch)
    assert rel_l2(y, truth) < 1e-6
    assert rel_l2(z, x.astype(np.complex128) * N) < 2e-6


@pytest.mark.parametrize("N,batch", [(8232, 3), (9000, 3), (10000, 2), (10080, 3), (12000, 2), (12288, 3), (13125, 2), (14641, 3), (15000, 2), (15625, 2), (16128, 2), (16384 - 184, 2), (16807, 3)])
def test_long_mixed_radix_rows_in_one_pass(run, oracle, N, batch):
    """9000 ... 16000, 11^4, 5^6, 7^5: one pass of mixed_row_kernel with the whole row in one LDS buffer (mixed_table_6.inc; two butterflies per thread for 11^4, three for 7^5), where the
    generated table stops at 8192 points and the Four-Step plan takes two passes"""
    x = parity.seeded_complex(N * batch, False, N + batch)
    y, z, up = run.transform(x, (N,), batch, both=True)
    assert up == [1]
    h, ptr = run._alloc(x)
    app = api.App([N], batch, buffer_ptr=ptr, lib=run.lib)
    n, names = app.launch_info(False)
    app.delete()
    assert n == 1 and names.startswith("mixed_row_kernel"), (n, names)
    assert rel_l2(y, oracle.truth_c2c(x, (N,), batch)) < 1e-6
    assert rel_l2(z, x.astype(np.complex128) * N) < 2e-6


@pytest.mark.parametrize("kind,N,type,dst", [("r2c", 8400, 0, False), ("r2c", 16464, 0, False), ("r2c", 20000, 0, False), ("r2c", 30000, 0, False), ("r2c", 10125, 0, False),
                                             ("r2r", 10080, 2, False), ("r2r", 8400, 2, True), ("r2r", 12000, 3, False), ("r2r", 16200, 4, False), ("r2r", 10125, 4, True), ("r2r", 14406, 2, False)])
def test_real_transforms_on_the_long_rows(run, oracle, kind, N, type, dst):
    """real transforms whose complex length is one of the long rows (8193 ... 16807 points, tools/gen_long_rows_table.py): ONE launch of mixed_row_kernel between the
    table-driven maps where the real planners stopped at 8192 points and fell to the interpreter's multi-pass plans — R2C of even lengths on the half-length form (the long
    rows keep it: no full-length pairs), odd R2C and DCT / DST II-IV on the full-length forms"""
    if kind == "r2c":
        parity.check_r2c(run, oracle, (N,), 2, False)
        kw = dict(r2c=True)
    else:
        parity.check_r2r(run, oracle, (N,), 2, False, type, dst)
        kw = dict(dst=type) if dst else dict(dct=type)
    h, ptr = run._alloc(np.zeros(2 * (N + 2) * 2, np.float32))
    app = api.App([N], 2, buffer_ptr=ptr, lib=run.lib, **kw)
    n, names = app.launch_info(False)
    app.delete()
    assert n == 1 and names.startswith("mixed_row_kernel"), (n, names)


@pytest.mark.parametrize("N", [4116, 4200, 5040, 5625, 6300, 6561, 7203, 7560, 8000, 8064, 4158, 4225, 5005, 6006, 6591, 7007, 8190])
def test_fp64_rows_of_4097_to_8192_points_in_one_pass(run, oracle, N):
    """double precision: the two-buffer single-pass limit is 4096 points, a row of up to 8192 fits ONE LDS buffer — instances of mixed_row_kernel<double> for every 13-smooth length
    of that range (mixed_table_15 ... 19.inc) where the interpreter ran two passes; complex rows, and R2C / DCT-II rows whose complex length is one of them"""
    up = parity.check_c2c(run, oracle, (N,), 2, True, use_c_oracle=False)
    assert up == [1]
    h, ptr = run._alloc(np.zeros(4 * N, np.complex128))
    app = api.App([N], 2, dp=True, buffer_ptr=ptr, lib=run.lib)
    n, names = app.launch_info(False)
    app.delete()
    assert n == 1 and names.startswith("mixed_row_kernel<double>"), (n, names)
    if N in (5040, 6300):
        parity.check_r2c(run, oracle, (2 * N,), 2, True)
        parity.check_r2r(run, oracle, (N,), 2, True, 2, False)


def test_dst1_of_1782_reals_rader_stage_without_geometry(run, oracle):
    """regression (round 6): DST-I of 1782 reals embeds into 2 * 1783 complex points; 1783 is a Rader prime whose registry entry has no stage form (geometry 0), and
    mixrad_choose divided by its group count — SIGFPE at initializeVkFFT.  Found by a scan of plan creation over every transform kind and length up to 8300
    (no other length faults); the length now takes the fused Bluestein kernel"""
    parity.check_r2r(run, oracle, (1782,), 2, False, 1, True)


@pytest.mark.parametrize("kind", ["c2c", "r2c", "dct1", "dct2", "dct3", "dct4", "dst1", "dst2", "dst3", "dst4"])
def test_plan_creation_of_sampled_lengths_every_kind(run, kind):
    """initializeVkFFT + vkfftMI355XDescribePlan + deleteVkFFT for every 23rd length up to 8300 (and the neighbours of 1782 / 3566) of every transform kind: no plan may fault
    or fail (the full scan, every length, ran once in round 6: tools-free, ten processes, two minutes)"""
    kw = {"c2c": {}, "r2c": {"r2c": True}}.get(kind)
    if kw is None:
        kw = {kind[:3]: int(kind[3])}
    h, ptr = run._alloc(np.zeros(4 * 8400, np.float32))
    for N in sorted(set(list(range(2, 8300, 23)) + [1781, 1782, 1783, 3565, 3566, 3567])):
        app = api.App([N], 2, buffer_ptr=ptr, lib=run.lib, **kw)
        n, _ = app.launch_info(False)
        app.delete()
        assert n >= 1, (kind, N)


@pytest.mark.parametrize("N,batch,chunk_kib,lag,ring,queues,shape", [(59049, 7, 512, 2, 3, 1, 0), (59049, 11, 1024, 1, 2, 4, 0), (28561, 37, 256, 2, 3, 8, 0), (28561, 21, 128, 1, 2, 3, 0), (531441, 3, 4096, 1, 2, 1, 0), (177147, 5, 2048, 1, 2, 2, 0), (78125, 7, 512, 2, 3, 3, 0)])
def test_fused_fourstep_of_non_power_of_two_lengths_queue(run, oracle, monkeypatch, N, batch, chunk_kib, lag, ring, queues, shape):
    """the same kernel under forced chunk sizes, lags, rings and queue counts (ring slots reused, a partial last chunk, queues that are helped, the reversed sweep of
    the inverse), and against the separate passes it replaces (VKFFT_MI355X_MIXFUSED=0)"""
    monkeypatch.setenv("VKFFT_MI355X_FUSED_CHUNK_KIB", str(chunk_kib))
    monkeypatch.setenv("VKFFT_MI355X_FUSED_LAG", str(lag))
    monkeypatch.setenv("VKFFT_MI355X_FUSED_RING", str(ring))
    monkeypatch.setenv("VKFFT_MI355X_FUSED_QUEUES", str(queues))
    monkeypatch.setenv("VKFFT_MI355X_MXFV", str(shape))
    monkeypatch.setenv("VKFFT_MI355X_LONGROWS", "0")
    x = parity.seeded_complex(N * batch, False, N + batch)
    y, z, up = run.transform(x, (N,), batch, both=True)
    assert up == [2]
    assert rel_l2(y, oracle.truth_c2c(x, (N,), batch)) < 1e-6
    assert rel_l2(z, x.astype(np.complex128) * N) < 2e-6
    monkeypatch.setenv("VKFFT_MI355X_MIXFUSED", "0")
    y2, _ = run.transform(x, (N,), batch)
    assert rel_l2(y, y2.astype(np.complex128)) < 5e-7


@pytest.mark.parametrize("k,variant", [(k, v) for k in (9, 10, 11, 12) for v in range(2)] + [(k, v) for k in (13, 14, 15) for v in range(3)])
def test_register_lean_rows_every_variant(run, oracle, monkeypatch, k, variant):
    """kernel_pow2_lean.h: 32 points per thread, real / imaginary planes exchanged one after the other, in-place DIF butterflies, twiddles in chunks (with
    and without the prefetch across the exchange) — every registered shape of 2^13, 2^14 and the one-pass 2^15, next to the round-1 kernels they replace"""
    monkeypatch.setenv(f"VKFFT_MI355X_P2V{k}", str(variant))
    N = 1 << k
    B = (11 if variant == 0 else 3) if k == 15 else 3 if k >= 13 else 21  # (2^9 ... 2^12: several rows per workgroup, the last tile partly filled; 2^15 variant 0: persistent workgroups, several rows each)
    x = parity.seeded_complex(N * B, False, N + variant)
    y, z, up = run.transform(x, (N,), B, both=True)
    assert up == [1]
    truth = oracle.truth_c2c(x, (N,), B)
    assert rel_l2(y, truth) < 1e-6
    from helpers import assert_elementwise
    assert_elementwise(y, truth, "c2c", False, f"2^{k} variant {variant}")
    assert rel_l2(z, x.astype(np.complex128) * N) < 2e-6


@pytest.mark.parametrize("N", [1 << 10, 1 << 14, 1 << 15])
def test_register_lean_row_strides_padding_and_scale(run, oracle, N):
    """the packed row kernels behind the general plan features: padded batch stride, zero padding (read and write masks; 2^15: the pairs kernel has none and the
    padded plan takes the one-row kernel), normalised inverse"""
    rng = np.random.default_rng(7)
    pitch = N + 24
    buf = (rng.uniform(-1, 1, (3, pitch)) + 1j * rng.uniform(-1, 1, (3, pitch))).astype(np.complex64)
    h, ptr = run._alloc(buf.reshape(-1))
    app = api.App([N], 3, buffer_ptr=ptr, lib=run.lib, normalize=True, bufferStride=[pitch])
    app.forward(); y = run._fetch(h, np.complex64).reshape(3, pitch)
    app.inverse(); z = run._fetch(h, np.complex64).reshape(3, pitch); app.delete()
    assert rel_l2(y[:, :N], np.fft.fft(buf[:, :N].astype(np.complex128), axis=1)) < 1e-6
    assert np.array_equal(y[:, N:], buf[:, N:]) and np.array_equal(z[:, N:], buf[:, N:])  # the gap between rows is never touched
    assert rel_l2(z[:, :N], buf[:, :N]) < 2e-6
    import convpad
    assert convpad.zeropad_case(run, (N,), {0: (N // 2, N)}, batch=2) < 3e-6
    assert convpad.zeropad_case(run, (N,), {0: (N // 4 + 1, N // 2 + 3)}, batch=3) < 3e-6  # (an inner range with odd ends)


@pytest.mark.parametrize("k,variant,batch", [(15, 0, 5), (15, 1, 5), (16, 1, 3), (17, 1, 3), (18, 1, 3), (19, 1, 2), (20, 1, 2), (20, 2, 2), (21, 0, 1), (21, 1, 1), (22, 0, 1), (22, 1, 1)])
def test_fused_fourstep_every_registered_shape(run, oracle, monkeypatch, k, variant, batch):
    """every shape in the fused Four-Step registry besides the defaults the other tests run: index 0 = what ships (2^16 ... 2^20: the packed-pair
    software-pipelined form, kernel_pow2_fused_pk.h; 2^21 / 2^22: packed-pair tiles of two halves, kernel_pow2_fused_pkh.h), then the round-4 pipelined form,
    the round-2/3 shape and the register-lean plane-split form (2^21 / 2^22: the second orientation, the round-4 16-column shape, the round-3 shapes)"""
    monkeypatch.setenv(f"VKFFT_MI355X_FUV{k}", str(variant))
    monkeypatch.setenv("VKFFT_MI355X_ROW15", "0")
    if k <= 18:
        monkeypatch.setenv("VKFFT_MI355X_FUSED_CHUNK_KIB", str((8 << k) >> 10)); monkeypatch.setenv("VKFFT_MI355X_FUSED_LAG", "2")
        monkeypatch.setenv("VKFFT_MI355X_FUSED_RING", "3"); monkeypatch.setenv("VKFFT_MI355X_FUSED_QUEUES", "2")
    N = 1 << k
    x = parity.seeded_complex(N * batch, False, N + batch)
    y, z, up = run.transform(x, (N,), batch, both=True)
    assert up == [2]
    assert rel_l2(y, oracle.truth_c2c(x, (N,), batch)) < 1e-6
    assert rel_l2(z, x.astype(np.complex128) * N) < 2e-6


@pytest.mark.parametrize("N", [2 * 37, 3 * 41, 8 * 37, 7 * 127, 30 * 89, 32 * 101, 5 * 53, 4 * 61, 16 * 257, 9 * 113, 25 * 73, 21 * 43, 2 * 1297, 12 * 337, 64 * 37, 6 * 521,
                               29 * 97, 44 * 71, 104 * 37, 23 * 151, 19 * 97, 34 * 37, 62 * 61, 87 * 37, 110 * 37, 75 * 41, 98 * 37, 105 * 37, 37 * 37, 61 * 61, 53 * 53, 96 * 41, 94 * 37, 59 * 61])
def test_rader_stage_of_a_composite_length(run, oracle, monkeypatch, N):
    """kernel_mixrad.h: rows of M * P points, the Rader convolution of the prime P as a stage (cofactors below, equal to and above the thread groups of
    the prime's instance; one and two column steps in registers; odd radices as direct sums — 11, 13, the primes 17 ... 31, 15, 25, 49; P * P through the prime's
    convolution along the columns; 59 * 61 has no such plan — the cofactor is a prime above 55 — and must still be right through Bluestein); against the truth,
    against the Bluestein plan of the same length, a batch that leaves the last workgroup partly filled, and the inverse"""
    monkeypatch.setenv("VKFFT_MI355X_MIXRAD", "2")  # (every served length, also where the cost model prefers Bluestein)
    batch = 7
    x = parity.seeded_complex(N * batch, False, N)
    y, z, up = run.transform(x, (N,), batch, both=True)
    assert up == [1]
    truth = oracle.truth_c2c(x, (N,), batch)
    assert rel_l2(y, truth) < 3e-6, rel_l2(y, truth)
    assert rel_l2(z, x.astype(np.complex128) * N) < 6e-6
    monkeypatch.setenv("VKFFT_MI355X_MIXRAD", "0")
    yb, _ = run.transform(x, (N,), batch)
    assert rel_l2(y, yb) < 3e-6


def test_rader_stage_plan_is_taken(emu_lib):
    """the planner sends M * P rows to the Rader-stage kernel (one launch of the mixconv family with the composite parameter) and not through Bluestein"""
    buf = np.zeros(2670 * 4, np.complex64)
    a = api.App([2670], 4, buffer_ptr=buf.ctypes.data, lib=emu_lib)
    n, kern = a.launch_info(); a.delete()
    assert n == 1 and "mixrad" in kern, (n, kern)


@pytest.mark.parametrize("k,batch", [(14, 5), (15, 3), (16, 3), (17, 2), (18, 2), (19, 2), (20, 1)])
def test_fused_fourstep_fp64(run, oracle, monkeypatch, k, batch):
    """fp64 members of the fused Four-Step family (16-byte elements, 16-column tiles), several chunks"""
    monkeypatch.setenv("VKFFT_MI355X_FUSED_CHUNK_KIB", str((16 << k) >> 10))
    monkeypatch.setenv("VKFFT_MI355X_FUSED_LAG", "1")
    monkeypatch.setenv("VKFFT_MI355X_FUSED_RING", "2")
    N = 1 << k
    x = parity.seeded_complex(N * batch, True, 5 + k)
    y, z, up = run.transform(x, (N,), batch, both=True)
    assert up == [2]
    assert rel_l2(y, oracle.truth_c2c(x, (N,), batch, longdouble=True)) < 3e-15
    assert rel_l2(z, x * N) < 5e-15


@pytest.mark.parametrize("shape,dp", [((32, 16, 64), False), ((8, 256), False), ((16, 4, 128), True), ((4, 1 << 13), False)])
def test_column_kernel_with_64_bit_addresses(oracle, emu_lib, shape, dp):
    """pow2_col_kernel's wide-span form (tiles of 2 GiB and more: the z axis of a 1024^3 volume on one GPU) forced on small problems
    (VKFFT_MI355X_FORCE_BIGSPAN is read once per process: own subprocess); the last case runs the Four-Step passes of a strided 8192-point axis"""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = f"""
import sys, os, numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, 'tests'))
from vkfft_amd import api
import parity, helpers
from oracle import oracle as O
lib = api.load_test_double(os.path.join({root!r}, 'tests', 'hostemu', '_build', 'libvkfft_hostemu.so'))
run = helpers.Runner(lib, 'emu')
parity.check_c2c(run, O, {tuple(shape)!r}, 2, {dp!r}, use_c_oracle=False)
print('OK')
"""
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, VKFFT_MI355X_FORCE_BIGSPAN="1"), capture_output=True, text=True)
    assert out.returncode == 0 and "OK" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def _mixconv_cases():
    """(N, dp) over the row instances of kernel_mixconv.h: every 7th Rader prime and every 5th Bluestein ladder length (the longest prime row it serves)"""
    ent = [e for e in parity.mixconv_entries() if not e[2]]
    cases = []
    for dp in (False, True):
        primes = [v for d, r, c, v in ent if d == dp and r]
        ladder = [v for d, r, c, v in ent if d == dp and not r]
        cases += [(p, dp) for p in primes[::7] + primes[-1:]]
        cases += [(parity.mixconv_length_for(False, m), dp) for m in ladder[::5] + ladder[-1:] if parity.mixconv_length_for(False, m)]
    return cases


@pytest.mark.parametrize("chunk", range(4))
def test_one_kernel_cyclic_convolution_rows(run, oracle, chunk, monkeypatch):
    """Rader (prime p, transform length p-1) and Bluestein on a smooth padded length, forward and round trip, dense rows and a partial last tile"""
    monkeypatch.setenv("VKFFT_MI355X_MIXCONV", "2")  # always prefer the family: the cost model would give some of these lengths to the power-of-two kernel
    monkeypatch.setenv("VKFFT_MI355X_MIXRAD_PRIMES", "0")  # (the instances of kernel_mixconv.h themselves: a prime\'s rows otherwise run on kernel_mixrad.h with the tables in LDS)
    for N, dp in _mixconv_cases()[chunk::4]:
        batch = 3 if N > 512 else 37
        x = parity.seeded_complex(N * batch, dp, N)
        y, z, _ = run.transform(x, (N,), batch, both=True)
        e = rel_l2(y, oracle.truth_c2c(x, (N,), batch, longdouble=dp))
        assert e < (3e-15 if dp else 1.5e-6), (N, dp, e)
        e2 = rel_l2(z, x.astype(np.complex128) * N)
        assert e2 < (6e-15 if dp else 3e-6), (N, dp, e2)


def test_one_kernel_cyclic_convolution_is_chosen(run, monkeypatch):
    """the plan of a Rader prime / of a length just above a power of two is ONE launch of the family (not the interpreter, not the power-of-two kernel)"""
    monkeypatch.setenv("VKFFT_MI355X_MIXCONV", "2")
    monkeypatch.setenv("VKFFT_MI355X_MIXRAD_PRIMES", "0")  # (the instances of kernel_mixconv.h themselves: a prime\'s rows otherwise run on kernel_mixrad.h with the tables in LDS)
    for N in (257, 8191, 1046, 47):
        x = parity.seeded_complex(N * 2, False, N)
        h, ptr = run._alloc(x)
        app = api.App([N], 2, buffer_ptr=ptr, lib=run.lib)
        try:
            n, names = app.launch_info()
            assert n == 1 and "mixconv" in names, (N, n, names)
        finally:
            app.delete()


@pytest.mark.parametrize("shape", [(37, 37), (6, 73), (40, 47), (3, 547), (5, 1009), (12, 257, 3), (67, 67, 2), (37, 41, 3), (33, 37, 37)])
@pytest.mark.parametrize("dp", [False, True])
def test_one_kernel_cyclic_convolution_columns(run, oracle, shape, dp, monkeypatch):
    """strided axes: tiles of neighbouring columns (Rader primes 37, 73, 257, 547, 1009; Bluestein on a smooth length for 47), partial tiles"""
    monkeypatch.setenv("VKFFT_MI355X_MIXCONV", "2")
    monkeypatch.setenv("VKFFT_MI355X_MIXRAD_PRIMES", "0")  # (the instances of kernel_mixconv.h themselves: a prime\'s rows otherwise run on kernel_mixrad.h with the tables in LDS)
    parity.check_c2c(run, oracle, shape, 2, dp, kind="bluestein")


@pytest.mark.parametrize("N", [6, 14, 28, 37, 46, 61, 74, 90, 112, 175, 242, 1001, 1046 // 2 * 2 - 2, 2 * 257])
@pytest.mark.parametrize("dp", [False, True])
def test_real_rows_between_the_interpreters_maps_and_an_instance_transform(run, oracle, N, dp):
    """R2C / C2R and DCT / DST I-IV of lengths whose complex transform has a mixed-radix or Rader instance but no fused-map kernel: the ahead-of-time
    transform runs between the interpreter's gather-load and gather-store (mixed_row_kernel / mixconv_kernel OPS = 1), one launch"""
    parity.check_r2c(run, oracle, (N,), 5, dp)
    for type in (1, 2, 3, 4):
        parity.check_r2r(run, oracle, (N,), 3, dp, type, False)
    parity.check_r2r(run, oracle, (N,), 3, dp, 2, True)
    parity.check_r2r(run, oracle, (N,), 3, dp, 4, True)


@pytest.mark.parametrize("shape", [(84, 84), (42, 90), (28, 36, 10), (75, 45)])
def test_real_planes_of_smooth_lengths_outside_the_curated_list(run, oracle, shape):
    """R2C and DCT / DST planes whose axis lengths got fused-map instances late in round 3 (rows of every short 7-smooth length, strided DCT axes up to 256 reals)"""
    parity.check_r2c(run, oracle, shape, 2, False)
    for type in (2, 3, 4):
        parity.check_r2r(run, oracle, shape, 2, False, type, False)
    parity.check_r2r(run, oracle, shape, 2, False, 2, True)


@pytest.mark.parametrize("kind,N", [("r2c", 265), ("r2c", 328), ("r2c", 148), ("dct2", 265), ("dct2", 148), ("dct3", 111), ("dct4", 74), ("r2c", 2 * 1010), ("dct2", 889), ("dst2", 185)])
def test_real_rows_whose_complex_length_has_a_rader_stage(run, oracle, monkeypatch, kind, N):
    """R2C / C2R / DCT / DST rows whose complex transform length is M * P (a Rader prime and a small cofactor): the Rader-stage kernel between the generic
    pre- and post-maps (kernel_mixrad.h, `ops`), against the oracle and against the plan without it"""
    if kind == "r2c":
        parity.check_r2c(run, oracle, (N,), 5, False)
    else:
        parity.check_r2r(run, oracle, (N,), 5, False, int(kind[3]), kind.startswith("dst"))


@pytest.mark.parametrize("N,dp", [(169, False), (385, False), (286, False), (1001, False), (2 * 1573, False), (169, True), (286, True), (33 * 13, True)])
def test_real_rows_with_the_maps_inside_the_stages(run, oracle, N, dp):
    """R2C / C2R rows whose complex transform runs on a mixed-radix instance with eight or more threads per row: the plain load / store maps (packed complex
    side of the even split; full-length real forms of odd lengths: real in, N/2 + 1 bins out, Hermitian half in, real parts out) run inside the first /
    last stage instead of a staging pass (kernel_mixed.h, `DIRECT`); several rows so that the last workgroup is partly filled"""
    parity.check_r2c(run, oracle, (N,), 7, dp)


@pytest.mark.parametrize("shape,pad", [((2670,), 6), ((296, 5), 3), ((74, 4, 3), 1), ((889,), 0)])
def test_rader_stage_rows_with_padded_pitch_out_of_place_and_as_an_axis(emu_lib, monkeypatch, shape, pad):
    """the Rader-stage kernel as axis 0 of 1-D ... 3-D plans: padded row pitch, a separate formatted input buffer that must stay untouched, forward and the
    normalised inverse back into place"""
    monkeypatch.setenv("VKFFT_MI355X_MIXRAD", "2")
    nd = len(shape); B = 3
    pitches, acc = [], 1
    for s in shape:
        acc = acc * s + pad
        pitches.append(acc)
    total = pitches[-1] * B
    rng = np.random.default_rng(sum(shape))
    src = (rng.uniform(-1, 1, total) + 1j * rng.uniform(-1, 1, total)).astype(np.complex64)
    orig = src.copy()
    strides = [pitches[-1]] + [pitches[i - 1] if i > 0 else 1 for i in range(nd - 1, -1, -1)]
    idx = np.indices([B] + list(shape)[::-1]).reshape(nd + 1, -1)
    off = sum(idx[d] * strides[d] for d in range(nd + 1))
    view = lambda a: a[off].reshape([B] + list(shape)[::-1]).astype(np.complex128)
    dst = np.zeros(total, np.complex64)
    app = api.App(list(shape), B, buffer_ptr=dst.ctypes.data, isInputFormatted=1, inputBuffer=src.ctypes.data, inputBufferStride=pitches + [0] * (4 - nd),
                  bufferStride=pitches + [0] * (4 - nd), normalize=True, lib=emu_lib)
    app.forward()
    assert rel_l2(view(dst), np.fft.fftn(view(orig), axes=tuple(range(1, nd + 1)))) < 5e-6
    assert np.array_equal(src, orig)
    app.inverse()
    app.delete()
    assert rel_l2(view(dst), view(orig)) < 8e-6


ODD_DCT4 = [((3,), 4), ((5,), 4), ((9,), 3), ((15,), 3), ((45,), 3), ((105,), 2), ((37,), 3), ((47,), 3), ((111,), 3), ((1125,), 2), ((1451,), 2), ((243,), 2),
            ((24, 45), 1), ((24, 239), 1), ((35, 7, 3), 1), ((19683,), 1), ((10007,), 1)]


@pytest.mark.parametrize("shape,b", ODD_DCT4)
@pytest.mark.parametrize("dp", [False, True])
def test_dct4_dst4_of_odd_length_in_the_same_length_form(run, oracle, shape, b, dp):
    """DCT-IV / DST-IV of odd length: signed permutation -> N-point transform -> one rotation by a multiple of pi/4 per output (vkFFT_R2R.h:414-481,
    922-972, 1032) on every path that can carry it: fused-map instances, instance transforms between the generic maps (mixed-radix, Rader, Rader stage),
    fused Bluestein (47, 1451: 4096 points, not 8192), strided axes, maps as passes (239 strided, 10007), several passes (3^9)"""
    if dp and shape in ((19683,), (10007,)):
        pytest.skip("long rows: fp32 only here (emulator time)")
    parity.check_r2r(run, oracle, shape, b, dp, 4, False)
    parity.check_r2r(run, oracle, shape, b, dp, 4, True)


def test_dct4_of_an_odd_prime_length_pads_to_the_same_length_bluestein(run):
    """1451 reals: the fused Bluestein kernel on 4096 points (2 * 1451 - 1 <= 4096), one launch"""
    x = np.zeros(1451 * 2, dtype=np.float32)
    h, ptr = run._alloc(x)
    app = api.App([1451], 2, buffer_ptr=ptr, lib=run.lib, dct=4)
    try:
        n, names = app.launch_info()
        assert n == 1 and "pow2_blue_r2r" in names, (n, names)
    finally:
        app.delete()


@pytest.mark.parametrize("N", [13, 31, 55, 91, 169, 385, 37, 61, 127, 111, 205])
@pytest.mark.parametrize("batch", [1, 2, 7])
def test_two_real_rows_per_transform(run, oracle, monkeypatch, N, batch):
    """PassParams::pairRows (the reference's mergeSequencesR2C, vkFFT_SharedMemory.h:40): rows of the full-length real forms travel two per complex transform
    between the generic maps; odd row counts; the same plans with one row per transform agree"""
    parity.check_r2c(run, oracle, (N,), batch, False)
    for type, dst in [(1, False), (2, False), (3, False), (4, False), (2, True), (3, True), (4, True), (1, True)]:
        parity.check_r2r(run, oracle, (N,), batch, False, type, dst)
    rng = np.random.default_rng(N)
    x = rng.uniform(-1, 1, N * batch).astype(np.float32)
    a = run.transform(x, (N,), batch, both=True, dct=3)
    monkeypatch.setenv("VKFFT_MI355X_NO_ROW_PAIRS", "1")
    b = run.transform(x, (N,), batch, both=True, dct=3)
    assert rel_l2(a[0], b[0]) < 1e-6 and rel_l2(a[1], b[1]) < 1e-6
    if batch > 1 and N in (169, 37, 111):
        assert not np.array_equal(a[0], b[0])  # (the paired plan really is another computation)


@pytest.mark.parametrize("N", [9, 15, 25, 45, 105])
@pytest.mark.parametrize("dp", [False, True])
def test_two_real_rows_per_transform_preferred_over_a_fused_map_instance(run, oracle, monkeypatch, N, dp):
    monkeypatch.setenv("VKFFT_MI355X_PAIR_PREFER", "1")
    parity.check_r2c(run, oracle, (N,), 5, dp)
    for type, dst in [(2, False), (3, False), (4, False), (4, True)]:
        parity.check_r2r(run, oracle, (N,), 5, dp, type, dst)


@pytest.mark.parametrize("shape,pads", [((45,), {0: (20, 45)}), ((91,), {0: (40, 91)})])
def test_two_real_rows_per_transform_with_zero_padding(run, shape, pads):
    import convpad
    res = convpad.zeropad_semantics_case(run, shape, pads, r2c=True)
    for k, v in res.items():
        assert (v is True) if isinstance(v, bool) else v < 3e-6, (k, res)


@pytest.mark.parametrize("N", [13, 55, 169, 37, 111, 28])
def test_two_real_rows_per_transform_against_one_row_per_transform(run, monkeypatch, N):
    """the body of the device test with the chip full, on an odd number of rows: paired plan against the plan with one row per transform (R2C, DCT-II / -III / -IV)"""
    batch = 33
    rng = np.random.default_rng(N)
    x = rng.uniform(-1, 1, N * batch).astype(np.float32)
    kinds = [dict(dct=2), dict(dct=3)] + ([dict(dct=4)] if N % 2 else [])
    if N % 2:
        buf = np.zeros((batch, 2 * (N // 2 + 1)), np.float32); buf[:, :N] = x.reshape(batch, N)
        kinds.append(dict(r2c=True))
    for kw in kinds:
        data = buf.reshape(-1) if kw.get("r2c") else x
        monkeypatch.delenv("VKFFT_MI355X_NO_ROW_PAIRS", raising=False)
        a = run.transform(data, (N,), batch, both=True, **kw)
        monkeypatch.setenv("VKFFT_MI355X_NO_ROW_PAIRS", "1")
        b = run.transform(data, (N,), batch, both=True, **kw)
        monkeypatch.delenv("VKFFT_MI355X_NO_ROW_PAIRS", raising=False)
        fa, fb = a[0].astype(np.float64), b[0].astype(np.float64)
        if kw.get("r2c"):
            fa = fa.reshape(batch, -1); fb = fb.reshape(batch, -1)
            assert rel_l2(a[1].reshape(batch, -1)[:, :N], b[1].reshape(batch, -1)[:, :N]) < 1e-6, (N, kw)
        else:
            assert rel_l2(a[1], b[1]) < 1e-6, (N, kw)
        assert rel_l2(fa, fb) < 1e-6, (N, kw)


@pytest.mark.parametrize("N", [13, 28, 55, 100, 169, 286, 385])
@pytest.mark.parametrize("dp", [False, True])
def test_table_driven_maps_of_the_real_transforms(run, oracle, monkeypatch, N, dp):
    """kernel_tmaps.h: every family the planner builds tables for (R2C / C2R, DCT / DST I-IV), two rows per transform and an odd row count, against the oracle and
    against the generic maps of the same plans (VKFFT_MI355X_NO_TMAPS)"""
    batch = 5
    parity.check_r2c(run, oracle, (N,), batch, dp)
    for type, dst in [(1, False), (2, False), (3, False), (4, False), (1, True), (2, True), (3, True), (4, True)]:
        parity.check_r2r(run, oracle, (N,), batch, dp, type, dst)
    rng = np.random.default_rng(N)
    x = rng.uniform(-1, 1, N * batch).astype(np.float64 if dp else np.float32)
    for kw in (dict(dct=2), dict(dct=3), dict(dct=4), dict(dst=2), dict(dst=3), dict(dst=4), dict(dct=1), dict(dst=1)):
        monkeypatch.delenv("VKFFT_MI355X_NO_TMAPS", raising=False)
        a = run.transform(x, (N,), batch, both=True, **kw)
        monkeypatch.setenv("VKFFT_MI355X_NO_TMAPS", "1")
        b = run.transform(x, (N,), batch, both=True, **kw)
        tol = 1e-13 if dp else 2e-6
        assert rel_l2(a[0], b[0]) < tol and rel_l2(a[1], b[1]) < tol, (kw, rel_l2(a[0], b[0]), rel_l2(a[1], b[1]))


@pytest.mark.parametrize("N", [283, 298, 235])
def test_two_real_rows_per_fused_bluestein_transform(run, oracle, monkeypatch, N):
    """kernel_blue_r2r.h with PassParams::pairRows: rows whose embedding length needs Bluestein travel two per transform (real sequences through the even / odd
    split, real results as real and imaginary part); odd row count; against the oracle and against the plans with one row per transform"""
    batch = 5
    parity.check_r2c(run, oracle, (N,), batch, False)
    for type, dst in [(1, False), (2, False), (3, False), (4, False), (1, True), (2, True), (3, True), (4, True)]:
        parity.check_r2r(run, oracle, (N,), batch, False, type, dst)
    rng = np.random.default_rng(N)
    x = rng.uniform(-1, 1, N * batch).astype(np.float32)
    for kw in (dict(dct=2), dict(dct=3), dict(dct=4)):
        monkeypatch.delenv("VKFFT_MI355X_NO_BLUE_PAIRS", raising=False)
        a = run.transform(x, (N,), batch, both=True, **kw)
        monkeypatch.setenv("VKFFT_MI355X_NO_BLUE_PAIRS", "1")
        b = run.transform(x, (N,), batch, both=True, **kw)
        assert rel_l2(a[0], b[0]) < 3e-6 and rel_l2(a[1], b[1]) < 3e-6, (kw, rel_l2(a[0], b[0]), rel_l2(a[1], b[1]))


@pytest.mark.parametrize("N", [265, 355, 148, 316])
def test_table_driven_maps_around_the_rader_stage_kernel(run, oracle, monkeypatch, N):
    """kernel_mixrad.h between the table-driven maps (kernel_tmaps.h tm_rows_in / tm_rows_out): every family, against the oracle and against the generic maps"""
    batch = 5
    parity.check_r2c(run, oracle, (N,), batch, False)
    for type, dst in [(2, False), (3, False), (4, False), (2, True), (3, True), (4, True)]:
        parity.check_r2r(run, oracle, (N,), batch, False, type, dst)
    rng = np.random.default_rng(N)
    x = rng.uniform(-1, 1, N * batch).astype(np.float32)
    for kw in (dict(dct=2), dict(dct=3), dict(dct=4)):
        monkeypatch.delenv("VKFFT_MI355X_NO_MIXRAD_TMAPS", raising=False)
        a = run.transform(x, (N,), batch, both=True, **kw)
        monkeypatch.setenv("VKFFT_MI355X_NO_MIXRAD_TMAPS", "1")
        b = run.transform(x, (N,), batch, both=True, **kw)
        assert rel_l2(a[0], b[0]) < 3e-6 and rel_l2(a[1], b[1]) < 3e-6, (kw, rel_l2(a[0], b[0]), rel_l2(a[1], b[1]))


@pytest.mark.parametrize("N,pitch", [(13, 15), (13, 40), (55, 57), (31, 33)])
def test_staged_real_rows_leave_the_gap_between_rows_alone(run, N, pitch):
    """kernel_mixed.h staging tile (fewer than eight threads per row): rows in a pitch longer than the row — the tile is still copied in as one run, but the results go
    out scalar by scalar with the gaps left out (or, beyond N + 2 scalars of pitch, through the maps directly); DCT-II and its inverse, in place"""
    import scipy.fft as sf
    rng = np.random.default_rng(N + pitch)
    batch = 5
    buf = rng.uniform(-1, 1, (batch, pitch)).astype(np.float32)
    h, ptr = run._alloc(buf.reshape(-1))
    app = api.App([N], batch, buffer_ptr=ptr, lib=run.lib, dct=2, bufferStride=[pitch])
    app.forward(); y = run._fetch(h, np.float32).reshape(batch, pitch)
    app.inverse(); z = run._fetch(h, np.float32).reshape(batch, pitch); app.delete()
    assert rel_l2(y[:, :N], sf.dct(buf[:, :N].astype(np.float64), type=2, axis=1)) < 3e-6
    assert np.array_equal(y[:, N:], buf[:, N:]) and np.array_equal(z[:, N:], buf[:, N:])
    assert rel_l2(z[:, :N], buf[:, :N].astype(np.float64) * 2 * N) < 5e-6
