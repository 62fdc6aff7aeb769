"""CPU tests of the drop-in boundary: the real HIP library loads here (no GPU), exports every symbol that
include/vkFFT.h declares, its struct layout matches the FFI binding, error paths behave like the reference's
(vkFFT_InitializeApp.h:1468-1482, :428ff) and — having no CPU fallback — plan creation fails loudly without a GPU."""
import ctypes as C
import os
import re

import pytest

from vkfft_amd import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_exports_every_declared_symbol(product_lib):
    hdr = open(os.path.join(ROOT, "include", "vkFFT.h")).read()
    declared = re.findall(r"VKFFT_API\s+[\w\s\*]+?\b(\w+)\s*\(", hdr)
    assert set(declared) >= {"initializeVkFFT", "VkFFTAppend", "deleteVkFFT", "VkFFTGetVersion", "getVkFFTErrorString"}
    for name in declared:
        assert hasattr(product_lib, name), name
    assert set(declared) == set(api.EXPORTS)


def test_version_and_error_strings(product_lib):
    assert product_lib.VkFFTGetVersion() == 10304  # vkFFT.h:109
    assert product_lib.getVkFFTErrorString(0) == b"VKFFT_SUCCESS"
    assert product_lib.getVkFFTErrorString(3002) == b"VKFFT_ERROR_UNSUPPORTED_FFT_LENGTH"
    assert product_lib.getVkFFTErrorString(4039) == b"VKFFT_ERROR_FAILED_TO_LAUNCH_KERNEL"


def test_struct_sizes_match_binding(product_lib):
    sizes = (C.c_uint64 * 4)()
    product_lib.vkfftMI355XStructSizes(sizes)
    assert list(sizes) == [C.sizeof(api.VkFFTConfiguration), C.sizeof(api.VkFFTLaunchParams), C.sizeof(api.VkFFTPlan), C.sizeof(api.VkFFTApplication)]


def _init(lib, cfg, app=None):
    app = app if app is not None else api.VkFFTApplication()
    return lib.initializeVkFFT(C.byref(app), cfg), app


def test_validation_order_matches_reference(product_lib):
    lib = product_lib
    assert lib.initializeVkFFT(None, api.VkFFTConfiguration()) == 2015  # EMPTY_app
    dirty = api.VkFFTApplication(); dirty.actualNumBatches = 1
    assert _init(lib, api.VkFFTConfiguration(), dirty)[0] == 8          # NONZERO_APP_INITIALIZATION
    cfg = api.VkFFTConfiguration()
    assert _init(lib, cfg)[0] == 2001                                    # EMPTY_FFTdim
    cfg.FFTdim = 5
    assert _init(lib, cfg)[0] == 7                                       # FFTdim_GT_MAX
    cfg.FFTdim = 1
    assert _init(lib, cfg)[0] == 1002                                    # INVALID_DEVICE
    dev = C.c_int(0); cfg.device = C.pointer(dev)
    assert _init(lib, cfg)[0] == 2002                                    # EMPTY_size
    cfg.size[0] = 64; cfg.halfPrecision = 1
    assert _init(lib, cfg)[0] == 4                                       # out-of-scope feature rejected
    assert lib.VkFFTAppend(None, -1, None) == 2015


def test_no_cpu_fallback_without_gpu(product_lib):
    """On a box without a GPU plan creation must fail (no silent host path); on a GPU box it succeeds."""
    import torch
    cfg = api.VkFFTConfiguration(); cfg.FFTdim = 1; cfg.size[0] = 64
    dev = C.c_int(0); cfg.device = C.pointer(dev)
    rc, app = _init(product_lib, cfg)
    if torch.cuda.is_available():
        assert rc == 0
        product_lib.deleteVkFFT(C.byref(app))
    else:
        assert rc in (4051, 1002)  # FAILED_TO_GET_ATTRIBUTE / INVALID_DEVICE
        assert bytes(app) == bytes(C.sizeof(api.VkFFTApplication))  # app left zeroed, as the reference does on failure


def test_cli_driver_builds_and_prints_usage(product_lib):
    """tools/vkfft_cli.cpp (the counterpart of the reference's VkFFT_TestSuite flags) links the C-ABI and runs without a GPU
    as far as its usage text; every other mode needs a device and says so."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-s", "-C", root, "build/vkfft_mi355x_cli"])
    out = subprocess.run([os.path.join(root, "build", "vkfft_mi355x_cli"), "-h"], capture_output=True, text=True)
    assert out.returncode == 0 and "-benchmark_vkfft" in out.stdout and "-vkfft <id>" in out.stdout


def test_public_header_compiles_as_c99():
    """include/vkFFT.h is a C header like the reference's (callers may be plain C): compile a caller-side translation unit with
    gcc -std=c99 against it (HIP runtime API header from /opt/rocm)."""
    import subprocess, tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = ('#include "vkFFT.h"\n'
           'int main(void) { VkFFTConfiguration c = VKFFT_ZERO_INIT; VkFFTApplication a = VKFFT_ZERO_INIT; VkFFTLaunchParams l = VKFFT_ZERO_INIT;\n'
           '  (void)c; (void)a; (void)l; return VkFFTGetVersion() == 10304 ? 0 : 1; }\n')
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "caller.c"), "w").write(src)
        subprocess.check_call(["gcc", "-std=c99", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(root, "include"), "-I/opt/rocm/include",
                               "-c", os.path.join(d, "caller.c"), "-o", os.path.join(d, "caller.o")])


def test_environment_switches_are_documented():
    """every VKFFT_MI355X_* name the library reads (literal getenv calls in vkfft_amd/csrc) appears in INTEGRATION.md's switch table — by its full name or by the
    abbreviated `_SUFFIX` form the table uses inside a family — and the table names no switch the sources do not read"""
    import glob, re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = "".join(open(f, errors="replace").read() for f in glob.glob(os.path.join(root, "vkfft_amd", "csrc", "*")) if f.endswith((".cpp", ".hip", ".h")))
    read = set(re.findall(r'getenv\("(VKFFT_MI355X_[A-Z0-9_]+)"\)', src))
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    table = doc[doc.index("## Environment switches"):doc.index("## FFI stubs")]
    missing = [n for n in sorted(read) if n not in table and ("`" + n[len("VKFFT_MI355X"):]) not in table]
    assert not missing, missing
    named = set(re.findall(r"VKFFT_MI355X_[A-Z0-9_]+", table))
    built = {"VKFFT_MI355X_P2V", "VKFFT_MI355X_FUV", "VKFFT_MI355X_LIB"}  # (names assembled at run time: P2V<k>, FUV<k>; the Python stub's own)
    stale = [n for n in sorted(named) if n not in read and n not in built and not any(r.startswith(n) for r in read)]
    assert not stale, stale
