"""Convolution / zero-padding configurations on the CPU emulator build of the product sources."""
import pytest
import convpad
from helpers import Runner


@pytest.fixture(scope="module")
def run(emu_lib):
    return Runner(emu_lib, "emu")


@pytest.mark.parametrize("case", convpad.CONV_CASES, ids=lambda c: "x".join(map(str, c["shape"])) + "".join(f"-{k}{v}" for k, v in c.items() if k != "shape"))
def test_convolution(run, case):
    c = dict(case); shape = c.pop("shape")
    err = convpad.conv_case(run, shape, **c)
    assert err < (1e-13 if c.get("dp") else 3e-5), err


@pytest.mark.parametrize("case", convpad.ZEROPAD_CASES, ids=lambda c: "x".join(map(str, c["shape"])) + "".join(f"-{k}" for k in c if k not in ("shape", "pads")))
def test_zero_padding(run, case):
    c = dict(case); shape = c.pop("shape"); pads = c.pop("pads")
    err = convpad.zeropad_case(run, shape, pads, **c)
    assert err < (1e-13 if c.get("dp") else 3e-6), err


@pytest.mark.parametrize("case", convpad.CONV_ZEROPAD_CASES, ids=lambda c: "x".join(map(str, c["shape"])) + f"-m{c['m']}")
def test_convolution_of_zero_padded_systems(run, case):
    c = dict(case); shape = c.pop("shape"); pads = c.pop("pads")
    err = convpad.conv_zeropad_case(run, shape, pads, **c)
    assert err < (1e-13 if c.get("dp") else 3e-5), err


@pytest.mark.parametrize("case", convpad.ZEROPAD_SEMANTICS_CASES, ids=lambda c: "x".join(map(str, c["shape"])) + "".join(f"-{k}" for k in c if k not in ("shape", "pads")))
def test_zero_padding_never_touches_what_it_skips(run, case):
    """the caller's padded input range is not read (nor written: a separate input buffer stays bit-identical), the inverse leaves the padded range of
    its result alone, sequences inside the padded range of a later axis are not visited"""
    c = dict(case); shape = c.pop("shape"); pads = c.pop("pads")
    res = convpad.zeropad_semantics_case(run, shape, pads, **c)
    tol = 1e-13 if c.get("dp") else 3e-6
    for k, v in res.items():
        if isinstance(v, bool):
            assert v, (k, res)
        else:
            assert v < tol, (k, res)
