"""Barrier-hazard check: the CPU SIMT emulation normally runs the threads of a workgroup in ascending order between barriers,
which masks a missing barrier whenever the reader comes after the writer in that order.  Re-run the kernel-heavy part of the
suite with descending and pseudo-random thread orders (VKFFT_HOSTEMU_ORDER, tests/hostemu/hostemu_runtime.cpp): results must
not depend on the order."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("order", [1, 2])
def test_results_do_not_depend_on_the_thread_order(order):
    env = dict(os.environ, VKFFT_HOSTEMU_ORDER=str(order))
    sel = "opfft_table or mixed_radix_table or fourstep or bluestein or r2c or dct or random_plans or golden or cyclic_convolution or real_rows_between or mixed_radix or convolution or zero_padding or rader_stage or register_lean or every_registered_shape or maps_inside or two_real_rows or never_touches"
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_emu_parity.py"), os.path.join(ROOT, "tests", "test_emu_fuzz.py"), os.path.join(ROOT, "tests", "test_emu_convpad.py"),
                          "-x", "-q", "-m", "not gpu", "-k", sel, "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
