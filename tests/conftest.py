import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by `pytest -m gpu`)")


@pytest.fixture(scope="session")
def oracle():
    """CPU oracle (test infrastructure): C restatement + scipy ground truth."""
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def product_lib():
    """The real HIP library (C-ABI).  Built in-tree if missing; never substituted by anything else."""
    from vkfft_amd import api
    if not os.path.exists(api.lib_path()):
        subprocess.check_call(["make", "-s", "-C", ROOT])
    return api.load()


@pytest.fixture(scope="session")
def emu_lib():
    """CPU-emulated build of the same sources (tests/hostemu) — test double for host logic / index maps."""
    from vkfft_amd import api
    # one build at a time: pytest-xdist workers would otherwise compile into the same object files concurrently
    import fcntl
    os.makedirs(os.path.join(ROOT, "tests", "hostemu", "_build"), exist_ok=True)
    with open(os.path.join(ROOT, "tests", "hostemu", "_build", ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            subprocess.check_call([os.path.join(ROOT, "tests", "hostemu", "build.sh")])
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return api.load_test_double(os.path.join(ROOT, "tests", "hostemu", "_build", "libvkfft_hostemu.so"))


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_golden
    data = np.load(os.path.join(ROOT, "tests", "golden", "ref_vkfft.npz"))
    return make_golden, data
