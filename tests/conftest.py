import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def _gpu_visible():
    """Without touching the GPU (and without importing torch): does this box have an AMD compute device node?"""
    return os.path.exists("/dev/kfd")


# The code objects hipRTC produces are cached on disk (launch_custom.hip; the GPU suite was paying 10 - 20 s per variant and process).  On
# a box without a GPU -- where the suite's job is to show that the run-time compilation itself works -- the cache is switched off unless a
# test sets its own directory, so every variant there really goes through hipRTC.
if not _gpu_visible():
    os.environ.setdefault("CDKF_RTC_CACHE", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def hip_lib():
    """The loaded HIP C-ABI library; building it if the .so is missing (cross-compiles without a GPU)."""
    from cd_dynamax_amd import _ffi
    if not os.path.exists(_ffi.LIB_PATH):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "cd_dynamax_amd", "csrc"), "-j8"])
    return _ffi.lib()
