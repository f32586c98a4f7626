"""Shared pytest set-up.

Markers: `gpu` = needs a real MI355X (run with `-m gpu` on the GPU box); everything else runs
on CPU. The CPU oracle (oracle/) is TEST INFRASTRUCTURE: it is built here on demand (gcc, a
few seconds) and is the checker in every parity test; the product library is never routed
through it.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X GPU")


def _make(target_dir, *args):
    subprocess.run(["make", "-C", os.path.join(REPO, target_dir), *args], check=True,
                   stdout=subprocess.DEVNULL)


@pytest.fixture(scope="session")
def oracle_c():
    from oracle import pyoracle
    if not pyoracle.available("c"):
        _make("oracle", "libtracer_oracle.so")
    o = pyoracle.Oracle("c")
    o.set_math_mode(pyoracle.Oracle.MATH_LIBM)
    return o


@pytest.fixture(scope="session")
def oracle_ref():
    """The reference-linked host build; only where it has been built (needs /root/reference)."""
    from oracle import pyoracle
    if not pyoracle.available("ref"):
        if os.path.isdir("/root/reference/src"):
            _make("oracle", "ref")
        if not pyoracle.available("ref"):
            pytest.skip("oracle/_ref/libtracer_ref.so not built (no /root/reference here)")
    return pyoracle.Oracle("ref")


@pytest.fixture(scope="session")
def product_lib():
    """libtracer_amd.so (hipcc-built, in-tree). Host-side entry points work without a GPU."""
    import tracer_amd
    if not os.path.exists(tracer_amd.lib_path()):
        _make(os.path.join("tracer_amd", "csrc"), "-j8")
    return tracer_amd.load_library()


@pytest.fixture(scope="session")
def golden():
    class G:
        def __getitem__(self, name):
            return np.load(os.path.join(GOLDEN, name + ".npz"))
    return G()


def rgb_channels(pbo):
    pbo = np.asarray(pbo, np.uint32)
    return np.stack([(pbo >> 16) & 255, (pbo >> 8) & 255, pbo & 255], -1).astype(np.int32)
