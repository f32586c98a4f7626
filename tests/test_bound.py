"""Host check of tracer_amd/csrc/bound.h (no GPU): the world-space bounding spheres of the (ray, object) pre-test and the
mesh hulls, on random transforms -- tests/cpp/bound_check.cpp. The GPU side of the claim (the pre-test never rejects a pair
that the exact tests accept) is tests/test_gpu_pairs.py."""
import os
import subprocess

import pytest


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_bounding_spheres_hold_the_objects(seed):
    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cpp", "bin", "bound_check")
    if not os.path.exists(exe):
        pytest.skip("tests/cpp/bin/bound_check not built (run __graft_entry__.build())")
    r = subprocess.run([exe, "20000", str(seed)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.startswith("ok:"), r.stdout + r.stderr
