"""Accuracy of the portable sin/cos/acos/pow (oracle/pmath.h == tracer_amd/csrc/pmath.h
algorithm) against float64 references, and its measured effect on rendered radiance versus the
libm the reference calls. CPU only."""
import numpy as np
import pytest


def ulp_err(got, exact):
    got = np.asarray(got, np.float64); exact = np.asarray(exact, np.float64)
    ulp = np.spacing(np.abs(exact).astype(np.float32)).astype(np.float64)
    return np.abs(got - exact) / ulp


def test_sincos_accuracy(oracle_c):
    xs = np.concatenate([np.linspace(0, 2 * np.pi, 5001), [0.0, np.pi / 2, np.pi, 6.2831855]]).astype(np.float32)
    s = np.empty_like(xs); c = np.empty_like(xs)
    for i, x in enumerate(xs):
        s[i], c[i] = oracle_c.pm_sincos(float(x))
    xd = xs.astype(np.float64)
    # absolute error near the zeros of sin/cos (relative there is meaningless), ulp elsewhere
    big = np.abs(np.sin(xd)) > 1e-2
    assert ulp_err(s[big], np.sin(xd[big])).max() <= 2.0
    assert np.abs(s - np.sin(xd)).max() < 2e-7
    big = np.abs(np.cos(xd)) > 1e-2
    assert ulp_err(c[big], np.cos(xd[big])).max() <= 2.0
    assert np.abs(c - np.cos(xd)).max() < 2e-7


def test_acos_accuracy(oracle_c):
    xs = np.concatenate([np.linspace(-1, 1, 4001), [1.0, -1.0, 0.5, -0.5, 0.0]]).astype(np.float32)
    a = np.array([oracle_c.pm_acos(float(x)) for x in xs], np.float32)
    exact = np.arccos(xs.astype(np.float64))
    ok = exact > 1e-3
    assert ulp_err(a[ok], exact[ok]).max() <= 3.0
    assert np.abs(a - exact).max() < 5e-7
    assert float(oracle_c.pm_acos(1.0)) == 0.0


def test_gamma_pow_accuracy(oracle_c):
    xs = np.concatenate([np.geomspace(1e-6, 50, 3000), [0.0, 1.0, 0.5]]).astype(np.float32)
    g = np.float32(1.0 / 2.2)
    got = np.array([oracle_c.pm_pow(float(x), float(g)) for x in xs], np.float32)
    exact = np.power(xs.astype(np.float64), np.float64(g))
    nz = xs > 0
    assert ulp_err(got[nz], exact[nz]).max() <= 1.0
    assert float(oracle_c.pm_pow(0.0, float(g))) == 0.0
    assert np.isnan(oracle_c.pm_pow(-1.0, float(g)))


def test_portable_vs_libm_radiance(oracle_c):
    """Swapping glibc's sinf/cosf/acosf for the portable routines moves sampled directions by
    ~1e-7; per-sample radiance is a product of material colours, so it changes only where a
    path flips across a geometric edge. Measured here: the fraction of pixel-samples that
    differ by more than 1e-4 (the north-star tolerance) must stay below 1e-3."""
    from oracle.pyoracle import Oracle
    s = oracle_c.scene_create(0)
    w = h = 128
    cam = oracle_c.camera_default(w, h)
    bad = 0; total = 0
    try:
        for it in (1, 2, 3, 4):
            oracle_c.set_math_mode(Oracle.MATH_LIBM)
            a = oracle_c.pathtrace(s, w, h, cam, 21, it, 1, nthreads=4)["film"]
            oracle_c.set_math_mode(Oracle.MATH_PORTABLE)
            b = oracle_c.pathtrace(s, w, h, cam, 21, it, 1, nthreads=4)["film"]
            bad += int((np.abs(a - b).max(1) > 1e-4).sum()); total += w * h
    finally:
        oracle_c.set_math_mode(Oracle.MATH_LIBM)
    assert bad / total < 1e-3, (bad, total)


def test_division_shortcuts_match_ieee_operators():
    """div_by_uniform / udivmod of tracer_amd/csrc/pmath.h (used by the camera-ray and task-decoding
    code of the kernels) against plain `/` and `%`: every pixel coordinate of 30 image widths, 2^24
    random (width, coordinate, jitter) triples, 2^24 random + edge-case integer divisions."""
    import os, subprocess
    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cpp", "bin", "mathcheck")
    if not os.path.exists(exe):
        pytest.skip("tests/cpp/bin/mathcheck not built (run __graft_entry__.build())")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert " 0 mismatches; udivmod" in r.stdout and r.stdout.rstrip().endswith(" 0 mismatches")
