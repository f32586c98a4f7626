"""The claim of tracer_amd/csrc/sparse_xform.h, checked in IEEE float32 arithmetic on the CPU: leaving out the
terms of exactly-zero matrix entries gives the bits of the full left-to-right sum of type_mat4x4.inl:695-699
whenever each row keeps one entry with 2^-60 <= |m| < 2^60 and every coordinate is in [2^-60, 2^60) --
whatever the signs of the zeros, with zero translations, with sums that cancel. (The kernels' use of it is
covered on the GPU by test_axis_aligned_objects_every_pipeline.)"""
import numpy as np

F = np.float32
PATTERNS = {   # which (col, row) entries of the 3x3 part are zero
    "diag": lambda c, r: c != r,
    "block_x": lambda c, r: (c == 0) != (r == 0),
    "block_y": lambda c, r: (c == 1) != (r == 1),
    "block_z": lambda c, r: (c == 2) != (r == 2),
}


def full_row(m, d, v):          # ((m0*x + m1*y) + m2*z) + d
    return F(F(F(F(m[0] * v[0]) + F(m[1] * v[1])) + F(m[2] * v[2])) + d)


def sparse_row(m, d, v, keep):
    s = None
    for k in range(3):
        if keep[k]:
            t = F(m[k] * v[k])
            s = t if s is None else F(s + t)
    return F(s + d)


def test_skipping_zero_entries_keeps_every_bit():
    r = np.random.default_rng(7)
    zeros = np.array([0.0, -0.0], np.float32)
    checked = 0
    with np.errstate(over="ignore", under="ignore"):
        for name, is_zero in PATTERNS.items():
            for _ in range(4000):
                row = int(r.integers(3))
                keep = [not is_zero(c, row) for c in range(3)]
                m = np.zeros(3, np.float32)
                for c in range(3):
                    if keep[c]:
                        m[c] = F(r.choice([-1, 1]) * 2.0 ** r.uniform(-30, 30))
                    else:
                        m[c] = r.choice(zeros)
                # now and then: only one kept entry is in range (the "anchor"), the others are tiny, huge or zero
                if sum(keep) == 2 and r.random() < 0.3:
                    k = [c for c in range(3) if keep[c]][int(r.integers(2))]
                    m[k] = r.choice([F(0.0), F(-0.0), F(2.0 ** -140), F(2.0 ** 100)])
                d = r.choice([F(0.0), F(-0.0), F(r.normal()), F(5.0), F(-2.5)])
                v = (r.choice([-1, 1], 3) * 2.0 ** r.uniform(-60, 59.9, 3)).astype(np.float32)
                if r.random() < 0.3:      # force a cancellation between the kept terms, or against d
                    k = [c for c in range(3) if keep[c]][0]
                    if m[k] != 0:
                        v[k] = F(-d / m[k]) if abs(d) > 0 and 2.0 ** -60 <= abs(F(-d / m[k])) < 2.0 ** 60 else v[k]
                a = full_row(m, F(d), v)
                b = sparse_row(m, F(d), v, keep)
                assert a.tobytes() == b.tobytes() or (np.isnan(a) and np.isnan(b)), (name, m, d, v, a, b)
                checked += 1
    assert checked == 16000


def test_the_conditions_are_needed():
    """Outside the conditions the two sums do differ (sign of a zero result): what the wavefront-wide check in
    closest_hit() guards against."""
    m = np.array([0.25, -0.0, 0.0], np.float32)          # diagonal row 0, translation -0
    v = np.array([-0.0, -1.0, -1.0], np.float32)          # x == -0: violates the coordinate condition
    a = full_row(m, F(-0.0), v)
    b = sparse_row(m, F(-0.0), v, [True, False, False])
    assert a == 0 and b == 0 and np.signbit(a) != np.signbit(b)
