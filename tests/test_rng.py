"""The injected counter RNG (oracle/rng.h) against the published Philox-4x32-10 known-answer
vectors (Random123 kat_vectors: Salmon et al., SC'11) and its own contract."""
import numpy as np
import pytest

KAT = [
    ([0, 0, 0, 0], [0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
    ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
    ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
     [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]),
]


@pytest.mark.parametrize("ctr,key,expect", KAT)
def test_philox_known_answers(oracle_c, ctr, key, expect):
    assert [int(x) for x in oracle_c.philox(ctr, key)] == expect


def test_philox_known_answers_reference_build(oracle_ref):
    for ctr, key, expect in KAT:
        assert [int(x) for x in oracle_ref.philox(ctr, key)] == expect


def test_u01_is_half_open_at_zero(oracle_c):
    """cuRAND-style (0,1]: x*2^-32 + 2^-33 in fp32; never 0, can be exactly 1."""
    vals = np.stack([oracle_c.rand3(9, i, 1, 0) for i in range(4000)])
    assert vals.min() > 0.0 and vals.max() <= 1.0
    assert abs(vals.mean() - 0.5) < 0.01


def test_rand3_matches_philox_layout(oracle_c):
    """rand3(seed, index, iter, dim) = u01 of Philox(ctr={index,iter,dim,0}, key={lo,hi})[0..2]."""
    seed = (0x1234 << 32) | 0xdeadbeef
    o = oracle_c.philox([77, 5, 2, 0], [seed & 0xffffffff, seed >> 32])
    r = oracle_c.rand3(seed, 77, 5, 2)
    expect = (o[:3].astype(np.float32) * np.float32(2.0 ** -32)) + np.float32(2.0 ** -33)
    assert r.tobytes() == expect.astype(np.float32).tobytes()


def test_rand3_golden(oracle_c, golden):
    g = golden["units"]
    mine = np.stack([oracle_c.rand3(5, i * 7919, 3, i % 4) for i in range(64)])
    assert mine.tobytes() == g["rand3_seed5"].tobytes()
