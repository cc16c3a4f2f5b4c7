"""Philox4x32-10 known answers (Random123 kat_vectors) and the uniform mapping."""
import numpy as np


def _philox_py(ctr, key):
    """Independent pure-Python statement of Philox4x32-10 (Salmon et al. 2011)."""
    M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
    c, k = list(ctr), list(key)
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        c = [((p1 >> 32) ^ c[1] ^ k[0]) & 0xFFFFFFFF, p1 & 0xFFFFFFFF,
             ((p0 >> 32) ^ c[3] ^ k[1]) & 0xFFFFFFFF, p0 & 0xFFFFFFFF]
        k = [(k[0] + W0) & 0xFFFFFFFF, (k[1] + W1) & 0xFFFFFFFF]
    return c


KAT = [  # counter, key, expected  (Random123 examples/kat_vectors, philox4x32 10)
    ([0, 0, 0, 0], [0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
    ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
    ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
     [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]),
]


def test_philox_known_answers(oracle):
    for ctr, key, want in KAT:
        assert _philox_py(ctr, key) == want
        assert oracle.philox(ctr, key) == want


def test_philox_matches_python_on_random_inputs(oracle):
    rng = np.random.default_rng(0)
    for _ in range(200):
        ctr = [int(x) for x in rng.integers(0, 2 ** 32, 4)]
        key = [int(x) for x in rng.integers(0, 2 ** 32, 2)]
        assert oracle.philox(ctr, key) == _philox_py(ctr, key)


def test_uniform_mapping(oracle):
    seed = 0x123456789ABCDEF0
    for env, stream, ctr, blk in [(0, 1, 0, 0), (5, 2, 7, 3), (2 ** 32 - 1, 4, 123456, 1)]:
        o = _philox_py([env, stream, ctr, blk], [seed & 0xFFFFFFFF, seed >> 32])
        want = [((o[0] >> 5) * 67108864 + (o[1] >> 6)) / 9007199254740992.0,
                ((o[2] >> 5) * 67108864 + (o[3] >> 6)) / 9007199254740992.0]
        got = oracle.uniform2(seed, env, stream, ctr, blk)
        assert list(got) == want
        assert 0.0 <= got[0] < 1.0 and 0.0 <= got[1] < 1.0
    u = np.array([oracle.uniform2(42, e, 1, 0, 0) for e in range(4000)]).ravel()
    assert abs(u.mean() - 0.5) < 0.02 and abs(u.var() - 1 / 12) < 0.01
