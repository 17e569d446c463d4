"""Counter RNG: Philox4x32-10 against the Random123 known-answer vectors
(kat_vectors of the Random123 distribution), and the uniform mapping.  The
reference's own generator (rand 0.9.2 StdRng) is not in its tree and no reference
test pins any RNG output: "parity unpinned" at this boundary."""
import numpy as np

KATS = [
    ([0, 0, 0, 0], [0, 0], [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]),
    ([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2, [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]),
    ([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0],
     [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]),
]


def test_philox4x32_10_known_answers(orc):
    for ctr, key, exp in KATS:
        assert orc.philox(ctr, key) == exp


def test_u01_open_interval_and_exact_in_f32(orc):
    lo, hi = orc.u01(0), orc.u01(0xFFFFFFFF)
    assert lo == 2.0 ** -24 and hi == 1.0 - 2.0 ** -24
    rng = np.random.default_rng(1)
    for r in rng.integers(0, 2 ** 32, size=2000, dtype=np.uint64):
        u = orc.u01(int(r))
        assert 0.0 < u < 1.0
        assert float(np.float32(u)) == u                 # same value in f32 and f64
        assert u == (2 * (int(r) >> 9) + 1) / 2.0 ** 24


def test_streams_differ_by_pixel_sample_depth_block(orc):
    base = orc.philox([3, 5, 1, 0], [7, 9])
    assert orc.philox([4, 5, 1, 0], [7, 9]) != base     # sample
    assert orc.philox([3, 6, 1, 0], [7, 9]) != base     # depth
    assert orc.philox([3, 5, 0, 0], [7, 9]) != base     # block
    assert orc.philox([3, 5, 1, 0], [8, 9]) != base     # x
    assert orc.philox([3, 5, 1, 0], [7, 10]) != base    # y


def test_philox_uniformity_coarse(orc):
    vals = []
    for s in range(4000):
        vals += orc.philox([s, 0, 1, 0], [12, 34])
    u = np.array([orc.u01(v) for v in vals])
    assert abs(u.mean() - 0.5) < 0.01
    hist, _ = np.histogram(u, bins=16, range=(0, 1))
    assert hist.min() > 0.85 * len(u) / 16 and hist.max() < 1.15 * len(u) / 16


def test_roulette_word_uses_only_bits_the_other_draws_skip(orc):
    """The roulette uniform of a vertex is built from the low bits of its BLK_SURFACE words, which u01() of those words
    never reads: changing the roulette bits leaves the four sample uniforms alone and vice versa; the uniform itself
    is uniform and uncorrelated with them."""
    rng = np.random.default_rng(7)
    for _ in range(200):
        ds = [int(v) for v in rng.integers(0, 2 ** 32, size=4, dtype=np.uint64)]
        w = orc.rr_word(ds)
        assert w == ((ds[0] << 23) & 0xFFFFFFFF) | ((ds[1] & 0x1FF) << 14) | ((ds[2] & 0x1FF) << 5)
        top = [(v >> 9) << 9 for v in ds]                    # what u01 reads
        low = [v & 0x1FF for v in ds]
        other = [t | int(l) for t, l in zip(top, rng.integers(0, 512, size=4))]
        assert [orc.u01(a) for a in other] == [orc.u01(a) for a in ds]             # samples do not see the low bits
        swapped = [int(t) << 9 | l for t, l in zip(rng.integers(0, 2 ** 23, size=4), low)]
        assert orc.u01(orc.rr_word(swapped)) == orc.u01(w)                           # roulette does not see the top bits
    us, rs = [], []
    for s_ in range(6000):
        ds = orc.philox([s_, 2, 0, 0], [12, 34])
        rs.append(orc.u01(orc.rr_word(ds)))
        us.append([orc.u01(v) for v in ds])
    rs, us = np.array(rs), np.array(us)
    assert abs(rs.mean() - 0.5) < 0.012
    hist, _ = np.histogram(rs, bins=16, range=(0, 1))
    assert hist.min() > 0.8 * len(rs) / 16 and hist.max() < 1.2 * len(rs) / 16
    for k in range(4):
        assert abs(np.corrcoef(rs, us[:, k])[0, 1]) < 0.04
