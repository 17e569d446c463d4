"""SURVEY 8(d) bar (iii), the statistical part: two films that are INDEPENDENT estimates of the same image must differ
like Monte-Carlo noise.  Shared by the GPU-vs-f64-oracle tests (tests/test_gpu_parity.py) and the CPU test that sets the
Philox-addressed oracle against the sequential-stream oracle (tests/test_oracle_stream.py)."""
import numpy as np


def independent_films_look_like_noise(a, b, sigma, outliers=0.01):
    """a, b: grey films [H, W]; sigma: per-pixel sigma of ONE film's pixel mean (same for both).  Asserts: image-mean
    difference within 3 sigma; per-pixel z-scores with unit spread (robust estimate, heavy tails allowed); a sign test
    (the difference of two draws of one estimator is symmetric about 0 whatever its per-pixel distribution); bounded
    far outliers.  Returns (share of lit pixels, spread)."""
    lit = sigma > 0
    z = (a - b)[lit] / (np.sqrt(2.0) * sigma[lit])
    mean_sigma = np.sqrt(2.0 * (sigma[lit] ** 2).sum()) / lit.sum()            # sigma of the image-mean difference
    assert abs((a - b)[lit].mean()) <= 3.0 * mean_sigma, ((a - b)[lit].mean(), mean_sigma)
    spread = np.median(np.abs(z)) / 0.6745                                     # robust estimate of std(z), 1 for pure noise
    assert 0.75 <= spread <= 1.35, spread
    pos = np.mean(z > 0)
    assert abs(pos - 0.5) <= 3.0 * 0.5 / np.sqrt(z.size), pos
    # sigma is itself estimated from a few parts, and a pixel whose parts missed a rare bright sample underestimates it:
    # far outliers are bounded, not excluded
    assert np.mean(np.abs(z) > 4.0) <= outliers, np.mean(np.abs(z) > 4.0)
    return lit.mean(), spread
