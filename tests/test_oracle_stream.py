"""The Philox ADDRESSING of the build (counter = (x, y, sample, depth): one shared "surface" block per vertex, the
roulette uniform assembled from the low bits u01() skips, the light index / lobe draws in a second block, 23-bit
uniforms on the open interval, no redraw in random_range) checked against something other than itself: the f64
recursive oracle driven by the REFERENCE's own draw source -- one sequential StdRng (ChaCha12) stream per pixel, seeded
(y << 32) | x (main.rs:51-52), 53-bit uniforms on [0, 1), draws pulled in the reference's program order
(world.rs:299,255; shape.rs:111-112,211-212; material.rs:100-101; mirror.rs:42-43,232; rendering.rs:100).
Both films are estimates of the same image by the same integrator code; if the addressing scheme correlated draws that
the estimator needs independent, or biased a uniform, the films would differ by more than Monte-Carlo noise.

The StdRng restatement is unverified against the rand crate (oracle/pt_oracle.hpp); for THIS test that does not matter:
any generator of independent uniforms consumed sequentially serves as the independent second opinion."""
import os

import numpy as np
import pytest

from nobias import independent_films_look_like_noise

THREADS = min(8, os.cpu_count() or 1)


@pytest.mark.parametrize("scene,label,outliers", [(2, "C2", 0.01), (1, "C1", 0.05)])
def test_philox_addressing_agrees_with_the_sequential_stream(pt, orc, scene, label, outliers):
    """C2 (BASELINE configs[1] scene) and C1 (World::new(): triangle lights, GGX glass sphere -- the lobe draw, the
    conditional draws and two lights) at 64 x 64 x 4096 spp."""
    objs = pt.builtin_scene(scene)
    cam = pt.camera_new(width=64, height=64)
    spp, parts = 4096, 16
    sub = spp // parts
    grey = lambda f: f.mean(axis=-1)
    # Philox film from its 16 consecutive 256-spp parts (their mean IS the 4096-spp film); the parts give sigma
    pr = np.stack([grey(orc.render(cam, objs, pt.default_params(spp=sub, spp_offset=sub * k), orc.F64, orc.RECURSIVE, THREADS)[0])
                   for k in range(parts)])
    philox = pr.mean(axis=0)
    sigma = pr.std(axis=0, ddof=1) / np.sqrt(float(parts))
    lin, _, cnt = orc.render_stdrng(cam, objs, pt.default_params(spp=spp), THREADS)
    stream = grey(lin)
    lit, spread = independent_films_look_like_noise(stream, philox, sigma, outliers)
    print(f"{label}: image means philox {philox.mean():.6f} stream {stream.mean():.6f}, z-spread {spread:.3f}, "
          f"{cnt['vertices'] / (64 * 64 * spp):.4f} vertices per sample")
    assert lit > 0.95


def test_sequential_stream_is_one_stream_per_pixel(pt, orc):
    """The stream is consumed across all samples of a pixel (world.rs:296-312): sample k of a pixel depends on how many
    draws samples 0..k-1 took -- spp_offset replays them first (the skip-ahead of the reference's diagnostics,
    world.rs:634-652) -- and pixels are independent of each other (main.rs:51)."""
    objs = pt.builtin_scene(1)
    cam = pt.camera_new(width=400, height=400)                     # the reference's own size: pixels (79,176), (10,158)
    xy = [[79, 176], [10, 158], [200, 200]]
    lin, smp = orc.render_pixels_stdrng(cam, objs, pt.default_params(spp=12), xy)
    assert np.allclose(smp.mean(axis=1), lin, rtol=1e-13)
    lin2, smp2 = orc.render_pixels_stdrng(cam, objs, pt.default_params(spp=5, spp_offset=7), xy)
    assert np.array_equal(smp2, smp[:, 7:])
    lin3, smp3 = orc.render_pixels_stdrng(cam, objs, pt.default_params(spp=12), xy[::-1])
    assert np.array_equal(smp3[::-1], smp)
    # the full-film entry renders the same pixels
    row, _, _ = orc.render_stdrng(cam, objs, pt.default_params(spp=12, band_rows=1, band_index=176, band_count=400), 2)
    assert np.array_equal(row[0, 79], lin[0])
    # and it is a different stream from the Philox one: same estimator, different samples
    plin, _ = orc.render_pixels(cam, objs, pt.default_params(spp=12), xy)
    assert not np.array_equal(plin, lin)
