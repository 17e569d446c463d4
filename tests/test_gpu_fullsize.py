"""BASELINE.json's configurations at their FULL sizes, DEFAULT arithmetic (what bench.py measures), against the f64
recursive oracle -- the reference-faithful restatement (oracle/pt_oracle.hpp) -- and through size-independent
properties where the oracle cannot follow (C3: 4.3e9 samples, C5: 8.5e9 samples).

SURVEY 8(d)(ii) as written: per channel |d| <= 1e-3 + 1e-2*|ref| on >= 99.5 % of pixels, RGBA8 within 1 LSB on
>= 99.5 %, image-mean relative error <= 1e-3, and "pixels outside must be explainable by replaying that pixel in the
oracle": up to 20 of them are replayed sample by sample (pt_render_pixels = World::render_pixel, world.rs:293, with
the radiance of every camera sample, against orc.render_pixels) and must show the branch-flip signature -- all but a
few samples agree to 1e-4 relative and, without those few, the pixel is far inside the tolerance."""
import os
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
F64, F32, REC, ITER = 64, 32, 0, 1
THREADS = min(16, os.cpu_count() or 1)


@pytest.mark.parametrize("scene,label", [(2, "C2"), (1, "C1")])
def test_full_size_default_mode_meets_fp32_tolerance_against_f64(pt, orc, gpu_ctx, scene, label):
    """1024 x 1024 x 64 spp (BASELINE configs[1] and the reference's own scene at that size), exact_math = 0."""
    objs = pt.builtin_scene(scene)
    cam = pt.camera_new(width=1024, height=1024)
    spp = 64
    prm = pt.default_params(spp=spp)
    gpu_ctx.upload(objs)
    lin, rgba = gpu_ctx.render(cam, prm)
    got, got8 = lin.cpu().numpy().astype(np.float64), rgba.cpu().numpy()
    t = time.time()
    ref, ref8, _ = orc.render(cam, objs, prm, F64, REC, THREADS)
    t_orc = time.time() - t
    assert np.isfinite(got).all()
    ok = (np.abs(got - ref) <= 1e-3 + 1e-2 * np.abs(ref)).all(-1)
    ok8 = (np.abs(got8.astype(int) - ref8.astype(int)) <= 1).all(-1)
    mean_rel = abs(got.mean() - ref.mean()) / ref.mean()
    print(f"{label} 1024^2 x 64 default mode vs f64 recursive oracle ({t_orc:.1f} s on {THREADS} threads): "
          f"{ok.mean() * 100:.3f} % of pixels within tolerance, RGBA8 +-1 LSB on {ok8.mean() * 100:.3f} %, "
          f"image-mean relative error {mean_rel:.2e}")
    assert ok.mean() >= 0.995, ok.mean()
    assert ok8.mean() >= 0.995, ok8.mean()
    assert mean_rel <= 1e-3, mean_rel
    # ---- "pixels outside must be explainable": replay up to 20 of them, sample by sample
    bad = np.argwhere(~ok)
    if len(bad) == 0:
        return
    pick = bad[np.linspace(0, len(bad) - 1, min(20, len(bad))).astype(int)]
    xy = pick[:, ::-1].astype(np.uint32)                                     # (x, y)
    plin, _, psmp = gpu_ctx.render_pixels(cam, prm, xy, want_samples=True)
    assert np.array_equal(plin, lin.cpu().numpy()[pick[:, 0], pick[:, 1]])  # the replay IS that pixel of the film
    olin, osmp = orc.render_pixels(cam, objs, prm, xy, F64, REC)
    assert np.allclose(olin, ref[pick[:, 0], pick[:, 1]], rtol=1e-12, atol=0)
    d = psmp.astype(np.float64) - osmp
    agree = (np.abs(d) <= 1e-4 * np.abs(osmp) + 1e-6).all(-1)               # [pixel, sample]
    flips = (~agree).sum(1)
    # without the flipped samples the pixel is two orders of magnitude inside the tolerance: they carry the excess
    resid = np.abs((d * agree[..., None]).sum(1)) / spp
    print(f"{label}: {len(bad)} pixels outside; replayed {len(xy)}: flipped samples per pixel {flips.tolist()}, "
          f"largest residual of the agreeing samples {resid.max():.2e}")
    # The signature of a branch flip: every outlier has at least one sample whose path took another branch in f32 (a roulette
    # decision, a lobe pick, a grazing hit), and WITHOUT those samples it agrees to 1e-4 (next assertion).  How many samples
    # of a pixel flip is a property of where the pixel lies -- 1-4 almost everywhere; a pixel on the seam of two wall
    # spheres of C2 sees the seam in many of its samples -- so the bound on the count is on the replayed population, not on
    # the single worst pixel: typically a few (median <= 4), rarely more than 12 (at most a tenth of the outliers), never
    # the majority of a pixel's samples (that would be a disagreement of the arithmetic, not a flip).
    assert (flips >= 1).all(), flips
    assert np.median(flips) <= 4 and np.mean(flips > 12) <= 0.1 and flips.max() < spp // 2, flips
    assert (resid <= 1e-4).all(), resid.max()


def test_c3_full_size_4096_spp_is_the_mean_of_its_64_spp_renders(pt, gpu_ctx):
    """BASELINE configs[2]: 1024 x 1024 x 4096 spp = 4.3e9 samples (16 sample batches of 2^28 paths).  The film is
    the f64 mean over samples in sample order (world.rs:311-315), so it equals the mean of the 64 consecutive 64-spp
    renders (spp_offset = 64 k) up to the f32 rounding of each film; two runs are bitwise equal."""
    gpu_ctx.upload(pt.builtin_scene(2))
    cam = pt.camera_new(width=1024, height=1024)
    full, full8 = gpu_ctx.render(cam, pt.default_params(spp=4096))
    st = gpu_ctx.stats()
    assert st.batches == 16 and st.samples == 1024 * 1024 * 4096
    again, again8 = gpu_ctx.render(cam, pt.default_params(spp=4096))
    st2 = gpu_ctx.stats()
    import torch
    assert torch.equal(full, again) and torch.equal(full8, again8) and st2.vertices == st.vertices
    del again, again8
    acc = np.zeros((1024, 1024, 3), dtype=np.float64)
    verts = 0
    for k in range(64):
        part, _ = gpu_ctx.render(cam, pt.default_params(spp=64, spp_offset=64 * k), want_rgba=False)
        acc += part.cpu().numpy()
        verts += gpu_ctx.stats().vertices
    assert verts == st.vertices
    full = full.cpu().numpy().astype(np.float64)
    assert np.allclose(acc / 64.0, full, rtol=3e-7, atol=1e-9)
    assert np.isfinite(full).all() and 4.0 < st.vertices / st.samples < 6.0


def test_c5_full_job_partition_and_sample_split(pt, gpu_ctx):
    """BASELINE configs[4]: 3840 x 2160.  (i) The frame an 8-device pt_multi render assembles (partition into
    interleaved bands, 16 B/pixel pack, gather layout, row permutation -- pt_debug_multi_emulate runs them on this one
    GPU with device-to-device copies where the real path runs ncclGather) equals the single render bit for bit at
    64 spp.  (ii) The full 1024-spp job (8.5e9 samples, 32 sample batches, one GPU) equals the mean of its sixteen
    64-spp renders."""
    gpu_ctx.upload(pt.builtin_scene(2))
    cam = pt.camera_new(width=3840, height=2160)
    one, one8 = gpu_ctx.render(cam, pt.default_params(spp=64))
    st = gpu_ctx.stats()
    assert st.samples == 3840 * 2160 * 64 and st.batches == 2
    one, one8 = one.cpu().numpy(), one8.cpu().numpy()
    lin, rgba = gpu_ctx.multi_emulate(8, cam, pt.default_params(spp=64))
    assert np.array_equal(lin, one) and np.array_equal(rgba, one8)
    del lin, rgba
    full, _ = gpu_ctx.render(cam, pt.default_params(spp=1024), want_rgba=False)
    stf = gpu_ctx.stats()
    assert stf.samples == 3840 * 2160 * 1024 and stf.batches == 32
    acc = one.astype(np.float64)
    verts = st.vertices
    for k in range(1, 16):
        part, _ = gpu_ctx.render(cam, pt.default_params(spp=64, spp_offset=64 * k), want_rgba=False)
        acc += part.cpu().numpy()
        verts += gpu_ctx.stats().vertices
    assert verts == stf.vertices
    assert np.allclose(acc / 16.0, full.cpu().numpy().astype(np.float64), rtol=3e-7, atol=1e-9)
