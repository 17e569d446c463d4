"""Size-independent properties of the HIP path in its default (fast) arithmetic mode --
the mode bench.py measures -- checked at BASELINE.json's full
sizes (1024x1024, 64 spp and the 10 000-sphere scene): determinism, invariance to
how the work is scheduled (batches, queue segments, row bands), linearity in spp,
consistency of the two film planes, closed-form radiometry."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _render(pt, ctx, cam, **kw):
    lin, rgba = ctx.render(cam, pt.default_params(**kw))
    return lin.cpu().numpy(), rgba.cpu().numpy(), ctx.stats()


def test_full_size_c2_determinism_and_schedule_invariance(pt, gpu_ctx):
    """1024^2 x 64 spp, scene C2: two runs are bitwise equal, and the film does not depend on the sample
    batch size, the number of queue segments, or the row-band partition (1 / 2 / 8 tiles)."""
    gpu_ctx.upload(pt.builtin_scene(2))
    cam = pt.camera_new(width=1024, height=1024)
    a, a8, sa = _render(pt, gpu_ctx, cam, spp=64)
    b, b8, sb = _render(pt, gpu_ctx, cam, spp=64)
    assert np.array_equal(a, b) and np.array_equal(a8, b8) and sa.vertices == sb.vertices
    assert sa.samples == 1024 * 1024 * 64 and sa.batches == 1
    assert np.isfinite(a).all() and a.min() >= 0.0
    c, c8, sc = _render(pt, gpu_ctx, cam, spp=64, max_paths_in_flight=5 * 1024 * 1024)     # 13 ragged batches
    assert sc.batches == 13 and np.array_equal(a, c) and np.array_equal(a8, c8) and sc.vertices == sa.vertices
    d, d8, sd = _render(pt, gpu_ctx, cam, spp=64, workgroups=333)
    assert np.array_equal(a, d) and np.array_equal(a8, d8) and sd.vertices == sa.vertices
    for G, band in [(2, 64), (8, 16)]:
        frame = np.zeros_like(a)
        verts = 0
        for g in range(G):
            t, _, st = _render(pt, gpu_ctx, cam, spp=64, band_rows=band, band_index=g, band_count=G)
            frame[pt.tile_row_indices(1024, band, g, G)] = t
            verts += st.vertices
        assert np.array_equal(frame, a) and verts == sa.vertices
    # V/S for this scene (reported in DESIGN.md): forced 4 bounces, early exits on the light / open front
    assert 4.0 < sa.vertices / sa.samples < 6.0


def test_full_size_spp_linearity(pt, gpu_ctx):
    """The film is a mean over samples: halves rendered with spp_offset average to the whole."""
    gpu_ctx.upload(pt.builtin_scene(2))
    cam = pt.camera_new(width=1024, height=1024)
    full, _, _ = _render(pt, gpu_ctx, cam, spp=64)
    lo, _, _ = _render(pt, gpu_ctx, cam, spp=32, spp_offset=0)
    hi, _, _ = _render(pt, gpu_ctx, cam, spp=32, spp_offset=32)
    assert np.allclose((lo.astype(np.float64) + hi) / 2, full, rtol=3e-7, atol=1e-9)
    assert not np.array_equal(lo, hi)


def test_rgba8_plane_is_the_gamma_quantised_linear_plane(pt, gpu_ctx):     # world.rs:322-332
    gpu_ctx.upload(pt.builtin_scene(1))
    lin, rgba, _ = _render(pt, gpu_ctx, pt.camera_new(width=256, height=256), spp=16)
    q = np.floor(np.clip(np.sqrt(lin.astype(np.float64)), 0, 1) * 255.0)
    assert (np.abs(q - rgba[..., :3]) <= 1).all() and (q == rgba[..., :3]).mean() > 0.99
    assert np.all(rgba[..., 3] == 255)


def test_c4_ten_thousand_spheres_full_width_determinism(pt, gpu_ctx):
    """The tiled-LDS path at the C4 scene size: deterministic and band-invariant (1024 wide, 64 rows, 8 spp)."""
    gpu_ctx.upload(pt.builtin_scene(4, 10000))
    cam = pt.camera_new(width=1024, height=64)
    a, a8, sa = _render(pt, gpu_ctx, cam, spp=8, accel=0)
    for wg in (0, 0, 7, 64, 2048):           # repeated and differently scheduled renders: a race in the
        r, r8, sr = _render(pt, gpu_ctx, cam, spp=8, accel=0, workgroups=wg)   # workgroup-level compaction shows up here
        assert np.array_equal(a, r) and np.array_equal(a8, r8) and sa.vertices == sr.vertices, wg
    b, b8, sb = _render(pt, gpu_ctx, cam, spp=8, accel=0, max_paths_in_flight=3 * 1024 * 64, workgroups=100)
    assert np.array_equal(a, b) and np.array_equal(a8, b8) and sa.vertices == sb.vertices
    frame = np.zeros_like(a)
    for g in range(4):
        t, _, _ = _render(pt, gpu_ctx, cam, spp=8, accel=0, band_rows=8, band_index=g, band_count=4)
        frame[pt.tile_row_indices(64, 8, g, 4)] = t
    assert np.array_equal(frame, a)


def test_exact_mode_full_size_matches_fast_mode_statistically(pt, gpu_ctx):
    """1024^2 x 16 spp C2 in both arithmetic modes: same estimator, ulp-perturbed paths."""
    gpu_ctx.upload(pt.builtin_scene(2))
    cam = pt.camera_new(width=1024, height=1024)
    fast, _, sf = _render(pt, gpu_ctx, cam, spp=16)
    exact, _, se = _render(pt, gpu_ctx, cam, spp=16, exact_math=1)
    assert abs(fast.mean() - exact.mean()) <= 2e-5 * exact.mean()
    rel = np.abs(fast.astype(np.float64) - exact) / np.maximum(np.abs(exact), 1e-6)
    assert (rel.max(-1) <= 1e-3).mean() >= 0.99
    assert abs(int(sf.vertices) - int(se.vertices)) <= 1e-4 * se.vertices


def test_furnace_brdf_only_on_gpu(pt, gpu_ctx):
    """Convex Lambertian sphere (albedo rho) in a uniform emitter Le: outgoing radiance = rho * Le."""
    rho, le = 0.6, 2.0
    gpu_ctx.upload(pt.make_objects([(0, [0, 0, -3, 1.0], 0, [rho] * 3), (0, [0, 0, 0, 40.0], 1, [le] * 3)]))
    lin, _, _ = _render(pt, gpu_ctx, pt.camera_new(width=128, height=128, fov_degrees=10.0), spp=256, integrator=1)
    assert lin.mean() == pytest.approx(rho * le, rel=2e-3)


def test_mis_equals_brdf_only_without_roulette_on_gpu(pt, gpu_ctx):
    gpu_ctx.upload(pt.builtin_scene(2))
    cam = pt.camera_new(width=128, height=128)
    mis, _, _ = _render(pt, gpu_ctx, cam, spp=256, min_depth=60000, max_depth=60001)
    bo, _, st = _render(pt, gpu_ctx, cam, spp=1024, min_depth=60000, max_depth=60001, integrator=1)
    assert mis.mean() == pytest.approx(bo.mean(), rel=5e-3)
    assert st.max_depth_reached > 50                         # long paths drain through repeated bounce groups


def test_edge_sizes(pt, orc, gpu_ctx):
    """Smallest legal image, one sample, one-pixel-row tile, an empty tile, odd sizes."""
    objs = pt.builtin_scene(1)
    gpu_ctx.upload(objs)
    for (w, h, spp) in [(2, 2, 1), (3, 5, 1), (65, 2, 3), (2, 67, 2)]:
        cam = pt.camera_new(width=w, height=h)
        prm = pt.default_params(spp=spp, exact_math=1)
        lin, rgba = gpu_ctx.render(cam, prm)
        ref, ref8, _ = orc.render(cam, objs, prm, orc.F32, orc.ITERATIVE)
        assert np.array_equal(lin.cpu().numpy(), ref.astype(np.float32)) and np.array_equal(rgba.cpu().numpy(), ref8)
    cam = pt.camera_new(width=8, height=8)
    lin, rgba = gpu_ctx.render(cam, pt.default_params(spp=2, band_rows=64, band_index=1, band_count=2))
    assert lin.shape[0] == 0 and gpu_ctx.stats().samples == 0


def test_progressive_preview_ends_bit_identical_to_one_shot(pt, gpu_ctx):
    """pt_render_progressive (the reference's live window, main.rs:79-90): frames after k samples equal a
    one-shot render of k samples, the last frame equals the full render, and the callback can stop early."""
    gpu_ctx.upload(pt.builtin_scene(1))
    cam = pt.camera_new(width=96, height=64)
    frames = []
    lin, rgba = gpu_ctx.render_progressive(cam, pt.default_params(spp=10), 4,
                                           lambda done, total, f8, fl: frames.append((done, total, f8, fl)) and False)
    assert [f[0] for f in frames] == [4, 8, 10] and all(f[1] == 10 for f in frames)
    for done, _, f8, fl in frames:
        one, one8, _ = _render(pt, gpu_ctx, cam, spp=done)
        assert np.array_equal(fl, one) and np.array_equal(f8, one8)
    assert np.array_equal(lin, frames[-1][3]) and np.array_equal(rgba, frames[-1][2])
    seen = []
    gpu_ctx.render_progressive(cam, pt.default_params(spp=10), 3, lambda done, *_: seen.append(done) or done >= 6)
    assert seen == [3, 6]


def test_maximum_sizes(pt, orc, gpu_ctx):
    """The limits of the packed path state (row, x, sample-in-batch and depth are 16-bit fields): the widest and
    the tallest legal tile and more samples per pixel than one batch can index, bit-exact against the f32 oracle;
    one past the limit is a status code."""
    objs = pt.builtin_scene(2)
    gpu_ctx.upload(objs)
    for (w, h, spp) in [(65535, 2, 1), (2, 65535, 1), (2, 2, 70001)]:
        cam = pt.camera_new(width=w, height=h)
        prm = pt.default_params(spp=spp, exact_math=1)
        lin, rgba = gpu_ctx.render(cam, prm)
        st = gpu_ctx.stats()
        ref, ref8, cnt = orc.render(cam, objs, prm, orc.F32, orc.ITERATIVE, 16)
        assert np.array_equal(lin.cpu().numpy(), ref.astype(np.float32)), (w, h, spp)
        assert np.array_equal(rgba.cpu().numpy(), ref8) and st.vertices == cnt["vertices"]
        if spp > 65535:
            assert st.batches == 2                          # 65535 samples per pixel per batch at most
    for (w, h) in [(65536, 2), (2, 65536)]:
        with pytest.raises(pt._lib.PtError, match="65536"):
            gpu_ctx.render(pt.camera_new(width=w, height=h), pt.default_params(spp=1))
    # a taller image is fine as long as each tile has < 65536 rows
    cam = pt.camera_new(width=2, height=70000)
    lin, _ = gpu_ctx.render(cam, pt.default_params(spp=1, exact_math=1, band_rows=35000, band_index=1, band_count=2))
    ref, _, _ = orc.render(cam, objs, pt.default_params(spp=1, band_rows=35000, band_index=1, band_count=2), orc.F32, orc.ITERATIVE, 16)
    assert lin.shape[0] == 35000 and np.array_equal(lin.cpu().numpy(), ref.astype(np.float32))


def test_soak_mixed_entries_stay_deterministic(pt, gpu_ctx):
    """Forty calls in random order through every rendering entry of one context -- whole frames of changing size,
    spp, integrator, accel and arithmetic, band tiles, pixel lists, injected rays, multi-batch renders, progressive
    renders -- interleaved with scene changes.  Every call is made twice (not back to back): the second result must equal
    the first bit for bit, i.e. no entry leaves state behind that another one picks up."""
    rng = np.random.default_rng(2026)
    scenes = [pt.builtin_scene(1), pt.builtin_scene(2), pt.builtin_scene(4, 700)]
    jobs = []
    for k in range(20):
        sc = int(rng.integers(0, 3))
        w, h = int(rng.integers(8, 200)), int(rng.integers(8, 120))
        kind = ["frame", "frame", "tile", "pixels", "rays", "batches", "progressive"][int(rng.integers(0, 7))]
        jobs.append((sc, w, h, kind, int(rng.integers(1, 9)), int(rng.integers(0, 2)), int(rng.integers(0, 3)), int(rng.integers(0, 2)), int(rng.integers(0, 1000))))

    def run(job):
        sc, w, h, kind, spp, integ, accel, exact, seed = job
        gpu_ctx.upload(scenes[sc])
        cam = pt.camera_new(width=w, height=h)
        r = np.random.default_rng(seed)
        if kind == "frame":
            lin, rgba = gpu_ctx.render(cam, pt.default_params(spp=spp, integrator=integ, accel=accel, exact_math=exact))
            return lin.cpu().numpy(), rgba.cpu().numpy()
        if kind == "tile":
            lin, rgba = gpu_ctx.render(cam, pt.default_params(spp=spp, integrator=integ, accel=accel, band_rows=3, band_index=1, band_count=2))
            return lin.cpu().numpy(), rgba.cpu().numpy()
        if kind == "pixels":
            xy = np.stack([r.integers(0, w, 50), r.integers(0, h, 50)], 1)
            lin, rgba, smp = gpu_ctx.render_pixels(cam, pt.default_params(spp=spp, integrator=integ, accel=accel, exact_math=exact), xy, want_samples=True)
            return lin, rgba, smp
        if kind == "rays":
            rays = np.concatenate([r.uniform(-0.9, 0.9, (64, 3)) + [0, 0, -2], r.normal(size=(64, 3))], 1)
            return (gpu_ctx.ray_color(pt.default_params(spp=1, spp_offset=spp, integrator=integ, accel=accel), rays, r.integers(0, 400, (64, 2))),)
        if kind == "batches":
            lin, rgba = gpu_ctx.render(cam, pt.default_params(spp=spp + 3, integrator=integ, accel=accel, max_paths_in_flight=w * h * 2))
            assert gpu_ctx.stats().batches >= 2
            return lin.cpu().numpy(), rgba.cpu().numpy()
        lin, rgba = gpu_ctx.render_progressive(cam, pt.default_params(spp=spp + 2, integrator=integ, accel=accel), 2)
        return lin, rgba

    first = {}
    order = list(range(len(jobs))) * 2
    rng.shuffle(order)
    for k in order:
        out = run(jobs[k])
        assert all(np.isfinite(o).all() for o in out), jobs[k]
        if k in first:
            assert all(np.array_equal(a, b) for a, b in zip(first[k], out)), jobs[k]
        else:
            first[k] = out


def test_multi_device_objects_can_be_created_and_destroyed_repeatedly(pt):
    """pt_multi_create / destroy in a loop (contexts, RCCL communicator, events): no failure, frames stay the same."""
    objs = pt.builtin_scene(2)
    cam = pt.camera_new(width=64, height=48)
    ref = None
    for k in range(4):
        m = pt.Multi([0])
        m.upload(objs)
        lin, rgba = m.render_host(cam, pt.default_params(spp=3, band_rows=5))
        m.close()
        if ref is None:
            ref = (lin, rgba)
        assert np.array_equal(lin, ref[0]) and np.array_equal(rgba, ref[1])
