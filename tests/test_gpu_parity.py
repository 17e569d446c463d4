"""GPU parity proper: the HIP path, called through the C ABI, against the CPU oracle
on the same seeded inputs.  The library has two arithmetic modes (PtRenderParams.exact_math);
every case is rendered in both:

  * exact_math = 1 (IEEE div/sqrt) vs the f32 oracle (same arithmetic specification): the
    film must match BIT FOR BIT -- linear f32 plane, RGBA8 plane, vertex and shadow-ray
    counts.  This is the proof that the kernel logic (queues, compaction, RNG addressing,
    every branch of every material) is right.
  * both modes vs the f64 recursive oracle (the reference-faithful restatement): the stated
    FP32 tolerance of SURVEY 8d (ii): |d| <= 1e-3 + 1e-2*|ref| per channel on >= 99.5 % of
    pixels, RGBA8 within 1 LSB on >= 99.5 %, image mean within 1e-3 relative.
  * exact_math = 0 (default; hardware 1-ulp rcp/sqrt, what bench.py measures) additionally
    stays close to the f32 oracle: >= 99 % of pixels within 1e-3 relative, vertex count
    within 1e-4 (paths are the same paths, perturbed by ulps).
All four materials are bit-reproducible (OrenNayar's azimuth term is trig-free in the f32 specification)."""
import os

import numpy as np
import pytest

from nobias import independent_films_look_like_noise

pytestmark = pytest.mark.gpu
F64, F32, REC, ITER = 64, 32, 0, 1
THREADS = min(16, os.cpu_count() or 1)


def _with(prm, **kw):
    q = type(prm)()
    for name, _ in prm._fields_:
        setattr(q, name, getattr(prm, name))
    for k, v in kw.items():
        setattr(q, k, v)
    return q


def _check(pt, orc, ctx, objs, cam, prm, exact=True, f64_frac=0.995, f64_mean=1e-3, fast_close=0.99, vert_rel=1e-4):
    ctx.upload(objs)
    ref32, ref32_8, c32 = orc.render(cam, objs, prm, F32, ITER, THREADS)
    ref, ref8, _ = orc.render(cam, objs, prm, F64, REC, THREADS)
    out = None
    for exact_math in (1, 0):
        lin, rgba = ctx.render(cam, _with(prm, exact_math=exact_math))
        st = ctx.stats()
        got, got8 = lin.cpu().numpy(), rgba.cpu().numpy()
        assert not np.isnan(got).any()
        assert st.samples == got.shape[0] * got.shape[1] * prm.spp
        if exact_math and exact:
            assert np.array_equal(got, ref32.astype(np.float32)), \
                f"{(got != ref32.astype(np.float32)).any(-1).sum()} pixels differ from the f32 oracle"
            assert np.array_equal(got8, ref32_8)
            assert st.vertices == c32["vertices"] and st.shadow_rays == c32["shadow_rays"]
            assert st.max_depth_reached == c32["max_depth"]
        g = got.astype(np.float64)
        ok = (np.abs(g - ref) <= 1e-3 + 1e-2 * np.abs(ref)).all(-1)
        assert ok.mean() >= f64_frac, (exact_math, ok.mean())
        assert (np.abs(got8.astype(int) - ref8.astype(int)) <= 1).all(-1).mean() >= f64_frac
        if ref.mean() > 0:
            assert abs(g.mean() - ref.mean()) <= f64_mean * ref.mean(), exact_math
        if not exact_math:
            rel = np.abs(g - ref32) / np.maximum(np.abs(ref32), 1e-6)
            assert (rel.max(-1) <= 1e-3).mean() >= fast_close, (rel.max(-1) <= 1e-3).mean()
            assert abs(int(st.vertices) - c32["vertices"]) <= max(2, vert_rel * c32["vertices"])
        else:
            out = (got, st)
    return out


def test_config1_reference_scene_256x256x4(pt, orc, gpu_ctx):
    """BASELINE.json configs[0]: World::new()'s Cornell box + glass sphere, 256x256, 4 spp."""
    _check(pt, orc, gpu_ctx, pt.builtin_scene(1), pt.camera_new(width=256, height=256), pt.default_params(spp=4))


def test_config2_ten_sphere_cornell_small(pt, orc, gpu_ctx):
    _check(pt, orc, gpu_ctx, pt.builtin_scene(2), pt.camera_new(width=128, height=128), pt.default_params(spp=16))


def test_config4_ten_thousand_spheres_small(pt, orc, gpu_ctx):
    """The LDS-tiled scan (scene larger than one LDS tile), 10 000 spheres, 100 lights.  Bit-exact against
    the f32 oracle.  Against f64 this scene is chaotic (tests/test_oracle_integrator.py::
    test_c4_is_chaotic_paths_agree_only_as_a_prefix): per-pixel agreement is bounded by decorrelated paths,
    so the f64 bar here is >= 93 % of pixels within tolerance and the image mean within 1e-2
    (Monte-Carlo noise between decorrelated paths at 4 spp)."""
    _check(pt, orc, gpu_ctx, pt.builtin_scene(4, 10000), pt.camera_new(width=48, height=48), pt.default_params(spp=4, accel=0),
           f64_frac=0.93, f64_mean=1e-2, fast_close=0.88, vert_rel=2e-3)


def test_mixed_runs_triangles_and_spheres_tiled(pt, orc, gpu_ctx):
    """Object order alternates shape types (several runs) and exceeds one LDS tile."""
    base = list(pt.builtin_scene(1))
    many = list(pt.builtin_scene(4, 1500))
    objs = (pt._lib.PtObject * (len(base) + len(many)))(*(many[:700] + base[:6] + many[700:] + base[6:]))
    _check(pt, orc, gpu_ctx, objs, pt.camera_new(width=40, height=40), pt.default_params(spp=4, accel=0),   # the tiled scan
           f64_frac=0.95, f64_mean=1e-2, fast_close=0.93, vert_rel=2e-3)      # contains 1500 of the tiny C4 spheres: partly chaotic


@pytest.mark.parametrize("scene", [1, 2])
def test_brdf_only_integrator(pt, orc, gpu_ctx, scene):           # rendering.rs:214-265
    _check(pt, orc, gpu_ctx, pt.builtin_scene(scene), pt.camera_new(width=64, height=64),
           pt.default_params(spp=8, integrator=1))


def test_metal_and_oren_nayar_materials(pt, orc, gpu_ctx):
    objs = list(pt.builtin_scene(2))
    metal = pt.make_objects([(0, [-0.4, -0.6, -2.0, 0.4], 2, [0.15, 0.9, 0.7, 0.3, 1.0, 1.5])])[0]
    objs[6] = metal
    arr = (pt._lib.PtObject * len(objs))(*objs)
    _check(pt, orc, gpu_ctx, arr, pt.camera_new(width=64, height=64), pt.default_params(spp=8))
    on = pt.make_objects([(0, [0.4, -0.6, -2.0, 0.4], 3, [0.7, 0.7, 0.7, 0.5])])[0]
    objs[7] = on
    arr = (pt._lib.PtObject * len(objs))(*objs)
    _check(pt, orc, gpu_ctx, arr, pt.camera_new(width=64, height=64), pt.default_params(spp=8))
    # a scene of OrenNayar surfaces only (all walls and spheres), rough and smooth
    on_objs = list(pt.builtin_scene(2))
    for k, o in enumerate(on_objs):
        if o.mat_tag == 0:
            o.mat_tag = 3
            o.mat[3] = [0.0, 0.3, 0.6, 1.0][k % 4]
    arr = (pt._lib.PtObject * len(on_objs))(*on_objs)
    _check(pt, orc, gpu_ctx, arr, pt.camera_new(width=64, height=64), pt.default_params(spp=8))


def test_objects_inside_an_emissive_sphere(pt, orc, gpu_ctx):
    """An enclosing light (sky dome): every NEE sample is taken from INSIDE the emissive sphere, where the cone
    sampler's near root is negative (shape.rs:134-144).  Film bit-exact against the f32 oracle and within the FP32
    tolerance of the f64 recursive oracle (whose light_dir / distance are the reference's own point - from); the
    device function on its own as well.  (the f32 specification handed on a negative distance there.)"""
    from test_oracle_integrator import enclosing_light_scene
    objs = enclosing_light_scene(pt)
    _, st = _check(pt, orc, gpu_ctx, objs, pt.camera_new(width=64, height=64), pt.default_params(spp=16))
    assert st.shadow_rays > 0
    rng = np.random.default_rng(6)
    frm = rng.uniform([-1.5, -0.5, -4.0], [1.5, 2.0, 1.0], size=(4000, 3))
    r12 = rng.uniform(0.0, 1.0, size=(4000, 2))
    ref = orc.shape_sample(objs, frm, None, r12, F64)            # point3 normal3 pdf dir3 dist
    s32 = orc.shape_sample(objs, frm, None, r12, F32)
    for exact in (1, 0):
        got = gpu_ctx.debug_shape_sample(0, frm, r12=r12, exact_math=exact).astype(np.float64)   # point3 pdf dir3 dist
        assert (got[:, 7] > 0).all()
        assert np.allclose(got[:, 0:3], ref[:, 0:3], atol=5e-5) and np.allclose(got[:, 4:7], ref[:, 7:10], atol=2e-5)
        assert np.allclose(got[:, 7], ref[:, 10], rtol=2e-5, atol=1e-5)
    got = gpu_ctx.debug_shape_sample(0, frm, r12=r12, exact_math=1)
    assert np.array_equal(got[:, 4:8], s32[:, 7:11].astype(np.float32))


def test_no_lights_and_empty_scene(pt, orc, gpu_ctx):
    objs = pt.make_objects([(0, [0, 0, -3, 1.0], 0, [0.5] * 3)])
    got, st = _check(pt, orc, gpu_ctx, objs, pt.camera_new(width=16, height=16), pt.default_params(spp=2))
    assert not got.any() and st.shadow_rays == 0               # world.rs:252-254: no lights -> no NEE
    empty = (pt._lib.PtObject * 0)()
    gpu_ctx.upload(empty)
    lin, rgba = gpu_ctx.render(pt.camera_new(width=8, height=8), pt.default_params(spp=2))
    assert not lin.cpu().numpy().any() and np.all(rgba.cpu().numpy()[..., 3] == 255)
    assert gpu_ctx.stats().vertices == 8 * 8 * 2


def test_non_square_look_at_camera(pt, orc, gpu_ctx):              # camera.rs:94-130
    cam = pt.camera_look_at((0.6, 0.3, 1.8), (0.0, -0.3, -2.0), (0.0, 1.0, 0.0), 96, 40, 40.0)
    _check(pt, orc, gpu_ctx, pt.builtin_scene(1), cam, pt.default_params(spp=4))


def test_roulette_parameters(pt, orc, gpu_ctx):                    # rendering.rs:6-7,91-98
    _check(pt, orc, gpu_ctx, pt.builtin_scene(2), pt.camera_new(width=32, height=32),
           pt.default_params(spp=8, min_depth=1, max_depth=3))


def test_very_long_paths_hold_the_tolerance(pt, orc, gpu_ctx):
    """min_depth = 200: no roulette before the 200th vertex -- the paths that stay inside the box run hundreds of vertices, and
    ulp-level differences between f32 and f64 accumulate along them.  64 x 64 x 4 spp, where SURVEY 8d-ii's 0.5 % is 20 pixels
    (round 3 ran this on 16 x 16, where one pixel is 0.4 %, and had to bend the bar): >= 99.5 % of the pixels within the FP32
    tolerance of the f64 recursive oracle in both arithmetic modes, image mean within 1e-3, and every outlier (up to 20 replayed
    sample by sample) shows the flip signature: at least one sample disagrees, the rest agree to 1e-3.  Exact mode stays
    bit-identical to the f32 oracle; both modes within 5 % of its vertex count."""
    objs = pt.builtin_scene(2)
    cam = pt.camera_new(width=64, height=64)
    spp = 4
    prm = pt.default_params(spp=spp, min_depth=200, max_depth=300)
    gpu_ctx.upload(objs)
    ref32, ref32_8, c32 = orc.render(cam, objs, prm, F32, ITER, THREADS)
    ref, ref8, _ = orc.render(cam, objs, prm, F64, REC, THREADS)
    for exact_math in (1, 0):
        q = _with(prm, exact_math=exact_math)
        lin, rgba = gpu_ctx.render(cam, q)
        st = gpu_ctx.stats()
        got = lin.cpu().numpy()
        if exact_math:
            assert np.array_equal(got, ref32.astype(np.float32)) and np.array_equal(rgba.cpu().numpy(), ref32_8)
            assert st.vertices == c32["vertices"] and st.max_depth_reached == c32["max_depth"]
        assert abs(int(st.vertices) - c32["vertices"]) <= 5e-2 * c32["vertices"]
        g = got.astype(np.float64)
        ok = (np.abs(g - ref) <= 1e-3 + 1e-2 * np.abs(ref)).all(-1)
        print(f"exact_math {exact_math}: {st.vertices / (64 * 64 * spp):.0f} vertices per path (deepest {st.max_depth_reached}), "
              f"{(~ok).sum()} of 4096 pixels outside the tolerance (bar: 20)")
        assert st.max_depth_reached >= 200 and ok.mean() >= 0.995, ok.mean()
        assert abs(g.mean() - ref.mean()) <= 1e-3 * ref.mean()
        bad = np.argwhere(~ok)
        if len(bad) == 0:
            continue
        pick = bad[np.linspace(0, len(bad) - 1, min(20, len(bad))).astype(int)]
        xy = pick[:, ::-1].astype(np.uint32)
        plin, _, psmp = gpu_ctx.render_pixels(cam, q, xy, want_samples=True)
        assert np.array_equal(plin, got[pick[:, 0], pick[:, 1]])
        olin, osmp = orc.render_pixels(cam, objs, prm, xy, F64, REC)
        d = psmp.astype(np.float64) - osmp
        agree = (np.abs(d) <= 1e-3 * np.abs(osmp) + 1e-5).all(-1)       # (a 200-vertex path accumulates ~1e-5 per bounce off an R = 100 wall)
        flips = (~agree).sum(1)
        resid = np.abs((d * agree[..., None]).sum(1)) / spp
        assert (flips >= 1).all() and (resid <= 1e-3).all(), (flips, resid.max())


@pytest.mark.parametrize("scene,arg", [(1, 0), (2, 0), (4, 10000)])
def test_hit_scene_kernel_against_oracle(pt, orc, gpu_ctx, scene, arg):
    """World::hit_scene on 200k random rays: same object as the f32 oracle with the same t bits; same object
    as the f64 oracle except where two surfaces are closer than the f32 resolution, |dt| <= 1e-4*t + 2e-5."""
    rng = np.random.default_rng(11 + scene)
    n = 200_000
    o = np.stack([rng.uniform(-0.95, 0.95, n), rng.uniform(-0.95, 0.95, n), rng.uniform(-2.9, 1.9, n)], 1)
    d = rng.normal(size=(n, 3))
    rays = np.concatenate([o, d], 1)
    objs = pt.builtin_scene(scene, arg)
    gpu_ctx.upload(objs)
    ids, t = gpu_ctx.debug_hit_scene(rays, 0.001, float("inf"), exact_math=1)
    ids32, t32, _, _ = orc.hit_scene(objs, rays, 0.001, float("inf"), F32)
    assert np.array_equal(ids, ids32)
    assert np.array_equal(t[ids >= 0], t32[ids >= 0].astype(np.float32))
    ids64, t64, _, _ = orc.hit_scene(objs, rays, 0.001, float("inf"), F64)
    for exact_math in (1, 0):
        ids, t = gpu_ctx.debug_hit_scene(rays, 0.001, float("inf"), exact_math=exact_math)
        same = ids == ids64
        assert same.mean() >= 0.9999
        hit = same & (ids >= 0)
        # grazing hits are ill-conditioned (t moves with sqrt of the discriminant): judge the bulk, bound the tail
        dt = np.abs(t[hit] - t64[hit])
        assert np.mean(dt <= 1e-4 * np.abs(t64[hit]) + 2e-5) >= 0.9995 and dt.max() < 1e-3
    # a finite t_max clips like hit_scene(shadow, 0.001, dist - 0.001) (rendering.rs:63-65)
    ids_c, _ = gpu_ctx.debug_hit_scene(rays[:5000], 0.001, 0.5, exact_math=1)
    ids_c32, _, _, _ = orc.hit_scene(objs, rays[:5000], 0.001, 0.5, F32)
    assert np.array_equal(ids_c, ids_c32)


def _quad_scene(pt, rng, n_quads, n_single, n_spheres):
    """Parallelogram quads as two triangles fanned from one corner (the pairs the scan tests together), single triangles
    and spheres, shuffled so that pair runs, triangle runs and sphere runs alternate; a quad's halves stay adjacent."""
    groups = []
    for _ in range(n_quads):
        a = rng.uniform([-0.9, -0.9, -2.9], [0.9, 0.9, -1.1])
        e1, e2 = rng.normal(size=3) * 0.35, rng.normal(size=3) * 0.35
        b, c, d = a + e1, a + e1 + e2, a + e2
        col = list(rng.uniform(0.2, 0.9, 3))
        groups.append([(1, list(a) + list(b) + list(c), 0, col), (1, list(a) + list(c) + list(d), 0, col)])
    for _ in range(n_single):
        v = rng.uniform([-0.9, -0.9, -2.9], [0.9, 0.9, -1.1], size=(3, 3))
        groups.append([(1, list(v.ravel()), 0, list(rng.uniform(0.2, 0.9, 3)))])
    for k in range(n_spheres):
        c = rng.uniform([-0.8, -0.8, -2.8], [0.8, 0.8, -1.2])
        groups.append([(0, list(c) + [float(rng.uniform(0.05, 0.25))], 1 if k == 0 else 0, [9.0, 9.0, 9.0] if k == 0 else [0.7, 0.7, 0.7])])
    order = rng.permutation(len(groups))
    return pt.make_objects([spec for g in order for spec in groups[g]])


@pytest.mark.parametrize("n_quads,n_single,n_spheres", [(6, 0, 1), (5, 7, 6), (70, 40, 30)])   # the last one: > 128 objects, tiled scan
def test_triangle_pairs_are_tested_like_two_triangles(pt, orc, gpu_ctx, n_quads, n_single, n_spheres):
    """Consecutive triangles with the same v0 and plane normal (halves of a parallelogram; every wall of World::new())
    share one determinant / t / range test / hit point in the scan (tripair_test).  Same ids and t bits as the f32
    oracle, which tests every triangle on its own -- on random rays, on rays aimed at the shared diagonal (both halves
    accept: the later object wins, world.rs:281-287), at the corners and along the planes; with a finite t_max; through the
    BVH; and as a film."""
    rng = np.random.default_rng(500 + n_quads)
    objs = _quad_scene(pt, rng, n_quads, n_single, n_spheres)
    gpu_ctx.upload(objs)
    n = 60_000
    o = np.stack([rng.uniform(-0.95, 0.95, n), rng.uniform(-0.95, 0.95, n), rng.uniform(-2.9, 1.9, n)], 1)
    d = rng.normal(size=(n, 3))
    # rays through points of the quads' diagonals, corners and edges (exact in f64, rounded by the f32 conversion)
    tri = [np.array(ob.shape[:9]).reshape(3, 3) for ob in objs if ob.shape_tag == 1]
    aimed = []
    for v in tri[:200]:
        for w in ([1, 0, 0], [0, 0, 1], [0.5, 0, 0.5], [0.25, 0, 0.75], [0.5, 0.5, 0], [1 / 3, 1 / 3, 1 / 3]):
            tgt = np.array(w) @ v
            org = rng.uniform([-0.9, -0.9, -0.9], [0.9, 0.9, 1.5])
            aimed.append(np.concatenate([org, tgt - org]))
    rays = np.concatenate([np.concatenate([o, d], 1), np.array(aimed)], 0)
    for t_max in (float("inf"), 1.7):
        ids, t = gpu_ctx.debug_hit_scene(rays, 0.001, t_max, exact_math=1)
        ids32, t32, _, _ = orc.hit_scene(objs, rays, 0.001, t_max, F32)
        assert np.array_equal(ids, ids32)
        assert np.array_equal(t[ids >= 0], t32[ids >= 0].astype(np.float32))
        ids_b, t_b = gpu_ctx.debug_hit_scene(rays, 0.001, t_max, exact_math=1, accel=1)
        assert np.array_equal(ids_b, ids) and np.array_equal(t_b[ids >= 0], t[ids >= 0])
    cam = pt.camera_new(width=96, height=64)
    _check(pt, orc, gpu_ctx, objs, cam, pt.default_params(spp=4), f64_frac=0.97, fast_close=0.97, vert_rel=1e-2)


def test_pt_render_host_buffers_entry(pt, orc, gpu_ctx):
    """pt_render(): the one-shot entry with host buffers (= src/main.rs:43-60)."""
    objs = pt.builtin_scene(1)
    cam = pt.camera_new(width=40, height=24)
    prm = pt.default_params(spp=3, exact_math=1)
    lin, rgba = pt.render_host(cam, objs, prm)
    ref, ref8, _ = orc.render(cam, objs, prm, F32, ITER, THREADS)
    assert np.array_equal(lin, ref.astype(np.float32)) and np.array_equal(rgba, ref8)


def test_error_behaviour(pt, gpu_ctx):
    """Errors are status codes + message, never a crash (the reference panics: main.rs:59,66)."""
    fresh = pt.Context(0)
    with pytest.raises(pt._lib.PtError, match="no scene"):
        fresh.render(pt.camera_new(width=8, height=8), pt.default_params(spp=1))
    fresh.upload(pt.builtin_scene(2))
    with pytest.raises(pt._lib.PtError, match=">= 2"):
        fresh.render(pt.camera_new(width=1, height=8), pt.default_params(spp=1))
    with pytest.raises(pt._lib.PtError, match="spp"):
        fresh.render(pt.camera_new(width=8, height=8), pt.default_params(spp=0))
    with pytest.raises(pt._lib.PtError, match="band_index"):
        fresh.render(pt.camera_new(width=8, height=8), pt.default_params(spp=1, band_index=2, band_count=2))
    with pytest.raises(pt._lib.PtError, match="max_paths_in_flight"):
        fresh.render(pt.camera_new(width=64, height=64), pt.default_params(spp=1, max_paths_in_flight=100))
    bad = pt.make_objects([(7, [0] * 4, 0, [0.5] * 3)])
    with pytest.raises(pt._lib.PtError, match="shape_tag"):
        fresh.upload(bad)
    with pytest.raises(pt._lib.PtError):
        pt.Context(9999)
    fresh.close()


@pytest.mark.parametrize("scene,label,w,h,spp,integrator", [(2, "C2", 1024, 1024, 64, 0), (1, "C1", 1024, 1024, 64, 0),
                                                          (2, "C5 size", 3840, 2160, 8, 0), (1, "C1 BRDF-only", 1024, 1024, 32, 1)])
def test_full_size_exact_mode_is_bit_identical_to_the_f32_oracle(pt, orc, gpu_ctx, scene, label, w, h, spp, integrator):
    """BASELINE's sizes -- 1024^2 x 64 spp (6.7e7 samples, ~3e8 path vertices) and C5's 3840 x 2160 -- in exact arithmetic against
    the f32 oracle: every pixel of both film planes and every counter.  Events of probability 1e-8 per vertex
    (empty shadow intervals, paths trapped by total internal reflection to depth 50, roulette at the depth limit,
    NaNs) occur a few times at this size and not at 256^2 x 4."""
    import time
    objs = pt.builtin_scene(scene)
    cam = pt.camera_new(width=w, height=h)
    prm = pt.default_params(spp=spp, exact_math=1, integrator=integrator)
    gpu_ctx.upload(objs)
    lin, rgba = gpu_ctx.render(cam, prm)
    st = gpu_ctx.stats()
    t = time.time()
    ref, ref8, cnt = orc.render(cam, objs, prm, F32, ITER, THREADS)
    print(f"{label}: oracle f32 {time.time() - t:.1f} s on {THREADS} threads")
    got = lin.cpu().numpy()
    bad = np.argwhere(np.any(got != ref.astype(np.float32), axis=-1))
    assert len(bad) == 0, (label, len(bad), bad[:5].tolist())
    assert np.array_equal(rgba.cpu().numpy(), ref8)
    assert (st.vertices, st.shadow_rays, st.max_depth_reached) == (cnt["vertices"], cnt["shadow_rays"], cnt["max_depth"])


_REF_CACHE = {}


def _no_bias(pt, orc, ctx, objs, cam, spp, parts=8, same_frac=0.995, outliers=0.01, chaotic=False, mean_rel=1e-3, **kw):
    """SURVEY 8(d) parity bar (iii): GPU f32 (DEFAULT arithmetic, what bench.py measures) against the f64
    reference-faithful recursive oracle at `spp` samples per pixel.
    (1) Same sample indices: the two films differ only by f32 rounding and rare branch flips -- far below the
        Monte-Carlo noise.
    (2) Disjoint sample indices (spp_offset = spp): the GPU film is then an independent estimate; its difference to
        the oracle must look like noise -- image-mean difference within 3 sigma, per-pixel z-scores with unit spread
        (robust estimate: the glass sphere of C1 and the specular-free but tiny lights of C4 give heavy-tailed pixels,
        world.rs:417 hunts luminance > 10 samples), no excess of far outliers.
    The per-pixel sigma of an spp-sample mean is estimated from `parts` independent GPU renders of spp / parts samples."""
    ctx.upload(objs)
    key = (len(objs), tuple(objs[0].shape), cam.width, cam.height, spp)        # accel does not concern the oracle
    if key not in _REF_CACHE:
        _REF_CACHE[key] = orc.render(cam, objs, pt.default_params(spp=spp), F64, REC, THREADS)[0].mean(axis=-1)
    ref = _REF_CACHE[key]                                                      # grey value per pixel
    grey = lambda **p: ctx.render(cam, pt.default_params(**p, **kw))[0].cpu().numpy().astype(np.float64).mean(axis=-1)
    same = grey(spp=spp)
    other = grey(spp=spp, spp_offset=spp)
    sub = spp // parts
    pr = np.stack([grey(spp=sub, spp_offset=2 * spp + sub * k) for k in range(parts)])
    sigma = pr.std(axis=0, ddof=1) / np.sqrt(float(parts))                     # sigma of an spp-sample pixel mean
    lit = sigma > 0
    # (1) same samples: rounding-level agreement
    d_same = same - ref
    # image-mean relative error <= 1e-3.  In a chaotic scene f32 and f64 paths of the same sample decorrelate
    # after a few bounces, and the "same-sample" difference is then itself Monte-Carlo noise: there the bound is what (2)
    # applies to independent films, 3 sigma of the image-mean difference, where that is the larger one.
    mean_tol = mean_rel * ref.mean()
    print(f"same-sample image mean: GPU - f64 oracle = {d_same.mean():+.3e} = {d_same.mean() / ref.mean():+.2e} relative (bar {mean_rel:g})")
    if chaotic:
        mean_tol = max(mean_tol, 3.0 * np.sqrt(2.0 * (sigma[lit] ** 2).sum()) / lit.sum())
    assert abs(d_same.mean()) <= mean_tol, (d_same.mean(), ref.mean(), mean_tol)
    assert np.mean(np.abs(d_same[lit]) <= 0.5 * sigma[lit]) >= same_frac, np.mean(np.abs(d_same[lit]) <= 0.5 * sigma[lit])
    # (2) independent samples: differences are noise, not bias
    _, spread = independent_films_look_like_noise(other, ref, sigma, outliers)
    return lit.mean(), spread


def test_convergence_at_4096_spp_shows_no_bias(pt, orc, gpu_ctx):
    """C2 (BASELINE configs[1] scene), 128 x 128, 4096 spp."""
    # C2 holds the same-sample image mean to 1e-4 relative (ten times inside SURVEY's 1e-3): a systematic shift from the
    # fast-arithmetic substitutions (a = 1 sphere test, native sin/cos, roulette word from the low bits) would show here
    lit, _ = _no_bias(pt, orc, gpu_ctx, pt.builtin_scene(2), pt.camera_new(width=128, height=128), 4096, mean_rel=1e-4)
    assert lit > 0.95


def test_reference_scene_shows_no_bias_at_4096_spp(pt, orc, gpu_ctx):
    """C1 = World::new()'s own scene (world.rs:80-211: triangle walls, two triangle lights, the GGX glass sphere),
    96 x 96, 4096 spp, MIS: the glass BTDF makes the estimator heavy-tailed, hence 16 parts and robust statistics."""
    lit, _ = _no_bias(pt, orc, gpu_ctx, pt.builtin_scene(1), pt.camera_new(width=96, height=96), 4096, parts=16, same_frac=0.99,
                      outliers=0.05)       # caustic pixels under the glass sphere: a few samples carry the pixel
    assert lit > 0.95


@pytest.mark.parametrize("scene,label,outliers", [(2, "C2", 0.01), (1, "C1", 0.05)])
def test_gpu_film_agrees_with_the_reference_stream_oracle(pt, orc, gpu_ctx, scene, label, outliers):
    """The product against the oracle driven by the REFERENCE's own draw source: the f64 recursive form with one
    sequential StdRng (ChaCha12) stream per pixel, seeded (y << 32) | x and consumed in the reference's program order
    (main.rs:51-52; oracle/pt_oracle.hpp StdRngStream, restated from the published algorithm, unverified against the rand
    crate).  The GPU film (DEFAULT arithmetic, Philox-addressed draws) and that film are independent estimates of one
    image: at 96 x 96 x 4096 spp their difference must look like Monte-Carlo noise -- image-mean difference within 3
    sigma, robust z-spread near 1, no excess of far outliers (tests/nobias.py).  tests/test_oracle_stream.py makes the
    same comparison between the two draw sources inside the oracle; this one has the HIP path on one side."""
    objs = pt.builtin_scene(scene)
    cam = pt.camera_new(width=96, height=96)
    spp, parts = 4096, 16
    gpu_ctx.upload(objs)
    grey = lambda **p: gpu_ctx.render(cam, pt.default_params(**p))[0].cpu().numpy().astype(np.float64).mean(axis=-1)
    film = grey(spp=spp)
    sub = spp // parts
    pr = np.stack([grey(spp=sub, spp_offset=spp + sub * k) for k in range(parts)])
    sigma = pr.std(axis=0, ddof=1) / np.sqrt(float(parts))
    stream = orc.render_stdrng(cam, objs, pt.default_params(spp=spp), THREADS)[0].mean(axis=-1)
    lit, spread = independent_films_look_like_noise(film, stream, sigma, outliers)
    print(f"{label}: image means GPU {film.mean():.6f} reference-stream oracle {stream.mean():.6f}, z-spread {spread:.3f}")
    assert lit > 0.95


@pytest.mark.parametrize("accel", [0, 1])
def test_ten_thousand_spheres_show_no_bias_at_1024_spp(pt, orc, gpu_ctx, accel):
    """C4 (10 000 spheres, 100 lights) is chaotic -- f32 and f64 paths decorrelate after 3-4 bounces
    (tests/test_oracle_integrator.py::test_c4_is_chaotic_paths_agree_only_as_a_prefix) -- so the per-pixel bar (ii) of
    SURVEY 8(d) cannot hold there (test_config4_ten_thousand_spheres_small states what does); the bar that applies is
    (iii): at 1024 spp the GPU film, linear scan and BVH alike, is an unbiased estimate of what the f64 oracle
    estimates.  32 x 32 pixels of the full scene (the oracle's linear scan costs ~0.5 ms per sample and thread)."""
    objs = pt.builtin_scene(4, 10000)
    cam = pt.camera_new(width=32, height=32)
    # same-sample agreement is per-pixel only as far as paths stay correlated: demand it of 90 % of the pixels
    lit, spread = _no_bias(pt, orc, gpu_ctx, objs, cam, 1024, parts=16, same_frac=0.90, outliers=0.08, chaotic=True, accel=accel)
    assert lit > 0.5
