"""Integrator-level properties of the oracle (src/rendering.rs): the recursive
(reference-shaped) and iterative (wavefront-shaped) forms agree, the f32 arithmetic
mode stays within the stated FP32 tolerance of f64, and closed-form radiometry holds."""
import numpy as np
import pytest

F64, F32, REC, ITER = 64, 32, 0, 1
SPH, TRI = 0, 1
LAMBERT, EMISSIVE, MIRROR = 0, 1, 2


@pytest.mark.parametrize("scene,arg,integ", [(1, 0, 0), (2, 0, 0), (4, 300, 0), (1, 0, 1), (2, 0, 1)])
def test_recursive_equals_iterative_f64(pt, orc, scene, arg, integ):
    """Same vertices, same decisions; only the summation order of the radiance terms differs."""
    objs = pt.builtin_scene(scene, arg)
    cam = pt.camera_new(width=24, height=24)
    prm = pt.default_params(spp=8, integrator=integ)
    a, ra, ca = orc.render(cam, objs, prm, F64, REC, 4)
    b, rb, cb = orc.render(cam, objs, prm, F64, ITER, 4)
    assert np.allclose(a, b, rtol=1e-11, atol=1e-13)
    # the truncating u8 of two linear values 1e-11 apart can differ where sqrt(v) * 255.999 straddles an integer: a tie,
    # at most a channel or two per image, one LSB
    dq = np.abs(ra.astype(np.int16) - rb.astype(np.int16))
    assert dq.max() <= 1 and np.count_nonzero(dq) <= 2
    # the recursive form keeps walking beta == 0 paths, so it can only visit MORE vertices
    assert ca["vertices"] >= cb["vertices"] and ca["max_depth"] == cb["max_depth"]
    if scene != 1:     # no Mirror => no zero-throughput paths => identical vertex sets
        assert ca["vertices"] == cb["vertices"] and ca["shadow_rays"] == cb["shadow_rays"]


def _prefix_agreement(pt, orc, scene, arg, n_paths, seed):
    """Fraction of paths whose f32 and f64 versions hit the same objects up to each depth."""
    objs = pt.builtin_scene(scene, arg)
    cam = pt.camera_new(width=256, height=256)
    prm = pt.default_params(spp=1)
    rng = np.random.default_rng(seed)
    same, tot, dt0 = np.zeros(6), np.zeros(6), []
    for _ in range(n_paths):
        x, y, s = (int(v) for v in rng.integers(0, [256, 256, 64]))
        a = orc.trace_path(cam, objs, prm, x, y, s, F64)
        b = orc.trace_path(cam, objs, prm, x, y, s, F32)
        ok = True
        for k in range(min(6, max(len(a), len(b)))):
            tot[k] += 1
            ok = ok and k < len(a) and k < len(b) and a[k][1] == b[k][1]
            same[k] += ok
            if ok and k == 0 and a[0][1] >= 0:
                dt0.append(abs(a[0][2] - b[0][2]) / a[0][2])
    return same / np.maximum(tot, 1), tot, np.array(dt0)


def test_f32_paths_follow_f64_paths_on_the_cornell_scenes(pt, orc):
    """C1/C2: the f32 path visits the same objects as the f64 path at every depth."""
    for scene in (1, 2):
        frac, tot, dt0 = _prefix_agreement(pt, orc, scene, 0, 400, 3)
        assert np.all(frac[tot > 20] >= 0.995), frac
        assert np.median(dt0) < 2e-6


def test_c4_is_chaotic_paths_agree_only_as_a_prefix(pt, orc):
    """C4 (10 000 spheres of radius 0.005-0.03) is a Sinai billiard: a bounce off a sphere of radius r
    turns a position error e into a direction error e/r, about x15-50 per bounce.  f32 and f64 therefore
    agree exactly at depth 0-1 and decorrelate by depth 3-4 whatever the arithmetic; the f32 tolerance
    for this scene is stated on the path prefix and on image statistics, not per pixel."""
    frac, tot, dt0 = _prefix_agreement(pt, orc, 4, 10000, 500, 4)
    assert frac[0] == 1.0 and frac[1] >= 0.995 and frac[2] >= 0.97
    assert np.median(dt0) < 5e-7 and np.percentile(dt0, 99) < 5e-6
    assert frac[4] < 0.9            # documents the decorrelation; if this ever passes, tighten the C4 bar


@pytest.mark.parametrize("scene,arg", [(1, 0), (2, 0), (4, 300)])
def test_f32_mode_within_fp32_tolerance_of_f64(pt, orc, scene, arg):
    """The stated FP32 tolerance (SURVEY 8d ii): per channel |d| <= 1e-3 + 1e-2*|ref| on >= 99.5 % of pixels,
    RGBA8 within 1 LSB on >= 99.5 %, image mean within 1e-3 relative."""
    objs = pt.builtin_scene(scene, arg)
    cam = pt.camera_new(width=48, height=48)
    prm = pt.default_params(spp=16)
    ref, ref8, _ = orc.render(cam, objs, prm, F64, REC, 8)
    got, got8, _ = orc.render(cam, objs, prm, F32, ITER, 8)
    ok = (np.abs(got - ref) <= 1e-3 + 1e-2 * np.abs(ref)).all(-1)
    assert ok.mean() >= 0.995
    assert (np.abs(got8.astype(int) - ref8.astype(int)) <= 1).all(-1).mean() >= 0.995
    assert abs(got.mean() - ref.mean()) <= 1e-3 * ref.mean()


def test_primary_emitter_returns_emission_unweighted(pt, orc):    # rendering.rs:43-46
    objs = pt.make_objects([(SPH, [0, 0, -2, 50.0], EMISSIVE, [3.0, 2.0, 1.0])])   # camera is inside the emitter
    cam = pt.camera_new(width=8, height=8)
    lin, rgba, _ = orc.render(cam, objs, pt.default_params(spp=3), F64, REC)
    assert np.array_equal(lin, np.broadcast_to([3.0, 2.0, 1.0], lin.shape))
    assert np.array_equal(rgba[..., :3], np.full_like(rgba[..., :3], 255)) and np.all(rgba[..., 3] == 255)


def test_background_is_black_and_quantisation_truncates(pt, orc):   # rendering.rs:141, world.rs:327-332
    objs = pt.make_objects([(SPH, [0, 0, 50, 1.0], LAMBERT, [0.5] * 3)])    # behind the camera
    cam = pt.camera_new(width=8, height=8)
    lin, rgba, cnt = orc.render(cam, objs, pt.default_params(spp=2), F64, REC)
    assert not lin.any() and not rgba[..., :3].any() and np.all(rgba[..., 3] == 255)
    assert cnt["vertices"] == 8 * 8 * 2            # one (missing) iteration per sample
    # emission 0.25 -> sqrt = 0.5 -> 127.5 -> 127 (truncation, not rounding)
    objs = pt.make_objects([(SPH, [0, 0, -2, 50.0], EMISSIVE, [0.25, 0.25, 0.25])])
    _, rgba, _ = orc.render(cam, objs, pt.default_params(spp=1), F64, REC)
    assert np.all(rgba[..., :3] == 127)


def test_furnace_brdf_only(pt, orc):
    """A Lambertian sphere (albedo rho) inside a uniformly emitting enclosure Le: a convex body has no
    interreflection, so the radiance leaving it is exactly rho*Le.  BRDF-only integrator (rendering.rs:214-265).
    (Not valid for MIS: seen from INSIDE a sphere light the reference's cone sampler, shape.rs:97-104,
    degenerates to the hemisphere facing the centre and returns points behind the observer.)"""
    rho, le = 0.6, 2.0
    objs = pt.make_objects([(SPH, [0, 0, -3, 1.0], LAMBERT, [rho] * 3), (SPH, [0, 0, 0, 40.0], EMISSIVE, [le] * 3)])
    cam = pt.camera_new(width=16, height=16, fov_degrees=10.0)    # every pixel sees the sphere
    lin, _, _ = orc.render(cam, objs, pt.default_params(spp=256, integrator=1), F64, REC, 8)
    assert lin.mean() == pytest.approx(rho * le, rel=0.02)


def test_mis_agrees_with_brdf_only_when_roulette_is_off(pt, orc):
    """With one light seen from outside (Q2 inactive) and Russian roulette disabled (min_depth huge, Q1
    inactive) the reference's MIS estimator is unbiased, so its image mean equals BRDF-only's."""
    objs = pt.builtin_scene(2)
    cam = pt.camera_new(width=16, height=16)
    mis, _, _ = orc.render(cam, objs, pt.default_params(spp=256, min_depth=60000, max_depth=60001), F64, ITER, 8)
    bo, _, _ = orc.render(cam, objs, pt.default_params(spp=1024, min_depth=60000, max_depth=60001, integrator=1),
                          F64, ITER, 8)
    assert mis.mean() == pytest.approx(bo.mean(), rel=0.03)


def test_roulette_drops_direct_light_q1(pt, orc):
    """SURVEY Q1 (rendering.rs:100-102 vs :81): a path terminated by roulette returns 0 and loses the NEE
    term of that vertex, so with roulette ON the MIS image is darker than with roulette OFF."""
    objs = pt.builtin_scene(2)
    cam = pt.camera_new(width=16, height=16)
    on, _, _ = orc.render(cam, objs, pt.default_params(spp=256), F64, ITER, 8)
    off, _, _ = orc.render(cam, objs, pt.default_params(spp=256, min_depth=60000, max_depth=60001), F64, ITER, 8)
    assert 0.90 * off.mean() < on.mean() < 0.995 * off.mean()


def test_zero_emission_emissive_is_not_a_light(pt, orc):          # emit().length() > 0
    objs = pt.make_objects([(SPH, [0, 0, -3, 1.0], EMISSIVE, [0, 0, 0]), (SPH, [0, 0, 0, 40.0], EMISSIVE, [1.0] * 3)])
    cam = pt.camera_new(width=8, height=8, fov_degrees=10.0)
    lin, _, cnt = orc.render(cam, objs, pt.default_params(spp=4), F64, ITER)
    assert not lin.any()                   # black absorber: bsdf = 0 (material.rs:147), nothing reflected
    assert cnt["shadow_rays"] > 0          # it was shaded like a surface, i.e. not treated as an emitter


def test_spp_offset_partitions_the_sample_set(pt, orc):
    objs = pt.builtin_scene(2)
    cam = pt.camera_new(width=16, height=16)
    full, _, _ = orc.render(cam, objs, pt.default_params(spp=8), F64, REC)
    a, _, _ = orc.render(cam, objs, pt.default_params(spp=4, spp_offset=0), F64, REC)
    b, _, _ = orc.render(cam, objs, pt.default_params(spp=4, spp_offset=4), F64, REC)
    assert np.allclose((a + b) / 2, full, rtol=1e-13)


def test_row_bands_partition_the_image(pt, orc):                  # pixels are independent: the stream is addressed by (x, y), main.rs:51
    objs = pt.builtin_scene(1)
    cam = pt.camera_new(width=20, height=22)
    full, full8, _ = orc.render(cam, objs, pt.default_params(spp=2), F32, ITER)
    rebuilt = np.zeros_like(full)
    for g in range(3):
        prm = pt.default_params(spp=2, band_rows=4, band_index=g, band_count=3)
        tile, _, _ = orc.render(cam, objs, prm, F32, ITER)
        rows = pt.tile_row_indices(22, 4, g, 3)
        assert tile.shape[0] == len(rows)
        rebuilt[rows] = tile
    assert np.array_equal(rebuilt, full)


def enclosing_light_scene(pt):
    """Objects INSIDE an emissive sphere (a sky dome): every shading point has dc < r for the light, so the cone sampler's
    near root is negative and the light point lies behind the cone direction (shape.rs:134-144)."""
    return pt.make_objects([
        (SPH, [0.0, 0.0, -2.0, 6.0], EMISSIVE, [0.8, 0.9, 1.0]),          # the dome; the camera (0, 0, 2) is inside too
        (SPH, [0.0, -100.6, -2.0, 100.0], LAMBERT, [0.6, 0.6, 0.6]),      # floor: reaches outside the dome, visible part inside
        (SPH, [-0.5, -0.2, -2.0, 0.4], LAMBERT, [0.8, 0.3, 0.3]),
        (SPH, [0.5, -0.3, -1.6, 0.3], LAMBERT, [0.3, 0.8, 0.3]),
        (SPH, [0.1, 0.5, -2.4, 0.35], MIRROR, [0.2, 0.9, 0.9, 0.9, 1.0, 1.5]),
    ])


def test_light_sample_from_inside_an_emissive_sphere(pt, orc):
    """The f32 specification of SphereShape::sample_surface_from_point returns the cone direction and the near root as
    (light_dir, distance); from inside the sphere that root is negative and the reference's own
    (point - from).normalize() / .length() (shape.rs:139-144, rendering.rs:58-60) are -direction and |t|.  f32 and f64
    oracle must agree on direction, distance and point there (round 3 handed on a negative distance: ADVICE r3)."""
    objs = enclosing_light_scene(pt)
    rng = np.random.default_rng(5)
    n = 2000
    frm = rng.uniform([-1.5, -0.5, -4.0], [1.5, 2.0, 1.0], size=(n, 3))     # all inside the dome
    r12 = rng.uniform(0.0, 1.0, size=(n, 2))
    a = orc.shape_sample(objs, frm, None, r12, F64)        # object 0 of the array = the dome
    b = orc.shape_sample(objs, frm, None, r12, F32)
    # out11 = point3, normal3, pdf_omega, light_dir3, distance
    assert (a[:, 10] > 0).all() and (b[:, 10] > 0).all()
    assert np.allclose(a[:, 10], b[:, 10], rtol=2e-5, atol=1e-5)
    assert np.allclose(a[:, 7:10], b[:, 7:10], atol=2e-5)
    assert np.allclose(a[:, 0:3], b[:, 0:3], atol=5e-5)
    assert np.allclose(a[:, 6], 1.0 / (2.0 * np.pi)) and np.allclose(b[:, 6], 1.0 / (2.0 * np.pi), rtol=1e-6)
    # the point is on the sphere and distance / direction are the point's
    assert np.allclose(np.linalg.norm(a[:, 0:3] - np.array([0.0, 0.0, -2.0]), axis=1), 6.0, rtol=1e-9)
    assert np.allclose(frm + a[:, 7:10] * a[:, 10:11], a[:, 0:3], atol=1e-9)


def test_f32_film_inside_an_emissive_sphere_follows_f64(pt, orc):
    objs = enclosing_light_scene(pt)
    cam = pt.camera_new(width=48, height=48)
    prm = pt.default_params(spp=16)
    ref, ref8, _ = orc.render(cam, objs, prm, F64, REC, 8)
    got, got8, c = orc.render(cam, objs, prm, F32, ITER, 8)
    ok = (np.abs(got - ref) <= 1e-3 + 1e-2 * np.abs(ref)).all(-1)
    assert ok.mean() >= 0.995, ok.mean()
    assert abs(got.mean() - ref.mean()) <= 1e-3 * ref.mean()
    assert c["shadow_rays"] > 0 and ref.mean() > 0.1
