"""PtRenderParams.accel = 1 (BVH traversal instead of the reference's linear scan) must not
change a single bit: same object and same t for every ray of World::hit_scene, same film, same counters.
The linear scan is the specification (it is what the oracle does, world.rs:270-290); these tests compare the
two device paths with each other and the BVH path with the f32 oracle directly."""
import numpy as np
import pytest

from test_gpu_fuzz import random_scene

pytestmark = pytest.mark.gpu
F32, ITER = 32, 1


def _rays(rng, n, lo=(-0.95, -0.95, -2.9), hi=(0.95, 0.95, 1.9)):
    o = rng.uniform(lo, hi, (n, 3))
    d = rng.normal(size=(n, 3))
    return np.concatenate([o, d], 1)


def _same_hits(ctx, rays, t_min, t_max):
    for exact_math in (1, 0):
        i0, t0 = ctx.debug_hit_scene(rays, t_min, t_max, exact_math=exact_math, accel=0)
        i1, t1 = ctx.debug_hit_scene(rays, t_min, t_max, exact_math=exact_math, accel=1)
        assert np.array_equal(i0, i1), (exact_math, int((i0 != i1).sum()))
        assert np.array_equal(t0.view(np.uint32), t1.view(np.uint32)), exact_math
    return i1, t1


@pytest.mark.parametrize("scene,arg", [(1, 0), (2, 0), (4, 1000), (4, 10000)])
def test_bvh_hit_scene_equals_linear_scan(pt, orc, gpu_ctx, scene, arg):
    rng = np.random.default_rng(70 + scene + arg)
    objs = pt.builtin_scene(scene, arg)
    gpu_ctx.upload(objs)
    rays = _rays(rng, 300_000)
    ids, t = _same_hits(gpu_ctx, rays, 0.001, float("inf"))
    assert (ids >= 0).mean() > 0.03
    _same_hits(gpu_ctx, rays[:50_000], 0.001, 0.4)          # shadow-ray style clipping (rendering.rs:63-65)
    _same_hits(gpu_ctx, rays[:50_000], 0.3, 1.5)
    # and against the f32 oracle directly
    n = 40_000
    i1, t1 = gpu_ctx.debug_hit_scene(rays[:n], 0.001, float("inf"), exact_math=1, accel=1)
    i32, t32, _, _ = orc.hit_scene(objs, rays[:n], 0.001, float("inf"), F32)
    assert np.array_equal(i1, i32)
    assert np.array_equal(t1[i1 >= 0], t32[i1 >= 0].astype(np.float32))


def test_bvh_far_origins_axis_rays_and_nonfinite_rays(pt, gpu_ctx):
    """Where a slab test is fragile: origins far outside the scene (the padding scales with |o|), directions
    with zero components (1/0 = inf, 0*inf = NaN), origins on box planes, and NaN/inf rays (the linear scan
    lets NaN through the sphere test, Q10 -- the BVH path must reproduce that)."""
    rng = np.random.default_rng(5)
    objs = pt.builtin_scene(4, 2000)
    gpu_ctx.upload(objs)
    n = 60_000
    far = _rays(rng, n, lo=(-300, -300, -300), hi=(300, 300, 300))
    far[:, 3:] = -far[:, :3] + rng.normal(scale=2.0, size=(n, 3)) + np.array([0, 0, -2.0])   # aimed at the cloud
    ids, _ = _same_hits(gpu_ctx, far, 0.001, float("inf"))
    assert (ids >= 0).mean() > 0.003
    axis = _rays(rng, n)
    k = rng.integers(0, 3, n)
    axis[np.arange(n), 3 + k] = 0.0                                       # one zero component
    m = n // 3
    axis[np.arange(m), 3 + (k[:m] + 1) % 3] = 0.0                         # two zero components: axis-parallel rays
    axis[m:m + 500, 3:] = 0.0                                             # zero direction: normalize() keeps it, the tests go NaN
    _same_hits(gpu_ctx, axis, 0.001, float("inf"))
    # origins exactly on sphere-centre coordinates (box planes of many nodes are c +- r; also exercise equality)
    cen = np.array([[o.shape[0], o.shape[1], o.shape[2]] for o in objs])
    on = _rays(rng, n)
    pick = rng.integers(0, len(cen), n)
    on[:, 0] = cen[pick, 0]
    on[: n // 2, 3] = 0.0
    _same_hits(gpu_ctx, on, 0.001, float("inf"))
    bad = _rays(rng, 4096)
    bad[::4, 0] = np.nan
    bad[1::4, 4] = np.nan
    bad[2::4, 3] = np.inf
    bad[3::8, 2] = -np.inf
    _same_hits(gpu_ctx, bad, 0.001, float("inf"))


def test_bvh_ties_pick_the_highest_object_index(pt, gpu_ctx):
    """Coincident primitives give exactly equal t; the scan keeps the LAST one (accepts t <= closest)."""
    rng = np.random.default_rng(9)
    specs = []
    for i in range(40):
        c = rng.uniform([-1, -1, -3], [1, 1, -1])
        r = float(rng.uniform(0.1, 0.4))
        copies = int(rng.integers(1, 4))
        for _ in range(copies):
            specs.append((0, list(c) + [r], 0, [0.5, 0.5, 0.5]))
        if i % 5 == 0:
            v0 = rng.uniform([-1, -1, -3], [1, 1, -1]); v1 = v0 + rng.uniform(-1, 1, 3); v2 = v0 + rng.uniform(-1, 1, 3)
            for _ in range(2):
                specs.append((1, list(v0) + list(v1) + list(v2), 0, [0.5, 0.5, 0.5]))
    order = rng.permutation(len(specs))
    objs = pt.make_objects([specs[i] for i in order])
    gpu_ctx.upload(objs)
    ids, _ = _same_hits(gpu_ctx, _rays(rng, 200_000), 0.001, float("inf"))
    assert (ids >= 0).mean() > 0.2


@pytest.mark.parametrize("n_objs", [0, 1, 3, 4, 5, 9])
def test_bvh_tiny_scenes(pt, gpu_ctx, n_objs):
    """Root is the sentinel (empty scene), a single leaf (<= 4 objects) or a small tree."""
    rng = np.random.default_rng(20 + n_objs)
    specs = [(0, list(rng.uniform([-1, -1, -3], [1, 1, -1])) + [0.4], 1 if i == 0 else 0, [3.0, 3.0, 3.0]) for i in range(n_objs)]
    objs = pt.make_objects(specs)
    gpu_ctx.upload(objs)
    _same_hits(gpu_ctx, _rays(rng, 20_000), 0.001, float("inf"))
    cam = pt.camera_new(width=16, height=16)
    films = [gpu_ctx.render(cam, pt.default_params(spp=2, exact_math=1, accel=a))[0].cpu().numpy() for a in (0, 1)]
    assert np.array_equal(films[0], films[1], equal_nan=True)


@pytest.mark.parametrize("case", ["cornell", "ten_spheres", "random_1000", "fuzz_mixed", "fuzz_large", "brdf_only"])
def test_bvh_render_is_bit_identical_to_linear_render(pt, orc, gpu_ctx, case):
    rng = np.random.default_rng(33)
    kw = {}
    if case == "cornell":
        objs, cam = pt.builtin_scene(1), pt.camera_new(width=96, height=96)
    elif case == "ten_spheres":
        objs, cam = pt.builtin_scene(2), pt.camera_new(width=96, height=96)
    elif case == "random_1000":
        objs, cam = pt.builtin_scene(4, 1000), pt.camera_new(width=64, height=64)
    elif case == "fuzz_mixed":
        objs, cam = random_scene(pt, rng, 60), pt.camera_new(width=48, height=40)
    elif case == "fuzz_large":
        objs, cam = random_scene(pt, rng, 700), pt.camera_new(width=40, height=48)
    else:
        objs, cam, kw = pt.builtin_scene(4, 500), pt.camera_new(width=48, height=48), {"integrator": 1}
    gpu_ctx.upload(objs)
    for exact_math in (1, 0):
        out = []
        for accel in (0, 1):
            prm = pt.default_params(spp=6, exact_math=exact_math, accel=accel, **kw)
            lin, rgba = gpu_ctx.render(cam, prm)
            st = gpu_ctx.stats()
            out.append((lin.cpu().numpy(), rgba.cpu().numpy(), st.vertices, st.shadow_rays, st.max_depth_reached))
        assert np.array_equal(out[0][0], out[1][0], equal_nan=True), (case, exact_math)
        assert np.array_equal(out[0][1], out[1][1])
        assert out[0][2:] == out[1][2:]
    if case in ("random_1000", "cornell"):
        prm = pt.default_params(spp=6, exact_math=1, accel=1, **kw)
        ref, ref8, cnt = orc.render(cam, objs, prm, F32, ITER, 8)
        lin, rgba = gpu_ctx.render(cam, prm)
        assert np.array_equal(lin.cpu().numpy(), ref.astype(np.float32))
        assert np.array_equal(rgba.cpu().numpy(), ref8)
        assert gpu_ctx.stats().vertices == cnt["vertices"]


def test_bvh_render_with_hand_off_and_bands(pt, gpu_ctx):
    """Enough paths for the tail hand-off (continuation launches) and a banded tile, BVH vs linear."""
    objs = pt.builtin_scene(4, 300)
    gpu_ctx.upload(objs)
    cam = pt.camera_new(width=512, height=512)
    films = []
    for accel in (0, 1):
        prm = pt.default_params(spp=40, accel=accel, band_rows=32, band_index=1, band_count=2)
        films.append(gpu_ctx.render(cam, prm)[0].cpu().numpy())
    assert np.array_equal(films[0], films[1], equal_nan=True)


def test_bvh_refuses_non_finite_objects(pt, gpu_ctx):
    """accel = 1 with a NaN object: a status code, not a different picture (see tests/test_bvh_host.py)."""
    specs = [(0, [0.0, 0.0, -2.0, 0.5], 1, [3.0, 3.0, 3.0]), (0, [float("nan"), 0.0, -2.0, 0.3], 0, [0.5, 0.5, 0.5])]
    gpu_ctx.upload(pt.make_objects(specs))
    cam = pt.camera_new(width=8, height=8)
    gpu_ctx.render(cam, pt.default_params(spp=1, accel=0))
    with pytest.raises(RuntimeError, match="NaN/inf"):
        gpu_ctx.render(cam, pt.default_params(spp=1, accel=1))


def test_bvh_render_in_several_batches(pt, gpu_ctx):
    """spp split into sample batches by max_paths_in_flight (f64 film sums across batches), BVH vs linear."""
    objs = pt.builtin_scene(4, 400)
    gpu_ctx.upload(objs)
    cam = pt.camera_new(width=64, height=64)
    films = []
    for accel, cap in ((0, 0), (1, 64 * 64 * 3), (1, 64 * 64 * 7)):
        prm = pt.default_params(spp=16, accel=accel, max_paths_in_flight=cap)
        films.append(gpu_ctx.render(cam, prm)[0].cpu().numpy())
        if cap:
            assert gpu_ctx.stats().batches == -(-16 // (cap // (64 * 64)))
    assert np.array_equal(films[0], films[1], equal_nan=True) and np.array_equal(films[0], films[2], equal_nan=True)


def test_full_size_c4_bvh_film_equals_linear_scan_film(pt, gpu_ctx):
    """BASELINE config C4 at full size -- 10 000 spheres, 1024^2, 256 spp (2.7e8 samples, 4 sample batches), default
    arithmetic: the BVH render and the brute-force render give the same film bit for bit and count the same
    vertices and shadow rays."""
    gpu_ctx.upload(pt.builtin_scene(4, 10000))
    cam = pt.camera_new(width=1024, height=1024)
    out = []
    for accel in (1, 0):
        lin, rgba = gpu_ctx.render(cam, pt.default_params(spp=256, accel=accel))
        st = gpu_ctx.stats()
        out.append((lin.cpu().numpy(), rgba.cpu().numpy(), (st.samples, st.vertices, st.shadow_rays, st.max_depth_reached, st.batches)))
    assert out[0][2][0] == 1024 * 1024 * 256 and out[0][2][4] == 4
    assert out[0][2] == out[1][2]
    assert np.array_equal(out[0][0], out[1][0], equal_nan=True) and np.array_equal(out[0][1], out[1][1])
    assert np.isfinite(out[0][0]).all()


def test_bvh_film_does_not_depend_on_traversal_scheduling(pt, gpu_ctx):
    """Which lane traces which ray (refill threshold, leaf batching, number of queue segments) must not matter."""
    gpu_ctx.upload(pt.builtin_scene(4, 3000))
    cam = pt.camera_new(width=256, height=128)
    ref, ref8 = gpu_ctx.render(cam, pt.default_params(spp=8, accel=1))
    ref, ref8 = ref.cpu().numpy(), ref8.cpu().numpy()
    try:
        for refill, leaf, wg in [(1, 1, 0), (64, 64, 0), (20, 3, 0), (36, 24, 7), (50, 40, 999)]:
            gpu_ctx.set_tuning(bvh_refill=refill, bvh_leaf=leaf)          # pt_context_set_tuning
            lin, rgba = gpu_ctx.render(cam, pt.default_params(spp=8, accel=1, workgroups=wg))
            assert np.array_equal(lin.cpu().numpy(), ref) and np.array_equal(rgba.cpu().numpy(), ref8), (refill, leaf, wg)
    finally:
        gpu_ctx.set_tuning()


@pytest.mark.parametrize("t_min", [0.3, 0.05, 1e-6])
def test_bvh_with_unusual_t_min(pt, gpu_ctx, t_min):
    """A large t_min makes `distance - t_min` (the shadow ray's t_max, rendering.rs:64) negative or tiny for many
    light samples: the scan then finds nothing (empty interval) and the light counts as visible.  The BVH path
    must agree -- also in what it does with slots that have no shadow ray at all."""
    rng = np.random.default_rng(3)
    for objs in (pt.builtin_scene(4, 600), random_scene(pt, rng, 200), pt.builtin_scene(1)):
        gpu_ctx.upload(objs)
        cam = pt.camera_new(width=96, height=64)
        out = []
        for accel in (0, 1):
            lin, _ = gpu_ctx.render(cam, pt.default_params(spp=8, accel=accel, t_min=t_min))
            st = gpu_ctx.stats()
            out.append((lin.cpu().numpy(), st.vertices, st.shadow_rays))
        assert np.array_equal(out[0][0], out[1][0], equal_nan=True), t_min
        assert out[0][1:] == out[1][1:]
    rays = _rays(rng, 50_000)
    gpu_ctx.upload(pt.builtin_scene(4, 600))
    for t_max in (-0.5, 0.0, 0.5 * t_min, float("nan")):
        _same_hits(gpu_ctx, rays, t_min, t_max)


def test_bvh_on_a_triangle_mesh(pt, gpu_ctx):
    """A bumpy terrain of 2 * 100 * 100 triangles (shared edges and vertices: rays through edges hit two triangles
    at equal or nearly equal t) under a sphere light, with a glass and a metal sphere: film and counters of the BVH
    render equal the linear scan's, both arithmetic modes."""
    rng = np.random.default_rng(1)
    n = 100
    xs, zs = np.linspace(-2.0, 2.0, n + 1), np.linspace(-5.0, -1.0, n + 1)
    X, Z = np.meshgrid(xs, zs, indexing="ij")
    Y = -0.8 + 0.15 * np.sin(3 * X) * np.cos(2.5 * Z) + 0.02 * rng.standard_normal(X.shape)
    P = np.stack([X, Y, Z], -1)
    specs = []
    for i in range(n):
        for j in range(n):
            a, b, c, d = P[i, j], P[i + 1, j], P[i + 1, j + 1], P[i, j + 1]
            specs.append((1, list(a) + list(c) + list(b), 0, [0.7, 0.7, 0.7]))
            specs.append((1, list(a) + list(d) + list(c), 0, [0.4, 0.6, 0.4]))
    specs.append((0, [0.0, 1.6, -3.0, 0.5], 1, [12.0, 12.0, 12.0]))
    specs.append((0, [-0.7, -0.35, -2.6, 0.3], 2, [0.1, 0.95, 0.95, 0.95, 1.0, 1.5]))
    specs.append((0, [0.8, -0.3, -3.2, 0.35], 2, [0.05, 1.0, 1.0, 1.0, 0.0, 1.5]))
    objs = pt.make_objects(specs)
    depth, nodes, slots = pt.bvh_check(objs)
    assert slots == len(objs)
    gpu_ctx.upload(objs)
    cam = pt.camera_look_at((0.0, 1.0, 1.5), (0.0, -0.6, -3.0), (0.0, 1.0, 0.0), 128, 128, 40.0)
    for exact_math in (1, 0):
        out = []
        for accel in (1, 0):
            lin, rgba = gpu_ctx.render(cam, pt.default_params(spp=4, accel=accel, exact_math=exact_math))
            st = gpu_ctx.stats()
            out.append((lin.cpu().numpy(), rgba.cpu().numpy(), (st.vertices, st.shadow_rays, st.max_depth_reached)))
        assert out[0][2] == out[1][2]
        assert np.array_equal(out[0][0], out[1][0], equal_nan=True) and np.array_equal(out[0][1], out[1][1])
    assert out[0][0].mean() > 0.01


def test_auto_accel_picks_by_scene_size_and_never_changes_the_film(pt, gpu_ctx):
    """PT_ACCEL_AUTO (the default): linear scan up to 512 objects, BVH above, linear again for a scene the BVH
    refuses -- the film is the same in every case, only bounce-kernel choice differs."""
    cam = pt.camera_new(width=48, height=48)
    for n, refused in ((300, False), (900, False), (900, True)):
        objs = pt.builtin_scene(4, n)
        if refused:
            objs[5].shape[0] = float("nan")
        gpu_ctx.upload(objs)
        auto, _ = gpu_ctx.render(cam, pt.default_params(spp=4))
        lin, _ = gpu_ctx.render(cam, pt.default_params(spp=4, accel=0))
        assert pt.default_params().accel == 2
        assert np.array_equal(auto.cpu().numpy(), lin.cpu().numpy(), equal_nan=True), (n, refused)
        ids_a, t_a = gpu_ctx.debug_hit_scene(_rays(np.random.default_rng(n), 20000), accel=2)
        ids_l, t_l = gpu_ctx.debug_hit_scene(_rays(np.random.default_rng(n), 20000), accel=0)
        assert np.array_equal(ids_a, ids_l) and np.array_equal(t_a.view(np.uint32), t_l.view(np.uint32))
