"""Function-level GPU parity: the per-vertex DEVICE functions of the path kernels -- not images --
called through the C ABI (pt_debug_*) on the committed fixtures of tests/golden and on seeded inputs.

  * default and exact arithmetic vs the f64 reference-faithful oracle / the committed f64 fixtures:
    |d pdf|, |d f| <= 1e-4 relative (1e-3 for Mirror), same branch decisions;
  * exact arithmetic vs the f32 oracle: bit for bit.

Rows covered: a3 camera, a4/a5/a6/a7 hit records, a9 sample_light_point, a10/a11 shape sampling, a12-a15 the four
materials (eval and sample), a8 ray_color on arbitrary rays, a2 render_pixel on a pixel list."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu
F64, F32, REC, ITER = 64, 32, 0, 1

MATS = {
    "lambert": (0, [0.8, 0.6, 0.2]),
    "emissive": (1, [15, 15, 15]),
    "glass": (2, [0.3, 1, 1, 1, 0.0, 1.5]),
    "metal": (2, [0.2, 0.9, 0.7, 0.3, 1.0, 1.5]),
    "oren_nayar": (3, [0.7, 0.7, 0.7, 0.5]),
}
SHAPES = {"sphere": (0, [0.0, 0.79, -2.0, 0.2], [36] * 3),
          "triangle": (1, [-0.3, 0.99, -2.3, 0.3, 0.99, -2.3, 0.3, 0.99, -1.7], [15] * 3)}


def _close(got, ref, rtol, atol):
    """Rows where the f64 value is finite and moderate (the GGX lobes reach 1e8 at grazing angles, where a
    relative 1e-3 of the input error is amplified without bound)."""
    fin = np.isfinite(ref).all(1) & (np.abs(ref).max(1) < 1e6)
    return np.allclose(got[fin], ref[fin], rtol=rtol, atol=atol), fin


@pytest.mark.parametrize("name", list(MATS))
def test_bsdf_eval_device_function(pt, orc, gpu_ctx, name):
    """Material::bsdf_pdf (material.rs:86-91,139-148,221-265; mirror.rs:62-124,179-198) on the committed vectors."""
    tag, params = MATS[name]
    ob = pt.make_objects([(0, [0, 0, 0, 1], tag, params)])
    gpu_ctx.upload(ob)
    g = np.load(os.path.join(GOLDEN, f"bsdf_{name}.npz"))
    rtol = 1e-3 if tag == 2 else 1e-4
    for exact in (1, 0):
        got = gpu_ctx.debug_bsdf_eval(0, g["eval_in"], exact_math=exact).astype(np.float64)
        ok, fin = _close(got, g["eval_out"], rtol, 1e-5)
        assert fin.mean() > 0.9 and ok, (name, exact)
    ref32 = orc.bsdf_eval(ob, g["eval_in"], F32).astype(np.float32)
    got = gpu_ctx.debug_bsdf_eval(0, g["eval_in"], exact_math=1)
    assert np.array_equal(got, ref32, equal_nan=True), name


@pytest.mark.parametrize("name", list(MATS))
def test_bsdf_sample_device_function(pt, orc, gpu_ctx, name):
    """Material::bsdf_pdf_sample (material.rs:29-40,93-122; mirror.rs:17-60,200-305) on the committed vectors:
    direction, f, pdf, cos; the lobe decision (reflect / refract / failed sample) must agree with f64."""
    tag, params = MATS[name]
    ob = pt.make_objects([(0, [0, 0, 0, 1], tag, params)])
    gpu_ctx.upload(ob)
    g = np.load(os.path.join(GOLDEN, f"bsdf_{name}.npz"))
    ref = g["sample_out"]
    rtol = 1e-3 if tag == 2 else 1e-4
    for exact in (1, 0):
        got = gpu_ctx.debug_bsdf_sample(0, g["sample_in"], g["draws"], exact_math=exact).astype(np.float64)
        # same lobe: the sampled direction lies on the same side of the surface, failed samples (pdf = 1, f = 0) agree
        nrm = g["sample_in"][:, 3:6]
        side_ref = np.sign((ref[:, 0:3] * nrm).sum(1)); side = np.sign((got[:, 0:3] * nrm).sum(1))
        same = side_ref == side
        assert same.mean() >= 0.99, (name, exact, same.mean())          # a Fresnel coin flip within 1e-7 of u may differ
        assert np.allclose(got[same, 0:3], ref[same, 0:3], atol=5e-5), (name, exact)
        ok, fin = _close(got[same][:, 3:8], ref[same][:, 3:8], rtol, 1e-5)
        assert ok and fin.mean() > 0.9, (name, exact)
    ref32 = orc.bsdf_sample(ob, g["sample_in"], g["draws"], F32).astype(np.float32)
    got = gpu_ctx.debug_bsdf_sample(0, g["sample_in"], g["draws"], exact_math=1)
    assert np.array_equal(got, ref32, equal_nan=True), name


@pytest.mark.parametrize("name", list(SHAPES))
def test_shape_sample_device_function(pt, orc, gpu_ctx, name):
    """Shape::sample_surface_from_point (shape.rs:91-145, 200-242): sampled point, solid-angle pdf, direction and
    distance; and the look-ahead form (target given, no draws) that the MIS weight of rendering.rs:114-117 uses."""
    st, sv, em = SHAPES[name]
    ob = pt.make_objects([(st, sv, 1, em)])
    gpu_ctx.upload(ob)
    g = np.load(os.path.join(GOLDEN, f"light_{name}.npz"))
    ref = g["sampled"]                      # point3 normal3 pdf dir3 dist
    for exact in (1, 0):
        got = gpu_ctx.debug_shape_sample(0, g["frm"], r12=g["r12"], exact_math=exact).astype(np.float64)
        assert np.allclose(got[:, 0:3], ref[:, 0:3], atol=2e-5)
        assert np.allclose(got[:, 3], ref[:, 6], rtol=1e-4)
        assert np.allclose(got[:, 4:7], ref[:, 7:10], atol=2e-5) and np.allclose(got[:, 7], ref[:, 10], rtol=1e-4, atol=2e-5)
        got_t = gpu_ctx.debug_shape_sample(0, g["frm"], target=ref[:, 0:3], exact_math=exact).astype(np.float64)
        assert np.allclose(got_t[:, 0:3], ref[:, 0:3], atol=1e-6)                 # the point is handed back
        assert np.allclose(got_t[:, 3], g["with_target"][:, 6], rtol=1e-4)
    s32 = orc.shape_sample(ob, g["frm"], None, g["r12"], F32)
    got = gpu_ctx.debug_shape_sample(0, g["frm"], r12=g["r12"], exact_math=1)
    assert np.array_equal(got[:, 0:4], s32[:, [0, 1, 2, 6]].astype(np.float32))
    assert np.array_equal(got[:, 4:8], s32[:, 7:11].astype(np.float32))


@pytest.mark.parametrize("scene,arg", [(1, 0), (2, 0), (4, 300)])
def test_sample_light_point_device_function(pt, orc, gpu_ctx, scene, arg):
    """World::sample_light_point (world.rs:251-267): uniform pick by the index word, surface sample, pdf / n_lights,
    emission -- 2 triangle lights (C1), one sphere light (C2), 3 sphere lights among 300 objects."""
    objs = pt.builtin_scene(scene, arg)
    gpu_ctx.upload(objs)
    rng = np.random.default_rng(5 + scene)
    n = 4096
    frm = np.stack([rng.uniform(-0.9, 0.9, n), rng.uniform(-0.9, 0.5, n), rng.uniform(-2.9, -1.1, n)], 1)
    words = rng.integers(0, 2**32, size=(n, 4), dtype=np.uint64).astype(np.uint32)
    ref = orc.light_point(objs, frm, words, F64)
    for exact in (1, 0):
        got = gpu_ctx.debug_light_point(frm, words, exact_math=exact).astype(np.float64)
        assert np.array_equal(got[:, 7], ref[:, 7])                               # the same light object
        assert np.array_equal(got[:, 3:6], ref[:, 3:6])                           # its emission
        # the lights of scene 4 have radii 0.005 - 0.03: the sampled point is the root of a quadratic whose
        # coefficients cancel to ~r^2 = 1e-4 from terms of order 1 (shape.rs:131-136): ~1e-3 r in f32
        dp = np.abs(got[:, 0:3] - ref[:, 0:3]).max(1)
        if scene == 4:
            # near the cone's edge sqrt(disc) -> 0 and the point moves ALONG the sphere by up to ~0.1 r in f32 (still a
            # point of the light, and the pdf does not depend on it); the bulk agrees to 5e-5
            assert np.mean(dp <= 5e-5) >= 0.95 and dp.max() <= 0.03, (np.mean(dp <= 5e-5), dp.max())
        else:
            assert dp.max() <= 5e-5          # worst of 4096 points: one next to the r = 0.2 light (the quadratic of shape.rs:131-136 cancels there)
        assert np.mean(np.isclose(got[:, 6], ref[:, 6], rtol=1e-4)) >= 0.999      # grazing triangle samples: pdf ~ 1/cos
    ref32 = orc.light_point(objs, frm, words, F32).astype(np.float32)
    assert np.array_equal(gpu_ctx.debug_light_point(frm, words, exact_math=1), ref32)
    assert len(set(ref[:, 7])) == {1: 2, 2: 1, 4: 3}[scene]


def test_camera_ray_device_function(pt, orc, gpu_ctx):
    """Camera::get_ray_with_offset (camera.rs:139-147) with the jitter draws of world.rs:299: the ray of sample s of
    pixel (x, y), for the reference camera and a look_at camera (camera.rs:94-130)."""
    gpu_ctx.upload(pt.builtin_scene(2))
    rng = np.random.default_rng(9)
    for cam in (pt.camera_new(width=400, height=400), pt.camera_look_at((0.6, 0.3, 1.8), (0.0, -0.3, -2.0), (0, 1, 0), 96, 40, 40.0)):
        n = 2000
        xys = np.stack([rng.integers(0, cam.width, n), rng.integers(0, cam.height, n), rng.integers(0, 5000, n)], 1).astype(np.uint32)
        off = np.array([[orc.u01(w) for w in orc.philox((int(x), int(y), int(s), 0xFFFFFFFF), (0, 0))[:2]] for x, y, s in xys])
        # world.rs:299 hands get_ray_with_offset the flipped row HEIGHT-1-y
        flipped = np.stack([xys[:, 0], cam.height - 1 - xys[:, 1]], 1)
        ref = orc.camera_rays(cam, flipped, off, F64)
        for exact in (1, 0):
            got = gpu_ctx.debug_camera_rays(cam, xys, exact_math=exact).astype(np.float64)
            assert np.array_equal(got[:, 6:8], off)                               # the uniforms are exact in f32
            assert np.allclose(got[:, 0:6], ref, atol=3e-7)
        ref32 = orc.camera_rays(cam, flipped, off, F32).astype(np.float32)
        assert np.array_equal(gpu_ctx.debug_camera_rays(cam, xys, exact_math=1)[:, 0:6], ref32)


@pytest.mark.parametrize("scene,arg,accel", [(1, 0, 0), (2, 0, 0), (4, 3000, 0), (4, 3000, 1)])
def test_hit_records_device_function(pt, orc, gpu_ctx, scene, arg, accel):
    """HitRecord of the closest hit (shape.rs:84-88,194-197; base.rs:19-33): point, face-forwarded normal,
    front_face -- Shape::hit's whole return value, through the LDS scan, the tiled scan and the BVH."""
    objs = pt.builtin_scene(scene, arg)
    gpu_ctx.upload(objs)
    rng = np.random.default_rng(21 + scene)
    n = 20000
    o = np.stack([rng.uniform(-0.95, 0.95, n), rng.uniform(-0.95, 0.95, n), rng.uniform(-2.9, 1.9, n)], 1)
    rays = np.concatenate([o, rng.normal(size=(n, 3))], 1)
    ids64, t64, pn64, ff64 = orc.hit_scene(objs, rays, 0.001, float("inf"), F64)
    ids32, t32, pn32, ff32 = orc.hit_scene(objs, rays, 0.001, float("inf"), F32)
    ids, rec = gpu_ctx.debug_hit_records(rays, exact_math=1, accel=accel)
    assert np.array_equal(ids, ids32)
    hit = ids >= 0
    assert np.array_equal(rec[hit, 0], t32[hit].astype(np.float32))
    assert np.array_equal(rec[hit, 1:7], pn32[hit].astype(np.float32)) and np.array_equal(rec[hit, 7] != 0, ff32[hit] != 0)
    assert not rec[~hit].any()
    for exact in (1, 0):
        ids, rec = gpu_ctx.debug_hit_records(rays, exact_math=exact, accel=accel)
        same = (ids == ids64) & (ids >= 0)
        assert same.sum() >= 0.999 * (ids64 >= 0).sum()
        # point: the bar of the hit-distance tests, |dt| <= 1e-4 t + 2e-5 (the absolute floor is the R = 100 wall spheres
        # of C2: o - c is only good to ulp(100) = 7.6e-6 in f32), with room for the direction's own rounding; grazing
        # hits are ill-conditioned: bulk and worst case
        dp = np.abs(rec[same, 1:4] - pn64[same, 0:3]).max(1)
        tol = 5e-5 + 1e-4 * t64[same]
        assert np.mean(dp <= tol) >= 0.999 and dp.max() <= 2e-2, (np.mean(dp <= tol), dp.max())
        assert np.mean(np.abs(rec[same, 4:7] - pn64[same, 3:6]).max(1) <= 2e-3) >= 0.995          # normal (R = 0.005 spheres: 1/r amplifies)
        assert np.mean((rec[same, 7] != 0) == (ff64[same] != 0)) >= 0.9999


# ---------------------------------------------------------------- World::render_pixel and ray_color through the ABI
def test_pixel_replay_matches_the_full_film(pt, orc, gpu_ctx):
    """The reference's own diagnostics replay single pixels of the 400 x 400 scene with the seed of main.rs:51
    (world.rs:378 pixel (79, 176), world.rs:531 pixel (10, 158)).  pt_render_pixels = World::render_pixel for a pixel
    list: every listed pixel must equal that pixel of the full film bit for bit (same key, same samples), in both
    arithmetic modes and both integrators, and the per-sample radiance list (what the reference prints, world.rs:417)
    must sum to the pixel and equal the f32 oracle's samples."""
    objs = pt.builtin_scene(1)
    gpu_ctx.upload(objs)
    cam = pt.camera_new(width=400, height=400)
    rng = np.random.default_rng(1)
    xy = np.array([[79, 176], [10, 158], [0, 0], [399, 399], [200, 37], [79, 176]] +
                  [[int(rng.integers(0, 400)), int(rng.integers(0, 400))] for _ in range(300)], dtype=np.uint32)
    spp = 32
    for exact, integ in [(1, 0), (0, 0), (1, 1)]:
        prm = pt.default_params(spp=spp, exact_math=exact, integrator=integ)
        full, full8 = gpu_ctx.render(cam, prm)
        full, full8 = full.cpu().numpy(), full8.cpu().numpy()
        lin, rgba, smp = gpu_ctx.render_pixels(cam, prm, xy, want_samples=True)
        assert np.array_equal(lin, full[xy[:, 1], xy[:, 0]]) and np.array_equal(rgba, full8[xy[:, 1], xy[:, 0]])
        assert smp.shape == (len(xy), spp, 3)
        assert np.array_equal((smp.astype(np.float64).sum(1) / spp).astype(np.float32), lin)      # world.rs:311-315, in sample order
        # sample offsets address the same streams: samples 8..15 on their own
        part, _, psmp = gpu_ctx.render_pixels(cam, pt.default_params(spp=8, spp_offset=8, exact_math=exact, integrator=integ), xy[:6], want_samples=True)
        assert np.array_equal(psmp, smp[:6, 8:16])
        if exact:
            # the oracle's per-sample values: one-sample renders of the pixel's row
            for (x, y) in xy[:2]:
                for s in (0, 1, 17, 31):
                    row, _, _ = orc.render(cam, objs, pt.default_params(spp=1, spp_offset=s, integrator=integ, band_rows=1,
                                                                         band_index=int(y), band_count=400), F32, ITER, 1)
                    i = int(np.argmax((xy[:, 0] == x) & (xy[:, 1] == y)))
                    assert np.array_equal(smp[i, s], row[0, x].astype(np.float32)), (x, y, s)
    # multi-batch lists (film sums across sample batches) give the same pixels; samples then cannot be returned
    prm = pt.default_params(spp=spp, max_paths_in_flight=len(xy) * 5)
    lin_b, rgba_b, _ = gpu_ctx.render_pixels(cam, prm, xy)
    assert gpu_ctx.stats().batches > 1
    full, full8 = gpu_ctx.render(cam, pt.default_params(spp=spp))
    assert np.array_equal(lin_b, full.cpu().numpy()[xy[:, 1], xy[:, 0]])
    with pytest.raises(pt._lib.PtError, match="one sample batch"):
        gpu_ctx.render_pixels(cam, prm, xy, want_samples=True)
    with pytest.raises(pt._lib.PtError, match="outside"):
        gpu_ctx.render_pixels(cam, prm, [[400, 0]])
    lin0, rgba0, _ = gpu_ctx.render_pixels(cam, prm, np.zeros((0, 2), dtype=np.uint32))
    assert lin0.shape == (0, 3)


@pytest.mark.parametrize("scene,arg,accel", [(1, 0, 0), (2, 0, 0), (4, 2000, 0), (4, 2000, 1)])
def test_pixel_list_through_every_scan(pt, gpu_ctx, scene, arg, accel):
    """Pixel lists through the LDS scan, the tiled scan and the BVH kernels (their LIST variants)."""
    gpu_ctx.upload(pt.builtin_scene(scene, arg))
    cam = pt.camera_new(width=96, height=64)
    prm = pt.default_params(spp=6, accel=accel)
    full, full8 = gpu_ctx.render(cam, prm)
    rng = np.random.default_rng(3)
    xy = np.stack([rng.integers(0, 96, 700), rng.integers(0, 64, 700)], 1).astype(np.uint32)
    lin, rgba, _ = gpu_ctx.render_pixels(cam, prm, xy)
    assert np.array_equal(lin, full.cpu().numpy()[xy[:, 1], xy[:, 0]]) and np.array_equal(rgba, full8.cpu().numpy()[xy[:, 1], xy[:, 0]])


@pytest.mark.parametrize("integrator", [0, 1])
def test_ray_color_on_arbitrary_rays(pt, orc, gpu_ctx, integrator):
    """RenderingStrategy::ray_color(world, ray, 0, rng, 1) (rendering.rs:34-142, 214-265) for rays that no camera
    generated: exact arithmetic == the f32 oracle bit for bit; default arithmetic within tolerance of the f64
    recursive (reference-faithful) form."""
    objs = pt.builtin_scene(1)
    gpu_ctx.upload(objs)
    rng = np.random.default_rng(77)
    n = 3000
    o = np.stack([rng.uniform(-0.9, 0.9, n), rng.uniform(-0.9, 0.9, n), rng.uniform(-2.8, -1.2, n)], 1)
    rays = np.concatenate([o, rng.normal(size=(n, 3)) * rng.uniform(0.1, 5.0, (n, 1))], 1)      # not normalised: Ray::new does it
    xy = np.stack([rng.integers(0, 400, n), rng.integers(0, 400, n)], 1).astype(np.uint32)
    prm = pt.default_params(spp=1, spp_offset=11, integrator=integrator, exact_math=1)
    got = gpu_ctx.ray_color(prm, rays, xy)
    ref32 = orc.ray_color(objs, prm, rays, xy, F32, ITER).astype(np.float32)
    assert np.array_equal(got, ref32)
    ref = orc.ray_color(objs, prm, rays, xy, F64, REC)
    fast = gpu_ctx.ray_color(pt.default_params(spp=1, spp_offset=11, integrator=integrator), rays, xy).astype(np.float64)
    ok = (np.abs(fast - ref) <= 1e-3 + 1e-2 * np.abs(ref)).all(1)
    assert ok.mean() >= 0.99, ok.mean()                  # single samples: a branch flip is a whole different path
    assert abs(fast.mean() - ref.mean()) <= 2e-2 * ref.mean()
    # a ray that starts on a light and looks at it from outside the box sees Le (rendering.rs:43-45)
    le = gpu_ctx.ray_color(prm, [[0.0, 0.5, -2.0, 0.0, 1.0, 0.0]], [[1, 1]])
    assert np.array_equal(le, [[15.0, 15.0, 15.0]])


# ---------------------------------------------------------------- the entry is asynchronous / graph-capturable
# (1, 1): queue-form level-0 launch + continuation launch; (1, 0): the reference scene's default, the regenerating form with
# batched Mirror vertices; (2, 0): regenerating form (its chunk counters are cleared by the resolve of the same graph)
@pytest.mark.parametrize("scene,form", [(1, 1), (1, 0), (2, 0)])
def test_render_entry_is_capturable_into_a_graph(pt, gpu_ctx, scene, form):
    """pt_render_device enqueues and returns: no host synchronisation, no allocation once its buffers exist.  So it can
    be captured into a hipGraph (torch.cuda.CUDAGraph on the stream the context renders on) and replayed; the film of
    a replay is the film of a direct call.  1024 x 1024 x 8 spp = 8.4 M paths: large enough for the tail hand-off
    (level-0 launch + continuation launch that reads its path count on the device)."""
    import torch
    objs = pt.builtin_scene(scene)
    gpu_ctx.upload(objs)
    gpu_ctx.set_tuning(level0_form=form)
    cam = pt.camera_new(width=1024, height=1024)
    prm = pt.default_params(spp=8)
    ref, ref8 = gpu_ctx.render(cam, prm)                       # also creates every buffer of this size
    st = gpu_ctx.stats()
    assert st.bounce_launches == (2 if form == 1 else 1) and st.batches == 1     # regenerating waves run dry themselves
    dev = torch.device("cuda", 0)
    lin = torch.zeros_like(ref); rgba = torch.zeros_like(ref8)
    stream = torch.cuda.Stream(dev)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(stream):
        gpu_ctx.set_stream(stream.cuda_stream)
        with torch.cuda.graph(g, stream=stream):
            gpu_ctx.render_into(cam, prm, lin.data_ptr(), rgba.data_ptr())
        for _ in range(2):
            lin.zero_(); rgba.zero_()
            g.replay()
            stream.synchronize()
            assert torch.equal(lin, ref) and torch.equal(rgba, ref8)
    gpu_ctx.set_stream(None)
    gpu_ctx.set_tuning()
    del g


def test_tuning_knobs_do_not_change_the_film(pt, gpu_ctx):
    gpu_ctx.upload(pt.builtin_scene(1))
    cam = pt.camera_new(width=1024, height=640)
    prm = pt.default_params(spp=8)
    ref, ref8 = gpu_ctx.render(cam, prm)
    base = gpu_ctx.stats()
    try:
        for eb in (1, 16, 200):
            gpu_ctx.set_tuning(export_below=eb)
            lin, rgba = gpu_ctx.render(cam, prm)
            st = gpu_ctx.stats()
            assert torch_equal(lin, ref) and torch_equal(rgba, ref8)
            assert (st.vertices, st.shadow_rays) == (base.vertices, base.shadow_rays)
        # launches of consecutive batches strictly in order (PtTuning.in_order) or overlapping (default): three batches each,
        # and three renders enqueued back to back behind one synchronisation
        import torch
        small = pt.default_params(spp=8, max_paths_in_flight=3 * 1024 * 640)
        for in_order in (1, 0):
            gpu_ctx.set_tuning(in_order=in_order)
            lin, rgba = gpu_ctx.render(cam, small)
            st = gpu_ctx.stats()
            assert torch_equal(lin, ref) and torch_equal(rgba, ref8) and st.batches == 3
            assert (st.vertices, st.shadow_rays) == (base.vertices, base.shadow_rays)
            outs = [(torch.zeros_like(ref), torch.zeros_like(ref8)) for _ in range(3)]
            for lin_d, rgba_d in outs:
                gpu_ctx.render_into(cam, prm, lin_d.data_ptr(), rgba_d.data_ptr())
            gpu_ctx.sync()
            for lin_d, rgba_d in outs:
                assert torch_equal(lin_d, ref) and torch_equal(rgba_d, ref8)
            assert gpu_ctx.stats().vertices == 3 * base.vertices
    finally:
        gpu_ctx.set_tuning()


def test_pipelined_renders_with_changing_launch_sizes_keep_their_exchange_memory_apart(pt, gpu_ctx):
    """Renders enqueued back to back overlap their launches, and a launch that shares the device takes a smaller grid than one that
    runs alone -- so consecutive launches of the reference scene (k_paths_regen_split, per-wave exchange stacks in global memory)
    differ in size while several are in flight.  Each lane's region of that memory must not move under a launch that is still
    running (round 4 had the lane stride follow the current render's grid: a memory fault in bench.py --workload c1).  Five
    frames of 1024 x 1024 x 16 spp posted back to back -- the first alone on the device, the others sharing it, the last two with
    a larger and a smaller tuned grid -- all equal the isolated render bit for bit, and no stack invariant is reported."""
    import torch
    gpu_ctx.upload(pt.builtin_scene(1))
    cam = pt.camera_new(width=1024, height=1024)
    prm = pt.default_params(spp=16)
    ref, ref8 = gpu_ctx.render(cam, prm)
    base = gpu_ctx.stats()
    outs = [(torch.zeros_like(ref), torch.zeros_like(ref8)) for _ in range(5)]
    try:
        for k, (lin_d, rgba_d) in enumerate(outs):
            if k == 3:
                gpu_ctx.set_tuning(regen_workgroups=1900)
            if k == 4:
                gpu_ctx.set_tuning(regen_workgroups=300)
            gpu_ctx.render_into(cam, prm, lin_d.data_ptr(), rgba_d.data_ptr())
        gpu_ctx.sync()                                  # raises if a kernel reported a violated stack invariant
        for k, (lin_d, rgba_d) in enumerate(outs):
            assert torch_equal(lin_d, ref) and torch_equal(rgba_d, ref8), k
        assert gpu_ctx.stats().vertices == 5 * base.vertices
    finally:
        gpu_ctx.set_tuning()


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_sequences_of_pipelined_renders_equal_isolated_renders(pt, seed):
    """The scheduling machinery of round 4 under a random load: THREE contexts on the device (C2 -> k_paths_regen, the reference
    scene -> k_paths_regen_split with its per-lane exchange memory, an OrenNayar Cornell box -> the kernel without the Mirror
    code), both integrators, 28 renders posted back to back without a host wait, in random order over the contexts -- whole images and row-band tiles, one to four sample batches per render (max_paths_in_flight),
    small jobs that take the queue form in between, exact and default arithmetic, a tuned grid now and then.  Lanes, buffer sets,
    spare workgroups, the counters the resolves clear: whatever they do, every film must equal the one the same job gives when it
    runs alone and in order, bit for bit, and the statistics must add up."""
    import torch
    rng = np.random.default_rng(seed)
    dev = torch.device("cuda", 0)
    ctxs = [pt.Context(0), pt.Context(0), pt.Context(0)]
    try:
        ctxs[0].upload(pt.builtin_scene(2))
        ctxs[1].upload(pt.builtin_scene(1))
        ctxs[2].upload(_oren_nayar_cornell(pt))
        jobs = []
        for _ in range(28):
            which = int(rng.integers(0, 3))
            W, H = [(256, 256), (320, 200), (512, 128), (64, 48)][int(rng.integers(0, 4))]
            spp = int(rng.choice([4, 8, 12]))
            bands = int(rng.choice([1, 1, 2, 3]))
            kw = dict(spp=spp, exact_math=int(rng.integers(0, 2)), spp_offset=int(rng.integers(0, 1000)), integrator=int(rng.random() < 0.25))
            if bands > 1:
                kw.update(band_rows=int(rng.choice([8, 16, 50])), band_index=int(rng.integers(0, bands)), band_count=bands)
            if rng.random() < 0.4:
                kw["max_paths_in_flight"] = int(W * H * spp // int(rng.choice([2, 3, 4])) + 1)
            tune = int(rng.choice([0, 0, 0, 300, 1700]))
            jobs.append((which, pt.camera_new(width=W, height=H), kw, tune))
        # every job alone and in order
        refs = []
        for which, cam, kw, tune in jobs:
            ctxs[which].set_tuning(in_order=1, regen_workgroups=tune)
            lin, rgba = ctxs[which].render(cam, pt.default_params(**kw))
            refs.append((lin.clone(), rgba.clone(), ctxs[which].stats().vertices))
        # ... and all of them posted back to back
        streams = [torch.cuda.Stream(dev) for _ in ctxs]
        for c, st in zip(ctxs, streams):
            c.set_stream(st.cuda_stream)
        outs = [(torch.zeros_like(r[0]), torch.zeros_like(r[1])) for r in refs]
        for c in ctxs:
            c.sync()
            c.stats()
        for (which, cam, kw, tune), (lin_d, rgba_d) in zip(jobs, outs):
            ctxs[which].set_tuning(regen_workgroups=tune)
            ctxs[which].render_into(cam, pt.default_params(**kw), lin_d.data_ptr(), rgba_d.data_ptr())
        for c in ctxs:
            c.sync()
        bad = [(k, jobs[k][0], jobs[k][1].width, jobs[k][1].height, jobs[k][2], jobs[k][3], float(lin_d.abs().sum()), float(lin.abs().sum()))
               for k, ((lin_d, rgba_d), (lin, rgba, _)) in enumerate(zip(outs, refs))
               if not (torch_equal(lin_d.view(torch.int32), lin.view(torch.int32)) and torch_equal(rgba_d, rgba))]
        assert not bad, bad
        for which in range(len(ctxs)):
            assert ctxs[which].stats().vertices == sum(r[2] for r, j in zip(refs, jobs) if j[0] == which)
    finally:
        for c in ctxs:
            c.set_stream(None)
            c.close()


def test_renders_keep_their_order_across_a_change_of_stream(pt):
    """pt_context_set_stream while renders are in flight: the context hands its sample buffers, counters and statistics from
    render to render in the order of one stream, so what is already enqueued on the old stream must stay ahead of what the new
    one gets.  Six renders (queue form and regenerating form in turn) posted back to back, the stream changed before each one,
    one synchronisation at the end: every film equals the job run alone."""
    import torch
    dev = torch.device("cuda", 0)
    ctx = pt.Context(0)
    try:
        ctx.upload(pt.builtin_scene(2))
        jobs = [(pt.camera_new(width=w, height=h), pt.default_params(spp=spp, spp_offset=7 * k))
                for k, (w, h, spp) in enumerate([(400, 300, 8), (64, 64, 8), (512, 256, 4), (48, 32, 12), (320, 320, 6), (64, 48, 4)])]
        refs = [tuple(t.clone() for t in ctx.render(cam, prm)) for cam, prm in jobs]
        streams = [torch.cuda.Stream(dev) for _ in range(3)]
        outs = [(torch.zeros_like(a), torch.zeros_like(b)) for a, b in refs]
        ctx.sync()
        for k, ((cam, prm), (lin_d, rgba_d)) in enumerate(zip(jobs, outs)):
            ctx.set_stream(None if k == 3 else streams[k % 3].cuda_stream)
            ctx.render_into(cam, prm, lin_d.data_ptr(), rgba_d.data_ptr())
        ctx.sync()
        torch.cuda.synchronize(dev)
        for k, ((lin_d, rgba_d), (lin, rgba)) in enumerate(zip(outs, refs)):
            assert torch_equal(lin_d.view(torch.int32), lin.view(torch.int32)) and torch_equal(rgba_d, rgba), k
    finally:
        ctx.set_stream(None)
        ctx.close()


def _oren_nayar_cornell(pt):
    """C2 with every Lambertian surface OrenNayar (material.rs:166-296), rough and smooth: a scene without Mirror
    surfaces, which takes the regenerating kernel compiled without the GGX code by default (round 3)."""
    objs = list(pt.builtin_scene(2))
    for k, o in enumerate(objs):
        if o.mat_tag == 0:
            o.mat_tag = 3
            o.mat[3] = [0.0, 0.3, 0.6, 1.0][k % 4]
    return (pt._lib.PtObject * len(objs))(*objs)


@pytest.mark.parametrize("scene,exact", [(2, 0), (2, 1), (1, 0), (1, 1), ("oren_nayar", 0), ("oren_nayar", 1)])
def test_level0_forms_give_the_same_film(pt, gpu_ctx, scene, exact):
    """The level-0 launch of a large batch has two forms: the queue form (k_paths: state through HBM once per vertex,
    in-place compaction) and the regenerating form (k_paths_regen: a path stays in its lane's registers, a lane whose path
    ends takes the batch's next one from the chunk counters).  Which lane traces which path -- and in the regenerating
    form that depends on the timing of the atomics -- must not matter: both films, the vertex / shadow-ray counts and
    the deepest vertex are identical, for the diffuse scene (regeneration is the default there) and for the reference
    scene with its glass sphere, in both arithmetic modes, for several grid sizes of the regenerating launch (1
    workgroup; fewer and more than the device holds at once)."""
    gpu_ctx.upload(_oren_nayar_cornell(pt) if scene == "oren_nayar" else pt.builtin_scene(scene))
    cam = pt.camera_new(width=1024, height=640)
    prm = pt.default_params(spp=8, exact_math=exact)                  # 5.2 M paths: above the hand-off threshold
    try:
        gpu_ctx.set_tuning(level0_form=1)
        ref, ref8 = gpu_ctx.render(cam, prm)
        base = gpu_ctx.stats()
        assert base.bounce_launches == 2
        # 3 = the regenerating form that sets Mirror vertices aside and shades them in batches (k_paths_regen_split; the
        # default for the reference scene; for the diffuse scene the request falls back to the queue form)
        for form, wgs, eb in [(2, 0, 0), (2, 1, 0), (2, 333, 3), (2, 4000, 64), (3, 0, 0), (3, 1, 0), (3, 333, 0), (3, 4000, 0), (0, 0, 0)]:
            gpu_ctx.set_tuning(level0_form=form, regen_workgroups=wgs, export_below=eb)
            lin, rgba = gpu_ctx.render(cam, prm)
            st = gpu_ctx.stats()
            assert torch_equal(lin, ref) and torch_equal(rgba, ref8), (form, wgs)
            assert (st.vertices, st.shadow_rays, st.max_depth_reached) == (base.vertices, base.shadow_rays, base.max_depth_reached)
            if form == 0:     # the default is a regenerating form for all three scenes: one launch, no continuation launch
                assert st.bounce_launches == 1
    finally:
        gpu_ctx.set_tuning()


@pytest.mark.parametrize("scene,form", [(2, 2), (1, 3), (1, 2)])
def test_regenerating_form_with_ragged_batches_and_bands(pt, gpu_ctx, scene, form):
    """Batches whose path count is not a multiple of 64, several of them, a row band, a sample offset: the regenerating
    forms (2: k_paths_regen; 3: k_paths_regen_split, the reference scene's default) equal the queue form."""
    gpu_ctx.upload(pt.builtin_scene(scene))
    cam = pt.camera_new(width=1023, height=517)
    for kw in (dict(spp=19, max_paths_in_flight=1023 * 517 * 9), dict(spp=20, band_rows=100, band_index=1, band_count=2),
               dict(spp=7, spp_offset=1000003, max_paths_in_flight=1023 * 517 * 3)):
        prm = pt.default_params(**kw)
        try:
            gpu_ctx.set_tuning(level0_form=1)
            ref, ref8 = gpu_ctx.render(cam, prm)
            base = gpu_ctx.stats()
            gpu_ctx.set_tuning(level0_form=form)
            lin, rgba = gpu_ctx.render(cam, prm)
            st = gpu_ctx.stats()
        finally:
            gpu_ctx.set_tuning()
        assert torch_equal(lin, ref) and torch_equal(rgba, ref8), kw
        assert (st.vertices, st.shadow_rays, st.max_depth_reached, st.batches) == (base.vertices, base.shadow_rays, base.max_depth_reached, base.batches)


def test_statistics_add_up_over_pipelined_renders(pt, gpu_ctx):
    """PtStats cover the renders enqueued since they were last collected: one render per synchronisation gives that render's
    counters; three renders enqueued behind ONE synchronisation give their sums (vertices, shadow rays, samples, launches,
    batches; the launch times of all of them with profile = 1), and the film is the last render's.  A new scene starts afresh."""
    import torch
    gpu_ctx.upload(pt.builtin_scene(1))
    cam = pt.camera_new(width=256, height=256)
    prms = [pt.default_params(spp=6, profile=1), pt.default_params(spp=6, spp_offset=6, profile=1), pt.default_params(spp=9, spp_offset=50, profile=1)]
    singles = []
    for prm in prms:
        lin, rgba = gpu_ctx.render(cam, prm)                 # render + pt_sync
        singles.append((lin.clone(), gpu_ctx.stats()))
    dev = lin.device
    lin = torch.empty_like(lin); rgba = torch.empty((256, 256, 4), dtype=torch.uint8, device=dev)
    for prm in prms:
        gpu_ctx.render_into(cam, prm, lin.data_ptr(), rgba.data_ptr())       # no synchronisation in between
    gpu_ctx.sync()
    st = gpu_ctx.stats()
    assert torch.equal(lin, singles[-1][0])
    for name in ("vertices", "shadow_rays", "samples", "bounce_launches", "batches", "primary_launches", "primary_vertices"):
        assert getattr(st, name) == sum(getattr(s, name) for _, s in singles), name
    assert st.max_depth_reached == max(s.max_depth_reached for _, s in singles)
    assert st.primary_kernel_ms > 0 and abs(st.primary_kernel_ms - sum(s.primary_kernel_ms for _, s in singles)) < 0.5 * st.primary_kernel_ms
    # an invalid call between two pipelined renders fails with its message and leaves their statistics alone
    gpu_ctx.stats()
    gpu_ctx.render_into(cam, prms[0], lin.data_ptr(), rgba.data_ptr())
    with pytest.raises(pt._lib.PtError, match="spp"):
        gpu_ctx.render_into(cam, pt.default_params(spp=0), lin.data_ptr(), rgba.data_ptr())
    with pytest.raises(pt._lib.PtError, match="band_index"):
        gpu_ctx.render_into(cam, pt.default_params(spp=1, band_index=3, band_count=2), lin.data_ptr(), rgba.data_ptr())
    gpu_ctx.render_into(cam, prms[1], lin.data_ptr(), rgba.data_ptr())
    st2 = gpu_ctx.stats()
    for name in ("vertices", "shadow_rays", "samples", "bounce_launches", "batches"):
        assert getattr(st2, name) == getattr(singles[0][1], name) + getattr(singles[1][1], name), name
    assert torch.equal(lin, singles[1][0])
    # collected: the next render starts from zero again; so does a new scene
    lin2, _ = gpu_ctx.render(cam, prms[0])
    assert gpu_ctx.stats().vertices == singles[0][1].vertices
    gpu_ctx.render_into(cam, prms[0], lin.data_ptr(), rgba.data_ptr())
    gpu_ctx.upload(pt.builtin_scene(1))
    lin3, _ = gpu_ctx.render(cam, prms[0])
    assert gpu_ctx.stats().vertices == singles[0][1].vertices and torch.equal(lin3, lin2)


@pytest.mark.parametrize("scene", [1, 2])
def test_tiny_images_with_many_samples_take_the_regenerating_forms(pt, gpu_ctx, scene):
    """Batches above 2^17 paths take a regenerating form whatever the image size: images of a few pixels with tens of
    thousands of samples (a 64-path chunk then spans many samples of the same pixels; s_local up to 65534; two batches
    beyond that; a row band) equal the queue form -- films, counters, deepest vertex."""
    gpu_ctx.upload(pt.builtin_scene(scene))
    try:
        for (w, h, spp, kw) in [(2, 2, 40000, {}), (3, 5, 20000, {}), (7, 3, 65535, {}), (2, 2, 70000, {}),
                                (33, 17, 700, dict(band_rows=5, band_index=1, band_count=2))]:
            cam = pt.camera_new(width=w, height=h)
            prm = pt.default_params(spp=spp, **kw)
            gpu_ctx.set_tuning(level0_form=1)
            ref, ref8 = gpu_ctx.render(cam, prm)
            base = gpu_ctx.stats()
            gpu_ctx.set_tuning()
            lin, rgba = gpu_ctx.render(cam, prm)
            st = gpu_ctx.stats()
            assert torch_equal(lin, ref) and torch_equal(rgba, ref8), (w, h, spp)
            assert (st.vertices, st.shadow_rays, st.max_depth_reached, st.batches) == (base.vertices, base.shadow_rays, base.max_depth_reached, base.batches)
    finally:
        gpu_ctx.set_tuning()


def torch_equal(a, b):
    import torch
    return torch.equal(a, b)


# ---------------------------------------------------------------- multi-GPU behind the C ABI (one device here)
def test_multi_gpu_entry_with_one_device_over_rccl(pt, gpu_ctx):
    """pt_multi_*: contexts + ncclCommInitAll + ONE ncclGather of the packed film + row permutation, with the single
    device of this box: the whole N-device code path over real RCCL.  The frame equals the single-context render
    bit for bit, for several band heights (interleaved bands, ragged last band) and for the one-shot entries."""
    objs = pt.builtin_scene(1)
    cam = pt.camera_new(width=200, height=117)
    prm = pt.default_params(spp=6)
    gpu_ctx.upload(objs)
    ref, ref8 = gpu_ctx.render(cam, prm)
    ref, ref8 = ref.cpu().numpy(), ref8.cpu().numpy()
    m = pt.Multi([0])
    try:
        m.upload(objs)
        for band_rows in (0, 1, 7, 64, 500):
            lin, rgba = m.render_host(cam, pt.default_params(spp=6, band_rows=band_rows))
            assert np.array_equal(lin, ref) and np.array_equal(rgba, ref8), band_rows
        st = m.stats()
        assert st.samples == 200 * 117 * 6
        info = m.info()
        assert (info.n_devices, info.comm_count, info.threaded) == (1, 1, 0) and info.rccl_version >= 20000 and info.frames == 5
        # the host-thread form (default for more than one device) over the real communicator: the device's own thread makes
        # the render launches and the ncclGather call; three frames posted back to back, then ONE synchronisation
        import torch
        m.set_threads(True)
        assert m.info().threaded == 1
        dev = torch.device("cuda", 0)
        outs = [(torch.zeros((117, 200, 3), dtype=torch.float32, device=dev), torch.zeros((117, 200, 4), dtype=torch.uint8, device=dev)) for _ in range(3)]
        for k, (lin_d, rgba_d) in enumerate(outs):
            m.render_into(cam, pt.default_params(spp=6, band_rows=[7, 0, 64][k]), lin_d.data_ptr(), rgba_d.data_ptr())
        m.sync()
        for lin_d, rgba_d in outs:
            assert np.array_equal(lin_d.cpu().numpy(), ref) and np.array_equal(rgba_d.cpu().numpy(), ref8)
        assert m.stats().samples == 3 * 200 * 117 * 6 and m.info().frames == 8
        m.set_threads(False)
        lin, rgba = m.render_host(cam, pt.default_params(spp=6, band_rows=3))
        assert np.array_equal(lin, ref) and np.array_equal(rgba, ref8)
        # the exchange by copies (pt_multi_set_exchange: one DMA copy per device where the default form runs ncclGather), switched
        # on an object that has frames behind it, in both thread modes, frames posted back to back -- more of them than the ring of
        # send buffers holds -- and switched back: every frame the same bits
        frames_before = m.info().frames
        m.set_exchange("copy")
        assert m.info().exchange == pt._lib.PT_EXCHANGE_COPY
        for threads in (False, True):
            m.set_threads(threads)
            outs = [(torch.zeros((117, 200, 3), dtype=torch.float32, device=dev), torch.zeros((117, 200, 4), dtype=torch.uint8, device=dev)) for _ in range(11)]
            for k, (lin_d, rgba_d) in enumerate(outs):
                m.render_into(cam, pt.default_params(spp=6, band_rows=[7, 0, 64][k % 3]), lin_d.data_ptr(), rgba_d.data_ptr())
            m.sync()
            for lin_d, rgba_d in outs:
                assert np.array_equal(lin_d.cpu().numpy(), ref) and np.array_equal(rgba_d.cpu().numpy(), ref8)
        assert m.info().frames == frames_before + 22
        m.set_threads(False)
        m.set_exchange("rccl")
        assert m.info().exchange == pt._lib.PT_EXCHANGE_RCCL
        lin, rgba = m.render_host(cam, pt.default_params(spp=6, band_rows=5))
        assert np.array_equal(lin, ref) and np.array_equal(rgba, ref8)
        with pytest.raises(pt._lib.PtError, match="unknown mode"):
            pt._lib.check(pt._lib.lib().pt_multi_set_exchange(m._h, 7))
    finally:
        m.close()
    lin, rgba = pt.render_multi([0], cam, objs, prm)
    assert np.array_equal(lin, ref) and np.array_equal(rgba, ref8)
    lin, rgba = pt.render_host(cam, objs, pt.default_params(spp=6, n_devices=1))
    assert np.array_equal(lin, ref)
    with pytest.raises(pt._lib.PtError, match="twice"):
        pt.Multi([0, 0])
    pt._lib.lib().pt_shutdown()


@pytest.mark.parametrize("form", ["rccl", "copy", "rccl-threads", "shared3", "shared3-threads"])
def test_random_frame_sequences_through_the_multi_device_object(pt, gpu_ctx, form):
    """Frames of random size, band height and sample count posted back to back through pt_multi_* -- growing and shrinking tiles
    (re-allocation of the send ring and the receive buffer between frames in flight, a ring of a different length for a large
    frame), more frames than the ring holds -- over the real ncclGather with the one device, the exchange by copies, host threads, and
    three contexts sharing the device: every frame equals the plain render of the same job."""
    import torch
    rng = np.random.default_rng(len(form))
    dev = torch.device("cuda", 0)
    objs = pt.builtin_scene(2)
    gpu_ctx.upload(objs)
    jobs = []
    for k in range(14):
        W, H = [(96, 64), (200, 117), (512, 300), (33, 9), (640, 480)][int(rng.integers(0, 5))]
        jobs.append((pt.camera_new(width=W, height=H), dict(spp=int(rng.choice([2, 4, 6])), spp_offset=int(rng.integers(0, 100)),
                                                           band_rows=int(rng.choice([0, 1, 7, 32])))))
    refs = []
    for cam, kw in jobs:
        lin, rgba = gpu_ctx.render(cam, pt.default_params(spp=kw["spp"], spp_offset=kw["spp_offset"]))
        refs.append((lin.clone(), rgba.clone()))
    m = pt.Multi([0, 0, 0], shared_device=0) if form.startswith("shared3") else pt.Multi([0])
    try:
        m.upload(objs)
        if form == "copy":
            m.set_exchange("copy")
        if form.endswith("threads"):
            m.set_threads(True)
        outs = [(torch.zeros_like(a), torch.zeros_like(b)) for a, b in refs]
        for (cam, kw), (lin_d, rgba_d) in zip(jobs, outs):
            m.render_into(cam, pt.default_params(**kw), lin_d.data_ptr(), rgba_d.data_ptr())
        m.sync()
        for k, ((lin_d, rgba_d), (lin, rgba)) in enumerate(zip(outs, refs)):
            assert torch_equal(lin_d.view(torch.int32), lin.view(torch.int32)) and torch_equal(rgba_d, rgba), (k, jobs[k][1])
        assert m.stats().samples == sum(c.width * c.height * kw["spp"] for c, kw in jobs)
    finally:
        m.close()


@pytest.mark.parametrize("n", [2, 3, 8])
def test_multi_gpu_partition_pack_unpack_for_n_devices(pt, gpu_ctx, n):
    """The band partition, the 16 B/pixel pack and the row permutation of pt_multi_* for n = 2, 3, 8 devices, emulated on
    the one GPU of this box (pt_debug_multi_emulate: device-to-device copies where the real path runs ncclGather):
    the frame is bitwise independent of n, also with ragged bands (117 rows) and devices that own nothing."""
    objs = pt.builtin_scene(2)
    gpu_ctx.upload(objs)
    cam = pt.camera_new(width=160, height=117)
    ref, ref8 = gpu_ctx.render(cam, pt.default_params(spp=5))
    ref, ref8 = ref.cpu().numpy(), ref8.cpu().numpy()
    for band_rows in (0, 1, 10, 117):
        lin, rgba = gpu_ctx.multi_emulate(n, cam, pt.default_params(spp=5, band_rows=band_rows))
        assert np.array_equal(lin, ref) and np.array_equal(rgba, ref8), (n, band_rows)


@pytest.mark.parametrize("scene,kw", [(2, {}), (1, dict(band_rows=5, band_index=1, band_count=3)), (2, dict(max_paths_in_flight=300000)),
                                      (1, dict(exact_math=1)), (4, dict(accel=1))])
def test_packed_render_is_the_two_planes_packed(pt, gpu_ctx, scene, kw):
    """pt_render_device_packed: the film resolve writes the 16 B/pixel send record of the multi-GPU gather itself.  Bit for bit
    what pt_film_pack makes of the two planes of pt_render_device -- whole image, a row-band tile, several sample batches
    (the f64 film sums in between), exact arithmetic, the BVH kernels."""
    import torch
    gpu_ctx.upload(pt.builtin_scene(scene, 300 if scene == 4 else 0))
    cam = pt.camera_new(width=257, height=131)
    prm = pt.default_params(spp=7, **kw)
    lin, rgba = gpu_ctx.render(cam, prm)
    rows = lin.shape[0]
    packed = torch.zeros((rows, 257, 16), dtype=torch.uint8, device=lin.device)
    gpu_ctx.render_packed_into(cam, prm, packed.data_ptr())
    gpu_ctx.sync()
    want = torch.empty_like(packed)
    pt._lib.check(pt._lib.lib().pt_film_pack(None, lin.data_ptr(), rgba.data_ptr(), rows * 257, want.data_ptr()))
    torch.cuda.synchronize()
    assert torch.equal(packed, want)
    assert torch.equal(packed[..., :12].contiguous().view(torch.float32).reshape(rows, 257, 3), lin) and torch.equal(packed[..., 12:], rgba)
    with pytest.raises(pt._lib.PtError, match="null"):
        gpu_ctx.render_packed_into(cam, prm, 0)
    with pytest.raises(pt._lib.PtError, match="aligned"):
        gpu_ctx.render_packed_into(cam, prm, packed.data_ptr() + 4)


@pytest.mark.parametrize("n", [2, 8])
def test_n_device_frames_from_host_threads_posted_back_to_back(pt, gpu_ctx, n):
    """The single-process multi-device object with n > 1 "devices" on this one-GPU box: n contexts on device 0
    (pt_debug_multi_create_shared; device-to-device copies stand in for ncclGather, everything else is the real object).
    One host thread per device feeds its stream; SIX frames (different sample offsets, band heights, with and without the
    RGBA8 plane) are posted back to back and synchronised ONCE: every frame equals the one-context render of the same
    parameters bit for bit, the statistics are the sums, and the one-thread form gives the same frames."""
    import torch
    objs = pt.builtin_scene(1)
    gpu_ctx.upload(objs)
    cam = pt.camera_new(width=192, height=117)
    dev = torch.device("cuda", 0)
    jobs = [dict(spp=5, spp_offset=3 * k, band_rows=[0, 1, 10, 117, 4, 33][k]) for k in range(6)]
    refs, verts = [], 0
    for j in jobs:
        lin, rgba = gpu_ctx.render(cam, pt.default_params(spp=j["spp"], spp_offset=j["spp_offset"]))
        verts += gpu_ctx.stats().vertices
        refs.append((lin.clone(), rgba.clone()))
    m = pt.Multi([0] * n, shared_device=0)
    try:
        m.upload(objs)
        for threaded in (True, False):
            m.set_threads(threaded)
            outs = [(torch.zeros((117, 192, 3), dtype=torch.float32, device=dev), torch.zeros((117, 192, 4), dtype=torch.uint8, device=dev)) for _ in jobs]
            for k, (j, (lin_d, rgba_d)) in enumerate(zip(jobs, outs)):
                m.render_into(cam, pt.default_params(**j), lin_d.data_ptr(), rgba_d.data_ptr() if k != 2 else 0)
            m.sync()
            for k, ((lin_d, rgba_d), (lin, rgba)) in enumerate(zip(outs, refs)):
                assert torch.equal(lin_d, lin), (n, threaded, k)
                assert k == 2 or torch.equal(rgba_d, rgba), (n, threaded, k)
            st = m.stats()
            assert st.samples == 6 * 192 * 117 * 5 and st.vertices == verts
            info = m.info()
            assert (info.n_devices, info.comm_count, info.threaded) == (n, 0, 1 if threaded else 0)
            assert info.enqueue_us_max > 0 and info.enqueue_us_sum >= info.enqueue_us_max
        # a larger frame re-allocates the exchange buffers between frames: still right
        big = pt.camera_new(width=320, height=200)
        m.set_threads(True)
        lin_h, rgba_h = m.render_host(big, pt.default_params(spp=3))
        ref, ref8 = gpu_ctx.render(big, pt.default_params(spp=3))
        assert np.array_equal(lin_h, ref.cpu().numpy()) and np.array_equal(rgba_h, ref8.cpu().numpy())
    finally:
        m.close()


def test_large_pixel_list_takes_the_tail_hand_off(pt, gpu_ctx):
    """A pixel list large enough for the two-launch form (level-0 launch exports, continuation launch reads its count
    on the device) in the LIST variants of the kernels: every pixel of a 512 x 512 image, shuffled, 24 spp = 6.3 M paths
    (> 2^22).  Equals the full film, pixel by pixel."""
    gpu_ctx.upload(pt.builtin_scene(1))
    cam = pt.camera_new(width=512, height=512)
    prm = pt.default_params(spp=24)
    full, full8 = gpu_ctx.render(cam, prm)
    rng = np.random.default_rng(12)
    order = rng.permutation(512 * 512)
    xy = np.stack([order % 512, order // 512], 1).astype(np.uint32)
    lin, rgba, _ = gpu_ctx.render_pixels(cam, prm, xy)
    st = gpu_ctx.stats()
    assert st.bounce_launches == 2 and st.samples == 512 * 512 * 24
    assert np.array_equal(lin, full.cpu().numpy()[xy[:, 1], xy[:, 0]]) and np.array_equal(rgba, full8.cpu().numpy()[xy[:, 1], xy[:, 0]])


# ---------------------------------------------------------------- the per-process form's film exchange on the GPU
@pytest.mark.gpu
@pytest.mark.parametrize("H,W,band,n", [(23, 20, 3, 3), (16, 8, 16, 2), (9, 5, 1, 4), (128, 64, 16, 8), (37, 33, 4, 1)])
def test_film_pack_and_unpack_kernels_of_the_per_process_form(pt, H, W, band, n):
    """pathtrace_amd/dist.py (one process per GPU, gather by torch.distributed) packs and unpacks the film with the
    library's kernels (pt_film_pack / pt_film_unpack, the ones pt_multi_* runs around its ncclGather) when the tiles are
    on a GPU.  Here the ranks' tiles are packed one after another on this GPU, laid out as the gather would deliver them,
    and unpacked: both planes come back bit for bit, for ragged bands, ranks without a band and a single rank -- and
    equal what the CPU form of the same class (the one the gloo tests run) produces."""
    import torch
    from pathtrace_amd.dist import FilmGather
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(H * 131 + W)
    lin = torch.randn((H, W, 3), generator=g, dtype=torch.float32)
    lin[0, 0, 0] = float("inf"); lin[H - 1, W - 1, 2] = -0.0                   # bit patterns, not values
    rgba = torch.randint(0, 256, (H, W, 4), generator=g, dtype=torch.uint8)
    rows = [pt.tile_row_indices(H, band, r, n) for r in range(n)]
    for device in (dev, torch.device("cpu")):
        ranks = [FilmGather(H, W, band, r, n, device) for r in range(n)]
        root = ranks[0]
        recv = torch.zeros((n, root.max_rows, W, 16), dtype=torch.uint8, device=device)
        for r in range(n):
            idx = torch.as_tensor(rows[r], dtype=torch.int64)
            ranks[r]._pack(lin[idx].to(device), rgba[idx].to(device))
            recv[r] = ranks[r].send
        out_lin, out_rgba = root._unpack(recv.view(n * root.max_rows, W, 16))
        assert torch.equal(out_lin.cpu().view(torch.int32), lin.view(torch.int32)), str(device)
        assert torch.equal(out_rgba.cpu(), rgba), str(device)
    with pytest.raises(pt._lib.PtError, match="max_rows"):
        pt._lib.check(pt._lib.lib().pt_film_unpack(None, recv.data_ptr(), W, H, band, n, 0, out_lin.data_ptr(), None))


@pytest.mark.gpu
def test_streamed_film_exchange_over_rccl_keeps_every_frame(pt):
    """The per-process form's exchange on a GPU (pathtrace_amd/dist.py, FilmGather): a ring of send tiles, and gather + wait +
    row permutation on a stream of their own, so that a late collective holds up nothing but the frame that wants its tile
    back.  One rank over the real backend (RCCL, world size 1, the gather issued regardless): 21 different frames -- more than
    two rounds of the ring -- are rendered by the film resolve straight into the ring's tiles and started back to back without a
    host wait; the frames collected on the way and the last one equal plain renders bit for bit, and the tiles are taken in turn."""
    import socket
    import torch
    import torch.distributed as dist
    from pathtrace_amd.dist import FilmGather
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    ctx = None
    try:
        dev = torch.device("cuda", 0)
        H, W, band = 96, 64, 8
        cam = pt.camera_new(width=W, height=H)
        ctx = pt.Context(0)
        ctx.upload(pt.builtin_scene(2))
        frames = 21
        prms = [pt.default_params(spp=8, spp_offset=8 * k, band_rows=band, band_index=0, band_count=1) for k in range(frames)]
        ref = [ctx.render(cam, p) for p in prms]
        assert not torch.equal(ref[0][0], ref[1][0])
        fg = FilmGather(H, W, band, 0, 1, dev, always_collective=True)
        assert len(fg._sends) == FilmGather.RING_MAX and fg._streamed
        ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
        ptrs, checked = [], 0
        for k in range(frames):
            ptrs.append(fg.send.data_ptr())
            ctx.render_packed_into(cam, prms[k], fg.send.data_ptr())
            fg.start_prepacked()
            if k % 5 == 4 or k + 1 == frames:
                lin, rgba = fg.finish()
                torch.cuda.current_stream(dev).synchronize()
                assert torch.equal(lin.view(torch.int32), ref[k][0].view(torch.int32)), k
                assert torch.equal(rgba, ref[k][1]), k
                checked += 1
        assert checked == 5 and fg.finish() == (None, None)
        ring = FilmGather.RING_MAX
        assert len(set(ptrs[:ring])) == ring and ptrs[ring:2 * ring] == ptrs[:ring] and ptrs[2 * ring] == ptrs[0]
        ctx.sync()
        # the forms that pack two existing planes (start / the blocking call) and the one-call entry of the per-process form
        from pathtrace_amd.dist import render_distributed
        fg2 = FilmGather(H, W, band, 0, 1, dev, always_collective=True)
        for k in (3, 11):
            fg2.start(ref[k][0], ref[k][1])
        lin, rgba = fg2.finish()
        torch.cuda.current_stream(dev).synchronize()
        assert torch.equal(lin.view(torch.int32), ref[11][0].view(torch.int32)) and torch.equal(rgba, ref[11][1])
        lin, rgba = fg2(ref[5][0], ref[5][1])
        torch.cuda.current_stream(dev).synchronize()
        assert torch.equal(lin.view(torch.int32), ref[5][0].view(torch.int32)) and torch.equal(rgba, ref[5][1])
        lin, rgba = render_distributed(ctx, cam, pt.default_params(spp=8, spp_offset=8 * 7), 0, 1, band_rows=band)
        torch.cuda.current_stream(dev).synchronize()
        assert torch.equal(lin.view(torch.int32), ref[7][0].view(torch.int32)) and torch.equal(rgba, ref[7][1])
    finally:
        if ctx is not None:
            ctx.set_stream(None)
            ctx.close()
        dist.destroy_process_group()
