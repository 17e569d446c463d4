"""Randomised scenes through the C ABI: spheres and triangles in arbitrary object order (so the LDS
scan sees many runs), all four materials, several lights, random cameras, integrators and roulette
policies.  exact_math = 1 must equal the f32 oracle bit for bit (NaNs included, all four materials); the default mode
must stay within the FP32 tolerance of it."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
F32, ITER = 32, 1


def random_scene(pt, rng, n_objs):
    specs = []
    for i in range(n_objs):
        kind = rng.integers(0, 2)
        if kind == 0:
            c = rng.uniform([-1, -1, -3], [1, 1, -1])
            shape = (0, list(c) + [float(rng.uniform(0.05, 0.5))])
        else:
            v0 = rng.uniform([-1.2, -1.2, -3.2], [1.2, 1.2, -0.8])
            shape = (1, list(v0) + list(v0 + rng.uniform(-1, 1, 3)) + list(v0 + rng.uniform(-1, 1, 3)))
        m = rng.integers(0, 10)
        if i < 2 or m == 0:
            mat = (1, list(rng.uniform(2, 20, 3)))                       # emissive (at least two lights)
        elif m <= 5:
            mat = (0, list(rng.uniform(0.1, 0.9, 3)))                    # lambert
        elif m <= 7:
            mat = (2, [float(rng.uniform(0.05, 0.6))] + list(rng.uniform(0.5, 1, 3)) +
                   [float(rng.choice([0.0, 1.0])), float(rng.uniform(1.1, 1.8))])   # GGX glass / metal
        elif m == 8:
            mat = (3, list(rng.uniform(0.2, 0.9, 3)) + [float(rng.uniform(0, 1))])  # oren-nayar
        else:
            mat = (1, [0.0, 0.0, 0.0])                                   # black "emissive": not a light (Q9)
        specs.append((shape[0], shape[1], mat[0], mat[1]))
    # a big enclosing diffuse sphere so that paths keep bouncing
    specs.append((0, [0.0, 0.0, -2.0, 6.0], 0, [0.7, 0.7, 0.7]))
    return pt.make_objects(specs)


@pytest.mark.parametrize("seed", range(int(os.environ.get("PT_FUZZ_SEEDS", "8"))))   # PT_FUZZ_SEEDS=64 for a longer campaign
def test_random_scene_matches_f32_oracle(pt, orc, gpu_ctx, seed):
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(3, 40)) if seed % 8 < 6 else int(rng.integers(150, 400))   # two in eight exceed one LDS blob
    objs = random_scene(pt, rng, n)
    if seed % 4 == 3:                     # Lambertian / emissive only: the kernels' DIFFUSE variants
        for o in objs:
            if o.mat_tag not in (0, 1):
                o.mat_tag = 0
                o.mat[0], o.mat[1], o.mat[2] = 0.6, 0.5, 0.4
    w, h = int(rng.integers(8, 48)), int(rng.integers(8, 48))
    if rng.uniform() < 0.5:
        cam = pt.camera_new(width=w, height=h, fov_degrees=float(rng.uniform(25, 60)))
    else:
        cam = pt.camera_look_at(tuple(rng.uniform([-0.8, -0.8, 0.5], [0.8, 0.8, 2.5])), (0.0, 0.0, -2.0), (0.0, 1.0, 0.0),
                                w, h, float(rng.uniform(25, 60)))
    prm = pt.default_params(spp=int(rng.integers(1, 6)), integrator=int(rng.integers(0, 2)),
                            min_depth=int(rng.integers(0, 6)), max_depth=int(rng.integers(6, 30)), exact_math=1, accel=0)
    gpu_ctx.upload(objs)
    lin, rgba = gpu_ctx.render(cam, prm)
    st = gpu_ctx.stats()
    ref, ref8, cnt = orc.render(cam, objs, prm, F32, ITER, 8)
    got = lin.cpu().numpy()
    # every material, OrenNayar included (its azimuth term is trig-free in the f32 specification)
    assert np.array_equal(got, ref.astype(np.float32), equal_nan=True), \
        f"seed {seed}: {(got != ref.astype(np.float32)).any(-1).sum()} pixels differ"
    assert np.array_equal(rgba.cpu().numpy(), ref8)
    assert st.vertices == cnt["vertices"] and st.shadow_rays == cnt["shadow_rays"]
    prm.accel = 1                         # the BVH path on the same scene: the same film
    lin_b, rgba_b = gpu_ctx.render(cam, prm)
    assert np.array_equal(lin_b.cpu().numpy(), got, equal_nan=True) and np.array_equal(rgba_b.cpu().numpy(), ref8)
    assert gpu_ctx.stats().vertices == cnt["vertices"]
    prm.accel = 0
    fin = np.isfinite(ref).all(-1)
    prm.exact_math = 0
    fast, _ = gpu_ctx.render(cam, prm)
    for img in [fast.cpu().numpy()]:
        g = img.astype(np.float64)
        ok = (np.abs(g - ref) <= 1e-3 + 1e-2 * np.abs(ref)).all(-1)
        assert ok[fin].mean() >= 0.97, (seed, ok[fin].mean())


@pytest.mark.parametrize("seed", range(int(os.environ.get("PT_FUZZ_DIRECT_SEEDS", "6"))))
def test_random_scene_regenerating_default_matches_f32_oracle(pt, orc, gpu_ctx, seed):
    """The regenerating kernels pinned DIRECTLY to the f32 oracle (the test below pins them to the queue form, which the test
    above pins to the oracle): random scenes in LDS at 200 x 200 x 4..6 spp = 160 000 .. 240 000 paths, above the 2^17
    threshold from which a batch takes a regenerating form by default -- k_paths_regen compiled for the scene's material set
    (diffuse only / no Mirror / every material) or k_paths_regen_split where a minority of the objects is Mirror.  exact_math = 1:
    film, RGBA8, vertex and shadow-ray counts equal the oracle's bit for bit, NaNs included."""
    rng = np.random.default_rng(5000 + seed)
    objs = random_scene(pt, rng, int(rng.integers(3, 40)))
    kind = seed % 3
    for o in objs:
        if kind == 0 and o.mat_tag not in (0, 1):          # Lambertian / emissive only
            o.mat_tag = 0
            o.mat[0], o.mat[1], o.mat[2] = 0.6, 0.5, 0.4
        if kind == 1 and o.mat_tag == 2:                    # no Mirror: the GGX objects become OrenNayar
            o.mat_tag = 3
            o.mat[0], o.mat[1], o.mat[2], o.mat[3] = 0.6, 0.5, 0.4, 0.5
    cam = pt.camera_look_at(tuple(rng.uniform([-0.8, -0.8, 0.5], [0.8, 0.8, 2.5])), (0.0, 0.0, -2.0), (0.0, 1.0, 0.0),
                            200, 200, float(rng.uniform(25, 60)))
    prm = pt.default_params(spp=int(rng.integers(4, 7)), integrator=int(rng.integers(0, 2)),
                            min_depth=int(rng.integers(0, 6)), max_depth=int(rng.integers(6, 30)), exact_math=1, accel=0)
    gpu_ctx.upload(objs)
    lin, rgba = gpu_ctx.render(cam, prm)
    st = gpu_ctx.stats()
    assert st.bounce_launches == 1 and st.batches == 1          # one regenerating launch, no continuation launch
    ref, ref8, cnt = orc.render(cam, objs, prm, F32, ITER, 8)
    got = lin.cpu().numpy()
    assert np.array_equal(got, ref.astype(np.float32), equal_nan=True), \
        f"seed {seed}: {(got != ref.astype(np.float32)).any(-1).sum()} pixels differ"
    assert np.array_equal(rgba.cpu().numpy(), ref8)
    assert st.vertices == cnt["vertices"] and st.shadow_rays == cnt["shadow_rays"] and st.max_depth_reached == cnt["max_depth"]


@pytest.mark.parametrize("seed", range(int(os.environ.get("PT_FUZZ_REGEN_SEEDS", "6"))))
def test_random_scene_regenerating_form_equals_queue_form(pt, gpu_ctx, seed):
    """The regenerating level-0 kernel on random scenes (all four materials or diffuse only, several lights, random camera,
    integrator and roulette limits, both arithmetic modes) at sizes above the hand-off threshold: the film, the counters and
    the deepest vertex equal those of the queue form -- which the test above pins to the f32 oracle on the same kind of
    scene -- NaNs included.  Which lane traces which path there depends on the timing of the chunk counters."""
    import torch
    rng = np.random.default_rng(77000 + seed)
    objs = random_scene(pt, rng, int(rng.integers(3, 40)))
    if seed % 2 == 1:                     # Lambertian / emissive only: the DIFFUSE variant (the product default for such scenes)
        for o in objs:
            if o.mat_tag not in (0, 1):
                o.mat_tag = 0
                o.mat[0], o.mat[1], o.mat[2] = 0.6, 0.5, 0.4
    w, h = int(rng.integers(500, 800)), int(rng.integers(400, 700))
    spp = -(-(1 << 22) // (w * h)) + int(rng.integers(1, 6))              # more than 2^22 paths in the batch
    cam = pt.camera_look_at(tuple(rng.uniform([-0.8, -0.8, 0.5], [0.8, 0.8, 2.5])), (0.0, 0.0, -2.0), (0.0, 1.0, 0.0),
                            w, h, float(rng.uniform(25, 60)))
    prm = pt.default_params(spp=spp, integrator=int(rng.integers(0, 2)), min_depth=int(rng.integers(0, 6)),
                            max_depth=int(rng.integers(6, 30)), exact_math=int(seed % 3 == 0), accel=0)
    gpu_ctx.upload(objs)
    try:
        gpu_ctx.set_tuning(level0_form=1)
        ref, ref8 = gpu_ctx.render(cam, prm)
        base = gpu_ctx.stats()
        eb = int(rng.choice([0, 0, 5, 64]))
        gpu_ctx.set_tuning(level0_form=2, regen_workgroups=int(rng.choice([0, 0, 97, 3000])), export_below=eb)
        lin, rgba = gpu_ctx.render(cam, prm)
        st = gpu_ctx.stats()
        # the forms really were the two: level-0 + continuation launch; one regenerating launch whose waves run dry
        # (with a hand-over threshold: + the continuation launch)
        assert base.bounce_launches == 2 and st.bounce_launches == (1 if eb == 0 else 2)
        # the form that batches the Mirror vertices of a wave (k_paths_regen_split), whatever share of the scene is Mirror
        gpu_ctx.set_tuning(level0_form=3, regen_workgroups=int(rng.choice([0, 0, 97, 3000])))
        lin3, rgba3 = gpu_ctx.render(cam, prm)
        st3 = gpu_ctx.stats()
    finally:
        gpu_ctx.set_tuning()
    a, b = lin.cpu().numpy(), ref.cpu().numpy()
    assert np.array_equal(a, b, equal_nan=True), f"seed {seed}: {(a != b).any(-1).sum()} pixels differ"
    assert torch.equal(rgba, ref8)
    assert (st.vertices, st.shadow_rays, st.max_depth_reached) == (base.vertices, base.shadow_rays, base.max_depth_reached)
    assert np.array_equal(lin3.cpu().numpy(), b, equal_nan=True) and torch.equal(rgba3, ref8), f"seed {seed}: split form"
    assert (st3.vertices, st3.shadow_rays, st3.max_depth_reached) == (base.vertices, base.shadow_rays, base.max_depth_reached)
