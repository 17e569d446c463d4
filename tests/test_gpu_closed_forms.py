"""Function-level pins that do NOT go through oracle/: the per-vertex device functions behind the C ABI (pt_debug_*) against numpy f64
restatements of the reference's formulas, written here from the cited lines, and against closed forms the formulas imply.

tests/test_gpu_quadrature.py does this for the integrator (MIS weights, roulette) and GGX; this file covers what lies beneath it:

  (a) World::hit_scene (world.rs:270-290) over SphereShape::hit (shape.rs:53-88: both roots, the acceptance window t_min <= root <= t_max)
      and TriangleShape::hit (shape.rs:160-198: Moeller-Trumbore, |a| < 1e-8 parallel, u in [0, 1], v >= 0, u + v <= 1): object and
      distance of 200 000 random rays against a direct f64 evaluation, for spheres, single triangles and the triangle PAIRS the
      kernels test together; HitRecord::new (base.rs:19-33): point, face-forwarded normal, front_face.
  (b) Shape::sample_surface_from_point: the sphere's cone sampling (shape.rs:91-145) -- pdf = 1 / (2 pi (1 - cos theta_max)), points
      on the near cap, directions uniform in the cone (mean cosine (1 + cos theta_max) / 2) -- and the triangle's area sampling
      (shape.rs:200-242): the point v0 + (1 - sqrt r1)(v1 - v0) + r2 sqrt r1 (v2 - v0) draw for draw, pdf_omega = d^2 / (A cos), and
      E[1 / pdf_omega] = the triangle's solid angle by Van Oosterom & Strackee's closed form.
  (c) Lambertian and OrenNayar bsdf_pdf (material.rs:86-91, 221-265, coefficients :182-193) and their cosine-weighted sampler
      (material.rs:93-119): values against the formulas (the reference's atan2 form of cos(phi_i - phi_o), not the kernels' atan2-free
      one), draw for draw directions, and the energy integral of the Lambertian = albedo.
  (d) Camera::new / Camera::look_at / get_ray_with_offset (camera.rs:50-147) as World::render_pixel calls it (world.rs:297-299).
  (e) World::sample_light_point (world.rs:251-267) over the light list of world.rs:213-225: pick, point, emission, pdf / n.

Tolerances are f32 ones against f64 (1e-4 relative, looser where a formula is ill-conditioned -- stated at the assertion).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SPH, TRI = 0, 1
LAMBERT, EMISSIVE, MIRROR, OREN = 0, 1, 2, 3
T_MIN = 1e-3


def _norm(v):
    return v / np.linalg.norm(v, axis=-1, keepdims=True)


# ------------------------------------------------------------------ (a) intersection
def _sphere_hit(o, d, c, r, t_min, t_max):
    """SphereShape::hit, shape.rs:53-88 (d need not be unit there; here it is).  -> t or inf"""
    oc = o - c
    a = (d * d).sum(-1)
    hb = (oc * d).sum(-1)
    cc = (oc * oc).sum(-1) - r * r
    disc = hb * hb - a * cc
    sq = np.sqrt(np.maximum(disc, 0.0))
    r0, r1 = (-hb - sq) / a, (-hb + sq) / a
    ok0 = (disc >= 0) & ~((r0 < t_min) | (t_max < r0))
    ok1 = (disc >= 0) & ~((r1 < t_min) | (t_max < r1))
    return np.where(ok0, r0, np.where(ok1, r1, np.inf))


def _triangle_hit(o, d, v0, v1, v2, t_min, t_max):
    """TriangleShape::hit, shape.rs:160-198.  -> t or inf"""
    e1, e2 = v1 - v0, v2 - v0
    h = np.cross(d, e2)
    a = (e1 * h).sum(-1)
    with np.errstate(divide="ignore", invalid="ignore"):
        f = 1.0 / a
        s = o - v0
        u = f * (s * h).sum(-1)
        q = np.cross(s, e1)
        v = f * (d * q).sum(-1)
        t = f * (e2 * q).sum(-1)
    ok = (np.abs(a) >= 1e-8) & (u >= 0.0) & (u <= 1.0) & ~((v < 0.0) | (u + v > 1.0)) & ~((t < t_min) | (t > t_max))
    return np.where(ok, t, np.inf)


def _hit_scene(specs, o, d, t_min=T_MIN):
    """World::hit_scene, world.rs:270-290: objects in order, each with t_max = the closest so far (so an equal t REPLACES)."""
    n = o.shape[0]
    best_t = np.full(n, np.inf)
    best_id = np.full(n, -1, dtype=np.int64)
    for i, (st, sv, _, _) in enumerate(specs):
        sv = np.asarray(sv, dtype=np.float64)
        if st == SPH:
            t = _sphere_hit(o, d, sv[:3], sv[3], t_min, best_t)
        else:
            t = _triangle_hit(o, d, sv[0:3], sv[3:6], sv[6:9], t_min, best_t)
        take = np.isfinite(t)
        best_t = np.where(take, t, best_t)
        best_id = np.where(take, i, best_id)
    return best_id, best_t


def _mixed_scene(rng, n_sph, n_tri, n_quads):
    specs = []
    for _ in range(n_sph):
        specs.append((SPH, list(rng.uniform(-2, 2, 3)) + [rng.uniform(0.15, 0.7)], LAMBERT, [0.7, 0.6, 0.5]))
    for _ in range(n_tri):
        v0 = rng.uniform(-2, 2, 3)
        specs.append((TRI, list(v0) + list(v0 + rng.uniform(-1.2, 1.2, 3)) + list(v0 + rng.uniform(-1.2, 1.2, 3)), LAMBERT, [0.5, 0.5, 0.5]))
    for _ in range(n_quads):        # two triangles that share v0 and the plane: what the kernels test as a pair
        v0 = rng.uniform(-2, 2, 3)
        a, b = rng.uniform(-1.5, 1.5, 3), rng.uniform(-1.5, 1.5, 3)
        specs.append((TRI, list(v0) + list(v0 + a) + list(v0 + a + b), LAMBERT, [0.5, 0.5, 0.5]))
        specs.append((TRI, list(v0) + list(v0 + a + b) + list(v0 + b), LAMBERT, [0.5, 0.5, 0.5]))
    order = rng.permutation(len(specs) - 2 * n_quads)
    head = [specs[i] for i in order]
    return head[: len(head) // 2] + specs[len(specs) - 2 * n_quads:] + head[len(head) // 2:]


@pytest.mark.parametrize("exact_math", [0, 1])
@pytest.mark.parametrize("accel", [0, 1])
def test_hit_scene_against_the_formulas_of_shape_rs(pt, gpu_ctx, exact_math, accel):
    rng = np.random.default_rng(20264)
    specs = _mixed_scene(rng, n_sph=14, n_tri=16, n_quads=5)
    gpu_ctx.upload(pt.make_objects(specs))
    n = 200_000
    o = rng.uniform(-3, 3, (n, 3))
    d = _norm(rng.normal(size=(n, 3)))
    d[: n // 4] = _norm(rng.uniform(-1.5, 1.5, (n // 4, 3)) - o[: n // 4])          # a quarter aimed into the crowd
    ref_id, ref_t = _hit_scene(specs, o, d)
    ids, ts = gpu_ctx.debug_hit_scene(np.concatenate([o, d], 1), t_min=T_MIN, exact_math=exact_math, accel=accel)
    same = ids == ref_id
    # f32 against f64: a ray that grazes a sphere or a triangle's edge within rounding may fall the other way
    assert same.mean() >= 0.9995, same.mean()
    hit = same & (ref_id >= 0)
    assert hit.sum() > n // 4
    rel = np.abs(ts[hit].astype(np.float64) - ref_t[hit]) / np.maximum(ref_t[hit], 1e-2)
    assert np.quantile(rel, 0.999) <= 1e-4 and rel.max() <= 5e-3, (np.quantile(rel, 0.999), rel.max())   # near-tangent sphere hits lose digits
    # the rays that disagree do so at a boundary: the f64 distances of the two candidates are close, or one of them grazes
    bad = np.flatnonzero(~same)
    for i in bad[:50]:
        cand = []
        for k in (ids[i], ref_id[i]):
            if k >= 0:
                sid, tk = _hit_scene([specs[k]], o[i:i + 1], d[i:i + 1])
                cand.append(tk[0])
        assert len(cand) < 2 or not np.isfinite(cand).all() or abs(cand[0] - cand[1]) <= 1e-3 * max(cand), (i, cand)


def test_hit_records_follow_base_rs(pt, gpu_ctx):
    rng = np.random.default_rng(77)
    specs = _mixed_scene(rng, n_sph=10, n_tri=10, n_quads=3)
    gpu_ctx.upload(pt.make_objects(specs))
    n = 50_000
    o = rng.uniform(-3, 3, (n, 3))
    d = _norm(rng.uniform(-1.5, 1.5, (n, 3)) - o)
    ref_id, ref_t = _hit_scene(specs, o, d)
    ids, rec = gpu_ctx.debug_hit_records(np.concatenate([o, d], 1), t_min=T_MIN)
    ok = (ids == ref_id) & (ref_id >= 0)
    assert ok.sum() > n // 3
    point = o + ref_t[:, None] * d
    outward = np.zeros((n, 3))
    for i, (st, sv, _, _) in enumerate(specs):
        sv = np.asarray(sv, dtype=np.float64)
        m = ref_id == i
        if st == SPH:
            outward[m] = (point[m] - sv[:3]) / sv[3]                                       # shape.rs:86
        else:
            outward[m] = _norm(np.cross(sv[3:6] - sv[0:3], sv[6:9] - sv[0:3]))            # shape.rs:195
    front = (d * outward).sum(-1) < 0.0                                                   # base.rs:20
    normal = np.where(front[:, None], outward, -outward)
    r = rec[ok].astype(np.float64)
    assert np.abs(r[:, 0] - ref_t[ok]).max() <= 5e-3 * ref_t[ok].max()
    assert np.abs(r[:, 1:4] - point[ok]).max() <= 2e-3
    # a ray that runs in the surface (d . n ~ 0) may be called front or back by either arithmetic
    clear = np.abs((d * outward).sum(-1))[ok] > 1e-4
    assert (r[clear, 7] == front[ok][clear]).all()
    assert np.abs(r[clear, 4:7] - normal[ok][clear]).max() <= 2e-3


# ------------------------------------------------------------------ (b) light sampling
def _uniforms(rng, n, k):
    words = rng.integers(0, 1 << 32, size=(n, k), dtype=np.uint64).astype(np.uint32)
    return words, (((words >> 9).astype(np.float64) * 2.0) + 1.0) / 16777216.0


@pytest.mark.parametrize("dist", [0.3, 1.5, 12.0])
def test_sphere_light_sampling_is_uniform_in_its_cone(pt, gpu_ctx, dist):
    c, r = np.array([0.2, 0.9, -1.0]), 0.25
    gpu_ctx.upload(pt.make_objects([(SPH, list(c) + [r], EMISSIVE, [5, 5, 5])]))
    rng = np.random.default_rng(int(dist * 10))
    n = 1 << 18
    frm = c + _norm(rng.normal(size=3)) * (r + dist)
    r12 = rng.random((n, 2))
    out = gpu_ctx.debug_shape_sample(0, np.tile(frm, (n, 1)), r12=r12).astype(np.float64)      # point3, pdf, dir3, distance
    dc = np.linalg.norm(c - frm)
    cos_max = np.sqrt(max(0.0, 1.0 - r * r / (dc * dc)))                                      # shape.rs:98-100
    pdf = 1.0 / (2.0 * np.pi * (1.0 - cos_max))                                               # shape.rs:103-104
    assert np.abs(out[:, 3] / pdf - 1.0).max() <= 2e-4 + 3e-7 / (1.0 - cos_max)               # (1 - cos) cancels in f32 for a far light
    w = (c - frm) / dc
    cosang = (out[:, 4:7] * w).sum(-1)
    assert cosang.min() >= cos_max - 1e-5                                                     # inside the cone
    on = np.linalg.norm(out[:, 0:3] - c, axis=1)
    assert np.abs(on / r - 1.0).max() <= 1e-3 * max(1.0, dc / r * 0.05)                       # on the sphere ...
    assert ((out[:, 0:3] - c) @ w).max() <= 1e-3 * r                                          # ... on the cap that faces the point
    assert np.abs(np.linalg.norm(out[:, 0:3] - frm, axis=1) / out[:, 7] - 1.0).max() <= 1e-3
    # uniform in solid angle: cos theta = 1 - r1 (1 - cos_max) (shape.rs:113), so its mean is (1 + cos_max) / 2 ...
    se = (1.0 - cos_max) / np.sqrt(12.0 * n)
    assert abs(cosang.mean() - 0.5 * (1.0 + cos_max)) <= 5.0 * se + 1e-6
    # ... and draw for draw
    assert np.abs(cosang - (1.0 - r12[:, 0] + r12[:, 0] * cos_max)).max() <= 2e-5


def _triangle_solid_angle(p, a, b, c):
    """Van Oosterom & Strackee 1983: tan(Omega / 2) = |A . (B x C)| / (|A||B||C| + (A.B)|C| + (A.C)|B| + (B.C)|A|)"""
    A, B, C = a - p, b - p, c - p
    la, lb, lc = np.linalg.norm(A), np.linalg.norm(B), np.linalg.norm(C)
    num = abs(np.dot(A, np.cross(B, C)))
    den = la * lb * lc + np.dot(A, B) * lc + np.dot(A, C) * lb + np.dot(B, C) * la
    return 2.0 * np.arctan2(num, den)


@pytest.mark.parametrize("frm", [(0.0, -0.9, -1.8), (0.6, 0.2, -2.4), (-0.9, 0.95, -1.2)])
def test_triangle_light_sampling_against_its_solid_angle(pt, gpu_ctx, frm):
    v0, v1, v2 = np.array([-0.3, 0.99, -2.3]), np.array([0.3, 0.99, -2.3]), np.array([0.3, 0.99, -1.7])
    gpu_ctx.upload(pt.make_objects([(TRI, list(v0) + list(v1) + list(v2), EMISSIVE, [5, 5, 5])]))
    rng = np.random.default_rng(5)
    n = 1 << 18
    frm = np.asarray(frm, dtype=np.float64)
    r12 = rng.random((n, 2))
    out = gpu_ctx.debug_shape_sample(0, np.tile(frm, (n, 1)), r12=r12).astype(np.float64)
    s1 = np.sqrt(r12[:, 0])
    point = v0 + (v1 - v0) * (1.0 - s1)[:, None] + (v2 - v0) * (r12[:, 1] * s1)[:, None]        # shape.rs:212-217
    assert np.abs(out[:, 0:3] - point).max() <= 1e-5
    nrm = np.cross(v1 - v0, v2 - v0)
    area = 0.5 * np.linalg.norm(nrm)
    nrm = _norm(nrm)
    to = point - frm
    dd = np.linalg.norm(to, axis=1)
    cosl = np.abs((to / dd[:, None]) @ nrm)                                                    # shape.rs:228 (two-sided)
    pdf = dd * dd / (area * cosl)                                                              # shape.rs:231-233
    assert np.abs(out[:, 3] / pdf - 1.0).max() <= 2e-4
    assert np.abs(out[:, 4:7] - to / dd[:, None]).max() <= 1e-5 and np.abs(out[:, 7] / dd - 1.0).max() <= 1e-5
    inv = 1.0 / out[:, 3]
    omega = _triangle_solid_angle(frm, v0, v1, v2)
    assert abs(inv.mean() - omega) <= 5.0 * inv.std(ddof=1) / np.sqrt(n) + 1e-6 * omega


# ------------------------------------------------------------------ (c) diffuse materials
def _tangent_frame(nrm):
    up = np.where((np.abs(nrm[:, 1]) > 0.999)[:, None], np.array([1.0, 0.0, 0.0]), np.array([0.0, 1.0, 0.0]))     # material.rs:109-113
    t = _norm(np.cross(up, nrm))
    return t, np.cross(nrm, t)


def _oren_nayar(albedo, sigma, i, o, nrm):
    """OrenNayar::bsdf_pdf, material.rs:221-265 with the coefficients of :182-193 (i = -ray.direction)"""
    s2 = sigma * sigma
    A, B = 1.0 - 0.5 * s2 / (s2 + 0.33), 0.45 * s2 / (s2 + 0.09)
    ci, co = np.maximum((i * nrm).sum(-1), 0.0), np.maximum((o * nrm).sum(-1), 0.0)
    si, so = np.sqrt(np.maximum(1.0 - ci * ci, 0.0)), np.sqrt(np.maximum(1.0 - co * co, 0.0))
    t, b = _tangent_frame(nrm)
    phi_i = np.arctan2((i * b).sum(-1), (i * t).sum(-1))
    phi_o = np.arctan2((o * b).sum(-1), (o * t).sum(-1))
    cphi = np.maximum(np.cos(phi_i - phi_o), 0.0)
    with np.errstate(divide="ignore", invalid="ignore"):
        first = ci > co
        tan_beta = np.where(first, np.where(ci > 1e-6, si / ci, 0.0), np.where(co > 1e-6, so / co, 0.0))
    sin_alpha = np.where(first, so, si)
    term = A + B * cphi * sin_alpha * tan_beta
    return albedo[None, :] * (term / np.pi)[:, None], co / np.pi


@pytest.mark.parametrize("sigma", [0.0, 0.35, 1.0])
def test_lambertian_and_oren_nayar_values_against_material_rs(pt, gpu_ctx, sigma):
    albedo = np.array([0.8, 0.45, 0.2])
    gpu_ctx.upload(pt.make_objects([(SPH, [0, 0, 0, 1.0], LAMBERT, list(albedo)),
                                    (SPH, [3, 0, 0, 1.0], OREN, list(albedo) + [sigma])]))
    rng = np.random.default_rng(11 + int(sigma * 100))
    n = 100_000
    nrm = _norm(rng.normal(size=(n, 3)))
    nrm[:100] = np.array([0.0, 1.0, 0.0])                      # the frame's other branch (|n.y| > 0.999)
    i = _norm(rng.normal(size=(n, 3)))
    i = np.where(((i * nrm).sum(-1) < 0)[:, None], -i, i)       # the incoming ray arrives from the normal's side
    o = _norm(rng.normal(size=(n, 3)))                          # any direction: below the surface the cosine clamps to 0
    inp = np.concatenate([-i, o, nrm, np.ones((n, 1))], 1)
    lam = gpu_ctx.debug_bsdf_eval(0, inp).astype(np.float64)
    co = np.maximum((o * nrm).sum(-1), 0.0)
    assert np.abs(lam[:, :3] - albedo / np.pi).max() <= 1e-6                                    # material.rs:87-88
    assert np.abs(lam[:, 3] - co / np.pi).max() <= 2e-6                                         # material.rs:77-80
    f, pdf = _oren_nayar(albedo, sigma, i, o, nrm)
    on = gpu_ctx.debug_bsdf_eval(1, inp).astype(np.float64)
    assert np.abs(on[:, 3] - pdf).max() <= 2e-6
    # tan(beta) is unbounded towards grazing angles (compare relative to the term's own size), and it is SET to 0 below a cosine of
    # 1e-6 (material.rs:239-251): directions within f32 rounding of that switch are left out
    ci = np.maximum((i * nrm).sum(-1), 0.0)
    safe = np.maximum(ci, co) > 1e-4
    assert safe.mean() > 0.999
    assert (np.abs(on[safe, :3] - f[safe]) / (np.abs(f[safe]) + 1e-3)).max() <= 2e-3
    if sigma == 0.0:
        assert np.abs(on[:, :3] - albedo / np.pi).max() <= 1e-6                                 # A = 1, B = 0: Lambertian


def test_cosine_weighted_sampler_draw_for_draw_and_lambertian_energy(pt, gpu_ctx):
    albedo = np.array([0.8, 0.45, 0.2])
    gpu_ctx.upload(pt.make_objects([(SPH, [0, 0, 0, 1.0], LAMBERT, list(albedo)),
                                    (SPH, [3, 0, 0, 1.0], OREN, list(albedo) + [0.6])]))
    rng = np.random.default_rng(3)
    n = 1 << 18
    nrm = np.tile(_norm(np.array([0.3, 0.8, -0.5])), (n, 1))
    nrm[: n // 8] = np.array([0.0, -1.0, 0.0])
    i = _norm(nrm + 0.7 * _norm(rng.normal(size=(n, 3))))
    words, uni = _uniforms(rng, n, 4)
    inp = np.concatenate([-i, nrm, np.ones((n, 1))], 1)
    for obj in (0, 1):
        s = gpu_ctx.debug_bsdf_sample(obj, inp, words).astype(np.float64)                       # wo3, f3, pdf, cos
        phi = 2.0 * np.pi * uni[:, 0]                                                           # material.rs:100-107
        ct = np.sqrt(uni[:, 1])
        st = np.sqrt(1.0 - ct * ct)
        t, b = _tangent_frame(nrm)
        wo = _norm(t * (st * np.cos(phi))[:, None] + b * (st * np.sin(phi))[:, None] + nrm * ct[:, None])
        assert np.abs(s[:, 0:3] - wo).max() <= 2e-4
        assert np.abs(s[:, 7] - ct).max() <= 2e-4 and np.abs(s[:, 6] - ct / np.pi).max() <= 1e-4
        if obj == 0:
            w = s[:, 3:6] * s[:, 7:8] / s[:, 6:7]                                               # f cos / pdf = albedo, sample by sample
            assert np.abs(w - albedo).max() <= 1e-4


# ------------------------------------------------------------------ (d) camera
def _camera_new(origin, width, height, screen_distance, fov_degrees):
    """Camera::new, camera.rs:50-82"""
    origin = np.asarray(origin, dtype=np.float64)
    vh = 2.0 * np.tan(np.radians(fov_degrees) / 2.0) * screen_distance
    vw = vh * (width / height)
    hor, ver = np.array([vw, 0.0, 0.0]), np.array([0.0, vh, 0.0])
    return origin, origin - hor / 2.0 - ver / 2.0 - np.array([0.0, 0.0, screen_distance]), hor, ver


def _camera_look_at(origin, target, up, width, height, fov_degrees):
    """Camera::look_at, camera.rs:94-130"""
    origin, target, up = (np.asarray(a, dtype=np.float64) for a in (origin, target, up))
    w = _norm(origin - target)
    u = _norm(np.cross(up, w))
    v = np.cross(w, u)
    vh = 2.0 * np.tan(np.radians(fov_degrees) / 2.0)
    vw = vh * (width / height)
    hor, ver = u * vw, v * vh
    return origin, origin - hor / 2.0 - ver / 2.0 - w, hor, ver


@pytest.mark.parametrize("kind", ["new", "look_at"])
def test_camera_rays_against_camera_rs(pt, gpu_ctx, kind):
    """Camera::new / look_at and get_ray_with_offset (camera.rs:139-147) as World::render_pixel calls it (world.rs:297-299: film row
    y looks through camera row HEIGHT - 1 - y): the host-side constructors and the device's ray against the formulas, with the
    jitter the device reports for the sample."""
    W, H = 317, 201
    if kind == "new":
        cam = pt.camera_new(origin=(0.1, -0.2, 2.5), width=W, height=H, screen_distance=1.3, fov_degrees=41.0)
        org, llc, hor, ver = _camera_new((0.1, -0.2, 2.5), W, H, 1.3, 41.0)
    else:
        cam = pt.camera_look_at((1.0, 0.7, 2.0), (0.0, -0.1, -2.0), (0.1, 1.0, 0.0), W, H, 33.0)
        org, llc, hor, ver = _camera_look_at((1.0, 0.7, 2.0), (0.0, -0.1, -2.0), (0.1, 1.0, 0.0), W, H, 33.0)
    for got, want in ((cam.origin, org), (cam.lower_left, llc), (cam.horizontal, hor), (cam.vertical, ver)):
        assert np.abs(np.array(list(got)) - want).max() <= 1e-12
    gpu_ctx.upload(pt.builtin_scene(2))
    rng = np.random.default_rng(9)
    n = 50_000
    xys = np.stack([rng.integers(0, W, n), rng.integers(0, H, n), rng.integers(0, 4096, n)], 1).astype(np.uint32)
    xys[:4] = [[0, 0, 0], [W - 1, 0, 1], [0, H - 1, 2], [W - 1, H - 1, 3]]
    out = gpu_ctx.debug_camera_rays(cam, xys).astype(np.float64)               # origin3, direction3, ox, oy
    ox, oy = out[:, 6], out[:, 7]
    assert (ox > 0).all() and (ox < 1).all() and (oy > 0).all() and (oy < 1).all()
    assert abs(ox.mean() - 0.5) <= 5.0 / np.sqrt(12.0 * n) and abs(oy.mean() - 0.5) <= 5.0 / np.sqrt(12.0 * n)
    u = (xys[:, 0] + ox) / (W - 1)                                            # camera.rs:140-141
    v = ((H - 1 - xys[:, 1].astype(np.float64)) + oy) / (H - 1)               # world.rs:298
    d = _norm(llc + hor * u[:, None] + ver * v[:, None] - org)                # camera.rs:143-146, Ray::new normalises (camera.rs:10-16)
    assert np.abs(out[:, 0:3] - org).max() <= 1e-6
    assert np.abs(out[:, 3:6] - d).max() <= 2e-6


# ------------------------------------------------------------------ (e) World::sample_light_point
def test_sample_light_point_against_world_rs(pt, gpu_ctx):
    """World::sample_light_point (world.rs:251-267) over World::new's light list (world.rs:213-225: the objects whose emission is not
    zero, in object order): which light an index word picks (uniformly: floor(u * n)), that light's own point sampling (shape.rs:106-131
    for a sphere, :208-217 for a triangle) draw for draw, its emission, and pdf = pdf_shape / n."""
    tri_a = [-0.3, 0.99, -2.3, 0.3, 0.99, -2.3, 0.3, 0.99, -1.7]
    tri_b = [0.9, -0.2, -2.0, 0.9, 0.5, -2.4, 0.9, 0.4, -1.6]
    sph = [-0.5, 0.3, -1.5, 0.2]
    specs = [(SPH, [0.4, -0.6, -2.0, 0.4], LAMBERT, [0.7, 0.7, 0.7]),
             (SPH, sph, EMISSIVE, [4.0, 3.0, 2.0]),
             (TRI, [-1, -1, -3, 1, -1, -3, 0, 1, -3], LAMBERT, [0.5, 0.5, 0.5]),
             (TRI, tri_a, EMISSIVE, [9.0, 9.0, 8.0]),
             (SPH, [0.0, -0.9, -1.2, 0.1], EMISSIVE, [0.0, 0.0, 0.0]),       # emits nothing: not a light
             (TRI, tri_b, EMISSIVE, [1.0, 2.0, 3.0])]
    lights = [1, 3, 5]
    gpu_ctx.upload(pt.make_objects(specs))
    rng = np.random.default_rng(21)
    n = 120_000
    frm = np.array([0.1, -0.5, -1.9]) + rng.uniform(-0.2, 0.2, (n, 3))
    words, uni = _uniforms(rng, n, 4)
    out = gpu_ctx.debug_light_point(frm, words).astype(np.float64)          # point3, emission3, pdf, light object
    pick = ((words[:, 0].astype(np.uint64) * 3) >> 32).astype(np.int64)
    assert (out[:, 7].astype(np.int64) == np.array(lights)[pick]).all()
    assert np.abs(np.bincount(pick, minlength=3) / n - 1.0 / 3.0).max() < 0.01
    r1, r2 = uni[:, 1], uni[:, 2]
    for k, obj in enumerate(lights):
        m = pick == k
        sv = np.asarray(specs[obj][1], dtype=np.float64)
        assert np.abs(out[m, 3:6] - np.asarray(specs[obj][3])).max() == 0.0
        f = frm[m]
        if specs[obj][0] == SPH:
            c, r = sv[:3], sv[3]
            tc = c - f
            d2 = (tc * tc).sum(-1)
            cmax = np.sqrt(np.maximum(1.0 - r * r / d2, 0.0))
            pdf_shape = 1.0 / (2.0 * np.pi * (1.0 - cmax))
            ct = 1.0 - r1[m] + r1[m] * cmax
            st = np.sqrt(np.maximum(1.0 - ct * ct, 0.0))
            phi = 2.0 * np.pi * r2[m]
            w = _norm(tc)
            up = np.where((np.abs(w[:, 1]) > 0.999)[:, None], np.array([1.0, 0.0, 0.0]), np.array([0.0, 1.0, 0.0]))
            u = _norm(np.cross(up, w))
            v = np.cross(w, u)
            dr = _norm(u * (st * np.cos(phi))[:, None] + v * (st * np.sin(phi))[:, None] + w * ct[:, None])
            oc = f - c
            hb = (oc * dr).sum(-1)
            disc = hb * hb - ((oc * oc).sum(-1) - r * r)
            point = f + dr * (-hb - np.sqrt(np.maximum(disc, 0.0)))[:, None]
            tol = 2e-4          # the root near the cone's rim is a small difference of two larger numbers
        else:
            v0, v1, v2 = sv[0:3], sv[3:6], sv[6:9]
            s1 = np.sqrt(r1[m])
            point = v0 + (v1 - v0) * (1.0 - s1)[:, None] + (v2 - v0) * (r2[m] * s1)[:, None]
            nn = np.cross(v1 - v0, v2 - v0)
            area = 0.5 * np.linalg.norm(nn)
            to = point - f
            dd = np.linalg.norm(to, axis=1)
            pdf_shape = dd * dd / (area * np.abs((to / dd[:, None]) @ _norm(nn)))
            tol = 1e-5
        assert np.abs(out[m, 0:3] - point).max() <= tol, obj
        assert np.abs(out[m, 6] / (pdf_shape / 3.0) - 1.0).max() <= 5e-4, obj       # world.rs:260
