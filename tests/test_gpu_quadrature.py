"""Pins of the integrator that do NOT go through oracle/: expected values computed here by numpy quadrature straight from the
reference's formulas (file:line cited at each), against Monte-Carlo means of the HIP path through the C ABI.

The reference pins nothing beyond Vector3, and the oracle is a restatement by the same hand as the kernels; these
tests are the statement that can be made without either: the quirks of the reference's estimator are reproduced, not invented.

  (i)   One-bounce radiance: a ray that hits a Lambertian surface point x of a scene whose OTHER surfaces are black.  What
        MisStrategy::ray_color (rendering.rs:34-142) returns then has a closed-form expectation -- the NEE term with
        w_nee = (p_l / n) / (p_l / n + p_b)  (world.rs:260, rendering.rs:73) plus the emitter-hit term of the BSDF-sampled
        ray with w_bsdf = p_b / (p_b + p_l)  (rendering.rs:117: p_l NOT divided by the light count n = SURVEY Q2):
            E = rho Le / pi * Int_lights cos_x cos_l / d^2 * V * (w_nee + w_bsdf) dA
        For World::new()'s two triangle lights (n = 2, shape.rs:200-242) the weights do not sum to 1: the reference is
        biased by ~1-2 % and so must the kernels be.  For C2's one sphere light (shape.rs:91-145) they do, and E is the
        analytic irradiance of a sphere.
  (ii)  The same with min_depth = 0: roulette at the first vertex (rendering.rs:91-102) drops the NEE term of a killed
        path, the surviving emitter-hit term is divided by rr:  E = lum(rho) * E_nee + E_bsdf.
        (The variant "image mean at min_depth 0 = image mean at min_depth 4" does NOT hold for the reference's MIS
        estimator, precisely because of Q1; it does hold for BrdfOnlyStrategy, rendering.rs:246-262, tested below.)
  (iii) GGX (mirror.rs): Mirror::bsdf_pdf and Mirror::bsdf_pdf_sample of the device against numpy restatements of the formulas
        (brdf :62-88, btdf :90-124, get_f :126-132, get_g1 / get_g :136-175, sample_ggx_vndf :17-60, bsdf_pdf_sample :200-305),
        draw for draw; energy of the eval side (reflection lobe <= 1, transmission lobe <= (1 / eta)^2: radiance scales with
        the squared index); and the sampler's mean weight E[f cos / pdf] against Int f_eval cos dw for the metal, with F = 1
        on both sides -- a metal's sampled reflection carries F = 1 instead of Schlick's coloured F (mirror.rs:225-231).
        For glass the reference's sampler and eval are not consistent with each other (see the test): printed, not asserted.
  (iv)  Furnace tests -- a convex body inside a uniformly emitting sphere must send back albedo * Le: BrdfOnlyStrategy does (Lambertian,
        OrenNayar by its directional albedo, GGX metal by its sampler's mean weight, with and without roulette); MisStrategy loses
        25-45 %, because the reference samples a sphere light seen from INSIDE over half of the directions only (shape.rs:98-136) while
        the look-ahead pdf claims all of them -- a quirk beyond SURVEY's Q1-Q10, reproduced and pinned by quadrature.
  (v)   An integrating sphere (Lambertian sphere lit by a small sphere at its centre): the wall radiance with ALL orders of
        interreflection is rho Le s / (1 - rho (1 - s)); MIS without roulette and BRDF-only with roulette return it, MIS with the
        reference's roulette returns the value the Q1 recursion predicts (up to 12 % less).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N_STREAMS = 1 << 20
SPH, TRI, LAMBERT, EMISSIVE, MIRROR = 0, 1, 0, 1, 2


# ------------------------------------------------------------------ helpers (numpy, f64)
def _norm(v):
    return v / np.linalg.norm(v, axis=-1, keepdims=True)


def _sphere_blocks(o, d, tmax, c, r):
    """Does the segment o + t d, t in (1e-3, tmax - 1e-3), meet the sphere (c, r)?  d unit; arrays broadcast."""
    oc = o - c
    hb = (oc * d).sum(-1)
    disc = hb * hb - ((oc * oc).sum(-1) - r * r)
    sq = np.sqrt(np.maximum(disc, 0.0))
    t0, t1 = -hb - sq, -hb + sq
    lo, hi = 1e-3, tmax - 1e-3
    return (disc > 0) & (((t0 > lo) & (t0 < hi)) | ((t1 > lo) & (t1 < hi)))


def _gpu_mean(gpu_ctx, pt, origin, target, n=N_STREAMS, **params):
    """Mean and standard error of pt_ray_color over n independent RNG streams for ONE ray (origin -> target)."""
    d = np.asarray(target, dtype=np.float64) - np.asarray(origin, dtype=np.float64)
    rays = np.tile(np.concatenate([origin, d]), (n, 1))
    idx = np.arange(n, dtype=np.uint32)
    xy = np.stack([idx & 0xFFFF, idx >> 16], 1).astype(np.uint32)
    rgb = gpu_ctx.ray_color(pt.default_params(spp=1, **params), rays, xy).astype(np.float64)
    assert np.isfinite(rgb).all()
    return rgb.mean(0), rgb.std(0, ddof=1) / np.sqrt(n)


def _black_except(pt, objs, keep):
    """The scene with every non-emissive object black (Lambertian, albedo 0) except object `keep`; the glass sphere of
    World::new() becomes a black Lambertian sphere: an occluder, nothing more."""
    out = []
    for k, o in enumerate(objs):
        sv = list(o.shape)
        if o.mat_tag == EMISSIVE or k == keep:
            assert k != keep or o.mat_tag == LAMBERT
            out.append((o.shape_tag, sv, o.mat_tag, list(o.mat)))
        else:
            out.append((o.shape_tag, sv, LAMBERT, [0.0, 0.0, 0.0]))
    return pt.make_objects(out)


def _check(mean, sem, want, what):
    tol = 4.0 * sem + 2e-4 * np.abs(want)
    assert np.all(np.abs(mean - want) <= tol), (what, mean, want, (mean - want) / np.maximum(sem, 1e-30))


# ------------------------------------------------------------------ (i) + (ii): World::new(), two triangle lights
LE_TRI, LS, LY, BD = 15.0, 0.3, 0.99, -2.0                      # world.rs:167-182
GLASS_C, GLASS_R = np.array([0.4, -0.6, -2.0]), 0.4                # world.rs:202-210 (black occluder here)


def _triangle_light_terms(x, n, m=700):
    """E_nee / rho and E_bsdf / rho at surface point x (normal n) by midpoint quadrature over the light square
    [-0.3, 0.3] x {0.99} x [-2.3, -1.7] (two triangles of area 0.18 each, both in the plane y = 0.99, world.rs:167-182)."""
    u = (np.arange(m) + 0.5) / m
    xs, zs = np.meshgrid(-LS + 2 * LS * u, BD - LS + 2 * LS * u, indexing="ij")
    y = np.stack([xs, np.full_like(xs, LY), zs], -1)
    to = y - x
    d = np.linalg.norm(to, axis=-1)
    l = to / d[..., None]
    cos_x = np.maximum((l * n).sum(-1), 0.0)                       # p_b = max(0, n.l) / pi (material.rs:80); |n.l| in NEE (rendering.rs:68)
    cos_l = np.abs(l[..., 1])                                      # two-sided light, normal (0, +-1, 0) (shape.rs:222)
    vis = ~_sphere_blocks(x, l, d, GLASS_C, GLASS_R)
    area_tri = 0.5 * (2 * LS) * (2 * LS)
    p_l = d * d / (area_tri * cos_l)                               # shape.rs:225-229 (solid-angle pdf of ONE triangle)
    p_b = cos_x / np.pi
    w_nee = (p_l / 2.0) / (p_l / 2.0 + p_b)                        # world.rs:260 (/ n_lights), rendering.rs:73
    w_bsdf = p_b / (p_b + p_l)                                     # rendering.rs:117 (NOT / n_lights: Q2)
    geom = cos_x * cos_l / (d * d) * vis * (LE_TRI / np.pi)
    dA = (2 * LS / m) ** 2
    return float((geom * w_nee).sum() * dA), float((geom * w_bsdf).sum() * dA), float((geom).sum() * dA)


# (object index in World::new() order: left 0-1, right 2-3, back 4-5, floor 6-7, ceiling 8-9, lights 10-11, sphere 12; point; normal)
TRI_POINTS = [
    ("floor, light unobstructed", 7, [-0.5, -1.0, -2.2], [0.0, 1.0, 0.0]),
    ("floor beside the sphere, light partly hidden", 7, [0.88, -1.0, -1.55], [0.0, 1.0, 0.0]),
    ("floor, front", 6, [0.1, -1.0, -1.2], [0.0, 1.0, 0.0]),
    ("left wall", 0, [-1.0, 0.0, -2.0], [1.0, 0.0, 0.0]),
    ("back wall", 4, [0.3, -0.2, -3.0], [0.0, 0.0, 1.0]),
    ("right wall, high", 2, [1.0, 0.5, -1.5], [-1.0, 0.0, 0.0]),
]


def _which_triangle(pt, objs, gpu_ctx, origin, target):
    ids, t = gpu_ctx.debug_hit_scene(np.concatenate([origin, np.asarray(target) - origin])[None, :], 1e-3, float("inf"))
    return int(ids[0]), float(t[0])


@pytest.mark.parametrize("label,obj,x,n", TRI_POINTS, ids=[p[0] for p in TRI_POINTS])
def test_one_bounce_radiance_under_two_triangle_lights_carries_the_reference_bias(pt, gpu_ctx, label, obj, x, n):
    x, n = np.array(x), np.array(n)
    base = pt.builtin_scene(1)
    origin = x + 0.25 * n + 0.03 * np.array([0.3, 0.2, -0.5])
    gpu_ctx.upload(base)
    hit, _ = _which_triangle(pt, base, gpu_ctx, origin, x)          # which of the wall's two triangles holds x
    assert hit in (obj, obj ^ 1) and base[hit].mat_tag == LAMBERT, (hit, obj)
    rho = np.array(list(base[hit].mat)[:3])
    objs = _black_except(pt, base, hit)
    gpu_ctx.upload(objs)
    e_nee, e_bsdf, e_unweighted = _triangle_light_terms(x, n)
    # the bias is there to be seen: the weights sum to less than 1 by about p_b / p_l
    assert 0.002 < 1.0 - (e_nee + e_bsdf) / e_unweighted < 0.05
    mean, sem = _gpu_mean(gpu_ctx, pt, origin, x)
    _check(mean, sem, rho * (e_nee + e_bsdf), label)
    # the unbiased value is NOT what the reference computes (and here it is distinguishable):
    assert np.abs(mean - rho * e_unweighted)[np.argmax(rho)] > 4.0 * sem[np.argmax(rho)], "Q2 bias not visible"
    # (ii) roulette from the first vertex on (min_depth = 0): rr = min(luminance(next_throughput), 1) = lum(rho)
    # (rendering.rs:89-98, math.rs:133-135); a killed path returns 0 and loses its NEE term (Q1), the survivor's
    # emitter-hit term is divided by rr (rendering.rs:119)
    rr = min(0.2126 * rho[0] + 0.7152 * rho[1] + 0.0722 * rho[2], 1.0)
    mean0, sem0 = _gpu_mean(gpu_ctx, pt, origin, x, min_depth=0)
    _check(mean0, sem0, rho * (rr * e_nee + e_bsdf), label + ", min_depth 0")
    # exact arithmetic mode: same expectation
    mean_x, sem_x = _gpu_mean(gpu_ctx, pt, origin, x, n=1 << 18, exact_math=1)
    _check(mean_x, sem_x, rho * (e_nee + e_bsdf), label + ", exact arithmetic")


# ------------------------------------------------------------------ (i): C2, one sphere light -> the weights sum to 1
LIGHT_C, LIGHT_R, LE_SPH = np.array([0.0, 0.79, -2.0]), 0.2, 36.0   # world.rs:184-190 (pt_scenes.cpp scene 2)


def _sphere_light_expectation(x, n, occluders, m_theta=400, m_phi=800):
    """(1 / pi) Le Int_cone max(0, n.w) V dw over the cone of directions that meet the light sphere (shape.rs:97-103),
    by midpoint quadrature in (cos theta, phi)."""
    to_c = LIGHT_C - x
    dc = np.linalg.norm(to_c)
    w = to_c / dc
    cos_max = np.sqrt(1.0 - (LIGHT_R / dc) ** 2)
    up = np.array([1.0, 0.0, 0.0]) if abs(w[1]) > 0.999 else np.array([0.0, 1.0, 0.0])
    u = _norm(np.cross(up, w)); v = np.cross(w, u)
    ct = 1.0 - (np.arange(m_theta) + 0.5) / m_theta * (1.0 - cos_max)
    ph = (np.arange(m_phi) + 0.5) / m_phi * 2 * np.pi
    ct, ph = np.meshgrid(ct, ph, indexing="ij")
    st = np.sqrt(1.0 - ct * ct)
    dirs = u * (st * np.cos(ph))[..., None] + v * (st * np.sin(ph))[..., None] + w * ct[..., None]
    # distance to the light surface along each direction
    hb = -(to_c * dirs).sum(-1)
    t_l = -hb - np.sqrt(np.maximum(hb * hb - (dc * dc - LIGHT_R ** 2), 0.0))
    vis = np.ones_like(t_l, dtype=bool)
    for c, r in occluders:
        vis &= ~_sphere_blocks(x, dirs, t_l, np.array(c), r)
    cosx = np.maximum((dirs * n).sum(-1), 0.0)
    dw = (1.0 - cos_max) / m_theta * (2 * np.pi / m_phi)
    return float((cosx * vis).sum() * dw * LE_SPH / np.pi), bool(vis.all())


SPH_RAYS = [
    ("floor, left front", [-0.85, -0.5, -1.3], [0.0, -1.0, 0.0]),
    ("floor, behind the right sphere", [0.75, -0.5, -2.75], [0.0, -1.0, 0.0]),
    ("left wall", [-0.5, 0.2, -1.5], [-1.0, 0.0, 0.0]),
    ("back wall, low", [-0.3, -0.5, -2.5], [0.0, 0.0, -1.0]),
    ("top of the left sphere", [-0.4, 0.3, -2.0], [0.0, -1.0, 0.0]),
    ("floor between the spheres, light partly hidden by the blue sphere", [0.0, -0.5, -2.85], [0.0, -1.0, 0.0]),
]


@pytest.mark.parametrize("label,o,d", SPH_RAYS, ids=[p[0] for p in SPH_RAYS])
def test_one_bounce_radiance_under_the_sphere_light_is_the_sphere_irradiance(pt, gpu_ctx, label, o, d):
    o, d = np.array(o), np.array(d)
    base = pt.builtin_scene(2)
    gpu_ctx.upload(base)
    ids, t = gpu_ctx.debug_hit_scene(np.concatenate([o, d])[None, :], 1e-3, float("inf"))
    hit = int(ids[0])
    assert hit >= 0 and base[hit].mat_tag == LAMBERT and base[hit].shape_tag == SPH
    c, r = np.array(list(base[hit].shape)[:3]), base[hit].shape[3]
    # the hit point and normal in f64 from the geometry (not from the device's t)
    oc = o - c
    hb = (oc * d).sum()
    tt = -hb - np.sqrt(hb * hb - (oc @ oc - r * r))
    x = o + tt * d
    n = (x - c) / r
    assert abs(tt - float(t[0])) < 1e-3
    rho = np.array(list(base[hit].mat)[:3])
    occl = [(list(ob.shape)[:3], ob.shape[3]) for k, ob in enumerate(base) if ob.mat_tag != EMISSIVE and k != hit]
    want, unobstructed = _sphere_light_expectation(x, n, occl)
    if unobstructed:
        # closed form: a sphere of radiance Le wholly above the horizon gives the irradiance pi Le (r / D)^2 cos(theta_c)
        D = np.linalg.norm(LIGHT_C - x)
        cos_c = ((LIGHT_C - x) @ n) / D
        assert cos_c > LIGHT_R / D
        assert abs(want - LE_SPH * (LIGHT_R / D) ** 2 * cos_c) < 2e-4 * want
    gpu_ctx.upload(_black_except(pt, base, hit))
    mean, sem = _gpu_mean(gpu_ctx, pt, o, o + d)
    _check(mean, sem, rho * want, label)


# ------------------------------------------------------------------ roulette is unbiased where nothing is dropped
def test_brdf_only_image_mean_does_not_depend_on_min_depth(pt, gpu_ctx):
    """BrdfOnlyStrategy (rendering.rs:214-265) has no NEE term for the roulette to drop: f Li cos / (pdf rr) keeps the
    expectation, so the image mean with roulette from the first vertex (min_depth 0) equals the mean with the
    reference's min_depth 4 within the Monte-Carlo error (estimated from 16 independent sub-renders each)."""
    gpu_ctx.upload(pt.builtin_scene(2))
    cam = pt.camera_new(width=96, height=96)
    def means(min_depth):
        return np.array([gpu_ctx.render(cam, pt.default_params(spp=256, spp_offset=256 * k, integrator=1, min_depth=min_depth))[0]
                         .cpu().numpy().astype(np.float64).mean() for k in range(16)])
    a, b = means(4), means(0)
    se = np.sqrt(a.var(ddof=1) / 16 + b.var(ddof=1) / 16)
    assert abs(a.mean() - b.mean()) < 3.0 * se, (a.mean(), b.mean(), se)
    # ... whereas the MIS estimator's mean drops with min_depth = 0 (Q1: the NEE term of a killed path is lost) -- by far
    # more than the noise; it is the reference's behaviour, pinned point-wise above
    def means_mis(min_depth):
        return np.array([gpu_ctx.render(cam, pt.default_params(spp=256, spp_offset=256 * k, min_depth=min_depth))[0]
                         .cpu().numpy().astype(np.float64).mean() for k in range(8)])
    c, e = means_mis(4), means_mis(0)
    assert c.mean() - e.mean() > 10.0 * np.sqrt(c.var(ddof=1) / 8 + e.var(ddof=1) / 8)


# ------------------------------------------------------------------ (iii) GGX, mirror.rs
def _ggx_eval(i, o, n, rough, color, metallic, ior, eta):
    """Mirror::bsdf_pdf (mirror.rs:179-198) -> f[..., 3], written from brdf() :62-88, btdf() :90-124, get_f :126-132,
    get_g :153-175.  i = -ray.direction (unit), o, n unit; arrays broadcast over leading axes."""
    a2 = (rough * rough) ** 2
    i_n, o_n = (i * n).sum(-1), (o * n).sum(-1)
    refl = i_n * o_n > 0
    def D(nh):
        den = nh * nh * (a2 - 1.0) + 1.0
        return a2 / (np.pi * den * den)
    def lam(c):
        return (np.sqrt(a2 + (1.0 - a2) * c * c) - c) / (2.0 * c)
    def G(ci, co):
        ok = (ci > 0) & (co > 0)
        ci_, co_ = np.where(ok, ci, 1.0), np.where(ok, co, 1.0)
        return np.where(ok, 1.0 / (1.0 + lam(ci_) + lam(co_)), 0.0)
    f0d = ((1.0 - ior) / (1.0 + ior)) ** 2
    f0 = f0d * (1.0 - metallic) + np.asarray(color) * metallic
    def F(c):
        return f0 + (1.0 - f0) * ((1.0 - c) ** 5)[..., None]
    # reflection
    h = _norm(i + o)
    ci, co = np.maximum(i_n, 0.0), np.maximum(o_n, 0.0)
    with np.errstate(divide="ignore", invalid="ignore"):
        f_r = (D((n * h).sum(-1)) * G(ci, co))[..., None] * F(np.maximum((i * h).sum(-1), 0.0)) / (4.0 * ci * co)[..., None]
        # transmission
        ht = -_norm(i * eta + o)
        ih, oh = (i * ht).sum(-1), (o * ht).sum(-1)
        ai, ao = np.abs(i_n), np.abs(o_n)
        den = eta * ih + oh
        f_t = (1.0 - F(np.abs(ih))) * (D((n * ht).sum(-1)) * G(ai, ao) * np.abs(ih) * np.abs(oh) / (ai * ao * den * den))[..., None]
    if metallic > 0.99:
        f_t = np.zeros_like(f_t)
    f = np.where(refl[..., None], f_r, f_t)
    return np.where(np.isfinite(f), f, 0.0)


GGX_MATS = {"glass": (0.3, [1.0, 1.0, 1.0], 0.0, 1.5),              # World::new()'s sphere, world.rs:202-210
            "rough glass": (0.6, [1.0, 1.0, 1.0], 0.0, 1.5),
            "metal": (0.4, [0.9, 0.7, 0.3], 1.0, 1.5)}


def _ggx_sample(i, n, rough, color, metallic, ior, eta, r1, r2, u):
    """Mirror::bsdf_pdf_sample (mirror.rs:200-305) with sample_ggx_vndf (:17-60), vectorised over the draws
    -> (wo[k, 3], f[k, 3], pdf[k], cos[k]); failures are the tuple (n, 0, 1, 0) (:215-217, :264, :299)."""
    k = len(r1)
    color = np.asarray(color, dtype=np.float64)
    alpha = rough * rough
    a2 = alpha * alpha
    up = np.array([1.0, 0.0, 0.0]) if abs(n[1]) > 0.999 else np.array([0.0, 1.0, 0.0])
    t = _norm(np.cross(up, n)); b = np.cross(n, t)
    vl = np.array([i @ t, i @ b, i @ n])
    vh = _norm(np.array([alpha * vl[0], alpha * vl[1], vl[2]]))
    lensq = vh[0] ** 2 + vh[1] ** 2
    T1 = np.array([-vh[1], vh[0], 0.0]) / np.sqrt(lensq) if lensq > 0 else np.array([1.0, 0.0, 0.0])
    T2 = np.cross(vh, T1)
    r = np.sqrt(r1); phi = 2 * np.pi * r2
    p1 = r * np.cos(phi); p2 = r * np.sin(phi)
    sv = 0.5 * (1.0 + vh[2])
    p2 = (1.0 - sv) * np.sqrt(1.0 - p1 * p1) + sv * p2
    nh = T1 * p1[:, None] + T2 * p2[:, None] + vh * np.sqrt(np.maximum(0.0, 1.0 - p1 * p1 - p2 * p2))[:, None]
    ne = _norm(np.stack([alpha * nh[:, 0], alpha * nh[:, 1], np.maximum(0.0, nh[:, 2])], 1))
    h = _norm(t * ne[:, 0:1] + b * ne[:, 1:2] + n * ne[:, 2:3])
    i_h = h @ i
    i_n = i @ n
    f0d = ((1.0 - ior) / (1.0 + ior)) ** 2
    f0 = f0d * (1.0 - metallic) + color * metallic
    F = f0 + (1.0 - f0) * ((1.0 - i_h) ** 5)[:, None]
    cos2t = 1.0 - eta * eta * (1.0 - i_h * i_h)
    forced = (cos2t < 0) | (metallic > 0.99)
    rr_f = np.where(forced, 1.0, F[:, 0])
    F = np.where(forced[:, None], 1.0, F)
    refl = u < rr_f
    n_h = h @ n
    D = a2 / (np.pi * (n_h * n_h * (a2 - 1.0) + 1.0) ** 2)
    def lam(c):
        return (np.sqrt(a2 + (1.0 - a2) * c * c) - c) / (2.0 * c)
    def G(ci, co):
        ok = (ci > 0) & (co > 0)
        ci_, co_ = np.where(ok, ci, 1.0), np.where(ok, co, 1.0)
        return np.where(ok, 1.0 / (1.0 + lam(ci_) + lam(co_)), 0.0)
    def G1(c):
        return np.where(c > 0, 2.0 * c / (c + np.sqrt(a2 + (1.0 - a2) * c * c)), 0.0)
    with np.errstate(divide="ignore", invalid="ignore"):
        # reflect (:238-266)
        o_r = _norm(2.0 * i_h[:, None] * h - i)
        on_r = np.maximum(0.0, o_r @ n)
        in_r = max(0.0, i_n)
        f_r = F * (D * G(np.full(k, in_r), on_r))[:, None] / (4.0 * in_r * on_r * rr_f)[:, None]
        pdf_r = G1(np.full(k, in_r)) * D * np.maximum(0.0, i_h) / in_r / (4.0 * np.abs(i_h))
        # refract (:268-301)
        cos_t = np.sqrt(np.maximum(cos2t, 0.0))
        o_t = _norm(h * (eta * i_h - cos_t)[:, None] - i * eta)
        o_h = (o_t * h).sum(-1)
        on_t = np.abs(o_t @ n)
        in_t = abs(i_n)
        den = eta * i_h + o_h
        f_t = (1.0 - F) * (D * G(np.full(k, in_t), on_t) * np.abs(i_h) * np.abs(o_h) / (in_t * on_t * den * den * (1.0 - rr_f)))[:, None]
        pdf_t = G1(np.full(k, in_t)) * D * np.maximum(0.0, i_h) / in_t * (np.abs(o_h) / (den * den))
    wo = np.where(refl[:, None], o_r, o_t)
    f = np.where(refl[:, None], f_r, f_t)
    pdf = np.where(refl, pdf_r, pdf_t)
    cosv = np.where(refl, on_r, on_t)
    bad = (i_h <= 0) | ~np.isfinite(f).all(-1) | ~np.isfinite(pdf) | (pdf <= 0)
    wo = np.where(bad[:, None], n, wo); f = np.where(bad[:, None], 0.0, f)
    pdf = np.where(bad, 1.0, pdf); cosv = np.where(bad, 0.0, cosv)
    return wo, f, pdf, cosv


# Eval-side integral minus the sampler's mean weight for glass (mean over the channels), from the numpy restatement of
# mirror.rs:17-305 with 2^18 draws: the reference's btdf() and its sampler disagree, more so at grazing incidence and for rough
# surfaces (sampler / eval side: glass 0.999 / 1.002, 0.996 / 1.004, 0.969 / 0.998; rough glass 0.983 / 1.012, 0.949 / 1.040,
# 0.890 / 1.145 at cos_i = 0.95, 0.6, 0.25).  Bounds = the value +- 0.005 (the sampler's mean carries ~6e-4 of noise).
GLASS_GAP = {("glass", 0.95): (-0.0024, 0.0076), ("glass", 0.6): (0.0024, 0.0124), ("glass", 0.25): (0.0236, 0.0336),
             ("rough glass", 0.95): (0.0247, 0.0347), ("rough glass", 0.6): (0.0859, 0.0959), ("rough glass", 0.25): (0.2506, 0.2606)}


@pytest.mark.parametrize("mat", list(GGX_MATS))
@pytest.mark.parametrize("cos_i", [0.95, 0.6, 0.25])
def test_ggx_eval_and_sampler_against_the_formulas_of_mirror_rs(pt, gpu_ctx, mat, cos_i):
    rough, color, metallic, ior = GGX_MATS[mat]
    objs = pt.make_objects([(SPH, [0, 0, 0, 1.0], MIRROR, [rough] + color + [metallic, ior])])
    gpu_ctx.upload(objs)
    n = np.array([0.0, 0.0, 1.0])
    i = np.array([np.sqrt(1 - cos_i ** 2), 0.0, cos_i])
    dir_in = -i
    eta = 1.0 / ior                                                  # entering: front_face -> 1 / ior (rendering.rs:20-25)
    # --- eval: the device against the formulas, at random directions of both hemispheres
    rng = np.random.default_rng(7)
    o = _norm(rng.normal(size=(20000, 3)))
    o = o[np.abs(o[:, 2]) > 0.02]
    want = _ggx_eval(i, o, n, rough, color, metallic, ior, eta)
    inp = np.concatenate([np.tile(dir_in, (len(o), 1)), o, np.tile(n, (len(o), 1)), np.full((len(o), 1), eta)], 1)
    got = gpu_ctx.debug_bsdf_eval(0, inp, exact_math=0).astype(np.float64)[:, :3]
    scale = np.maximum(np.abs(want), 1e-3)
    assert np.mean(np.all(np.abs(got - want) <= 2e-3 * scale + 1e-6, axis=1)) > 0.999
    # --- sampler: the device against the formulas, draw for draw (uniforms = the device's: (2k + 1) / 2^24 of the word's top 23 bits)
    ns = 1 << 18
    words = rng.integers(0, 1 << 32, size=(ns, 4), dtype=np.uint64).astype(np.uint32)
    uni = (((words >> 9).astype(np.float64) * 2.0) + 1.0) / 16777216.0
    sin = np.tile(np.concatenate([dir_in, n, [eta]]), (ns, 1))
    s = gpu_ctx.debug_bsdf_sample(0, sin, words, exact_math=0).astype(np.float64)      # wo3, f3, pdf, cos
    wo, f, pdf, cosv = _ggx_sample(i, n, rough, color, metallic, ior, eta, uni[:, 0], uni[:, 1], uni[:, 2])
    failed_gpu = (s[:, 6] == 1.0) & (s[:, 7] == 0.0) & (np.abs(s[:, 3:6]).sum(-1) == 0)
    failed_ref = (pdf == 1.0) & (cosv == 0.0) & (np.abs(f).sum(-1) == 0)
    assert np.mean(failed_gpu == failed_ref) > 0.9995                     # (a lobe pick or a validity test on the edge may flip in f32)
    both = ~failed_gpu & ~failed_ref & ((wo[:, 2] > 0) == (s[:, 2] > 0))
    assert both.mean() > 0.5
    close = (np.abs(s[both, 0:3] - wo[both]).max(-1) < 5e-4)
    close &= np.all(np.abs(s[both, 3:6] - f[both]) <= 5e-3 * np.abs(f[both]) + 1e-6, axis=1)
    close &= np.abs(s[both, 6] - pdf[both]) <= 5e-3 * pdf[both]
    close &= np.abs(s[both, 7] - cosv[both]) <= 5e-4
    assert close.mean() > 0.995, close.mean()
    # --- energy of the eval side, lobe by lobe: Int f |cos| dw by midpoint quadrature in (cos theta_o, phi)
    m_t, m_p = 1500, 3000
    ct = (np.arange(m_t) + 0.5) / m_t
    ph = (np.arange(m_p) + 0.5) / m_p * 2 * np.pi
    ctg, phg = np.meshgrid(ct, ph, indexing="ij")
    stg = np.sqrt(1 - ctg * ctg)
    dw = (1.0 / m_t) * (2 * np.pi / m_p)
    def lobe_integral(sign, col):
        og = np.stack([stg * np.cos(phg), stg * np.sin(phg), sign * ctg], -1)
        return (_ggx_eval(i, og, n, rough, col, metallic, ior, eta) * ctg[..., None]).sum((0, 1)) * dw
    e_refl, e_trans = lobe_integral(1.0, color), lobe_integral(-1.0, color)
    assert np.all(e_refl <= 1.0 + 1e-3) and np.all(e_refl > 0.02), e_refl
    assert np.all(e_trans <= (1.0 / eta) ** 2 * (1.0 + 1e-3)), e_trans    # radiance grows by (n_t / n_i)^2 across the interface
    if metallic > 0.99:
        assert not e_trans.any()
    # --- the sampler's mean weight E[f cos / pdf] against the eval side's integral.  They agree where the reference's two
    # sides are consistent: a metal, with F = 1 on the eval side too (what its sampler carries, mirror.rs:225-231).  For
    # glass they do NOT, in the reference itself: the sampler's transmission weight is (1 - F) G / G1 <= 1 by construction,
    # while btdf() (mirror.rs:90-124) takes |i.n|, |o.n| and has no test that the half vector faces the incident side, so it
    # is non-zero for directions no microfacet refracts into and its integral exceeds the sampler's mean (rough glass at
    # grazing incidence: 1.15 against 0.89).  That inconsistency is the reference's and is reproduced on both sides; what
    # is PINNED here (round 5; round 4 printed the figures) is that the device carries exactly it:
    #   (1) the device sampler's mean weight = the numpy sampler's mean weight (same draws), 4 sigma of the paired difference;
    #   (2) the device's eval-side integral = the numpy eval-side integral on the same quadrature grid;
    #   (3) the gap between the two sides is the expected one (GLASS_GAP: a "fix" of either side on the device, or in the
    #       numpy restatement, moves it) -- DESIGN.md 1 lists the rows.
    w = s[:, 3:6] * (s[:, 7] / s[:, 6])[:, None]
    mean, sem = w.mean(0), w.std(0, ddof=1) / np.sqrt(ns)
    assert np.all(mean <= 1.0 + 4.0 * sem)                                  # F G2 / G1 <= 1 per lobe (Heitz 2018)
    w_ref = f * (cosv / pdf)[:, None]
    mean_ref = w_ref.mean(0)
    agree = (failed_gpu == failed_ref) & ((wo[:, 2] > 0) == (s[:, 2] > 0))   # draws on which no lobe pick / validity test flipped in f32
    dw_pair = (w - w_ref)[agree]
    sem_pair = dw_pair.std(0, ddof=1) / np.sqrt(len(dw_pair))
    assert np.all(np.abs(dw_pair.mean(0)) <= 4.0 * sem_pair + 2e-4), (mat, cos_i, dw_pair.mean(0), sem_pair)
    assert np.all(np.abs(mean - mean_ref) <= 4.0 * sem + 2e-3 * (1.0 - agree.mean()) + 2e-4), (mat, cos_i, mean, mean_ref, sem)
    # (2) eval side on the device: the same midpoint rule on a grid the debug entry takes in one call
    g_t, g_p = 500, 1000
    gct = (np.arange(g_t) + 0.5) / g_t
    gph = (np.arange(g_p) + 0.5) / g_p * 2 * np.pi
    gctg, gphg = np.meshgrid(gct, gph, indexing="ij")
    gstg = np.sqrt(1 - gctg * gctg)
    gdw = (1.0 / g_t) * (2 * np.pi / g_p)
    dev_total, ref_total = np.zeros(3), np.zeros(3)
    for sign in (1.0, -1.0):
        og = np.stack([gstg * np.cos(gphg), gstg * np.sin(gphg), sign * gctg], -1).reshape(-1, 3)
        col = [1.0, 1.0, 1.0] if metallic > 0.99 else color
        if metallic > 0.99 and sign < 0:
            continue
        inp = np.concatenate([np.tile(dir_in, (len(og), 1)), og, np.tile(n, (len(og), 1)), np.full((len(og), 1), eta)], 1)
        if metallic > 0.99:      # (the metal's eval side is compared with F = 1, which needs another material record: numpy only)
            ref_total += (_ggx_eval(i, og, n, rough, col, metallic, ior, eta) * gctg.reshape(-1, 1)).sum(0) * gdw
            continue
        got_f = gpu_ctx.debug_bsdf_eval(0, inp, exact_math=0).astype(np.float64)[:, :3]
        dev_total += (got_f * gctg.reshape(-1, 1)).sum(0) * gdw
        ref_total += (_ggx_eval(i, og, n, rough, col, metallic, ior, eta) * gctg.reshape(-1, 1)).sum(0) * gdw
    if metallic > 0.99:
        total = lobe_integral(1.0, [1.0, 1.0, 1.0])
        assert np.all(np.abs(mean - total) <= 4.0 * sem + 5e-3 * total), (mat, cos_i, mean, total, (mean - total) / sem)
    else:
        total = e_refl + e_trans
        assert np.all(np.abs(dev_total - ref_total) <= 2e-3 * ref_total), (mat, cos_i, dev_total, ref_total)
        assert np.all(np.abs(ref_total - total) <= 1e-2 * total), (mat, cos_i, ref_total, total)          # the coarser grid is fine enough
        # (3) the reference's own inconsistency, as a committed expectation: eval-side integral - sampler's mean weight
        gap = float((total - mean_ref).mean())
        lo, hi = GLASS_GAP[(mat, cos_i)]
        assert lo <= gap <= hi, (mat, cos_i, gap, total, mean_ref)
    print(f"{mat}, cos_i {cos_i}: sampler mean weight {mean} (numpy {mean_ref}), eval-side integral {total}")


# ------------------------------------------------------------------ (iv) furnace: a convex Lambertian body inside a uniformly emitting sphere
def _inside_sphere_light_mis_factor(x, n, c_light, m=1200):
    """What MisStrategy::ray_color returns, in units of rho * Le, at a point x (normal n) of a convex Lambertian body INSIDE a sphere
    light, by midpoint quadrature over the hemisphere of n.  From inside, SphereShape::sample_surface_from_point (shape.rs:91-145)
    has sin^2(theta_max) = r^2 / d^2 > 1, so cos_theta_max clamps to 0: it draws `direction` in the hemisphere about w = towards the
    centre at pdf 1 / (2 pi), takes the root -half_b - sqrt(disc) -- negative from inside, the point BEHIND x -- and returns
    light_dir = normalize(point - x) = -direction.  So NEE only ever sends shadow rays into the hemisphere that faces AWAY from the
    centre, while the look-ahead pdf of a BSDF-sampled emitter hit (rendering.rs:117, same function with target_hit) is 1 / (2 pi)
    in EVERY direction: towards the centre's side the emitter-hit term is weighted p_b / (p_b + p_l) < 1 and nothing makes up
    the rest.  The reference loses that energy; so must the kernels."""
    w = _norm(np.asarray(c_light) - x)
    up = np.array([1.0, 0.0, 0.0]) if abs(n[1]) > 0.999 else np.array([0.0, 1.0, 0.0])
    t = _norm(np.cross(up, n))
    b = np.cross(n, t)
    u = (np.arange(m) + 0.5) / m
    th, ph = np.meshgrid(0.5 * np.pi * u, 2.0 * np.pi * u, indexing="ij")
    dirs = (np.sin(th) * np.cos(ph))[..., None] * t + (np.sin(th) * np.sin(ph))[..., None] * b + np.cos(th)[..., None] * n
    cos = np.cos(th)
    p_b, p_l = cos / np.pi, 1.0 / (2.0 * np.pi)
    weight = p_b / (p_b + p_l) + np.where(dirs @ w <= 0.0, p_l / (p_l + p_b), 0.0)
    dw = np.sin(th) * (0.5 * np.pi / m) * (2.0 * np.pi / m)
    return float((cos / np.pi * weight * dw).sum())


@pytest.mark.parametrize("integrator", [0, 1], ids=["mis", "brdf_only"])
@pytest.mark.parametrize("aim", [(0.0, 0.0), (0.3, 0.2), (-0.42, 0.1)], ids=["centre", "off-centre", "near the rim"])
def test_furnace_a_convex_body_inside_a_sphere_light(pt, gpu_ctx, integrator, aim):
    """Inside a sphere that emits Le uniformly, a CONVEX Lambertian body sees Le from every direction of its hemisphere and nothing
    of itself: the radiance it sends back is rho * Le whatever the point, the normal or the view -- a closed form for a whole
    estimator.  BrdfOnlyStrategy (rendering.rs:214-265) returns exactly that.  MisStrategy does NOT: the reference samples a sphere
    light seen from inside over half of the directions only and weights the other half as if it sampled them too
    (_inside_sphere_light_mis_factor; not among SURVEY's Q1-Q10) -- 25-45 % of the energy is lost, depending on how the normal lies
    to the direction of the centre.  Both are pinned here without oracle/: the MIS mean against the quadrature of the reference's
    own estimator, with the unbiased value excluded.  One bounce, min_depth 4: no roulette.  2^20 streams, 4 sigma."""
    le = np.array([1.7, 1.1, 0.6])
    rho = np.array([0.8, 0.5, 0.2])
    c_light, c_body, r_body = np.array([0.1, -0.2, -0.3]), np.array([0.0, 0.0, -2.0]), 0.5
    objs = pt.make_objects([(SPH, list(c_light) + [9.0], EMISSIVE, list(le)),           # the enclosure (camera and body inside)
                            (SPH, list(c_body) + [r_body], LAMBERT, list(rho))])
    gpu_ctx.upload(objs)
    origin = np.array([0.0, 0.0, 1.0])
    target = np.array([aim[0], aim[1], -2.0])
    d = _norm(target - origin)
    ids, _ = gpu_ctx.debug_hit_scene(np.concatenate([origin, d])[None, :], 1e-3, float("inf"))
    assert int(ids[0]) == 1
    oc = origin - c_body
    hb = oc @ d
    x = origin + (-hb - np.sqrt(hb * hb - (oc @ oc - r_body * r_body))) * d
    n = (x - c_body) / r_body
    mean, sem = _gpu_mean(gpu_ctx, pt, origin, target, integrator=integrator)
    if integrator == 1:
        _check(mean, sem, rho * le, f"furnace, BRDF only, {aim}")
    else:
        k = _inside_sphere_light_mis_factor(x, n, c_light)
        assert 0.4 < k < 0.9
        _check(mean, sem, rho * le * k, f"furnace, MIS, {aim}")
        assert np.all(np.abs(mean - rho * le) > 100.0 * sem)            # the unbiased value is far outside
    # a ray that misses the body sees the enclosure itself
    mean, sem = _gpu_mean(gpu_ctx, pt, origin, np.array([2.0, 1.5, -2.0]), n=1 << 12, integrator=integrator)
    assert np.abs(mean - le).max() <= 1e-5


def _oren_nayar_albedo(sigma, i, n, m=1500):
    """Int f_OrenNayar(i, o) cos_o d omega_o / rho for incoming direction i (pointing away from the surface), material.rs:221-265"""
    s2 = sigma * sigma
    A, B = 1.0 - 0.5 * s2 / (s2 + 0.33), 0.45 * s2 / (s2 + 0.09)
    up = np.array([1.0, 0.0, 0.0]) if abs(n[1]) > 0.999 else np.array([0.0, 1.0, 0.0])
    t = _norm(np.cross(up, n))
    b = np.cross(n, t)
    u = (np.arange(m) + 0.5) / m
    th, ph = np.meshgrid(0.5 * np.pi * u, 2.0 * np.pi * u, indexing="ij")
    co, so = np.cos(th), np.sin(th)
    ci = max(float(i @ n), 0.0)
    si = np.sqrt(max(1.0 - ci * ci, 0.0))
    phi_i = np.arctan2(i @ b, i @ t)
    cphi = np.maximum(np.cos(phi_i - ph), 0.0)
    first = ci > co
    tan_beta = np.where(first, si / ci if ci > 1e-6 else 0.0, np.where(co > 1e-6, so / np.maximum(co, 1e-300), 0.0))
    sin_alpha = np.where(first, so, si)
    f = (A + B * cphi * sin_alpha * tan_beta) / np.pi
    return float((f * co * so * (0.5 * np.pi / m) * (2.0 * np.pi / m)).sum())


@pytest.mark.parametrize("min_depth", [4, 0], ids=["no roulette", "roulette from the first vertex"])
@pytest.mark.parametrize("sigma", [0.0, 0.6, 1.0])
def test_furnace_oren_nayar_body_brdf_only(pt, gpu_ctx, sigma, min_depth):
    """The furnace with an OrenNayar body (material.rs:166-296) under BrdfOnlyStrategy: the radiance sent back is Le times the
    directional albedo  Int f(i, o) cos_o d omega_o  of the reference's BRDF for the view direction -- by quadrature of the formula --,
    with and without Russian roulette at the first vertex (rendering.rs:246-262: survivors are divided by the survival
    probability, so the mean must not move).  sigma = 0 is the Lambertian value rho."""
    le = np.array([1.7, 1.1, 0.6])
    rho = np.array([0.8, 0.5, 0.2])
    c_body, r_body = np.array([0.0, 0.0, -2.0]), 0.5
    objs = pt.make_objects([(SPH, [0.1, -0.2, -0.3, 9.0], EMISSIVE, list(le)),
                            (SPH, list(c_body) + [r_body], 3, list(rho) + [sigma])])
    gpu_ctx.upload(objs)
    origin, target = np.array([0.0, 0.0, 1.0]), np.array([0.33, -0.18, -2.0])
    d = _norm(target - origin)
    oc = origin - c_body
    hb = oc @ d
    x = origin + (-hb - np.sqrt(hb * hb - (oc @ oc - r_body * r_body))) * d
    n = (x - c_body) / r_body
    k = _oren_nayar_albedo(sigma, -d, n)
    if sigma == 0.0:
        assert abs(k - 1.0) < 1e-5
    mean, sem = _gpu_mean(gpu_ctx, pt, origin, target, integrator=1, min_depth=min_depth)
    _check(mean, sem, rho * le * k, f"OrenNayar furnace sigma {sigma} min_depth {min_depth}")


@pytest.mark.parametrize("aim", [(0.0, 0.0), (0.36, -0.2)], ids=["centre", "oblique"])
def test_furnace_metal_body_brdf_only(pt, gpu_ctx, aim):
    """The furnace with a GGX metal body (mirror.rs) under BrdfOnlyStrategy: one vertex on the body, the sampled direction then meets
    the enclosure, so the radiance is Le * E[f cos / pdf] over Mirror::bsdf_pdf_sample's own draws -- evaluated here with the numpy
    restatement of the sampler (_ggx_sample, mirror.rs:200-305) on an independent set of uniforms; the integrator hands the material
    eta_ratio = 1 / ior on a front face (rendering.rs:20-25).  Pins how the integrator applies the sampler's weight (the cosine, the
    F = 1 of a metal's sampled reflection, a failed sample = nothing); 4 sigma of both Monte-Carlo errors together."""
    rough, color, metallic, ior = GGX_MATS["metal"]
    le = np.array([1.7, 1.1, 0.6])
    c_body, r_body = np.array([0.0, 0.0, -2.0]), 0.5
    gpu_ctx.upload(pt.make_objects([(SPH, [0.1, -0.2, -0.3, 9.0], EMISSIVE, list(le)),
                                    (SPH, list(c_body) + [r_body], MIRROR, [rough] + color + [metallic, ior])]))
    origin, target = np.array([0.0, 0.0, 1.0]), np.array([aim[0], aim[1], -2.0])
    d = _norm(target - origin)
    oc = origin - c_body
    hb = oc @ d
    x = origin + (-hb - np.sqrt(hb * hb - (oc @ oc - r_body * r_body))) * d
    n = (x - c_body) / r_body
    rng = np.random.default_rng(5)
    k = 1 << 20
    _, f, pdf, cosv = _ggx_sample(-d, n, rough, color, metallic, ior, 1.0 / ior, rng.random(k), rng.random(k), rng.random(k))
    w = f * (cosv / pdf)[:, None]
    want, want_sem = w.mean(0), w.std(0, ddof=1) / np.sqrt(k)
    mean, sem = _gpu_mean(gpu_ctx, pt, origin, target, integrator=1)
    tol = 4.0 * np.sqrt(sem ** 2 + (le * want_sem) ** 2) + 2e-4 * le * want
    assert np.all(np.abs(mean - le * want) <= tol), (mean, le * want, (mean - le * want) / np.sqrt(sem ** 2 + (le * want_sem) ** 2))
    assert np.all(want > 0.3) and np.all(want < 1.0)


# ------------------------------------------------------------------ (v) an integrating sphere: infinitely many bounces in closed form
def _direct_split(s, m=200_000):
    """The one-bounce radiance rho * Le * s of a wall point under the central sphere light, split into its NEE part and its
    BSDF-sampled part by the MIS weights of rendering.rs:73,117 (one light: they sum to 1), as fractions of rho * Le.
    The light fills the cone of half-angle asin(sqrt s) about the normal; p_l = 1 / (2 pi (1 - cos_max)), p_b = cos / pi."""
    cmax = np.sqrt(1.0 - s)
    th = (np.arange(m) + 0.5) / m * np.arccos(cmax)
    dth = np.arccos(cmax) / m
    p_l, p_b = 1.0 / (2.0 * np.pi * (1.0 - cmax)), np.cos(th) / np.pi
    ring = np.cos(th) / np.pi * 2.0 * np.pi * np.sin(th) * dth
    return float((ring * p_l / (p_l + p_b)).sum()), float((ring * p_b / (p_b + p_l)).sum())


@pytest.mark.parametrize("case", ["mis, no roulette", "mis, roulette from depth 4 (Q1)", "brdf only, roulette from depth 4", "brdf only, roulette from depth 0"])
def test_integrating_sphere_every_bounce_in_closed_form(pt, gpu_ctx, case):
    """A Lambertian sphere of radius R seen from inside, lit by a small emissive sphere of radius r at its centre.  By symmetry the
    wall's radiance L is the same everywhere, and a wall point receives  pi Le s  from the light (s = (r / R)^2: a sphere of radiance
    Le centred on the normal) plus  pi (1 - s) L  from the rest of the wall (the light hides the cone it fills):
        L = rho (Le s + (1 - s) L)   =>   L = rho Le s / (1 - rho (1 - s))
    -- all orders of interreflection, exactly.  What the estimators make of it:
      * MisStrategy without roulette (min_depth = max_depth = 50; the 50 bounces carry all but rho^50 of it): L.
      * BrdfOnlyStrategy with roulette: L too -- a survivor is divided by its survival probability (rendering.rs:237-262).
      * MisStrategy WITH roulette (the reference's defaults): less.  A path killed at a vertex returns zero (rendering.rs:100-102),
        which drops that vertex's NEE term as well: with D_nee + D_bsdf = rho Le s the direct light split by the MIS
        weights, and rr_d = min(1, lum(T_d rho)) the survival probability at depth d >= 4 (T_{d+1} = T_d rho / rr_d),
            E_d = rr_d D_nee + D_bsdf + rho (1 - s) E_{d+1}.
    2^20 streams per case, 4 sigma; the MIS roulette case must also EXCLUDE the unbiased L."""
    R, r = 2.0, 0.3
    le = np.array([20.0, 14.0, 9.0])
    rho = np.array([0.7, 0.5, 0.25])
    s = (r / R) ** 2
    gpu_ctx.upload(pt.make_objects([(SPH, [0, 0, 0, R], LAMBERT, list(rho)), (SPH, [0, 0, 0, r], EMISSIVE, list(le))]))
    origin = np.array([0.6, 0.3, 0.2])
    target = origin + np.array([1.0, 0.4, 0.3])
    L = rho * le * s / (1.0 - rho * (1.0 - s))
    mis = case.startswith("mis")
    min_depth = 50 if "no roulette" in case else 0 if "depth 0" in case else 4
    want = L
    if mis and min_depth == 4:
        d_nee, d_bsdf = _direct_split(s)
        assert abs(d_nee + d_bsdf - s) < 1e-6 * s
        T, rr = np.ones(3), []
        for d in range(90):                                    # T_d and rr_d down the path (rendering.rs:89-98); (rho (1 - s))^90 ~ 1e-14
            nxt = T * rho
            lum = min(1.0, 0.2126 * nxt[0] + 0.7152 * nxt[1] + 0.0722 * nxt[2])
            p = 1.0 if d < 4 else lum * 0.5 ** (d - 4) if d >= 50 else lum
            rr.append(p)
            T = np.minimum(nxt / p, 1e30)                       # (beyond depth 50 the carried throughput explodes; its luminance clamps at 1)
        E = np.zeros(3)
        for d in reversed(range(90)):
            E = rho * le * (rr[d] * d_nee + d_bsdf) + rho * (1.0 - s) * E
        want = E
        assert want[0] < 0.9 * L[0] and np.all(want < L)        # Q1 costs the red channel (albedo 0.7: long paths) 12 %
    mean, sem = _gpu_mean(gpu_ctx, pt, origin, target, integrator=0 if mis else 1, min_depth=min_depth, max_depth=50)
    _check(mean, sem, want, case)
    if mis and min_depth == 4:
        assert abs(mean[0] - L[0]) > 20.0 * sem[0]


@pytest.mark.parametrize("case", ["mis, no roulette", "brdf only, roulette from depth 4"])
def test_integrating_sphere_image_through_the_throughput_kernels(pt, gpu_ctx, case):
    """The same closed form through the whole-frame path -- camera rays, the regenerating path kernel (a batch of 2^22 paths), the
    film resolve -- instead of pt_ray_color's pixel lists: a camera inside the integrating sphere that looks away from the light
    sees the wall in every pixel, so every pixel's expectation is L = rho Le s / (1 - rho (1 - s)).  256 x 256 x 64 spp; the mean
    over the pixels against L at 4 standard errors (pixels are independent), and no pixel far from it."""
    R, r = 2.0, 0.3
    le, rho = np.array([20.0, 14.0, 9.0]), np.array([0.7, 0.5, 0.25])
    s = (r / R) ** 2
    gpu_ctx.upload(pt.make_objects([(SPH, [0, 0, 0, R], LAMBERT, list(rho)), (SPH, [0, 0, 0, r], EMISSIVE, list(le))]))
    cam = pt.camera_look_at((0.6, 0.3, 0.2), (1.6, 0.7, 0.5), (0.0, 1.0, 0.0), 256, 256, 35.0)
    mis = case.startswith("mis")
    prm = pt.default_params(spp=64, integrator=0 if mis else 1, min_depth=50 if mis else 4, max_depth=50)
    lin, _ = gpu_ctx.render(cam, prm)
    st = gpu_ctx.stats()
    assert st.samples == 256 * 256 * 64 and st.bounce_launches == 1
    img = lin.cpu().numpy().astype(np.float64).reshape(-1, 3)
    L = rho * le * s / (1.0 - rho * (1.0 - s))
    mean, sem = img.mean(0), img.std(0, ddof=1) / np.sqrt(img.shape[0])
    _check(mean, sem, L, case)
    assert np.all(np.abs(img - L) < 60.0 * sem * np.sqrt(img.shape[0]))          # every pixel within its own spread of L
