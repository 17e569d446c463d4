"""ctypes binding of the CPU ORACLE (oracle/liborc.so).  TEST INFRASTRUCTURE ONLY:
importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg,
never from pathtrace_amd/.  See oracle/pt_oracle.hpp for what pins the oracle."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liborc.so")
_lib = None

F64, F32 = 64, 32
RECURSIVE, ITERATIVE = 0, 1


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} not found: run `make -C oracle` (or __graft_entry__.build())")
        _lib = C.CDLL(LIB_PATH)
        _lib.orc_u01.restype = C.c_double
        _lib.orc_u01.argtypes = [C.c_uint32]
        _lib.orc_vec3.restype = C.c_double
        _lib.orc_vec3.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p]
        _lib.orc_tile_rows.restype = C.c_uint32
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def philox(ctr, key, rounds=None):
    """Philox4x32 with `rounds` rounds; default = the round count of the render draws (orc_draw_rounds)."""
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().orc_philox4x32(c, k, C.c_int(lib().orc_draw_rounds() if rounds is None else rounds), o)
    return [int(x) for x in o]


def u01(r):
    return lib().orc_u01(C.c_uint32(r))


def rr_word(ds):
    """Roulette word of a vertex from its four BLK_SURFACE words (pt_oracle.hpp rr_word)."""
    f = lib().orc_rr_word
    f.restype = C.c_uint32
    return int(f((C.c_uint32 * 4)(*[int(v) for v in ds])))


def render(cam, objs, params, precision=F64, form=RECURSIVE, threads=1):
    """-> (linear float64[rows,W,3], rgba uint8[rows,W,4], counters dict)"""
    rows = lib().orc_tile_rows(C.c_uint32(cam.height), C.c_uint32(params.band_rows), C.c_uint32(params.band_index),
                               C.c_uint32(params.band_count or 1))
    lin = np.zeros((rows, cam.width, 3), dtype=np.float64)
    rgba = np.zeros((rows, cam.width, 4), dtype=np.uint8)
    cnt = np.zeros(4, dtype=np.uint64)
    rc = lib().orc_render(C.byref(cam), objs, C.c_uint32(len(objs)), C.byref(params), C.c_int(precision),
                          C.c_int(form), C.c_int(threads), _p(lin), _p(rgba), _p(cnt))
    if rc:
        raise RuntimeError(f"orc_render failed: {rc}")
    return lin, rgba, {"vertices": int(cnt[0]), "shadow_rays": int(cnt[1]), "scans": int(cnt[2]),
                       "max_depth": int(cnt[3])}


def render_pixels(cam, objs, params, xy, precision=F64, form=RECURSIVE):
    """World::render_pixel for a pixel list -> (linear float64[n,3], samples float64[n,spp,3] in sample order)"""
    xy = np.ascontiguousarray(xy, dtype=np.uint32).reshape(-1, 2)
    n = xy.shape[0]
    lin = np.zeros((n, 3), dtype=np.float64)
    smp = np.zeros((n, params.spp, 3), dtype=np.float64)
    rc = lib().orc_render_pixels(C.byref(cam), objs, C.c_uint32(len(objs)), C.byref(params), C.c_int(precision), C.c_int(form),
                                 _p(xy), C.c_uint32(n), _p(lin), _p(smp))
    if rc:
        raise RuntimeError(f"orc_render_pixels failed: {rc}")
    return lin, smp


def render_stdrng(cam, objs, params, threads=1):
    """orc.render(F64, RECURSIVE) drawing from the reference's own generator: one sequential StdRng (ChaCha12) stream
    per pixel, seeded (y << 32) | x (main.rs:51-52).  Restated from the published algorithm, unverified against the
    rand crate.  -> (linear float64[rows,W,3], rgba uint8[rows,W,4], counters dict)"""
    rows = lib().orc_tile_rows(C.c_uint32(cam.height), C.c_uint32(params.band_rows), C.c_uint32(params.band_index),
                               C.c_uint32(params.band_count or 1))
    lin = np.zeros((rows, cam.width, 3), dtype=np.float64)
    rgba = np.zeros((rows, cam.width, 4), dtype=np.uint8)
    cnt = np.zeros(4, dtype=np.uint64)
    rc = lib().orc_render_stdrng(C.byref(cam), objs, C.c_uint32(len(objs)), C.byref(params), C.c_int(threads), _p(lin), _p(rgba), _p(cnt))
    if rc:
        raise RuntimeError(f"orc_render_stdrng failed: {rc}")
    return lin, rgba, {"vertices": int(cnt[0]), "shadow_rays": int(cnt[1]), "scans": int(cnt[2]), "max_depth": int(cnt[3])}


def render_pixels_stdrng(cam, objs, params, xy):
    """World::render_pixel for a pixel list from the sequential StdRng stream -> (linear float64[n,3], samples [n,spp,3])"""
    xy = np.ascontiguousarray(xy, dtype=np.uint32).reshape(-1, 2)
    n = xy.shape[0]
    lin = np.zeros((n, 3), dtype=np.float64)
    smp = np.zeros((n, params.spp, 3), dtype=np.float64)
    rc = lib().orc_render_pixels_stdrng(C.byref(cam), objs, C.c_uint32(len(objs)), C.byref(params), _p(xy), C.c_uint32(n), _p(lin), _p(smp))
    if rc:
        raise RuntimeError(f"orc_render_pixels_stdrng failed: {rc}")
    return lin, smp


def chacha_block(state16, rounds):
    i = (C.c_uint32 * 16)(*[int(v) for v in state16])
    o = (C.c_uint32 * 16)()
    lib().orc_chacha_block(i, C.c_int(rounds), o)
    return [int(v) for v in o]


def stdrng_seed_key(seed):
    k = (C.c_uint32 * 8)()
    lib().orc_stdrng_seed_key(C.c_uint64(seed), k)
    return [int(v) for v in k]


def stdrng_draw(seed, mode, n, arg=0, rounds=12):
    """mode: "u32" | "u64" | "f64" | "range" (arg = n of 0..n) | "mixed" (alternating u32, u64 -> two lists)"""
    m = {"u32": 0, "u64": 1, "f64": 2, "range": 3, "mixed": 4}[mode]
    o32 = np.zeros(n, dtype=np.uint32); o64 = np.zeros(n, dtype=np.uint64); of = np.zeros(n, dtype=np.float64)
    lib().orc_stdrng_draw(C.c_uint64(seed), C.c_int(rounds), C.c_int(m), C.c_uint32(arg), C.c_uint32(n), _p(o32), _p(o64), _p(of))
    return {0: o32, 1: o64, 2: of, 3: o32, 4: (o32, o64)}[m]


def hit_scene(objs, rays, t_min=0.001, t_max=float("inf"), precision=F64):
    rays = np.ascontiguousarray(rays, dtype=np.float64).reshape(-1, 6)
    n = rays.shape[0]
    ids = np.empty(n, dtype=np.int32)
    ts = np.empty(n, dtype=np.float64)
    pn = np.empty((n, 6), dtype=np.float64)
    ff = np.empty(n, dtype=np.uint8)
    lib().orc_hit_scene(objs, C.c_uint32(len(objs)), C.c_int(precision), _p(rays), C.c_uint32(n), C.c_double(t_min),
                        C.c_double(t_max), _p(ids), _p(ts), _p(pn), _p(ff))
    return ids, ts, pn, ff


def camera_rays(cam, xy, off, precision=F64):
    xy = np.ascontiguousarray(xy, dtype=np.uint32).reshape(-1, 2)
    off = np.ascontiguousarray(off, dtype=np.float64).reshape(-1, 2)
    out = np.empty((xy.shape[0], 6), dtype=np.float64)
    lib().orc_camera_rays(C.byref(cam), C.c_int(precision), C.c_uint32(xy.shape[0]), _p(xy), _p(off), _p(out))
    return out


def shape_sample(obj_array, frm, target=None, r12=None, precision=F64):
    frm = np.ascontiguousarray(frm, dtype=np.float64).reshape(-1, 3)
    n = frm.shape[0]
    tg = np.ascontiguousarray(target, dtype=np.float64).reshape(-1, 3) if target is not None else None
    rr = np.ascontiguousarray(r12, dtype=np.float64).reshape(-1, 2) if r12 is not None else None
    out = np.empty((n, 11), dtype=np.float64)
    lib().orc_shape_sample(obj_array, C.c_int(precision), _p(frm), _p(tg), _p(rr), C.c_uint32(n), _p(out))
    return out


def bsdf_eval(obj_array, inp, precision=F64):
    inp = np.ascontiguousarray(inp, dtype=np.float64).reshape(-1, 10)
    out = np.empty((inp.shape[0], 4), dtype=np.float64)
    lib().orc_bsdf_eval(obj_array, C.c_int(precision), _p(inp), C.c_uint32(inp.shape[0]), _p(out))
    return out


def bsdf_sample(obj_array, inp, draws, precision=F64):
    inp = np.ascontiguousarray(inp, dtype=np.float64).reshape(-1, 7)
    draws = np.ascontiguousarray(draws, dtype=np.uint32).reshape(-1, 4)
    out = np.empty((inp.shape[0], 8), dtype=np.float64)
    lib().orc_bsdf_sample(obj_array, C.c_int(precision), _p(inp), _p(draws), C.c_uint32(inp.shape[0]), _p(out))
    return out


def light_point(objs, frm, words, precision=F64):
    """World::sample_light_point: words n x 4 (index word, r1 word, r2 word, -) -> n x 8 = point3, emission3, pdf, light obj"""
    frm = np.ascontiguousarray(frm, dtype=np.float64).reshape(-1, 3)
    words = np.ascontiguousarray(words, dtype=np.uint32).reshape(-1, 4)
    out = np.empty((frm.shape[0], 8), dtype=np.float64)
    lib().orc_light_point(objs, C.c_uint32(len(objs)), C.c_int(precision), _p(frm), _p(words), C.c_uint32(frm.shape[0]), _p(out))
    return out


def ray_color(objs, params, rays, xy, precision=F64, form=RECURSIVE):
    """RenderingStrategy::ray_color(world, ray, 0, rng(key xy, sample params.spp_offset), 1) -> n x 3"""
    rays = np.ascontiguousarray(rays, dtype=np.float64).reshape(-1, 6)
    xy = np.ascontiguousarray(xy, dtype=np.uint32).reshape(-1, 2)
    out = np.empty((rays.shape[0], 3), dtype=np.float64)
    lib().orc_ray_color(objs, C.c_uint32(len(objs)), C.byref(params), C.c_int(precision), C.c_int(form), _p(rays), _p(xy),
                        C.c_uint32(rays.shape[0]), _p(out))
    return out


def sincos2pi(u, precision=F32):
    u = np.ascontiguousarray(u, dtype=np.float64)
    out = np.empty((u.shape[0], 2), dtype=np.float64)
    lib().orc_sincos2pi(C.c_int(precision), _p(u), C.c_uint32(u.shape[0]), _p(out))
    return out


def vec3(op, a, b=None, c=None, s=0.0, precision=F64):
    A = np.asarray(a, dtype=np.float64)
    B = np.asarray(b, dtype=np.float64) if b is not None else None
    Cc = np.asarray(c, dtype=np.float64) if c is not None else None
    out = np.zeros(3, dtype=np.float64)
    ret = lib().orc_vec3(C.c_int(precision), C.c_int(op), _p(A), _p(B), _p(Cc), C.c_double(s), _p(out))
    return ret, out


def trace_path(cam, objs, params, x, y, sample, precision=F64, max_vertices=128):
    """Per-vertex records (24 doubles, layout in oracle_capi.cpp) of one camera sample, iterative form."""
    rec = np.zeros((max_vertices, 24), dtype=np.float64)
    n = lib().orc_trace_path(C.byref(cam), objs, C.c_uint32(len(objs)), C.byref(params), C.c_int(precision),
                             C.c_uint32(x), C.c_uint32(y), C.c_uint32(sample), _p(rec), C.c_int(max_vertices))
    return rec[: min(n, max_vertices)]
