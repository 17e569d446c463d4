"""Thin Python surface over the C ABI, used by tests/, bench.py and smoke().

The product's host language is C++ (pathtrace_amd/host/pathtrace.hpp mirrors the
reference's Camera/World/Object surface); this module only marshals arguments.
Device buffers come from torch (device memory + streams are what torch is here for).
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import (PtCamera, PtObject, PtRenderParams, PtStats, check, lib)


def _d3(v):
    return (C.c_double * 3)(*[float(x) for x in v])


def camera_new(origin=(0.0, 0.0, 2.0), width=400, height=400, screen_distance=1.0, fov_degrees=35.0):
    """Camera::new (src/camera.rs:50-82); defaults = World::new's camera (src/world.rs:67-73)."""
    cam = PtCamera()
    check(lib().pt_camera_new(_d3(origin), width, height, screen_distance, fov_degrees, C.byref(cam)))
    return cam


def camera_look_at(origin, target, up, width, height, fov_degrees):
    """Camera::look_at (src/camera.rs:94-130)."""
    cam = PtCamera()
    check(lib().pt_camera_look_at(_d3(origin), _d3(target), _d3(up), width, height, fov_degrees, C.byref(cam)))
    return cam


def default_params(**over):
    """Reference constants (world.rs:18, rendering.rs:6-7) with overrides."""
    p = PtRenderParams()
    lib().pt_default_params(C.byref(p))
    for k, v in over.items():
        if not hasattr(p, k):
            raise AttributeError(f"PtRenderParams has no field {k}")
        setattr(p, k, v)
    return p


def builtin_scene(scene_id, arg=0):
    """Scenes of SURVEY 8(d): 1 reference Cornell box, 2 ten-sphere Cornell, 4 random spheres (arg = n)."""
    n = C.c_uint32(0)
    check(lib().pt_builtin_scene(scene_id, arg, None, 0, C.byref(n)))
    objs = (PtObject * n.value)()
    check(lib().pt_builtin_scene(scene_id, arg, objs, n.value, C.byref(n)))
    return objs


def make_objects(specs):
    """specs: iterable of (shape_tag, shape_values, mat_tag, mat_values) -> PtObject array."""
    specs = list(specs)
    objs = (PtObject * len(specs))()
    for o, (st, sv, mt, mv) in zip(objs, specs):
        o.shape_tag, o.mat_tag = st, mt
        for i, x in enumerate(sv):
            o.shape[i] = float(x)
        for i, x in enumerate(mv):
            o.mat[i] = float(x)
    return objs


def tile_rows(height, band_rows, band_index, band_count):
    return int(lib().pt_tile_rows(height, band_rows, band_index, band_count))


def tile_row_indices(height, band_rows, band_index, band_count):
    """Image rows of the tile, ascending (host mirror of the partition rule in pathtrace_amd.h)."""
    br = band_rows if band_rows else max(height, 1)
    bc = band_count if band_count else 1
    return [y for y in range(height) if (y // br) % bc == band_index]


class Context:
    """One GPU context (pt_context_create).  Fails loudly without a HIP device."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        check(lib().pt_context_create(device, C.byref(self._h)))
        self.device = device
        self._objs = None

    def close(self):
        if self._h:
            lib().pt_context_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, objs):
        self._objs = objs   # keep alive
        check(lib().pt_scene_upload(self._h, objs, len(objs)))

    def set_stream(self, hip_stream_ptr):
        """Render on a caller-owned stream.  0 is the handle of HIP's legacy default stream (torch's default stream):
        it is passed on as PT_STREAM_LEGACY_DEFAULT, so the render is ordered against the caller's other work there;
        None restores the context's own stream."""
        if hip_stream_ptr is None:
            ptr = None
        else:
            ptr = C.c_void_p(hip_stream_ptr if hip_stream_ptr else _lib.PT_STREAM_LEGACY_DEFAULT)
        check(lib().pt_context_set_stream(self._h, ptr))

    def set_tuning(self, export_below=0, bvh_refill=0, bvh_leaf=0, cont_workgroups=0, level0_form=0, regen_workgroups=0, in_order=0):
        """Scheduling knobs (pt_context_set_tuning); 0 = library default.  Results never depend on them."""
        t = _lib.PtTuning(export_below, bvh_refill, bvh_leaf, cont_workgroups, level0_form, regen_workgroups, in_order)
        check(lib().pt_context_set_tuning(self._h, C.byref(t)))

    def render_into(self, cam, params, linear_ptr, rgba_ptr):
        """pt_render_device on raw device pointers (asynchronous; call sync())."""
        check(lib().pt_render_device(self._h, C.byref(cam), C.byref(params), C.c_void_p(linear_ptr),
                                     C.c_void_p(rgba_ptr) if rgba_ptr else None))

    def fail_after(self, n):
        """Test hook (pt_debug_fail_after): the n-th stream operation of the NEXT render fails as a HIP call would; n < 0: none."""
        check(lib().pt_debug_fail_after(self._h, int(n)))

    def scan_layout(self):
        """-> (spheres, single triangles, triangle pairs) one linear scan of the uploaded scene tests (pt_debug_scan_layout)"""
        a, b, c = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
        check(lib().pt_debug_scan_layout(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def render_packed_into(self, cam, params, packed_ptr):
        """pt_render_device_packed: the tile as 16 B per pixel (linear RGB + RGBA8), the send-buffer form of the film gather."""
        check(lib().pt_render_device_packed(self._h, C.byref(cam), C.byref(params), C.c_void_p(packed_ptr)))

    def sync(self):
        check(lib().pt_sync(self._h))

    def stats(self):
        s = PtStats()
        check(lib().pt_get_stats(self._h, C.byref(s)))
        return s

    def render(self, cam, params, want_rgba=True):
        """Render the tile into fresh torch device tensors; returns (linear[rows,W,3] f32, rgba[rows,W,4] u8)."""
        import torch
        rows = tile_rows(cam.height, params.band_rows, params.band_index, params.band_count or 1)
        dev = torch.device("cuda", self.device)
        lin = torch.empty((rows, cam.width, 3), dtype=torch.float32, device=dev)
        rgba = torch.empty((rows, cam.width, 4), dtype=torch.uint8, device=dev) if want_rgba else None
        self.set_stream(torch.cuda.current_stream(dev).cuda_stream)
        self.render_into(cam, params, lin.data_ptr(), rgba.data_ptr() if want_rgba else 0)
        self.sync()
        return lin, rgba

    def render_progressive(self, cam, params, spp_step, on_frame=None):
        """pt_render_progressive: on_frame(spp_done, spp_total, rgba[rows,W,4], linear[rows,W,3]) -> truthy to stop."""
        rows = tile_rows(cam.height, params.band_rows, params.band_index, params.band_count or 1)
        lin = np.zeros((rows, cam.width, 3), dtype=np.float32)
        rgba = np.zeros((rows, cam.width, 4), dtype=np.uint8)

        def _cb(user, done, total, p8, pf):
            return int(bool(on_frame(done, total, rgba.copy(), lin.copy()))) if on_frame else 0

        cb = _lib.PROGRESS_FN(_cb)
        check(lib().pt_render_progressive(self._h, C.byref(cam), C.byref(params), spp_step, C.cast(cb, C.c_void_p), None,
                                          lin.ctypes.data_as(C.c_void_p), rgba.ctypes.data_as(C.c_void_p)))
        return lin, rgba

    def debug_hit_scene(self, rays, t_min=0.001, t_max=float("inf"), exact_math=0, accel=0):
        rays = np.ascontiguousarray(rays, dtype=np.float64).reshape(-1, 6)
        n = rays.shape[0]
        ids = np.empty(n, dtype=np.int32)
        ts = np.empty(n, dtype=np.float32)
        check(lib().pt_debug_hit_scene(self._h, rays.ctypes.data_as(C.POINTER(C.c_double)), n, t_min, t_max, exact_math, accel,
                                       ids.ctypes.data_as(C.POINTER(C.c_int32)),
                                       ts.ctypes.data_as(C.POINTER(C.c_float))))
        return ids, ts


def _f64(a, cols):
    return np.ascontiguousarray(a, dtype=np.float64).reshape(-1, cols)


def _pd(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


def _pu(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32)) if a is not None else None


def _pf(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class _ContextFunctions:
    """Function-level entries of the C ABI (pt_debug_*, pt_render_pixels, pt_ray_color): the per-vertex device
    functions on arbitrary inputs.  Mixed into Context."""

    def debug_hit_records(self, rays, t_min=0.001, t_max=float("inf"), exact_math=0, accel=0):
        """-> (ids int32[n], rec float32[n, 8] = t, point3, normal3, front_face)"""
        rays = _f64(rays, 6)
        n = rays.shape[0]
        ids = np.empty(n, dtype=np.int32)
        rec = np.empty((n, 8), dtype=np.float32)
        check(lib().pt_debug_hit_records(self._h, _pd(rays), n, t_min, t_max, exact_math, accel,
                                         ids.ctypes.data_as(C.POINTER(C.c_int32)), _pf(rec)))
        return ids, rec

    def debug_bsdf_eval(self, obj, inp, exact_math=0):
        """inp n x (ray dir3, wo3, normal3, eta) -> float32[n, 4] = f3, pdf"""
        inp = _f64(inp, 10)
        out = np.empty((inp.shape[0], 4), dtype=np.float32)
        check(lib().pt_debug_bsdf_eval(self._h, obj, _pd(inp), inp.shape[0], exact_math, _pf(out)))
        return out

    def debug_bsdf_sample(self, obj, inp, words, exact_math=0):
        """inp n x (ray dir3, normal3, eta), words n x 4 raw u32 (r1, r2, lobe, -) -> float32[n, 8] = wo3, f3, pdf, cos"""
        inp = _f64(inp, 7)
        words = np.ascontiguousarray(words, dtype=np.uint32).reshape(-1, 4)
        out = np.empty((inp.shape[0], 8), dtype=np.float32)
        check(lib().pt_debug_bsdf_sample(self._h, obj, _pd(inp), _pu(words), inp.shape[0], exact_math, _pf(out)))
        return out

    def debug_shape_sample(self, obj, frm, target=None, r12=None, exact_math=0):
        """-> float32[n, 8] = point3, pdf_omega, light_dir3, distance"""
        frm = _f64(frm, 3)
        tg = _f64(target, 3) if target is not None else None
        rr = _f64(r12, 2) if r12 is not None else None
        out = np.empty((frm.shape[0], 8), dtype=np.float32)
        check(lib().pt_debug_shape_sample(self._h, obj, _pd(frm), _pd(tg), _pd(rr), frm.shape[0], exact_math, _pf(out)))
        return out

    def debug_light_point(self, frm, words, exact_math=0):
        """World::sample_light_point: words n x 4 (index word, r1 word, r2 word, -) -> float32[n, 8] = point3, emission3,
        pdf, light object"""
        frm = _f64(frm, 3)
        words = np.ascontiguousarray(words, dtype=np.uint32).reshape(-1, 4)
        out = np.empty((frm.shape[0], 8), dtype=np.float32)
        check(lib().pt_debug_light_point(self._h, _pd(frm), _pu(words), frm.shape[0], exact_math, _pf(out)))
        return out

    def debug_camera_rays(self, cam, xys, exact_math=0):
        """xys n x (x, y film row, sample) -> float32[n, 8] = origin3, direction3, ox, oy"""
        xys = np.ascontiguousarray(xys, dtype=np.uint32).reshape(-1, 3)
        out = np.empty((xys.shape[0], 8), dtype=np.float32)
        check(lib().pt_debug_camera_rays(self._h, C.byref(cam), _pu(xys), xys.shape[0], exact_math, _pf(out)))
        return out

    def multi_emulate(self, n_virtual, cam, params):
        """pt_debug_multi_emulate: the frame an n_virtual-device pt_multi render assembles, produced on this one context."""
        lin = np.empty((cam.height, cam.width, 3), dtype=np.float32)
        rgba = np.empty((cam.height, cam.width, 4), dtype=np.uint8)
        check(lib().pt_debug_multi_emulate(self._h, n_virtual, C.byref(cam), C.byref(params), lin.ctypes.data_as(C.c_void_p),
                                           rgba.ctypes.data_as(C.c_void_p)))
        return lin, rgba

    def render_pixels(self, cam, params, xy, want_samples=False):
        """pt_render_pixels = World::render_pixel for a pixel list.  -> (linear f32[n,3], rgba u8[n,4], samples
        f32[n,spp,3] or None)"""
        xy = np.ascontiguousarray(xy, dtype=np.uint32).reshape(-1, 2)
        n = xy.shape[0]
        lin = np.empty((n, 3), dtype=np.float32)
        rgba = np.empty((n, 4), dtype=np.uint8)
        smp = np.empty((n, params.spp, 3), dtype=np.float32) if want_samples else None
        check(lib().pt_render_pixels(self._h, C.byref(cam), C.byref(params), _pu(xy), n, lin.ctypes.data_as(C.c_void_p),
                                     rgba.ctypes.data_as(C.c_void_p), smp.ctypes.data_as(C.c_void_p) if want_samples else None))
        return lin, rgba, smp

    def ray_color(self, params, rays, xy):
        """pt_ray_color = RenderingStrategy::ray_color(world, ray, 0, rng(key xy, sample spp_offset), 1) -> f32[n,3]"""
        rays = _f64(rays, 6)
        xy = np.ascontiguousarray(xy, dtype=np.uint32).reshape(-1, 2)
        assert xy.shape[0] == rays.shape[0]
        out = np.empty((rays.shape[0], 3), dtype=np.float32)
        check(lib().pt_ray_color(self._h, C.byref(params), _pd(rays), _pu(xy), rays.shape[0], out.ctypes.data_as(C.c_void_p)))
        return out


for _n, _f in vars(_ContextFunctions).items():
    if not _n.startswith("__"):
        setattr(Context, _n, _f)


class Multi:
    """pt_multi_*: ONE process, several GPUs, one RCCL gather of the film to the first device."""

    def __init__(self, devices, shared_device=None):
        """devices: HIP ordinals (distinct).  shared_device = d: the DEBUG object of len(devices) contexts that all sit on
        device d, the gather emulated by device-to-device copies (pt_debug_multi_create_shared; no RCCL)."""
        self._h = C.c_void_p()
        if shared_device is not None:
            check(lib().pt_debug_multi_create_shared(shared_device, len(devices), C.byref(self._h)))
            devices = [shared_device] * len(devices)
        else:
            arr = (C.c_int * len(devices))(*devices)
            check(lib().pt_multi_create(arr, len(devices), C.byref(self._h)))
        self.devices = list(devices)
        self._objs = None

    def set_threads(self, enabled):
        check(lib().pt_multi_set_threads(self._h, 1 if enabled else 0))

    def set_exchange(self, mode):
        """pt_multi_set_exchange: "rccl" (one ncclGather per frame, default) or "copy" (one DMA copy per device, no kernel)."""
        check(lib().pt_multi_set_exchange(self._h, {"rccl": _lib.PT_EXCHANGE_RCCL, "copy": _lib.PT_EXCHANGE_COPY}[mode]))

    def info(self):
        i = _lib.PtMultiInfo()
        check(lib().pt_multi_info(self._h, C.byref(i)))
        return i

    def close(self):
        if self._h:
            lib().pt_multi_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, objs):
        self._objs = objs
        check(lib().pt_multi_scene_upload(self._h, objs, len(objs)))

    def set_tuning(self, export_below=0, bvh_refill=0, bvh_leaf=0, cont_workgroups=0, level0_form=0, regen_workgroups=0, in_order=0):
        t = _lib.PtTuning(export_below, bvh_refill, bvh_leaf, cont_workgroups, level0_form, regen_workgroups, in_order)
        check(lib().pt_multi_set_tuning(self._h, C.byref(t)))

    def render_into(self, cam, params, linear_ptr, rgba_ptr):
        check(lib().pt_multi_render_device(self._h, C.byref(cam), C.byref(params), C.c_void_p(linear_ptr),
                                           C.c_void_p(rgba_ptr) if rgba_ptr else None))

    def sync(self):
        check(lib().pt_multi_sync(self._h))

    def stats(self):
        s = PtStats()
        check(lib().pt_multi_get_stats(self._h, C.byref(s)))
        return s

    def render_host(self, cam, params):
        lin = np.empty((cam.height, cam.width, 3), dtype=np.float32)
        rgba = np.empty((cam.height, cam.width, 4), dtype=np.uint8)
        check(lib().pt_multi_render_host(self._h, C.byref(cam), C.byref(params), lin.ctypes.data_as(C.c_void_p),
                                         rgba.ctypes.data_as(C.c_void_p)))
        return lin, rgba


def render_multi(devices, cam, objs, params):
    """pt_render_multi: the one-shot multi-GPU entry with host buffers."""
    arr = (C.c_int * len(devices))(*devices)
    lin = np.empty((cam.height, cam.width, 3), dtype=np.float32)
    rgba = np.empty((cam.height, cam.width, 4), dtype=np.uint8)
    check(lib().pt_render_multi(arr, len(devices), C.byref(cam), objs, len(objs), C.byref(params),
                                lin.ctypes.data_as(C.c_void_p), rgba.ctypes.data_as(C.c_void_p)))
    return lin, rgba


def bvh_check(objs):
    """pt_debug_bvh_check (host only): build + verify the accel = 1 BVH; returns (depth, nodes, leaf slots)."""
    d, nn, nl = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
    check(lib().pt_debug_bvh_check(objs, len(objs), C.byref(d), C.byref(nn), C.byref(nl)))
    return d.value, nn.value, nl.value


def render_host(cam, objs, params):
    """pt_render: the one-shot host-buffer entry (= src/main.rs:43-60)."""
    rows = tile_rows(cam.height, params.band_rows, params.band_index, params.band_count or 1)
    lin = np.empty((rows, cam.width, 3), dtype=np.float32)
    rgba = np.empty((rows, cam.width, 4), dtype=np.uint8)
    check(lib().pt_render(C.byref(cam), objs, len(objs), C.byref(params), lin.ctypes.data_as(C.c_void_p),
                          rgba.ctypes.data_as(C.c_void_p)))
    return lin, rgba
