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
        check(lib().pt_context_set_stream(self._h, C.c_void_p(hip_stream_ptr)))

    def render_into(self, cam, params, linear_ptr, rgba_ptr):
        """pt_render_device on raw device pointers (asynchronous; call sync())."""
        check(lib().pt_render_device(self._h, C.byref(cam), C.byref(params), C.c_void_p(linear_ptr),
                                     C.c_void_p(rgba_ptr) if rgba_ptr else None))

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
