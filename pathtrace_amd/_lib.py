"""ctypes binding of the C ABI declared in include/pathtrace_amd.h.

This is the same binding a foreign-language host would write (see INTEGRATION.md
for the Rust `extern "C"` form).  Loading fails loudly when the shared library is
missing: there is no Python or CPU fallback for the rendering path.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PATHTRACE_AMD_LIB: alternative build of the same library (A/B experiments of kernel variants)
LIB_PATH = os.environ.get("PATHTRACE_AMD_LIB") or os.path.join(_HERE, "libpathtrace_amd.so")


class PtCamera(C.Structure):
    _fields_ = [
        ("origin", C.c_double * 3),
        ("lower_left", C.c_double * 3),
        ("horizontal", C.c_double * 3),
        ("vertical", C.c_double * 3),
        ("width", C.c_uint32),
        ("height", C.c_uint32),
    ]


class PtObject(C.Structure):
    _fields_ = [
        ("shape_tag", C.c_uint32),
        ("mat_tag", C.c_uint32),
        ("shape", C.c_double * 9),
        ("mat", C.c_double * 6),
    ]


class PtRenderParams(C.Structure):
    _fields_ = [
        ("spp", C.c_uint32),
        ("spp_offset", C.c_uint32),
        ("min_depth", C.c_uint32),
        ("max_depth", C.c_uint32),
        ("integrator", C.c_uint32),
        ("t_min", C.c_double),
        ("band_rows", C.c_uint32),
        ("band_index", C.c_uint32),
        ("band_count", C.c_uint32),
        ("max_paths_in_flight", C.c_uint64),
        ("profile", C.c_uint32),
        ("workgroups", C.c_uint32),
        ("exact_math", C.c_uint32),
        ("accel", C.c_uint32),
        ("n_devices", C.c_uint32),
    ]


class PtTuning(C.Structure):
    _fields_ = [
        ("export_below", C.c_uint32),
        ("bvh_refill", C.c_uint32),
        ("bvh_leaf", C.c_uint32),
        ("cont_workgroups", C.c_uint32),
        ("level0_form", C.c_uint32),
        ("regen_workgroups", C.c_uint32),
        ("in_order", C.c_uint32),
    ]


class PtStats(C.Structure):
    _fields_ = [
        ("samples", C.c_uint64),
        ("vertices", C.c_uint64),
        ("shadow_rays", C.c_uint64),
        ("bounce_launches", C.c_uint32),
        ("batches", C.c_uint32),
        ("max_depth_reached", C.c_uint32),
        ("reserved", C.c_uint32),
        ("bounce_kernel_ms", C.c_double),
        ("total_ms", C.c_double),
        ("primary_vertices", C.c_uint64),
        ("primary_kernel_ms", C.c_double),
        ("primary_launches", C.c_uint32),
        ("reserved2", C.c_uint32),
        ("samples_expected", C.c_uint64),
    ]


class PtSchedJob(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("n_batches", "regen", "split", "hand_off", "regen_export", "profile", "in_order", "capturing",
                                          "grid", "regen_grid", "cont_grid", "regen_capacity", "fixed_grid", "counter_words")] + [("xchg_need", C.c_uint64)]


class PtSchedOp(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("kind", "stream", "event", "pool", "set", "lane", "level", "own_queue", "ovf_par", "batch", "grid",
                                          "seq", "core", "flags", "zero_words", "reserved")] + [("xchg_off", C.c_uint64), ("xchg_len", C.c_uint64)]


class PtMultiInfo(C.Structure):
    _fields_ = [
        ("n_devices", C.c_uint32),
        ("comm_count", C.c_uint32),
        ("rccl_version", C.c_uint32),
        ("threaded", C.c_uint32),
        ("frames", C.c_uint64),
        ("enqueue_us_sum", C.c_double),
        ("enqueue_us_max", C.c_double),
        ("exchange", C.c_uint32),
        ("reserved_", C.c_uint32),
    ]


PT_EXCHANGE_RCCL, PT_EXCHANGE_COPY = 0, 1
PT_SHAPE_SPHERE, PT_SHAPE_TRIANGLE = 0, 1
PT_MAT_LAMBERT, PT_MAT_EMISSIVE, PT_MAT_MIRROR, PT_MAT_OREN_NAYAR = 0, 1, 2, 3
PT_INTEGRATOR_MIS, PT_INTEGRATOR_BRDF_ONLY = 0, 1
PT_STREAM_LEGACY_DEFAULT = 1      # pt_context_set_stream: HIP's legacy default stream (handle 0 means "the context's own")

# every symbol include/pathtrace_amd.h declares: name -> (restype, argtypes)
_P = C.POINTER
SYMBOLS = {
    "pt_camera_new": (C.c_int, [_P(C.c_double), C.c_uint32, C.c_uint32, C.c_double, C.c_double, _P(PtCamera)]),
    "pt_camera_look_at": (C.c_int, [_P(C.c_double), _P(C.c_double), _P(C.c_double), C.c_uint32, C.c_uint32,
                                    C.c_double, _P(PtCamera)]),
    "pt_default_params": (None, [_P(PtRenderParams)]),
    "pt_tile_rows": (C.c_uint32, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
    "pt_builtin_scene": (C.c_int, [C.c_uint32, C.c_uint32, _P(PtObject), C.c_uint32, _P(C.c_uint32)]),
    "pt_context_create": (C.c_int, [C.c_int, _P(C.c_void_p)]),
    "pt_context_destroy": (C.c_int, [C.c_void_p]),
    "pt_context_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "pt_context_set_tuning": (C.c_int, [C.c_void_p, _P(PtTuning)]),
    "pt_scene_upload": (C.c_int, [C.c_void_p, _P(PtObject), C.c_uint32]),
    "pt_render_device": (C.c_int, [C.c_void_p, _P(PtCamera), _P(PtRenderParams), C.c_void_p, C.c_void_p]),
    "pt_render_device_packed": (C.c_int, [C.c_void_p, _P(PtCamera), _P(PtRenderParams), C.c_void_p]),
    "pt_sync": (C.c_int, [C.c_void_p]),
    "pt_get_stats": (C.c_int, [C.c_void_p, _P(PtStats)]),
    "pt_debug_raw_stats": (C.c_int, [C.c_void_p, _P(C.c_uint64)]),
    "pt_debug_scan_layout": (C.c_int, [C.c_void_p, _P(C.c_uint32), _P(C.c_uint32), _P(C.c_uint32)]),
    "pt_render_host": (C.c_int, [C.c_void_p, _P(PtCamera), _P(PtRenderParams), C.c_void_p, C.c_void_p]),
    "pt_render_progressive": (C.c_int, [C.c_void_p, _P(PtCamera), _P(PtRenderParams), C.c_uint32, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p]),
    "pt_render": (C.c_int, [_P(PtCamera), _P(PtObject), C.c_uint32, _P(PtRenderParams), C.c_void_p, C.c_void_p]),
    "pt_debug_hit_scene": (C.c_int, [C.c_void_p, _P(C.c_double), C.c_uint32, C.c_double, C.c_double, C.c_uint32,
                                     C.c_uint32, _P(C.c_int32), _P(C.c_float)]),
    "pt_debug_hit_records": (C.c_int, [C.c_void_p, _P(C.c_double), C.c_uint32, C.c_double, C.c_double, C.c_uint32,
                                       C.c_uint32, _P(C.c_int32), _P(C.c_float)]),
    "pt_debug_bsdf_eval": (C.c_int, [C.c_void_p, C.c_uint32, _P(C.c_double), C.c_uint32, C.c_uint32, _P(C.c_float)]),
    "pt_debug_bsdf_sample": (C.c_int, [C.c_void_p, C.c_uint32, _P(C.c_double), _P(C.c_uint32), C.c_uint32, C.c_uint32,
                                       _P(C.c_float)]),
    "pt_debug_shape_sample": (C.c_int, [C.c_void_p, C.c_uint32, _P(C.c_double), _P(C.c_double), _P(C.c_double), C.c_uint32,
                                        C.c_uint32, _P(C.c_float)]),
    "pt_debug_light_point": (C.c_int, [C.c_void_p, _P(C.c_double), _P(C.c_uint32), C.c_uint32, C.c_uint32, _P(C.c_float)]),
    "pt_debug_camera_rays": (C.c_int, [C.c_void_p, _P(PtCamera), _P(C.c_uint32), C.c_uint32, C.c_uint32, _P(C.c_float)]),
    "pt_debug_bvh_check": (C.c_int, [_P(PtObject), C.c_uint32, _P(C.c_uint32), _P(C.c_uint32), _P(C.c_uint32)]),
    "pt_render_pixels": (C.c_int, [C.c_void_p, _P(PtCamera), _P(PtRenderParams), _P(C.c_uint32), C.c_uint32, C.c_void_p,
                                   C.c_void_p, C.c_void_p]),
    "pt_ray_color": (C.c_int, [C.c_void_p, _P(PtRenderParams), _P(C.c_double), _P(C.c_uint32), C.c_uint32, C.c_void_p]),
    "pt_shutdown": (None, []),
    "pt_multi_create": (C.c_int, [_P(C.c_int), C.c_uint32, _P(C.c_void_p)]),
    "pt_multi_destroy": (C.c_int, [C.c_void_p]),
    "pt_multi_device_count": (C.c_uint32, [C.c_void_p]),
    "pt_multi_set_threads": (C.c_int, [C.c_void_p, C.c_int]),
    "pt_multi_set_exchange": (C.c_int, [C.c_void_p, C.c_uint32]),
    "pt_multi_info": (C.c_int, [C.c_void_p, _P(PtMultiInfo)]),
    "pt_debug_multi_create_shared": (C.c_int, [C.c_int, C.c_uint32, _P(C.c_void_p)]),
    "pt_debug_feeder_selftest": (C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_int32, _P(C.c_uint64), _P(C.c_uint32)]),
    "pt_debug_sched_create": (C.c_int, [_P(C.c_void_p)]),
    "pt_debug_sched_destroy": (None, [C.c_void_p]),
    "pt_debug_sched_render": (C.c_int, [C.c_void_p, _P(PtSchedJob), C.c_uint32, C.c_uint32, _P(PtSchedOp), C.c_uint32, _P(C.c_uint32), _P(C.c_uint32)]),
    "pt_debug_sched_sync": (C.c_int, [C.c_void_p, C.c_uint32]),
    "pt_debug_fail_after": (C.c_int, [C.c_void_p, C.c_int64]),
    "pt_multi_scene_upload": (C.c_int, [C.c_void_p, _P(PtObject), C.c_uint32]),
    "pt_multi_set_tuning": (C.c_int, [C.c_void_p, _P(PtTuning)]),
    "pt_multi_render_device": (C.c_int, [C.c_void_p, _P(PtCamera), _P(PtRenderParams), C.c_void_p, C.c_void_p]),
    "pt_multi_sync": (C.c_int, [C.c_void_p]),
    "pt_multi_get_stats": (C.c_int, [C.c_void_p, _P(PtStats)]),
    "pt_multi_render_host": (C.c_int, [C.c_void_p, _P(PtCamera), _P(PtRenderParams), C.c_void_p, C.c_void_p]),
    "pt_film_pack": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]),
    "pt_film_unpack": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]),
    "pt_debug_multi_emulate": (C.c_int, [C.c_void_p, C.c_uint32, _P(PtCamera), _P(PtRenderParams), C.c_void_p, C.c_void_p]),
    "pt_render_multi": (C.c_int, [_P(C.c_int), C.c_uint32, _P(PtCamera), _P(PtObject), C.c_uint32, _P(PtRenderParams),
                                  C.c_void_p, C.c_void_p]),
    "pt_last_error": (C.c_char_p, []),
    "pt_abi_version": (C.c_uint32, []),
}

PROGRESS_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint8), C.POINTER(C.c_float))

_lib = None


def lib():
    """Load libpathtrace_amd.so (built by `__graft_entry__.build()` / csrc/Makefile)."""
    global _lib
    if _lib is None:
        # torch ships its own libamdhip64.so with the same SONAME as /opt/rocm's.  A process must
        # run ONE HIP runtime: load torch's first so that this library binds to it too (device
        # pointers and streams are exchanged with torch).  Without torch the system runtime is used.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build the HIP library first "
                "(python -c 'import __graft_entry__ as g; g.build()').  There is no fallback path.")
        h = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(h, name)   # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = h
    return _lib


class PtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"pathtrace_amd error {code}: {msg}")
        self.code = code


def check(code):
    if code != 0:
        raise PtError(code, lib().pt_last_error().decode("utf-8", "replace"))
