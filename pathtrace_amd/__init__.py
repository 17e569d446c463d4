"""pathtrace_amd -- MI355X-native wavefront path tracer behind a C ABI.

Drop-in for the per-pixel rendering hot path of roxas1533/pathtrace
(src/rendering.rs and what it calls).  The product is libpathtrace_amd.so
(pathtrace_amd/csrc); this package is the ctypes harness tests and bench use.
"""
from . import _lib, api  # noqa: F401
from .api import (Context, Multi, builtin_scene, bvh_check, camera_look_at, camera_new, default_params, make_objects,  # noqa: F401
                  render_host, render_multi, tile_rows, tile_row_indices)
