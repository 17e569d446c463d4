"""pt_render_host (host buffers, PCIe-inclusive) vs pt_render_device for C2 1024^2 x 64."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ctypes as C
import numpy as np
import pathtrace_amd as pt
from pathtrace_amd._lib import lib, check
objs = pt.builtin_scene(2); cam = pt.camera_new(width=1024, height=1024); prm = pt.default_params(spp=64)
ctx = pt.Context(0); ctx.upload(objs)
lin = np.empty((1024, 1024, 3), np.float32); rgba = np.empty((1024, 1024, 4), np.uint8)
for name, fn in [("pt_render_host", lambda: check(lib().pt_render_host(ctx._h, C.byref(cam), C.byref(prm), lin.ctypes.data_as(C.c_void_p), rgba.ctypes.data_as(C.c_void_p)))),
                 ("pt_render_device + sync", lambda: ctx.render(cam, prm))]:
    fn(); ts = []
    for _ in range(8):
        t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
    print(f"{name}: best {min(ts)*1e3:.2f} ms  median {sorted(ts)[4]*1e3:.2f} ms  -> {67.108864/min(ts)/1e3*1e3:.0f} Msamples/s")
