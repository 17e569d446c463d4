"""Measurement build (PT_DRAIN_TIMING): do the waves of launch k + 1 start while launch k runs dry?  `reps` renders of one rank's
share of C2 at N ranks back to back, then the per-wave stamps of the last two launches (one per lane)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import pathtrace_amd as pt
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
ctx = pt.Context(0); ctx.upload(pt.builtin_scene(2))
cam = pt.camera_new(width=1024, height=1024)
prm = pt.default_params(spp=64, band_rows=16 if n > 1 else 0, band_index=0, band_count=n)
rows = pt.tile_rows(1024, prm.band_rows, 0, n)
dev = torch.device("cuda", 0)
lin = torch.empty((rows, 1024, 3), dtype=torch.float32, device=dev); rgba = torch.empty((rows, 1024, 4), dtype=torch.uint8, device=dev)
ctx.render_into(cam, prm, lin.data_ptr(), rgba.data_ptr()); ctx.sync(); ctx.stats()
for _ in range(reps): ctx.render_into(cam, prm, lin.data_ptr(), rgba.data_ptr())
ctx.sync()
nw = 256 * 6 * 4
fn = pt._lib.lib().pt_debug_wave_dump
fn.restype = C.c_int; fn.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
d = []
for plane in (0, 1, 2):
    buf = np.zeros((nw, 4), dtype=np.uint32)
    pt._lib.check(fn(ctx._h, buf.ctypes.data_as(C.c_void_p), nw | (plane << 30)))
    d.append(buf.astype(np.int64))
d.sort(key=lambda k: np.median(k[:, 2]))
base = d[0][:, 0].min()
for name, k in (("launch k", d[0]), ("launch k + 1", d[1]), ("launch k + 2", d[2])):
    b, x, e = (k[:, 0] - base) * 0.01, (k[:, 1] - base) * 0.01, (k[:, 2] - base) * 0.01
    q = lambda a: " ".join(f"{np.percentile(a, p):8.1f}" for p in (0, 1, 10, 25, 50, 75, 90, 99, 100))
    print(f"{name}: wave begin   p0/1/10/25/50/75/90/99/100 = {q(b)} us")
    print(f"{name}: out of work  p0/1/10/25/50/75/90/99/100 = {q(x)} us")
    print(f"{name}: wave end     p0/1/10/25/50/75/90/99/100 = {q(e)} us")
ctx.close()
