"""One-off: accel=1 vs linear scan on the 100k-triangle terrain at 256x256x4 (both arithmetic modes), films and counters."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import pathtrace_amd as pt
sys.argv = [sys.argv[0], "224", "64", "1"]
exec(open(os.path.join(os.path.dirname(__file__), "..", "mesh_bench.py")).read().split("cam = pt.camera_look_at")[0])
cam = pt.camera_look_at((0.0, 1.0, 1.5), (0.0, -0.6, -3.0), (0.0, 1.0, 0.0), 256, 256, 40.0)
ctx = pt.Context(0); ctx.upload(objs)
for em in (1, 0):
    out = []
    for accel in (1, 0):
        t = time.time(); lin, rgba = ctx.render(cam, pt.default_params(spp=4, accel=accel, exact_math=em)); st = ctx.stats()
        out.append((lin.cpu().numpy(), rgba.cpu().numpy(), st.vertices, st.shadow_rays)); print(f"exact={em} accel={accel}: {time.time()-t:.2f} s", flush=True)
    print("  identical:", np.array_equal(out[0][0], out[1][0], equal_nan=True) and np.array_equal(out[0][1], out[1][1]) and out[0][2:] == out[1][2:])
