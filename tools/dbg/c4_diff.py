"""Where do the accel=1 and accel=0 films of full-size C4 differ?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import pathtrace_amd as pt
ctx = pt.Context(0); ctx.upload(pt.builtin_scene(4, 10000))
cam = pt.camera_new(width=1024, height=1024)
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 256
f = {}
for accel in (1, 0):
    lin, _ = ctx.render(cam, pt.default_params(spp=spp, accel=accel)); f[accel] = lin.cpu().numpy(); st = ctx.stats()
    print(accel, st.vertices, st.shadow_rays)
d = np.any(f[0] != f[1], axis=-1)
ys, xs = np.nonzero(d)
print("differing pixels:", d.sum())
for y, x in list(zip(ys, xs))[:10]:
    print(y, x, f[0][y, x], f[1][y, x], (f[1][y, x] - f[0][y, x]) * spp)
# isolate the sample for the first few pixels: render a 1-row band with 1 spp at a time
for y, x in list(zip(ys, xs))[:3]:
    for s in range(spp):
        r = {}
        for accel in (1, 0):
            prm = pt.default_params(spp=1, spp_offset=s, accel=accel, band_rows=1, band_index=int(y), band_count=1024)
            lin, _ = ctx.render(cam, prm); r[accel] = lin.cpu().numpy()[0, x]
        if not np.array_equal(r[0], r[1]):
            print("pixel", y, x, "sample", s, "linear", r[0], "bvh", r[1])
