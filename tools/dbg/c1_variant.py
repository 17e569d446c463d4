"""How much of C1's time is the GGX sphere?  C1 as is, C1 with the glass sphere made Lambertian, C1 without triangles' walls replaced (n/a)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pathtrace_amd as pt
def run(name, objs):
    ctx = pt.Context(0); ctx.upload(objs); cam = pt.camera_new(width=1024, height=1024)
    best = 1e9
    for _ in range(3):
        t = time.time(); ctx.render(cam, pt.default_params(spp=64, profile=1)); best = min(best, time.time() - t); st = ctx.stats()
    print(f"{name}: {best*1e3:.2f} ms  V/S {st.vertices/st.samples:.2f}  ms per Gvertex {best*1e3/(st.vertices/1e9):.1f}", flush=True)
    ctx.close()
objs = pt.builtin_scene(1)
print([ (o.shape_tag, o.mat_tag) for o in objs])
run("C1", objs)
o2 = pt.builtin_scene(1)
for o in o2:
    if o.mat_tag == 2:
        o.mat_tag = 0; o.mat[0] = o.mat[1] = o.mat[2] = 0.8
run("C1, glass -> Lambert (DIFFUSE kernel)", o2)
o3 = pt.builtin_scene(1)
for o in o3:
    if o.mat_tag == 2:
        o.mat_tag = 3; o.mat[0] = o.mat[1] = o.mat[2] = 0.8; o.mat[3] = 0.0     # OrenNayar sigma 0 == Lambert, generic kernel
run("C1, glass -> OrenNayar(0) (generic kernel)", o3)
