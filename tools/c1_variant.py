import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pathtrace_amd as pt
def run(name, objs):
    cam = pt.camera_new(width=1024, height=1024); prm = pt.default_params(spp=64, profile=1)
    ctx = pt.Context(0); ctx.upload(objs)
    for _ in range(3): ctx.render(cam, prm)
    st = ctx.stats(); print(f"{name}: kernel {st.bounce_kernel_ms:.2f} ms  V/S {st.vertices/st.samples:.2f}", flush=True); ctx.close()
objs = pt.builtin_scene(1); run("C1 as is (glass)", objs)
o2 = list(objs); lam = pt.make_objects([(0, list(objs[12].shape[:4]), 0, [0.8, 0.8, 0.8])])[0]; o2[12] = lam
run("C1 glass->lambert", (pt._lib.PtObject * 13)(*o2))
o3 = list(objs[:12]); run("C1 without sphere", (pt._lib.PtObject * 12)(*o3))
