"""ms per 1024^2 x 64 render of the all-OrenNayar Cornell scene (C2 with OrenNayar walls and spheres), queue form vs the
regenerating kernel compiled without the GGX code: python tools/r03/on_forms.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import pathtrace_amd as pt
dev = torch.device("cuda", 0)
objs = list(pt.builtin_scene(2))
for k, o in enumerate(objs):
    if o.mat_tag == 0:
        o.mat_tag = 3; o.mat[3] = [0.0, 0.3, 0.6, 1.0][k % 4]
objs = (pt._lib.PtObject * len(objs))(*objs)
cam = pt.camera_new(width=1024, height=1024); prm = pt.default_params(spp=64)
lin = torch.empty((1024, 1024, 3), dtype=torch.float32, device=dev); rgba = torch.empty((1024, 1024, 4), dtype=torch.uint8, device=dev)
for form in (1, 2, 0):
    ctx = pt.Context(0); ctx.upload(objs); ctx.set_tuning(level0_form=form)
    st = torch.cuda.Stream(dev); ctx.set_stream(st.cuda_stream)
    with torch.cuda.stream(st):
        for _ in range(2): ctx.render_into(cam, prm, lin.data_ptr(), rgba.data_ptr())
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(8): ctx.render_into(cam, prm, lin.data_ptr(), rgba.data_ptr())
        e1.record(st); st.synchronize()
    ctx.sync(); ctx.render_into(cam, prm, lin.data_ptr(), rgba.data_ptr()); ctx.sync(); s = ctx.stats()      # the statistics of ONE render
    print(f"OrenNayar Cornell, level0_form {form}: {e0.elapsed_time(e1) / 8:.3f} ms per render, {s.bounce_launches} path launches", flush=True)
    ctx.set_stream(None); ctx.close()
