"""What the GGX code costs C1: the reference scene (World::new(), 12 triangles + the rough-glass sphere) against the same
geometry with the sphere Lambertian (kernel compiled without the GGX / OrenNayar code), ms per 1024^2 x 64 render and
per million vertices, queue form and regenerating form.
    python tools/r03/c1_ggx_cost.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import pathtrace_amd as pt

dev = torch.device("cuda", 0)
cam = pt.camera_new(width=1024, height=1024); prm = pt.default_params(spp=64)
lin = torch.empty((1024, 1024, 3), dtype=torch.float32, device=dev); rgba = torch.empty((1024, 1024, 4), dtype=torch.uint8, device=dev)
for label, lambert in (("C1 (glass sphere)", False), ("C1 with a Lambertian sphere", True)):
    objs = pt.builtin_scene(1)
    if lambert:
        o = objs[len(objs) - 1]
        o.mat_tag = 0                                   # PT_MAT_LAMBERT
        o.mat[0], o.mat[1], o.mat[2] = 0.8, 0.8, 0.8
    for form in (1, 2, 3):
        ctx = pt.Context(0); ctx.upload(objs); ctx.set_tuning(level0_form=form)
        st = torch.cuda.Stream(dev); ctx.set_stream(st.cuda_stream)
        with torch.cuda.stream(st):
            for _ in range(2): ctx.render_into(cam, prm, lin.data_ptr(), rgba.data_ptr())
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(8): ctx.render_into(cam, prm, lin.data_ptr(), rgba.data_ptr())
            e1.record(st); st.synchronize()
        ms = e0.elapsed_time(e1) / 8
        ctx.sync(); ctx.render_into(cam, prm, lin.data_ptr(), rgba.data_ptr()); ctx.sync(); s = ctx.stats()      # the statistics of ONE render
        print(f"{label}, level0_form {form}: {ms:.3f} ms per render, {s.vertices / 1e6:.1f} M vertices, "
              f"{ms * 1e3 / (s.vertices / 1e6):.2f} us per M vertices, {s.bounce_launches} path launches", flush=True)
        ctx.set_stream(None); ctx.close()
