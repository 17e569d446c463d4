"""A scene of 120 spheres (LDS-resident, 15 KB blob): queue form vs the regenerating default, ms per 1024^2 x 16 render and
bit equality of the films.  The regenerating launch sizes its grid by the kernel's occupancy WITH that blob."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import pathtrace_amd as pt
dev = torch.device("cuda", 0)
objs = pt.builtin_scene(4, int(sys.argv[1]) if len(sys.argv) > 1 else 120)
cam = pt.camera_new(width=1024, height=1024); prm = pt.default_params(spp=16)
lin = torch.empty((1024, 1024, 3), dtype=torch.float32, device=dev); rgba = torch.empty((1024, 1024, 4), dtype=torch.uint8, device=dev)
ref = None
for form, wg in ((1, 0), (0, 0), (2, 1536), (2, 1280), (2, 1024), (2, 768)):
    ctx = pt.Context(0); ctx.upload(objs); ctx.set_tuning(level0_form=form, regen_workgroups=wg)
    st = torch.cuda.Stream(dev); ctx.set_stream(st.cuda_stream)
    with torch.cuda.stream(st):
        for _ in range(2): ctx.render_into(cam, prm, lin.data_ptr(), rgba.data_ptr())
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(5): ctx.render_into(cam, prm, lin.data_ptr(), rgba.data_ptr())
        e1.record(st); st.synchronize()
    ctx.sync(); ctx.render_into(cam, prm, lin.data_ptr(), rgba.data_ptr()); ctx.sync(); s = ctx.stats()      # the statistics of ONE render
    same = "" if ref is None else f", film == queue form: {bool(torch.equal(lin, ref))}"
    if ref is None: ref = lin.clone()
    print(f"{len(objs)} spheres, level0_form {form}, regen_workgroups {wg or 'default'}: {e0.elapsed_time(e1) / 5:.3f} ms per render, {s.bounce_launches} path launches{same}", flush=True)
    ctx.set_stream(None); ctx.close()
