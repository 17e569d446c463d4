"""ms per 1024^2 x 64 render of C1 (reference scene) for level-0 forms: python tools/r03/c1_forms.py <form>[:<regen_workgroups>] ..."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import pathtrace_amd as pt
dev = torch.device("cuda", 0)
scene = int(os.environ.get("PT_SCENE", "1"))
cam = pt.camera_new(width=1024, height=1024); prm = pt.default_params(spp=64)
lin = torch.empty((1024, 1024, 3), dtype=torch.float32, device=dev); rgba = torch.empty((1024, 1024, 4), dtype=torch.uint8, device=dev)
for spec in sys.argv[1:]:
    form, _, wg = spec.partition(":")
    ctx = pt.Context(0); ctx.upload(pt.builtin_scene(scene)); ctx.set_tuning(level0_form=int(form), regen_workgroups=int(wg or 0))
    st = torch.cuda.Stream(dev); ctx.set_stream(st.cuda_stream)
    with torch.cuda.stream(st):
        for _ in range(2): ctx.render_into(cam, prm, lin.data_ptr(), rgba.data_ptr())
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(8): ctx.render_into(cam, prm, lin.data_ptr(), rgba.data_ptr())
        e1.record(st); st.synchronize()
    ms = e0.elapsed_time(e1) / 8
    ctx.sync(); ctx.render_into(cam, prm, lin.data_ptr(), rgba.data_ptr()); ctx.sync(); s = ctx.stats()      # the statistics of ONE render
    print(f"{os.environ.get('PATHTRACE_AMD_LIB', 'default lib')}: scene {scene} level0_form {spec}: {ms:.3f} ms per render, {s.bounce_launches} path launches", flush=True)
    ctx.set_stream(None); ctx.close()
