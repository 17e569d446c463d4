"""ms per device-resident render of small jobs (progressive-preview steps, tiles of a multi-GPU job, pixel-list sized batches):
    python tools/small_batches.py [level0_form [job index [scene id: 2 = C2 (default), 1 = the reference's World::new()]]]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pathtrace_amd as pt

form = int(sys.argv[1]) if len(sys.argv) > 1 else 0
scene = int(sys.argv[3]) if len(sys.argv) > 3 else 2
ctx = pt.Context(0); ctx.upload(pt.builtin_scene(scene)); ctx.set_tuning(level0_form=form)
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(dev); ctx.set_stream(st.cuda_stream)
only = int(sys.argv[2]) if len(sys.argv) > 2 else -1      # index of the one job to run (for a kernel trace)
for k, (w, h, spp) in enumerate([(256, 256, 4), (512, 512, 2), (512, 512, 8), (1024, 1024, 1), (1024, 1024, 2), (1024, 1024, 4), (1024, 1024, 8), (1024, 1024, 16)]):
    if only >= 0 and k != only: continue
    cam = pt.camera_new(width=w, height=h); prm = pt.default_params(spp=spp)
    lin = torch.empty((h, w, 3), dtype=torch.float32, device=dev); rgba = torch.empty((h, w, 4), dtype=torch.uint8, device=dev)
    with torch.cuda.stream(st):
        for _ in range(3): ctx.render_into(cam, prm, lin.data_ptr(), rgba.data_ptr())
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(20): ctx.render_into(cam, prm, lin.data_ptr(), rgba.data_ptr())
        e1.record(st); st.synchronize()
    ms = e0.elapsed_time(e1) / 20
    ctx.sync(); ctx.render_into(cam, prm, lin.data_ptr(), rgba.data_ptr()); ctx.sync(); s = ctx.stats()      # the statistics of ONE render
    print(f"scene {scene} form {form}: {w}x{h}x{spp} = {w*h*spp/1e6:.2f} M paths: {ms:.3f} ms  ({w*h*spp/ms/1e3:.0f} Msamples/s, {s.bounce_launches} path launches)", flush=True)
ctx.set_stream(None); ctx.close()
