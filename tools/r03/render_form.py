"""Three 1024^2 x 64 renders of a built-in scene with a given PtTuning.level0_form (for rocprofv3 counter passes).
    python3 tools/r03/render_form.py <scene id> <level0_form>"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import pathtrace_amd as pt
scene, form = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda", 0)
cam = pt.camera_new(width=1024, height=1024); prm = pt.default_params(spp=64)
lin = torch.empty((1024, 1024, 3), dtype=torch.float32, device=dev); rgba = torch.empty((1024, 1024, 4), dtype=torch.uint8, device=dev)
ctx = pt.Context(0); ctx.upload(pt.builtin_scene(scene)); ctx.set_tuning(level0_form=form)
for _ in range(3): ctx.render_into(cam, prm, lin.data_ptr(), rgba.data_ptr())
ctx.sync(); s = ctx.stats()
print(f"scene {scene} form {form}: {s.total_ms:.3f} ms, {s.vertices} vertices, {s.bounce_launches} launches")
ctx.close()
