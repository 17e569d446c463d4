import sys, os, time
sys.path.insert(0, "/root/repo")
import pathtrace_amd as pt
t=time.time(); ctx = pt.Context(0); print("context %.3f"%(time.time()-t))
t=time.time(); ctx.upload(pt.builtin_scene(1)); print("upload %.3f"%(time.time()-t))
cam = pt.camera_new(width=400, height=400)
for mp in (0, 1<<24, 1<<22):
    c2 = pt.Context(0); c2.upload(pt.builtin_scene(1))
    prm = pt.default_params(spp=3000, max_paths_in_flight=mp)
    t=time.time(); lin, rgba = c2.render(cam, prm); print("max_paths %d first render %.3f"%(mp, time.time()-t), c2.stats().batches)
    t=time.time(); lin, rgba = c2.render(cam, prm); print("   second render %.3f"%(time.time()-t))
    c2.close()
