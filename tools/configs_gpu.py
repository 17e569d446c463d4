"""Throughput of the BASELINE.json configs on one GPU (default fast arithmetic)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pathtrace_amd as pt

def run(name, objs, W, H, spp, reps=2, **kw):
    cam = pt.camera_new(width=W, height=H); prm = pt.default_params(spp=spp, profile=1, **kw)
    ctx = pt.Context(0); ctx.upload(objs)
    ctx.set_tuning(bvh_refill=int(os.environ.get("TUNE_BVH_REFILL", "0")), bvh_leaf=int(os.environ.get("TUNE_BVH_LEAF", "0")),
                   export_below=int(os.environ.get("TUNE_EXPORT_BELOW", "0")))     # read by this script, not by the library
    best = None
    for _ in range(reps):
        t = time.time(); lin, rgba = ctx.render(cam, prm); dt = time.time() - t; st = ctx.stats()
        best = dt if best is None else min(best, dt)
    print(f"{name}: {W}x{H}x{spp} {st.samples/1e6:.0f} Msamples in {best*1e3:.1f} ms = {st.samples/best/1e6:.1f} Msamples/s  "
          f"V/S {st.vertices/st.samples:.2f} shadow/S {st.shadow_rays/st.samples:.2f} maxdepth {st.max_depth_reached} "
          f"batches {st.batches} kernel_ms {st.bounce_kernel_ms:.1f}", flush=True)
    ctx.close()

which = sys.argv[1:] or ["c1", "c2", "c3", "c4s", "c5"]
if "c1" in which: run("C1 ref Cornell+glass", pt.builtin_scene(1), 1024, 1024, 64)
if "c1s" in which: run("C1 config0", pt.builtin_scene(1), 256, 256, 4)
if "c2" in which: run("C2", pt.builtin_scene(2), 1024, 1024, 64)
if "c3" in which: run("C3", pt.builtin_scene(2), 1024, 1024, 4096, reps=2)
if "c4s" in which: run("C4 (4 spp probe)", pt.builtin_scene(4, 10000), 1024, 1024, 4, reps=1, accel=0)
if "c4" in which: run("C4 (linear scan)", pt.builtin_scene(4, 10000), 1024, 1024, 256, reps=1, accel=0)
if "c5" in which: run("C5 on one GPU", pt.builtin_scene(2), 3840, 2160, 64, reps=3)
if "c4m" in which: run("C4 (64 spp, linear scan)", pt.builtin_scene(4, 10000), 1024, 1024, 64, reps=1, accel=0)
if "c4b" in which: run("C4 BVH (64 spp)", pt.builtin_scene(4, 10000), 1024, 1024, 64, reps=2, accel=1)
if "c4bf" in which: run("C4 BVH", pt.builtin_scene(4, 10000), 1024, 1024, 256, reps=2, accel=1)
if "c1b" in which: run("C1 BVH", pt.builtin_scene(1), 1024, 1024, 64, accel=1)
if "c2b" in which: run("C2 BVH", pt.builtin_scene(2), 1024, 1024, 64, accel=1)
if "c6b" in which: run("100k spheres BVH", pt.builtin_scene(4, 100000), 1024, 1024, 64, accel=1)
if "c5f" in which: run("C5 full (1024 spp) on one GPU", pt.builtin_scene(2), 3840, 2160, 1024, reps=1)
