"""One rank's share of C2 at 8 ranks, 30 renders back to back, for several grids of the regenerating launch
(PtTuning.regen_workgroups; default = what the device holds: 256 CUs x 6)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import pathtrace_amd as pt
scene = int(sys.argv[3]) if len(sys.argv) > 3 else 2
ctx = pt.Context(0); ctx.upload(pt.builtin_scene(scene))
cam = pt.camera_new(width=1024, height=1024)
dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
prm = pt.default_params(spp=64, band_rows=16 if n > 1 else 0, band_index=0, band_count=n)
rows = pt.tile_rows(1024, prm.band_rows, 0, n)
lin = torch.empty((rows, 1024, 3), dtype=torch.float32, device=dev); rgba = torch.empty((rows, 1024, 4), dtype=torch.uint8, device=dev)
for rep in range(1):
    for wg in [int(v) for v in (sys.argv[2].split(",") if len(sys.argv) > 2 else "0,1536,1408,1280,1152,1024,768".split(","))]:
        ctx.set_tuning(regen_workgroups=wg)
        for _ in range(3): ctx.render_into(cam, prm, lin.data_ptr(), rgba.data_ptr())
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(30): ctx.render_into(cam, prm, lin.data_ptr(), rgba.data_ptr())
        ctx.sync()
        print(f"scene {scene} N={n} regen_workgroups {wg:5d}: {(time.perf_counter() - t0) / 30 * 1e3:.4f} ms per render", flush=True)
ctx.close()
