"""Do the lanes engage whatever stream the caller renders on?  One rank's share of C2 at 8 ranks, 30 renders back to back, on the
context's own stream, on a torch stream, and on HIP's legacy default stream (torch's default stream: what bench.py's
one-process-per-GPU form renders on)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import pathtrace_amd as pt
ctx = pt.Context(0); ctx.upload(pt.builtin_scene(2))
cam = pt.camera_new(width=1024, height=1024)
dev = torch.device("cuda", 0)
prm = pt.default_params(spp=64, band_rows=16, band_index=0, band_count=8)
rows = pt.tile_rows(1024, 16, 0, 8)
lin = torch.empty((rows, 1024, 3), dtype=torch.float32, device=dev); rgba = torch.empty((rows, 1024, 4), dtype=torch.uint8, device=dev)
ts = torch.cuda.Stream(dev)
for name, handle in (("context's own stream", None), ("torch stream", ts.cuda_stream), ("legacy default stream", 0)):
    ctx.set_stream(handle)
    for _ in range(3): ctx.render_into(cam, prm, lin.data_ptr(), rgba.data_ptr())
    ctx.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30): ctx.render_into(cam, prm, lin.data_ptr(), rgba.data_ptr())
    ctx.sync(); torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / 30 * 1e3:.4f} ms per render (wall)", flush=True)
ctx.set_stream(None); ctx.close()
