"""Host time of every pt_multi_render_device call when frames are posted back to back (one device, real RCCL gather): does the call
return at once, or does something in it wait for the device?   python tools/r04/multi_host_times.py [frames [threads]]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import pathtrace_amd as pt

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 24
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dev = torch.device("cuda", 0)
objs = pt.builtin_scene(2)
for W, H in ((1024, 1024), (1024, 128)):
    cam = pt.camera_new(width=W, height=H); prm = pt.default_params(spp=64)
    lin = torch.empty((H, W, 3), dtype=torch.float32, device=dev); rgba = torch.empty((H, W, 4), dtype=torch.uint8, device=dev)
    m = pt.Multi([0]); m.upload(objs); m.set_threads(threads)
    for _ in range(4): m.render_into(cam, prm, lin.data_ptr(), rgba.data_ptr())
    m.sync()
    ts = []
    t00 = time.perf_counter()
    for _ in range(frames):
        t0 = time.perf_counter(); m.render_into(cam, prm, lin.data_ptr(), rgba.data_ptr()); ts.append((time.perf_counter() - t0) * 1e6)
    t_posted = (time.perf_counter() - t00) * 1e3
    m.sync(); t_all = (time.perf_counter() - t00) * 1e3
    m.close()
    print(f"{W}x{H} threads={threads}: all {frames} frames posted after {t_posted:.2f} ms, complete after {t_all:.2f} ms; per call us: " + " ".join(f"{t:.0f}" for t in ts), flush=True)
