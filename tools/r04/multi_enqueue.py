"""What a frame of the single-process multi-device form costs the HOST, and what posting frames back to back buys: the C2 job
(1024 x 1024 x 64 spp) through a PtMulti of n contexts that all sit on device 0 (pt_debug_multi_create_shared: device-to-
device copies where the real object calls ncclGather), n = 1, 2, 4, 8, with one host thread per device and with the one-thread
form.  The GPU time per frame is the same job each time (one device does all the tiles); the figures of interest are the
host's: enqueue time per device and frame (pt_multi_info), and whether the host keeps ahead of the device.
    python tools/r04/multi_enqueue.py [frames]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import pathtrace_amd as pt

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 20
objs = pt.builtin_scene(2)
cam = pt.camera_new(width=1024, height=1024)
dev = torch.device("cuda", 0)
lin = torch.empty((1024, 1024, 3), dtype=torch.float32, device=dev); rgba = torch.empty((1024, 1024, 4), dtype=torch.uint8, device=dev)
ref = None
for n in (1, 2, 4, 8):
    for threaded in (True, False):
        m = pt.Multi([0] * n, shared_device=0)
        m.upload(objs)
        m.set_threads(threaded)
        prm = pt.default_params(spp=64)
        for _ in range(3): m.render_into(cam, prm, lin.data_ptr(), rgba.data_ptr())
        m.sync()
        i0 = m.info()
        t0 = time.perf_counter()
        for _ in range(frames): m.render_into(cam, prm, lin.data_ptr(), rgba.data_ptr())
        t_post = time.perf_counter() - t0
        m.sync()
        t_all = time.perf_counter() - t0
        i1 = m.info()
        f = i1.frames - i0.frames
        us_sum = (i1.enqueue_us_sum * i1.frames - i0.enqueue_us_sum * i0.frames) / f
        us_max = (i1.enqueue_us_max * i1.frames - i0.enqueue_us_max * i0.frames) / f
        if ref is None: ref = lin.clone()
        same = bool(torch.equal(lin, ref))
        print(f"n={n} {'threads' if threaded else 'one thread'}: {t_all / frames * 1e3:.3f} ms per frame (posting took {t_post / frames * 1e6:.0f} us per frame); "
              f"enqueue per frame: all devices {us_sum:.0f} us, slowest device {us_max:.0f} us; frame == n=1 frame: {same}", flush=True)
        m.close()
