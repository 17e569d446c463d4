"""The single-process multi-device form on ONE device with the real RCCL gather kernel (pt_multi_create over [0]) against a plain
context, frames posted back to back: ms per frame for the whole C2 image and for a frame of one rank's share at 8 ranks
(1024 x 128 pixels).  What it shows: the cost of the gather's kernel arriving beside three overlapping regenerating launches
(pt_multi.cpp head) at realistic launch sizes; with `copy` the exchange by copies (pt_multi_set_exchange) instead.
    python tools/r04/share_multi.py [frames [rccl|copy]]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import pathtrace_amd as pt

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 48
exchange = sys.argv[2] if len(sys.argv) > 2 else "rccl"
dev = torch.device("cuda", 0)
objs = pt.builtin_scene(2)
for W, H in ((1024, 1024), (1024, 128)):
    cam = pt.camera_new(width=W, height=H)
    prm = pt.default_params(spp=64)
    lin = torch.empty((H, W, 3), dtype=torch.float32, device=dev); rgba = torch.empty((H, W, 4), dtype=torch.uint8, device=dev)
    packed = torch.empty((H, W, 16), dtype=torch.uint8, device=dev)
    res = {}
    ctx = pt.Context(0); ctx.upload(objs)
    st = torch.cuda.Stream(dev); ctx.set_stream(st.cuda_stream)
    for rep in range(2):
        for _ in range(4): ctx.render_packed_into(cam, prm, packed.data_ptr())
        ctx.sync(); t0 = time.perf_counter()
        for _ in range(frames): ctx.render_packed_into(cam, prm, packed.data_ptr())
        ctx.sync(); res["plain"] = (time.perf_counter() - t0) / frames * 1e3
    ctx.set_stream(None); ctx.close()
    m = pt.Multi([0]); m.upload(objs); m.set_exchange(exchange)
    for rep in range(2):
        for _ in range(4): m.render_into(cam, prm, lin.data_ptr(), rgba.data_ptr())
        m.sync(); t0 = time.perf_counter()
        for _ in range(frames): m.render_into(cam, prm, lin.data_ptr(), rgba.data_ptr())
        m.sync(); res["multi"] = (time.perf_counter() - t0) / frames * 1e3
    m.close()
    print(f"{W}x{H}: plain {res['plain']:.3f} ms per frame, single-process form, exchange = {exchange}: {res['multi']:.3f} (+{res['multi'] - res['plain']:.3f})", flush=True)
