"""Frames posted back to back for a kernel trace: `plain` (one context, packed resolve) or `multi` (pt_multi over [0]: + RCCL gather +
row permutation).   rocprofv3 --kernel-trace --output-format csv -d DIR -o st -- python3 tools/r04/frames_trace.py plain|multi W H frames"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import pathtrace_amd as pt

mode, W, H, frames = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
dev = torch.device("cuda", 0)
objs = pt.builtin_scene(2)
cam = pt.camera_new(width=W, height=H); prm = pt.default_params(spp=64)
lin = torch.empty((H, W, 3), dtype=torch.float32, device=dev); rgba = torch.empty((H, W, 4), dtype=torch.uint8, device=dev)
packed = torch.empty((H, W, 16), dtype=torch.uint8, device=dev)
if mode == "plain":
    ctx = pt.Context(0); ctx.upload(objs)
    for _ in range(3): ctx.render_packed_into(cam, prm, packed.data_ptr())
    ctx.sync()
    for _ in range(frames): ctx.render_packed_into(cam, prm, packed.data_ptr())
    ctx.sync(); ctx.close()
else:
    m = pt.Multi([0]); m.upload(objs)
    for _ in range(3): m.render_into(cam, prm, lin.data_ptr(), rgba.data_ptr())
    m.sync()
    for _ in range(frames): m.render_into(cam, prm, lin.data_ptr(), rgba.data_ptr())
    m.sync(); m.close()
