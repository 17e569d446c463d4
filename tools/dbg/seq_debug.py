"""debug: the random pipelined sequence of tests/test_gpu_functions.py, printing which jobs differ and how"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import pathtrace_amd as pt
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rng = np.random.default_rng(seed)
dev = torch.device("cuda", 0)
def oren_nayar_cornell():
    objs = list(pt.builtin_scene(2))
    for k, o in enumerate(objs):
        if o.mat_tag == 0:
            o.mat_tag = 3
            o.mat[3] = [0.0, 0.3, 0.6, 1.0][k % 4]
    return (pt._lib.PtObject * len(objs))(*objs)
ctxs = [pt.Context(0), pt.Context(0), pt.Context(0)]
ctxs[0].upload(pt.builtin_scene(2)); ctxs[1].upload(pt.builtin_scene(1)); ctxs[2].upload(oren_nayar_cornell())
jobs = []
for _ in range(28):
    which = int(rng.integers(0, 3))
    W, H = [(256, 256), (320, 200), (512, 128), (64, 48)][int(rng.integers(0, 4))]
    spp = int(rng.choice([4, 8, 12]))
    bands = int(rng.choice([1, 1, 2, 3]))
    kw = dict(spp=spp, exact_math=int(rng.integers(0, 2)), spp_offset=int(rng.integers(0, 1000)), integrator=int(rng.random() < 0.25))
    if bands > 1:
        kw.update(band_rows=int(rng.choice([8, 16, 50])), band_index=int(rng.integers(0, bands)), band_count=bands)
    if rng.random() < 0.4:
        kw["max_paths_in_flight"] = int(W * H * spp // int(rng.choice([2, 3, 4])) + 1)
    tune = int(rng.choice([0, 0, 0, 300, 1700]))
    jobs.append((which, (W, H), kw, tune))
refs = []
for which, (W, H), kw, tune in jobs:
    ctxs[which].set_tuning(in_order=1, regen_workgroups=tune)
    lin, rgba = ctxs[which].render(pt.camera_new(width=W, height=H), pt.default_params(**kw))
    refs.append((lin.clone(), rgba.clone(), ctxs[which].stats().vertices))
streams = [torch.cuda.Stream(dev) for _ in ctxs]
for c, st in zip(ctxs, streams): c.set_stream(st.cuda_stream)
outs = [(torch.zeros_like(r[0]), torch.zeros_like(r[1])) for r in refs]
for c in ctxs: c.sync(); c.stats()
for (which, (W, H), kw, tune), (lin_d, rgba_d) in zip(jobs, outs):
    ctxs[which].set_tuning(regen_workgroups=tune)
    ctxs[which].render_into(pt.camera_new(width=W, height=H), pt.default_params(**kw), lin_d.data_ptr(), rgba_d.data_ptr())
for c in ctxs: c.sync()
for k, ((lin_d, rgba_d), (lin, rgba, _)) in enumerate(zip(outs, refs)):
    same = torch.equal(lin_d.view(torch.int32), lin.view(torch.int32))
    nd = int((lin_d.view(torch.int32) != lin.view(torch.int32)).any(-1).sum())
    print(k, "ok " if same else "BAD", jobs[k], "pixels differing", nd, "of", lin.shape[0] * lin.shape[1], "sum out", float(lin_d.sum()), "sum ref", float(lin.sum()), flush=True)
