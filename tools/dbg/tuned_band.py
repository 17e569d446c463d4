"""debug: a row-band tile with a tuned regenerating grid, in order vs default"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import pathtrace_amd as pt
ctx = pt.Context(0); ctx.upload(pt.builtin_scene(2))
cam = pt.camera_new(width=256, height=256)
for spp in (4, 8, 12):
  for kw in (dict(), dict(band_rows=8, band_index=1, band_count=3), dict(band_rows=16, band_index=0, band_count=2)):
    res = {}
    for tune in (0, 300, 1700):
        for in_order in (1, 0):
            ctx.set_tuning(in_order=in_order, regen_workgroups=tune)
            lin, rgba = ctx.render(cam, pt.default_params(spp=spp, **kw))
            st = ctx.stats()
            res[(tune, in_order)] = (lin.clone(), st.vertices, st.samples)
    base = res[(0, 1)]
    print(spp, kw, {k: (bool(torch.equal(v[0], base[0])), float(v[0].sum()), v[1], v[2]) for k, v in res.items()}, flush=True)
ctx.close()
