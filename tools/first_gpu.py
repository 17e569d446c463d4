"""Diagnostic: GPU vs float oracle on small cases + a first throughput number."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import pathtrace_amd as pt
from oracle import orc

def cmp(name, objs, W, H, spp, **kw):
    cam = pt.camera_new(width=W, height=H)
    prm = pt.default_params(spp=spp, **kw)
    ctx = pt.Context(0); ctx.upload(objs)
    lin, rgba = ctx.render(cam, prm); st = ctx.stats()
    got = lin.cpu().numpy()
    ref, ref_rgba, cnt = orc.render(cam, objs, prm, orc.F32, orc.ITERATIVE, threads=16)
    ref32 = ref.astype(np.float32)
    exact = (got == ref32).all(axis=-1).mean()
    err = np.abs(got.astype(np.float64) - ref)
    ok = (err <= 1e-3 + 1e-2*np.abs(ref)).all(axis=-1).mean()
    rg = (np.abs(rgba.cpu().numpy().astype(int) - ref_rgba.astype(int)) <= 1).all(axis=-1).mean()
    print(f"{name}: V gpu={st.vertices} orc={cnt['vertices']} shadow gpu={st.shadow_rays} orc={cnt['shadow_rays']} "
          f"maxdepth gpu={st.max_depth_reached} orc={cnt['max_depth']} exact={exact:.5f} tol_ok={ok:.5f} rgba<=1={rg:.5f} "
          f"maxerr={err.max():.3e} mean gpu={got.mean():.6f} ref={ref.mean():.6f} nan={np.isnan(got).sum()}", flush=True)
    ctx.close()

cmp("C1 64x64x8", pt.builtin_scene(1), 64, 64, 8)
cmp("C2 128x128x8", pt.builtin_scene(2), 128, 128, 8)
cmp("C4(2000) 64x64x4", pt.builtin_scene(4, 2000), 64, 64, 4)
cmp("C1 brdf-only 64x64x8", pt.builtin_scene(1), 64, 64, 8, integrator=1)
cmp("C2 batched 64x64x8", pt.builtin_scene(2), 64, 64, 8, max_paths_in_flight=64*64*3)

# throughput, C2 1024^2 x 64
objs = pt.builtin_scene(2)
cam = pt.camera_new(width=1024, height=1024)
ctx = pt.Context(0); ctx.upload(objs)
for cap, wg in ((1<<26, 0), (1<<26, 512), (1<<26, 2048), (1<<26, 4096), (1<<24, 0), (1<<22, 0)):
    prm = pt.default_params(spp=64, max_paths_in_flight=cap, profile=1, workgroups=wg)
    for it in range(3):
        t = time.time(); lin, rgba = ctx.render(cam, prm); dt = time.time() - t
        st = ctx.stats()
    print(f"C2 1024^2x64 cap={cap} wg={wg}: wall {dt*1e3:.1f} ms  total_ms {st.total_ms:.1f} bounce_ms {st.bounce_kernel_ms:.1f} "
          f"launches {st.bounce_launches} batches {st.batches} V={st.vertices} V/S={st.vertices/st.samples:.2f} "
          f"Msamples/s={st.samples/dt/1e6:.0f}  alg GB/s={(252*st.vertices+64*st.samples)/(st.bounce_kernel_ms*1e-3)/1e9:.0f}", flush=True)
