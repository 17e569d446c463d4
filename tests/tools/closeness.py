"""How close is a library build to the f32 oracle?  PATHTRACE_AMD_LIB selects the build."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import pathtrace_amd as pt
from oracle import orc

def cmp(name, objs, W, H, spp, **kw):
    cam = pt.camera_new(width=W, height=H); prm = pt.default_params(spp=spp, **kw)
    ctx = pt.Context(0); ctx.upload(objs)
    lin, rgba = ctx.render(cam, prm); st = ctx.stats(); got = lin.cpu().numpy().astype(np.float64)
    r32, r32_8, c32 = orc.render(cam, objs, prm, orc.F32, orc.ITERATIVE, 16)
    r64, r64_8, _ = orc.render(cam, objs, prm, orc.F64, orc.RECURSIVE, 16)
    rel32 = np.abs(got - r32) / np.maximum(np.abs(r32), 1e-6)
    ok64 = (np.abs(got - r64) <= 1e-3 + 1e-2 * np.abs(r64)).all(-1).mean()
    ok64_o = (np.abs(r32 - r64) <= 1e-3 + 1e-2 * np.abs(r64)).all(-1).mean()
    print(f"{name}: V gpu={st.vertices} orc32={c32['vertices']} | vs f32 oracle: exact px {(got.astype(np.float32)==r32.astype(np.float32)).all(-1).mean():.4f} "
          f"px within 1e-5 rel {(rel32.max(-1)<=1e-5).mean():.5f} within 1e-3 {(rel32.max(-1)<=1e-3).mean():.5f} rgba equal {(rgba.cpu().numpy()==r32_8).all(-1).mean():.5f} "
          f"| vs f64: tol_ok gpu {ok64:.5f} (f32 oracle {ok64_o:.5f}) mean rel gpu {abs(got.mean()-r64.mean())/r64.mean():.2e} (f32 oracle {abs(r32.mean()-r64.mean())/r64.mean():.2e})", flush=True)
    ctx.close()

cmp("C1 256x256x4", pt.builtin_scene(1), 256, 256, 4)
cmp("C2 256x256x16", pt.builtin_scene(2), 256, 256, 16)
cmp("C4 10k 48x48x4", pt.builtin_scene(4, 10000), 48, 48, 4)
cmp("C1 brdf 128x128x8", pt.builtin_scene(1), 128, 128, 8, integrator=1)
