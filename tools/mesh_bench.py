"""accel = 1 on a triangle mesh: a bumpy terrain of 2*n*n triangles under one sphere light (+ a few glossy spheres)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pathtrace_amd as pt

def terrain(n, seed=1):
    rng = np.random.default_rng(seed)
    xs = np.linspace(-2.0, 2.0, n + 1); zs = np.linspace(-5.0, -1.0, n + 1)
    X, Z = np.meshgrid(xs, zs, indexing="ij")
    Y = -0.8 + 0.15 * np.sin(3 * X) * np.cos(2.5 * Z) + 0.02 * rng.standard_normal(X.shape)
    P = np.stack([X, Y, Z], -1)
    specs = []
    for i in range(n):
        for j in range(n):
            a, b, c, d = P[i, j], P[i + 1, j], P[i + 1, j + 1], P[i, j + 1]
            col = [0.75, 0.75, 0.75] if (i // 8 + j // 8) % 2 else [0.35, 0.55, 0.35]
            specs.append((1, list(a) + list(c) + list(b), 0, col))
            specs.append((1, list(a) + list(d) + list(c), 0, col))
    specs.append((0, [0.0, 1.6, -3.0, 0.5], 1, [12.0, 12.0, 12.0]))
    specs.append((0, [-0.7, -0.35, -2.6, 0.3], 2, [0.1, 0.95, 0.95, 0.95, 1.0, 1.5]))
    specs.append((0, [0.8, -0.3, -3.2, 0.35], 2, [0.05, 1.0, 1.0, 1.0, 0.0, 1.5]))
    return pt.make_objects(specs)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 224
W = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 16
t = time.time(); objs = terrain(n); print(f"{len(objs)} objects built in {time.time()-t:.1f} s", flush=True)
t = time.time(); print("bvh_check", pt.bvh_check(objs), f"{time.time()-t:.2f} s", flush=True)
cam = pt.camera_look_at((0.0, 1.0, 1.5), (0.0, -0.6, -3.0), (0.0, 1.0, 0.0), W, W, 40.0)
ctx = pt.Context(0); ctx.upload(objs)
for rep in range(2):
    t = time.time(); lin, rgba = ctx.render(cam, pt.default_params(spp=spp, accel=1, profile=1)); dt = time.time() - t; st = ctx.stats()
    print(f"accel=1: {st.samples/1e6:.0f} Msamples in {dt*1e3:.1f} ms = {st.samples/dt/1e6:.1f} Msamples/s  V/S {st.vertices/st.samples:.2f} "
          f"shadow/S {st.shadow_rays/st.samples:.2f} kernel_ms {st.bounce_kernel_ms:.1f} mean {float(lin.mean()):.4f}", flush=True)
if "--check" in sys.argv:
    camc = pt.camera_look_at((0.0, 1.0, 1.5), (0.0, -0.6, -3.0), (0.0, 1.0, 0.0), 96, 96, 40.0)
    a, _ = ctx.render(camc, pt.default_params(spp=2, accel=1)); b, _ = ctx.render(camc, pt.default_params(spp=2, accel=0))
    print("bit-identical to the linear scan (96x96x2):", bool((a == b).all()))
