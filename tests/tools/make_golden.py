"""Generate tests/golden/* from the f64 RECURSIVE oracle (the reference-faithful
restatement, oracle/pt_oracle.hpp).  The reference tree holds no golden vector for
this path, so these fixtures pin the oracle against drift and give the
GPU tests a committed f64 target; they are data only (inputs + expected outputs).

    python tests/tools/make_golden.py          # rewrites tests/golden/
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import pathtrace_amd as pt
from oracle import orc

OUT = os.path.join(ROOT, "tests", "golden")
SCENES = {"c1": (1, 0), "c2": (2, 0), "c4_300": (4, 300)}
IMAGES = {"c1": (32, 32, 16), "c2": (32, 32, 16), "c4_300": (32, 32, 8)}


def write_luminance_csv(path, lin):
    """The reference's only on-disk format: World::export_luminance, src/world.rs:344-369."""
    h, w, _ = lin.shape
    with open(path, "w") as f:
        f.write("x,y,r,g,b,luminance\n")
        for y in range(h):
            for x in range(w):
                r, g, b = lin[y, x]
                lum = 0.2126 * r + 0.7152 * g + 0.0722 * b
                f.write(f"{x},{y},{r:.6f},{g:.6f},{b:.6f},{lum:.6f}\n")


def random_rays(rng, n):
    """Rays from inside/around the box towards random directions, plus camera-like rays."""
    o = np.stack([rng.uniform(-0.95, 0.95, n), rng.uniform(-0.95, 0.95, n), rng.uniform(-2.9, 1.9, n)], axis=1)
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return np.concatenate([o, d], axis=1)


def main():
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(20251205)
    for name, (sid, arg) in SCENES.items():
        objs = pt.builtin_scene(sid, arg)
        w, h, spp = IMAGES[name]
        cam = pt.camera_new(width=w, height=h)
        prm = pt.default_params(spp=spp)
        lin, rgba, cnt = orc.render(cam, objs, prm, orc.F64, orc.RECURSIVE, threads=8)
        write_luminance_csv(os.path.join(OUT, f"{name}_{w}x{h}x{spp}_luminance.csv"), lin)
        np.save(os.path.join(OUT, f"{name}_{w}x{h}x{spp}_rgba8.npy"), rgba)
        rays = random_rays(rng, 512)
        ids, ts, pn, ff = orc.hit_scene(objs, rays, 0.001, float("inf"), orc.F64)
        np.savez(os.path.join(OUT, f"{name}_hits.npz"), rays=rays, ids=ids, t=ts, point_normal=pn, front_face=ff)
        print(name, "image mean", lin.mean(), "hits", (ids >= 0).mean(), cnt)

    # per-material BSDF vectors (one object each)
    mats = {
        "lambert": pt.make_objects([(0, [0, 0, 0, 1], 0, [0.8, 0.6, 0.2])]),
        "emissive": pt.make_objects([(0, [0, 0, 0, 1], 1, [15, 15, 15])]),
        "glass": pt.make_objects([(0, [0, 0, 0, 1], 2, [0.3, 1, 1, 1, 0.0, 1.5])]),
        "metal": pt.make_objects([(0, [0, 0, 0, 1], 2, [0.2, 0.9, 0.7, 0.3, 1.0, 1.5])]),
        "oren_nayar": pt.make_objects([(0, [0, 0, 0, 1], 3, [0.7, 0.7, 0.7, 0.5])]),
    }
    n = 256
    for name, ob in mats.items():
        nrm = rng.normal(size=(n, 3)); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
        din = rng.normal(size=(n, 3)); din /= np.linalg.norm(din, axis=1, keepdims=True)
        # incoming ray direction points INTO the surface side the normal faces (d.n < 0), like a face-forwarded hit
        flip = (din * nrm).sum(1) > 0
        din[flip] *= -1
        wo = rng.normal(size=(n, 3)); wo /= np.linalg.norm(wo, axis=1, keepdims=True)
        eta = np.where(rng.uniform(size=n) < 0.5, 1.0 / 1.5, 1.5)
        ev_in = np.concatenate([din, wo, nrm, eta[:, None]], axis=1)
        ev = orc.bsdf_eval(ob, ev_in, orc.F64)
        sm_in = np.concatenate([din, nrm, eta[:, None]], axis=1)
        draws = rng.integers(0, 2**32, size=(n, 4), dtype=np.uint64).astype(np.uint32)
        sm = orc.bsdf_sample(ob, sm_in, draws, orc.F64)
        np.savez(os.path.join(OUT, f"bsdf_{name}.npz"), eval_in=ev_in, eval_out=ev, sample_in=sm_in, draws=draws,
                 sample_out=sm)
    # light sampling vectors
    shapes = {
        "sphere": pt.make_objects([(0, [0.0, 0.79, -2.0, 0.2], 1, [36, 36, 36])]),
        "triangle": pt.make_objects([(1, [-0.3, 0.99, -2.3, 0.3, 0.99, -2.3, 0.3, 0.99, -1.7], 1, [15, 15, 15])]),
    }
    for name, ob in shapes.items():
        frm = np.stack([rng.uniform(-0.9, 0.9, n), rng.uniform(-0.9, 0.5, n), rng.uniform(-2.9, -1.1, n)], axis=1)
        r12 = rng.integers(0, 2**23, size=(n, 2)).astype(np.float64)
        r12 = (2 * r12 + 1) / 2.0**24          # the generator's open-interval grid
        out = orc.shape_sample(ob, frm, None, r12, orc.F64)
        tgt = out[:, 0:3].copy()
        out_t = orc.shape_sample(ob, frm, tgt, None, orc.F64)
        np.savez(os.path.join(OUT, f"light_{name}.npz"), frm=frm, r12=r12, sampled=out, with_target=out_t)
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
