#!/usr/bin/env python3
"""luminance.csv (the reference's only on-disk output, World::export_luminance, src/world.rs:344-369) of the reference's
own job -- World::new()'s scene, 400 x 400, 3000 spp (world.rs:16-18) -- from the f64 recursive ORACLE (test
infrastructure, CPU).

    python tools/oracle_luminance.py --rng stdrng > oracle_stdrng.csv      # one sequential StdRng stream per pixel
    python tools/oracle_luminance.py --rng philox > oracle_philox.csv      # the build's addressed draws
    (on a box with a Rust toolchain)  cargo run --release                  # writes luminance.csv
    examples/luminance_diff oracle_stdrng.csv luminance.csv 1e-5 1e-5 0.999 1e-6

With --rng stdrng the oracle draws what the reference draws (ChaCha12 behind rand 0.9.2's StdRng, seed (y << 32) | x,
restated from the published algorithm and UNVERIFIED against the crate): if the restatement is exact, the two files agree
to the `{:.6}` the format prints, pixel for pixel -- the number-for-number check this image cannot run (no cargo, no
crates).  If it is not, or with --rng philox / a GPU film, the films are independent estimates of one image: read
`mean_rel` and `rmse` of the diff (3000 spp: per-pixel noise ~1e-2 relative, image mean ~1e-4).
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rng", choices=["stdrng", "philox"], default="stdrng")
    ap.add_argument("--width", type=int, default=400)      # world.rs:16
    ap.add_argument("--height", type=int, default=400)     # world.rs:17
    ap.add_argument("--spp", type=int, default=3000)       # world.rs:18
    ap.add_argument("--scene", type=int, default=1)
    ap.add_argument("--integrator", type=int, default=0, help="0 MIS (default cargo feature), 1 BRDF-only")
    ap.add_argument("--threads", type=int, default=len(os.sched_getaffinity(0)))
    a = ap.parse_args()
    import pathtrace_amd as pt
    from oracle import orc
    objs = pt.builtin_scene(a.scene)
    cam = pt.camera_new(width=a.width, height=a.height)
    prm = pt.default_params(spp=a.spp, integrator=a.integrator)
    if a.rng == "stdrng":
        lin, _, _ = orc.render_stdrng(cam, objs, prm, a.threads)
    else:
        lin, _, _ = orc.render(cam, objs, prm, orc.F64, orc.RECURSIVE, a.threads)
    out = sys.stdout
    out.write("x,y,r,g,b,luminance\n")                                        # world.rs:352
    for y in range(a.height):
        for x in range(a.width):
            r, g, b = lin[y, x]
            out.write(f"{x},{y},{r:.6f},{g:.6f},{b:.6f},{0.2126 * r + 0.7152 * g + 0.0722 * b:.6f}\n")   # world.rs:355-365


if __name__ == "__main__":
    main()
