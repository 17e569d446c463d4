"""Where the fixed cost of the regenerating kernel sits (measurement build: tools/build_variant_full.sh drain -DPT_DRAIN_TIMING,
run with PATHTRACE_AMD_LIB=pathtrace_amd/libpt_drain.so): per render of one rank's share of C2 at N ranks, device-side
wall_clock64 stamps (100 MHz) of the first wave's start, the first / last wave that found the batch used up, and the last wave's end.
    python tools/r04/drain_timing.py [workload scene id, default 2]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import pathtrace_amd as pt

scene = int(sys.argv[1]) if len(sys.argv) > 1 else 2
ctx = pt.Context(0); ctx.upload(pt.builtin_scene(scene))
cam = pt.camera_new(width=1024, height=1024)
M = (1 << 64) - 1
for n in (1, 2, 4, 8, 16, 64):
    prm = pt.default_params(spp=64, band_rows=max(1, 128 // n) if n > 1 else 0, band_index=0, band_count=n)
    for rep in range(3):
        lin, rgba = ctx.render(cam, prm)
        st = ctx.stats()
        raw = (C.c_uint64 * 16)()
        pt._lib.check(pt._lib.lib().pt_debug_raw_stats(ctx._h, raw))
    t0, tx0, tx1, t1, dsum = (M - raw[8]), (M - raw[9]), raw[10], raw[11], raw[12]
    tick = 0.01   # us per tick (100 MHz)
    waves = 256 * 6 * 4
    print(f"N={n:2d}: {lin.shape[0]:4d} rows, total_ms {st.total_ms:.3f}: kernel span {(t1 - t0) * tick:8.1f} us = "
          f"start -> first wave out of work {(tx0 - t0) * tick:8.1f} | -> last wave out of work {(tx1 - tx0) * tick:7.1f} | "
          f"-> last wave done {(t1 - tx1) * tick:7.1f} us; mean drain per wave {dsum * tick / waves:7.1f} us; deepest vertex {st.max_depth_reached}", flush=True)
ctx.close()

# ---- per-wave picture of the N = 8 share: when did each wave run out of work / end, how much did it process
ctx = pt.Context(0); ctx.upload(pt.builtin_scene(scene))
import numpy as np
for n in (8, 1):
    prm = pt.default_params(spp=64, band_rows=16 if n > 1 else 0, band_index=0, band_count=n)
    for rep in range(2):
        lin, rgba = ctx.render(cam, prm)
    nw = 256 * 6 * 4
    buf = np.zeros((nw, 4), dtype=np.uint32)
    fn = pt._lib.lib().pt_debug_wave_dump
    fn.restype = C.c_int; fn.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
    pt._lib.check(fn(ctx._h, buf.ctypes.data_as(C.c_void_p), nw))
    t0 = buf[:, 0].astype(np.int64); tx = buf[:, 1].astype(np.int64); t1 = buf[:, 2].astype(np.int64); v = buf[:, 3].astype(np.int64)
    base = t0.min()
    b, x, e = (t0 - base) * 0.01, (tx - base) * 0.01, (t1 - base) * 0.01
    print(f"N={n}: waves {nw}; begin: min {b.min():.1f} max {b.max():.1f} us; out of work: p1 {np.percentile(x,1):.1f} p50 {np.percentile(x,50):.1f} p99 {np.percentile(x,99):.1f} max {x.max():.1f}; "
          f"end: p1 {np.percentile(e,1):.1f} p50 {np.percentile(e,50):.1f} p90 {np.percentile(e,90):.1f} p99 {np.percentile(e,99):.1f} max {e.max():.1f}")
    print(f"      drain per wave (end - out of work): p50 {np.percentile(e-x,50):.1f} p90 {np.percentile(e-x,90):.1f} p99 {np.percentile(e-x,99):.1f} max {(e-x).max():.1f} us")
    print(f"      vertices per wave: min {v.min()} p10 {np.percentile(v,10):.0f} p50 {np.percentile(v,50):.0f} p90 {np.percentile(v,90):.0f} max {v.max()}")
    wg = np.arange(nw) // 4
    # by launch order of the workgroup (sixths of the grid = age rank on the SIMD, if workgroups fill the CUs in order)
    for k in range(6):
        sel = (wg * 6 // (nw // 4)) == k
        print(f"      workgroups {k}/6: vertices per wave mean {v[sel].mean():.0f}, out of work mean {x[sel].mean():.1f}, end mean {e[sel].mean():.1f} max {e[sel].max():.1f}")
    late = np.argsort(e)[-8:]
    for w in late:
        print(f"      late wave {w} (workgroup {w // 4}): begin {b[w]:.1f} out of work {x[w]:.1f} end {e[w]:.1f} vertices {v[w]}")
ctx.close()
