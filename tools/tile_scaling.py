"""Time of one rank's share of the C2 job at N = 1, 2, 4, 8 on ONE GPU: the rank-0 row bands of bench.py's strong-scaling
split (interleaved 16-row bands), device-resident, HIP events around `reps` renders.  The per-rank fixed cost (tail,
launch gaps) is what limits strong scaling; the RCCL gather is not in this number.
    python tools/tile_scaling.py [band_rows [export_below]]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pathtrace_amd as pt

band = int(sys.argv[1]) if len(sys.argv) > 1 else 16
ctx = pt.Context(0); ctx.upload(pt.builtin_scene(2))
ctx.set_tuning(export_below=int(sys.argv[2]) if len(sys.argv) > 2 else 0)
cam = pt.camera_new(width=1024, height=1024)
dev = torch.device("cuda", 0)
for n in (1, 2, 4, 8):
    prm = pt.default_params(spp=64, band_rows=band if n > 1 else 0, band_index=0, band_count=n)
    rows = pt.tile_rows(1024, prm.band_rows, 0, n) if n > 1 else 1024
    lin = torch.empty((rows, 1024, 3), dtype=torch.float32, device=dev); rgba = torch.empty((rows, 1024, 4), dtype=torch.uint8, device=dev)
    st = torch.cuda.Stream(dev); ctx.set_stream(st.cuda_stream)
    with torch.cuda.stream(st):
        for _ in range(3): ctx.render_into(cam, prm, lin.data_ptr(), rgba.data_ptr())
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(st)
        reps = 20
        for _ in range(reps): ctx.render_into(cam, prm, lin.data_ptr(), rgba.data_ptr())
        e1.record(st); st.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"N={n}: {rows} rows, {ms:.3f} ms per render -> {1024*1024*64/ms/1e3:.0f} Msamples/s if all N ranks took this long; efficiency vs N=1 ideal below", flush=True)
    if n == 1: base = ms
    else: print(f"      ideal {base/n:.3f} ms, efficiency {base/n/ms:.2f}")
ctx.set_stream(None); ctx.close()
