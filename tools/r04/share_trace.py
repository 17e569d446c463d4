"""One rank's share of the C2 job at N = 8 (rank 0's interleaved 16-row bands: 128 rows x 1024 x 64 spp), rendered `reps`
times back to back on one stream -- the per-render fixed cost under the microscope.  Run it under
    rocprofv3 --kernel-trace --memory-copy-trace -d gpurun_out/share_trace -o st -- python3 tools/r04/share_trace.py
and feed the kernel trace to tools/r04/trace_gaps.py: duration of every kernel of a render and the idle gaps between them.
    python tools/r04/share_trace.py [n_ranks [reps [packed]]]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import pathtrace_amd as pt

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
packed = len(sys.argv) > 3 and sys.argv[3] == "1"
ctx = pt.Context(0); ctx.upload(pt.builtin_scene(2))
cam = pt.camera_new(width=1024, height=1024)
dev = torch.device("cuda", 0)
prm = pt.default_params(spp=64, band_rows=16 if n > 1 else 0, band_index=0, band_count=n)
rows = pt.tile_rows(1024, prm.band_rows, 0, n)
lin = torch.empty((rows, 1024, 3), dtype=torch.float32, device=dev); rgba = torch.empty((rows, 1024, 4), dtype=torch.uint8, device=dev)
pk = torch.empty((rows, 1024, 16), dtype=torch.uint8, device=dev)
st = torch.cuda.Stream(dev); ctx.set_stream(st.cuda_stream)
def one():
    if packed: ctx.render_packed_into(cam, prm, pk.data_ptr())
    else: ctx.render_into(cam, prm, lin.data_ptr(), rgba.data_ptr())
with torch.cuda.stream(st):
    for _ in range(3): one()
    st.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record(st)
    for _ in range(reps): one()
    t_host = time.perf_counter() - t0
    e1.record(st); st.synchronize()
ms = e0.elapsed_time(e1) / reps
print(f"N={n}: {rows} rows, {ms:.4f} ms per render on the stream ({'packed' if packed else 'two planes'}); host enqueue {t_host / reps * 1e6:.1f} us per render", flush=True)
ctx.set_stream(None); ctx.close()
