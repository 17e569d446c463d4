"""The scheduler and the self-verifying statistics on the GPU.

tests/test_sched_cpu.py checks the PLAN of a render against the lane protocol without a GPU; here the same planner runs inside
pt_render_device*: samples are counted on the device and compared with pixels x spp by pt_sync, a stream operation that fails
half-way leaves a context the next renders are right in, and a refused frame does not hang the copy-mode exchange."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _eq(a, b):
    import torch
    return bool(torch.equal(a, b))


# scene, level0_form, width, height, spp, max_paths (0 = default): every launch form render_impl knows
FORMS = [
    ("regen C2", 2, 0, 1024, 512, 8, 0),
    ("split C1", 1, 0, 1024, 512, 8, 0),
    ("queue + continuation C1", 1, 1, 1024, 1024, 8, 0),
    ("queue small C2", 2, 0, 64, 64, 4, 0),
    ("regen, three batches", 2, 0, 1024, 512, 9, 3 * 1024 * 512),
    ("queue, overlapped batches", 1, 1, 256, 256, 12, 4 * 256 * 256),
]


@pytest.mark.parametrize("name,scene,form,w,h,spp,cap", FORMS, ids=[f[0] for f in FORMS])
def test_samples_are_counted_on_the_device(pt, gpu_ctx, name, scene, form, w, h, spp, cap):
    """PtStats.samples = paths whose radiance the kernels wrote to the sample buffer (a device counter), samples_expected =
    pixels x spp (host arithmetic); pt_sync fails if they differ.  One render, then four enqueued back to back."""
    import torch
    gpu_ctx.upload(pt.builtin_scene(scene))
    gpu_ctx.set_tuning(level0_form=form)
    try:
        cam = pt.camera_new(width=w, height=h)
        prm = pt.default_params(spp=spp, max_paths_in_flight=cap)
        ref, ref8 = gpu_ctx.render(cam, prm)
        st = gpu_ctx.stats()
        assert st.samples == st.samples_expected == w * h * spp
        one = (st.vertices, st.shadow_rays)
        outs = [(torch.zeros_like(ref), torch.zeros_like(ref8)) for _ in range(4)]
        for lin, rgba in outs:
            gpu_ctx.render_into(cam, prm, lin.data_ptr(), rgba.data_ptr())
        st = gpu_ctx.stats()
        assert st.samples == st.samples_expected == 4 * w * h * spp
        assert (st.vertices, st.shadow_rays) == (4 * one[0], 4 * one[1])
        for lin, rgba in outs:
            assert _eq(lin, ref) and _eq(rgba, ref8)
    finally:
        gpu_ctx.set_tuning()


def test_pixel_lists_rays_and_bvh_renders_count_their_samples(pt, gpu_ctx):
    gpu_ctx.upload(pt.builtin_scene(1))
    cam = pt.camera_new(width=200, height=120)
    xy = np.stack([np.arange(300) % 200, np.arange(300) % 120], 1).astype(np.uint32)
    gpu_ctx.render_pixels(cam, pt.default_params(spp=7), xy)
    st = gpu_ctx.stats()
    assert st.samples == st.samples_expected == 300 * 7
    rays = np.tile(np.array([[0.0, 0.0, 2.0, 0.1, 0.05, -1.0]]), (500, 1))
    gpu_ctx.ray_color(pt.default_params(spp=1), rays, np.zeros((500, 2), np.uint32))
    st = gpu_ctx.stats()
    assert st.samples == st.samples_expected == 500
    gpu_ctx.upload(pt.builtin_scene(4, 2000))
    cam = pt.camera_new(width=256, height=128)
    for accel in (0, 1):
        gpu_ctx.render(cam, pt.default_params(spp=4, accel=accel))
        st = gpu_ctx.stats()
        assert st.samples == st.samples_expected == 256 * 128 * 4, accel


def test_graph_replays_count_multiples_of_the_captured_render(pt, gpu_ctx):
    """A captured render runs zero or more times: `samples` is then samples_expected + k x its size, and pt_sync accepts exactly
    that.  After a capture the context no longer trusts its "statistics are zero" flag (a replay may have run since)."""
    import torch
    gpu_ctx.upload(pt.builtin_scene(2))
    cam = pt.camera_new(width=512, height=512)
    prm = pt.default_params(spp=8)
    ref, ref8 = gpu_ctx.render(cam, prm)
    n = 512 * 512 * 8
    dev = torch.device("cuda", 0)
    lin = torch.zeros_like(ref); rgba = torch.zeros_like(ref8)
    stream = torch.cuda.Stream(dev)
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.stream(stream):
            gpu_ctx.set_stream(stream.cuda_stream)
            with torch.cuda.graph(g, stream=stream):
                gpu_ctx.render_into(cam, prm, lin.data_ptr(), rgba.data_ptr())
            for _ in range(3):
                g.replay()
            stream.synchronize()
            st = gpu_ctx.stats()
            assert st.samples_expected == 0 and st.samples == 3 * n       # replays add to the period they run in
            assert _eq(lin, ref)
            # a replay after the collection, then a direct render: the direct render's statistics are its own
            g.replay()
            stream.synchronize()
            lin2, rgba2 = gpu_ctx.render(cam, prm)
            st = gpu_ctx.stats()
            assert st.samples == st.samples_expected == n
            assert _eq(lin2, ref) and _eq(rgba2, ref8)
            # ... and a direct lanes render right behind a replay waits for it (found by tests/test_sched_cpu.py)
            outs = [(torch.zeros_like(ref), torch.zeros_like(ref8)) for _ in range(3)]
            for k, (l, r8) in enumerate(outs):
                g.replay()
                gpu_ctx.render_into(cam, prm, l.data_ptr(), r8.data_ptr())
            gpu_ctx.sync()
            for l, r8 in outs:
                assert _eq(l, ref) and _eq(r8, ref8)
            assert _eq(lin, ref) and _eq(rgba, ref8)
    finally:
        gpu_ctx.set_stream(None)
        del g


@pytest.mark.parametrize("name,scene,form,w,h,spp,cap", FORMS, ids=[f[0] for f in FORMS])
def test_a_stream_operation_that_fails_half_way(pt, gpu_ctx, name, scene, form, w, h, spp, cap):
    """pt_debug_fail_after makes the n-th stream operation of the next render fail as a HIP call would.  Whatever n: the call
    reports the failure, the next render of the same job is bit-identical to the reference, pipelined renders behind it too,
    and the statistics verify (the cut-off render left counters and sample buffers in any state)."""
    import torch
    gpu_ctx.upload(pt.builtin_scene(scene))
    gpu_ctx.set_tuning(level0_form=form)
    try:
        cam = pt.camera_new(width=w, height=h)
        prm = pt.default_params(spp=spp, max_paths_in_flight=cap)
        ref, ref8 = gpu_ctx.render(cam, prm)
        base = gpu_ctx.stats()
        lin = torch.zeros_like(ref); rgba = torch.zeros_like(ref8)
        hit = 0
        for cut in range(0, 40):
            # two good renders in flight, then the one that is cut short
            gpu_ctx.render_into(cam, prm, lin.data_ptr(), rgba.data_ptr())
            gpu_ctx.render_into(cam, prm, lin.data_ptr(), rgba.data_ptr())
            gpu_ctx.fail_after(cut)
            try:
                gpu_ctx.render_into(cam, prm, lin.data_ptr(), rgba.data_ptr())
                failed = False
            except pt._lib.PtError as e:
                failed = True
                assert "injected failure" in str(e)
            gpu_ctx.fail_after(-1)
            if not failed:
                gpu_ctx.sync()
                break                                   # the plan has fewer operations than `cut`
            hit += 1
            outs = [(torch.zeros_like(ref), torch.zeros_like(ref8)) for _ in range(3)]
            for l, r8 in outs:
                gpu_ctx.render_into(cam, prm, l.data_ptr(), r8.data_ptr())
            st = gpu_ctx.stats()                        # would raise: the device's sample count must match
            assert st.samples == st.samples_expected == 3 * w * h * spp
            assert (st.vertices, st.shadow_rays) == (3 * base.vertices, 3 * base.shadow_rays)
            for l, r8 in outs:
                assert _eq(l, ref) and _eq(r8, ref8), cut
        assert hit >= 3
    finally:
        gpu_ctx.fail_after(-1)
        gpu_ctx.set_tuning()


@pytest.mark.parametrize("threaded", [False, True])
def test_a_refused_frame_does_not_hang_the_exchange_by_copies(pt, gpu_ctx, threaded):
    """in PT_EXCHANGE_COPY mode -- and in the shared-device debug object, which always copies -- a frame whose
    render was refused (spp = 0) left the host latches behind, and the NEXT frame's enqueue waited for ever.  The latches now
    advance on every way out: the refused frame reports its error, the next frame is right, close() returns."""
    import torch
    objs = pt.builtin_scene(2)
    gpu_ctx.upload(objs)
    cam = pt.camera_new(width=192, height=117)
    ref, ref8 = gpu_ctx.render(cam, pt.default_params(spp=5))
    dev = torch.device("cuda", 0)
    m = pt.Multi([0, 0, 0], shared_device=0)
    try:
        m.upload(objs)
        m.set_threads(threaded)
        lin = torch.zeros((117, 192, 3), dtype=torch.float32, device=dev)
        rgba = torch.zeros((117, 192, 4), dtype=torch.uint8, device=dev)
        m.render_into(cam, pt.default_params(spp=5), lin.data_ptr(), rgba.data_ptr())
        m.sync()
        assert _eq(lin, ref)
        bad = pt.default_params(spp=5)
        bad.spp = 0
        with pytest.raises(pt._lib.PtError):
            m.render_into(cam, bad, lin.data_ptr(), rgba.data_ptr())       # one thread: refused at once; threads: at the next sync
            m.sync()
        lin.zero_(); rgba.zero_()
        for _ in range(3):                                                  # the frames after it complete and are right
            m.render_into(cam, pt.default_params(spp=5), lin.data_ptr(), rgba.data_ptr())
        m.sync()
        assert _eq(lin, ref) and _eq(rgba, ref8)
    finally:
        m.close()


def test_an_unaligned_rgba8_pointer_is_refused(pt, gpu_ctx):
    """the resolve stores a pixel's RGBA8 as one 32-bit word; a byte-offset pointer is an argument error, not a GPU fault."""
    import torch
    gpu_ctx.upload(pt.builtin_scene(2))
    cam = pt.camera_new(width=64, height=64)
    lin = torch.zeros((64, 64, 3), dtype=torch.float32, device="cuda:0")
    raw = torch.zeros(64 * 64 * 4 + 8, dtype=torch.uint8, device="cuda:0")
    with pytest.raises(pt._lib.PtError) as e:
        gpu_ctx.render_into(cam, pt.default_params(spp=1), lin.data_ptr(), raw.data_ptr() + 1)
    assert e.value.code == 1 and "aligned" in str(e.value)
    gpu_ctx.render_into(cam, pt.default_params(spp=1), lin.data_ptr(), raw.data_ptr() + 4)
    gpu_ctx.sync()


def _bench(*args):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "1", "--no-cpu-baseline", *args],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


@pytest.mark.parametrize("args", [[], ["--workload", "c1"], ["--in-order"], ["--level0-form", "1"], ["--force-multi"], ["--force-dist"],
                                  ["--gpus", "2", "--shared-device"]],
                         ids=["default", "c1", "in_order", "queue_form", "force_multi", "force_dist", "shared2"])
def test_the_bench_record_verifies_itself(args):
    """bench.py keeps the device counters of its timed (overlapped) region and compares the last timed frame, bit for bit, with
    the job rendered alone and in order by a context of its own -- in every mode."""
    d = _bench(*args)
    c = d["config"]
    assert "verification_failed" not in d
    assert c["frame_equals_single_gpu"] is True and c["timed_region_counters_equal_steps_x_per_step"] is True
    assert c["timed_region_samples"] == 4 * c["samples_per_step"] == 4 * 1024 * 1024 * 64
    assert abs(c["timed_region_vertices_per_sample"] - c["vertices_per_sample"]) < 1e-9
    if "--force-multi" in args:
        x = c["exchange_copy"]
        assert x["frame_equals_single_gpu"] is True and x["value"] > 0 and c["rccl"]["ncclCommCount"] == 1
    r = d["roofline"]
    assert 0.0 < r["frac"] < 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 2e-3
    if "--level0-form" in args:
        # the queue form (north_star's SoA path queue in HBM) is priced against the HBM peak: 128 B per continuing vertex + 12 B per sample
        assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and 0.0 < r["valu_frac"] < 1.0
        per_launch = 1024 * 1024 * 64 * r["vertex_share"]        # the level-0 launch's share of the job (the continuation launch takes the tails)
        v = c["vertices_per_sample"] * per_launch
        assert abs(r["algorithmic_bytes_per_launch"] - (128.0 * (v - per_launch) + 12.0 * per_launch)) <= 1e-3 * r["algorithmic_bytes_per_launch"]
    else:
        assert r["bound"] == "valu" and r["unit"] == "TFLOP/s"
