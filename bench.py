#!/usr/bin/env python3
"""bench.py -- Msamples/s of the rendering hot path on N MI355X of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
        --master-port P bench.py --gpus N --steps K --warmup W

Both launch forms work for every N (launch_mode() below): under torch.distributed.run it is one process per GPU
(pathtrace_amd/dist.py: pack kernel, ONE dist.gather over RCCL, unpack kernel); started as a plain process with
--gpus N > 1 it is the library's single-process form (pt_multi_*: one context per device, ONE ncclGather inside
an ncclGroup, include/pathtrace_amd.h) -- no launcher needed and no re-exec after the GPU has been touched.

A step = one complete render of the workload through the C ABI (pt_render_device):
camera-ray generation, every bounce launch, film resolve, and for N > 1 the single
RCCL gather of the framebuffer to rank 0.  Scene and camera are resident in HBM
before the timed region; outputs stay in HBM (no PCIe in the timed region).

Workload (BASELINE.json configs[1]): 10-sphere diffuse Cornell scene ("C2"),
1024x1024, 64 spp, MIS integrator, reference path-depth policy.  N > 1 is STRONG scaling, as
BASELINE.json's metric asks ("at 1024^2/64spp, 1/2/4/8 GPU"): the job stays 1024x1024x64 spp,
its rows are dealt to the ranks in interleaved bands and ONE gather per step assembles the
frame on rank 0.  `--workload c5` is the configuration built for 8 GPUs (3840x2160x1024 spp).

One JSON line on stdout (rank 0).  `roofline` prices the dominant kernel (the level-0 launch
of a sample batch: k_paths_regen for the default workload) against what bounds it: the f32 VALU (bound "valu": there is no dense contraction
on this path, so the schema's "mfma" slot does not apply; the peak is the same 157.3 TFLOP/s
f32 rate).  achieved = ALGORITHMIC flops of SURVEY 8(d) -- F_isect per primitive test of every
scan the launch ran (23 per sphere, 51 per triangle) + 200 per path vertex -- divided by the
HIP-event time of those launches measured in this run.  `hbm_frac` and `valu_issue_frac` come
from the committed rocprofv3 counter passes of the same command and carry their source file.
`cpu_baseline` times the oracle (CPU restatement of the reference, f64 recursive,
std::thread over pixels like rayon) on a bounded sample of the same workload; the
Rust reference itself cannot be built here or on the GPU box (no cargo/rustc).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WIDTH = HEIGHT = 1024
SPP = 64
# --workload: the other BASELINE.json configs through the same harness (default c2 = the headline config).
# name -> (scene id, scene arg, width, height, spp, description)
WORKLOADS = {
    "c1": (1, 0, 1024, 1024, 64, "C1: reference Cornell box + GGX glass sphere (World::new(), 13 objects)"),
    "c2": (2, 0, 1024, 1024, 64, "C2: 10-sphere diffuse Cornell scene"),
    "c3": (2, 0, 1024, 1024, 4096, "C3: 10-sphere diffuse Cornell scene, steady state"),
    "c4": (4, 10000, 1024, 1024, 256, "C4: 10 000 random spheres (100 lights)"),
    "c5": (2, 0, 3840, 2160, 1024, "C5: 10-sphere diffuse Cornell scene, 4K"),
    # the reference's literal job: World::new() at WIDTH = HEIGHT = 400, SAMPLE_NUM = 3000 (world.rs:16-18) -- the only
    # configuration the reference itself ships; `host_buffers` in the JSON line is the same job through pt_render with HOST
    # film buffers (main.rs:58-66 ends with the film on the host), PCIe included
    "ref": (1, 0, 400, 400, 3000, "REF: World::new() as the reference ships it (world.rs:16-18)"),
}
BYTES_PER_VERTEX = 252      # extend 32 + shade 144 + shadow/accumulate 68 + compaction 8 (a five-kernel pipeline;
BYTES_PER_SAMPLE = 64       # the fused kernel never makes those round trips: informational only)
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s achievable float4 copy)
VALU_PEAK_TFLOPS = 157.3    # MI355X_MICROARCH.md: peak FP32 vector (= the f32 MFMA rate)
F_SPHERE, F_TRIANGLE, F_SHADE = 23, 51, 200     # flops per primitive test / per shaded vertex (Lambert)
# ... where 51 prices the reference's Moeller-Trumbore test (two cross products per ray, shape.rs:163-188).  The kernels run the
# plane form on per-triangle constants -- d.n 5, s 3, s.n 5, t 1, hit point 6, u 5, v 5, u + v 1 = 31 flops -- and test two
# triangles of one parallelogram TOGETHER (tripair_test: determinant, t and hit point once = 20, then 2 x 11): the model prices
# what is executed, so that `frac` does not credit work nobody does.
F_TRIANGLE_PLANE, F_TRIANGLE_PAIR = 31, 42
PROFILE_DIR = os.path.join(ROOT, "profiles", "r05")
REGEN_MIN_PATHS = 1 << 17     # pt_api.cpp kRegenMinPaths: batches above this over a scene in LDS take a regenerating form


def launch_mode(gpus, env, force_dist=False, force_multi=False):
    """How this process takes part in an N-GPU run -> (mode, world, rank, local_rank).
      "dist"    one process per GPU under torch.distributed.run (WORLD_SIZE > 1 in the environment, or --force-dist):
                the environment's world size wins over --gpus;
      "multi"   ONE plain process and --gpus N > 1 (or --force-multi): the library's pt_multi_* path over N devices;
      "single"  one GPU, no collective."""
    world = int(env.get("WORLD_SIZE", "1"))
    rank = int(env.get("RANK", "0"))
    local_rank = int(env.get("LOCAL_RANK", "0"))
    if force_dist and force_multi:
        raise SystemExit("--force-dist and --force-multi exclude each other")
    if world > 1 or force_dist:
        if force_multi:
            raise SystemExit("--force-multi is the single-process form: do not start it under torch.distributed.run")
        return "dist", world, rank, local_rank
    if gpus > 1 or force_multi:
        return "multi", max(1, gpus), 0, 0
    return "single", 1, 0, 0


def cpu_baseline(pt, objs, width, height, spp_full, desc):
    """Oracle (kind "port") on the host: the workload's scene at the bench camera, a row subset spread over the image.
    Threads = the box's CPU share for one GPU (16), never more than the affinity mask allows; `all_cores` repeats a bounded
    run on every CPU the process may use (north_star: "host cores (core count stated)"; the reference's rayon loop,
    main.rs:48, takes all of them).  A short probe sizes the samples so that each timed run is ~10 s of wall time."""
    from oracle import orc
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 16))
    cam = pt.camera_new(width=width, height=height)

    def run(spp, band_count, threads):
        prm = pt.default_params(spp=spp, band_rows=1, band_index=0, band_count=band_count)
        t0 = time.perf_counter()
        lin, _, _ = orc.render(cam, objs, prm, orc.F64, orc.RECURSIVE, threads=threads)
        return lin.shape[0], time.perf_counter() - t0

    def timed(threads, budget_s, rate=None):
        if rate is None:
            rows, dt = run(1, 64, threads)                  # probe: every 64th row x 1 spp
            rate = rows * width / dt                        # samples / s
        # full spp on every band_count-th row; fewer samples per pixel only if even every 64th row is too much
        band_count, spp = 64, spp_full
        for bc in (32, 16, 8, 4, 2, 1):
            if len(range(0, height, bc)) * width * spp_full / rate <= budget_s:
                band_count = bc
        rows = len(range(0, height, band_count))
        if rows * width * spp_full / rate > 2.5 * budget_s:
            spp = int(max(1, 2.5 * budget_s * rate / (rows * width)))
        rows, dt = run(spp, band_count, threads)
        return rows, spp, band_count, dt

    rows, spp, band_count, dt = timed(cores, 12.0)
    samples = rows * width * spp
    rate16 = samples / dt
    out = {
        "value": round(samples / dt / 1e6, 4),
        "unit": "Msamples/s",
        "cores": cores,
        "kind": "port",
        "sample": f"{desc}, {width}x{height} camera, every {band_count}th row ({rows} rows), {spp} of {spp_full} spp = "
                  f"{samples} samples in {dt:.2f} s wall; oracle f64 recursive (reference-shaped: linear scan, "
                  f"3 scans per vertex), {cores} threads over pixels ({avail} CPUs visible)",
    }
    if avail > cores:
        # sized by the rate just measured, not by a probe: a 1-spp probe on hundreds of threads mostly times their start-up
        # (and where the container's CPU quota is below the visible CPU count, more threads are slower, not faster)
        rows, spp, band_count, dt = timed(avail, 4.0, rate16)
        out["all_cores"] = {"value": round(rows * width * spp / dt / 1e6, 4), "unit": "Msamples/s", "cores": avail,
                            "sample": f"same oracle on every visible CPU: every {band_count}th row ({rows} rows), {spp} spp in {dt:.2f} s wall"}
    else:
        out["all_cores"] = {"value": out["value"], "unit": "Msamples/s", "cores": avail, "sample": "the figure above already uses every visible CPU"}
    # BASELINE.json configs[0]: the reference's own scene, 256 x 256, 4 spp, ONE thread (the scalar port)
    c1_objs = pt.builtin_scene(1)
    c1_cam = pt.camera_new(width=256, height=256)
    t0 = time.perf_counter()
    orc.render(c1_cam, c1_objs, pt.default_params(spp=4), orc.F64, orc.RECURSIVE, threads=1)
    c1_dt = time.perf_counter() - t0
    out["config0_single_thread"] = {"value": round(256 * 256 * 4 / c1_dt / 1e6, 4), "unit": "Msamples/s", "cores": 1,
                                    "sample": f"C1 reference scene, 256x256, 4 spp = 262144 samples in {c1_dt:.2f} s"}
    return out


def counter_problems(tag, timed, want_samples, steps, ref=None, ref_steps=0):
    """The device counters of a timed region of `steps` renders of a deterministic job -> list of problems (empty: fine).
    timed: {"samples": finished samples counted on the device, "samples_expected": the library's own pixels x spp sum,
    "vertices", "shadow_rays"}; ref: the same counters of `ref_steps` renders of the job run one by one (None: only
    divisibility by `steps` can be checked)."""
    problems = []
    if timed["samples"] != want_samples or timed["samples_expected"] != want_samples:
        problems.append(f"{tag}: the device finished {timed['samples']} samples, the library expected {timed['samples_expected']}, "
                        f"{steps} renders of the job have {want_samples}")
    for key in ("vertices", "shadow_rays"):
        if ref is not None:
            if timed[key] * ref_steps != ref[key] * steps:
                problems.append(f"{tag}: {key} {timed[key]} != {steps} x the per-render {ref[key] / max(ref_steps, 1)}")
        elif timed[key] % steps:
            problems.append(f"{tag}: {key} {timed[key]} is not a multiple of {steps} renders")
    return problems


def profile_summary(world, workload, accel, level0_form=0):
    """Counter-derived figures of the dominant kernel from the committed rocprofv3 passes of THIS command
    (profiles/r05/roofline_<workload>[_bvh | _queue].json, written by tools/profile_workload.sh on the GPU box).  PMC counters
    cannot be read from inside this process, so these are the last profiled values, tagged with their source file;
    None when no profile of this exact workload exists (or N > 1)."""
    if world != 1:
        return None
    if level0_form not in (0, 1):
        return None         # a forced regenerating form has no committed profile of its own
    name = f"roofline_{workload}{'_bvh' if accel == 1 else ''}{'_queue' if level0_form == 1 else ''}.json"
    path = os.path.join(PROFILE_DIR, name)
    if not os.path.exists(path):
        return None
    try:
        with open(path) as f:
            d = json.load(f)
        d["_file"] = os.path.relpath(path, ROOT)
        return d
    except Exception:
        return None


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--max-paths", type=int, default=0, help="PtRenderParams.max_paths_in_flight (0 = default)")
    ap.add_argument("--workgroups", type=int, default=0)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--accel", type=int, default=0, choices=[0, 1, 2],
                    help="PtRenderParams.accel: 0 = the reference's linear scan (every reported config), 1 = BVH (same film), "
                         "2 = PT_ACCEL_AUTO, the product default (the BVH for C4-sized scenes)")
    ap.add_argument("--cont-workgroups", type=int, default=0, help="PtTuning.cont_workgroups (0 = library default)")
    ap.add_argument("--export-below", type=int, default=0, help="PtTuning.export_below (0 = library default)")
    ap.add_argument("--level0-form", type=int, default=0, help="PtTuning.level0_form (0 = library default, 1 = queue form, 2 = regenerating form, 3 = regenerating form with batched Mirror vertices)")
    ap.add_argument("--regen-workgroups", type=int, default=0, help="PtTuning.regen_workgroups (0 = library default)")
    ap.add_argument("--in-order", action="store_true",
                    help="round 3's method: PtTuning.in_order = 1 (consecutive launches never overlap) and HIP events around every launch of "
                         "the timed steps; default: the timed steps overlap and the launch times come from 3 in-order steps after them")
    ap.add_argument("--exchange", default="rccl", choices=["rccl", "copy"],
                    help="single-process multi-device form: how the tiles reach device 0 -- one ncclGather per frame (default) or one DMA "
                         "copy per device (pt_multi_set_exchange: no kernel takes part in the exchange)")
    ap.add_argument("--weak", action="store_true", help="N > 1: weak scaling (64*N spp) instead of the strong-scaling default")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo = rehearsal on a box with fewer GPUs than ranks (ranks share devices, the gather "
                         "goes through host memory); the driver's runs use nccl (RCCL)")
    ap.add_argument("--force-dist", action="store_true",
                    help="with one rank: still initialise the process group and run every collective of the N > 1 path "
                         "(rehearsal of the one-process-per-GPU code over real RCCL on a one-GPU box)")
    ap.add_argument("--force-multi", action="store_true",
                    help="with --gpus 1: still run the single-process multi-device path (pt_multi_*: ncclCommInitAll, the "
                         "ncclGather, the row permutation) -- its rehearsal over real RCCL on a one-GPU box")
    ap.add_argument("--shared-device", action="store_true",
                    help="REHEARSAL of the single-process form with --gpus N > 1 on a one-GPU box: N contexts on device 0, device-to-"
                         "device copies where the real object calls ncclGather (pt_debug_multi_create_shared); everything else -- host "
                         "threads, frames posted back to back, the packed resolve, the row permutation, this script's N > 1 branches -- "
                         "is the real path.  The line it prints is marked as a rehearsal and is not a measurement of N GPUs")
    args = ap.parse_args(argv)

    mode, world, rank, local_rank = launch_mode(args.gpus, os.environ, args.force_dist, args.force_multi)

    # the pool's host driver only supports dmabuf IPC: without this RCCL / cross-process device memory fails with
    # "hipIpcGetMemHandle: invalid argument" (it is exported on the boxes already; kept for environments built by hand)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    import pathtrace_amd as pt
    from pathtrace_amd.dist import FilmGather, default_band_rows

    args.gpus = world
    if args.shared_device and mode != "multi":
        raise SystemExit("--shared-device rehearses the single-process form: start it plainly with --gpus N")
    if mode == "multi" and not args.shared_device and torch.cuda.device_count() < world:
        raise SystemExit(f"--gpus {world}: this host shows {torch.cuda.device_count()} GPU(s)")
    dev_index = local_rank if (mode != "dist" or args.backend == "nccl") else local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist_path = mode == "dist"                  # one process per GPU: process group + dist.gather
    if dist_path:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    comm_dev = dev if args.backend == "nccl" else torch.device("cpu")

    global WIDTH, HEIGHT, SPP
    scene_id, scene_arg, WIDTH, HEIGHT, SPP, wl_desc = WORKLOADS[args.workload]
    objs = pt.builtin_scene(scene_id, scene_arg)
    cam = pt.camera_new(width=WIDTH, height=HEIGHT)
    spp = SPP * world if args.weak else SPP
    band_rows = default_band_rows(HEIGHT, world) if mode != "single" else 0
    tuning = dict(cont_workgroups=args.cont_workgroups, export_below=args.export_below, level0_form=args.level0_form,
                  regen_workgroups=args.regen_workgroups, in_order=1 if args.in_order else 0)
    # The timed steps are enqueued back to back and carry NO per-launch events: the library then lets the launch of step k + 1 start
    # while the last waves of step k run dry (lanes, pt_api.cpp), which is how a host that renders frame after frame uses it.  A
    # launch that overlaps its neighbours has no duration of its own -- an event pair around it would span its wait for wave
    # slots -- so the launch times for the `roofline` object come from a few extra steps AFTER the timed region, launched
    # strictly one after the other with HIP events around every launch (PtRenderParams.profile = 1).  `--in-order` is round 3's
    # method: every timed step in order and timed launch by launch (no overlap; `value` is then ~6 % lower on C2).
    live_profile = 1 if args.in_order else 0
    common = dict(spp=spp, profile=live_profile, max_paths_in_flight=args.max_paths, workgroups=args.workgroups, accel=args.accel)
    if mode == "multi":
        # ONE process, `world` devices: every device renders its interleaved bands, ONE ncclGather to device 0 (pt_multi.cpp)
        prm = pt.default_params(band_rows=band_rows, **common)
        ctx = pt.Multi(list(range(world)), shared_device=0 if args.shared_device else None)
        ctx.upload(objs)
        ctx.set_tuning(**tuning)
        if args.exchange != "rccl":
            ctx.set_exchange(args.exchange)
        rows = HEIGHT                           # the frame is assembled on device 0 by the library
    else:
        prm = pt.default_params(band_rows=band_rows, band_index=rank, band_count=world, **common)
        ctx = pt.Context(dev_index)
        ctx.upload(objs)
        ctx.set_tuning(**tuning)
        ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
        rows = pt.tile_rows(HEIGHT, band_rows, rank, world)
    lin = torch.empty((rows, WIDTH, 3), dtype=torch.float32, device=dev)
    rgba = torch.empty((rows, WIDTH, 4), dtype=torch.uint8, device=dev)

    acc = {"vertices": 0, "samples": 0, "samples_expected": 0, "bounce_ms": 0.0, "launches": 0, "total_ms": 0.0,
           "p_vertices": 0, "p_ms": 0.0, "p_launches": 0, "shadow_rays": 0}

    def reset_acc():
        for key in acc:
            acc[key] = 0 if isinstance(acc[key], int) else 0.0

    # one process per GPU: the single exchange step of the path is one gather of the framebuffer (f32 + RGBA8 packed) per
    # step.  It is launched asynchronously, so the gather of step k runs (on the backend's stream) while step k + 1 renders;
    # the last one is completed inside the timed region.  (Single-process form: the gather is inside render_into.)
    film_gather = FilmGather(HEIGHT, WIDTH, band_rows, rank, world, comm_dev, always_collective=True) if dist_path else None

    # The steps are enqueued back to back: render, (pack, gather) are ordered by the stream and nothing in a step needs the
    # host; the library adds up the counters and HIP-event launch times of the renders enqueued since the last
    # synchronisation (pathtrace_amd.h: PtStats), so they are read ONCE, after the last step, and cover every timed launch.
    # (One process per GPU: a rank's share of the job is ~1 ms at 8 GPUs, a host round trip per step several percent of it.)
    # The single-process multi-device form does the same since round 4: pt_multi_render_device posts a frame to the devices'
    # host threads and returns, and stream order keeps consecutive frames apart on every device (its send buffer is written
    # by a resolve that follows the previous frame's gather in that device's stream; the root's receive buffer by a gather
    # that follows the previous frame's row permutation).  Only the gloo rehearsal synchronises per step.
    async_steps = mode in ("single", "multi") or (dist_path and args.backend == "nccl")
    # one process per GPU over RCCL: the film resolve writes the gather's send buffer itself (pt_render_device_packed)
    packed_dist = dist_path and args.backend == "nccl"

    def step(last):
        if packed_dist:
            ctx.render_packed_into(cam, prm, film_gather.send.data_ptr())
        else:
            ctx.render_into(cam, prm, lin.data_ptr(), rgba.data_ptr())
        if not async_steps or last:
            ctx.sync()
        if packed_dist:
            film_gather.start_prepacked()
        elif dist_path:
            film_gather.start(lin.to(comm_dev), rgba.to(comm_dev))
        if not async_steps or last:
            st = ctx.stats()
            acc["vertices"] += st.vertices
            acc["shadow_rays"] += st.shadow_rays
            acc["samples"] += st.samples                        # counted on the DEVICE (PtStats.samples; pt_sync compares it with ...
            acc["samples_expected"] += st.samples_expected      # ... pixels x spp of the renders enqueued and fails on a mismatch)
            acc["bounce_ms"] += st.bounce_kernel_ms
            acc["launches"] += st.bounce_launches
            acc["total_ms"] += st.total_ms
            acc["p_vertices"] += st.primary_vertices
            acc["p_ms"] += st.primary_kernel_ms
            acc["p_launches"] += st.primary_launches

    def barrier():
        if dist_path:
            dist.barrier()

    def device_sync():
        for d in (range(world) if mode == "multi" and not args.shared_device else [dev_index]):
            torch.cuda.synchronize(d)

    def timed_region(n_warm):
        """n_warm untimed steps, then EXACTLY args.steps timed steps between barrier + device synchronisation on both sides.
        -> (seconds: max over ranks, device counters of the timed steps, the LAST timed frame as (linear, rgba8) clones on rank 0)"""
        for k in range(n_warm):
            step(k + 1 == n_warm)
        reset_acc()                             # the warm-up steps are not part of the statistics
        if dist_path:
            film_gather.finish()
        barrier()
        device_sync()
        t0 = time.perf_counter()
        for k in range(args.steps):
            step(k + 1 == args.steps)
        fr = fr8 = None
        if dist_path:
            fr, fr8 = film_gather.finish()      # the last frame; earlier ones were completed by the next start()
        device_sync()
        barrier()
        dt = time.perf_counter() - t0
        if dist_path:
            t = torch.tensor([dt], dtype=torch.float64, device=comm_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        counters = dict(acc)
        # the frame the LAST timed step produced, kept for the bitwise check below (later steps reuse the buffers)
        if dist_path:
            last = (fr.clone(), fr8.clone()) if rank == 0 else None
        else:
            last = (lin.clone(), rgba.clone())
        return dt, counters, last

    elapsed, timed, last_frame = timed_region(args.warmup)
    # Single-process multi-device form: BOTH exchanges in the one run -- `value` with ONE ncclGather per frame (north_star's form),
    # `config.exchange_copy` with one DMA copy per device in its place (pt_multi_set_exchange; no kernel takes part) -- so that the
    # first run on a multi-GPU node settles the default by measurement.  Each with its own bitwise frame check.
    copy_run = None
    if mode == "multi" and not args.shared_device and args.exchange == "rccl":
        ctx.set_exchange("copy")
        c_elapsed, c_timed, c_last = timed_region(max(1, args.warmup))
        copy_run = {"elapsed": c_elapsed, "timed": c_timed, "last": c_last}
        ctx.set_exchange("rccl")
    if dist_path:
        tot = torch.tensor([timed["vertices"], timed["samples"]], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        job_samples = float(tot[1].item())
    else:
        job_samples = float(timed["samples"])   # pt_multi_get_stats sums over the devices

    post_pass = not args.in_order
    if post_pass:
        # launch times for the roofline: three more steps, in order, with HIP events around every launch, outside the timed region
        prm.profile = 1
        reset_acc()
        async_was, async_steps = async_steps, False
        for k in range(3):
            step(True)
        if dist_path:
            film_gather.finish()
        async_steps = async_was
        prm.profile = 0
    prof_steps = 3 if post_pass else args.steps       # the steps acc's launch times and counters cover
    multi_info = ctx.info() if mode == "multi" else None

    host_buffers = None
    if args.workload == "ref" and rank == 0 and world == 1:
        # the same job through the one-shot entry with HOST buffers (scene upload, render, film over PCIe): what replaces
        # main.rs:43-66 as it stands
        pt.render_host(cam, objs, pt.default_params(spp=spp))                   # creates the cached context and its buffers
        t1 = time.perf_counter()
        reps = max(1, min(args.steps, 5))
        for _ in range(reps):
            h_lin, h_rgba = pt.render_host(cam, objs, pt.default_params(spp=spp))
        dt = (time.perf_counter() - t1) / reps
        assert (h_lin == last_frame[0].cpu().numpy()).all() and (h_rgba == last_frame[1].cpu().numpy()).all()     # same film as the device-resident step
        host_buffers = {"value": round(WIDTH * HEIGHT * spp / dt / 1e6, 2), "unit": "Msamples/s", "ms_per_render": round(dt * 1e3, 3),
                        "what": "pt_render(): scene upload + render + both film planes copied to host memory (PCIe), blocking; "
                                f"mean of {reps} calls; film bitwise equal to the device-resident step's"}

    # ---- the record verifies itself: the number above rests on overlapped launches, so
    # (b) the DEVICE counters of the timed region must be exactly those of `steps` renders of the job -- samples counted where a
    #     path's radiance is written (PtStats.samples; the library compares it with pixels x spp itself and fails pt_sync on a
    #     mismatch), vertices and shadow rays = steps x the per-step figures of the in-order steps after the region (the job is
    #     deterministic) -- and
    # (c) the LAST timed frame must equal, bit for bit, the job rendered once, alone and in order, on one GPU by a context of its
    #     own: in every mode and for every N.  Any difference -> the JSON line says so and the exit code is non-zero.
    problems = []

    want = rows * WIDTH * spp * args.steps          # this process's share of the job (multi: all devices, summed by the library)
    per_step = acc if post_pass else None
    problems += counter_problems("timed region", timed, want, args.steps, per_step, prof_steps)
    if copy_run:
        problems += counter_problems("timed region (exchange by copies)", copy_run["timed"], want, args.steps, per_step, prof_steps)
    frame_ok = None
    if rank == 0:
        ref_ctx = pt.Context(dev_index)
        ref_ctx.upload(objs)
        ref_ctx.set_tuning(in_order=1)
        ref_lin, ref_rgba = ref_ctx.render(cam, pt.default_params(spp=spp, max_paths_in_flight=args.max_paths, accel=args.accel))
        ref_ctx.close()

        def same(fr):
            return (fr is not None and tuple(fr[0].shape) == (HEIGHT, WIDTH, 3) and bool(torch.equal(fr[0], ref_lin.to(fr[0].device)))
                    and bool(torch.equal(fr[1], ref_rgba.to(fr[1].device))))
        frame_ok = same(last_frame)
        if not frame_ok:
            problems.append("the last timed frame differs from the job rendered alone, in order, on one GPU")
        if copy_run:
            copy_run["frame_ok"] = same(copy_run["last"])
            if not copy_run["frame_ok"]:
                problems.append("exchange by copies: the last timed frame differs from the job rendered alone, in order, on one GPU")
        del ref_lin, ref_rgba
    if dist_path:
        # every rank learns whether any rank found a problem (a rank must not leave the others waiting in a collective)
        flag = torch.tensor([1.0 if problems else 0.0], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if flag.item() and not problems:
            problems.append("another rank reported a problem")
        if problems and rank != 0:
            print(f"[bench rank {rank}] " + "; ".join(problems), file=sys.stderr, flush=True)

    if rank == 0:
        # the dominant kernel = the level-0 launch of each batch (camera rays + every bounce until its waves hand
        # their sparse tails over); it processes p_vertices of the vertices and all of the camera samples.
        # Single-process multi-device runs: counters are sums over the devices, launch times the slowest device's
        # (pt_multi_get_stats), so the per-device figures below divide the work by the device count.
        n_stat = world if mode == "multi" else 1
        n_sph = sum(1 for o in objs if o.shape_tag == 0)
        n_tri = len(objs) - n_sph
        # what one linear scan tests: spheres, single triangles, triangle pairs (the library's own grouping)
        lay_ctx = ctx if mode != "multi" else pt.Context(0)
        if mode == "multi":
            lay_ctx.upload(objs)
        l_sph, l_tri, l_pair = lay_ctx.scan_layout()
        if mode == "multi":
            lay_ctx.close()
        assert l_sph == n_sph and l_tri + 2 * l_pair == n_tri
        f_scan = F_SPHERE * l_sph + F_TRIANGLE_PLANE * l_tri + F_TRIANGLE_PAIR * l_pair     # one linear scan of the scene
        share = acc["p_vertices"] / max(acc["vertices"], 1)                  # the level-0 launches' share of the work
        scans = acc["p_vertices"] + acc["shadow_rays"] * share               # closest-hit scans + visibility scans
        alg_flops = (f_scan * scans + F_SHADE * acc["p_vertices"]) / n_stat
        p_launches = acc["p_launches"] / n_stat
        achieved_tf = alg_flops / (acc["p_ms"] * 1e-3) / 1e12 if acc["p_ms"] > 0 else 0.0
        alg_bytes = (BYTES_PER_VERTEX * acc["p_vertices"] + BYTES_PER_SAMPLE * acc["samples"]) / n_stat
        avg_ms = acc["p_ms"] / max(p_launches, 1)
        prof = profile_summary(world if mode != "multi" else 0, args.workload, args.accel, args.level0_form)
        accel_name = {0: "linear scan (reference)", 1: "BVH traversal (accel=1, same film as the linear scan)",
                      2: "PT_ACCEL_AUTO (product default: BVH above ~512 sphere tests per scan, same film)"}[args.accel]
        diffuse = all(o.mat_tag in (0, 1) for o in objs)
        if args.accel == 1 or (args.accel == 2 and len(objs) > 512):
            kernel = "k_paths_bvh<MIS, OVF=false%s>" % (", DIFFUSE" if diffuse else "")
        elif len(objs) <= 128:
            # large batches over a scene in LDS: a regenerating form by default (pt_api.cpp; PtTuning.level0_form = 1: the queue form)
            big = acc["samples"] / n_stat / max(p_launches, 1) > REGEN_MIN_PATHS
            n_mirror = sum(1 for o in objs if o.mat_tag == 2)
            if big and (args.level0_form == 3 or (args.level0_form == 0 and not diffuse and 0 < 2 * n_mirror <= len(objs))):
                kernel = "k_paths_regen_split<MIS> (regenerating form, the Mirror vertices of a wave shaded in batches of 64)"
            elif big and args.level0_form in (0, 2):
                kernel = "k_paths_regen<MIS, %s>" % ("DIFFUSE" if diffuse else "no Mirror code" if n_mirror == 0 else "every material")
            else:
                kernel = "k_paths<kModeLds, MIS, OVF=false%s>" % (", DIFFUSE" if diffuse else "")
        else:
            kernel = "k_paths<kModeTiled, MIS, OVF=false>"
        roof = {
            "kernel": kernel + (": the one path-kernel launch of a sample batch (camera rays + every bounce of every path; a lane "
                                "whose path ends takes the batch's next one)" if kernel.startswith("k_paths_regen") else
                                ": the level-0 launch of a sample batch (camera rays + every bounce until the waves hand "
                                "over their sparse tails)") +
                      (", rank 0" if mode != "multi" else ", per device (work / devices over the slowest device's launch time)") +
                      ".  Bound by f32 VALU issue (no contraction on this path: the "
                      "schema's mfma slot does not apply; same 157.3 TFLOP/s f32 peak)",
            "bound": "valu",
            "achieved": round(achieved_tf, 2),
            "peak": VALU_PEAK_TFLOPS,
            "unit": "TFLOP/s",
            "frac": round(achieved_tf / VALU_PEAK_TFLOPS, 4),
            "algorithmic_flops_per_launch": round(alg_flops / max(p_launches, 1)),
            "flops_model": f"{f_scan} per scan of the scene ({l_sph} spheres x {F_SPHERE} + {l_tri} single triangles x {F_TRIANGLE_PLANE} "
                           f"(plane form) + {l_pair} triangle pairs x {F_TRIANGLE_PAIR}) x (vertices + visibility scans) + {F_SHADE} per vertex "
                           f"(SURVEY 8d prices a triangle at {F_TRIANGLE} = the reference's Moeller-Trumbore test; the kernels execute less)",
            "avg_launch_ms": round(avg_ms, 4),
            "launches": int(p_launches),
            "vertex_share": round(share, 4),
            "all_path_kernels_ms_per_step": round(acc["bounce_ms"] / max(prof_steps, 1), 4),
            "traffic": None, "hbm_frac": None, "valu_issue_frac": None,
            "algorithmic_bytes_per_launch_unfused_pipeline": round(alg_bytes / max(p_launches, 1)),
        }
        bvh_run = args.accel == 1 or (args.accel == 2 and len(objs) > 512)
        if bvh_run:
            # a BVH traversal has no closed-form algorithmic flop count (the scan model would price work it never does
            # and give a "fraction" above 1): price it with the f32 flops it EXECUTED per the counter pass, if there is one
            roof["achieved"] = roof["frac"] = None
            roof["algorithmic_flops_per_launch"] = None
            roof["flops_model"] = "executed f32 flops of the launch from the rocprofv3 counter pass (SQ_INSTS_VALU_FLOPS_FP32 x 64 lanes x lane utilisation)"
            cp = (prof or {}).get("counters_per_launch", {})
            if cp.get("SQ_INSTS_VALU_FLOPS_FP32") and prof.get("lane_utilisation") and avg_ms > 0:
                ex = cp["SQ_INSTS_VALU_FLOPS_FP32"] * 64.0 * prof["lane_utilisation"]
                roof["achieved"] = round(ex / (avg_ms * 1e-3) / 1e12, 2)
                roof["frac"] = round(roof["achieved"] / VALU_PEAK_TFLOPS, 4)
        if prof:
            src = prof["_file"]
            if prof.get("hbm_bytes_per_launch"):
                roof["traffic"] = int(prof["hbm_bytes_per_launch"])
                # counter bytes over THIS run's launch time, against the 8 TB/s peak
                roof["hbm_frac"] = round(prof["hbm_bytes_per_launch"] / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if avg_ms > 0 else None
            if prof.get("valu_issue_frac"):
                roof["valu_issue_frac"] = round(min(prof["valu_issue_frac"], 1.0), 4)     # against 2 cycles per wave64 VALU instruction
                roof["valu_insts_per_launch"] = int(prof["valu_insts_per_launch"])
            roof["counters_source"] = f"{src}: replayed from the committed rocprofv3 passes of this command ({prof.get('kernel')}, " \
                                      f"{prof.get('avg_launch_ms_kernel_trace', 0):.3f} ms per launch in the profiled process), not measured in this run"
        if kernel.startswith("k_paths<kModeLds") and acc["p_vertices"] > 0 and avg_ms > 0:
            # The queue form (north_star's literal organisation: SoA path queue in HBM, compacted by ballot / prefix sum) is the one
            # kernel of this library that HBM bounds: every vertex whose path goes on writes its 64-byte state and the next pass reads
            # it, every sample writes its 12 bytes.  `achieved` is that algorithmic traffic over the launch time against the 8 TB/s
            # peak; the VALU figures of the same launch move to valu_*.
            p_samples = acc["samples"] * share / n_stat / max(p_launches, 1)
            p_vertices = acc["p_vertices"] / n_stat / max(p_launches, 1)
            q_bytes = 128.0 * max(p_vertices - p_samples, 0.0) + 12.0 * p_samples
            roof["valu_achieved_tflops"], roof["valu_frac"] = roof["achieved"], roof["frac"]
            roof["bound"], roof["unit"], roof["peak"] = "hbm", "GB/s", HBM_PEAK_GBS
            roof["achieved"] = round(q_bytes / (avg_ms * 1e-3) / 1e9, 1)
            roof["frac"] = round(roof["achieved"] / HBM_PEAK_GBS, 4)
            roof["algorithmic_bytes_per_launch"] = int(q_bytes)
            roof["bytes_model"] = "128 B per vertex whose path goes on (64 B state written, 64 B read by the next pass) + 12 B per sample"
            roof["kernel"] = roof["kernel"].replace("Bound by f32 VALU issue (no contraction on this path: the schema's mfma slot does not apply; "
                                                    "same 157.3 TFLOP/s f32 peak)", "Bound by HBM (the path queue's round trip per vertex)")
        if world == 1:
            tiles = "whole image" + (" (through the single-process multi-device path: ncclCommInitAll over 1 device, ONE ncclGather, row "
                                     "permutation; frame checked bitwise against the plain render)" if mode == "multi" else "")
        elif mode == "multi":
            tiles = (f"interleaved bands of {band_rows} rows over {world} devices driven by ONE process (pt_multi_*): ONE ncclGather "
                     f"(RCCL, the calls of the {world} devices inside one ncclGroup) of the packed f32 + RGBA8 tiles to device 0 per step, "
                     f"then the row permutation there; steps posted back to back")
        else:
            tiles = (f"interleaved bands of {band_rows} rows over {world} ranks, ONE {args.backend} gather of the packed f32 + RGBA8 "
                     f"frame to rank 0 per step, overlapped with the next step's rendering")
        out = {
            "metric": ("Msamples/sec (pixels x spp / s) at 1024^2/64spp" if args.workload == "c2" and not args.weak
                       else f"Msamples/sec (pixels x spp / s) at {WIDTH}x{HEIGHT}/{SPP}spp") +
                      ("" if args.accel == 0 else f", hit_scene = {'BVH' if args.accel == 1 else 'PT_ACCEL_AUTO'}"),
            "value": round(job_samples / elapsed / 1e6, 2),
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak" if args.weak else "strong",       # the job (1024^2 x 64 spp) is fixed as N grows unless --weak
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{wl_desc}, {WIDTH}x{HEIGHT}, {spp} spp, MIS, min_depth 4 / max_depth 50",
                "hit_scene": accel_name,
                "samples_per_step": int(job_samples / args.steps),
                "vertices_per_sample": round(acc["vertices"] / max(acc["samples"], 1), 3),
                # the same figure from the device counters of the TIMED (overlapped) region, rank 0 / all devices of the process;
                # `samples` there are counted on the device where a path's radiance is written (PtStats.samples)
                "timed_region_vertices_per_sample": round(timed["vertices"] / max(timed["samples"], 1), 3),
                "timed_region_samples": int(timed["samples"]),
                "timed_region_counters_equal_steps_x_per_step": not any(p.startswith("timed region") for p in problems),
                "frame_equals_single_gpu": bool(frame_ok),
                "launch": {"single": "one process, one GPU", "multi": "one process, all GPUs (pt_multi_*)",
                           "dist": "one process per GPU (torch.distributed)"}[mode] +
                          (f" -- REHEARSAL: {world} contexts on ONE device, copies in place of ncclGather; not a measurement of {world} GPUs"
                           if args.shared_device else ""),
                "tiles": tiles,
            },
            "roofline": roof,
        }
        if multi_info is not None:
            # proof that N ranks took part: the communicator's own count and the library version, plus what a frame costs the host
            out["config"]["rccl"] = {"ncclCommCount": int(multi_info.comm_count), "version": int(multi_info.rccl_version),
                                     "host_threads": int(multi_info.threaded) * world,
                                     "exchange": "copies by the DMA engines (pt_multi_set_exchange)" if int(multi_info.exchange) else "ncclGather",
                                     "enqueue_us_per_frame_slowest_device": round(multi_info.enqueue_us_max, 1),
                                     "enqueue_us_per_frame_all_devices": round(multi_info.enqueue_us_sum, 1)}
        elif dist_path:
            out["config"]["rccl"] = {"world_size": dist.get_world_size(), "backend": args.backend,
                                     "version": ".".join(str(v) for v in torch.cuda.nccl.version()) if args.backend == "nccl" else None}
        if post_pass:
            roof["launch_times_from"] = ("3 extra steps after the timed region, launched strictly in order with HIP events around every launch "
                                         "(the timed steps are enqueued back to back without per-launch events: their launches overlap -- the next "
                                         "one fills the device while the last waves of this one run dry -- and have no duration of their own)")
            step_flops = alg_flops / 3.0                  # the level-0 launches of ONE step of this rank (single-process form: per device)
            roof["frac_of_timed_step"] = round(step_flops / (elapsed / args.steps) / 1e12 / VALU_PEAK_TFLOPS, 4)
            roof["frac_of_timed_step_note"] = ("algorithmic flops of this process's launches of one timed step / the step's wall time (resolve, "
                                               "gather and everything else included) / peak: what the overlap buys, against `frac` of a launch on its own")
        else:
            roof["launch_times_from"] = "HIP events around every path-kernel launch of the timed steps (--in-order: launches strictly one after the other)"
        if host_buffers:
            out["host_buffers"] = host_buffers
        if copy_run:
            out["config"]["exchange_copy"] = {
                "value": round(float(copy_run["timed"]["samples"]) / copy_run["elapsed"] / 1e6, 2), "unit": "Msamples/s",
                "ms_per_step": round(copy_run["elapsed"] / args.steps * 1e3, 3), "frame_equals_single_gpu": bool(copy_run["frame_ok"]),
                "what": "the same timed region with pt_multi_set_exchange(PT_EXCHANGE_COPY): one DMA copy per device in place of the "
                        "ncclGather (no kernel takes part in the exchange); `value` above is the ncclGather form"}
        if problems:
            out["verification_failed"] = problems
        if world == 1 and not args.no_cpu_baseline and not problems:
            out["cpu_baseline"] = cpu_baseline(pt, objs, WIDTH, HEIGHT, SPP, wl_desc.split(":")[0] + " scene")
        print(json.dumps(out), flush=True)
    ctx.close()
    if dist_path:
        dist.destroy_process_group()
    if problems:
        print("bench.py: the record does not verify: " + "; ".join(problems), file=sys.stderr, flush=True)
        sys.exit(1)


if __name__ == "__main__":
    main()
