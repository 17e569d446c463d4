#!/usr/bin/env python3
"""bench.py -- Msamples/s of the rendering hot path on N MI355X of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
        --master-port P bench.py --gpus N --steps K --warmup W

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
}
BYTES_PER_VERTEX = 252      # extend 32 + shade 144 + shadow/accumulate 68 + compaction 8 (a five-kernel pipeline;
BYTES_PER_SAMPLE = 64       # the fused kernel never makes those round trips: informational only)
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s achievable float4 copy)
VALU_PEAK_TFLOPS = 157.3    # MI355X_MICROARCH.md: peak FP32 vector (= the f32 MFMA rate)
F_SPHERE, F_TRIANGLE, F_SHADE = 23, 51, 200     # flops per primitive test / per shaded vertex (Lambert)
PROFILE_DIR = os.path.join(ROOT, "profiles", "r02")


def cpu_baseline(pt, objs):
    """Oracle (kind "port") on the host: C2 at the bench camera, a row subset spread over the image.
    Threads = the box's CPU share for one GPU (16), never more than the affinity mask allows.  A short
    probe sizes the sample so that the timed run is ~10 s of wall time (bounded)."""
    from oracle import orc
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 16))
    cam = pt.camera_new(width=WIDTH, height=HEIGHT)

    def run(spp, band_count):
        prm = pt.default_params(spp=spp, band_rows=1, band_index=0, band_count=band_count)
        t0 = time.perf_counter()
        lin, _, _ = orc.render(cam, objs, prm, orc.F64, orc.RECURSIVE, threads=cores)
        return lin.shape[0], time.perf_counter() - t0

    rows, dt = run(1, 16)                                   # probe: 64 rows x 1 spp
    rate = rows * WIDTH / dt                                # samples / s
    # timed run: full 64 spp on every band_count-th row, band_count chosen for ~10 s of wall time
    band_count = 16
    for bc in (8, 4, 2, 1):
        if (HEIGHT // bc) * WIDTH * SPP / rate <= 12.0:
            band_count = bc
    spp = SPP if (HEIGHT // band_count) * WIDTH * SPP / rate <= 30.0 else int(max(1, 30.0 * rate / (64 * WIDTH)))
    rows, dt = run(spp, band_count)
    samples = rows * WIDTH * spp
    # BASELINE.json configs[0]: the reference's own scene, 256 x 256, 4 spp, ONE thread (the scalar port)
    c1_objs = pt.builtin_scene(1)
    c1_cam = pt.camera_new(width=256, height=256)
    t0 = time.perf_counter()
    orc.render(c1_cam, c1_objs, pt.default_params(spp=4), orc.F64, orc.RECURSIVE, threads=1)
    c1_dt = time.perf_counter() - t0
    return {
        "value": round(samples / dt / 1e6, 4),
        "unit": "Msamples/s",
        "cores": cores,
        "kind": "port",
        "sample": f"C2 scene, {WIDTH}x{HEIGHT} camera, every {band_count}th row ({rows} rows), {spp} of {SPP} spp = "
                  f"{samples} samples in {dt:.2f} s wall; oracle f64 recursive (reference-shaped: linear scan, "
                  f"3 scans per vertex), {cores} threads over pixels ({avail} CPUs visible)",
        "config0_single_thread": {"value": round(256 * 256 * 4 / c1_dt / 1e6, 4), "unit": "Msamples/s", "cores": 1,
                                  "sample": f"C1 reference scene, 256x256, 4 spp = 262144 samples in {c1_dt:.2f} s"},
    }


def profile_summary(world, workload, accel):
    """Counter-derived figures of the dominant kernel from the committed rocprofv3 passes of THIS command
    (profiles/r02/roofline_<workload>[_bvh].json, written by tools/profile_workload.sh on the GPU box).  PMC counters
    cannot be read from inside this process, so these are the last profiled values, tagged with their source file;
    None when no profile of this exact workload exists (or N > 1)."""
    if world != 1:
        return None
    name = f"roofline_{workload}{'_bvh' if accel == 1 else ''}.json"
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


def main():
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
    ap.add_argument("--level0-form", type=int, default=0, help="PtTuning.level0_form (0 = library default, 1 = queue form, 2 = regenerating form)")
    ap.add_argument("--regen-workgroups", type=int, default=0, help="PtTuning.regen_workgroups (0 = library default)")
    ap.add_argument("--weak", action="store_true", help="N > 1: weak scaling (64*N spp) instead of the strong-scaling default")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo = rehearsal on a box with fewer GPUs than ranks (ranks share devices, the gather "
                         "goes through host memory); the driver's runs use nccl (RCCL)")
    ap.add_argument("--force-dist", action="store_true",
                    help="with one rank: still initialise the process group and run every collective of the N > 1 path "
                         "(rehearsal of that code over real RCCL on a one-GPU box)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import pathtrace_amd as pt
    from pathtrace_amd.dist import FilmGather, default_band_rows

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N > 1 with torch.distributed.run (one process per GPU)")
        args.gpus = world
    dev_index = local_rank if args.backend == "nccl" else local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    multi = world > 1 or args.force_dist        # run the distributed code path
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    comm_dev = dev if args.backend == "nccl" else torch.device("cpu")

    global WIDTH, HEIGHT, SPP
    scene_id, scene_arg, WIDTH, HEIGHT, SPP, wl_desc = WORKLOADS[args.workload]
    if args.workload != "c2":
        args.no_cpu_baseline = True          # the CPU baseline leg is defined on the headline config
    objs = pt.builtin_scene(scene_id, scene_arg)
    cam = pt.camera_new(width=WIDTH, height=HEIGHT)
    spp = SPP * world if args.weak else SPP
    band_rows = default_band_rows(HEIGHT, world) if multi else 0
    prm = pt.default_params(spp=spp, band_rows=band_rows, band_index=rank, band_count=world, profile=1,
                            max_paths_in_flight=args.max_paths, workgroups=args.workgroups, accel=args.accel)
    ctx = pt.Context(dev_index)
    ctx.upload(objs)
    ctx.set_tuning(cont_workgroups=args.cont_workgroups, export_below=args.export_below, level0_form=args.level0_form,
                   regen_workgroups=args.regen_workgroups)
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    rows = pt.tile_rows(HEIGHT, band_rows, rank, world)
    lin = torch.empty((rows, WIDTH, 3), dtype=torch.float32, device=dev)
    rgba = torch.empty((rows, WIDTH, 4), dtype=torch.uint8, device=dev)

    acc = {"vertices": 0, "samples": 0, "bounce_ms": 0.0, "launches": 0, "total_ms": 0.0,
           "p_vertices": 0, "p_ms": 0.0, "p_launches": 0, "shadow_rays": 0}

    # the single exchange step of the path: one gather of the framebuffer (f32 + RGBA8 packed) per step.  It is
    # launched asynchronously, so the gather of step k runs (on the backend's stream) while step k + 1 renders; the
    # last one is completed inside the timed region.
    film_gather = FilmGather(HEIGHT, WIDTH, band_rows, rank, world, comm_dev, always_collective=True) if multi else None

    def step(record):
        ctx.render_into(cam, prm, lin.data_ptr(), rgba.data_ptr())
        ctx.sync()
        if multi:
            film_gather.start(lin.to(comm_dev), rgba.to(comm_dev))
        if record:
            st = ctx.stats()
            acc["vertices"] += st.vertices
            acc["shadow_rays"] += st.shadow_rays
            acc["samples"] += st.samples
            acc["bounce_ms"] += st.bounce_kernel_ms
            acc["launches"] += st.bounce_launches
            acc["total_ms"] += st.total_ms
            acc["p_vertices"] += st.primary_vertices
            acc["p_ms"] += st.primary_kernel_ms
            acc["p_launches"] += st.primary_launches

    def barrier():
        if multi:
            dist.barrier()

    frame = frame8 = None
    for _ in range(args.warmup):
        step(False)
    if multi:
        film_gather.finish()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    if multi:
        frame, frame8 = film_gather.finish()      # the last frame; earlier ones were completed by the next start()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if multi:
        t = torch.tensor([elapsed], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tot = torch.tensor([acc["vertices"], acc["samples"]], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        job_samples = float(tot[1].item())
    else:
        job_samples = float(acc["samples"])

    if rank == 0:
        if multi:
            assert frame is not None and tuple(frame.shape) == (HEIGHT, WIDTH, 3) and torch.isfinite(frame).all()
            if world == 1:
                assert torch.equal(frame, lin) and torch.equal(frame8, rgba)     # --force-dist: the gathered frame is the tile
        else:
            assert torch.isfinite(lin).all()
        # the dominant kernel = the level-0 launch of each batch (camera rays + every bounce until its waves hand
        # their sparse tails over); it processes p_vertices of the vertices and all of the camera samples
        n_sph = sum(1 for o in objs if o.shape_tag == 0)
        n_tri = len(objs) - n_sph
        f_scan = F_SPHERE * n_sph + F_TRIANGLE * n_tri                       # one linear scan of the scene
        share = acc["p_vertices"] / max(acc["vertices"], 1)                  # the level-0 launches' share of the work
        scans = acc["p_vertices"] + acc["shadow_rays"] * share               # closest-hit scans + visibility scans
        alg_flops = f_scan * scans + F_SHADE * acc["p_vertices"]
        achieved_tf = alg_flops / (acc["p_ms"] * 1e-3) / 1e12 if acc["p_ms"] > 0 else 0.0
        alg_bytes = BYTES_PER_VERTEX * acc["p_vertices"] + BYTES_PER_SAMPLE * acc["samples"]
        avg_ms = acc["p_ms"] / max(acc["p_launches"], 1)
        prof = profile_summary(world, args.workload, args.accel)
        accel_name = {0: "linear scan (reference)", 1: "BVH traversal (accel=1, same film as the linear scan)",
                      2: "PT_ACCEL_AUTO (product default: BVH above ~512 sphere tests per scan, same film)"}[args.accel]
        diffuse = all(o.mat_tag in (0, 1) for o in objs)
        if args.accel == 1 or (args.accel == 2 and len(objs) > 512):
            kernel = "k_paths_bvh<MIS, OVF=false%s>" % (", DIFFUSE" if diffuse else "")
        elif len(objs) <= 128:
            # large batches over a scene in LDS: the regenerating form where the library takes it (pt_api.cpp: diffuse scenes
            # by default, PtTuning.level0_form), the queue form otherwise
            big = acc["samples"] / max(acc["p_launches"], 1) > (1 << 22)
            if big and (args.level0_form == 2 or (args.level0_form == 0 and diffuse)):
                kernel = "k_paths_regen<MIS, %s>" % ("DIFFUSE" if diffuse else "generic")
            else:
                kernel = "k_paths<kModeLds, MIS, OVF=false%s>" % (", DIFFUSE" if diffuse else "")
        else:
            kernel = "k_paths<kModeTiled, MIS, OVF=false>"
        roof = {
            "kernel": kernel + (": the one path-kernel launch of a sample batch (camera rays + every bounce of every path; a lane "
                                "whose path ends takes the batch's next one)" if kernel.startswith("k_paths_regen") else
                                ": the level-0 launch of a sample batch (camera rays + every bounce until the waves hand "
                                "over their sparse tails)") +
                      ", rank 0.  Bound by f32 VALU issue (no contraction on this path: the "
                      "schema's mfma slot does not apply; same 157.3 TFLOP/s f32 peak)",
            "bound": "valu",
            "achieved": round(achieved_tf, 2),
            "peak": VALU_PEAK_TFLOPS,
            "unit": "TFLOP/s",
            "frac": round(achieved_tf / VALU_PEAK_TFLOPS, 4),
            "algorithmic_flops_per_launch": round(alg_flops / max(acc["p_launches"], 1)),
            "flops_model": f"{f_scan} per scan of the scene ({n_sph} spheres x {F_SPHERE} + {n_tri} triangles x {F_TRIANGLE}) x "
                           f"(vertices + visibility scans) + {F_SHADE} per vertex (SURVEY 8d)",
            "avg_launch_ms": round(avg_ms, 4),
            "launches": acc["p_launches"],
            "vertex_share": round(share, 4),
            "all_path_kernels_ms_per_step": round(acc["bounce_ms"] / max(args.steps, 1), 4),
            "traffic": None, "hbm_frac": None, "valu_issue_frac": None,
            "algorithmic_bytes_per_launch_unfused_pipeline": round(alg_bytes / max(acc["p_launches"], 1)),
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
        out = {
            "metric": ("Msamples/sec (pixels x spp / s) at 1024^2/64spp" if args.workload == "c2"
                       else f"Msamples/sec (pixels x spp / s) at {WIDTH}x{HEIGHT}/{SPP}spp") +
                      ("" if args.accel == 0 else f", hit_scene = {'BVH' if args.accel == 1 else 'PT_ACCEL_AUTO'}"),
            "value": round(job_samples / elapsed / 1e6, 2),
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak" if (args.weak or world == 1) else "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{wl_desc}, {WIDTH}x{HEIGHT}, {spp} spp, MIS, min_depth 4 / max_depth 50",
                "hit_scene": accel_name,
                "samples_per_step": int(job_samples / args.steps),
                "vertices_per_sample": round(acc["vertices"] / max(acc["samples"], 1), 3),
                "tiles": "whole image" if world == 1 else f"interleaved bands of {band_rows} rows over {world} ranks, "
                                                          f"ONE {args.backend} gather of the packed f32 + RGBA8 frame to rank 0 per step, overlapped with the next step's rendering",
            },
            "roofline": roof,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(pt, objs)
        print(json.dumps(out), flush=True)
    ctx.close()
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
