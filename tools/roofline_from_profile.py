"""Summary of one tools/profile_workload.sh directory for bench.py's `roofline` object (profiles/rNN/roofline_<tag>.json):
the dominant kernel, its average duration from the kernel trace (timed dispatches = all but the first, which is the
warm-up step), HBM bytes per launch from the TCC passes (FETCH_SIZE KB x 1024 x 2: gfx950 reports half of wide coalesced
reads, MI355X_MICROARCH.md "HBM"; WRITE_SIZE KB x 1024), VALU instruction counts and mix, busy cycles.
usage: python tools/roofline_from_profile.py <dir> <tag>"""
import collections, csv, glob, json, os, sys

d, tag = sys.argv[1], sys.argv[2]
SIMDS = 1024


def short(k):
    return k.replace("void ", "").replace("(ptk::BounceArgs)", "").replace("(ptk::ResolveArgs)", "")


# kernel trace: durations per dispatch
dur = collections.defaultdict(list)
for f in glob.glob(f"{d}/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[short(r["Kernel_Name"])].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
if not dur:
    sys.exit("no kernel trace found")
tot = {k: sum(x[1] for x in v) for k, v in dur.items()}
dom = max(tot, key=tot.get)
# Since round 5 bench.py ends with one more render of the job -- alone, in order, by a context of its own: the reference its last
# timed frame is compared with -- which is the LAST dispatch of the dominant kernel in the process (and the first launch of a fresh
# context: slower).  It is no launch-time sample.
def verifies(path):
    try:
        return "frame_equals_single_gpu" in json.loads(open(path).read().strip().splitlines()[-1]).get("config", {})
    except Exception:
        return False
drop_last = verifies(f"{d}/bench_trace.json")
if drop_last:
    dur[dom] = sorted(dur[dom])[:-1]
# bench.py's timed steps overlap their launches (the next one fills the device while the last waves of this one run dry): a
# dispatch that waits for wave slots behind its predecessors has no duration of its own in the trace.  The launch time comes
# from the dispatches that ran ALONE (bench.py's in-order steps after the timed region; every dispatch with --in-order).
allv = sorted(dur[dom])
alone = [x for i, x in enumerate(allv) if all(j == i or y[0] + y[1] <= x[0] or x[0] + x[1] <= y[0] for j, y in enumerate(allv))]
if len(alone) == len(allv):
    timed = [x[1] for x in alone][1:] or [x[1] for x in allv]      # every launch in order (--in-order): all but the warm-up dispatch
else:
    timed = [x[1] for x in alone][-3:]                             # bench.py's three in-order steps after the overlapping timed steps
                                                                   # (the warm-up dispatch and the tail of the sequence also ran alone)
avg_ns = sum(timed) / len(timed)
overlapped = [x for x in allv if x not in alone]

# the --in-order run of the same command: every dispatch of the dominant kernel but the warm-up one
io = []
for f in glob.glob(f"{d}/trace_in_order/**/*kernel_trace.csv", recursive=True):
    rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in csv.DictReader(open(f)) if short(r["Kernel_Name"]) == dom)
    if verifies(f"{d}/bench_trace_in_order.json"):
        rows = rows[:-1]                                 # bench.py's verification render (see above)
    io += [x[1] for x in rows[len(rows) // 5:]]          # not the warm-up steps (5 of 25: clocks still ramping up)

# counters: per-dispatch average for the dominant kernel
cnt = collections.defaultdict(float)
nd = collections.defaultdict(set)
for f in glob.glob(f"{d}/pmc_*/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if short(r["Kernel_Name"]) != dom:
            continue
        cnt[r["Counter_Name"]] += float(r["Counter_Value"])
        nd[r["Counter_Name"]].add((f, r["Dispatch_Id"]))
c = {k: cnt[k] / len(nd[k]) for k in cnt}

out = {
    "tag": tag,
    "kernel": dom,
    "dispatches_timed": len(timed),
    "avg_launch_ms_kernel_trace": avg_ns / 1e6,
    "avg_launch_note": "in-order dispatches of the dominant kernel: bench.py's three event-timed steps after the timed region (all but the warm-up dispatch with --in-order)",
    "avg_launch_ms_in_order_run": (sum(io) / len(io) / 1e6) if io else None,
    "in_order_run_note": "trace_in_order/: the same command with --in-order (every launch on its own): the dispatches of its 20 timed steps; the average of the --stats summary there is a launch time",
    "overlapping_dispatches": len(overlapped),
    "overlapping_dispatches_period_ms": ((max(x[0] + x[1] for x in overlapped) - min(x[0] for x in overlapped)) / len(overlapped) / 1e6) if overlapped else None,
    "overlapping_dispatches_avg_trace_duration_ms": (sum(x[1] for x in overlapped) / len(overlapped) / 1e6) if overlapped else None,
    "all_kernels_avg_ms": {k: sum(x[1] for x in v) / len(v) / 1e6 for k, v in dur.items()},
    "counters_per_launch": c,
    "source": f"rocprofv3 --kernel-trace --stats and one --pmc pass per run of `python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline ...` (tools/profile_workload.sh {tag})",
}
if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
    out["fetch_bytes_per_launch"] = c["FETCH_SIZE"] * 1024 * 2
    out["write_bytes_per_launch"] = c["WRITE_SIZE"] * 1024
    out["hbm_bytes_per_launch"] = out["fetch_bytes_per_launch"] + out["write_bytes_per_launch"]
    out["hbm_note"] = "FETCH_SIZE (KB) x 1024 x 2 (gfx950 counts half of wide coalesced reads) + WRITE_SIZE (KB) x 1024"
if "SQ_INSTS_VALU" in c and "GRBM_GUI_ACTIVE" in c:
    cyc = c["GRBM_GUI_ACTIVE"] / 8.0                                   # the counter sums the 8 XCDs
    out["gpu_cycles_per_launch"] = cyc
    out["valu_insts_per_launch"] = c["SQ_INSTS_VALU"]
    out["cycles_per_valu_inst_per_simd"] = cyc * SIMDS / c["SQ_INSTS_VALU"]
    out["valu_issue_frac"] = 2.0 / out["cycles_per_valu_inst_per_simd"]   # against 2 cycles per wave64 VALU instruction (MI355X_MICROARCH.md)
    out["clock_ghz_profiled"] = cyc / avg_ns if avg_ns else None
if "SQ_THREAD_CYCLES_VALU" in c and "SQ_ACTIVE_INST_VALU" in c:
    out["lane_utilisation"] = c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"]) if c["SQ_ACTIVE_INST_VALU"] else None
if "SQ_LDS_BANK_CONFLICT" in c and c.get("SQ_LDS_IDX_ACTIVE"):
    out["lds_bank_conflict_frac"] = c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]
    out["lds_note"] = "SQ_LDS_BANK_CONFLICT (extra cycles) / SQ_LDS_IDX_ACTIVE (all LDS-array cycles), MI355X_MICROARCH.md"
mix = {k[len("SQ_INSTS_VALU_"):]: v for k, v in c.items() if k.startswith("SQ_INSTS_VALU_")}
if mix:
    out["valu_mix_per_launch"] = mix
json.dump(out, sys.stdout, indent=1)
