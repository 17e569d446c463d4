"""Steady-state summary of a tools/r04/frames_trace.py kernel trace: the last `frames` regenerating launches -- period between their
ends, time from a launch's end to the start / end of its resolve, and for every gather kernel how long it sat between becoming
ready (end of its frame's resolve) and ending.   python tools/r04/frames_summary.py DIR frames"""
import csv, glob, sys
d, frames = sys.argv[1], int(sys.argv[2])
f = glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
def sel(key): return [(int(r["Start_Timestamp"]) / 1e3, int(r["End_Timestamp"]) / 1e3) for r in rows if key in r["Kernel_Name"]]
regen, res, rccl, unp = sel("k_paths_regen")[-frames:], sel("k_resolve")[-frames:], sel("rccl")[-frames:], sel("k_film_unpack")[-frames:]
t0 = regen[0][0]
ends = sorted(e for _, e in regen)
per = [b - a for a, b in zip(ends, ends[1:])]
mid = per[len(per) // 4: -len(per) // 4] or per
print(f"{len(regen)} launches; period between launch ends: mean of the middle half {sum(mid) / len(mid):.1f} us (min {min(per):.1f}, max {max(per):.1f})")
print(f"whole sequence: first launch start to last resolve end {(max(e for _, e in res) - t0):.1f} us = {(max(e for _, e in res) - t0) / len(regen):.1f} per frame")
rs = sorted(res)
lag = [s - e for e, (s, _) in zip(ends, rs)]
print(f"launch end -> resolve start: mean {sum(lag) / len(lag):.1f} us, max {max(lag):.1f}; resolve duration mean {sum(e - s for s, e in rs) / len(rs):.1f}, max {max(e - s for s, e in rs):.1f}")
if rccl:
    rc = sorted(rccl)
    wait = [e - s for s, e in rc]
    print("gather kernel start -> end (us): " + " ".join(f"{w:.0f}" for w in wait))
    late = [ge - re for (_, re), (_, ge) in zip(rs, rc)]
    print("resolve end -> gather end (us):  " + " ".join(f"{w:.0f}" for w in late))
print("launch start / end (us from the first start), in start order:")
for s, e in sorted(regen): print(f"  {s - t0:9.1f} {e - t0:9.1f}  ({e - s:8.1f})")
