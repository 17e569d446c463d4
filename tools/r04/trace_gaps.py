"""Kernel trace of tools/r04/share_trace.py (rocprofv3 --kernel-trace CSV) -> per render: duration of every kernel and the
idle time between consecutive kernels.  python tools/r04/trace_gaps.py <dir with *_kernel_trace.csv> [skip_first_renders]"""
import csv, glob, os, sys
d = sys.argv[1]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 3
f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ev = [(r["Kernel_Name"].split("(")[0][:60], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
# a render = everything from one path kernel to the next
starts = [i for i, e in enumerate(ev) if "k_paths" in e[0]]
acc = {}
n = 0
for a, b in zip(starts[skip:-1], starts[skip + 1:]):
    seg = ev[a:b + 1]
    n += 1
    for k in range(len(seg) - 1):
        name, s, e = seg[k]
        acc.setdefault(("run", k, name), []).append(e - s)
        acc.setdefault(("gap", k, name + " -> " + seg[k + 1][0]), []).append(seg[k + 1][1] - e)
    acc.setdefault(("period", 0, "path kernel start -> next path kernel start"), []).append(seg[-1][1] - seg[0][1])
print(f"{f}: {n} renders")
for key in sorted(acc, key=lambda k: (k[1], k[0] != "run")):
    v = acc[key]
    print(f"  {key[0]:6s} {key[2]:100s} mean {sum(v) / len(v) / 1e3:9.2f} us  (min {min(v) / 1e3:.2f}, max {max(v) / 1e3:.2f}, n {len(v)})")
