"""rocprofv3 --kernel-trace CSV -> the kernels in start order with start / end relative to the first one (us), queue id.
python tools/r04/trace_list.py <dir> [first [count]]"""
import csv, glob, os, sys
d = sys.argv[1]
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
count = int(sys.argv[3]) if len(sys.argv) > 3 else 40
f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
for r in rows[first:first + count]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{r['Kernel_Name'].split('(')[0][-40:]:42s} queue {r.get('Queue_Id','?'):>3s} start {s/1e3:10.1f} end {e/1e3:10.1f} dur {(e-s)/1e3:9.1f} us  grid {r.get('Grid_Size','?')}")
