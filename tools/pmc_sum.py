"""Sum rocprofv3 counter_collection.csv files per kernel: python tools/pmc_sum.py <dir> [...]"""
import collections, csv, glob, json, sys
out = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for d in sys.argv[1:]:
    for f in glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "ptk::" not in k:
                continue
            k = k.replace("void ptk::", "").replace("(ptk::BounceArgs)", "").replace("(ptk::ResolveArgs)", "")
            out[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[k].add((f, r["Dispatch_Id"]))
for k, v in out.items():
    print(k, "dispatches/pass", len(disp[k]) // max(1, len(set(f for f, _ in disp[k]))))
    for c, x in sorted(v.items()):
        print(f"   {c:28s} {x:.4g}")
