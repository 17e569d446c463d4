"""Per-kernel, per-counter AVERAGE PER DISPATCH from rocprofv3 counter_collection.csv files.
usage: python tools/pmc_sum.py <dir> [...]   (json on stdout)"""
import collections, csv, glob, json, sys
tot = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(lambda: collections.defaultdict(set))
for d in sys.argv[1:]:
    for f in glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "k_paths" not in k and "k_resolve" not in k and "k_bounce" not in k:
                continue
            k = k.replace("void ", "").replace("(ptk::BounceArgs)", "").replace("(ptk::ResolveArgs)", "")
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[k][r["Counter_Name"]].add((f, r["Dispatch_Id"]))
out = {k: {c: {"per_dispatch": tot[k][c] / len(disp[k][c]), "dispatches": len(disp[k][c])} for c in sorted(tot[k])}
       for k in tot}
json.dump(out, sys.stdout, indent=1)
