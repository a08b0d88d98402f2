"""Counters of every launch of one kernel in dispatch order, for the LAST n launches (rocprofv3 --pmc csv dirs).
    python scripts/pmc_per_dispatch.py k_chol_level 46 dirA dirB ..."""
import csv, glob, sys, collections
kern, n, dirs = sys.argv[1], int(sys.argv[2]), sys.argv[3:]
cols = collections.OrderedDict()
grid = {}
for d in dirs:
    per = collections.defaultdict(dict)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if kern not in r["Kernel_Name"]: continue
            did = int(r["Dispatch_Id"])
            per[did][r["Counter_Name"]] = per[did].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            grid[did] = int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0) // 256
    ids = sorted(per)[-n:]
    for j, did in enumerate(ids):
        for c, v in per[did].items(): cols.setdefault(c, {})[j] = v
        cols.setdefault("workgroups", {})[j] = grid[did]
names = ["workgroups"] + [c for c in cols if c != "workgroups"]
print("launch " + " ".join(f"{c[-24:]:>24}" for c in names))
for j in range(n):
    print(f"{j:6d} " + " ".join(f"{cols[c].get(j, float('nan')):24.0f}" for c in names))
