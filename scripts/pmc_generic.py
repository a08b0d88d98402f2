"""Per-kernel sums of arbitrary rocprofv3 --pmc counters (csv output dirs given on the command line), averaged per launch.
    python scripts/pmc_generic.py k_chol_level dirA dirB ...  [--out file]"""
import csv, glob, sys, collections
args = [a for a in sys.argv[1:] if not a.startswith("--out")]
out = next((sys.argv[i + 1] for i, a in enumerate(sys.argv) if a == "--out"), None)
kern, dirs = args[0], args[1:]
agg = collections.defaultdict(lambda: [0, 0.0])
for d in dirs:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if kern not in r["Kernel_Name"]: continue
            a = agg[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
lines = [f"# rocprofv3 --pmc counters of {kern} (python scripts/prof_solve.py 3 0; one pass per counter group), per launch", f"{'launches':>9} {'mean_per_launch':>18}  counter"]
for k in sorted(agg): lines.append(f"{agg[k][0]:9d} {agg[k][1] / max(1, agg[k][0]):18.1f}  {k}")
txt = "\n".join(lines); print(txt)
if out: open(out, "w").write(txt + "\n")
