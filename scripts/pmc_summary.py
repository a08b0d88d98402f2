"""Per-kernel HBM traffic from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs, as MI355X_MICROARCH.md
prescribes).  Units: the counters are in KiB... rocprofv3 reports FETCH_SIZE/WRITE_SIZE in kilobytes; on gfx950 FETCH_SIZE
reports half of the bytes of wide coalesced streaming reads (guide: double it before comparing with a byte count).
    python scripts/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write [out.txt]"""
import csv, glob, sys, collections
def load(d, name):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != name: continue
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
    return agg
fe, wr = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
cmd = sys.argv[4] if len(sys.argv) > 4 else "python scripts/prof_solve.py 3 0` (speculation off)`"
lines = [f"# HBM traffic per launch from rocprofv3 --pmc (separate FETCH_SIZE / WRITE_SIZE passes of `{cmd}`)",
         "# counter values are kilobytes; FETCH_SIZE x2 = the gfx950 correction for wide coalesced reads (MI355X_MICROARCH.md, HBM section); WRITE_SIZE uncalibrated",
         f"{'launches':>9} {'fetch_KB':>12} {'fetch_x2_KB':>12} {'write_KB':>12}  kernel"]
for k in sorted(fe, key=lambda k: -fe[k][1]):
    n, f = fe[k]
    w = wr.get(k, [1, 0.0])
    lines.append(f"{n:9d} {f/n:12.2f} {2*f/n:12.2f} {w[1]/max(1,w[0]):12.2f}  {k}")
txt = "\n".join(lines)
print(txt)
if len(sys.argv) > 3: open(sys.argv[3], "w").write(txt + "\n")
