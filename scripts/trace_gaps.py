"""Per-kernel durations and inter-kernel gaps from a rocprofv3 --kernel-trace CSV.
    python scripts/trace_gaps.py <dir-with-*_kernel_trace.csv> [kernel-substring]"""
import csv, glob, sys, collections
d = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "k_chol_level"
files = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)))
rows.sort()
agg = collections.defaultdict(lambda: [0, 0.0])
for s, e, n, gsz in rows:
    k = n.split("(")[0]
    agg[k][0] += 1; agg[k][1] += (e - s) / 1e3
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{c:7d} {t:12.1f} us  avg {t/c:8.2f}  {k}")
sel = [(s, e, gsz) for s, e, n, gsz in rows if pat in n]
if sel:
    # split into bursts (gap > 200 us)
    bursts = [[sel[0]]]
    for a, b in zip(sel, sel[1:]):
        if b[0] - a[1] > 200000: bursts.append([])
        bursts[-1].append(b)
    bl = bursts[-1]
    dur = [(e - s) / 1e3 for s, e, _ in bl]
    gap = [(b[0] - a[1]) / 1e3 for a, b in zip(bl, bl[1:])]
    print(f"last burst of {pat}: {len(bl)} launches, span {(bl[-1][1]-bl[0][0])/1e3:.1f} us, sum dur {sum(dur):.1f}, sum gap {sum(gap):.1f}")
    print("  dur  min/med/max", min(dur), sorted(dur)[len(dur)//2], max(dur))
    if gap: print("  gap  min/med/max", min(gap), sorted(gap)[len(gap)//2], max(gap))
    print("  first 12 (dur, grid):", [(round(x, 1), g // 256) for x, (_, _, g) in zip(dur[:12], bl[:12])])
    mid = len(bl) // 2
    print("  middle 8:", [(round(x, 1), g // 256) for x, (_, _, g) in zip(dur[mid:mid + 8], bl[mid:mid + 8])])
    print("  last 24:", [(round(x, 1), g // 256) for x, (_, _, g) in zip(dur[-24:], bl[-24:])])
