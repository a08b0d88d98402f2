"""Per-launch durations of k_chol_level / k_back_group of the last solve in a rocprofv3 kernel trace (csv)."""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)))
rows.sort()
chol = [(s, e, g) for s, e, n, g in rows if "k_chol_level" in n]
n = int(sys.argv[2])
last = chol[-n:]
print("launch: duration_us gap_us workgroups")
for i, (s, e, g) in enumerate(last):
    gap = (s - last[i - 1][1]) / 1e3 if i else 0.0
    print(f"{i:4d}: {(e - s) / 1e3:7.2f} {gap:6.2f} {g // 256}")
print("total us", (last[-1][1] - last[0][0]) / 1e3, "sum dur", sum(e - s for s, e, g in last) / 1e3)
