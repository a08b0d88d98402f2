"""Timeline of an LM run from a rocprofv3 kernel trace: per solve (k_fold_flags ends it) and per linearisation."""
import csv, glob, sys
d = sys.argv[1]
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
rows.sort()
t0 = rows[0][0]
ev = []
for s, e, n, q, st in rows:
    if "k_linearize<2" in n: ev.append((s, f"LIN start  q={q} st={st}"))
    if "k_fold_flags" in n: ev.append((e, f"  solve END q={q} st={st}"))
    if "k_point" in n: ev.append((s, f"  solve BEGIN q={q} st={st}"))
    if "k_back_level" in n: pass
last = None
for t, m in ev[-60:]:
    print(f"{(t - t0) / 1e3:12.1f} us  {m}" + (f"   (+{(t - last) / 1e3:.1f})" if last else ""))
    last = t
