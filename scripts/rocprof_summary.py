"""Turn a rocprofv3 (rocpd sqlite) kernel trace into the text summary kept under profiles/.

    python scripts/rocprof_summary.py gpurun_out/prof/r1/bench_results.db profiles/r01_kernel_stats.txt "<command>"
"""
import sqlite3
import sys

db, out, cmd = sys.argv[1], sys.argv[2], (sys.argv[3] if len(sys.argv) > 3 else "")
c = sqlite3.connect(db)
rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
with open(out, "w") as f:
    f.write(f"# rocprofv3 --kernel-trace --stats summary (top_kernels view of the rocpd database)\n# command: {cmd}\n")
    f.write("# durations in microseconds\n")
    f.write(f"{'calls':>8} {'total_us':>14} {'avg_us':>10} {'pct':>7}  kernel\n")
    for name, calls, tot, avg, pct in rows:
        f.write(f"{calls:8d} {tot:14.3f} {avg:10.3f} {pct:7.2f}  {name}\n")
print(open(out).read()[:1500])
