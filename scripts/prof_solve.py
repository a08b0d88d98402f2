"""Profiling driver: a few damped solves / LM iterations of BASELINE config 2 (for rocprofv3 --kernel-trace).
    python scripts/prof_solve.py [solves=3] [lm_iters=0]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynosam_amd import synth
from dynosam_amd.optimizer import Context, LevenbergMarquardtParams

n_solve = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n_lm = int(sys.argv[2]) if len(sys.argv) > 2 else 0
g = synth.make_hybrid_graph(synth.config(int(os.environ.get("CFG", "2"))))
ctx = Context()
ctx.set_profiling(True)
ctx.upload(g)
if os.environ.get("NOSPEC"):
    ctx.set_speculation(False)
for k in range(n_solve):
    t = time.perf_counter()
    d, dec = ctx.solve_damped(1e-5 * 10 ** k)
    print("solve", k, "ms", 1e3 * (time.perf_counter() - t), "lin decrease", dec, flush=True)
if n_lm:
    P = LevenbergMarquardtParams()
    P.max_iterations = n_lm
    P.relative_error_tol = 1e-300
    P.absolute_error_tol = 0.0
    ctx.set_values(g.var_state)
    ctx.optimize(P)
    ctx.reset_kernel_stats()
    ctx.set_values(g.var_state)
    t = time.perf_counter()
    r = ctx.optimize(P)
    dt = time.perf_counter() - t
    print("LM", r.iterations, r.inner_iterations, "ms/iter", 1e3 * dt / r.iterations, flush=True)
    for s in ctx.kernel_stats():
        print("   %-44s launches %6d total %9.3f ms  avg %8.2f us" % (s["name"], s["launches"], s["total_ms"], 1e3 * s["total_ms"] / max(1, s["launches"])), flush=True)
ctx.close()
