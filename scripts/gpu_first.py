"""First end-to-end GPU check: linearisation parity, one damped solve, full LM vs the oracle."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dynosam_amd import synth
from dynosam_amd.optimizer import Context, LevenbergMarquardtParams
from oracle import oracle_py as O

CASES = {"tiny": (1, dict(frames=12, static_points=60, dynamic_points_per_object=20)), "cfg1": (1, {}), "cfg2": (2, {})}
for cfgn, kw in [CASES[a] for a in sys.argv[1:]]:
    g = synth.make_hybrid_graph(synth.config(cfgn, **kw))
    print(f"--- cfg {cfgn} {kw}: vars {g.n_vars} factors {g.n_factors}", flush=True)
    og = O.OracleGraph(g)
    ctx = Context()
ctx.set_profiling(True)
    t = time.time(); ctx.upload(g); print("upload s", time.time() - t, "n", ctx.graph.n_vars, flush=True)
    e_gpu, e_ref = ctx.error(), og.error()
    print("error", e_gpu, e_ref, abs(e_gpu - e_ref) / e_ref, flush=True)
    J, b, e = ctx.linearize()
    Jr, br, er = og.linearize()
    print("max|J-Jref|", np.abs(J - Jr).max(), "rel", np.abs(J - Jr).max() / np.abs(Jr).max(), "max|b-bref|", np.abs(b - br).max(), "max|e-eref|", np.abs(e - er).max(), flush=True)
    d, dec = ctx.solve_damped(1e-5)
    bad, dr, decr = og.solve_damped(1e-5)
    print("delta max diff", np.abs(d - dr).max(), "max", np.abs(dr).max(), "lin decrease", dec, decr, flush=True)
    O.set_threads(min(8, len(os.sched_getaffinity(0))))
    lim = 3 if g.n_factors > 20000 else 0
    P = LevenbergMarquardtParams()
    if lim: P.max_iterations = lim
    t = time.time(); r = ctx.optimize(P); tg = time.time() - t
    t = time.time(); rr, _ = og.optimize(P); tc = time.time() - t
    print("trace gpu", [(r.trace_lambda[i], r.trace_error[i], r.trace_accepted[i]) for i in range(r.trace_len)][:6], flush=True)
    print("trace cpu", [(rr.trace_lambda[i], rr.trace_error[i], rr.trace_accepted[i]) for i in range(rr.trace_len)][:6], flush=True)
    print("GPU LM: iters", r.iterations, r.inner_iterations, "err", r.error_before, "->", r.error_after, "time", tg, r.solve_seconds, flush=True)
    print("CPU LM: iters", rr.iterations, rr.inner_iterations, "err", rr.error_before, "->", rr.error_after, "time", tc, flush=True)
    print("rel final cost diff", abs(r.error_after - rr.error_after) / rr.error_after, flush=True)
    v = ctx.values(); vr = og.state()
    print("max |values diff|", np.abs(v - vr).max(), flush=True)
    for s in ctx.kernel_stats():
        print("   %-20s launches %6d total %9.3f ms  avg %8.2f us" % (s["name"], s["launches"], s["total_ms"], 1e3 * s["total_ms"] / max(1, s["launches"])), flush=True)
    if lim:
        ctx.reset_kernel_stats()
        ctx.set_values(g.var_state)
        t = time.time(); r = ctx.optimize(); tg = time.time() - t
        print("FULL GPU LM: iters", r.iterations, r.inner_iterations, "err", r.error_before, "->", r.error_after, "time", tg, "iters/s", r.iterations / tg, flush=True)
        for s in ctx.kernel_stats():
            print("   %-20s launches %6d total %9.3f ms  avg %8.2f us" % (s["name"], s["launches"], s["total_ms"], 1e3 * s["total_ms"] / max(1, s["launches"])), flush=True)
    ctx.close()
