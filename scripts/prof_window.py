import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynosam_amd import synth
from dynosam_amd.optimizer import Context, LevenbergMarquardtParams
g = synth.make_hybrid_graph(synth.config(2, frames=21, static_points=40 * 21, dynamic_points_per_object=2 * 21))
print(g.n_factors, g.n_vars)
ctx = Context()
ctx.set_profiling(True)
t = time.perf_counter(); ctx.upload(g); print("upload ms", 1e3 * (time.perf_counter() - t))
for rep in range(3):
    ctx.set_values(g.var_state)
    ctx.reset_kernel_stats()
    t = time.perf_counter(); r = ctx.optimize(); dt = time.perf_counter() - t
    print(f"optimize {rep}: {1e3*dt:.2f} ms, {r.iterations} it {r.inner_iterations} solves -> {1e3*dt/r.iterations:.3f} ms/it; solve_seconds {1e3*r.solve_seconds:.2f}")
for s in ctx.kernel_stats():
    print("   %-44s launches %6d total %9.3f ms  avg %8.2f us" % (s["name"], s["launches"], s["total_ms"], 1e3 * s["total_ms"] / max(1, s["launches"])))
if os.environ.get("NOGRAPH"):
    ctx.set_graphs(False)
    ctx.set_values(g.var_state)
    t = time.perf_counter(); r = ctx.optimize(); dt = time.perf_counter() - t
    print(f"no graphs: {1e3*dt:.2f} ms, {1e3*dt/r.iterations:.3f} ms/it")
P = LevenbergMarquardtParams(); P.verbosity = 2; P.max_iterations = 3
ctx.set_values(g.var_state); ctx.optimize(P)
