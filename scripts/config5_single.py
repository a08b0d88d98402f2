"""BASELINE config 5 (2 000 frames, 50 objects, 1.97 M factors) on ONE GPU: upload time, LM iteration time, cost trace."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynosam_amd import synth
from dynosam_amd.optimizer import Context, LevenbergMarquardtParams
t = time.perf_counter(); g = synth.make_hybrid_graph(synth.config(5)); print("generate s", time.perf_counter() - t, g.n_factors, g.n_vars, flush=True)
ctx = Context()
t = time.perf_counter(); ctx.upload(g); print("upload s", time.perf_counter() - t, flush=True)
P = LevenbergMarquardtParams(); P.max_iterations = int(sys.argv[1]) if len(sys.argv) > 1 else 5; P.relative_error_tol = 1e-300; P.absolute_error_tol = 0.0
t = time.perf_counter(); r = ctx.optimize(P); dt = time.perf_counter() - t
print(f"LM {r.iterations} it / {r.inner_iterations} solves: {1e3*dt/max(1,r.iterations):.2f} ms/iter, error {r.error_before:.6g} -> {r.error_after:.6g}", flush=True)
ctx.set_values(g.var_state)
t = time.perf_counter(); r = ctx.optimize(P); dt = time.perf_counter() - t
print(f"second run: {1e3*dt/max(1,r.iterations):.2f} ms/iter", flush=True)
print("trace", [(float(r.trace_lambda[i]), float(r.trace_error[i]), int(r.trace_accepted[i])) for i in range(r.trace_len)])
ctx.close()
