import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynosam_amd import synth
from dynosam_amd.optimizer import Context, LevenbergMarquardtParams
g = synth.make_hybrid_graph(synth.config(int(os.environ.get("CFG", "2"))))
ctx = Context(); ctx.upload(g)
if os.environ.get("NOGRAPH"): ctx.set_graphs(False)
P = LevenbergMarquardtParams(); P.max_iterations = int(os.environ.get("ITERS", "6")); P.relative_error_tol = 1e-300; P.absolute_error_tol = 0.0
ctx.optimize(P)
ctx.set_values(g.var_state)
P.verbosity = 2
ctx.optimize(P)
