import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dynosam_amd import synth
from dynosam_amd.optimizer import Context, LevenbergMarquardtParams
g = synth.make_hybrid_graph(synth.config(1, frames=40, objects=2, static_points=400, dynamic_points_per_object=60, seed=21))
c = Context(); c.upload(g)
r0 = c.optimize(); print("plain", r0.iterations, r0.error_after)
for thr in (1e-3, 1e-2, 3e-2, 0.1, 0.3):
    P = LevenbergMarquardtParams(); P.relinearize_threshold = thr
    c.set_values(g.var_state)
    r = c.optimize(P)
    tot = r.factors_linearized + r.factors_reused
    print(thr, "iters", r.iterations, "err", r.error_after, "rel", abs(r.error_after - r0.error_after) / r0.error_after, "reused frac", r.factors_reused / tot, "vars relin", r.variables_relinearized)
