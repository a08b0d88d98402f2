"""diagnostic: GPU vs oracle optimum, difference by variable class (poses / points) on config 2, stereo-static config 1"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dynosam_amd import synth
from dynosam_amd.optimizer import Context
from oracle import oracle_py as O
O.lib(); O.set_threads(8)
for name, g in (("config2", synth.make_hybrid_graph(synth.config(2))), ("stereo", synth.to_stereo_static(synth.make_hybrid_graph(synth.config(1)), behind=5))):
    og = O.OracleGraph(g); ro, _ = og.optimize()
    c = Context(); c.upload(g); r = c.optimize()
    v, vo = c.values(), og.state()
    d = np.abs(v - vo)
    pose = g.var_type == 0
    print(name, "cost rel", abs(r.error_after - ro.error_after) / ro.error_after, "pose max", d[pose].max(), "point max", d[~pose].max())
    dp = d[~pose][:, :3].max(axis=1)
    idx = np.argsort(-dp)[:5]
    pts = vo[~pose][:, :3]
    X0 = vo[pose][0]
    print("  worst points: diff", dp[idx], "norm of point", np.linalg.norm(pts[idx], axis=1))
    print("  quantiles of point diff", np.quantile(dp, [0.5, 0.9, 0.99, 0.999]))
    # re-linearise both at their own optimum: gradient norm? use damped solve step size as convergence measure
    dd, _ = c.solve_damped(1e-5)
    print("  GPU Gauss-Newton step at its optimum: pose max", np.abs(dd[pose]).max(), "point max", np.abs(dd[~pose]).max())
    bad, do, _ = og.solve_damped(1e-5)
    print("  oracle step at its optimum: pose max", np.abs(do[pose]).max(), "point max", np.abs(do[~pose]).max())
    c.close()
