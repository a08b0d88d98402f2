import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dynosam_amd import synth
from dynosam_amd.optimizer import Context, LevenbergMarquardtParams
from oracle import oracle_py as O
O.lib(); O.set_threads(8)
g = synth.make_hybrid_graph(synth.config(5))
P = LevenbergMarquardtParams(); P.max_iterations = 3
og = O.OracleGraph(g); ro, _ = og.optimize(P); vo = og.state()
bad, d, _ = og.solve_damped(1e-5)
print("D", np.abs(vo - g.var_state).max(), "s", np.abs(d).max())
c = Context(); c.upload(g); r = c.optimize(P); v = c.values()
diff = np.abs(v - vo)
pose = g.var_type == 0
print("pose max", diff[pose].max(), "point max", diff[~pose].max())
i = np.unravel_index(np.argmax(diff), diff.shape)
print("worst", i, "type", g.var_type[i[0]], "key", hex(int(g.var_keys[i[0]])), "v", v[i[0]], "vo", vo[i[0]], "init", g.var_state[i[0]])
print("trace err", [r.trace_error[k] for k in range(r.trace_len)], [ro.trace_error[k] for k in range(ro.trace_len)])
# step sizes of each iteration
print("quantiles pose diff", np.quantile(diff[pose].max(axis=1), [0.5, 0.9, 0.99, 1.0]))
print("quantiles point diff", np.quantile(diff[~pose].max(axis=1), [0.5, 0.9, 0.99, 1.0]))
