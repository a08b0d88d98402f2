"""GPU vs oracle values: config 2 after k outer iterations; config 1 / stereo-static run into the minimiser (tolerances 1e-12)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dynosam_amd import synth
from dynosam_amd.optimizer import Context, LevenbergMarquardtParams
from oracle import oracle_py as O
O.set_threads(8)
def cmp(tag, g, P):
    og = O.OracleGraph(g); c = Context(); c.upload(g)
    ro, _ = og.optimize(P); r = c.optimize(P)
    v, vo = c.values(), og.state()
    rel = np.abs(v - vo) / np.maximum(1.0, np.abs(vo))
    tr = [bool(r.trace_accepted[i]) for i in range(r.trace_len)] == [bool(ro.trace_accepted[i]) for i in range(ro.trace_len)]
    print(f"{tag}: oracle it {ro.iterations}/{ro.inner_iterations} cost {ro.error_after!r} | gpu it {r.iterations}/{r.inner_iterations} cost {r.error_after!r} | same trace {tr} | max rel dv {rel.max():.3e}  moved {np.abs(vo - g.var_state).max():.3f}", flush=True)
    c.close()
g2 = synth.make_hybrid_graph(synth.config(2))
for k in (3, 6, 10, 15, 25):
    P = LevenbergMarquardtParams(); P.max_iterations = k; P.relative_error_tol = 1e-300; P.absolute_error_tol = 0.0
    cmp(f"config2 k={k}", g2, P)
Pt = LevenbergMarquardtParams(); Pt.relative_error_tol = Pt.absolute_error_tol = 1e-12; Pt.max_iterations = 400
cmp("config1 tight", synth.make_hybrid_graph(synth.config(1)), Pt)
cmp("stereo-static tight", synth.to_stereo_static(synth.make_hybrid_graph(synth.config(1)), behind=5), Pt)
cmp("wcme tight", synth.make_wcme_graph(synth.config(1, frames=40, objects=2, static_points=200, dynamic_points_per_object=40)), Pt)
