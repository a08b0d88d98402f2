"""WCPE stream through the sliding window: where does the marginalisation become indeterminate?  (GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from test_formulation import make_stream, F, to12, from12, compose
from dynosam_amd.synth import se3_exp
from dynosam_amd import sliding_window as SW, symbols as S
from dynosam_amd._lib import DynoError
from oracle import window_oracle as WO
pk, _ = make_stream(n_frames=16, seed=4)
rng = np.random.default_rng(6)
for p in pk[1:]:
    p.X_world = to12(compose(from12(p.X_world), se3_exp(np.concatenate([rng.normal(0, 0.002, 3), rng.normal(0, 0.02, 3)]))))
    p.static[:, 1:] += rng.normal(0, 0.01, p.static[:, 1:].shape)
    p.dynamic[:, 2:] += rng.normal(0, 0.01, p.dynamic[:, 2:].shape)
wf = F.WorldPoseFormulation()
sw = SW.SlidingWindowOptimization(window_size=6, overlap=3)
orig = sw.ctx.marginalize
def marg(keys):
    try:
        return orig(keys)
    except DynoError as e:
        print("GPU:", e)
        g = marg.g
        st = sw.ctx.values()
        w = WO.WindowOracle(g.with_state(st))
        H, gvec, idx = w.hessian(st) if hasattr(w, "hessian") else (None, None, None)
        print("keys to marginalise:", [(chr(S.symbol_chr(k)), S.labeled_index(k) if chr(S.symbol_chr(k)) in "LH" else k & 0xffffffff) for k in keys if g.var_type[g.key_index(k)] == 0])
        try:
            rb, rp = w.marginalize(keys, st)
            print("oracle marginal ok: Lambda eig min/max", np.linalg.eigvalsh(rp.Lambda)[[0, -1]])
        except Exception as ee:   # noqa: BLE001
            print("oracle:", type(ee).__name__, ee)
        raise
sw.ctx.marginalize = marg
for p in pk:
    span = wf.update(p)
    vals, blocks = wf.new_values_and_factors(span)
    import dynosam_amd.sliding_window as M
    _fl = M.flatten
    def fl(values, blocks_, prior):
        g = _fl(values, blocks_, prior); marg.g = g; return g
    M.flatten = fl
    r = sw.update(blocks, vals, p.frame_id)
    M.flatten = _fl
    if r.optimized:
        print("frame", p.frame_id, "window ok", r.report.iterations, r.report.error_before, "->", r.report.error_after)
        wf.set_values(list(r.result), [v[1] for v in r.result.values()])
