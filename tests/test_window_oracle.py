"""CPU checks of the sliding-window oracle (oracle/window_oracle.py): a linear container re-expresses its factor
exactly at the linearisation point; marginalising reproduces the optimum of the full problem on the retained
variables (the defining property of a Schur-complement marginal for a quadratic); no GPU."""
import numpy as np
import pytest

from dynosam_amd import synth
from dynosam_amd.graph import FlatGraph
from oracle import window_oracle as WO


def tiny(noise=1.0):
    return synth.make_hybrid_graph(synth.config(1, frames=8, static_points=24, dynamic_points_per_object=8, static_track=(3, 6),
                                                dynamic_track=(3, 6), noise_scale=noise, seed=5))


def old_keys(g, cutoff):
    return [int(k) for k, f in zip(g.var_keys, g.meta["var_frame"]) if f < cutoff]


def test_marginal_of_the_linearised_problem_keeps_the_retained_optimum():
    g = tiny()
    w = WO.WindowOracle(g)
    x = g.var_state
    H, gv, c = w.normal_equations(x)
    lam = 1e-9
    full = np.linalg.solve(H + lam * np.eye(w.n), gv)
    keys = old_keys(g, 4)
    blocks, prior = w.marginalize(keys, x)
    keep = np.array([i for i, k in enumerate(g.var_keys) if int(k) not in set(keys)])
    g2 = FlatGraph(g.var_keys[keep], g.var_type[keep], x[keep], [], {}, prior)
    remap = -np.ones(g.n_vars, int); remap[keep] = np.arange(len(keep))
    for b in blocks:
        b.var_idx = remap[b.var_idx].astype(np.int32)
        assert (b.var_idx >= 0).all()
    g2.blocks = blocks
    w2 = WO.WindowOracle(g2)
    H2, g2v, c2 = w2.normal_equations(g2.var_state)
    red = np.linalg.solve(H2 + lam * np.eye(w2.n), g2v)
    sel = np.concatenate([np.arange(w.off[v], w.off[v + 1]) for v in keep])
    assert np.abs(red - full[sel]).max() <= 1e-6 * max(1.0, np.abs(full).max())
    # constants: the minimum of the reduced quadratic equals the minimum of the full one
    qfull = c - 0.5 * gv @ np.linalg.solve(H, gv)
    qred = c2 - 0.5 * g2v @ np.linalg.solve(H2, g2v)
    assert abs(qfull - qred) <= 1e-8 * max(1.0, abs(qfull))
    # and at the linearisation point the carried graph's error is its own constant term
    assert abs(w2.error(g2.var_state) - c2) <= 1e-9 * max(1.0, c2)


def test_linear_container_relinearises_like_gtsam():
    g = tiny()
    w = WO.WindowOracle(g)
    blocks, prior = w.marginalize([], g.var_state)       # nothing marginalised: every factor becomes a container
    assert prior is None and sum(b.count for b in blocks) == g.n_factors
    gl = FlatGraph(g.var_keys, g.var_type, g.var_state, blocks, {})
    wl = WO.WindowOracle(gl)
    H, gv, c = w.normal_equations(g.var_state)
    Hl, gl_, cl = wl.normal_equations(g.var_state)
    assert np.allclose(H, Hl, atol=1e-9 * np.abs(H).max()) and np.allclose(gv, gl_, atol=1e-9 * np.abs(gv).max()) and abs(c - cl) < 1e-9 * c
    # moved away from the linearisation point the Jacobian stays, b shifts by A Local(lin, x)
    rng = np.random.default_rng(0)
    x2 = w.retract(g.var_state, 1e-3 * rng.normal(size=w.n))
    H2, _, _ = wl.normal_equations(x2)
    assert np.allclose(H2, Hl, atol=1e-12 * np.abs(H).max())


def test_window_lm_converges_and_matches_the_c_oracle_without_priors():
    from oracle import oracle_py as O
    g = tiny()
    r1, _ = WO.WindowOracle(g).optimize()
    og = O.OracleGraph(g); og.set_dense(True)
    r2, _ = og.optimize()
    assert r1.iterations == r2.iterations and abs(r1.error_after - r2.error_after) <= 1e-7 * r2.error_after


def test_schur_solve_of_the_window_oracle_equals_its_dense_solve():
    """full-density windows solve the damped normal equations through the Schur complement of the uncoupled points
    (WindowOracle._solve); it must be the same system as the plain dense Cholesky the small windows use"""
    g = synth.make_hybrid_graph(synth.config(1, frames=10, static_points=60, dynamic_points_per_object=16, static_track=(3, 6),
                                             dynamic_track=(3, 6), seed=7))
    # marginalise everything older than frame 4 except the points born in frames 2-3: they stay, next to marginalised poses
    keys = [int(k) for k, f, t in zip(g.var_keys, g.meta["var_frame"], g.var_type) if f < 4 and not (t == 1 and f >= 2)]
    w = WO.WindowOracle(g)
    blocks, prior = w.marginalize(keys, g.var_state)        # the next window: containers + a prior that names points
    keep = np.array([i for i, k in enumerate(g.var_keys) if int(k) not in set(keys)])
    remap = -np.ones(g.n_vars, int); remap[keep] = np.arange(len(keep))
    for b in blocks:
        b.var_idx = remap[b.var_idx].astype(np.int32)
    g2 = FlatGraph(g.var_keys[keep], g.var_type[keep], g.var_state[keep], blocks, {}, prior)
    w2 = WO.WindowOracle(g2)
    assert 0 < len(w2.free_pts) < (g2.var_type == 1).sum()  # some points are coupled by the prior, the others are eliminated
    H, gv, _ = w2.normal_equations(g2.var_state)
    for lam in (1e-5, 1.0):
        w2.SCHUR_MIN_DIM = 10 ** 9
        dense = w2._solve(H, gv, lam)
        w2.SCHUR_MIN_DIM = 0
        schur = w2._solve(H, gv, lam)
        assert np.abs(schur - dense).max() <= 1e-6 * max(1.0, np.abs(dense).max())   # (cond ~ 1e12: the damped-solve tolerance used throughout)
