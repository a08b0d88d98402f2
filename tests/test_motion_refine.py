"""MotionOnlyRefinementOptimizer mirror (dynosam_amd/motion_refine.py): the graph it builds from existing factor classes is
the reference's (monocular projection = zero-baseline stereo class with a rank-2 square-root information), the oracle solves
it, outliers are rejected, and the GPU follows the oracle."""
import numpy as np
import pytest

from dynosam_amd import graph as G
from dynosam_amd import motion_refine as MR
from dynosam_amd.synth import act, compose, inverse, se3_exp, to12

K = (554.0, 554.0, 0.0, 320.0, 240.0)


def scene(n=25, seed=0, n_out=3):
    rng = np.random.default_rng(seed)
    X0 = se3_exp(rng.normal(0, 0.02, 6))
    X1 = compose(X0, se3_exp(np.array([0.003, 0.002, 0.0, 0.014, 0.038, 0.0])))
    H = se3_exp(np.array([0.0, 0.0, 0.03, 0.15, 0.02, 0.05]))
    m0 = np.array([act(X0, p) for p in rng.uniform([-2, -1.5, 6], [2, 1.5, 12], (n, 3))])
    m1 = np.array([act(H, p) for p in m0])
    proj = lambda X, p: np.array([K[0] * q[0] / q[2] + K[3] for q in [act(inverse(X), p)]] + [K[1] * q[1] / q[2] + K[4] for q in [act(inverse(X), p)]])
    kp0 = np.array([proj(X0, p) for p in m0]) + rng.normal(0, 0.2, (n, 2))
    kp1 = np.array([proj(X1, p) for p in m1]) + rng.normal(0, 0.2, (n, 2))
    l0 = m0 + rng.normal(0, 0.002, m0.shape)
    l1 = m1 + rng.normal(0, 0.002, m1.shape)
    kp1[:n_out] += rng.choice([-1, 1], (n_out, 2)) * 40.0       # gross pixel errors on a few tracklets
    H0 = compose(H, se3_exp(rng.normal(0, 0.01, 6)))
    return dict(X0=to12(X0), X1=to12(X1), H=to12(H), H0=to12(H0), tr=np.arange(100, 100 + n), kp0=kp0, kp1=kp1, l0=l0, l1=l1)


def oracle_solver(oracle):
    def solve(g, max_iterations):
        og = oracle.OracleGraph(g)
        P = oracle.default_params() if hasattr(oracle, "default_params") else None
        e0 = og.error()
        from dynosam_amd.optimizer import LevenbergMarquardtParams
        P = LevenbergMarquardtParams(); P.max_iterations = max_iterations
        r, _ = og.optimize(P)
        return og.state(), e0, r.error_after
    return solve


def test_projection_factor_is_the_zero_baseline_stereo_class(oracle):
    s = scene(6, seed=1, n_out=0)
    g = MR.build_graph(K, 3, 4, 2, s["X0"], s["X1"], s["H0"], s["tr"], s["kp0"], s["kp1"], s["l0"], s["l1"], MR.MotionRefineParams())
    assert [b.type for b in g.blocks] == [G.F_PRIOR_POSE3, G.F_STEREO_POINT, G.F_LANDMARK_TERNARY]
    assert np.array_equal(np.sort(np.concatenate([b.slot for b in g.blocks])), np.arange(2 + 3 * 6))
    # per-factor errors of the oracle = Huber(|(du, dv)| / sigma) of the pinhole projection, computed by hand
    og = oracle.OracleGraph(g)
    J, b, e = og.linearize()
    p = MR.MotionRefineParams()
    X0 = (np.asarray(s["X0"][:9]).reshape(3, 3), np.asarray(s["X0"][9:]))
    q = act(inverse(X0), s["l0"][0])
    du = np.array([K[0] * q[0] / q[2] + K[3] - s["kp0"][0][0], K[1] * q[1] / q[2] + K[4] - s["kp0"][0][1]])
    d = np.linalg.norm(du) / p.projection_sigma
    want = 0.5 * d * d if d <= p.k_huber else p.k_huber * (d - 0.5 * p.k_huber)
    assert abs(e[2] - want) <= 1e-12 * max(1.0, want)      # factor 2 = the first projection factor


def test_recovers_motion_and_rejects_outliers(oracle):
    s0 = scene(n_out=0)
    res0 = MR.optimize(oracle_solver(oracle), K, 3, 4, 2, s0["X0"], s0["X1"], s0["H0"], s0["tr"], s0["kp0"], s0["kp1"], s0["l0"], s0["l1"])
    # (the points carry no prior, so depth along the rays - and with it the scale of the object's translation - is only held
    # by the LM damping: the cost must drop, the motion need not approach the truth)
    assert len(res0["outliers"]) == 0 and res0["error_after"] < 0.05 * res0["error_before"]
    s = scene()
    # with the reference's sigmas (2 px vs 1 mm) the robust projection factors absorb a bad pixel and the ternary test stays
    # silent; trusting the pixels makes the motion factors of the corrupted tracklets stick out
    res = MR.optimize(oracle_solver(oracle), K, 3, 4, 2, s["X0"], s["X1"], s["H0"], s["tr"], s["kp0"], s["kp1"], s["l0"], s["l1"],
                      MR.MotionRefineParams(projection_sigma=0.002, landmark_motion_sigma=0.01, k_huber=10.0))
    assert {100, 101, 102} <= set(res["outliers"].tolist()) and len(res["inliers"]) >= 18


@pytest.mark.gpu
def test_gpu_follows_the_oracle(oracle):
    s = scene(40, seed=3, n_out=4)
    args = (K, 3, 4, 2, s["X0"], s["X1"], s["H0"], s["tr"], s["kp0"], s["kp1"], s["l0"], s["l1"])
    ref = MR.optimize(oracle_solver(oracle), *args)
    solve = MR.gpu_solver()
    got = MR.optimize(solve, *args)
    pr = MR.MotionRefineParams(projection_sigma=0.002, landmark_motion_sigma=0.01, k_huber=10.0)     # the outlier rounds
    ref2, got2 = MR.optimize(oracle_solver(oracle), *args, pr), MR.optimize(solve, *args, pr)
    assert len(ref2["outliers"]) >= 4 and np.array_equal(got2["outliers"], ref2["outliers"])
    # (a stiff, scale-deficient problem cut off after 5 iterations per round: the two LM trajectories may separate; both must
    # have removed the same factors and collapsed the cost)
    assert got2["error_after"] < 1e-3 * got2["error_before"] and ref2["error_after"] < 1e-3 * ref2["error_before"]
    assert np.array_equal(got["outliers"], ref["outliers"])
    assert np.abs(got["best_result"] - ref["best_result"]).max() <= 1e-6
    assert abs(got["error_after"] - ref["error_after"]) <= 1e-6 * max(ref["error_after"], 1e-9)
    solve.ctx.close()
