"""CPU checks of the per-object flow + pose refinement oracle (oracle/refine_oracle.py): the restated
Pose3FlowProjectionFactor Jacobian against central differences, recovery of a known pose, outlier rejection."""
import numpy as np

from dynosam_amd.synth import act, compose, inverse, se3_exp, to12
from oracle import refine_oracle as R

K = (554.0, 560.0, 0.0, 320.0, 240.0)


def scene(n=60, seed=0, n_out=4, noise=0.3):
    rng = np.random.default_rng(seed)
    Xp = se3_exp(rng.normal(0, 0.05, 6))
    Xk = compose(Xp, se3_exp(np.array([0.01, -0.02, 0.005, 0.1, 0.05, 0.2])))
    kp = np.stack([rng.uniform(50, 590, n), rng.uniform(50, 430, n)], -1)
    depth = rng.uniform(4, 20, n)
    proj = np.array([R._project(K, act(inverse(Xk), act(Xp, R._backproject(K, kp[i], depth[i])))) for i in range(n)])
    flow = proj - kp + rng.normal(0, noise, (n, 2))
    flow[:n_out] += rng.choice([-1, 1], (n_out, 2)) * rng.uniform(60, 90, (n_out, 2))
    X0 = compose(Xk, se3_exp(rng.normal(0, 0.01, 6)))
    return dict(X_prev=to12(Xp), pose_init=to12(X0), kp_prev=kp, depth=depth, flow=flow), Xk


def test_pose_jacobian_matches_central_differences():
    pr, Xk = scene(4, seed=1, n_out=0)
    P = R.FlowPoseProblem(K, pr["X_prev"], pr["pose_init"], pr["kp_prev"], pr["depth"], pr["flow"])
    for i in range(4):
        r0, J, ok = R.flow_factor(K, P.X_prev, P.kp[i], P.depth[i], P.f0[i], P.X0)
        assert ok
        num = np.zeros((2, 6))
        for c in range(6):
            d = np.zeros(6); d[c] = 1e-6
            rp = R.flow_factor(K, P.X_prev, P.kp[i], P.depth[i], P.f0[i], compose(P.X0, se3_exp(d)))[0]
            rm = R.flow_factor(K, P.X_prev, P.kp[i], P.depth[i], P.f0[i], compose(P.X0, se3_exp(-d)))[0]
            num[:, c] = (rp - rm) / 2e-6
        assert np.abs(J - num).max() <= 1e-5 * max(1.0, np.abs(num).max())   # the reference's closed form is the retract derivative
    # behind the camera: constant residual, zero Jacobians
    r, J, ok = R.flow_factor(K, P.X_prev, P.kp[0], -3.0, P.f0[0], P.X0)
    assert not ok and np.all(r == 2 * K[0]) and not J.any()


def test_recovers_pose_and_rejects_gross_outliers():
    pr, Xk = scene()
    res = R.FlowPoseProblem(K, pr["X_prev"], pr["pose_init"], pr["kp_prev"], pr["depth"], pr["flow"]).optimize()
    assert not res["inlier"][:4].any() and res["inlier"][4:].all()
    assert res["error_after"] < res["error_before"]
    assert np.abs(res["pose"] - to12(Xk)).max() < 0.5 * np.abs(pr["pose_init"] - to12(Xk)).max()   # 0.3 px flow noise, Huber far in its linear regime
    # exact data, exact start: nothing to do
    pr2, Xk2 = scene(30, seed=2, n_out=0, noise=0.0)
    pr2["pose_init"] = to12(Xk2)
    res2 = R.FlowPoseProblem(K, pr2["X_prev"], pr2["pose_init"], pr2["kp_prev"], pr2["depth"], pr2["flow"]).optimize()
    assert res2["error_before"] < 1e-20 and res2["inlier"].all() and np.abs(res2["pose"] - to12(Xk2)).max() < 1e-12


def test_camera_fixtures_of_the_reference():
    """The pinhole projection / back-projection the refinement factors restate (oracle/refine_oracle.py _project / _backproject; the kernels carry
    the same two formulas) against the reference's own Camera tests: dynosam/test/test_camera.cc:42-72 (project), :74-110 (back-projection at the
    principal point of the default test camera, helpers.hpp:87-95), :112-140 (top-left corner with uneven focal lengths)."""
    import numpy as np
    from oracle import refine_oracle as RO
    K = (1.0, 1.0, 0.0, 3.0, 2.0)                                                        # :51-57 fx, fy, u0, v0
    lmks = [(0.0, 0.0, 1.0), (0.0, 0.0, 2.0), (0.0, 1.0, 2.0), (0.0, 10.0, 20.0), (1.0, 0.0, 2.0)]          # :44-48
    want = [(3.0, 2.0), (3.0, 2.0), (3.0, 1.0 / 2.0 + 2.0), (3.0, 1.0 / 2.0 + 2.0), (1.0 / 2.0 + 3.0, 2.0)]  # :58-63
    for P, kp in zip(lmks, want):
        assert np.allclose(RO._project(K, np.array(P)), kp, atol=1e-12)
    Kd = (554.256, 554.256, 0.0, 640 / 2, 480 / 2)                                       # helpers.hpp:87-95
    for depth in (2.0, 3.0, 4.5):                                                         # :81-86, :98-108
        assert np.allclose(RO._backproject(Kd, np.array([Kd[3], Kd[4]]), depth), [0.0, 0.0, depth], atol=1e-4)
    fx, fy, cu, cv = 30.9 / 2.2, 12.0 / 23.0, 390.8, 142.2                               # :117-120
    got = RO._backproject((fx, fy, 0.0, cu, cv), np.array([0.0, 0.0]), 2.0)              # :130-134
    assert np.allclose(got, [2.0 / fx * (-cu), 2.0 / fy * (-cv), 2.0], atol=1e-4)        # :136-139
    for P in lmks:                                                                        # the two are inverses of each other
        assert np.allclose(RO._backproject(K, RO._project(K, np.array(P)), P[2]), P, atol=1e-12)
