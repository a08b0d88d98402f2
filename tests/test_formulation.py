"""Incremental HYBRID formulation port (dynosam_amd/formulation.py) on a noiseless synthetic stream and
on the reference's real fixture: the reference's gates / insertion order / keyframe rules, and geometric consistency (the
graph built frame by frame has zero error at the values it was initialised with when the data are exact)."""
import os

import numpy as np
import pytest

from dynosam_amd import graph as G
from dynosam_amd import symbols as S
from dynosam_amd import formulation as F
from dynosam_amd import tracks
from dynosam_amd.synth import act, compose, from12, inverse, se3_exp as expmap_se3, to12

HERE = os.path.dirname(os.path.abspath(__file__))
FIX = os.path.join(HERE, "golden", "small_frontend_tracks.npz")


def make_stream(n_frames=9, gap=None, seed=0):
    """exact camera-frame measurements of static points and of points riding on one rigidly moving object"""
    rng = np.random.default_rng(seed)
    X = [(np.eye(3), np.zeros(3))]
    dX = expmap_se3(np.array([0.003, 0.002, 0.0, 0.014, 0.038, 0.0]))
    for _ in range(n_frames - 1):
        X.append(compose(X[-1], dX))
    Hstep = expmap_se3(np.array([0.0, 0.0, 0.02, 0.1, 0.0, 0.02]))          # world motion of the object per frame
    L = [(np.eye(3), np.array([1.0, 0.5, 8.0]))]
    for _ in range(n_frames - 1):
        L.append(compose(Hstep, L[-1]))
    stat = rng.uniform([-4, -3, 5], [4, 3, 20], (30, 3))
    s_win = [(int(a), int(a + d)) for a, d in zip(rng.integers(0, n_frames - 2, 30), rng.integers(1, 6, 30))]
    body = rng.normal(0, 0.4, (12, 3))
    d_win = [(int(a), int(a + d)) for a, d in zip(rng.integers(0, 3, 12), rng.integers(2, n_frames, 12))]
    pk = []
    for k in range(n_frames):
        st = [(100 + i, *act(inverse(X[k]), stat[i])) for i, (a, b) in enumerate(s_win) if a <= k <= b]
        dy = []
        if gap is None or not (gap[0] <= k <= gap[1]):
            dy = [(500 + i, 1, *act(inverse(X[k]), act(L[k], body[i]))) for i, (a, b) in enumerate(d_win) if a <= k <= b]
        seen_before = k > 0 and (gap is None or not (gap[0] <= k - 1 <= gap[1]))
        mot = {1: to12(Hstep)} if seen_before and dy else {}     # the frontend needs the object in k-1 to estimate a motion
        T = to12(compose(inverse(X[k - 1]), X[k])) if k else None
        pk.append(F.FramePacket(k, to12(X[k]), T, np.array(st).reshape(-1, 4), np.array(dy).reshape(-1, 5), mot))
    return pk, dict(s_win=s_win, d_win=d_win)


def build(pk, **kw):
    hf = F.HybridFormulation(**kw)
    spans = [hf.update(p) for p in pk]
    return hf, spans


def test_gates_counts_and_slot_order():
    pk, info = make_stream()
    hf, spans = build(pk)
    g = hf.graph()
    assert np.all(g.var_keys[1:] > g.var_keys[:-1])
    assert np.array_equal(np.sort(np.concatenate([b.slot for b in g.blocks])), np.arange(g.n_factors))
    n_frames = len(pk)
    # static tracklet observed n times (n >= 2): enters at its 2nd observation with that frame only -> n - 1 factors
    ptp = [f for f in hf.factors if f[0] == G.F_POSE_TO_POINT]
    for i, (a, b) in enumerate(info["s_win"]):
        n = min(b, n_frames - 1) - a + 1
        got = sum(1 for f in ptp if int(f[1][1]) == int(S.StaticLandmarkSymbol(100 + i)))
        assert got == max(0, n - 1), (i, n, got)
    # dynamic tracklet observed n >= 3 consecutive times: pair (2nd, 3rd) when it qualifies, then one per frame -> n - 1
    hm = [f for f in hf.factors if f[0] == G.F_HYBRID_MOTION]
    for i, (a, b) in enumerate(info["d_win"]):
        n = min(b, n_frames - 1) - a + 1
        got = sum(1 for f in hm if int(f[1][2]) == int(S.HybridDynamicKey(500 + i)))
        assert got == (n - 1 if n >= 3 else 0), (i, n, got)
    # per spin: states first, then static factors in tracklet order, then dynamic ones, then priors / smoothing on motions
    for (a, b), p in zip(spans, pk):
        types = [hf.factors[s][0] for s in range(a, b)]
        assert types[0] == (G.F_PRIOR_POSE3 if p.frame_id == 0 else G.F_BETWEEN_POSE3)
        rank = {G.F_POSE_TO_POINT: 1, G.F_HYBRID_MOTION: 2, G.F_PRIOR_POSE3: 3, G.F_HYBRID_SMOOTHING: 3}
        r = [rank[t] for t in types[1:]]
        assert r == sorted(r)
        st_keys = [int(hf.factors[s][1][1]) for s in range(a, b) if hf.factors[s][0] == G.F_POSE_TO_POINT]
        assert st_keys == sorted(st_keys)


def test_keyframe_is_k_minus_one_of_the_first_pair_and_smoothing_needs_three_motions():
    pk, info = make_stream()
    hf, _ = build(pk)
    # the first dynamic tracklets qualify at frame 2 (seen 0,1,2): pair (1,2) -> keyframe 1, L_e = (I, centroid at frame 1)
    (s0, end, L_e), = hf.key_frames[1]
    assert s0 == 1 and end is None and np.array_equal(L_e[0], np.eye(3))
    pts = np.array([hf.dyn_meas[t][1] for t in hf.obj_lmks_at[(1, 1)]])
    assert np.allclose(L_e[1], act(hf.X_init[1], pts.mean(0)))
    priors = [f for f in hf.factors if f[0] == G.F_PRIOR_POSE3]
    assert [int(f[1][0]) for f in priors] == [int(S.CameraPoseSymbol(0)), int(S.ObjectMotionSymbol(1, 1))]
    motions = sorted(S.labeled_index(k) for k in hf.theta if chr(S.symbol_chr(k)) == "H")
    assert motions == list(range(1, len(pk)))
    sm = [f for f in hf.factors if f[0] == G.F_HYBRID_SMOOTHING]
    assert [[int(x) for x in f[1]] for f in sm] == [[int(S.ObjectMotionSymbol(1, k - 2)), int(S.ObjectMotionSymbol(1, k - 1)), int(S.ObjectMotionSymbol(1, k))]
                                                   for k in range(3, len(pk))]


def test_exact_data_gives_zero_error_at_the_initial_values(oracle):
    pk, _ = make_stream()
    hf, _ = build(pk)
    og = oracle.OracleGraph(hf.graph())
    # the sigma = 1e-6 priors and 1e-4 Huber make this a sharp test of the pose / motion / point conventions
    assert og.error() < 1e-12
    r, _ = og.optimize()
    assert r.error_after < 1e-12


def test_gap_longer_than_two_frames_starts_a_new_keyframe():
    pk, _ = make_stream(n_frames=14, gap=(5, 8))
    hf, _ = build(pk)
    rs = hf.key_frames[1]
    assert len(rs) == 2 and rs[0][0] == 1 and rs[0][1] == rs[1][0] == 9 and rs[1][1] is None
    priors = [int(f[1][0]) for f in hf.factors if f[0] == G.F_PRIOR_POSE3]
    assert int(S.ObjectMotionSymbol(1, rs[1][0])) in priors
    # no motion variable inside the gap
    motions = sorted(S.labeled_index(k) for k in hf.theta if chr(S.symbol_chr(k)) == "H")
    assert not any(5 <= m <= 8 for m in motions)


def test_estimates_feed_later_initialisations():
    """computeInitialH composes the frontend motion with the CURRENT estimate of the previous motion (updateTheta)."""
    pk, _ = make_stream()
    hf = F.HybridFormulation()
    for p in pk[:4]:
        hf.update(p)
    k3 = int(S.ObjectMotionSymbol(1, 3))
    bumped = compose(expmap_se3(np.array([0, 0, 0, 0.5, 0, 0])), from12(hf.theta[k3]))
    hf.set_values([k3], [to12(bumped)])
    hf.update(pk[4])
    want = compose(from12(pk[4].motions[1]), bumped)
    assert np.allclose(hf.theta[int(S.ObjectMotionSymbol(1, 4))], to12(want))


@pytest.fixture(scope="module")
def real_incremental():
    fr, X, obs, mot, _ = tracks.load_fixture(FIX)
    hf, spans = build(F.packets_from_arrays(fr, X, obs, mot))
    return hf, spans


def test_real_fixture_incremental_graph(real_incremental, oracle):
    hf, spans = real_incremental
    g = hf.graph()
    assert np.all(g.var_keys[1:] > g.var_keys[:-1])
    assert np.array_equal(np.sort(np.concatenate([b.slot for b in g.blocks])), np.arange(g.n_factors))
    batch = tracks.build_hybrid_graph(*tracks.load_fixture(FIX))
    # without back-tracking the incremental graph is a strict subset of the batch builder's factors
    assert g.n_factors < batch.n_factors and set(int(k) for k in g.var_keys) <= set(int(k) for k in batch.var_keys) | set(int(k) for k in g.var_keys)
    og = oracle.OracleGraph(g)
    e0 = og.error()
    r, _ = og.optimize()
    assert r.error_after < 0.3 * e0 and r.iterations > 2


@pytest.mark.gpu
def test_gpu_follows_the_oracle_on_the_incremental_real_graph(real_incremental, oracle):
    from dynosam_amd.optimizer import Context
    g = real_incremental[0].graph()
    c, og = Context(), oracle.OracleGraph(g)
    c.upload(g)
    assert abs(c.error() - og.error()) <= 1e-12 * og.error()
    rep = c.optimize()
    rr, _ = og.optimize()
    assert rep.iterations == rr.iterations and rep.inner_iterations == rr.inner_iterations
    assert abs(rep.error_after - rr.error_after) <= 1e-6 * rr.error_after
    assert np.abs(c.values() - og.state()).max() <= 1e-5
    c.close()


@pytest.mark.gpu
def test_backend_loop_formulation_plus_sliding_window_on_the_gpu():
    """RegularBackendModule with optimization_mode = sliding window: every spin the formulation appends its new values /
    factors, SlidingWindowOptimization::update solves + marginalises on the GPU when the window is full, and the optimised
    values flow back (updateTheta) so that later initialisations (computeInitialH, getInitialOrLinearizedSensorPose) use them."""
    from dynosam_amd.sliding_window import SlidingWindowOptimization
    from dynosam_amd.synth import se3_exp
    pk, _ = make_stream(n_frames=16, seed=3)
    rng = np.random.default_rng(5)
    truth = [from12(p.X_world) for p in pk]
    for p in pk[1:]:   # noisy frontend: perturbed camera pose estimates and measurements (odometry stays consistent with them)
        p.X_world = to12(compose(from12(p.X_world), se3_exp(np.concatenate([rng.normal(0, 0.002, 3), rng.normal(0, 0.02, 3)]))))
        p.static[:, 1:] += rng.normal(0, 0.01, p.static[:, 1:].shape)
        p.dynamic[:, 2:] += rng.normal(0, 0.01, p.dynamic[:, 2:].shape)
    hf = F.HybridFormulation()
    sw = SlidingWindowOptimization(window_size=6, overlap=3)
    n_opt = 0
    for p in pk:
        span = hf.update(p)
        vals, blocks = hf.new_values_and_factors(span)
        r = sw.update(blocks, vals, p.frame_id)
        if r.optimized:
            n_opt += 1
            assert r.report.error_after < r.report.error_before
            hf.set_values(list(r.result), [v[1] for v in r.result.values()])
    assert n_opt >= 3
    # optimised camera poses are closer to the truth than the frontend's estimates
    est_err = np.mean([np.linalg.norm(from12(hf.theta[int(S.CameraPoseSymbol(k))])[1] - truth[k][1]) for k in range(1, 13)])
    ini_err = np.mean([np.linalg.norm(from12(pk[k].X_world)[1] - truth[k][1]) for k in range(1, 13)])
    assert est_err < 0.7 * ini_err
    sw.ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["wcme", "wcpe"])
def test_backend_loop_world_centric_formulations_plus_sliding_window_on_the_gpu(kind):
    """the same backend loop with the world-centric formulations: their per-frame points form chains that every window cuts, so
    the marginal names chain points that stay coupled (ternary / LandmarkMotionPose factors) with the next, still eliminated ones"""
    from dynosam_amd._lib import IndeterminantLinearSystemException
    from dynosam_amd.sliding_window import NativeSlidingWindowOptimization
    from dynosam_amd.synth import se3_exp
    pk, _ = make_stream(n_frames=16, seed=4)
    rng = np.random.default_rng(6)
    for p in pk[1:]:
        p.X_world = to12(compose(from12(p.X_world), se3_exp(np.concatenate([rng.normal(0, 0.002, 3), rng.normal(0, 0.02, 3)]))))
        p.static[:, 1:] += rng.normal(0, 0.01, p.static[:, 1:].shape)
        p.dynamic[:, 2:] += rng.normal(0, 0.01, p.dynamic[:, 2:].shape)
    wf = F.WorldMotionFormulation() if kind == "wcme" else F.WorldPoseFormulation()
    sw = NativeSlidingWindowOptimization(window_size=6, overlap=3)
    n_opt = 0
    for p in pk:
        span = wf.update(p)
        vals, blocks = wf.new_values_and_factors(span)
        try:
            r = sw.update(blocks, vals, p.frame_id)
        except IndeterminantLinearSystemException:
            # WCPE has no prior on the object poses (WorldPoseEstimator.cc adds none): once fewer than three points of the object
            # remain in the frames being marginalised the eliminated block is singular - the oracle's dense elimination fails on
            # the same window (scripts/dbg_wcpe_window.py), GTSAM's EliminatePreferCholesky throws this exception there
            assert kind == "wcpe" and n_opt >= 2
            break
        if r.optimized:
            n_opt += 1
            assert r.report.error_after < r.report.error_before
            keys, _vt, st = sw.result_values()
            wf.set_values([int(k) for k in keys], list(st))
    assert n_opt >= (3 if kind == "wcme" else 2)
    sw.close(); sw.ctx.close()


# ---- world-centric formulations and the stereo static updater (widened) ----
def test_wcme_builder_exact_data_zero_error_and_structure(oracle):
    pk, info = make_stream()
    wf = F.WorldMotionFormulation()
    for p in pk:
        wf.update(p)
    g = wf.graph()
    types = {b.type: b.count for b in g.blocks}
    # one PoseToPointFactor per dynamic observation in the graph + one ternary factor per consecutive pair
    n_pts = sum(1 for k in wf.theta if chr(S.symbol_chr(k)) == "m")
    dyn_ptp = sum(1 for f in wf.factors if f[0] == G.F_POSE_TO_POINT and chr(S.symbol_chr(int(f[1][1]))) == "m")
    assert dyn_ptp == n_pts and types[G.F_LANDMARK_TERNARY] == n_pts - len(wf.dyn_in_map)   # a tracklet with n points has n - 1 links
    # motions H_k: frontend translation, identity rotation; smoothing = BetweenFactor(H_{k-1}, H_k, I) once per consecutive pair
    for k in wf.theta:
        if chr(S.symbol_chr(k)) == "H":
            assert np.array_equal(wf.theta[k][:9], np.eye(3).reshape(-1))
    btw = [f for f in wf.factors if f[0] == G.F_BETWEEN_POSE3 and chr(S.symbol_chr(int(f[1][0]))) == "H"]
    pairs = [(S.labeled_index(int(f[1][0])), S.labeled_index(int(f[1][1]))) for f in btw]
    assert len(pairs) == len(set(pairs)) and all(b == a + 1 for a, b in pairs) and len(pairs) >= 4
    # exact data: the points are exact, the motions are initialised with the wrong (identity) rotation -> optimisable to zero
    og = oracle.OracleGraph(g)
    r, _ = og.optimize()
    assert r.error_after < 1e-10 * max(1.0, r.error_before)
    # slots: insertion order; the ternary factor of a pair follows the two point factors of that pair
    assert np.array_equal(np.sort(np.concatenate([b.slot for b in g.blocks])), np.arange(g.n_factors))


def test_wcpe_builder_exact_data_and_the_reference_s_repeated_smoothing_factor(oracle):
    pk, _ = make_stream()
    wf = F.WorldPoseFormulation()
    for p in pk:
        wf.update(p)
    g = wf.graph()
    types = {b.type: b.count for b in g.blocks}
    assert types[G.F_LANDMARK_MOTION_POSE] > 20 and types[G.F_LANDMARK_POSE_SMOOTHING] > 3
    sm = [tuple(S.labeled_index(int(x)) for x in f[1]) for f in wf.factors if f[0] == G.F_LANDMARK_POSE_SMOOTHING]
    from collections import Counter
    c = Counter(sm)
    assert all(b == a + 1 and d == b + 1 for a, b, d in sm)
    assert max(c.values()) == 2                    # objectUpdateContext runs for both affected frames of every spin (no guard in the reference)
    og = oracle.OracleGraph(g)
    r, _ = og.optimize()
    assert r.error_after < 1e-10 * max(1.0, r.error_before)


def test_stereo_static_updater_triangulates_and_gates(oracle):
    pk, info = make_stream()
    cal = F.StereoCalibration(fx=700.0, fy=700.0, u0=320.0, v0=240.0, baseline=0.12, pixel_sigma=1.0)
    hf = F.HybridFormulation(static_formulation="stereo", stereo=cal)
    for p in pk:
        hf.update(p)
    g = hf.graph()
    st = [b for b in g.blocks if b.type == G.F_STEREO_POINT]
    assert len(st) == 1 and st[0].count > 20 and not any(b.type == G.F_POSE_TO_POINT for b in g.blocks)
    # measurement = (uL, uL - fx b / depth, v); every factor carries the fake stereo calibration
    assert np.allclose(st[0].consts, cal.k6()[None])
    assert np.all(st[0].meas[:, 0] - st[0].meas[:, 1] > 0.5)
    # a tracklet enters with ALL its observations so far once it can be triangulated (two or more frames), then one per frame
    fac = [f for f in hf.factors if f[0] == G.F_STEREO_POINT]
    n_frames = len(pk)
    for i, (a, b) in enumerate(info["s_win"]):
        n = min(b, n_frames - 1) - a + 1
        got = sum(1 for f in fac if int(f[1][1]) == int(S.StaticLandmarkSymbol(100 + i)))
        assert got == (n if n >= 2 else 0), (i, n, got)
    # exact data: the triangulated initial points are the true points -> zero error
    og = oracle.OracleGraph(g)
    assert og.error() < 1e-9
    # a point at infinity-like disparity is never inserted (disparity gate) and a wild observation rejects the tracklet
    far = F.FramePacket(0, pk[0].X_world, None, np.array([[900, 0.0, 0.0, 400.0]]), np.zeros((0, 5)), {})
    far2 = F.FramePacket(1, pk[1].X_world, pk[1].T_k_1_k, np.array([[900, 0.0, 0.0, 400.0]]), np.zeros((0, 5)), {})
    h2 = F.HybridFormulation(static_formulation="stereo", stereo=cal)
    h2.update(far); h2.update(far2)
    assert int(S.StaticLandmarkSymbol(900)) not in h2.theta
