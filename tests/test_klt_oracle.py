"""CPU checks of the sparse pyramidal LK oracle (oracle/klt_oracle.py): the restated cv::calcOpticalFlowPyrLK recovers the
exactly known flow of the synthetic scene, the flow-back check rejects occluded / border tracks, and the building blocks
(grey conversion, pyrDown, Scharr) satisfy the identities their definitions imply.  Parity with the OpenCV binary itself is
unpinned (OpenCV is not in this image); see the oracle's header."""
import numpy as np
import pytest

from dynosam_amd import synth_images as SI
from oracle import klt_oracle as K


@pytest.fixture(scope="module")
def scene():
    p = SI.make_pair(width=640, height=480, objects=3, seed=4)
    p["g0"], p["g1"] = K.gray_u8(p["rgb0"]), K.gray_u8(p["rgb1"])
    return p


def test_gray_is_the_opencv_fixed_point_formula():
    rgb = np.array([[[255, 255, 255], [0, 0, 0], [255, 0, 0], [0, 255, 0], [0, 0, 255], [12, 200, 77]]], np.uint8)
    assert K.gray_u8(rgb).tolist() == [[255, 0, 76, 150, 29, int((12 * 4899 + 200 * 9617 + 77 * 1868 + 8192) >> 14)]]


def test_pyrdown_and_scharr_identities():
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (37, 52), dtype=np.uint8)
    d = K.pyr_down(img)
    assert d.shape == (19, 26)
    assert np.array_equal(K.pyr_down(np.full((30, 30), 77, np.uint8)), np.full((15, 15), 77, np.uint8))   # kernel sums to 256
    # a horizontal ramp: dx = 32 * slope in the interior, dy = 0
    ramp = np.tile(np.arange(40, dtype=np.uint8) * 3, (20, 1))
    dx, dy = K.scharr(ramp)
    assert np.all(dx[:, 1:-1] == 2 * 3 * 16) and np.all(dy == 0)
    assert np.all(dx[:, 0] == 0) and np.all(dx[:, -1] == 0)   # reflect-101: the two neighbours coincide
    # transposing the image swaps the two derivatives
    dxt, dyt = K.scharr(np.ascontiguousarray(img.T))
    dx, dy = K.scharr(img)
    assert np.array_equal(dxt.T, dy) and np.array_equal(dyt.T, dx)


def test_pyramid_stops_before_a_level_not_larger_than_the_window(scene):
    assert [p.shape for p in K.build_pyramid(scene["g0"], 3)] == [(480, 640), (240, 320), (120, 160), (60, 80)]
    assert len(K.build_pyramid(scene["g0"], 5)) == 5   # 20x15 is not larger than 21x21


def test_recovers_known_flow_and_flow_back_check(scene):
    rng = np.random.default_rng(1)
    pts = np.stack([rng.uniform(25, 615, 80), rng.uniform(25, 455, 80)], -1).astype(np.float32)
    cur, back, good, fwd = K.track_points(scene["g0"], scene["g1"], pts)
    yi, xi = np.round(pts[:, 1]).astype(int), np.round(pts[:, 0]).astype(int)
    # keep points whose whole window moves rigidly: same label over the window in frame 0, visible in frame 1
    m0 = scene["mask0"]
    rigid = np.array([np.all(m0[max(0, y - 12):y + 13, max(0, x - 12):x + 13] == m0[y, x]) for x, y in zip(xi, yi)]) & scene["valid"][yi, xi]
    assert rigid.sum() > 40
    gt = scene["flow_gt"][yi, xi]
    # background flow is an integer translation: sub-0.01 px there, affine objects are evaluated at the rounded pixel
    err = np.linalg.norm(cur - pts - gt, axis=1)
    # (a window that is rigid in frame 0 can still be entered by a moving object in frame 1: those fail the flow-back check)
    sel = rigid & (good == 1)
    assert sel.sum() >= 0.95 * rigid.sum()
    assert np.median(err[sel]) < 0.01 and err[sel].max() < 0.2
    assert np.all(np.linalg.norm(back[good == 1] - pts[good == 1], axis=1) <= 0.5)


def test_initial_flow_and_failure_paths(scene):
    pts = np.array([[100.0, 100.0], [320.0, 200.0], [-40.0, 50.0], [700.0, 100.0]], np.float32)
    cur, st = K.calc_pyr_lk(scene["g0"], scene["g1"], pts)
    assert st.tolist()[:2] == [1, 1] and st.tolist()[2:] == [0, 0]          # window origin outside [-21, w): status cleared
    # a good initial guess converges to the same place as the cold start (within the stop criterion)
    cur2, st2 = K.calc_pyr_lk(scene["g0"], scene["g1"], pts[:2], init_pts=cur[:2] + 0.3)
    assert st2.tolist() == [1, 1] and np.abs(cur2 - cur[:2]).max() < 0.05
    # a textureless image fails the min-eigenvalue test at level 0
    flat = np.full((480, 640), 90, np.uint8)
    _, st3 = K.calc_pyr_lk(flat, flat, pts[:2])
    assert st3.tolist() == [0, 0]
    # fewer than 10 successes with an initial flow -> retried without it (StaticFeatureTracker.cc:491-503)
    bad_init = pts[:2] + np.float32(400.0)
    c4, _, good4, _ = K.track_points(scene["g0"], scene["g1"], pts[:2], init_pts=bad_init)
    assert good4.tolist() == [1, 1] and np.abs(c4 - cur[:2]).max() == 0.0


def _rot(axis, deg):
    a = np.asarray(axis, float) / np.linalg.norm(axis)
    t = np.deg2rad(deg)
    Kx = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + np.sin(t) * Kx + (1 - np.cos(t)) * Kx @ Kx


def test_predict_keypoints_given_rotation_properties():
    """predictKeypointsGivenRotation (FeatureTrackerBase.cc:50-105) restated in oracle/klt_oracle.py - properties the reference's arithmetic must
    have: the identity and any rotation whose quaternion w is within 1e-4 of 1 copy the points; a rotation about the optical axis turns the points
    about the principal point; two rotations compose; a point that would leave the (shrunken) image or fall behind the camera keeps its place."""
    from oracle import klt_oracle as KO
    K = np.array([[500.0, 0, 320.0], [0, 500.0, 240.0], [0, 0, 1.0]])
    rng = np.random.default_rng(5)
    pts = np.stack([rng.uniform(60, 580, 200), rng.uniform(60, 420, 200)], 1).astype(np.float32)
    W, H = 640, 480
    assert np.array_equal(KO.predict_keypoints_given_rotation(pts, np.eye(3), K, W, H), pts)
    assert np.array_equal(KO.predict_keypoints_given_rotation(pts, _rot([0, 1, 0], 1.0), K, W, H), pts)      # |1 - |w|| = 3.8e-5 < 1e-4: copied
    # about the optical axis: x' - c = R2d (x - c)
    Rz = _rot([0, 0, 1], 8.0)
    out = KO.predict_keypoints_given_rotation(pts, Rz, K, W, H)
    c = np.array([320.0, 240.0])
    want = (pts - c) @ Rz[:2, :2].T + c
    inside = (want[:, 0] > 1) & (want[:, 0] < W - 1) & (want[:, 1] > 1) & (want[:, 1] < H - 1)
    assert inside.sum() > 150 and np.abs(out[inside] - want[inside]).max() < 2e-3
    assert np.array_equal(out[~inside], pts[~inside])                                                         # would leave the image: kept where it was
    # composition (both rotations above the copy threshold), away from the border
    R1, R2 = _rot([0.2, 1, 0.1], 4.0), _rot([1, -0.3, 0.2], 3.0)
    mid = pts[(np.abs(pts[:, 0] - 320) < 120) & (np.abs(pts[:, 1] - 240) < 90)]
    a = KO.predict_keypoints_given_rotation(KO.predict_keypoints_given_rotation(mid, R1, K, W, H), R2, K, W, H)
    b = KO.predict_keypoints_given_rotation(mid, R2 @ R1, K, W, H)
    assert np.abs(a - b).max() < 5e-3
    # the shrunken image: a larger margin keeps more points in place
    out0 = KO.predict_keypoints_given_rotation(pts, _rot([0, 1, 0], 12.0), K, W, H)
    out1 = KO.predict_keypoints_given_rotation(pts, _rot([0, 1, 0], 12.0), K, W, H, shrink_row=60, shrink_col=120)
    assert (out1 == pts).all(1).sum() > (out0 == pts).all(1).sum()
    # behind the camera (a 120 degree turn): nothing moves to a negative depth
    far = KO.predict_keypoints_given_rotation(pts, _rot([0, 1, 0], 120.0), K, W, H)
    H3 = KO.rotation_homography(_rot([0, 1, 0], 120.0), K).astype(np.float64)
    z = H3[2, 0] * pts[:, 0] + H3[2, 1] * pts[:, 1] + H3[2, 2]
    assert np.array_equal(far[z <= 0], pts[z <= 0])
