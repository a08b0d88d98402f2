"""CPU checks of the frontend bookkeeping: the library's anms::RangeTree restatement (host-side integer code inside
libdynogfx.so, no GPU needed) against the numpy oracle, bit for bit; requiresSampling's decisions; sampleDynamic's candidate
rule on a hand-made image."""
import numpy as np
import pytest

from dynosam_amd import flow as F
from oracle import tracker_oracle as TO


@pytest.mark.parametrize("seed,n,k", [(0, 400, 50), (1, 5000, 50), (2, 3000, 200), (3, 60, 50), (4, 10, 3), (5, 2500, 37)])
def test_anms_range_tree_matches_oracle(seed, n, k):
    rng = np.random.default_rng(seed)
    w, h = 640, 480
    # distinct integer pixels of a blob (what sampleDynamic feeds) ...
    cx, cy = rng.integers(100, 540), rng.integers(100, 380)
    px = np.unique(np.stack([np.clip(rng.normal(cx, 40, n).astype(int), 0, w - 1), np.clip(rng.normal(cy, 30, n).astype(int), 0, h - 1)], -1), axis=0)
    order = np.lexsort((px[:, 0], px[:, 1]))            # row-major
    a = px[order].astype(np.float32)
    got = F.anms_range_tree(a, k, 0.01, w, h)
    ref = TO.anms_range_tree(a, k, 0.01, w, h)
    assert np.array_equal(got, ref) and len(got) > 0
    # ... and corner lists with a loose tolerance (the static detector's call: tolerance 0.1, strongest first)
    b = np.stack([rng.uniform(0, w - 1, n), rng.uniform(0, h - 1, n)], -1).astype(np.float32)
    assert np.array_equal(F.anms_range_tree(b, k, 0.1, w, h), TO.anms_range_tree(b, k, 0.1, w, h))


def test_anms_degenerate_requests():
    a = np.array([[3, 4], [10, 10], [300, 200]], np.float32)
    assert len(F.anms_range_tree(a, 0, 0.01, 640, 480)) == 0 == len(TO.anms_range_tree(a, 0, 0.01, 640, 480))
    assert list(F.anms_range_tree(a, 1, 0.01, 640, 480)) == [0] == list(TO.anms_range_tree(a, 1, 0.01, 640, 480))
    assert len(F.anms_range_tree(np.zeros((0, 2), np.float32), 5, 0.01, 640, 480)) == 0


def test_requires_sampling_decisions():
    kp = np.stack([np.linspace(100, 160, 30), np.linspace(80, 140, 30)], -1)
    young, old = np.full(30, 3), np.full(30, 24)
    # new object: always; tracked object with healthy tracks covering its box: no
    objs, why = TO.requires_sampling([1, 2], [(100, 80, 61, 61), (300, 300, 20, 20)], {1: dict(age=young, kp=kp)}, {1})
    assert objs == [2] and why[2]["new"]
    # > 80 % of the tracks older than max_age - max(3, buffer)
    assert TO.requires_sampling([1], [(100, 80, 61, 61)], {1: dict(age=old, kp=kp)}, {1})[0] == [1]
    # too few tracks
    assert TO.requires_sampling([1], [(100, 80, 61, 61)], {1: dict(age=young[:5], kp=kp[:5])}, {1})[1][1]["too_few"]
    # tracks cover a corner of the detection only
    assert TO.requires_sampling([1], [(100, 80, 400, 300)], {1: dict(age=young, kp=kp)}, {1})[1][1]["small_iou"]
    # known object without tracked features: skipped
    assert TO.requires_sampling([1], [(100, 80, 61, 61)], {}, {1})[0] == []


def test_sample_dynamic_candidate_rule():
    h, w = 48, 64
    mask = np.zeros((h, w), np.int32); mask[10:30, 20:50] = 3; mask[35:45, 5:15] = 7
    flow = np.ones((h, w, 2), np.float32); flow[12, 25] = (0, 1); flow[13, 26] = (1, 0)
    det = np.full((h, w), 255, np.uint8); det[10:15, 40:50] = 0
    r = TO.sample_dynamic(mask, flow, det, [3], [10], shrink_row=11, shrink_col=0, next_tracklet_id=100)
    assert r["n_zero_flow"][0] == 2 and (r["label"] == 3).all()
    ys = r["kp"][:, 1]
    assert (ys > 11).all() and len(r["kp"]) == r["n_sampled"][0] > 0
    assert list(r["tracklet_id"]) == list(range(100, 100 + len(r["kp"]))) and r["next_tracklet_id"] == 100 + len(r["kp"])
    assert np.allclose(r["predicted_kp"], r["kp"] + 1.0)
    inside_blank = (r["kp"][:, 1] < 15) & (r["kp"][:, 0] >= 40)
    assert not inside_blank.any()


def test_track_dynamic_klt_frame_keeps_features_on_their_objects():
    """the restated trackDynamicKLT on a small rendered stream: tracked features stay on their object and age by one, features
    beyond max_dynamic_feature_age are dropped, new objects / thinned objects are re-sampled up to max_features, ids are unique"""
    from dynosam_amd import synth_images as SI
    from dynosam_amd.feature_tracker import boarder_thickness
    from oracle import klt_oracle as KO
    from oracle import mask_oracle as MO
    from oracle import tracker_oracle as TO
    W, H = 320, 240
    rgb, mask = SI.make_sequence(W, H, objects=2, frames=6, seed=29)
    prev, tid, pg = None, 7, None
    seen = set()
    for k in range(5):
        g = KO.gray_u8(rgb[k])
        bm = MO.boundary_mask(mask[k], boarder_thickness(W, H), True)
        dyn, ts, st, tid2 = TO.track_dynamic_klt_frame(prev, pg, g, mask[k], dict(boundary_mask=bm["boundary_mask"], objects=bm["objects"], inner_boxes=bm["inner_boxes"]),
                                                       tid, max_age=3, age_buffer=1)
        ids, kp, age, obj = dyn["tracklet_id"], dyn["kp"], dyn["age"], dyn["object_id"]
        assert len(np.unique(ids)) == len(ids)
        assert (mask[k][kp[:, 1].astype(int), kp[:, 0].astype(int)] == obj).all()
        assert age.max() <= 3
        new = ids[age == 0]
        assert np.array_equal(new, np.arange(tid, tid2)) and not (set(new.tolist()) & seen)      # fresh ids, in order
        if prev is not None:
            a = {int(t): i for i, t in enumerate(prev["tracklet_id"])}
            for i in np.nonzero(age > 0)[0]:
                j = a[int(ids[i])]
                assert age[i] == prev["age"][j] + 1 and obj[i] == prev["object_id"][j]
                assert np.abs(kp[i] - prev["kp"][j]).max() < 12.0                                  # the scene moves a few pixels per frame
            assert (age > 0).sum() >= (60 if k <= 3 else 5)          # frame 4: the first generation turns 4 and is dropped
        for o in ts:
            assert st[o]["object_resampled"] and (obj == o).sum() <= 50 + 5          # ANMS stops within its tolerance band or when the search width repeats
        if k == 0:
            assert ts == sorted(bm["objects"]) and all(st[o]["object_new"] for o in ts)
        seen |= set(ids.tolist())
        prev, pg, tid = dict(tracklet_id=ids, kp=kp, age=age, object_id=obj), g, tid2


def test_propogate_mask_rules():
    """FeatureTracker::propogateMask restated (oracle/tracker_oracle.py::propogate_mask): the 150-vote gate, background must win the vote
    (ties go to the smallest label, i.e. to background), zero flow components and the shrunken border are skipped, labels are processed
    one after the other on the same mask"""
    H, W = 120, 160
    prev = np.zeros((H, W), np.int32)
    prev[30:70, 40:90] = 3
    prev[80:110, 100:150] = 5
    flow = np.zeros((H, W, 2), np.float32)
    flow[..., 0], flow[..., 1] = 2.5, -1.25
    rng = np.random.default_rng(0)

    def feats(label, n):
        ys, xs = np.nonzero(prev == label)
        sel = rng.choice(len(ys), n, replace=False)
        kp = np.stack([xs[sel], ys[sel]], 1).astype(float)
        return np.full(n, label), kp + [2.5, -1.25]

    cur = np.zeros((H, W), np.int32)                                     # both objects vanished from the current mask
    o3, p3 = feats(3, 200)
    out, done = TO.propogate_mask(o3, p3, prev, flow, cur)
    assert done == [3] and (out == 3).sum() == (prev == 3).sum()         # every pixel moved by (2.5, -1.25): truncation keeps them distinct
    ys, xs = np.nonzero(out == 3)
    assert ys.min() == 28 and xs.min() == 42                             # int(30 - 1.25) = 28, int(40 + 2.5) = 42
    assert not (out == 5).any() and np.array_equal(cur, np.zeros_like(cur))   # the input mask is not modified
    o3b, p3b = feats(3, 149)
    assert TO.propogate_mask(o3b, p3b, prev, flow, cur)[1] == []         # 149 votes: "not enough points to track object"
    cur2 = cur.copy()
    cur2[20:80, 30:100] = 3                                              # the detector still sees the object: the vote says 3, nothing is warped
    out2, done2 = TO.propogate_mask(o3, p3, prev, flow, cur2)
    assert done2 == [] and np.array_equal(out2, cur2)
    # a tie between background and another label goes to background (the smaller key)
    cur3 = cur.copy()
    u, v = p3[:, 0].astype(int), p3[:, 1].astype(int)
    cur3[v[:100], u[:100]] = 7
    n7 = int((cur3[v, u] == 7).sum())
    if n7 * 2 == len(u):
        assert TO.propogate_mask(o3, p3, prev, flow, cur3)[1] == [3]
    # zero flow component: those pixels stay behind; shrunken border: targets outside it are dropped
    flow2 = flow.copy()
    flow2[30:50, :, 1] = 0.0
    out4, _ = TO.propogate_mask(o3, p3, prev, flow2, cur)
    assert (out4 == 3).sum() == (prev[50:70] == 3).sum()
    out5, _ = TO.propogate_mask(o3, p3, prev, flow, cur, shrink_row=40, shrink_col=0)
    ys5, _ = np.nonzero(out5 == 3)
    assert ys5.min() == 41 and ys5.max() == int(69 - 1.25)
    # two labels, one after the other: both warped, the second one's vote reads the mask the first one left
    o5, p5 = feats(5, 160)
    out6, done6 = TO.propogate_mask(np.concatenate([o3, o5]), np.concatenate([p3, p5]), prev, flow, cur)
    assert done6 == [3, 5] and (out6 == 5).sum() > 0 and (out6 == 3).sum() > 0


def test_reference_fixtures_of_determine_outlier_ids_and_the_chi_square_quantile():
    """Reference-held known answers on the tracker's path: determineOutlierIds (dynosam/test/test_tools.cc:41-68) - the outlier list of
    KltFeatureTracker::trackPoints (StaticFeatureTracker.cc:600-606) - and chi_squared_quantile (dynosam/test/test_numerical.cc:38-50), whose
    values at (2, 0.99) / (3, 0.99) are the outlier thresholds of the two per-object refinements (FactorGraphTools.hpp:81-90: 0.5 x the quantile)."""
    import os
    from scipy.stats import chi2
    from oracle import refine_oracle as RO
    from dynosam_amd import motion_refine as MR
    assert TO.determine_outlier_ids([1, 2], [1, 2, 3, 4, 5]).tolist() == [3, 4, 5]                                  # :41-49
    assert TO.determine_outlier_ids([3, 1, 100], [12, 45, 1, 85, 3, 100]).tolist() == [12, 45, 85]                  # :51-59 unordered inputs
    assert TO.determine_outlier_ids([12, 45, 1, 85, 3, 100], [12, 45, 1, 85, 3, 100]).tolist() == []                # :61-68
    assert chi2.ppf(0.99, 6) == pytest.approx(16.811893829770927, rel=1e-13)                                         # test_numerical.cc:42-43: boost == scipy here
    assert chi2.ppf(0.5, 3) == pytest.approx(2.3659738843753377, rel=1e-13)                                          # :47-48
    assert RO.CHI2_2_099 == pytest.approx(chi2.ppf(0.99, 2), rel=1e-14) and MR.CHI2_3_099 == pytest.approx(chi2.ppf(0.99, 3), rel=1e-14)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    assert "0.5 * 9.210340371976182" in open(os.path.join(root, "dynosam_amd", "csrc", "dynoflow.hip")).read()     # the constants the kernels carry
    assert "0.5 * 11.344866730144373" in open(os.path.join(root, "dynosam_amd", "csrc", "motion_refine.h")).read()


def test_the_static_half_uses_the_reference_shrunken_image_test():
    """KltFeatureTracker::detectFeatures / trackPoints keep a keypoint when camera_->isKeypointContained(kp) && isWithinShrunkenImage(kp)
    (StaticFeatureTracker.cc:402,586).  isWithinShrunkenImage (FeatureTrackerBase.cc:313-326) casts the coordinates to int (functional_keypoint::u / v,
    dynosam_cv/include/dynosam_cv/Feature.hpp:46-55) and compares STRICTLY: row > shrink_row && row < rows - shrink_row && col > shrink_col && col <
    cols - shrink_col - so even without shrinking the first row and column are outside."""
    mask = np.zeros((48, 64), np.int32)
    kp = np.array([[0.5, 10.5], [10.5, 0.5], [1.0, 1.0], [0.99, 5.0], [63.5, 5.0], [62.9, 46.9], [5.0, 47.2], [30.0, 20.0]])
    assert TO._usable_static(kp, mask, 0, 0).tolist() == [False, False, True, False, True, True, True, True]
    kp = np.array([[10.0, 5.99], [10.0, 6.0], [10.0, 6.99], [10.0, 7.0], [8.9, 20.0], [9.0, 20.0], [55.99, 20.0], [56.0, 20.0], [30.0, 41.0], [30.0, 42.0]])
    #      shrink_row 6, shrink_col 8 on 64 x 48:   rows 7..41, columns 9..55 remain
    assert TO._usable_static(kp, mask, 6, 8).tolist() == [False, False, False, True, False, True, True, False, True, False]
    mask[20, 30] = 2
    assert not TO._usable_static(np.array([[30.4, 20.7]]), mask, 0, 0)[0]                       # motion_mask.at<int>(v, u) != background
