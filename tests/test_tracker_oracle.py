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
