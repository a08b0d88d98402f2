"""Frontend path on the GPU (through the C-ABI of include/dynoflow.h) against oracle/flow_oracle.py.

Tolerances: pyramid levels and descriptors are bit-exact (same operation order, correctly rounded f32 ops);
the coarse MFMA arg-max may differ from numpy's float32 matmul only on near-ties (accumulation order), so
>= 99.5 % of the matches and >= 99 % of the final flow vectors (1e-3 px) must agree; trackDynamic's
per-feature bookkeeping is integer/byte work and must be bit-exact; end-point error against the exactly
known synthetic flow: median < 0.2 px, > 92 % of the visible pixels below 1 px."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from dynosam_amd import synth_images as SI  # noqa: E402
from oracle import flow_oracle as FO  # noqa: E402


@pytest.fixture(scope="module")
def scene():
    return SI.make_pair(width=640, height=480, objects=3, seed=4)


@pytest.fixture(scope="module")
def tracker(scene):
    from dynosam_amd.flow import FlowTracker
    t = FlowTracker(640, 480)
    t.upload(scene["rgb0"], scene["mask0"], scene["rgb1"], scene["mask1"])
    t.flow, t.match = t.dense_flow()
    return t


def test_pyramid_and_descriptors_bit_exact(scene, tracker):
    for f, key in ((0, "rgb0"), (1, "rgb1")):
        pyr = FO.pyramid(scene[key])
        for lvl in range(4):
            assert np.array_equal(tracker.level(f, lvl), pyr[lvl]), (f, lvl)
        assert np.array_equal(tracker.descriptors(f), FO.descriptors(pyr[3])), f


def test_coarse_matches_and_flow_agree_with_oracle(scene, tracker):
    flow, match = FO.dense_flow(scene["rgb0"], scene["rgb1"])
    assert (tracker.match == match).mean() >= 0.995
    close = np.abs(tracker.flow - flow).max(-1) <= 1e-3
    assert close.mean() >= 0.99


def test_end_point_error_against_exact_flow(scene, tracker):
    e = np.linalg.norm(tracker.flow - scene["flow_gt"], axis=-1)[scene["valid"]]
    assert np.median(e) < 0.2 and (e < 1.0).mean() > 0.92


def test_track_dynamic_bit_exact(scene, tracker):
    rng = np.random.default_rng(7)
    ys, xs = np.nonzero(scene["mask0"] > 0)
    pick = rng.choice(len(xs), 600, replace=False)
    kp = np.stack([xs[pick] + rng.uniform(0, 1, 600), ys[pick] + rng.uniform(0, 1, 600)], -1)
    kp[:40] = rng.uniform(-5, 650, (40, 2))                      # some outside / on the background
    prev = scene["mask0"][np.clip(kp[:, 1].astype(int), 0, 479), np.clip(kp[:, 0].astype(int), 0, 639)].copy()
    prev[40:80] = 1 + (prev[40:80] % 3)                          # some with a different previous label
    prev = np.maximum(prev, 1)
    age = rng.integers(0, 30, 600)
    det = np.full((480, 640), 255, np.uint8); det[200:260, 300:360] = 0
    kw = dict(shrink_row=3, shrink_col=5, max_age=25, min_distance=2, next_tracklet_id=5000)
    got = tracker.track_dynamic(kp, prev, age, np.arange(600), detection_mask=det, **kw)
    ref = FO.track_dynamic(kp, prev, age, np.arange(600), tracker.flow, scene["mask0"], detection_mask=det, **kw)
    for k in ("code", "label", "new_age", "new_tracklet_id", "flow", "predicted_kp"):
        assert np.array_equal(got[k], ref[k]), k
    assert got["next_tracklet_id"] == ref["next_tracklet_id"]
    assert (got["code"] == FO.KEPT).sum() > 100 and len(set(got["code"])) >= 5
