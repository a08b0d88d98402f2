"""CPU tests of the frontend path's oracle (oracle/flow_oracle.py) and host logic: no GPU.
The dense-flow producer has no reference arithmetic (the reference consumes an off-line RAFT image):
the oracle is checked against the exactly known flow of the synthetic scene instead."""
import numpy as np
import pytest

from dynosam_amd import synth_images as SI
from oracle import flow_oracle as FO


@pytest.fixture(scope="module")
def scene():
    return SI.make_pair(width=320, height=256, objects=2, seed=11, max_flow=6.0)


def test_oracle_flow_recovers_the_known_motion(scene):
    flow, match = FO.dense_flow(scene["rgb0"], scene["rgb1"])
    e = np.linalg.norm(flow - scene["flow_gt"], axis=-1)[scene["valid"]]
    # errors concentrate at object boundaries (patches straddling two motions): robust statistics
    assert np.median(e) < 0.2 and (e < 1.0).mean() > 0.92 and e.mean() < 0.8


def test_bf16_rounding_is_nearest_even():
    x = np.array([1.0, 1.00390625, 1.01171875, -2.5, 3.140625, 1e-3], np.float32)   # 1+2^-8 ties to even
    b = FO.to_bf16_bits(x)
    assert b[0] == 0x3F80 and b[1] == 0x3F80 and b[2] == 0x3F82
    assert np.all(np.abs(FO.bf16_to_f32(b) - x) <= np.abs(x) * 2.0 ** -8)


def test_track_dynamic_branches_in_reference_order():
    H, W = 48, 64
    mask = np.zeros((H, W), np.int32); mask[10:30, 10:40] = 3; mask[30:40, 10:40] = 4
    flow = np.zeros((H, W, 2), np.float32); flow[..., 0] = 1.5; flow[..., 1] = -0.5
    flow[12, 12] = (0.0, 1.0)
    kp = [(15.7, 15.2), (16.2, 15.9), (5.0, 5.0), (15.0, 32.0), (-1.0, 3.0), (12.3, 12.9), (38.9, 11.0), (20.0, 20.0)]
    prev = [3, 3, 3, 3, 3, 3, 3, 3]
    r = FO.track_dynamic(kp, prev, [0, 1, 2, 3, 4, 5, 30, 7], range(8), flow, mask, max_age=25, min_distance=2, next_tracklet_id=100)
    # 0 kept; 1 falls inside the disc blanked by 0; 2 background; 3 label 4 != 3; 4 not contained; 5 zero flow x; 6 kept, too old -> new tracklet
    assert list(r["code"]) == [FO.KEPT, FO.MASKED_OUT, FO.BACKGROUND, FO.LABEL_CHANGED, FO.NOT_CONTAINED, FO.ZERO_FLOW, FO.KEPT, FO.KEPT]
    assert r["new_tracklet_id"][6] == 100 and r["new_age"][6] == 0 and r["next_tracklet_id"] == 101
    assert r["new_age"][0] == 1 and np.allclose(r["predicted_kp"][0], (17.2, 14.7))


def test_shrunken_image_test_uses_truncated_coordinates():
    H, W = 32, 32
    mask = np.ones((H, W), np.int32)
    flow = np.zeros((H, W, 2), np.float32); flow[..., 0] = 20.0; flow[..., 1] = 0.25
    r = FO.track_dynamic([(20.0, 10.0), (5.0, 10.0)], [1, 1], [0, 0], [0, 1], flow, mask, shrink_row=2, shrink_col=2)
    assert list(r["code"]) == [FO.OUTSIDE_SHRUNKEN, FO.KEPT]
