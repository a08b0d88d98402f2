"""CPU checks of the Shi-Tomasi oracle (oracle/gftt_oracle.py): response properties, mask / min-distance / ordering rules of
cv::goodFeaturesToTrack as restated there.  Parity with the OpenCV binary is unpinned (see the oracle's header)."""
import numpy as np

from oracle import gftt_oracle as G


def _checker(h=96, w=128, cell=16):
    ys, xs = np.mgrid[0:h, 0:w]
    return (((ys // cell) + (xs // cell)) % 2 * 200 + 20).astype(np.uint8)


def test_response_is_zero_on_flat_and_edge_regions_and_peaks_at_corners():
    img = _checker()
    eig = G.min_eigen_val(img)
    assert eig.dtype == np.float32 and eig.shape == img.shape
    assert eig[8, 8] == 0.0                                  # flat
    assert abs(eig[8, 16]) < 1e-9 and abs(eig[16, 8]) < 1e-9   # pure edges: rank-1 structure tensor
    assert eig[16, 16] > 1e-3 and eig[15:18, 15:18].max() == eig.max()   # checker corner


def test_corners_sit_on_the_checker_lattice_sorted_and_separated():
    img = _checker()
    c, eig = G.good_features_to_track(img, None, max_corners=100, quality_level=0.01, min_distance=8.0)
    assert len(c) == 5 * 7                                   # interior lattice crossings
    assert np.all(np.abs((c + 0.5) / 16 - np.round((c + 0.5) / 16)) <= 1.0 / 16)
    v = eig[c[:, 1].astype(int), c[:, 0].astype(int)]
    assert np.all(np.diff(v) <= 0)                           # strongest first
    d = np.linalg.norm(c[:, None] - c[None], axis=-1) + 1e9 * np.eye(len(c))
    assert d.min() >= 8.0


def test_mask_limits_max_and_candidates_and_max_corners_truncates():
    img = _checker()
    mask = np.zeros(img.shape, np.uint8)
    mask[:, :64] = 255
    c, _ = G.good_features_to_track(img, mask, max_corners=100, quality_level=0.01, min_distance=8.0)
    assert len(c) > 0 and np.all(c[:, 0] < 64)
    c3, _ = G.good_features_to_track(img, mask, max_corners=3, quality_level=0.01, min_distance=8.0)
    assert np.array_equal(c3, c[:3])
    assert len(G.good_features_to_track(img, np.zeros(img.shape, np.uint8))[0]) == 0
    flat = np.full((64, 64), 50, np.uint8)
    assert len(G.good_features_to_track(flat)[0]) == 0


def test_ties_prefer_the_higher_address():
    # two identical isolated blobs: equal responses; greaterThanPtr puts the higher address first
    img = np.full((64, 96), 30, np.uint8)
    img[20:24, 20:24] = 220
    img[20:24, 60:64] = 220
    c, eig = G.good_features_to_track(img, None, max_corners=2, quality_level=0.5, min_distance=1.0)
    assert len(c) == 2
    i0, i1 = c[0, 1] * 96 + c[0, 0], c[1, 1] * 96 + c[1, 0]
    assert eig[int(c[0, 1]), int(c[0, 0])] == eig[int(c[1, 1]), int(c[1, 0])] and i0 > i1


def test_block_size_and_harris_variants():
    """TrackerParams::GFFTParams: a larger box only adds neighbours to the sums (block 1 = the pixel's own products), and the Harris response is
    det - k trace^2 of the same sums"""
    rng = np.random.default_rng(8)
    g = rng.integers(0, 256, (40, 50)).astype(np.uint8)
    e1, e3, e5 = G.min_eigen_val(g, 1), G.min_eigen_val(g, 3), G.min_eigen_val(g, 5)
    assert e1.shape == e3.shape == e5.shape == g.shape
    assert np.abs(e1).max() < 1e-6                                   # rank-one structure tensor of a single pixel: smaller eigenvalue 0 (up to fp32 rounding)
    assert (e5[5:-5, 5:-5] * 25 >= e3[5:-5, 5:-5] * 9 - 1e-3).all()  # without the 1 / block_size^2 of the scaled products: more positive semi-definite terms, no smaller eigenvalue (Weyl)
    hr = G.min_eigen_val(g, 3, True, 0.04)
    # independent restatement in float64 from the same fp32 products
    ys, xs = np.arange(40), np.arange(50)
    gi = g.astype(np.int64)
    r = lambda i, n: np.where(np.mod(i, 2 * (n - 1)) >= n, 2 * (n - 1) - np.mod(i, 2 * (n - 1)), np.mod(i, 2 * (n - 1)))
    sy = gi[r(ys - 1, 40)] + 2 * gi + gi[r(ys + 1, 40)]; sx = gi[:, r(xs - 1, 50)] + 2 * gi + gi[:, r(xs + 1, 50)]
    dx = (sy[:, r(xs + 1, 50)] - sy[:, r(xs - 1, 50)]) / (4 * 3 * 255.0); dy = (sx[r(ys + 1, 40)] - sx[r(ys - 1, 40)]) / (4 * 3 * 255.0)
    box = lambda c: sum(c[r(ys + oy, 40)][:, r(xs + ox, 50)] for oy in (-1, 0, 1) for ox in (-1, 0, 1))
    a, b, c = box(dx * dx), box(dx * dy), box(dy * dy)
    assert np.abs(hr - (a * c - b * b - 0.04 * (a + c) ** 2)).max() < 1e-5
    c1, _ = G.good_features_to_track(g, None, 50, 0.01, 3.0, 3, True, 0.04)
    assert len(c1) > 5
