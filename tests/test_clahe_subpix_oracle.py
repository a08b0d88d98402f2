"""Properties of the CPU oracles of the detector's CLAHE pre-filter and sub-pixel refinement (oracle/clahe_oracle.py,
oracle/subpix_oracle.py; OpenCV is absent, so these are what pins the restatements down): CLAHE of a constant image is a
constant, its luts are monotone and end at 255, contrast goes up, sizes that are not multiples of 8 work; cornerSubPix moves an
integer corner of a synthetic checker crossing onto the true sub-pixel crossing, leaves a converged point alone, and returns the
initial corner when the estimate runs away."""
import numpy as np

from oracle import clahe_oracle as CO, subpix_oracle as SO


def test_clahe_luts_are_monotone_and_the_filter_raises_contrast():
    rng = np.random.default_rng(0)
    img = (60 + 40 * rng.random((480, 640)) + 30 * np.sin(np.arange(640) / 40.0)[None, :]).astype(np.uint8)
    luts = CO.tile_luts(img)
    assert luts.shape == (64, 256) and (np.diff(luts.astype(int), axis=1) >= 0).all() and (luts[:, -1] == 255).all()
    out = CO.clahe(img)
    assert out.shape == img.shape and out.dtype == np.uint8 and out.std() > 1.5 * img.std()
    flat = np.full((480, 640), 93, np.uint8)
    f = CO.clahe(flat)
    assert (f == f[0, 0]).all()
    # clip limit: no bin of a clipped histogram exceeds limit + redistribution
    area, clip = 80 * 60, max(int(2.0 * 80 * 60 / 256), 1)
    assert clip == 37
    odd = (255 * rng.random((50, 70))).astype(np.uint8)          # 70 = 8 * 8 + 6, 50 = 8 * 6 + 2: padded with reflect-101
    assert CO.clahe(odd).shape == (50, 70)


def _crossing(cx, cy, w=96, h=80, sharp=1.2):
    """two smooth edges crossing at (cx, cy): a saddle / checker corner"""
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float64)
    ang = 0.3
    u = (xs - cx) * np.cos(ang) + (ys - cy) * np.sin(ang)
    v = -(xs - cx) * np.sin(ang) + (ys - cy) * np.cos(ang)
    return np.clip(128 + 100 * np.tanh(u / sharp) * np.tanh(v / sharp), 0, 255).round().astype(np.uint8)


def test_corner_sub_pix_finds_the_crossing():
    for cx, cy in ((40.3, 33.7), (51.85, 40.1), (30.5, 30.5)):
        img = _crossing(cx, cy)
        start = np.array([[round(cx) + 1, round(cy) - 1]], np.float32)
        out, it = SO.corner_sub_pix(img, start)
        assert abs(out[0, 0] - cx) < 0.05 and abs(out[0, 1] - cy) < 0.05 and 1 <= it[0] <= 40, (out, it)
        again, it2 = SO.corner_sub_pix(img, out)
        assert np.abs(again - out).max() < 2e-3 and it2[0] <= 3
    # the guard: a start 5+ px from where the iteration ends keeps the initial point
    img = _crossing(40.0, 40.0)
    far = np.array([[47.0, 46.0]], np.float32)
    out, _ = SO.corner_sub_pix(img, far)
    assert np.array_equal(out, far) or np.abs(out - [40, 40]).max() < 0.1
    # near the border the replicate path is taken and the result stays inside the image
    out, _ = SO.corner_sub_pix(_crossing(3.4, 4.2), np.array([[3.0, 4.0]], np.float32))
    assert 0 <= out[0, 0] < 96 and 0 <= out[0, 1] < 80


def test_get_rect_sub_pix_paths_agree_in_the_interior():
    """the 8u -> 32f fast path (running `prev` term) and the four-tap path compute the same bilinear sample up to fp32 rounding"""
    rng = np.random.default_rng(3)
    img = (255 * rng.random((64, 64))).astype(np.uint8)
    P = SO.get_rect_sub_pix(img, (13, 13), (np.float32(30.37), np.float32(28.81)))
    cx, cy = 30.37 - 6.0, 28.81 - 6.0
    ix, iy = int(np.floor(cx)), int(np.floor(cy))
    a, b = cx - ix, cy - iy
    g = img.astype(np.float64)
    want = ((1 - a) * (1 - b) * g[iy:iy + 13, ix:ix + 13] + a * (1 - b) * g[iy:iy + 13, ix + 1:ix + 14] + (1 - a) * b * g[iy + 1:iy + 14, ix:ix + 13]
            + a * b * g[iy + 1:iy + 14, ix + 1:ix + 14])
    assert np.abs(P - want).max() < 1e-3
    assert SO.weights().shape == (11, 11) and abs(SO.weights()[5, 5] - 1.0) < 1e-7 and abs(SO.weights()[0, 5] - np.exp(-1.0)) < 1e-6


def test_other_windows_and_a_zero_zone():
    """SubPixelCornerRefinementParams::window_size / zero_zone are configuration fields (TrackerParams.cc:63-68): a larger or non-square window still
    finds the crossing, the zero zone only removes the centre's weights"""
    img = _crossing(30.37, 28.81)
    start = np.array([[31.0, 28.0]], np.float32)
    for win, win_h, zz in ((5, None, (-1, -1)), (7, None, (-1, -1)), (4, 8, (-1, -1)), (6, 6, (1, 1)), (5, 5, (0, 0))):
        out, it = SO.corner_sub_pix(img, start, win, win_h=win_h, zero_zone=zz)
        assert abs(float(out[0, 0]) - 30.37) < 0.15 and abs(float(out[0, 1]) - 28.81) < 0.15, (win, win_h, zz, out)
    m = SO.weights(6, 4, (1, 2))
    assert m.shape == (9, 13) and (m[2:7, 5:8] == 0).all() and m[1, 6] > 0 and m[4, 4] > 0
    assert (SO.weights(5, 5, (5, 5)) > 0).all()                     # a zero zone as large as the window is ignored (cornersubpix.cpp)
