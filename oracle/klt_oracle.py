"""CPU oracle of the sparse pyramidal Lucas-Kanade tracker (TEST INFRASTRUCTURE ONLY - never imported by the product).

Restates cv::calcOpticalFlowPyrLK the way KltFeatureTracker::trackPoints drives it
(dynosam/src/frontend/vision/StaticFeatureTracker.cc:447-534): forward pass 21x21 window, maxLevel 3, criteria
30 iterations / eps 0.03, optional OPTFLOW_USE_INITIAL_FLOW; reverse pass from the tracked points with maxLevel 5 and the
default criteria 30 / 0.01; a track survives iff both passes succeed and the reverse pass lands within 0.5 px.

The arithmetic lives in OpenCV 4.10.0 (docker/Dockerfile.amd64:67-93), which is NOT in /root/reference and cannot be
imported here (no cv2): PARITY UNPINNED against the OpenCV binary.  What is restated is the published algorithm of
modules/video/src/lkpyramid.cpp [recalled]: 8-bit grey pyramid by the 5-tap binomial pyrDown with rounding, int16 Scharr
derivatives with reflect-101 inside / zero outside the image, W_BITS = 14 fixed-point bilinear taps, the 2x2 normal
matrix with the min-eigenvalue test (1e-4), the Newton iteration with the 0.01 px oscillation stop, and the level-0 bounds
test that clears the status.  Sums over the window are accumulated as exact integers (OpenCV's scalar path accumulates
float products, its SIMD paths int32 lanes - the summation order is not part of the algorithm), every other float
operation is IEEE fp32 in the source order, so the device kernel can be compared BIT-EXACTLY with this file.
"""
from __future__ import annotations

import numpy as np

W_BITS = 14
WIN = 21
HALF = np.float32((WIN - 1) * 0.5)
FLT_SCALE = np.float32(1.0 / (1 << 20))
FLT_EPSILON = np.float32(1.1920929e-07)
MIN_EIG = np.float32(1e-4)
f32 = np.float32


def gray_u8(rgb: np.ndarray) -> np.ndarray:
    """cv::cvtColor RGB2GRAY on 8-bit data: (R*4899 + G*9617 + B*1868 + 8192) >> 14."""
    r, g, b = (rgb[..., k].astype(np.int64) for k in range(3))
    return ((r * 4899 + g * 9617 + b * 1868 + (1 << 13)) >> 14).astype(np.uint8)


def _reflect101(i: np.ndarray, n: int) -> np.ndarray:
    if n == 1:
        return np.zeros_like(i)
    p = 2 * (n - 1)
    i = np.mod(i, p)
    return np.where(i >= n, p - i, i)


def pyr_down(img: np.ndarray) -> np.ndarray:
    """cv::pyrDown on u8: separable [1 4 6 4 1], BORDER_REFLECT_101, (sum + 128) >> 8, size (w+1)/2 x (h+1)/2."""
    h, w = img.shape
    oh, ow = (h + 1) // 2, (w + 1) // 2
    src = img.astype(np.int64)
    k = np.array([1, 4, 6, 4, 1], dtype=np.int64)
    xs = 2 * np.arange(ow)[:, None] + np.arange(-2, 3)[None, :]
    ys = 2 * np.arange(oh)[:, None] + np.arange(-2, 3)[None, :]
    rows = (src[:, _reflect101(xs, w)] * k).sum(-1)          # h x ow
    out = (rows[_reflect101(ys, h), :] * k[None, :, None]).sum(1)
    return ((out + 128) >> 8).astype(np.uint8)


def build_pyramid(img: np.ndarray, max_level: int):
    """cv::buildOpticalFlowPyramid's level rule: stop before a level that is not larger than the window."""
    pyr = [img]
    for _ in range(max_level):
        h, w = pyr[-1].shape
        if (w + 1) // 2 <= WIN or (h + 1) // 2 <= WIN:
            break
        pyr.append(pyr_down(pyr[-1]))
    return pyr


def scharr(img: np.ndarray):
    """calcSharrDeriv: dx = t0[x+1] - t0[x-1], t0 = 3 up + 10 mid + 3 down; dy = 3 t1[x-1] + 10 t1[x] + 3 t1[x+1], t1 = down - up."""
    h, w = img.shape
    s = img.astype(np.int64)
    up, dn = s[_reflect101(np.arange(h) - 1, h)], s[_reflect101(np.arange(h) + 1, h)]
    t0 = (up + dn) * 3 + s * 10
    t1 = dn - up
    xl, xr = _reflect101(np.arange(w) - 1, w), _reflect101(np.arange(w) + 1, w)
    dx = t0[:, xr] - t0[:, xl]
    dy = (t1[:, xl] + t1[:, xr]) * 3 + t1 * 10
    return dx, dy


def _sample_img(img: np.ndarray, ys: np.ndarray, xs: np.ndarray) -> np.ndarray:
    h, w = img.shape
    return img[_reflect101(ys, h)[:, None], _reflect101(xs, w)[None, :]].astype(np.int64)


def _sample_deriv(d: np.ndarray, ys: np.ndarray, xs: np.ndarray) -> np.ndarray:
    h, w = d.shape
    ok = ((ys >= 0) & (ys < h))[:, None] & ((xs >= 0) & (xs < w))[None, :]
    return np.where(ok, d[np.clip(ys, 0, h - 1)[:, None], np.clip(xs, 0, w - 1)[None, :]], 0)


def _weights(fx: np.float32, fy: np.float32, ix: int, iy: int):
    a, b = f32(fx - f32(ix)), f32(fy - f32(iy))
    one = f32(1.0)
    sc = f32(1 << W_BITS)
    iw00 = int(np.rint(f32(f32(f32(one - a) * f32(one - b)) * sc)))
    iw01 = int(np.rint(f32(f32(a * f32(one - b)) * sc)))
    iw10 = int(np.rint(f32(f32(f32(one - a) * b) * sc)))
    return iw00, iw01, iw10, (1 << W_BITS) - iw00 - iw01 - iw10


def _descale(x, n):
    return (x + (1 << (n - 1))) >> n


def _bilinear(tap, w4, n):
    iw00, iw01, iw10, iw11 = w4
    return _descale(tap[:-1, :-1] * iw00 + tap[:-1, 1:] * iw01 + tap[1:, :-1] * iw10 + tap[1:, 1:] * iw11, n)


def lk_level(I, dIx, dIy, J, prev_pt, next_pt, level, max_count, eps2):
    """One point, one level.  Returns (next_pt, status_cleared)."""
    h, w = I.shape
    px, py = f32(prev_pt[0] - HALF), f32(prev_pt[1] - HALF)
    ix, iy = int(np.floor(px)), int(np.floor(py))
    if ix < -WIN or ix >= w or iy < -WIN or iy >= h:
        return next_pt, level == 0
    w4 = _weights(px, py, ix, iy)
    ys, xs = iy + np.arange(WIN + 1), ix + np.arange(WIN + 1)
    Iw = _bilinear(_sample_img(I, ys, xs), w4, W_BITS - 5)
    Ixw = _bilinear(_sample_deriv(dIx, ys, xs), w4, W_BITS)
    Iyw = _bilinear(_sample_deriv(dIy, ys, xs), w4, W_BITS)
    A11 = f32(f32(int((Ixw * Ixw).sum())) * FLT_SCALE)
    A12 = f32(f32(int((Ixw * Iyw).sum())) * FLT_SCALE)
    A22 = f32(f32(int((Iyw * Iyw).sum())) * FLT_SCALE)
    D = f32(f32(A11 * A22) - f32(A12 * A12))
    dd = f32(A11 - A22)
    disc = f32(np.sqrt(f32(f32(dd * dd) + f32(f32(4.0) * f32(A12 * A12)))))
    min_eig = f32(f32(f32(A22 + A11) - disc) / f32(2 * WIN * WIN))
    if min_eig < MIN_EIG or D < FLT_EPSILON:
        return next_pt, level == 0
    D = f32(f32(1.0) / D)
    nx, ny = f32(next_pt[0] - HALF), f32(next_pt[1] - HALF)
    pdx = pdy = f32(0.0)
    cleared = False
    out = (f32(next_pt[0]), f32(next_pt[1]))
    for j in range(max_count):
        jx, jy = int(np.floor(nx)), int(np.floor(ny))
        if jx < -WIN or jx >= w or jy < -WIN or jy >= h:
            cleared = level == 0
            break
        wj = _weights(nx, ny, jx, jy)
        Jw = _bilinear(_sample_img(J, jy + np.arange(WIN + 1), jx + np.arange(WIN + 1)), wj, W_BITS - 5)
        diff = Jw - Iw
        b1 = f32(f32(int((diff * Ixw).sum())) * FLT_SCALE)
        b2 = f32(f32(int((diff * Iyw).sum())) * FLT_SCALE)
        dx = f32(f32(f32(A12 * b2) - f32(A22 * b1)) * D)
        dy = f32(f32(f32(A12 * b1) - f32(A11 * b2)) * D)
        nx, ny = f32(nx + dx), f32(ny + dy)
        out = (f32(nx + HALF), f32(ny + HALF))
        if f32(f32(dx * dx) + f32(dy * dy)) <= eps2:
            break
        if j > 0 and abs(f32(dx + pdx)) < f32(0.01) and abs(f32(dy + pdy)) < f32(0.01):
            out = (f32(out[0] - f32(dx * f32(0.5))), f32(out[1] - f32(dy * f32(0.5))))
            break
        pdx, pdy = dx, dy
    if level == 0 and not cleared:
        fx, fy = f32(out[0] - HALF), f32(out[1] - HALF)
        jx, jy = int(np.floor(fx)), int(np.floor(fy))
        if jx < -WIN or jx >= w or jy < -WIN or jy >= h:
            cleared = True
    return out, cleared


def calc_pyr_lk(prev_gray, next_gray, prev_pts, init_pts=None, max_level=3, max_count=30, eps=0.03):
    """cv::calcOpticalFlowPyrLK(prev, next, prev_pts, next_pts, status, err, Size(21,21), max_level, criteria, flags)."""
    pI, pJ = build_pyramid(prev_gray, max_level), build_pyramid(next_gray, max_level)
    top = len(pI) - 1
    derivs = [scharr(im) for im in pI]
    eps2 = f32(f32(eps) * f32(eps))
    n = len(prev_pts)
    nxt = np.zeros((n, 2), np.float32)
    status = np.ones(n, np.uint8)
    for i in range(n):
        cur = (f32(0), f32(0))
        for level in range(top, -1, -1):
            sc = f32(1.0 / (1 << level))
            pp = (f32(f32(prev_pts[i][0]) * sc), f32(f32(prev_pts[i][1]) * sc))
            if level == top:
                cur = (f32(f32(init_pts[i][0]) * sc), f32(f32(init_pts[i][1]) * sc)) if init_pts is not None else pp
            else:
                cur = (f32(cur[0] * f32(2.0)), f32(cur[1] * f32(2.0)))
            cur, cleared = lk_level(pI[level], derivs[level][0], derivs[level][1], pJ[level], pp, cur, level, max_count, eps2)
            if cleared:
                status[i] = 0
        nxt[i] = cur
    return nxt, status


def track_points(prev_gray, next_gray, prev_pts, init_pts=None, min_success=10):
    """KltFeatureTracker::trackPoints' optical-flow part (StaticFeatureTracker.cc:447-534)."""
    prev_pts = np.asarray(prev_pts, np.float32).reshape(-1, 2)
    cur, st = calc_pyr_lk(prev_gray, next_gray, prev_pts, init_pts, 3, 30, 0.03)
    if init_pts is not None and int(st.sum()) < min_success:
        cur, st = calc_pyr_lk(prev_gray, next_gray, prev_pts, None, 3, 30, 0.03)
    back, rst = calc_pyr_lk(next_gray, prev_gray, cur, None, 5, 30, 0.01)
    dx = (prev_pts[:, 0] - back[:, 0]).astype(np.float32)
    dy = (prev_pts[:, 1] - back[:, 1]).astype(np.float32)
    dist = np.sqrt((dx * dx + dy * dy).astype(np.float32)).astype(np.float32)
    good = (st != 0) & (rst != 0) & (dist <= np.float32(0.5))
    return cur, back, good.astype(np.uint8), st


def _eigen_quat_w(R: np.ndarray) -> float:
    """w of Eigen::Quaterniond(R) (Eigen/src/Geometry/Quaternion.h, quaternionbase_assign_impl<Mat, 3, 3>), which is what
    gtsam::Rot3::toQuaternion() returns: t = trace; t > 0: w = 0.5 sqrt(t + 1); else from the largest diagonal entry"""
    R = np.asarray(R, np.float64).reshape(3, 3)
    t = R[0, 0] + R[1, 1] + R[2, 2]
    if t > 0.0:
        return 0.5 * np.sqrt(t + 1.0)
    i = 0
    if R[1, 1] > R[0, 0]:
        i = 1
    if R[2, 2] > R[i, i]:
        i = 2
    j, k = (i + 1) % 3, (i + 2) % 3
    tt = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0)
    return (R[k, j] - R[j, k]) * (0.5 / tt)


def _matx33f_mul(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """cv::Matx33f * cv::Matx33f: every entry a float accumulator over k = 0, 1, 2, one rounding per operation"""
    c = np.zeros((3, 3), np.float32)
    for i in range(3):
        for j in range(3):
            s = np.float32(0)
            for k in range(3):
                s = np.float32(s + np.float32(a[i, k] * b[k, j]))
            c[i, j] = s
    return c


def _eigen_inverse3(m: np.ndarray) -> np.ndarray:
    """Eigen::Matrix3d::inverse() (Eigen/src/LU/InverseImpl.h, compute_inverse_size3_helper): cyclic cofactors, the determinant along
    column 0, every entry cofactor * (1 / det)"""
    def cof(i, j):
        i1, i2, j1, j2 = (i + 1) % 3, (i + 2) % 3, (j + 1) % 3, (j + 2) % 3
        return m[i1, j1] * m[i2, j2] - m[i1, j2] * m[i2, j1]
    det = (cof(0, 0) * m[0, 0] + cof(1, 0) * m[1, 0]) + cof(2, 0) * m[2, 0]
    inv_det = 1.0 / det
    out = np.zeros((3, 3))
    for i in range(3):
        for j in range(3):
            out[j, i] = cof(i, j) * inv_det
    return out


def rotation_homography(R_km1_k: np.ndarray, K: np.ndarray) -> np.ndarray:
    """H = K_cv * R * K_inv_cv of FeatureTrackerBase::predictKeypointsGivenRotation (dynosam/src/frontend/vision/FeatureTrackerBase.cc:63-72):
    K inverted in double (Eigen), then everything cast to float and multiplied as cv::Matx33f, left to right"""
    K = np.asarray(K, np.float64).reshape(3, 3)
    Kf, Rf, Kif = K.astype(np.float32), np.asarray(R_km1_k, np.float64).reshape(3, 3).astype(np.float32), _eigen_inverse3(K).astype(np.float32)
    return _matx33f_mul(_matx33f_mul(Kf, Rf), Kif)


def predict_keypoints_given_rotation(pts_km1: np.ndarray, R_km1_k: np.ndarray, K: np.ndarray, width: int, height: int, shrink_row: int = 0, shrink_col: int = 0) -> np.ndarray:
    """FeatureTrackerBase::predictKeypointsGivenRotation (FeatureTrackerBase.cc:50-105), float32 as the original: a rotation whose quaternion has
    |1 - |w|| < 1e-4 copies the points; else p2 = H (x, y, 1), re-homogenised when p2.z > 0 (the previous point otherwise), and kept only if
    it lies within the shrunken image (isWithinShrunkenImage, :313-326: the coordinates truncated to int) - the previous point otherwise"""
    pts = np.ascontiguousarray(pts_km1, np.float32).reshape(-1, 2)
    if abs(1.0 - abs(_eigen_quat_w(R_km1_k))) < 1e-4:
        return pts.copy()
    H = rotation_homography(R_km1_k, K)
    out = pts.copy()
    one = np.float32(1)
    for i, (x, y) in enumerate(pts):
        p2 = [np.float32(np.float32(np.float32(np.float32(H[r, 0] * x) + np.float32(H[r, 1] * y)) + np.float32(H[r, 2] * one))) for r in range(3)]
        if p2[2] > np.float32(0):
            nx, ny = np.float32(p2[0] / p2[2]), np.float32(p2[1] / p2[2])
        else:
            nx, ny = x, y
        col, row = int(np.float64(nx)), int(np.float64(ny))          # Keypoint is a double vector; u(), v() assigned to int truncate
        if row > shrink_row and row < height - shrink_row and col > shrink_col and col < width - shrink_col:
            out[i] = (nx, ny)
    return out
