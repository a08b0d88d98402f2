"""CPU oracle of the Shi-Tomasi corner detector (TEST INFRASTRUCTURE ONLY - never imported by the product).

Restates cv::goodFeaturesToTrack as the reference's detector runs it (dynosam/src/frontend/vision/FeatureDetector.cc:58-89,
`cv::cuda::createGoodFeaturesToTrackDetector` / :96-111 `cv::GFTTDetector::create`; parameters
TrackerParams.hpp:74-75,108,111: quality_level 0.001, block_size 3, max 2000 corners, min distance 8, Harris off):

  eig = cornerMinEigenVal(img, blockSize 3, Sobel aperture 3)       Sobel * 1/(4*3*255), un-normalised 3x3 box sums of
                                                                     (dx^2, dx dy, dy^2), (a+c) - sqrt((a-c)^2 + b^2) with
                                                                     a = xx/2, b = xy, c = yy/2
  maxVal = max of eig over the mask;  eig = eig > maxVal*quality ? eig : 0
  candidates: interior pixels with eig != 0, eig == max over the 3x3 neighbourhood, mask != 0
  sorted by descending response (ties: higher address first), greedy minimum-distance filter on a cell grid, first maxCorners

The arithmetic lives in OpenCV 4.10.0 (docker/Dockerfile.amd64:67-93; not in /root/reference, no cv2 in this image):
PARITY UNPINNED against the OpenCV binary [algorithm of modules/imgproc/src/{corner,featureselect}.cpp recalled]. OpenCV's
separable-filter float summation order is an implementation detail; here every fp32 operation is spelt out (Sobel sum as an
exact integer times the fp32 scale, box sum row-major) and the device kernels follow the same order, so the comparison
with the GPU is BIT-EXACT.
"""
from __future__ import annotations

import numpy as np

f32 = np.float32


def _reflect101(i, n):
    p = 2 * (n - 1)
    i = np.mod(i, p)
    return np.where(i >= n, p - i, i)


def min_eigen_val(gray: np.ndarray, block_size: int = 3, use_harris: bool = False, k: float = 0.04) -> np.ndarray:
    """cornerMinEigenVal (or, use_harris, cornerHarris) with Sobel aperture 3 and a block_size x block_size box (anchor block_size // 2)"""
    h, w = gray.shape
    g = gray.astype(np.int64)
    ys, xs = np.arange(h), np.arange(w)
    yu, yd, xl, xr = _reflect101(ys - 1, h), _reflect101(ys + 1, h), _reflect101(xs - 1, w), _reflect101(xs + 1, w)
    sm_y = g[yu] + 2 * g + g[yd]            # smoothing along y
    sm_x = g[:, xl] + 2 * g + g[:, xr]
    dxi = sm_y[:, xr] - sm_y[:, xl]
    dyi = sm_x[yd] - sm_x[yu]
    scale = f32(1.0 / (4.0 * float(block_size) * 255.0))
    dx, dy = dxi.astype(f32) * scale, dyi.astype(f32) * scale
    cxx, cxy, cyy = dx * dx, dx * dy, dy * dy
    a0 = block_size // 2
    def box(c):
        acc = np.zeros((h, w), f32)
        for oy in range(-a0, block_size - a0):
            for ox in range(-a0, block_size - a0):
                acc = (acc + c[_reflect101(ys + oy, h)][:, _reflect101(xs + ox, w)]).astype(f32)
        return acc
    if use_harris:      # calcHarris: a c - b^2 - k (a + c)^2, every operation in fp32
        a, b, c = box(cxx), box(cxy), box(cyy)
        s = a + c
        return ((a * c - b * b) - (f32(k) * s) * s).astype(f32)
    a, b, c = box(cxx) * f32(0.5), box(cxy), box(cyy) * f32(0.5)
    amc = a - c
    return ((a + c) - np.sqrt(amc * amc + b * b, dtype=f32)).astype(f32)


def good_features_to_track(gray, mask=None, max_corners=2000, quality_level=0.001, min_distance=8.0, block_size=3, use_harris=False, k=0.04):
    h, w = gray.shape
    eig = min_eigen_val(gray, block_size, use_harris, k)
    m = np.ones((h, w), bool) if mask is None else (np.asarray(mask) != 0)
    if not m.any():
        return np.zeros((0, 2), f32), eig
    max_val = float(eig[m].max())
    thr = f32(max_val * quality_level)
    t = np.where(eig > thr, eig, f32(0))
    dil = t.copy()
    for oy in (-1, 0, 1):
        for ox in (-1, 0, 1):
            sh = np.full_like(t, -np.inf)
            ys0, ys1 = max(0, -oy), min(h, h - oy)
            xs0, xs1 = max(0, -ox), min(w, w - ox)
            sh[ys0:ys1, xs0:xs1] = t[ys0 + oy:ys1 + oy, xs0 + ox:xs1 + ox]
            dil = np.maximum(dil, sh)
    cand = (t != 0) & (t == dil) & m
    cand[0, :] = cand[-1, :] = False
    cand[:, 0] = cand[:, -1] = False
    idx = np.flatnonzero(cand)
    vals = t.ravel()[idx]
    order = np.lexsort((-idx, -vals.astype(np.float64)))      # response descending, ties: higher address first
    idx = idx[order]
    out = []
    if min_distance >= 1:
        cell = int(np.rint(min_distance))
        gw, gh = (w + cell - 1) // cell, (h + cell - 1) // cell
        grid = [[] for _ in range(gw * gh)]
        md2 = f32(min_distance * min_distance)
        for i in idx:
            y, x = int(i // w), int(i % w)
            xc, yc = x // cell, y // cell
            good = True
            for yy in range(max(0, yc - 1), min(gh - 1, yc + 1) + 1):
                for xx in range(max(0, xc - 1), min(gw - 1, xc + 1) + 1):
                    for (px, py) in grid[yy * gw + xx]:
                        ddx, ddy = f32(x) - px, f32(y) - py
                        if f32(ddx * ddx + ddy * ddy) < md2:
                            good = False
                            break
                    if not good:
                        break
                if not good:
                    break
            if good:
                grid[yc * gw + xc].append((f32(x), f32(y)))
                out.append((x, y))
                if max_corners > 0 and len(out) == max_corners:
                    break
    else:
        for i in idx[:max_corners if max_corners > 0 else None]:
            out.append((int(i % w), int(i // w)))
    return np.array(out, f32).reshape(-1, 2), eig
