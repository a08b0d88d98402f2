"""CPU oracle of the detector's sub-pixel corner refinement (TEST INFRASTRUCTURE ONLY - never imported by the product).

Restates cv::cornerSubPix as the reference's detector runs it (dynosam/src/frontend/vision/FeatureDetector.cc:224-238 with
TrackerParams::SubPixelCornerRefinementParams, TrackerParams.hpp:64-69: window (5, 5) -> 11 x 11, zero zone (-1, -1),
TermCriteria(EPS + COUNT, 40, 0.001); use_subpixel_corner_refinement defaults to true, :99):

  weights   w(i, j) = exp(-y^2) exp(-x^2), x = (j - 5) / 5, y = (i - 5) / 5  (fp32)
  iterate   13 x 13 fp32 patch around the current estimate by cv::getRectSubPix (8-bit source, bilinear; the 8u -> 32f fast path
            with its running `prev` term when the patch lies inside the image, the replicate-border generic path otherwise);
            central differences gx, gy on the inner 11 x 11; sums (fp64, row-major) a = S w gx^2, b = S w gx gy, c = S w gy^2,
            bb1 = S (w gx^2 px + w gx gy py), bb2 = S (w gx gy px + w gy^2 py);  new = cur + [c bb1 - b bb2, -b bb1 + a bb2] / det
            until 40 iterations, a step of squared length <= 1e-6, det ~ 0, or the estimate leaves the image
  guard     a result further than the half window (5 px) from the start in x or y is discarded: the initial corner stays

The arithmetic lives in OpenCV 4.10.0 (docker/Dockerfile.amd64:67-93; not in /root/reference, no cv2 in this image): PARITY UNPINNED
against the OpenCV binary [modules/imgproc/src/{cornersubpix,samplers}.cpp, recalled].  One stated deviation: the six distinct
weights exp(-t) are formed as float(exp(double(t))) instead of expf(t) (so that this file and the device agree whatever libm's
expf does in the last bit).  Every operation is spelt out with one rounding each and the device kernel follows the same order:
the comparison with the GPU is BIT-EXACT.
"""
from __future__ import annotations

import math

import numpy as np

f32 = np.float32


def weights(win=5, win_h=None, zero_zone=(-1, -1)):
    """mask[i][j] of cv::cornerSubPix: window half sizes (win, win_h) = (width, height), the zero zone's centre block set to 0"""
    ww, wh = win, (win if win_h is None else win_h)
    def axis(w):
        v = np.zeros(2 * w + 1, f32)
        for i in range(2 * w + 1):
            y = f32(f32(i - w) / f32(w))
            v[i] = f32(math.exp(float(f32(-y * y))))
        return v
    m = (axis(wh)[:, None] * axis(ww)[None, :]).astype(f32)          # mask[i][j] = (float)(vy * exp(-x*x))
    zw, zh = zero_zone
    if zw >= 0 and zh >= 0 and zw * 2 + 1 < 2 * ww + 1 and zh * 2 + 1 < 2 * wh + 1:
        m[wh - zh:wh + zh + 1, ww - zw:ww + zw + 1] = 0
    return m


def get_rect_sub_pix(img: np.ndarray, size, center) -> np.ndarray:
    """cv::getRectSubPix(8U -> 32F), size = (w, h), center = (x, y) fp32"""
    H, W = img.shape
    ww, wh = size
    cx = f32(f32(center[0]) - f32(f32(ww - 1) * f32(0.5)))
    cy = f32(f32(center[1]) - f32(f32(wh - 1) * f32(0.5)))
    ipx, ipy = int(math.floor(cx)), int(math.floor(cy))
    a, b = f32(cx - f32(ipx)), f32(cy - f32(ipy))
    out = np.zeros((wh, ww), f32)
    if 0 <= ipx and ipx + ww < W and 0 <= ipy and ipy + wh < H:
        a = max(a, f32(0.0001))
        a12, a22 = f32(a * f32(f32(1.0) - b)), f32(a * b)
        b1, b2 = f32(f32(1.0) - b), b
        s = (1.0 - float(a)) / float(a)
        A = img[ipy:ipy + wh + 1, ipx:ipx + ww + 1].astype(f32)
        prev = (f32(f32(1.0) - a) * ((b1 * A[:-1, 0]).astype(f32) + (b2 * A[1:, 0]).astype(f32)).astype(f32)).astype(f32)
        for j in range(ww):
            t = ((a12 * A[:-1, j + 1]).astype(f32) + (a22 * A[1:, j + 1]).astype(f32)).astype(f32)
            out[:, j] = (prev + t).astype(f32)
            prev = (t.astype(np.float64) * s).astype(f32)
        return out
    a11, a12 = f32(f32(f32(1.0) - a) * f32(f32(1.0) - b)), f32(a * f32(f32(1.0) - b))
    a21, a22 = f32(f32(f32(1.0) - a) * b), f32(a * b)
    b1, b2 = f32(f32(1.0) - b), b
    rx = min(max(-ipx, 0), ww)
    rw = ww if ipx < W - ww else max(W - ipx - 1, 0)
    g = img.astype(f32)
    for i in range(wh):
        ya, yb = min(max(ipy + i, 0), H - 1), min(max(ipy + i + 1, 0), H - 1)
        for j in range(ww):
            if rx <= j < rw:
                x0 = ipx + j
                out[i, j] = f32(f32(f32(f32(g[ya, x0] * a11) + f32(g[ya, x0 + 1] * a12)) + f32(g[yb, x0] * a21)) + f32(g[yb, x0 + 1] * a22))
            else:
                xc = min(max(ipx + j, 0), W - 1)
                out[i, j] = f32(f32(g[ya, xc] * b1) + f32(g[yb, xc] * b2))
    return out


def corner_sub_pix(img: np.ndarray, corners, win=5, max_count=40, epsilon=0.001, win_h=None, zero_zone=(-1, -1)):
    """returns ([n, 2] f32 refined corners, [n] iterations used); (win, win_h) = SubPixelCornerRefinementParams::window_size (half sizes)"""
    img = np.ascontiguousarray(img, np.uint8)
    H, W = img.shape
    wh = win if win_h is None else win_h
    wnx, wny = 2 * win + 1, 2 * wh + 1
    mask = weights(win, wh, zero_zone).astype(np.float64)
    px = (np.arange(wnx) - win).astype(np.float64)[None, :] * np.ones((wny, 1))
    py = (np.arange(wny) - wh).astype(np.float64)[:, None] * np.ones((1, wnx))
    eps = max(epsilon, 0.0) ** 2
    max_iters = min(max(max_count, 1), 100)
    out = np.array(corners, f32).reshape(-1, 2).copy()
    iters = np.zeros(len(out), np.int32)
    seq = lambda v: float(np.cumsum(v.ravel())[-1])          # sequential (row-major) fp64 accumulation
    for k in range(len(out)):
        cT = (f32(out[k, 0]), f32(out[k, 1]))
        cI = cT
        it = 0
        while True:
            P = get_rect_sub_pix(img, (wnx + 2, wny + 2), cI)
            tgx = (P[1:-1, 2:] - P[1:-1, :-2]).astype(f32).astype(np.float64)
            tgy = (P[2:, 1:-1] - P[:-2, 1:-1]).astype(f32).astype(np.float64)
            gxx, gxy, gyy = tgx * tgx * mask, tgx * tgy * mask, tgy * tgy * mask
            a, b, c = seq(gxx), seq(gxy), seq(gyy)
            bb1, bb2 = seq(gxx * px + gxy * py), seq(gxy * px + gyy * py)
            det = a * c - b * b
            if abs(det) <= np.finfo(np.float64).eps ** 2:
                break
            scale = 1.0 / det
            nx = f32(float(cI[0]) + c * scale * bb1 - b * scale * bb2)
            ny = f32(float(cI[1]) - b * scale * bb1 + a * scale * bb2)
            dx, dy = f32(nx - cI[0]), f32(ny - cI[1])
            err = float(f32(f32(dx * dx) + f32(dy * dy)))
            cI = (nx, ny)
            if cI[0] < 0 or cI[0] >= W or cI[1] < 0 or cI[1] >= H:
                break
            it += 1
            if not (it < max_iters and err > eps):
                break
        iters[k] = it
        if abs(f32(cI[0] - cT[0])) > win or abs(f32(cI[1] - cT[1])) > wh:
            cI = cT
        out[k] = cI
    return out, iters
