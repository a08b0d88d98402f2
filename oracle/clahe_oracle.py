"""CPU oracle of the detector's CLAHE pre-filter (TEST INFRASTRUCTURE ONLY - never imported by the product).

Restates cv::CLAHE::apply for 8-bit images as the reference's detector runs it (dynosam/src/frontend/vision/FeatureDetector.cc:186-199:
`clahe_ = cv::createCLAHE(2.0, cv::Size(8, 8))`, `clahe_->apply(processed_image, processed_image)` when
TrackerParams::use_clahe_filter, default true - TrackerParams.hpp:101):

  tiles      8 x 8; an image whose sides are not multiples of 8 is extended to the right / bottom with BORDER_REFLECT_101 for the
             histograms only
  per tile   256-bin histogram; clipLimit = max(1, int(2.0 * tileArea / 256)); the excess over the limit is redistributed:
             every bin += excess / 256, then the first `excess % 256` bins met with stride max(256 / residual, 1) get one more
  lut        lut[i] = saturate_cast<uchar>(cumsum(hist)[i] * lutScale), lutScale = 255.f / tileArea (fp32, round half to even)
  output     bilinear blend of the four surrounding tiles' luts: tile coordinate x / tileWidth - 0.5 (fp32), indices clamped,
             res = (l11 * xa1 + l12 * xa) * ya1 + (l21 * xa1 + l22 * xa) * ya, saturate_cast<uchar>(res)

The arithmetic lives in OpenCV 4.10.0 (docker/Dockerfile.amd64:67-93; not in /root/reference, no cv2 in this image): PARITY UNPINNED
against the OpenCV binary [modules/imgproc/src/clahe.cpp, recalled].  Every fp32 operation is spelt out (one rounding each) and the
device kernels follow the same order, so the comparison with the GPU is BIT-EXACT.
"""
from __future__ import annotations

import numpy as np

f32 = np.float32


def _reflect101(i, n):
    p = 2 * (n - 1)
    i = np.mod(i, p)
    return np.where(i >= n, p - i, i)


def tile_luts(gray: np.ndarray, clip_limit: float = 2.0, tiles=(8, 8)) -> np.ndarray:
    """[tilesY * tilesX, 256] u8"""
    h, w = gray.shape
    tx, ty = tiles
    if w % tx or h % ty:
        # (cv::copyMakeBorder(src, srcExt, 0, tilesY - rows % tilesY, 0, tilesX - cols % tilesX): a side that IS a multiple gets a whole
        #  extra tile row / column in OpenCV; restated literally)
        ys = _reflect101(np.arange(h + (ty - h % ty)), h)
        xs = _reflect101(np.arange(w + (tx - w % tx)), w)
        src = gray[ys][:, xs]
    else:
        src = gray
    hh, ww = src.shape
    tw, th = ww // tx, hh // ty
    area = tw * th
    lut_scale = f32(255.0) / f32(area)
    clip = max(int(clip_limit * area / 256), 1) if clip_limit > 0 else 0
    luts = np.zeros((ty * tx, 256), np.uint8)
    for k in range(ty * tx):
        r, c = k // tx, k % tx
        hist = np.bincount(src[r * th:(r + 1) * th, c * tw:(c + 1) * tw].ravel(), minlength=256).astype(np.int64)
        if clip > 0:
            clipped = int(np.maximum(hist - clip, 0).sum())
            hist = np.minimum(hist, clip)
            batch, residual = clipped // 256, clipped % 256
            hist += batch
            if residual:
                step = max(256 // residual, 1)
                i = 0
                while i < 256 and residual > 0:
                    hist[i] += 1
                    i += step
                    residual -= 1
        s = np.cumsum(hist)
        v = (s.astype(f32) * lut_scale).astype(f32)
        luts[k] = np.clip(np.rint(v), 0, 255).astype(np.uint8)      # cvRound: to nearest, ties to even
    return luts


def clahe(gray: np.ndarray, clip_limit: float = 2.0, tiles=(8, 8)) -> np.ndarray:
    gray = np.ascontiguousarray(gray, np.uint8)
    h, w = gray.shape
    tx, ty = tiles
    luts = tile_luts(gray, clip_limit, tiles)
    ww = w if (w % tx == 0 and h % ty == 0) else w + (tx - w % tx)
    hh = h if (w % tx == 0 and h % ty == 0) else h + (ty - h % ty)
    tw, th = ww // tx, hh // ty
    inv_tw, inv_th = f32(1.0) / f32(tw), f32(1.0) / f32(th)
    xf = (np.arange(w).astype(f32) * inv_tw - f32(0.5)).astype(f32)
    yf = (np.arange(h).astype(f32) * inv_th - f32(0.5)).astype(f32)
    x1, y1 = np.floor(xf).astype(np.int64), np.floor(yf).astype(np.int64)
    xa, ya = (xf - x1.astype(f32)).astype(f32), (yf - y1.astype(f32)).astype(f32)
    xa1, ya1 = (f32(1.0) - xa).astype(f32), (f32(1.0) - ya).astype(f32)
    x2, y2 = np.minimum(x1 + 1, tx - 1), np.minimum(y1 + 1, ty - 1)
    x1, y1 = np.maximum(x1, 0), np.maximum(y1, 0)
    v = gray.astype(np.int64)
    L = luts.astype(f32)
    l11 = L[(y1[:, None] * tx + x1[None, :]), v]
    l12 = L[(y1[:, None] * tx + x2[None, :]), v]
    l21 = L[(y2[:, None] * tx + x1[None, :]), v]
    l22 = L[(y2[:, None] * tx + x2[None, :]), v]
    top = ((l11 * xa1[None, :]).astype(f32) + (l12 * xa[None, :]).astype(f32)).astype(f32)
    bot = ((l21 * xa1[None, :]).astype(f32) + (l22 * xa[None, :]).astype(f32)).astype(f32)
    res = ((top * ya1[:, None]).astype(f32) + (bot * ya[:, None]).astype(f32)).astype(f32)
    return np.clip(np.rint(res), 0, 255).astype(np.uint8)
