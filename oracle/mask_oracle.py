"""CPU oracle of the object boundary mask (TEST INFRASTRUCTURE ONLY).

Restates vision_tools::computeObjectMaskBoundaryMask (dynosam/src/frontend/vision/VisionTools.cc:361-449) with
findObjectBoundingBox (:285-322), as FeatureTracker::objectDetection drives it (FeatureTracker.cc:1170-1205):
  per object   obj = (mask == id) dilated by a 1x11 RECT (vertical); its contours are drawn FILLED with the id into one 8-bit
               image ("thicc_boarder") - with all contours of the tree drawn together the even-odd fill reproduces the
               dilated component itself, so the image is the grey-scale dilation of the label image by the 1x11 element
               (objects in ascending id order: the larger id wins an overlap); bounding box = box of the dilated object
  outer border = dilate(thicc, ELLIPSE(2t+1)) - thicc,   inner border = thicc - erode(thicc, ELLIPSE(21))   (saturating u8)
  boundary_mask = 255 (detection mask) or 0 everywhere, borders set to 0 / 255; labelled_boundary_mask = outer | inner
  inner boxes  = findObjectBoundingBox(eroded, id)
OpenCV morphology [algorithm recalled: structuring elements of getStructuringElement, constant border that never wins].
Integer / byte work: the GPU must match bit for bit."""
from __future__ import annotations

import numpy as np


def ellipse(r: int) -> np.ndarray:
    """cv::getStructuringElement(MORPH_ELLIPSE, Size(2r+1, 2r+1)) as half-widths dx[i] per row (row i covers c-dx .. c+dx)"""
    k = 2 * r + 1
    c = r
    inv_r2 = 1.0 / (r * r) if r else 0.0
    out = np.zeros((k, k), bool)
    for i in range(k):
        dy = i - r
        dx = int(np.rint(c * np.sqrt((r * r - dy * dy) * inv_r2)))
        out[i, max(c - dx, 0):min(c + dx + 1, k)] = True
    return out


def _morph(img: np.ndarray, se: np.ndarray, op: str) -> np.ndarray:
    h, w = img.shape
    kh, kw = se.shape
    ay, ax = kh // 2, kw // 2
    pad_val = 0 if op == "dilate" else 255
    P = np.full((h + kh - 1, w + kw - 1), pad_val, np.uint8)
    P[ay:ay + h, ax:ax + w] = img
    out = np.full((h, w), pad_val, np.uint8)
    f = np.maximum if op == "dilate" else np.minimum
    for i in range(kh):
        for j in range(kw):
            if se[i, j]:
                out = f(out, P[i:i + h, j:j + w])
    return out


def _boxes(lab: np.ndarray, ids):
    """bounding boxes (x, y, w, h) of (lab == id) dilated by the 1x11 element; (0,0,0,0) if absent"""
    out = []
    se = np.ones((11, 1), bool)
    for j in ids:
        d = _morph((lab == j).astype(np.uint8), se, "dilate")
        ys, xs = np.nonzero(d)
        out.append((int(xs.min()), int(ys.min()), int(xs.max() - xs.min() + 1), int(ys.max() - ys.min() + 1)) if len(xs) else (0, 0, 0, 0))
    return out


def boundary_mask(mask: np.ndarray, thickness: int, use_as_feature_detection_mask: bool):
    mask = np.asarray(mask)
    ids = sorted(int(v) for v in np.unique(mask) if v != 0)
    assert all(0 < j <= 255 for j in ids)
    lab = np.where((mask > 0) & (mask <= 255), mask, 0).astype(np.uint8)
    thicc = _morph(lab, np.ones((11, 1), bool), "dilate")
    dil = _morph(thicc, ellipse(thickness), "dilate")
    ero = _morph(thicc, ellipse(10), "erode")
    outer = np.where(dil > thicc, dil - thicc, 0).astype(np.uint8)
    inner = np.where(thicc > ero, thicc - ero, 0).astype(np.uint8)
    base, fill = (255, 0) if use_as_feature_detection_mask else (0, 255)
    bm = np.full(mask.shape, base, np.uint8)
    bm[(outer != 0) | (inner != 0)] = fill
    return dict(boundary_mask=bm, labelled=(outer | inner).astype(np.uint8), objects=ids, boxes=_boxes(lab, ids), inner_boxes=_boxes(ero, ids))
