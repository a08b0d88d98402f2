"""CPU restatement (numpy) of the dense-flow / dynamic-feature propagation path of dynoflow.hip.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.

 * `track_dynamic` restates the per-feature logic of FeatureTracker::trackDynamic
   (dynosam/src/frontend/vision/FeatureTracker.cc:380-470) — integer/byte work, compared bit-exactly.
   Pinned pieces: Camera::isKeypointContained (dynosam_cv/src/Camera.cc:71-74),
   FeatureTrackerBase::isWithinShrunkenImage (FeatureTrackerBase.cc:313-326),
   functional_keypoint::u/v = static_cast<int> (dynosam_cv/include/dynosam_cv/Feature.hpp:46-56).
   The disc cv::circle(..., FILLED) blanks is restated from OpenCV's rasteriser as recalled (OpenCV is
   not installed here): row dy has half-width floor(sqrt(r^2 + r - dy^2)).
 * `dense_flow` restates THIS repository's flow producer (the reference consumes an off-line RAFT
   image, README.md:204, and holds no arithmetic for it): PARITY UNPINNED against the reference;
   it pins the HIP kernels against an independent numpy statement of the same algorithm.
"""
from __future__ import annotations

import numpy as np

DC = 64
F32 = np.float32


def fma32(a, b, c):
    """fmaf(a, b, c) for float32 arrays (product exact in float64)."""
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(F32)


def gray(rgb: np.ndarray) -> np.ndarray:
    r, g, b = (rgb[..., k].astype(F32) for k in range(3))
    return fma32(F32(0.114) * np.ones_like(b), b, fma32(F32(0.587) * np.ones_like(g), g, F32(0.299) * r))


def down(img: np.ndarray) -> np.ndarray:
    a, b, c, d = img[0::2, 0::2], img[0::2, 1::2], img[1::2, 0::2], img[1::2, 1::2]
    return (F32(0.25) * ((a + b) + (c + d))).astype(F32)


def pyramid(rgb: np.ndarray, levels: int = 4):
    out = [gray(rgb)]
    for _ in range(1, levels):
        out.append(down(out[-1]))
    return out


def to_bf16_bits(x: np.ndarray) -> np.ndarray:
    u = x.astype(F32).view(np.uint32).astype(np.uint64)
    u = u + 0x7FFF + ((u >> 16) & 1)
    return ((u >> 16) & 0xFFFF).astype(np.uint16)


def bf16_to_f32(b: np.ndarray) -> np.ndarray:
    return (b.astype(np.uint32) << 16).view(F32)


def descriptors(img: np.ndarray) -> np.ndarray:
    """[h*w, 64] bf16 bit patterns: 8x8 patch (offsets -4..3, edge clamp), zero mean, unit norm."""
    h, w = img.shape
    ys, xs = np.mgrid[0:h, 0:w]
    v = np.zeros((h, w, DC), F32)
    s = np.zeros((h, w), F32)
    for dy in range(8):
        for dx in range(8):
            t = img[np.clip(ys + dy - 4, 0, h - 1), np.clip(xs + dx - 4, 0, w - 1)]
            v[..., dy * 8 + dx] = t
            s = (s + t).astype(F32)
    mean = (s * F32(1.0 / DC)).astype(F32)
    v = (v - mean[..., None]).astype(F32)
    q = np.zeros((h, w), F32)
    for k in range(DC):
        q = fma32(v[..., k], v[..., k], q)
    nrm = np.sqrt(q).astype(F32)
    with np.errstate(divide="ignore"):
        inv = np.where(nrm > F32(1e-3), (F32(1.0) / nrm).astype(F32), F32(0.0)).astype(F32)
    d = (v * inv[..., None]).astype(F32)
    return to_bf16_bits(d).reshape(h * w, DC)


def corr_argmax(da_bits: np.ndarray, db_bits: np.ndarray, w: int, h: int, R: int):
    """best match q (index in frame k+1) for every pixel p of frame k, |dx|,|dy| <= R; ties -> lowest q;
    no positive correlation -> q = p."""
    A, B = bf16_to_f32(da_bits), bf16_to_f32(db_bits)
    n = w * h
    C = A @ B.T                      # float32 accumulate (products of bf16 are exact in float32)
    px, py = np.arange(n) % w, np.arange(n) // w
    ok = (np.abs(px[None, :] - px[:, None]) <= R) & (np.abs(py[None, :] - py[:, None]) <= R)
    C = np.where(ok, C, -np.inf)
    best = C.argmax(1)               # first maximum = lowest index
    bv = C[np.arange(n), best]
    best = np.where(bv > 0, best, np.arange(n))
    return best.astype(np.int32), C


def _cost(pa, B, xs, ys, fx, fy, dx, dy):
    h, w = B.shape
    c = np.zeros(xs.shape, F32)
    k = 0
    for v in range(5):
        for u in range(5):
            b = B[np.clip(ys + v - 2 + fy + dy, 0, h - 1), np.clip(xs + u - 2 + fx + dx, 0, w - 1)]
            d = (pa[k] - b).astype(F32)
            c = fma32(d, d, c)
            k += 1
    return c


def refine(A, B, fin, r, final=False):
    """one coarse-to-fine step: fin = integer flow [h/2, w/2, 2] of the coarser level."""
    h, w = A.shape
    ys, xs = np.mgrid[0:h, 0:w]
    fp = fin[ys >> 1, xs >> 1]
    fx, fy = 2 * fp[..., 0], 2 * fp[..., 1]
    pa = [A[np.clip(ys + v - 2, 0, h - 1), np.clip(xs + u - 2, 0, w - 1)] for v in range(5) for u in range(5)]
    bc = np.full((h, w), np.inf, F32)
    bx, by = np.zeros((h, w), np.int64), np.zeros((h, w), np.int64)
    for dy in range(-r, r + 1):
        for dx in range(-r, r + 1):
            c = _cost(pa, B, xs, ys, fx, fy, dx, dy)
            upd = c < bc
            bc = np.where(upd, c, bc)
            bx = np.where(upd, dx, bx)
            by = np.where(upd, dy, by)
    if not final:
        return np.stack([fx + bx, fy + by], -1).astype(np.int32)
    cxm, cxp = _cost(pa, B, xs, ys, fx, fy, bx - 1, by), _cost(pa, B, xs, ys, fx, fy, bx + 1, by)
    cym, cyp = _cost(pa, B, xs, ys, fx, fy, bx, by - 1), _cost(pa, B, xs, ys, fx, fy, bx, by + 1)
    dxx = ((cxm - F32(2.0) * bc).astype(F32) + cxp).astype(F32)
    dyy = ((cym - F32(2.0) * bc).astype(F32) + cyp).astype(F32)
    with np.errstate(divide="ignore", invalid="ignore"):
        ox = np.where(dxx > 0, (F32(0.5) * (cxm - cxp).astype(F32)).astype(F32) / dxx, F32(0.0)).astype(F32)
        oy = np.where(dyy > 0, (F32(0.5) * (cym - cyp).astype(F32)).astype(F32) / dyy, F32(0.0)).astype(F32)
    ox, oy = np.clip(ox, F32(-0.5), F32(0.5)), np.clip(oy, F32(-0.5), F32(0.5))
    return np.stack([(fx + bx).astype(F32) + ox, (fy + by).astype(F32) + oy], -1).astype(F32)


def dense_flow(rgb0: np.ndarray, rgb1: np.ndarray, R: int = 6):
    """returns (flow [H, W, 2] float32, coarse match [h3*w3] int32)."""
    p0, p1 = pyramid(rgb0), pyramid(rgb1)
    h3, w3 = p0[3].shape
    d0, d1 = descriptors(p0[3]), descriptors(p1[3])
    match, _ = corr_argmax(d0, d1, w3, h3, R)
    n = w3 * h3
    cflow = np.stack([match % w3 - np.arange(n) % w3, match // w3 - np.arange(n) // w3], -1).reshape(h3, w3, 2)
    f2 = refine(p0[2], p1[2], cflow, 2)
    f1 = refine(p0[1], p1[1], f2, 1)
    flow = refine(p0[0], p1[0], f1, 1, final=True)
    return flow, match


KEPT, MASKED_OUT, NOT_CONTAINED, BACKGROUND, LABEL_CHANGED, OUTSIDE_SHRUNKEN, ZERO_FLOW = range(7)


def track_dynamic(kp, prev_label, age, tracklet_id, flow, motion_mask, detection_mask=None, shrink_row=0, shrink_col=0,
                  max_age=25, min_distance=2, next_tracklet_id=0):
    """FeatureTracker.cc:380-470 for a list of previous dynamic features, in order.  flow: [H, W, 2] float32."""
    H, W = motion_mask.shape
    det = np.full((H, W), 255, np.uint8) if detection_mask is None else detection_mask.copy()
    n = len(kp)
    code = np.zeros(n, np.int32); label = np.zeros(n, np.int32); new_age = np.array(age, np.int32).copy()
    new_tid = np.array(tracklet_id, np.int64).copy(); fl = np.zeros((n, 2)); pk = np.zeros((n, 2))
    for i in range(n):
        kx, ky = float(kp[i][0]), float(kp[i][1])
        x, y = int(kx), int(ky)                      # static_cast<int>: truncation toward zero
        inb = 0 <= x < W and 0 <= y < H
        contained = (kx >= 0.0) and (kx < W) and (ky >= 0.0) and (ky < H)
        lab = int(motion_mask[y, x]) if inb else 0
        f = flow[y, x] if inb else np.zeros(2, np.float32)
        label[i] = lab
        fl[i] = (float(f[0]), float(f[1]))
        pk[i] = (kx + float(f[0]), ky + float(f[1]))
        if inb and det[y, x] == 0:
            code[i] = MASKED_OUT; continue
        if not contained or not inb:
            code[i] = NOT_CONTAINED; continue
        if lab == 0:
            code[i] = BACKGROUND; continue
        if lab != int(prev_label[i]):
            code[i] = LABEL_CHANGED; continue
        pc, pr = int(pk[i][0]), int(pk[i][1])
        if not (pr > shrink_row and pr < (H - shrink_row) and pc > shrink_col and pc < (W - shrink_col)):
            code[i] = OUTSIDE_SHRUNKEN; continue
        if f[0] == 0 or f[1] == 0:
            code[i] = ZERO_FLOW; continue
        na, tid = int(age[i]) + 1, int(tracklet_id[i])
        if na > max_age:
            tid = next_tracklet_id; next_tracklet_id += 1; na = 0
        new_age[i], new_tid[i] = na, tid
        code[i] = KEPT
        r = min_distance
        for dy in range(-r, r + 1):
            yy, v = y + dy, r * r + r - dy * dy
            if yy < 0 or yy >= H or v < 0:
                continue
            hw = int(np.floor(np.sqrt(v)))
            det[yy, max(0, x - hw):min(W - 1, x + hw) + 1] = 0
    return dict(code=code, label=label, new_age=new_age, new_tracklet_id=new_tid, flow=fl, predicted_kp=pk,
                next_tracklet_id=next_tracklet_id, detection_mask=det)
