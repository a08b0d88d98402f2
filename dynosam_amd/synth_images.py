"""Synthetic RGB + motion-mask frame pairs with exactly known optical flow (BASELINE config 4):
a textured static scene translated by a uniform flow plus rigid objects moving by their own affine maps.
Textures are sums of sinusoids, so frame k+1 is evaluated analytically at the warped coordinates (no
interpolation) and the ground-truth flow is exact.  Host-side input plumbing for tests and bench.py."""
from __future__ import annotations

import numpy as np


def _texture(rng, n_waves=28):
    wl = np.exp(rng.uniform(np.log(5.0), np.log(90.0), n_waves))
    ang = rng.uniform(0, 2 * np.pi, n_waves)
    k = np.stack([np.cos(ang), np.sin(ang)], -1) * (2 * np.pi / wl)[:, None]
    amp = rng.uniform(0.3, 1.0, n_waves) * (wl / 90.0) ** 0.35
    ph = rng.uniform(0, 2 * np.pi, (3, n_waves))

    def f(x, y):
        arg = x[..., None] * k[:, 0] + y[..., None] * k[:, 1]
        sn, cs = np.sin(arg), np.cos(arg)   # sin(arg + ph) = sin(arg) cos(ph) + cos(arg) sin(ph)
        out = sn @ (amp * np.cos(ph)).T + cs @ (amp * np.sin(ph)).T
        return out / (np.sqrt((amp ** 2).sum() / 2) * 2.2)
    return f


def make_pair(width=640, height=480, objects=3, seed=4, max_flow=8.0):
    """returns dict(rgb0, rgb1 [H,W,3] u8, mask0, mask1 [H,W] i32, flow_gt [H,W,2] f32, valid [H,W] bool)."""
    rng = np.random.default_rng(seed)
    ys, xs = np.mgrid[0:height, 0:width].astype(np.float64)
    bg = _texture(rng)
    u_bg = np.round(rng.uniform(-max_flow, max_flow, 2))          # uniform camera-induced flow (integer pixels)
    objs = []
    for j in range(objects):
        c = np.array([rng.uniform(0.2, 0.8) * width, rng.uniform(0.25, 0.75) * height])
        half = np.array([rng.uniform(40, 90), rng.uniform(30, 70)])
        th, sc = rng.uniform(-0.03, 0.03), 1.0 + rng.uniform(-0.03, 0.03)
        M = sc * np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
        t = rng.uniform(-max_flow, max_flow, 2)
        objs.append(dict(c=c, half=half, M=M, t=t, tex=_texture(rng)))

    def render(frame):
        img = bg(xs - frame * u_bg[0], ys - frame * u_bg[1])
        mask = np.zeros((height, width), np.int32)
        for j, o in enumerate(objs):
            if frame == 0:
                ox, oy = xs, ys
            else:  # object coordinates of the pixel in frame 1: x = M^-1 (x' - c - t) + c
                Mi = np.linalg.inv(o["M"])
                dx, dy = xs - o["c"][0] - o["t"][0], ys - o["c"][1] - o["t"][1]
                ox, oy = Mi[0, 0] * dx + Mi[0, 1] * dy + o["c"][0], Mi[1, 0] * dx + Mi[1, 1] * dy + o["c"][1]
            inside = (np.abs(ox - o["c"][0]) <= o["half"][0]) & (np.abs(oy - o["c"][1]) <= o["half"][1])
            img = np.where(inside[..., None], o["tex"](ox, oy), img)
            mask = np.where(inside, j + 1, mask)
        return np.clip(np.round(127.5 + 105.0 * img), 0, 255).astype(np.uint8), mask

    rgb0, mask0 = render(0)
    rgb1, mask1 = render(1)
    flow = np.zeros((height, width, 2))
    flow[..., 0], flow[..., 1] = u_bg[0], u_bg[1]
    for j, o in enumerate(objs):
        sel = mask0 == j + 1
        dx, dy = xs - o["c"][0], ys - o["c"][1]
        fx = (o["M"][0, 0] - 1) * dx + o["M"][0, 1] * dy + o["t"][0]
        fy = o["M"][1, 0] * dx + (o["M"][1, 1] - 1) * dy + o["t"][1]
        flow[..., 0] = np.where(sel, fx, flow[..., 0])
        flow[..., 1] = np.where(sel, fy, flow[..., 1])
    # valid = the pixel's target is inside frame 1 and still shows the same surface (not occluded), away from the border
    tx, ty = xs + flow[..., 0], ys + flow[..., 1]
    inb = (tx >= 8) & (tx < width - 8) & (ty >= 8) & (ty < height - 8) & (xs >= 8) & (xs < width - 8) & (ys >= 8) & (ys < height - 8)
    txi, tyi = np.clip(np.round(tx).astype(int), 0, width - 1), np.clip(np.round(ty).astype(int), 0, height - 1)
    valid = inb & (mask1[tyi, txi] == mask0)
    return dict(rgb0=rgb0, rgb1=rgb1, mask0=mask0, mask1=mask1, flow_gt=flow.astype(np.float32), valid=valid, u_bg=u_bg)


def make_sequence(width=640, height=480, objects=3, frames=10, seed=4, max_flow=6.0, return_flow=False):
    """a short stream for the composed tracker: the static scene slides by an integer flow per frame, every object moves rigidly
    (constant translation + small rotation / scale per frame about its own moving centre).  returns (rgb [F,H,W,3] u8, mask [F,H,W] i32);
    with return_flow also the exact flow images [F,H,W,2] f32 (frame f -> f+1: what ImageContainer::opticalFlow() of frame f holds)."""
    rng = np.random.default_rng(seed)
    ys, xs = np.mgrid[0:height, 0:width].astype(np.float64)
    bg = _texture(rng)
    u_bg = np.round(rng.uniform(-max_flow, max_flow, 2))
    objs = []
    for j in range(objects):
        c = np.array([rng.uniform(0.25, 0.75) * width, rng.uniform(0.3, 0.7) * height])
        half = np.array([rng.uniform(35, 70), rng.uniform(30, 55)])
        objs.append(dict(c=c, half=half, th=rng.uniform(-0.01, 0.01), sc=1.0 + rng.uniform(-0.004, 0.004), t=rng.uniform(-max_flow, max_flow, 2), tex=_texture(rng)))
    rgbs, masks = [], []
    for f in range(frames):
        img = bg(xs - f * u_bg[0], ys - f * u_bg[1])
        mask = np.zeros((height, width), np.int32)
        for j, o in enumerate(objs):
            th, sc = f * o["th"], o["sc"] ** f
            Mi = np.linalg.inv(sc * np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]]))
            cf = o["c"] + f * o["t"]
            dx, dy = xs - cf[0], ys - cf[1]
            ox, oy = Mi[0, 0] * dx + Mi[0, 1] * dy + o["c"][0], Mi[1, 0] * dx + Mi[1, 1] * dy + o["c"][1]
            inside = (np.abs(ox - o["c"][0]) <= o["half"][0]) & (np.abs(oy - o["c"][1]) <= o["half"][1])
            img = np.where(inside[..., None], o["tex"](ox, oy), img)
            mask = np.where(inside, j + 1, mask)
        rgbs.append(np.clip(np.round(127.5 + 105.0 * img), 0, 255).astype(np.uint8)); masks.append(mask)
    if not return_flow:
        return np.stack(rgbs), np.stack(masks)
    flows = []
    for f in range(frames):
        fl = np.zeros((height, width, 2))
        fl[..., 0], fl[..., 1] = u_bg[0], u_bg[1]
        for j, o in enumerate(objs):
            th, sc = f * o["th"], o["sc"] ** f
            Mi = np.linalg.inv(sc * np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]]))
            th1, sc1 = (f + 1) * o["th"], o["sc"] ** (f + 1)
            M1 = sc1 * np.array([[np.cos(th1), -np.sin(th1)], [np.sin(th1), np.cos(th1)]])
            A = M1 @ Mi                                              # x' = c_{f+1} + A (x - c_f)
            cf, cf1 = o["c"] + f * o["t"], o["c"] + (f + 1) * o["t"]
            dx, dy = xs - cf[0], ys - cf[1]
            sel = masks[f] == j + 1
            fl[..., 0] = np.where(sel, cf1[0] + A[0, 0] * dx + A[0, 1] * dy - xs, fl[..., 0])
            fl[..., 1] = np.where(sel, cf1[1] + A[1, 0] * dx + A[1, 1] * dy - ys, fl[..., 1])
        flows.append(fl.astype(np.float32))
    return np.stack(rgbs), np.stack(masks), np.stack(flows)
