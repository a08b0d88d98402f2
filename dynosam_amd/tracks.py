"""Frontend tracks -> flat HYBRID-formulation factor graph (first step).

Input is the array form of what the frontend hands the backend per frame (VisionImuPacket / the reference's
`small_frontend.bson` records: camera pose estimate T_world_camera, per-observation {frame, tracklet, object, point in the
camera frame}, per-object world motions H_W_{k-1,k}, propagated object poses L_W_k) — see tests/golden/make_small_frontend.py.
The builder restates, on flat arrays and WITHOUT the reference's Map / accessor machinery, the graph shape of
  * Formulation<MAP>::updateStaticObservations / StaticFormulationUpdater (Formulation-impl.hpp:145-235): one
    PoseToPointFactor per observation of a static tracklet seen >= min_static_observations times, landmark initialised by
    back-projecting its first observation through the camera pose estimate (:218-229)
  * HybridFormulation (HybridEstimator.cc:573-811): per object a keyframe e = first frame it is seen, L_e its pose there,
    variables eH_k (prior Identity at e, :744-746), one HybridMotionFactor per observation of a tracklet seen >=
    min_dynamic_observations times, point initialised by projectToObject3 (:647-657), HybridSmoothingFactor over
    consecutive motions
  * VisionImuBackendModule (VisionImuBackendModule.hpp:88-243): prior on the first camera pose, BetweenFactor odometry
  * noise: BackendParams.cc:33-80 defaults through NoiseModels::fromBackendParams (BackendDefinitions.cc:124-194)
Slots are assigned in insertion order, frame by frame (Formulation-impl.hpp:625).  It is NOT yet the full port of the
reference's update functions (no late insertion of back-tracked observations, one keyframe per object): §8f row 1 proper.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from . import symbols as S
from .graph import (F_BETWEEN_POSE3, F_HYBRID_MOTION, F_HYBRID_SMOOTHING, F_POSE_TO_POINT, F_PRIOR_POSE3, VAR_POINT3, VAR_POSE3,
                    FactorBlock, FlatGraph)
from .synth import act, compose, from12, inverse, to12


@dataclass
class BackendParams:
    """dynosam/src/backend/BackendParams.cc:33-80 (code defaults)"""
    static_point_noise_sigma: float = 0.2
    dynamic_point_noise_sigma: float = 0.2
    odometry_rotation_sigma: float = 0.02
    odometry_translation_sigma: float = 0.01
    constant_object_motion_rotation_sigma: float = 0.01
    constant_object_motion_translation_sigma: float = 0.1
    k_huber_3d_points: float = 1e-4
    use_robust_kernels: bool = True
    min_static_observations: int = 2
    min_dynamic_observations: int = 3
    prior_sigma: float = 1e-6


def build_hybrid_graph(frames, X_world, observations, motions, object_poses, params: BackendParams = BackendParams()) -> FlatGraph:
    frames = np.asarray(frames, dtype=np.int64)
    fidx = {int(f): i for i, f in enumerate(frames)}
    obs = np.asarray(observations, dtype=np.float64)
    o_frame, o_track, o_obj = obs[:, 0].astype(np.int64), obs[:, 1].astype(np.int64), obs[:, 2].astype(np.int64)
    o_pt = obs[:, 3:6]
    X = from12(np.asarray(X_world, dtype=np.float64))
    K = len(frames)
    # ---- observation gates: a tracklet enters the graph once it has been seen often enough ----
    def gate(sel, nmin):
        tr, cnt = np.unique(o_track[sel], return_counts=True)
        return set(int(t) for t, c in zip(tr, cnt) if c >= nmin)
    is_dyn = o_obj > 0
    s_ok, d_ok = gate(~is_dyn, params.min_static_observations), gate(is_dyn, params.min_dynamic_observations)
    use = np.array([(int(t) in (d_ok if dy else s_ok)) for t, dy in zip(o_track, is_dyn)])
    # ---- objects: keyframe, L_e, initial motions eH_k = H_{k-1,k} ... H_{e,e+1} ----
    objs = sorted(set(int(o) for o in o_obj[is_dyn & use]))
    mot = {(int(m[0]), int(m[1])): from12(m[2:]) for m in np.asarray(motions)}
    last = np.asarray(object_poses)
    H_keys, H_state, H_of, L_e = [], [], {}, {}
    for j in objs:
        seen = sorted(set(int(f) for f in o_frame[(o_obj == j) & use]))
        e = seen[0]
        cand = last[(last[:, 1] == j) & (last[:, 2] == e)]
        if not len(cand):
            raise ValueError(f"no propagated pose for object {j} at its keyframe {e}")
        L_e[j] = from12(cand[-1, 3:])
        Hk = (np.eye(3), np.zeros(3))
        for k in range(e, int(frames[-1]) + 1):
            if k > e:
                if (k, j) not in mot:
                    break
                Hk = compose(mot[(k, j)], Hk)
            H_of[(j, k)] = len(H_keys)
            H_keys.append(S.ObjectMotionSymbol(j, k))
            H_state.append(to12(Hk))
    # ---- variables ----
    X_keys = [S.CameraPoseSymbol(int(f)) for f in frames]
    st_tracks = sorted(t for t in s_ok)
    dy_tracks = sorted(t for t in d_ok)
    l_of, m_of = {t: i for i, t in enumerate(st_tracks)}, {t: i for i, t in enumerate(dy_tracks)}
    l_init, m_init = np.zeros((len(st_tracks), 3)), np.zeros((len(dy_tracks), 3))
    l_set, m_set = np.zeros(len(st_tracks), bool), np.zeros(len(dy_tracks), bool)
    # ---- factors, frame by frame (insertion order = slot) ----
    rows = {t: [] for t in (F_PRIOR_POSE3, F_BETWEEN_POSE3, F_POSE_TO_POINT, F_HYBRID_MOTION, F_HYBRID_SMOOTHING)}
    slot = 0

    def add(ftype, var, meas=(), noise=(), hk=0.0, consts=()):
        nonlocal slot
        rows[ftype].append((slot, var, np.asarray(meas, float), np.asarray(noise, float), hk, np.asarray(consts, float)))
        slot += 1

    iso6 = lambda sr, st: [sr] * 3 + [st] * 3
    Rs = np.eye(3).reshape(-1) / params.static_point_noise_sigma
    Rd = np.eye(3).reshape(-1) / params.dynamic_point_noise_sigma
    hub = params.k_huber_3d_points if params.use_robust_kernels else 0.0
    n_H = len(H_keys)
    for i, f in enumerate(frames):
        f = int(f)
        if i == 0:
            add(F_PRIOR_POSE3, [("X", i)], to12((X[0][i], X[1][i])), iso6(params.prior_sigma, params.prior_sigma))
        else:
            rel = compose(inverse((X[0][i - 1], X[1][i - 1])), (X[0][i], X[1][i]))
            add(F_BETWEEN_POSE3, [("X", i - 1), ("X", i)], to12(rel), iso6(params.odometry_rotation_sigma, params.odometry_translation_sigma))
        for k in np.nonzero((o_frame == f) & use)[0]:
            t = int(o_track[k])
            if not is_dyn[k]:
                if not l_set[l_of[t]]:
                    l_init[l_of[t]] = act((X[0][i], X[1][i]), o_pt[k]); l_set[l_of[t]] = True
                add(F_POSE_TO_POINT, [("X", i), ("l", l_of[t])], o_pt[k], Rs, hub)
            else:
                j = int(o_obj[k])
                if (j, f) not in H_of:
                    continue
                h = H_of[(j, f)]
                if not m_set[m_of[t]]:   # projectToObject3: m = L_e^-1 H^-1 X z
                    pw = act((X[0][i], X[1][i]), o_pt[k])
                    m_init[m_of[t]] = act(inverse(L_e[j]), act(inverse(from12(H_state[h])), pw)); m_set[m_of[t]] = True
                add(F_HYBRID_MOTION, [("X", i), ("H", h), ("m", m_of[t])], o_pt[k], Rd, hub, to12(L_e[j]))
        for j in objs:
            if all((j, f - d) in H_of for d in (0, 1, 2)):
                add(F_HYBRID_SMOOTHING, [("H", H_of[(j, f - 2)]), ("H", H_of[(j, f - 1)]), ("H", H_of[(j, f)])], (),
                    iso6(params.constant_object_motion_rotation_sigma, params.constant_object_motion_translation_sigma), 0.0, to12(L_e[j]))
            if (j, f) in H_of and (j, f - 1) not in H_of:      # the object's keyframe: prior Identity on eH_e
                add(F_PRIOR_POSE3, [("H", H_of[(j, f)])], to12((np.eye(3), np.zeros(3))), iso6(params.prior_sigma, params.prior_sigma))
    keep_l, keep_m = np.nonzero(l_set)[0], np.nonzero(m_set)[0]
    l_keys = [S.StaticLandmarkSymbol(st_tracks[i]) for i in keep_l]
    m_keys = [S.HybridDynamicKey(dy_tracks[i]) for i in keep_m]
    l_new, m_new = {int(o): n for n, o in enumerate(keep_l)}, {int(o): n for n, o in enumerate(keep_m)}
    keys = np.array(H_keys + X_keys + l_keys + m_keys, dtype=np.uint64)
    vtype = np.array([VAR_POSE3] * (n_H + K) + [VAR_POINT3] * (len(l_keys) + len(m_keys)), dtype=np.uint8)
    pad = lambda p: np.concatenate([p, np.zeros((len(p), 9))], -1) if len(p) else np.zeros((0, 12))
    state = np.concatenate([np.array(H_state).reshape(n_H, 12), to12(X), pad(l_init[keep_l]), pad(m_init[keep_m])], 0)
    order = np.argsort(keys, kind="stable")
    inv = np.empty_like(order); inv[order] = np.arange(len(order))
    base = {"H": 0, "X": n_H, "l": n_H + K, "m": n_H + K + len(l_keys)}

    def vid(kind, idx):
        if kind == "l": idx = l_new[idx]
        if kind == "m": idx = m_new[idx]
        return inv[base[kind] + idx]

    blocks = []
    for ftype, rws in rows.items():
        if not rws:
            continue
        blocks.append(FactorBlock(ftype, [r[0] for r in rws], np.array([[vid(*v) for v in r[1]] for r in rws]), np.array([r[2] for r in rws]),
                                  np.array([r[3] for r in rws]), np.array([r[4] for r in rws]) if any(r[4] > 0 for r in rws) else None,
                                  np.array([r[5] for r in rws]) if rws[0][5].size else None))
    return FlatGraph(keys[order], vtype[order], state[order], blocks, dict(frames=K, objects=len(objs), n_factors=slot))


def load_fixture(path):
    z = np.load(path)
    return z["frames"], z["X_world"], z["observations"], z["motions"], z["object_poses"]
