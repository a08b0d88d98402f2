"""Convert the reference's own frontend fixture dynosam/test/data/small_frontend.bson (9 real frames of frontend output,
consumed by dynosam/test/test_rgbd_backend.cc:87-140 through FrontendOfflinePipeline) into the array fixture
tests/golden/small_frontend_tracks.npz.  Runs in the build container only (/root/reference is not on the GPU box);
the .npz is committed.  BSON is parsed with the 40-line reader below (no third-party module)."""
import os
import struct
import sys

import numpy as np

SRC = "/root/reference/dynosam/test/data/small_frontend.bson"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "small_frontend_tracks.npz")


def parse_doc(buf, off, as_array=False):
    size = struct.unpack_from("<i", buf, off)[0]
    end = off + size
    off += 4
    out = [] if as_array else {}
    while off < end - 1:
        t = buf[off]; off += 1
        e = buf.index(b"\x00", off); name = buf[off:e].decode(); off = e + 1
        if t == 0x01: v = struct.unpack_from("<d", buf, off)[0]; off += 8
        elif t == 0x02: n = struct.unpack_from("<i", buf, off)[0]; v = buf[off + 4:off + 4 + n - 1].decode(); off += 4 + n
        elif t == 0x03: v, off = parse_doc(buf, off)
        elif t == 0x04: v, off = parse_doc(buf, off, True)
        elif t == 0x08: v = bool(buf[off]); off += 1
        elif t == 0x0A: v = None
        elif t == 0x10: v = struct.unpack_from("<i", buf, off)[0]; off += 4
        elif t == 0x12: v = struct.unpack_from("<q", buf, off)[0]; off += 8
        elif t == 0x11: v = struct.unpack_from("<Q", buf, off)[0]; off += 8
        else: raise ValueError(hex(t))
        if as_array: out.append(v)
        else: out[name] = v
    return out, end


def pose12(q):
    """gtsam::Pose3 from the json form {qw,qx,qy,qz,tx,ty,tz}: row-major R then t (gtsam::Rot3::Quaternion)"""
    w, x, y, z = q["qw"], q["qx"], q["qy"], q["qz"]
    n = np.sqrt(w * w + x * x + y * y + z * z)
    w, x, y, z = w / n, x / n, y / n, z / n
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    return np.concatenate([R.reshape(-1), [q["tx"], q["ty"], q["tz"]]])


def main():
    doc, _ = parse_doc(open(SRC, "rb").read(), 0)
    frames, X, Xgt, obs, motions, obj_poses = [], [], [], [], [], []
    for fid, fr in doc["data"]:
        frames.append(fr["frame_id"])
        X.append(pose12(fr["T_world_camera"]))
        Xgt.append(pose12(fr["ground_truth"]["X_world"]))
        for lm in fr["static_landmarks"] + fr["dynamic_landmarks"]:
            assert lm["reference_frame"] == "local"
            p = np.array(lm["value"]).reshape(-1)
            obs.append([fr["frame_id"], lm["tracklet_id"], lm["object_id"], p[0], p[1], p[2]])
        for oid, m in fr["estimated_motions"]:
            assert m["reference_frame"] == "global"
            motions.append(np.concatenate([[fr["frame_id"], oid], pose12(m["estimate"])]))
        for oid, lst in fr["propogated_object_poses"]:
            for k, q in lst:
                obj_poses.append(np.concatenate([[fr["frame_id"], oid, k], pose12(q)]))
    np.savez_compressed(OUT, frames=np.array(frames), X_world=np.array(X), X_world_gt=np.array(Xgt), observations=np.array(obs),
                        motions=np.array(motions), object_poses=np.array(obj_poses))
    print(OUT, os.path.getsize(OUT), "bytes;", len(obs), "observations over", len(frames), "frames")


if __name__ == "__main__":
    main()
