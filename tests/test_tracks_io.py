"""Tracks wire format (dynosam_amd/tracks_io.py): write -> read round trip is exact (integers and IEEE doubles), a stream
with unknown length reads until EOF, and the reference's real 9-frame fixture written in this format drives the incremental
HYBRID formulation to the same graph as the array fixture does."""
import io
import os

import numpy as np

from dynosam_amd import formulation as FM
from dynosam_amd import tracks_io as TIO

HERE = os.path.dirname(os.path.abspath(__file__))


def fixture_packets():
    z = np.load(os.path.join(HERE, "golden", "small_frontend_tracks.npz"))
    pk = FM.packets_from_arrays(z["frames"], z["X_world"], z["observations"], z["motions"])
    poses = z["object_poses"]
    out = []
    for i, p in enumerate(pk):
        st = np.concatenate([p.static[:, :1], np.zeros((len(p.static), 2)), p.static[:, 1:]], 1) if len(p.static) else np.zeros((0, 6))
        dy = np.concatenate([p.dynamic[:, :2], np.zeros((len(p.dynamic), 2)), p.dynamic[:, 2:]], 1) if len(p.dynamic) else np.zeros((0, 7))
        op = {int(r[1]): r[3:] for r in poses if int(r[0]) == p.frame_id and int(r[2]) == p.frame_id}
        out.append(TIO.TrackPacket(p.frame_id, 0.1 * i, p.X_world, p.T_k_1_k, dict(p.motions), op, st, dy))
    return pk, out


def test_round_trip_is_exact(tmp_path):
    rng = np.random.default_rng(0)
    pose = lambda: np.concatenate([np.linalg.qr(rng.normal(size=(3, 3)))[0].reshape(-1), rng.normal(size=3)])
    pk = []
    for k in range(4):
        ns, nd = int(rng.integers(0, 6)), int(rng.integers(0, 5))
        st = np.concatenate([rng.integers(0, 1 << 40, (ns, 1)).astype(float), rng.normal(size=(ns, 5))], 1)
        dy = np.concatenate([rng.integers(0, 1 << 40, (nd, 1)).astype(float), rng.integers(1, 9, (nd, 1)).astype(float), rng.normal(size=(nd, 5))], 1)
        pk.append(TIO.TrackPacket(10 + k, 0.033 * k, pose(), None if k == 0 else pose(), {3: pose(), 7: pose()}, {7: pose()}, st, dy,
                                  rng.normal(size=(ns, 9)) if k % 2 else None, rng.normal(size=(nd, 9)) if k % 2 else None))
    path = str(tmp_path / "t.dytr")
    TIO.write_tracks(path, pk)
    back = list(TIO.read_tracks(path))
    assert len(back) == 4
    for a, b in zip(pk, back):
        assert (a.frame_id, a.timestamp) == (b.frame_id, b.timestamp) and np.array_equal(a.X_world, b.X_world)
        assert (a.T_k_1_k is None) == (b.T_k_1_k is None) and (a.T_k_1_k is None or np.array_equal(a.T_k_1_k, b.T_k_1_k))
        assert sorted(a.motions) == sorted(b.motions) and all(np.array_equal(a.motions[o], b.motions[o]) for o in a.motions)
        assert sorted(a.object_poses) == sorted(b.object_poses) and np.array_equal(a.object_poses[7], b.object_poses[7])
        assert np.array_equal(a.static, b.static) and np.array_equal(a.dynamic, b.dynamic)
        if a.static_cov is not None and len(a.static):
            assert np.array_equal(a.static_cov, b.static_cov)
        if a.dynamic_cov is not None and len(a.dynamic):
            assert np.array_equal(a.dynamic_cov, b.dynamic_cov)


def test_stream_of_unknown_length_reads_to_eof(tmp_path):
    _, out = fixture_packets()
    path = str(tmp_path / "s.dytr")
    with open(path, "wb") as f:
        TIO.write_header(f)                   # n_frames unknown
        for p in out[:5]:
            TIO.write_packet(f, p)
    assert [p.frame_id for p in TIO.read_tracks(path)] == [p.frame_id for p in out[:5]]


def test_real_fixture_through_the_wire_format_builds_the_same_graph(tmp_path):
    pk, out = fixture_packets()
    path = str(tmp_path / "small_frontend.dytr")
    TIO.write_tracks(path, out)
    f1, f2 = FM.HybridFormulation(), FM.HybridFormulation()
    for p in pk:
        f1.update(p)
    for p in TIO.read_tracks(path):
        f2.update(TIO.to_frame_packet(p))
    g1, g2 = f1.graph(), f2.graph()
    assert np.array_equal(g1.var_keys, g2.var_keys) and np.array_equal(g1.var_state, g2.var_state) and g1.n_factors == g2.n_factors > 0
    for a, b in zip(g1.blocks, g2.blocks):
        assert a.type == b.type and np.array_equal(a.slot, b.slot) and np.array_equal(a.var_idx, b.var_idx) and np.array_equal(a.meas, b.meas)


def test_pose_only_objects_carry_no_motion_and_version_1_files_still_read(tmp_path):
    """format version 2: an object record says whether it holds a motion (version 1 wrote an identity H for a pose-only
    object, which a reader could not tell from a real frontend motion); a version-1 file is still read"""
    import struct
    rng = np.random.default_rng(1)
    pose = lambda: np.concatenate([np.linalg.qr(rng.normal(size=(3, 3)))[0].reshape(-1), rng.normal(size=3)])
    p = TIO.TrackPacket(3, 0.5, pose(), pose(), {2: pose()}, {2: pose(), 5: pose()}, np.zeros((0, 6)), np.zeros((0, 7)))
    path = str(tmp_path / "v2.dytr")
    TIO.write_tracks(path, [p])
    (b,) = list(TIO.read_tracks(path))
    assert sorted(b.motions) == [2] and sorted(b.object_poses) == [2, 5] and np.array_equal(b.object_poses[5], p.object_poses[5])
    assert TIO.to_frame_packet(b).motions.keys() == {2}
    # the same frame written by hand in the version-1 layout (id | H | has_pose | L)
    path1 = str(tmp_path / "v1.dytr")
    with open(path1, "wb") as f:
        f.write(b"DYTR" + struct.pack("<III", 1, 1, 0) + struct.pack("<qd", 3, 0.5) + p.X_world.astype("<f8").tobytes() + b"\x01" + p.T_k_1_k.astype("<f8").tobytes())
        f.write(struct.pack("<I", 1) + struct.pack("<i", 2) + p.motions[2].astype("<f8").tobytes() + b"\x01" + p.object_poses[2].astype("<f8").tobytes())
        f.write(struct.pack("<I", 0) + struct.pack("<I", 0))
    (c,) = list(TIO.read_tracks(path1))
    assert np.array_equal(c.motions[2], p.motions[2]) and np.array_equal(c.object_poses[2], p.object_poses[2])
