"""The reference's own frontend fixture (dynosam/test/data/small_frontend.bson, converted by
tests/golden/make_small_frontend.py into tests/golden/small_frontend_tracks.npz): 9 real frames, one moving object.
CPU part: the tracks -> HYBRID graph builder indexes like the reference (ascending keys, slots in insertion order, the
observation gates) and the oracle solves the resulting graph; GPU part: the HIP path follows the oracle on it."""
import os

import numpy as np
import pytest

from dynosam_amd import graph as G
from dynosam_amd import symbols as S
from dynosam_amd import tracks

HERE = os.path.dirname(os.path.abspath(__file__))
FIX = os.path.join(HERE, "golden", "small_frontend_tracks.npz")


@pytest.fixture(scope="module")
def real_graph():
    return tracks.build_hybrid_graph(*tracks.load_fixture(FIX))


def test_builder_indexing(real_graph):
    g = real_graph
    assert np.all(g.var_keys[1:] > g.var_keys[:-1])
    slots = np.sort(np.concatenate([b.slot for b in g.blocks]))
    assert np.array_equal(slots, np.arange(g.n_factors))                    # slot = insertion order, no gaps
    chars = [chr(S.symbol_chr(int(k))) for k in g.var_keys]
    assert chars.count("X") == 9 and chars.count("H") == 9
    z = np.load(FIX)
    obs = z["observations"]
    # gates (BackendParams.cc:75-80): static tracklets need 2 observations, dynamic ones 3
    for is_dyn, nmin, ftype in ((False, 2, G.F_POSE_TO_POINT), (True, 3, G.F_HYBRID_MOTION)):
        sel = (obs[:, 2] > 0) == is_dyn
        tr, cnt = np.unique(obs[sel, 1], return_counts=True)
        expected = int(cnt[cnt >= nmin].sum())
        got = sum(b.count for b in g.blocks if b.type == ftype)
        assert got == expected, (ftype, got, expected)
    # a factor's variables: HybridMotion = (X_k, H_k of the same frame, m of the tracklet)
    hm = [b for b in g.blocks if b.type == G.F_HYBRID_MOTION][0]
    kx, kh = g.var_keys[hm.var_idx[:, 0]], g.var_keys[hm.var_idx[:, 1]]
    assert np.array_equal(kx & np.uint64((1 << 48) - 1), kh & np.uint64((1 << 48) - 1))


def test_oracle_solves_the_real_graph(real_graph, oracle):
    og = oracle.OracleGraph(real_graph)
    e0 = og.error()
    r, _ = og.optimize()
    assert r.error_after < 0.2 * e0 and r.iterations > 3
    og2 = oracle.OracleGraph(real_graph); og2.set_dense(True)
    r2, _ = og2.optimize()
    assert r2.iterations == r.iterations and abs(r2.error_after - r.error_after) <= 1e-9 * r.error_after


@pytest.mark.gpu
def test_gpu_follows_the_oracle_on_real_tracks(real_graph, oracle):
    from dynosam_amd.optimizer import Context
    g = real_graph
    c, og = Context(), oracle.OracleGraph(g)
    c.upload(g)
    assert abs(c.error() - og.error()) <= 1e-12 * og.error()
    J, b, e = c.linearize()
    Jr, br, er = og.linearize()
    assert np.abs(J - Jr).max() <= 1e-9 * np.abs(Jr).max()        # includes the numerically differentiated smoothing factors
    rep = c.optimize()
    rr, _ = og.optimize()
    assert rep.iterations == rr.iterations and rep.inner_iterations == rr.inner_iterations
    assert [rep.trace_accepted[i] for i in range(rep.trace_len)] == [rr.trace_accepted[i] for i in range(rr.trace_len)]
    assert abs(rep.error_after - rr.error_after) <= 1e-6 * rr.error_after
    assert np.abs(c.values() - og.state()).max() <= 1e-5
