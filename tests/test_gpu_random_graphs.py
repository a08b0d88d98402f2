"""Randomised parity sweep: small graphs of every formulation with random sizes, track lengths, object lifetimes, noise and robust
kernels - GPU (through the C-ABI) against the C oracle: graph error, one damped solve, the LM accept / reject trace and final cost,
and (hybrid) a marginalisation with random old keys against the window oracle.  Seeds are fixed: failures reproduce."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from dynosam_amd import synth  # noqa: E402
from oracle import oracle_py as O  # noqa: E402
from oracle import window_oracle as WO  # noqa: E402


def random_config(rng):
    frames = int(rng.integers(4, 15))
    lo_s, lo_d = int(rng.integers(2, 4)), int(rng.integers(2, 4))
    return dict(frames=frames, objects=int(rng.integers(1, 4)), static_points=int(rng.integers(8, 70)),
                dynamic_points_per_object=int(rng.integers(4, 25)), static_track=(lo_s, lo_s + int(rng.integers(0, 6))),
                dynamic_track=(lo_d, lo_d + int(rng.integers(0, 6))), object_lifetime=int(rng.choice([0, 0, max(3, frames // 2)])),
                robust=bool(rng.integers(0, 2)), noise_scale=float(rng.choice([0.0, 0.5, 1.0, 2.0])), seed=int(rng.integers(0, 10_000)))


@pytest.mark.parametrize("case", range(24))
def test_random_graph_follows_the_oracle(case):
    from dynosam_amd.optimizer import Context
    rng = np.random.default_rng(1000 + case)
    kw = random_config(rng)
    kind = ["hybrid", "wcme", "wcpe"][case % 3]
    if kind != "hybrid":
        kw["objects"] = max(1, kw["objects"]); kw["object_lifetime"] = 0
    make = dict(hybrid=synth.make_hybrid_graph, wcme=synth.make_wcme_graph, wcpe=synth.make_wcpe_graph)[kind]
    g = make(synth.config(1, **kw))
    c = Context(); c.upload(g)
    og = O.OracleGraph(g)
    e_ref = og.error()
    assert abs(c.error() - e_ref) <= 1e-9 * max(1.0, abs(e_ref)), (kind, kw)
    lam = float(10.0 ** rng.integers(-6, 0))
    bad, d_ref, _dec = og.solve_damped(lam)
    if not bad:
        d, _ = c.solve_damped(lam)
        assert np.abs(d - d_ref).max() <= 1e-6 * max(1.0, np.abs(d_ref).max()), (kind, kw, lam)
    r = c.optimize()
    rr, _ = og.optimize()
    assert (r.iterations, r.inner_iterations) == (rr.iterations, rr.inner_iterations), (kind, kw)
    assert [bool(r.trace_accepted[i]) for i in range(r.trace_len)] == [bool(rr.trace_accepted[i]) for i in range(rr.trace_len)], (kind, kw)
    assert abs(r.error_after - rr.error_after) <= 1e-6 * max(rr.error_after, 1e-9), (kind, kw)
    if kind == "hybrid" and kw["frames"] >= 6:
        cut = int(rng.integers(2, kw["frames"] - 2))
        keys = [int(k) for k, f in zip(g.var_keys, g.meta["var_frame"]) if f < cut]
        st = c.values()
        try:
            rb, rp = WO.WindowOracle(g.with_state(st)).marginalize(keys, st)
        except np.linalg.LinAlgError:
            rp = None                                   # the oracle finds the elimination indeterminate: so must the device
        if rp is None:
            from dynosam_amd._lib import IndeterminantLinearSystemException
            with pytest.raises(IndeterminantLinearSystemException):
                c.marginalize(keys)
        else:
            _blocks, prior = c.marginalize(keys)
            assert np.array_equal(prior.keys, rp.keys), (kw, cut)
            assert np.abs(prior.Lambda - rp.Lambda).max() <= 1e-7 * max(1.0, np.abs(rp.Lambda).max()), (kw, cut)
            assert np.abs(prior.eta - rp.eta).max() <= 1e-7 * max(1.0, np.abs(rp.eta).max()), (kw, cut)
    c.close()
