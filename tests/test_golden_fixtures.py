"""The committed golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py) are
reproduced bit-for-bit by the generator's seeded inputs and to 1e-12 by the oracle."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from make_golden import CASES  # noqa: E402

from dynosam_amd import synth  # noqa: E402


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_reproduces_golden(oracle, name):
    kw = dict(CASES[name])
    g = synth.make_hybrid_graph(synth.config(kw.pop("n"), **kw))
    z = np.load(os.path.join(HERE, "golden", name + ".npz"))
    assert np.array_equal(g.var_state, z["init_state"])          # seeded inputs are bit-exact
    assert np.array_equal(g.var_keys, z["var_keys"])              # bit-exact variable indexing
    og = oracle.OracleGraph(g)
    r, _ = og.optimize()
    assert r.iterations == int(z["iterations"]) and r.inner_iterations == int(z["inner_iterations"])
    assert abs(r.error_after - float(z["error_after"])) <= 1e-9 * float(z["error_after"])
    assert np.allclose(og.state(), z["final_state"], atol=1e-7)
