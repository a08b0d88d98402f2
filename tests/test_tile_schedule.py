"""Host logic of the tile-sparse Cholesky (dynosam_amd/csrc/tile_sym.h): the symbolic analysis and
the level schedule the HIP kernels consume are executed on the CPU with dense tile arithmetic
(csrc/tile_sym_check.cpp, g++) and compared with a dense solve.  No GPU."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "dynosam_amd", "csrc")


@pytest.fixture(scope="module")
def checker(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("tsc") / "tile_sym_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", os.path.join(CSRC, "tile_sym_check.cpp"), "-o", exe])
    return exe


def run(exe, *args):
    out = subprocess.run([exe, *map(str, args)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    kv = dict(tok.split("=") for tok in out.stdout.split() if "=" in tok)
    return {k: float(v) for k, v in kv.items()}


@pytest.mark.parametrize("args", [(60, 5, 0, 1), (60, 5, 1, 1), (200, 14, 1, 2), (37, 3, 1, 3, 10), (5, 2, 1, 4), (1, 0, 1, 5),
                                  (120, 8, 1, 6, 5), (90, 6, 1, 7, 0, 20)])
def test_schedule_solves_the_system(checker, args):
    r = run(checker, *args)
    assert r["residual"] < 1e-10


def test_twisted_order_halves_the_dependent_launches(checker):
    band = run(checker, 1200, 84, 0, 7)       # BASELINE config-2 shape: 1200 pose-like variables, 14-frame band
    twisted = run(checker, 1200, 84, 1, 7, 0, 560)
    assert band["levels"] == band["nt"]       # a band in frame order is one chain
    assert twisted["levels"] < 0.56 * band["levels"]
    assert twisted["residual"] < 1e-10 and band["residual"] < 1e-10


def test_row_tasks_solve_the_system(checker):
    """levels whose off-diagonal updates are packed into row tasks (several targets per workgroup) - forced for every level"""
    for args in ((200, 14, 1, 2), (1200, 84, 1, 7, 0, 560), (120, 8, 1, 6, 5)):
        out = subprocess.run([checker, *map(str, args)], capture_output=True, text=True, timeout=300, env=dict(os.environ, TS_ROW_MIN="0"))
        assert out.returncode == 0, out.stdout + out.stderr
        kv = dict(tok.split("=") for tok in out.stdout.split() if "=" in tok)
        assert float(kv["residual"]) < 1e-10
        plain = run(checker, *args)
        assert float(kv["fwd_tasks"]) < 0.9 * plain["fwd_tasks"]       # and they really are fewer, fatter tasks (0.4x on the wide band)


def test_sorted_and_bitmap_structure_agree(checker):
    """the tile structure is de-duplicated through a bit per tile up to 8192 tile columns and by a sort of the list above; both give
    the same schedule (levels, stored tiles, tasks) and solve the system"""
    for args in ((200, 14, 1, 2), (1200, 84, 1, 7, 0, 560), (37, 3, 1, 3, 10)):
        a = run(checker, *args)
        out = subprocess.run([checker, *map(str, args)], capture_output=True, text=True, timeout=300, env=dict(os.environ, TS_BITMAP_MAX="0"))
        assert out.returncode == 0, out.stdout + out.stderr
        b = {k: float(v) for k, v in (tok.split("=") for tok in out.stdout.split() if "=" in tok)}
        assert b["residual"] < 1e-10
        assert {k: v for k, v in a.items() if k != "residual"} == {k: v for k, v in b.items() if k != "residual"}


def test_deferred_updates_solve_the_system(checker):
    """src_cap: a target takes a bounded number of sources per launch and the rest in later launches, up to the launch that finalises its
    column (tile_sym.h: build_phase) - same launches, same solution; also across the two phases of a sharded solve and with row tasks.
    Structure 2 = a star of chains eliminated chains first (ten chains + hub): the hub's tiles collect one update per chain and level"""
    star = (660, 3, 2, 5, 0, 10)
    for args, env in ((star, {}), (star, {"TS_NELIM": "40"}), (star, {"TS_ROW_MIN": "0"}), ((1200, 84, 1, 7, 0, 560), {}), ((37, 3, 1, 3, 10), {})):
        res = {}
        for cap in (0, 1, 2, 5):
            out = subprocess.run([checker, *map(str, args)], capture_output=True, text=True, timeout=300, env=dict(os.environ, TS_SRC_CAP=str(cap), **env))
            assert out.returncode == 0, out.stdout + out.stderr
            res[cap] = {k: float(v) for k, v in (tok.split("=") for tok in out.stdout.split() if "=" in tok)}
            assert res[cap]["residual"] < 1e-10
            assert res[cap]["fwd_launches"] == res[0]["fwd_launches"] and res[cap]["levels"] == res[0]["levels"]
        if args is star and "TS_NELIM" not in env:
            assert res[0]["max_src"] >= 3                              # the hub really collects several sources per launch
            if "TS_ROW_MIN" in env:                                    # every launch counts as wide: a small cap moves sources to later launches
                assert res[1]["early_src"] < res[0]["early_src"] and res[5]["early_src"] == res[0]["early_src"]
            else:                                                      # launches with few tasks never put off what has just arrived
                assert all(res[c]["early_src"] == res[0]["early_src"] and res[c]["max_src"] == res[0]["max_src"] for c in (1, 2, 5))


def test_split_tasks_solve_the_system(checker):
    """split_max: in a launch with many tasks a target with more sources is updated by up to three workgroups at once - the first in place, the others
    into zeroed scratch tiles (and scratch rhs segments for a diagonal target) that the target's task of the next launch adds and clears; the
    checker runs the tasks of a launch in random order and fails when a scratch tile is left non-zero at the end"""
    for args in ((660, 3, 2, 5, 0, 10), (660, 3, 2, 2, 20, 10), (1320, 2, 2, 9, 30, 10), (1200, 84, 1, 7, 0, 560)):
        res = {}
        for sp in (0, 1, 2):
            out = subprocess.run([checker, *map(str, args)], capture_output=True, text=True, timeout=300, env=dict(os.environ, TS_SPLIT=str(sp), TS_ROW_MIN="0"))
            assert out.returncode == 0, out.stdout + out.stderr
            res[sp] = {k: float(v) for k, v in (tok.split("=") for tok in out.stdout.split() if "=" in tok)}
            assert res[sp]["residual"] < 1e-10
            assert res[sp]["fwd_launches"] == res[0]["fwd_launches"]
        assert res[0]["scratch"] == 0
        if res[0]["max_src"] >= 2:
            assert res[1]["scratch"] > 0 and res[1]["fwd_tasks"] > res[0]["fwd_tasks"] and res[1]["max_src"] < res[0]["max_src"]


def test_split_tasks_in_a_two_phase_schedule_leave_no_scratch_at_the_phase_boundary(checker):
    """the sharded path (round 5: split tasks are on there too): phase A = the rank's interior, the all-reduce over [tiles | scratch | rhs],
    phase B = the separators.  A scratch tile handed out in one launch is added and cleared in the next launch of the SAME phase, never across
    the boundary - the checker fails if anything is left in a scratch tile or scratch rhs segment where the phases meet, or at the end"""
    used = 0
    for args, nel in (((660, 3, 2, 5, 0, 10), 40), ((1320, 2, 2, 9, 30, 10), 60), ((1200, 84, 1, 7, 0, 560), 150), ((660, 3, 2, 2, 20, 10), 55)):
        res = {}
        for sp in (0, 1, 2):
            out = subprocess.run([checker, *map(str, args)], capture_output=True, text=True, timeout=300,
                                 env=dict(os.environ, TS_SPLIT=str(sp), TS_ROW_MIN="0", TS_NELIM=str(nel)))
            assert out.returncode == 0, out.stdout + out.stderr
            res[sp] = {k: float(v) for k, v in (tok.split("=") for tok in out.stdout.split() if "=" in tok)}
            assert res[sp]["residual"] < 1e-10 and res[sp]["phases"] == 2
            assert res[sp]["fwd_launches"] == res[0]["fwd_launches"]
        assert res[0]["scratch"] == 0
        used += int(res[1]["scratch"] > 0)
    assert used >= 2
