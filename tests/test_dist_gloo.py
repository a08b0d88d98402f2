"""world_size-2 gloo test of the multi-GPU contract, on CPU: factors sharded by
point ownership, pose variables replicated, the reduced system is a SUM over ranks.  The oracle's
linearisation plays the per-rank kernel here (checker only); the all-reduce is real gloo."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _normal_eq(g, og):
    """dense J^T J, J^T b over the caller's variable order from the oracle's linearisation."""
    J, b, e = og.linearize()
    dim = np.where(g.var_type == 0, 6, 3)
    off = np.concatenate([[0], np.cumsum(dim)])
    n = off[-1]
    H, rhs = np.zeros((n, n)), np.zeros(n)
    f = 0
    for blk in g.blocks:
        for i in range(blk.count):
            cols = np.concatenate([off[v] + np.arange(dim[v]) for v in blk.var_idx[i]])
            A = np.concatenate([J[f, :, 6 * s:6 * s + dim[v]] for s, v in enumerate(blk.var_idx[i])], axis=1)
            H[np.ix_(cols, cols)] += A.T @ A
            rhs[cols] += A.T @ b[f]
            f += 1
    return H, rhs, e.sum()


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dynosam_amd import synth
    from oracle import oracle_py as O
    g = synth.make_hybrid_graph(synth.config(1, frames=10, static_points=24, dynamic_points_per_object=10))
    shard = g.shard(rank, world)
    H, rhs, err = _normal_eq(shard, O.OracleGraph(shard))
    payload = torch.from_numpy(np.concatenate([H.reshape(-1), rhs, [err]]))
    dist.all_reduce(payload, op=dist.ReduceOp.SUM)     # the one exchange step of the path
    # timing contract of bench.py: max over ranks
    t = torch.tensor([0.1 * (rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        Hf, rf, ef = _normal_eq(g, O.OracleGraph(g))
        full = np.concatenate([Hf.reshape(-1), rf, [ef]])
        q.put((float(np.abs(payload.numpy() - full).max() / np.abs(full).max()), float(t.item())))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_normal_equations_sum_to_full():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 500)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    rel, tmax = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert rel < 1e-12
    assert abs(tmax - 0.2) < 1e-12
