"""Per-rank timing of the sharded path with N in-process ranks on one GPU (ranks run one after the other inside each
collective-free segment only if the GPU is saturated; read the numbers as an upper bound)."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.zeros(1, device="cuda")
import numpy as np
from dynosam_amd import synth
from dynosam_amd.optimizer import Context, LevenbergMarquardtParams
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_multirank import FakeWorld

world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
cfg = synth.config(2)
cfg = synth.config(2, frames=cfg.frames * world, static_points=cfg.static_points * world, dynamic_points_per_object=cfg.dynamic_points_per_object * world)
g = synth.make_hybrid_graph(cfg)
fw = FakeWorld(world)
res = [None] * world
def body(r):
    c = Context(device=0, world_size=world, rank=r, allreduce=fw.allreduce(r))
    c.set_graphs(False)
    c.set_profiling(True)
    c.upload(g.shard(r, world))
    P = LevenbergMarquardtParams(); P.max_iterations = 6; P.relative_error_tol = 1e-300; P.absolute_error_tol = 0.0
    c.optimize(P); c.set_values(g.var_state); c.reset_kernel_stats()
    t = time.perf_counter(); rep = c.optimize(P); dt = time.perf_counter() - t
    res[r] = (dt, rep.iterations, rep.inner_iterations, rep.error_after, c.kernel_stats())
    c.close()
th = [threading.Thread(target=body, args=(r,)) for r in range(world)]
[t.start() for t in th]; [t.join() for t in th]
for r in range(world):
    dt, it, inner, err, ks = res[r]
    print(f"rank {r}: {1e3*dt/it:.2f} ms/iter ({it} it, {inner} solves) err {err:.6g}")
    for s in ks: print("     %-44s launches %6d total %9.3f ms" % (s["name"], s["launches"], s["total_ms"]))
print("factors", g.n_factors, "-> weak-scaling value", res[0][1] / res[0][0] * world)
