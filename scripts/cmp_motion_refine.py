import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
from oracle import oracle_py as O
O.lib(); O.set_threads(8)
import numpy as np
from test_gpu_motion_refine import reference, scene, problem, K
from dynosam_amd import motion_refine as MR
from dynosam_amd.flow import FlowTracker
t = FlowTracker(64, 48)
solve = MR.gpu_solver()
for args in [(100,11,0),(128,11,0),(129,11,0),(160,11,0),(200,11,0),(256,11,0),(256,12,0),(200,13,0)]:
    s = scene(args[0], seed=args[1], n_out=args[2])
    r = reference(O, s); g = MR.optimize_batch(t, K, [problem(s)])[0]
    one = MR.optimize(solve, K, 3, 4, 2, s["X0"], s["X1"], s["H0"], s["tr"], s["kp0"], s["kp1"], s["l0"], s["l1"])
    print(args, (g["iterations"], g["inner_iterations"]), (r["iterations"], r["inner_iterations"]), "vs oracle", abs(g["error_after"]-r["error_after"])/r["error_after"], np.abs(g["best_result"]-r["best_result"]).max(),
          "vs main", abs(g["error_after"]-one["error_after"])/one["error_after"], np.abs(g["best_result"]-one["best_result"]).max(), "main vs oracle", np.abs(one["best_result"]-r["best_result"]).max())
