import cProfile, pstats, time, sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from dynosam_amd import synth_images as SI
from dynosam_amd.flow import FlowTracker
sc = SI.make_pair(640, 480, objects=3, seed=4)
t = FlowTracker(640, 480, device=0)
t.upload(sc["rgb0"], sc["mask0"], sc["rgb1"], sc["mask1"])
rng = np.random.default_rng(7)
ys, xs = np.nonzero(sc["mask0"] > 0)
pick = rng.choice(len(xs), 1000, replace=False)
kp = np.stack([xs[pick] + 0.5, ys[pick] + 0.5], -1)
prev = sc["mask0"][ys[pick], xs[pick]]
zeros = np.zeros(1000, np.int64)
for _ in range(5):
    t.dense_flow(download=False); t.track_dynamic(kp, prev, zeros, zeros)
def loop(n):
    for _ in range(n):
        t.dense_flow(download=False)
        r = t.track_dynamic(kp, prev, zeros, zeros)
        tm = t.timing()
t0 = time.perf_counter(); loop(200); print("ms per frame", (time.perf_counter() - t0) / 200 * 1e3)
t0 = time.perf_counter()
for _ in range(200): t.dense_flow(download=False)
t.track_dynamic(kp, prev, zeros, zeros)
print("dense only ms", (time.perf_counter() - t0) / 200 * 1e3)
t0 = time.perf_counter()
for _ in range(200): t.track_dynamic(kp, prev, zeros, zeros)
print("track only ms", (time.perf_counter() - t0) / 200 * 1e3)
pr = cProfile.Profile(); pr.enable(); loop(200); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
