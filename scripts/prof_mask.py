import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dynosam_amd import synth_images as SI
from dynosam_amd.flow import FlowTracker
rgb, mask = SI.make_sequence(640, 480, objects=3, frames=2, seed=4)
t = FlowTracker(640, 480)
t.upload(rgb[0], mask[0], rgb[1], mask[1])
for _ in range(3):
    t.boundary_mask(mask[0], 10, True)
t0 = time.perf_counter()
for _ in range(20):
    t.boundary_mask(mask[0], 10, True)
print("boundary_mask ms/call", 1e3 * (time.perf_counter() - t0) / 20)
