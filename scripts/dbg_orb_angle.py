import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dynosam_amd import synth_images as SI
from dynosam_amd.flow import FlowTracker
from oracle import klt_oracle as K, orb_oracle as O
p = SI.make_pair(width=640, height=480, objects=3, seed=4)
g = K.gray_u8(p["rgb0"])
t = FlowTracker(640, 480); t.upload(p["rgb0"], p["mask0"], p["rgb1"], p["mask1"])
got = t.detect_orb(0); want = O.detect(g)
bad = np.nonzero(got["angle"] != want[3])[0]
print(len(bad), "of", len(want[3]))
for k in bad[:12]:
    print(k, got["angle"][k], want[3][k], got["angle"][k].view(np.uint32) - want[3][k].view(np.uint32), want[2][k])
