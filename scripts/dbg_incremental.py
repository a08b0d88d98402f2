import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from dynosam_amd import synth, sliding_window as SW, graph as G, symbols as S
from dynosam_amd._lib import IndeterminantLinearSystemException
from dynosam_amd.incremental import *
from dynosam_amd.optimizer import Context
g = synth.make_hybrid_graph(synth.config(1, frames=14, static_points=40, dynamic_points_per_object=10, static_track=(3, 6), dynamic_track=(3, 6), seed=3))
sm = FixedLagSmoother(lag=6.0, ctx=Context())
for k, blocks, vals in SW.frame_stream(g):
    a = UpdateArguments(blocks, vals, {key: float(k) for key in vals})
    snap = sm.snapshot()
    try:
        r = sm.update(a)
        print(k, "ok", r.error_before, r.error_after, len(sm.values), len(r.marginalized_keys), {kk: round(v, 2) for kk, v in r.timings_ms.items()})
    except IndeterminantLinearSystemException as e:
        key = e.nearby_variable
        print(k, "ILS key chr", chr(key >> 56), "label", (key >> 48) & 255, "index", key & ((1 << 48) - 1), "new vars", [(chr(x >> 56), (x >> 48) & 255, x & 0xffffffff) for x in vals if (x >> 56) in (ord('X'), ord('H'))])
        sm.restore(snap); sm.detect_indeterminate = False; r = sm.update(a); sm.detect_indeterminate = True
        print("   LM without the check:", r.error_before, r.error_after)
