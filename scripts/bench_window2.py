"""config 3 stream, second pass (buffers grown): per-window stage times with hipGraph replay on / off.  python scripts/bench_window2.py [frames]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dynosam_amd import synth, sliding_window as SW
from dynosam_amd.optimizer import Context
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 72
g = synth.make_hybrid_graph(synth.config(2, frames=frames, static_points=40 * frames, dynamic_points_per_object=2 * frames))
for graphs, native in ((True, False), (True, True), (False, True)):
    ctx = Context(); ctx.set_graphs(graphs)
    for rep in range(2):
        sw = (SW.NativeSlidingWindowOptimization if native else SW.SlidingWindowOptimization)(window_size=20, overlap=4, ctx=ctx)
        rows = []
        for k, blocks, vals in SW.frame_stream(g):
            t0 = time.perf_counter()
            r = sw.update(blocks, vals, k)
            if r.optimized:
                rows.append((k, r.n_factors if native else r.graph.n_factors, 1e3 * (time.perf_counter() - t0), r.report.iterations, r.report.inner_iterations, 1e3 * r.report.solve_seconds, r.timings_ms))
    print("graphs", graphs, "native", native)
    for k, nf, tot, it, inner, lm, tm in rows:
        print(f"  frame {k}: {nf} factors total {tot:.1f} ms | LM {tm['optimize']:.1f} ms ({it} it/{inner} solves, inside-library {lm:.1f}) flatten {tm['flatten']:.1f} upload {tm['upload']:.1f} download {tm['download']:.1f} marginalize {tm['marginalize']:.1f} bookkeeping {tm['bookkeeping']:.1f}")
    ctx.close()
