"""BASELINE config 3: sliding-window (20 keyframes, overlap 4) solves over a config-2-density stream, frontend stubbed
with the synthetic tracks.  Prints per-window host/GPU times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dynosam_amd import synth, sliding_window as SW

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 72
g = synth.make_hybrid_graph(synth.config(2, frames=frames, static_points=40 * frames, dynamic_points_per_object=2 * frames))
sw = SW.SlidingWindowOptimization(window_size=20, overlap=4)
sw.ctx.set_profiling(bool(os.environ.get('STATS')))
print("stream:", frames, "frames,", g.n_factors, "factors")
for k, blocks, vals in SW.frame_stream(g):
    t0 = time.perf_counter()
    if len(sw.frame_window) + 1 > sw.window_size:
        # time the pieces of optimize_window separately
        for kk in vals: sw.key_frame[int(kk)] = k
        sw.current_frame = k; sw.blocks += list(blocks); sw.values.update({int(a): b for a, b in vals.items()}); sw.frame_window.append(k)
        t1 = time.perf_counter()
        bl = sw._filter_valid(sw.blocks) + sw.prior_blocks
        gw = SW.flatten(sw.values, bl, sw.prior)
        t2 = time.perf_counter()
        sw.ctx.upload(gw)
        t3 = time.perf_counter()
        rep = sw.ctx.optimize(sw.params)
        t4 = time.perf_counter()
        if os.environ.get('STATS'):
            print('   solve_seconds %.1f ms' % (1e3 * rep.solve_seconds))
            for s_ in sw.ctx.kernel_stats(): print('      %-44s launches %6d total %9.3f ms' % (s_['name'], s_['launches'], s_['total_ms']))
            sw.ctx.reset_kernel_stats()
        if os.environ.get('WARM'):
            sw.ctx.set_values(gw.var_state); tw = time.perf_counter(); rw = sw.ctx.optimize(sw.params); print(f'   warm repeat LM {1e3*(time.perf_counter()-tw):.1f} ms ({rw.iterations} it)')
            for s_ in sw.ctx.kernel_stats(): print('      %-44s launches %6d total %9.3f ms' % (s_['name'], s_['launches'], s_['total_ms']))
        st = sw.ctx.values()
        result = {int(kx): (int(gw.var_type[i]), st[i].copy()) for i, kx in enumerate(gw.var_keys)}
        retained = {kx: v for kx, v in result.items() if sw.is_recent(kx)}
        to_marg = [kx for kx in result if kx not in retained]
        t5 = time.perf_counter()
        lb, pr = sw.ctx.marginalize(to_marg)
        t6 = time.perf_counter()
        sw.prior_blocks = [SW.keyed(b, gw.var_keys) for b in lb]; sw.prior = pr; sw.marginalized.update(to_marg)
        sw.frame_window = sw.frame_window[-sw.overlap:]; sw.blocks = []; sw.values = retained
        print(f"frame {k}: window {gw.n_factors} factors {gw.n_vars} vars | flatten {1e3*(t2-t1):.1f} ms  upload {1e3*(t3-t2):.1f}  "
              f"LM {1e3*(t4-t3):.1f} ({rep.iterations} it / {rep.inner_iterations} solves, {rep.error_before:.4g} -> {rep.error_after:.4g})  "
              f"download {1e3*(t5-t4):.1f}  marginalize {1e3*(t6-t5):.1f} (sep {0 if pr is None else len(pr.keys)} poses, {sum(b.count for b in lb)} containers)", flush=True)
    else:
        sw.update(blocks, vals, k)
