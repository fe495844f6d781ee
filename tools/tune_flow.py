"""Solver-mode timing sweep on one GPU (development aid): python tools/tune_flow.py [n] [mode:bpc:sleep[:k] ...]"""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, mgf_amd
from mgf_amd import scenes
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
specs = sys.argv[2:] or ['0:0:0', '1:0:2', '4:0:2']
ctx = mgf_amd.Context(0)
sc = scenes.sphere_pile(n, n, n)
for spec in specs:
    f = [int(x) for x in spec.split(':')] + [0, 0, 0]
    mode, bpc, sl, k = f[0], f[1], f[2], f[3]
    w = mgf_amd.World.from_scene(ctx, sc)
    w.set_option('phase_timing', 1)
    w.set_option('solver_mode', mode)
    if bpc: w.set_option('flow_blocks_per_cu', bpc)
    w.set_option('flow_sleep', sl)
    if k: w.set_option('flow_records_per_lane', k)
    ms = []; tot = []; C = 0
    for s in range(40):
        t0 = time.perf_counter(); st = w.step(float(sc['dt']), 10); el = time.perf_counter() - t0
        if s >= 10: ms.append(st.ms_solve); tot.append(el * 1e3); C = st.n_constraints
    v = w.state()["v"]
    extra = f" flow5 blocks {w.counter('flow5_blocks')} fallbacks {w.counter('flow5_fallbacks')} classes {w.counter('flow5_class0')}/{w.counter('flow5_class1')}/{w.counter('flow5_class2')}" if mode == 5 else ""
    print(f"n={n} mode {mode} blocks/cu {bpc} sleep {sl} k {k}: solve {np.mean(ms):.3f} ms  tick {np.mean(tot):.3f} ms  C={C} vsum={float(np.abs(v).sum()):.6f}{extra}", flush=True)
    del w
