"""why a settled tile set is slow: counters of the 8-tile run of config 4 late in the run"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, mgf_amd
from mgf_amd import scenes
ctx = mgf_amd.Context(0)
P = 8
scs = [scenes.sphere_pile_tile(128 // P, 128, 64, r, P) for r in range(P)]
worlds = []
for sc in scs:
    w = mgf_amd.World.from_scene(ctx, sc); w.set_tags(sc["tags"]); worlds.append(w)
T = mgf_amd.Tiles(ctx, worlds, [sc["x_range"] for sc in scs])
dt, it = float(scs[0]["dt"]), scs[0]["iters"]
names = ("capacity_retries", "flow6_fallbacks", "flow6_runs", "row_overflows", "solver_abort_fallbacks", "flow6_max_slots", "flow6_slot_cap", "flow6_max_foreign", "flow6_fcap", "flow6_fail_reason")
last = None
for s in range(1, 1701):
    if s % 100 == 1:
        torch.cuda.synchronize(); t0 = time.perf_counter()
    if s % 100 == 91:
        for w in worlds: w.set_option('time_solver_kernels', 1)
        solv = 0.0
    st = T.step(dt, it)
    if s % 100 > 90 or s % 100 == 0:
        solv += sum(float(x.ms_solver_kernels) for x in st)
    if s % 100 == 0:
        for w in worlds: w.set_option('time_solver_kernels', 0)
    if s % 100 == 0:
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 10
        cur = {k: [w.counter(k) for w in worlds] for k in names}
        d = {k: [a - b for a, b in zip(cur[k], last[k])] for k in names[:5]} if last else {k: cur[k] for k in names[:5]}
        print(f"tick {s}: {ms:.2f} ms per tick, solver kernels {solv / 10:.2f} ms per tick (last ten); per 100 ticks: " + ", ".join(f"{k} {d[k][3]}" for k in names[:5]) + f"; tile 3: slots {cur['flow6_max_slots'][3]}/{cur['flow6_slot_cap'][3]} foreign {cur['flow6_max_foreign'][3]}/{cur['flow6_fcap'][3]} reason {cur['flow6_fail_reason'][3]}; "
              f"constraints {int(st[3].n_constraints)}, hand-overs {sum(T.migrated(k) for k in range(P))}, owned {[len(w) for w in worlds]}", flush=True)
        last = cur
        xs = np.concatenate([w.state()["x"] for w in worlds])
        print(f"         bounds x [{xs[:,0].min():.1f}, {xs[:,0].max():.1f}] y [{xs[:,1].min():.1f}, {xs[:,1].max():.1f}] z [{xs[:,2].min():.1f}, {xs[:,2].max():.1f}]; bodies above y = 140: {(xs[:,1] > 140).sum()}, wide bodies {[w.counter('wide_bodies') for w in worlds]}, wide ticks {sum(w.counter('wide_ticks') for w in worlds)}", flush=True)
