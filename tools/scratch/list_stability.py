"""How often does a tick's (a, b) constraint list equal the previous tick's?  (if often, the solver's tables could be reused.)"""
import sys, numpy as np
sys.path.insert(0, '.')
import mgf_amd
from mgf_amd import scenes
ctx = mgf_amd.Context(0)
sc = scenes.sphere_pile(64, 64, 64)
dt, it = float(sc["dt"]), sc["iters"]
w = mgf_amd.World.from_scene(ctx, sc)
for start in (400, 1000, 2000):
    w.step_many(dt, it, start - w.counter("ticks") if False else (start if start == 400 else (600 if start == 1000 else 1000)))
    prev = None
    same = 0
    for k in range(12):
        w.step(dt, it)
        c = w.constraints()
        ab = np.stack([c["a"], c["b"]], axis=1)
        if prev is not None:
            eq = prev.shape == ab.shape and np.array_equal(prev, ab)
            same += int(eq)
            if not eq:
                sa = set(map(tuple, prev.tolist())); sb = set(map(tuple, ab.tolist()))
                print(f"  tick ~{start}+{k}: {len(ab)} constraints, {len(sa ^ sb)} differ from the previous tick's", flush=True)
        prev = ab
    print(f"around tick {start}: {same} of 11 consecutive lists identical", flush=True)
