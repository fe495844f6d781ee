"""World::step in a host loop (one C-ABI call and one wait per tick) against mgf_world_step_many (the next tick enqueued before the wait)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mgf_amd
from mgf_amd import scenes
ctx = mgf_amd.Context(0)
for name, sc, warm in (("config 2", scenes.sphere_pile(64, 64, 64), 5), ("config 5", scenes.dumbbell_field(64, 16, 64), 80)):
    dt, it = float(sc["dt"]), sc["iters"]
    w = mgf_amd.World.from_scene(ctx, sc)
    w.step_many(dt, it, warm)
    snap = w.clone()
    res = {}
    for mode in ("step_many", "step loop", "step_many", "step loop"):
        x = snap.clone()
        t0 = time.perf_counter()
        if mode == "step_many":
            x.step_many(dt, it, 20)
        else:
            for _ in range(20):
                x.step(dt, it)
        res.setdefault(mode, []).append((time.perf_counter() - t0) * 1e3 / 20)
        del x
    print(name, {k: [round(v, 4) for v in vs] for k, vs in res.items()}, flush=True)
