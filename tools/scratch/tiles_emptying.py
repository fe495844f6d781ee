"""tiles that run empty: a small pile drifting hard along +x through 4 tiles until the left tiles own nothing, against the oracle's tiles, bit for bit"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, mgf_amd
from mgf_amd import scenes
from mgf_amd.tiles import Tile, step_tiles_inprocess
from tests.oracle_engine import OracleEngine
ctx = mgf_amd.Context(0)
P, dims, drift, ticks = 4, (6, 6, 6), (25.0, 0.0, 0.0), int(sys.argv[1]) if len(sys.argv) > 1 else 160
scs = [scenes.sphere_pile_tile(*dims, r, P, drift=drift) for r in range(P)]
worlds = []
for sc in scs:
    w = mgf_amd.World.from_scene(ctx, sc); w.set_tags(sc["tags"]); worlds.append(w)
T = mgf_amd.Tiles(ctx, worlds, [sc["x_range"] for sc in scs])
ot = [Tile(OracleEngine(sc), sc["x_range"], r, P, sc["dt"], sc["iters"]) for r, sc in enumerate(scs)]
dt, it = float(scs[0]["dt"]), scs[0]["iters"]
t0 = time.time()
for s in range(1, ticks + 1):
    T.step(dt, it)
    step_tiles_inprocess(ot)
    if s % 20 == 0:
        owned = [len(w) for w in worlds]
        for k, w in enumerate(worlds):
            assert np.array_equal(w.tags(), ot[k].e.tags()), (s, k)
            g, o = w.state(), ot[k].e.state()
            for f in ("x", "q", "v", "omega"):
                assert np.array_equal(g[f].view(np.uint32), o[f].view(np.uint32)), (s, k, f)
        print(f"tick {s}: owned {owned}, bit-identical to the oracle's tiles [{time.time() - t0:.0f} s]", flush=True)
print("OK")
