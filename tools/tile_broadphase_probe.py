"""Development probe: broadphase time of a tile's bodies stepped as a plain world, and inside the tile protocol."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
import mgf_amd
from mgf_amd import scenes
from mgf_amd.tiles import HipEngine, Tile, step_tiles_inprocess

ctx = mgf_amd.Context(0)
sc = scenes.sphere_pile_tile(64, 64, 64, 0, 2)
w = mgf_amd.World.from_scene(ctx, sc)
w.set_option('phase_timing', 1)
bp = []
for s in range(40):
    st = w.step(float(sc["dt"]), 10)
    if s >= 10: bp.append(st.ms_broadphase)
print(f"plain world on tile 0's bodies: broadphase {np.mean(bp):.3f} ms, pair candidates {st.n_pair_candidates}, constraints {st.n_constraints}")
tiles = []
for r in range(2):
    s2 = scenes.sphere_pile_tile(64, 64, 64, r, 2)
    tiles.append(Tile(HipEngine(ctx, s2, 0), s2["x_range"], r, 2, s2["dt"], s2["iters"]))
bp = [[], []]
for s in range(40):
    stats = step_tiles_inprocess(tiles)
    if s >= 10:
        for r in range(2): bp[r].append(stats[r]["ms_broadphase"])
for r in range(2):
    print(f"tile {r}: broadphase {np.mean(bp[r]):.3f} ms, bodies {stats[r]['n_bodies']}, pair candidates {stats[r]['n_pair_candidates']}, constraints {stats[r]['n_constraints']}")
    sb = tiles[r].e.world
