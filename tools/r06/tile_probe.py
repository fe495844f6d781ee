"""r06: what the pair search of an x-slab tile of BASELINE config 4 sees - the brick kernel's slow queries, grid levels, cell fill - per tile, over the falling pile."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mgf_amd
from mgf_amd import scenes
ctx = mgf_amd.Context(0)
opts = [a.split("=") for a in sys.argv[1:] if "=" in a]
total, (nx, ny, nz) = 8, (16, 128, 64)
tsc = [scenes.sphere_pile_tile(nx, ny, nz, k, total, iters=10) for k in range(total)]
worlds = []
for sc in tsc:
    w = mgf_amd.World.from_scene(ctx, sc); w.set_tags(sc["tags"])
    for k, v in opts: w.set_option(k, int(v))
    worlds.append(w)
tiles = mgf_amd.Tiles(ctx, worlds, [sc["x_range"] for sc in tsc], first_tile=0, n_tiles_total=total, halo=1.0, refresh_every=2)
dt = float(tsc[0]["dt"])
names = ["pair_brick_slow_queries", "pair_brick_off_ticks", "grid_levels", "scene_ext_milli_x", "scene_ext_milli_y", "scene_ext_milli_z", "scene_rmax_milli_x"]
late = "--late-brick" in sys.argv
if late:
    for w in worlds: w.set_option("pair_brick", 0)
for t in range(71):
    if late and t == 3:
        for w in worlds: w.set_option("pair_brick", 1)
    tiles.step(dt, 10)
    if t in (0, 1, 2, 3, 4, 5, 10, 30, 70):
        for i in (0, 3):
            w = worlds[i]
            print(t, "tile", i, "n", len(w), {n: w.counter(n) for n in names}, flush=True)
