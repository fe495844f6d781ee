"""Development probe: how stale does the store's order get between two re-sorts (host_perm.inc)?
   python tools/store_probe.py [resort_every] [ticks]
Prints, per tick: ticks since the last re-sort, mean |slot - cell rank|, share of body pairs straddling 1024-body blocks cut from the
cell order and from the slots."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
import mgf_amd
from mgf_amd import scenes

every = int(sys.argv[1]) if len(sys.argv) > 1 else 32
ticks = int(sys.argv[2]) if len(sys.argv) > 2 else 100
ctx = mgf_amd.Context(0)
sc = scenes.sphere_pile(64, 64, 64)
w = mgf_amd.World.from_scene(ctx, sc)
w.set_option("resort_every", every)
dt, iters = float(sc["dt"]), sc["iters"]
last = 0
for t in range(ticks):
    w.step(dt, iters)
    r = w.counter("store_resorts")
    if t % 8 == 7 or r != last or t > ticks - 2:
        print(f"tick {t:4d} resorts {r:3d} displacement {w.counter('store_displacement_x1000') / 1000.0:10.1f} "
              f"cross by cell {w.counter('cross_block_pairs_by_cell_ppm') / 1e4:6.2f} %  by slot {w.counter('cross_block_pairs_by_slot_ppm') / 1e4:6.2f} %  "
              f"morton(x) {w.counter('cross_probe_morton') / 1e4:6.2f} %  hilbert(x) {w.counter('cross_probe_hilbert') / 1e4:6.2f} %  C {w.stats.n_constraints}")
    last = r
