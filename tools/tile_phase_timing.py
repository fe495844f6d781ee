"""Where a tile's tick spends its host time: every HipEngine call of the tile protocol wrapped in a wall-clock
accumulator (calls that wait for the GPU include the wait).  Two tiles in one process, one GPU."""
import argparse
import os
import sys
import time
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401

import mgf_amd  # noqa: E402
from mgf_amd import scenes  # noqa: E402
from mgf_amd.tiles import HipEngine, Tile, step_tiles_inprocess  # noqa: E402

ACC = defaultdict(float)
CALLS = defaultdict(int)


def wrap(obj, name):
    fn = getattr(obj, name)

    def timed(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            ACC[name] += time.perf_counter() - t0
            CALLS[name] += 1
    setattr(obj, name, timed)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nx", type=int, default=32)
    ap.add_argument("--ny", type=int, default=64)
    ap.add_argument("--nz", type=int, default=128)
    ap.add_argument("--ticks", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=15)
    ap.add_argument("--refresh-every", type=int, default=2)
    a = ap.parse_args()
    ctx = mgf_amd.Context(0)
    tiles = []
    for r in range(2):
        sc = scenes.sphere_pile_tile(a.nx, a.ny, a.nz, r, 2)
        e = HipEngine(ctx, sc, 0)
        tiles.append(Tile(e, sc["x_range"], r, 2, sc["dt"], sc["iters"], refresh_every=a.refresh_every))
    for _ in range(a.warmup):
        step_tiles_inprocess(tiles)
    for t in tiles:
        for name in ("begin_tick", "select_tile", "export_bodies", "kinds", "add_kinds", "import_ghosts", "collide", "solve_iterations",
                     "export_velocities", "import_ghost_velocities", "finish", "export_migrants", "apply_migration"):
            wrap(t.e, name)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.ticks):
        step_tiles_inprocess(tiles)
    torch.cuda.synchronize()
    total = time.perf_counter() - t0
    per = 1e3 / (a.ticks * 2)
    print(f"{total * per:.3f} ms per tile-tick; host time inside engine calls, ms per tile-tick:")
    for k, v in sorted(ACC.items(), key=lambda kv: -kv[1]):
        print(f"  {k:26s} {v * per:7.3f}  ({CALLS[k] / (a.ticks * 2):.1f} calls)")
    print(f"  {'(python between calls)':26s} {(total - sum(ACC.values())) * per:7.3f}")


if __name__ == "__main__":
    main()
