"""Cost of the tile protocol on one GPU: P tiles of an (P*nx) x ny x nz pile stepped in one process (shared
stream, so tiles run back to back), with and without the migration check.  Prints ms per tick."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401  (before libmgf_hip.so)

import mgf_amd  # noqa: E402
from mgf_amd import scenes  # noqa: E402
from mgf_amd.tiles import HipEngine, Tile, step_tiles_inprocess  # noqa: E402


def run(ctx, P, nx, ny, nz, ticks, warmup, **kw):
    tiles = []
    for r in range(P):
        sc = scenes.sphere_pile_tile(nx, ny, nz, r, P)
        tiles.append(Tile(HipEngine(ctx, sc, 0), sc["x_range"], r, P, sc["dt"], sc["iters"], **kw))
    for _ in range(warmup):
        step_tiles_inprocess(tiles)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(ticks):
        step_tiles_inprocess(tiles)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / ticks
    return ms, sum(t.n_migrated_in for t in tiles)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiles", type=int, default=2)
    ap.add_argument("--nx", type=int, default=32)
    ap.add_argument("--ny", type=int, default=32)
    ap.add_argument("--nz", type=int, default=64)
    ap.add_argument("--ticks", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--refresh-every", type=int, nargs="*", default=[2])
    a = ap.parse_args()
    ctx = mgf_amd.Context(0)
    for R in a.refresh_every:
        for migrate in (False, True):
            ms, moved = run(ctx, a.tiles, a.nx, a.ny, a.nz, a.ticks, a.warmup, migrate=migrate, refresh_every=R)
            print(f"tiles={a.tiles} bodies/tile={a.nx * a.ny * a.nz} refresh_every={R} migrate={migrate}: {ms:.3f} ms/tick ({moved} hand-overs)")


if __name__ == "__main__":
    main()
