"""Tile-tick cost of the native driver (mgf_tiles_step) on one GPU: P tiles of nx x ny x nz spheres stepped in one process,
wall time per tile-tick, against the same bodies as one world where that fits."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
import mgf_amd
from mgf_amd import scenes

ap = argparse.ArgumentParser()
ap.add_argument("--tiles", type=int, default=2)
ap.add_argument("--dims", type=int, nargs=3, default=[64, 64, 64])
ap.add_argument("--warmup", type=int, default=10)
ap.add_argument("--steps", type=int, default=60)
ap.add_argument("--refresh-every", type=int, default=2)
a = ap.parse_args()
ctx = mgf_amd.Context(0)
nx, ny, nz = a.dims
scs = [scenes.sphere_pile_tile(nx, ny, nz, r, a.tiles) for r in range(a.tiles)]
worlds = []
for sc in scs:
    w = mgf_amd.World.from_scene(ctx, sc); w.set_tags(sc["tags"]); worlds.append(w)
T = mgf_amd.Tiles(ctx, worlds, [sc["x_range"] for sc in scs], refresh_every=a.refresh_every)
dt = float(scs[0]["dt"])
for _ in range(a.warmup):
    T.step(dt, 10)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(a.steps):
    st = T.step(dt, 10)
torch.cuda.synchronize(); el = time.perf_counter() - t0
print(f"{a.tiles} tiles of {nx}x{ny}x{nz}, R={a.refresh_every}: {el * 1e3 / a.steps / a.tiles:.3f} ms per tile-tick "
      f"({el * 1e3 / a.steps:.3f} ms per tick of all tiles); constraints per tile {[int(s.n_constraints) for s in st]}, of which ghost copies {[int(s.n_ghost_constraints) for s in st]}")
