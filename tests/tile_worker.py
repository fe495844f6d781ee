"""Worker for the world_size-2 gloo test: one tile per process, oracle engine, real
torch.distributed point-to-point transport.  Launched by tests/test_tiles_cpu.py."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from mgf_amd import scenes  # noqa: E402
from mgf_amd.tiles import DistTransport, Tile, step_tile  # noqa: E402
from tests.oracle_engine import OracleEngine  # noqa: E402


def main():
    out_dir, nx, ny, nz, ticks = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    drift = float(sys.argv[6]) if len(sys.argv) > 6 else 0.0
    dist.init_process_group(backend="gloo")
    rank, ws = dist.get_rank(), dist.get_world_size()
    scene = scenes.sphere_pile_tile(nx, ny, nz, rank, ws, drift=(drift, 0.0, 0.0) if drift else None)
    tile = Tile(OracleEngine(scene), scene["x_range"], rank, ws, scene["dt"], scene["iters"])
    tr = DistTransport(dist, rank, ws)
    ncons = []
    for _ in range(ticks):
        ncons.append(step_tile(tile, tr)["n_constraints"])
    s = tile.e.state()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), ncons=np.asarray(ncons), tags=tile.e.tags(), **s)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
