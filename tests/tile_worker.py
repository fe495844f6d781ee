"""Worker for the multi-rank gloo tests: one tile per process, real torch.distributed point-to-point
transport.  Engine "oracle" (CPU, launched by tests/test_tiles_cpu.py) or "hip" (every rank on GPU 0,
payloads staged through the host, launched by tests/test_gpu_migration.py)."""
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
    engine = sys.argv[7] if len(sys.argv) > 7 else "oracle"
    dist.init_process_group(backend="gloo")
    rank, ws = dist.get_rank(), dist.get_world_size()
    scene = scenes.sphere_pile_tile(nx, ny, nz, rank, ws, drift=(drift, 0.0, 0.0) if drift else None)
    if engine == "hip":
        import mgf_amd
        from mgf_amd.tiles import HipEngine
        torch.cuda.set_device(0)
        eng = HipEngine(mgf_amd.Context(0), scene, 0)
    else:
        eng = OracleEngine(scene)
    tile = Tile(eng, scene["x_range"], rank, ws, scene["dt"], scene["iters"])
    tr = DistTransport(dist, rank, ws, host_staging=(engine == "hip"))
    ncons = []
    for _ in range(ticks):
        ncons.append(step_tile(tile, tr)["n_constraints"])
    s = tile.e.state()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), ncons=np.asarray(ncons), tags=tile.e.tags(), **s)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
