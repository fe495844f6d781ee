"""One rank of a multi-rank tile run (tests/test_gpu_multi_device.py): its own process, its own device (or, with the stand-in transport
of tests/fake_rccl, device 0 shared with the other ranks), its share of the tiles;
the RCCL id travels from rank 0 through a multiprocessing queue - nothing but the C-ABI (mgf_rccl_unique_id, mgf_tiles_connect,
mgf_tiles_step) touches the fabric."""
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def run_rank(rank, n_ranks, total_tiles, dims, drift, ticks, fail_rank, fail_tick, uid_q, out_q, device=None, rccl_lib=None, world_opts=None, extra_env=None):
    """device: the rank's device (default: its own, `rank`); rccl_lib: the library the C-ABI binds instead of librccl (MGF_RCCL_LIB -
    the tests' stand-in that lets several ranks share one device, tests/fake_rccl)."""
    try:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        for k, v in (extra_env or {}).items():
            os.environ[k] = v
        if rccl_lib:
            os.environ["MGF_RCCL_LIB"] = rccl_lib
            os.environ.setdefault("MGF_FAKE_RCCL_TIMEOUT_S", "90")
        import numpy as np
        import torch  # noqa: F401  (before libmgf_hip.so: see tests/conftest.py)
        import mgf_amd
        from mgf_amd import scenes
        if rccl_lib:
            mgf_amd.rccl_allow_override(True)
        per = total_tiles // n_ranks
        first = rank * per
        ctx = mgf_amd.Context(rank if device is None else device)
        halo = 1.0
        if dims == "two_kinds":  # (r06: two-part bodies on the left ranks, plain spheres on the right: ghost records of two widths across a RANK face)
            from tests.util import two_kinds_tile_scenes
            tile_scenes = two_kinds_tile_scenes(total_tiles)[first:first + per]
            halo = 2.0
        else:
            nx, ny, nz = dims
            tile_scenes = [scenes.sphere_pile_tile(nx, ny, nz, first + k, total_tiles, drift=drift) for k in range(per)]
        worlds = []
        for sc in tile_scenes:
            w = mgf_amd.World.from_scene(ctx, sc)
            w.set_tags(sc["tags"])
            for key, val in (world_opts or {}).items():
                w.set_option(key, val)
            worlds.append(w)
        T = mgf_amd.Tiles(ctx, worlds, [sc["x_range"] for sc in tile_scenes], first_tile=first, n_tiles_total=total_tiles, halo=halo)
        if rank == 0:
            uid = mgf_amd.rccl_unique_id()
            for _ in range(n_ranks - 1):
                uid_q.put(uid)
        else:
            uid = uid_q.get(timeout=120)
        T.connect(uid, rank, n_ranks)
        seen = T.preflight()
        if rank == fail_rank:
            T.set_option("test_fail_tick", fail_tick)
        dt, iters = float(tile_scenes[0]["dt"]), tile_scenes[0]["iters"]
        failed_at, err = None, None
        for t in range(ticks):
            try:
                T.step(dt, iters)
            except mgf_amd.MgfError as e:
                failed_at, err = t, str(e)
                break
        tiles_out = []
        if failed_at is None:
            for k, w in enumerate(worlds):
                st = w.state()
                tiles_out.append(dict(tile=first + k, tags=w.tags(), migrated_in=T.migrated(k), **{f: st[f] for f in ("x", "q", "v", "omega", "delta")}))
        out_q.put(dict(rank=rank, ranks_seen=seen, failed_at=failed_at, error=err, tiles=tiles_out,
                       bytes_out=T.counter("exchange_bytes_out"), bytes_in=T.counter("exchange_bytes_in"),
                       flow6_runs=sum(w.counter("flow6_runs") for w in worlds), flow_blocks=[w.counter("flow5_blocks") for w in worlds],
                       ticks_retried=T.counter("ticks_retried")))
    except Exception:
        out_q.put(dict(rank=rank, crash=traceback.format_exc()))
