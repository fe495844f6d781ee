#!/usr/bin/env python
"""Headline benchmark of the mgf hot path on MI355X (contract: see the task statement).

    python bench.py [--gpus N] [--steps K] [--warmup W]

A "step" is one physics tick (mgf_demo/world.rs::World::step) of the BASELINE.json config-2
workload: a 262 144-sphere pile (64^3 jittered lattice, r = 0.5, seed 0x6D6766) in an open box,
dt = 1/60, 10 solver iterations, per GPU (weak scaling: x-slab tiles side by side for N > 1).
metric = contact-constraint-iterations per second over the WHOLE tick (one unit = one
ContactConstraint::solve call, solver.rs:203), whole-job aggregate over all ranks.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SOLVE_BYTES_PER_UNIT = 288  # algorithmic bytes per ContactConstraint::solve call
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E peak 8 TB/s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--tile", type=int, nargs=3, default=[64, 64, 64], help="spheres per GPU (nx ny nz)")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=2)
    ap.add_argument("--solver-mode", type=int, default=None, help="override the library's default solver mode (development)")
    ap.add_argument("--refresh-every", type=int, default=None, help="solver iterations between ghost velocity refreshes (multi-GPU; default: mgf_amd.tiles.DEFAULT_REFRESH_EVERY)")
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=VALUE", help="library option for an experiment (mgf_world_set_option), repeatable")
    ap.add_argument("--no-migrate", action="store_true", help="multi-GPU: keep every body on its initial tile (development: cost of the hand-over check)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (default); gloo = host-staged exchange, for validating the multi-rank flow on one GPU")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    if world_size != args.gpus:
        if world_size == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N "
                             "--master-addr 127.0.0.1 --master-port P bench.py --gpus N ...")
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world_size}")

    import torch
    dist = None
    dev_index = int(os.environ.get("MGF_BENCH_DEVICE", local_rank))
    if world_size > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(dev_index)
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world_size, device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world_size)

    import mgf_amd
    from mgf_amd import scenes
    from mgf_amd.tiles import DEFAULT_REFRESH_EVERY, TiledWorld
    refresh_every = args.refresh_every or DEFAULT_REFRESH_EVERY

    nx, ny, nz = args.tile
    ctx = mgf_amd.Context(dev_index)
    tw = TiledWorld(ctx, rank, world_size, nx, ny, nz, iters=args.iters, dist=dist, device=dev_index,
                    host_staging=(args.backend == "gloo"), refresh_every=refresh_every, migrate=not args.no_migrate)
    red_dev = "cuda" if args.backend == "nccl" else "cpu"
    dt = tw.dt

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize() if torch.cuda.is_available() else None

    if args.solver_mode is not None:
        tw.world.set_option("solver_mode", args.solver_mode)
    for kv in args.opt:
        key, val = kv.split("=")
        tw.world.set_option(key, int(val))
    # HIP events around the dominant kernel (k_solve_flow) on the stream it is launched on, inside the timed region
    tw.world.set_option("time_solver_kernels", 1)
    for _ in range(args.warmup):
        tw.step()
    barrier()
    t0 = time.perf_counter()
    units = 0
    cons = 0
    phase = dict(ms_integrate=0.0, ms_broadphase=0.0, ms_narrowphase=0.0, ms_setup=0.0, ms_solve=0.0)
    launches = 0
    levels = []
    kms = 0.0
    if tw.tile is None:
        # one GPU: the K ticks through the C-ABI's own loop (mgf_world_step_many: World::step K times, one synchronisation
        # per tick as always) - what a compiled host does; the per-tick statistics come back as an array
        per_tick = tw.world.step_many(dt, args.iters, args.steps)
    else:
        per_tick = [tw.step() for _ in range(args.steps)]
    barrier()
    elapsed = time.perf_counter() - t0
    for st in per_tick:
        units += st["n_constraints"] * args.iters
        cons += st["n_constraints"]
        launches += st["solver_kernel_launches"]
        levels.append(st["n_levels"])
        kms += st["ms_solver_kernels"]
        for k in phase:
            phase[k] += st[k]
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        u = torch.tensor([units, cons], dtype=torch.float64, device=red_dev)
        dist.all_reduce(u, op=dist.ReduceOp.SUM)
        units_all, cons_all = float(u[0].item()), float(u[1].item())
    else:
        units_all, cons_all = float(units), float(cons)

    # ---- roofline of the dominant kernel, rank 0: algorithmic bytes of the timed launches / their HIP-event time
    roofline = None
    if rank == 0 and kms > 0 and launches > 0:
        achieved = units * SOLVE_BYTES_PER_UNIT / (kms * 1e-3) / 1e9
        mode = args.solver_mode if args.solver_mode is not None else 5
        kname = {6: "k_solve_flow6 (ContactConstraint::solve, block-local persistent dataflow launch with message channels", 5: "k_solve_flow5 (ContactConstraint::solve, block-local persistent dataflow launch", 1: "k_solve_flow (ContactConstraint::solve, persistent dataflow launch",
                 4: "k_solve_flowk (ContactConstraint::solve, persistent dataflow launch", 0: "k_solve (ContactConstraint::solve, one launch per dependency frontier"}[mode]
        roofline = {"bound": "hbm", "kernel": kname + ": " + ("all iterations of a tick" if world_size == 1 else f"{refresh_every} iteration(s) between ghost refreshes") + ")",
                    "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5),
                    "traffic": _pmc_traffic(),
                    "bytes_per_unit": SOLVE_BYTES_PER_UNIT,
                    "avg_launch_us": round(kms * 1e3 / launches, 3), "launches_per_step": round(launches / args.steps, 2),
                    "avg_units_per_launch": round(units / launches, 1),
                    "launches_timed": int(launches)}

    # ---- CPU baseline: the oracle (C++ restatement of mgf), 1 core, bounded sample -------------
    cpu = None
    if rank == 0 and world_size == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(nx, ny, nz, args.iters, args.cpu_steps)

    if rank == 0:
        out = {
            "metric": "contact_constraint_iters_per_sec", "value": units_all / elapsed,
            "unit": "constraint-iters/s", "n_gpus": world_size, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"BASELINE config 2: {nx * ny * nz} spheres/GPU ({nx}x{ny}x{nz} jittered lattice pile, r=0.5, "
                                   f"seed 0x6D6766) in an open box, dt=1/60, {args.iters} solver iters"
                                   + ("" if world_size == 1 else f"; {world_size} x-slab tiles side by side, ghost halo over RCCL"),
                       "bodies_per_gpu": nx * ny * nz, "bodies_total": nx * ny * nz * world_size, "iters": args.iters,
                       "dt": dt, "constraint_order": "canonical (i asc; terrain DFS; partners j<i asc)",
                       "parallelism": "1 GPU" if world_size == 1 else f"{world_size} spatial x-slabs, neighbour halo exchange (ghost bodies once per tick, ghost velocities every {refresh_every} solver iterations, bodies handed to the tile that holds their centre" + ("" if not args.no_migrate else " - DISABLED") + ")"},
            "physics_steps_per_sec": args.steps / elapsed,
            "constraints_per_step": cons_all / args.steps,
            "solver_levels_mean": float(np.mean(levels)), "solver_launches_per_step": launches / args.steps,
            "phase_ms_per_step_rank0": {k: v / args.steps for k, v in phase.items()},
            "solve_phase_constraint_iters_per_sec_rank0": units / (phase["ms_solve"] * 1e-3) if phase["ms_solve"] > 0 else None,
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def _pmc_traffic():
    """HBM bytes per k_solve launch from committed rocprofv3 PMC passes (profiles/), or None."""
    p = os.path.join(ROOT, "profiles", "pmc_k_solve_flow5.json")
    if os.path.exists(p):
        try:
            return json.load(open(p)).get("hbm_bytes_per_launch")
        except Exception:
            return None
    return None


def cpu_baseline(nx, ny, nz, iters, steps):
    """The reference is single-threaded Rust that cannot be built here; its CPU path is timed as the
    oracle's C++ restatement ("port") on 1 host core, on the first `steps` ticks of the same scene."""
    from mgf_amd import scenes
    from oracle import oracle as O
    scene = scenes.sphere_pile(nx, ny, nz)
    w = O.World(O.ORDER_DEMO)
    t = scene["terrain"]
    w.set_terrain(t["verts"], t["faces"], t["pos"])
    w.add_bodies(scene["comps"], scene["mass"], scene["restitution"], scene["friction"], scene["force"])
    w.set_state(v=scene["v0"])
    units = 0
    solve_s = 0.0
    t0 = time.perf_counter()
    for _ in range(steps):
        st = w.step(float(scene["dt"]), iters)
        units += st.n_constraints * iters
        solve_s += st.t_solve
    el = time.perf_counter() - t0
    return {"value": units / el, "unit": "constraint-iters/s", "cores": 1, "kind": "port",
            "sample": f"first {steps} ticks of the same {nx * ny * nz}-sphere scene, world.rs order, {el:.1f} s of CPU work",
            "physics_steps_per_sec": steps / el, "solve_phase_constraint_iters_per_sec": units / solve_s if solve_s > 0 else None,
            "host_cpus_visible": os.cpu_count()}


if __name__ == "__main__":
    main()
