#!/usr/bin/env python
"""Headline benchmark of the mgf hot path on MI355X (contract: see the task statement).

    python bench.py [--gpus N] [--steps K] [--warmup W]

A "step" is one physics tick (mgf_demo/world.rs::World::step): complete_motion, integrate, broadphase, narrowphase,
ContactConstraint::new for every contact, Solver::solve(10 iterations).
metric = contact-constraint-iterations per second over the WHOLE tick (one unit = one ContactConstraint::solve call,
solver.rs:203), whole-job aggregate over all ranks; a constraint across a tile face is counted once.

Workloads (BASELINE.json `configs`):
  N = 1   config 2 - a 262 144-sphere pile (64^3 jittered lattice, r = 0.5, seed 0x6D6766) in an open box, one world.
          The K timed ticks follow the W warm-up ticks of the falling pile; the window is repeated from a snapshot until a
          second of GPU time has been measured (median window reported), and a second window after tick 400 (the settled
          pile, twice the constraints) is reported under "settled".
  N > 1   config 4 - 1 048 576 spheres (128 x 128 x 64) cut into 8 x-slab tiles of 16 lattice columns; rank r owns 8 / N
          consecutive tiles (strong scaling: the same scene, the same 8-tile decomposition and the same results at every N),
          neighbour exchange between tiles of a rank by device copies and between ranks by RCCL send/recv over xGMI, all
          under the C-ABI (mgf_tiles_*).  `--scene weak` gives every rank one tile of --tile spheres instead;
          `--gpus 1 --scene config4` runs the 8 tiles on one GPU.
          `--scene config5 --gpus N` (N > 1), or `--scene config5_tiles` at any N: BASELINE config 5 as BASELINE.json states it - 65 536
          bodies of a sphere and a capsule each, cut into 8 x-slab tiles of 8 lattice columns, rank r owning 8 / N of them.
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
KERNEL_NAMES = {6: "k_solve_flow6 (ContactConstraint::solve, block-local persistent dataflow launch with message channels",
                5: "k_solve_flow5 (ContactConstraint::solve, block-local persistent dataflow launch",
                1: "k_solve_flow (ContactConstraint::solve, persistent dataflow launch",
                4: "k_solve_flowk (ContactConstraint::solve, persistent dataflow launch",
                0: "k_solve (ContactConstraint::solve, one launch per dependency frontier"}
SCENE_NOTE = "body order = SURVEY 8d's seeded Fisher-Yates shuffle (SplitMix64, seed 0x6D6766)"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--scene", default="auto", choices=["auto", "config2", "config3", "config5", "config4", "config5_tiles", "weak"],
                    help="auto = config 2 at N = 1, config 4 at N > 1; config3 / config5 = those configurations alone on one GPU (their own line: profiles)")
    ap.add_argument("--tile", type=int, nargs=3, default=[64, 64, 64], help="spheres per tile (nx ny nz): config 2's world, or --scene weak's tile")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--min-seconds", type=float, default=5.0, help="N = 1: repeat the timed window from a snapshot until this much has been measured "
                                                                     "(the settled window likewise; the nested configs 3 and 5 for a third of it each)")
    ap.add_argument("--no-settled", action="store_true", help="N = 1: skip the second window after tick 400")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="N = 1: skip the nested one-window runs of BASELINE configs 3 and 5")
    ap.add_argument("--no-order-check", action="store_true", help="N = 1: skip the canonical-vs-world.rs-order deviation (a few seconds of host BVH replay)")
    ap.add_argument("--solver-mode", type=int, default=None, help="override the library's default solver mode (development)")
    ap.add_argument("--refresh-every", type=int, default=None, help="solver iterations between ghost velocity refreshes (tiles)")
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=VALUE", help="library option for an experiment (mgf_world_set_option), repeatable")
    ap.add_argument("--no-one-gpu-reference", action="store_true", help="N > 1: do not measure the same scene's 8 tiles on ONE GPU of this box first (rank 0, before "
                    "the ranks connect: ~15 s); same_workload_on_one_gpu then comes from the committed profiles/ and says so")
    ap.add_argument("--no-settled-tiles", action="store_true", help="tiles: skip the second window after tick 400 (the pile has collapsed and come to rest)")
    ap.add_argument("--no-migrate", action="store_true", help="tiles: keep every body on its initial tile (development)")
    ap.add_argument("--transport", default="native", choices=["native", "torch"],
                    help="native = mgf_tiles_* (RCCL under the C-ABI); torch = the Python driver over torch.distributed (--scene weak only)")
    ap.add_argument("--rccl-lib", default=None, metavar="PATH",
                    help="development: bind this library instead of librccl under the C-ABI (tests/fake_rccl: several ranks on ONE GPU); the line "
                         "says so - such a run validates the multi-rank flow, it measures no fabric")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend of the rendezvous / barrier (and of --transport torch): nccl = RCCL; gloo = host-staged, "
                         "for validating the multi-rank flow on one GPU")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` exactly as the N = 1 command is issued: this process becomes the launcher of its own N ranks
        # (one per GPU, the contract's torch.distributed.run form); rank 0's JSON line is the only thing on stdout
        raise SystemExit(_spawn_ranks(args))
    if world_size != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world_size}")
    scene_kind = args.scene
    if scene_kind == "auto":
        scene_kind = "config2" if world_size == 1 else "config4"
    if scene_kind == "config5" and world_size > 1:
        scene_kind = "config5_tiles"
    if scene_kind in ("config2", "config3") and world_size > 1:
        raise SystemExit("configs 2 and 3 are single-GPU workloads here; use --scene config4, config5 or weak with --gpus N")
    if scene_kind in ("config4", "config5_tiles") and 8 % world_size:
        raise SystemExit("configs 4 and 5 are cut into 8 tiles: --gpus must be 1, 2, 4 or 8")

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
    red_dev = "cuda" if args.backend == "nccl" else "cpu"

    import mgf_amd
    from mgf_amd import scenes
    from mgf_amd.tiles import DEFAULT_REFRESH_EVERY
    refresh_every = args.refresh_every or DEFAULT_REFRESH_EVERY
    args.standin = _standin_lib(args)
    if args.standin:
        # development: the multi-rank flow validated on ONE GPU over the test-suite's stand-in transport (tests/fake_rccl) - never a measurement of xGMI
        os.environ["MGF_RCCL_LIB"] = args.standin
        os.environ.setdefault("MGF_FAKE_RCCL_TIMEOUT_S", "120")
        mgf_amd.rccl_allow_override(True)
    ctx = mgf_amd.Context(dev_index)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize() if torch.cuda.is_available() else None

    def configure(world):
        if args.solver_mode is not None:
            world.set_option("solver_mode", args.solver_mode)
        for kv in args.opt:
            key, val = kv.split("=")
            world.set_option(key, int(val))

    configure.instrument = _instrument
    mode = args.solver_mode if args.solver_mode is not None else 6
    if scene_kind == "config2":
        out = bench_single_world(args, ctx, mgf_amd, scenes, configure, mode)
    elif scene_kind in ("config3", "config5"):
        out = bench_other_config(args, ctx, mgf_amd, scenes, configure, mode, scene_kind, standalone=True)
    else:
        out = bench_tiles(args, ctx, mgf_amd, scenes, configure, mode, scene_kind, rank, world_size, dist, torch, red_dev, barrier, refresh_every)
    if rank == 0:
        if world_size == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(args.iters)
            except Exception as e:  # (the line with the GPU measurement is printed whatever happens to the CPU sample)
                out["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


INSTRUMENTATION_NOTE = ("value / ms_per_step: wall clock over whole ticks with NO HIP events in the stream (an event is a barrier packet: the eight "
                        "phase events and the two around the solver kernel cost ~50 us of an idle GPU per tick); roofline, phase_ms and gpu_event_ms: "
                        "the SAME ticks replayed from the same snapshot (mgf_world_clone) with the events on - options phase_timing, "
                        "time_solver_kernels - median of three replays")


def _instrument(world):
    """HIP events at the phase boundaries and around the dominant kernel (on the stream it is launched on): for the replays that
    feed `roofline`, never in the wall-clock windows."""
    world.set_option("phase_timing", 1)
    world.set_option("time_solver_kernels", 1)


def _instrumented_replays(snap, configure, dt, iters, steps, n=3):
    """(units, cons, launches, kernel ms, phase ms) of the median (by solver kernel time) of n replays of `steps` ticks from `snap`."""
    runs = []
    for _ in range(n):
        w = snap.clone()
        configure(w)
        _instrument(w)
        runs.append(_window_stats(w.step_many(dt, iters, steps), iters))
        del w
    runs.sort(key=lambda r: r[3])
    return runs[len(runs) // 2]


def _window_stats(per_tick, iters):
    units = sum(int(st["n_constraints"]) * iters for st in per_tick)
    cons = sum(int(st["n_constraints"]) for st in per_tick)
    launches = sum(int(st["solver_kernel_launches"]) for st in per_tick)
    kms = sum(float(st["ms_solver_kernels"]) for st in per_tick)
    phase = {k: sum(float(st[k]) for st in per_tick) / len(per_tick) for k in ("ms_integrate", "ms_broadphase", "ms_narrowphase", "ms_setup", "ms_solve")}
    phase["ms_total"] = sum(float(st["ms_total"]) for st in per_tick) / len(per_tick)  # HIP events around the whole tick on its stream
    return units, cons, launches, kms, phase


def _roofline_tick(per_tick, iters, elapsed_s):
    """The whole tick against the HBM roofline: SURVEY.md 8(d)'s algorithmic bytes of every phase, summed over the timed ticks, over the
    wall time of the window - complete_motion + integrate 284 B per body; pair search 32 B per leaf record a query accepts by its fat
    box (+ the query's own); sphere pair narrowphase 2 x 44 B in + 64 B out per emitted contact; ContactConstraint::new 332 B per
    constraint; ContactConstraint::solve 288 B per call."""
    n = sum(int(st["n_bodies"]) for st in per_tick)
    c = sum(int(st["n_constraints"]) for st in per_tick)
    ct = sum(int(st["n_terrain_constraints"]) for st in per_tick)
    acc = sum(int(st["n_pair_candidates"]) for st in per_tick)
    terms = {"integrate": 284 * n, "broadphase": 32 * (acc + n), "narrowphase": 152 * (c - ct), "setup": 332 * c, "solve": SOLVE_BYTES_PER_UNIT * c * iters}
    total = sum(terms.values())
    gbs = total / elapsed_s / 1e9
    return {"bound": "hbm", "achieved": round(gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 5),
            "bytes_per_tick": round(total / len(per_tick)), "bytes_per_tick_by_phase": {k: round(v / len(per_tick)) for k, v in terms.items()},
            "note": "sum over the phases of SURVEY 8(d)'s algorithmic bytes / wall time of the timed window (the whole tick, not its best kernel)"}


def _roofline(units, launches, kms, mode, what, window, workload="config2"):
    if not (kms > 0 and launches > 0):
        return None
    achieved = units * SOLVE_BYTES_PER_UNIT / (kms * 1e-3) / 1e9
    traffic, source = _pmc_traffic(mode, window, workload)
    return {"bound": "hbm", "kernel": KERNEL_NAMES[mode] + ": " + what + ")", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": source, "bytes_per_unit": SOLVE_BYTES_PER_UNIT,
            "avg_launch_us": round(kms * 1e3 / launches, 3), "avg_units_per_launch": round(units / launches, 1), "launches_timed": int(launches)}


def _standin_lib(args):
    """The library named to stand in for librccl (several ranks on ONE GPU over tests/fake_rccl: the multi-rank flow, never a
    measurement of xGMI), or None: --rccl-lib, or MGF_RCCL_LIB together with MGF_BENCH_ALLOW_RCCL_OVERRIDE=1."""
    if args.rccl_lib:
        return os.path.abspath(args.rccl_lib)
    if os.environ.get("MGF_RCCL_LIB") and os.environ.get("MGF_BENCH_ALLOW_RCCL_OVERRIDE") == "1":
        return os.environ["MGF_RCCL_LIB"]
    return None


def _spawn_ranks(args):
    """Re-run this command line as N ranks under torch.distributed.run on 127.0.0.1 (a free port) and return its exit code."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    extra = []
    if _standin_lib(args):
        import torch
        if torch.cuda.device_count() < args.gpus:
            # the stand-in transport's reason to exist: every rank on device 0, the rendezvous host-staged (RCCL refuses two ranks on a device)
            env.setdefault("MGF_BENCH_DEVICE", "0")
            if "--backend" not in sys.argv:
                extra = ["--backend", "gloo"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:] + extra
    return subprocess.call(cmd, env=env)


def bench_single_world(args, ctx, mgf_amd, scenes, configure, mode):
    nx, ny, nz = args.tile
    raster = os.environ.get("MGF_BENCH_BODY_ORDER", "shuffled") == "raster"  # development probe: how much the gathers cost (NOT the workload)
    scene = scenes.sphere_pile(nx, ny, nz, shuffle=not raster)
    dt = float(scene["dt"])
    world = mgf_amd.World.from_scene(ctx, scene)
    configure(world)
    for _ in range(args.warmup):
        world.step(dt, args.iters)
    # the timed window: K ticks through the C-ABI's own loop (mgf_world_step_many: World::step K times, one synchronisation per
    # tick), repeated from a snapshot (mgf_world_clone) so that the measurement covers >= --min-seconds whatever K is
    snap = world.clone()
    windows = []
    total = 0.0
    while not windows or (total < args.min_seconds and len(windows) < 1000):
        w = snap.clone()
        configure(w)
        import torch
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        per_tick = w.step_many(dt, args.iters, args.steps)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        windows.append((el, _window_stats(per_tick, args.iters), _roofline_tick(per_tick, args.iters, el)))
        total += el
        del w
    windows.sort(key=lambda x: x[0])
    elapsed, (units, cons, _l, _k, _p), tick_roof = windows[len(windows) // 2]
    ru, rc, launches, kms, phase = _instrumented_replays(snap, configure, dt, args.iters, args.steps)
    assert (ru, rc) == (units, cons), "a replay of the window did other work than the window"
    window_name = f"ticks {args.warmup}..{args.warmup + args.steps} of the falling pile"
    out = {
        "metric": "contact_constraint_iters_per_sec", "value": units / elapsed, "unit": "constraint-iters/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed * 1e3 / args.steps, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"BASELINE config 2: {nx * ny * nz} spheres ({nx}x{ny}x{nz} jittered lattice pile, r=0.5, seed 0x6D6766) in an open "
                               f"box, dt=1/60, {args.iters} solver iters; {window_name}; " + ("BODY ORDER = LATTICE RASTER (development probe, not the BASELINE workload)" if raster else SCENE_NOTE),
                   "bodies_total": nx * ny * nz, "iters": args.iters, "dt": dt,
                   "constraint_order": "canonical (i asc; terrain DFS; partners j<i asc)", "parallelism": "1 GPU",
                   "scaling_note": "this N = 1 line is BASELINE config 2 (the configuration the metric is quoted on: one world, the fused tick); the N > 1 lines "
                                   "are STRONG-scaling slices of BASELINE config 4 (1 048 576 spheres as 8 tiles through the tile protocol), another workload: "
                                   "a per-N efficiency belongs to config 4's own one-GPU figures, which every N > 1 line carries as same_workload_on_one_gpu and "
                                   "efficiency_vs_same_workload_on_one_gpu (`scaling` here follows the contract's default for a single-GPU line)"},
        "timed_region": {"windows": len(windows), "seconds": round(total, 3), "reported": "median window",
                         "ms_per_step_min": windows[0][0] * 1e3 / args.steps, "ms_per_step_max": windows[-1][0] * 1e3 / args.steps},
        "lib_sha256": _lib_sha256(),
        "physics_steps_per_sec": args.steps / elapsed, "constraints_per_step": cons / args.steps,
        "solver_launches_per_step": launches / args.steps, "phase_ms_per_step_rank0": phase,
        "solve_phase_constraint_iters_per_sec_rank0": units / (phase["ms_solve"] * args.steps * 1e-3) if phase["ms_solve"] > 0 else None,
        "roofline": _roofline(units, launches, kms, mode, "all iterations of a tick", ("transient", args.warmup, args.steps)),
        "roofline_tick": tick_roof,
        # GPU activity as the stream itself saw it (HIP events around every tick of the instrumented replay; no sampler needed)
        "gpu_event_ms": {"replayed_window": round(phase["ms_total"] * args.steps, 3), "solver_kernel_ms_replayed_window": round(kms, 3)},
        "instrumentation": INSTRUMENTATION_NOTE,
    }
    if not args.no_order_check:
        try:
            out["constraint_order_deviation"] = order_deviation(ctx, mgf_amd, scene, dt, args.iters)
            out["constraint_order_demo_cost"] = demo_order_cost(ctx, mgf_amd, scenes, scene, args.iters)
        except Exception as e:
            out["constraint_order_deviation"] = {"error": repr(e)}
    if not args.no_settled:
        # the settled pile (what the workload spends its life in): twice the constraints, deeper dependency graph
        while_ticks = max(0, 400 - args.warmup - args.steps)
        w = world
        w.step_many(dt, args.iters, args.steps)
        if while_ticks:
            w.step_many(dt, args.iters, while_ticks)
        import torch
        snap2 = w.clone()
        wins2, total2 = [], 0.0
        while not wins2 or (total2 < args.min_seconds and len(wins2) < 1000):
            x = snap2.clone()
            configure(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            per_tick = x.step_many(dt, args.iters, args.steps)
            torch.cuda.synchronize()
            el2 = time.perf_counter() - t0
            wins2.append((el2, _window_stats(per_tick, args.iters), _roofline_tick(per_tick, args.iters, el2)))
            total2 += wins2[-1][0]
            del x
        wins2.sort(key=lambda q: q[0])
        el, (u2, c2, _l2, _k2, _p2), tick_roof2 = wins2[len(wins2) // 2]
        _u, _c, l2, k2, p2 = _instrumented_replays(snap2, configure, dt, args.iters, args.steps)
        del snap2
        out["settled"] = {"window": f"ticks {args.warmup + args.steps + while_ticks}..{args.warmup + 2 * args.steps + while_ticks}", "value": u2 / el, "windows": len(wins2),
                          "ms_per_step": el * 1e3 / args.steps, "constraints_per_step": c2 / args.steps, "phase_ms_per_step": p2,
                          "roofline": _roofline(u2, l2, k2, mode, "all iterations of a tick", ("settled", 400, args.steps)), "roofline_tick": tick_roof2}
    if not args.no_other_configs:
        # BASELINE configs 3 and 5 on the same GPU, one short window each (their own lines: python bench.py --scene config3 / config5)
        del world, snap
        for kind in ("config3", "config5"):
            try:
                out[kind] = bench_other_config(args, ctx, mgf_amd, scenes, configure, mode, kind, standalone=False)
            except Exception as e:  # (a nested figure: never at the price of the line)
                out[kind] = {"error": repr(e)}
    return out


OTHER_CONFIGS = {
    # name, scene builder, warm-up ticks (the bodies have to arrive somewhere first), what it is
    "config3": ("BASELINE config 3: 131 072 capsules (r = 0.5, |d| = 0.5, random orientations, 128x32x32 lattice of pitch 2.6) over a 49 928-triangle "
                "heightfield (158 x 158 quads, heights U(-0.2, 0.2)), 10 solver iters", lambda sc: sc.capsule_field(128, 32, 32, quads=158), 150),
    "config5": ("BASELINE config 5 (this build's own definition of a rigid body of two components - the reference has none, SURVEY H8): 65 536 bodies, "
                "each a sphere (r = 0.5) + a capsule (|d| = 1, r = 0.3), 64x16x64 lattice of pitch 2.2 in an open box, 10 solver iters",
                lambda sc: sc.dumbbell_field(64, 16, 64), 80),
}


def bench_other_config(args, ctx, mgf_amd, scenes, configure, mode, kind, standalone):
    """One world of BASELINE config 3 / 5 on one GPU: W warm-up ticks, then the K timed ticks from a snapshot (repeated while
    --min-seconds lasts when it is the run's own line)."""
    import torch
    what, build, warm_default = OTHER_CONFIGS[kind]
    scene = build(scenes)
    dt = float(scene["dt"])
    warmup = args.warmup if standalone and args.warmup != 10 else warm_default
    steps = args.steps
    world = mgf_amd.World.from_scene(ctx, scene)
    configure(world)
    world.step_many(dt, args.iters, warmup)
    snap = world.clone()
    windows, total = [], 0.0
    budget = args.min_seconds if standalone else args.min_seconds / 3.0
    # (nested in the config-2 line: three windows, the median's reported - a clone's first window pays for its buffers)
    while len(windows) < 3 or (total < budget and len(windows) < 1000):
        w = snap.clone()
        configure(w)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        per_tick = w.step_many(dt, args.iters, steps)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        windows.append((el, _window_stats(per_tick, args.iters), per_tick[len(per_tick) - 1], _roofline_tick(per_tick, args.iters, el)))
        total += el
        del w
    windows.sort(key=lambda x: x[0])
    elapsed, (units, cons, _l, _k, _p), last, tick_roof = windows[len(windows) // 2]
    _u, _c, launches, kms, phase = _instrumented_replays(snap, configure, dt, args.iters, steps)
    res = {
        "value": units / elapsed, "unit": "constraint-iters/s", "ms_per_step": elapsed * 1e3 / steps, "steps": steps, "warmup": warmup,
        "windows": len(windows), "bodies": len(world), "constraints_per_step": cons / steps, "terrain_constraints_last_tick": int(last["n_terrain_constraints"]),
        "accepted_pairs_last_tick": int(last["n_pair_candidates"]), "phase_ms_per_step": phase,
        "gpu_event_ms_replayed_window": round(phase["ms_total"] * steps, 3),
        "roofline": _roofline(units, launches, kms, mode, "all iterations of a tick", (kind, warmup, steps), workload=kind),
        "roofline_tick": tick_roof,
        "store_resorts": world.counter("store_resorts"),
    }
    if not standalone:
        res["workload"] = what
        return res
    res.update({"metric": "contact_constraint_iters_per_sec", "n_gpus": 1, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
                "data": "synthetic", "physics_steps_per_sec": steps / elapsed,
                "config": {"workload": what + f"; ticks {warmup}..{warmup + steps}; {SCENE_NOTE}", "bodies_total": len(world), "iters": args.iters, "dt": dt,
                           "constraint_order": "canonical (i asc; terrain DFS; partners j<i asc)", "parallelism": "1 GPU"}})
    return res


def order_deviation(ctx, mgf_amd, scene, dt, iters, ticks=(1,)):
    """The HIP path inserts constraints in canonical order; the reference's own order (world.rs:233-291) is available as
    option constraint_order = 1 (host replay of the world BVH).  Same constraint SET, different Gauss-Seidel order: the
    deviation between the two on this scene, max |a - b| / max(1, |b|) over x, q, v, omega after the given ticks."""
    a, b = mgf_amd.World.from_scene(ctx, scene), mgf_amd.World.from_scene(ctx, scene)
    b.set_option("constraint_order", 1)
    out, k = {}, 0
    for target in ticks:
        while k < target:
            sa, sb = a.step(dt, iters), b.step(dt, iters)
            k += 1
        x, y = a.state(), b.state()
        dev = max(float(np.max(np.abs(x[f].astype(np.float64) - y[f].astype(np.float64)) / np.maximum(1.0, np.abs(y[f].astype(np.float64))))) for f in ("x", "q", "v", "omega"))
        out[f"after_tick_{target}"] = {"max_rel_deviation": dev, "constraints": int(sa.n_constraints), "same_constraint_count": int(sa.n_constraints) == int(sb.n_constraints)}
    out["note"] = "canonical order (the timed path) vs the reference's world.rs order (option constraint_order = demo); the contract bar is 1e-4"
    return out


def demo_order_cost(ctx, mgf_amd, scenes, scene2, iters):
    """What choosing the reference's own insertion order costs (option constraint_order = demo: the world BVH of world.rs:233-291 is
    replayed on the host every tick, single-threaded like the reference): wall ms per tick beside the canonical order's, at 512
    bodies (BASELINE config 1), the unmodified demo's 1332, and config 2's 262 144."""
    import torch
    out = {}
    cases = (("config1_balls_512", scenes.balls_demo(8), 120, 40), ("demo_balls_1332_iters20", scenes.balls_demo(11, extra_ball=True, iters=20), 120, 40),
             ("config2_spheres_262144", scene2, 4, 3))
    for name, sc, warm, ticks in cases:
        dt, it = float(sc["dt"]), (sc["iters"] if name.startswith("demo") else iters)
        row = {}
        for order, key in ((0, "canonical_ms_per_tick"), (1, "demo_order_ms_per_tick")):
            w = mgf_amd.World.from_scene(ctx, sc)
            w.set_option("constraint_order", order)
            for _ in range(warm):
                w.step(dt, it)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(ticks):
                st = w.step(dt, it)
            torch.cuda.synchronize()
            row[key] = (time.perf_counter() - t0) * 1e3 / ticks
            row["constraints_last_tick"] = int(st.n_constraints)
            del w
        row["ticks_timed"] = ticks
        out[name] = row
    out["note"] = ("demo order = the reference's own insertion order, bit-identical to the oracle in world.rs order (tests/test_world_snapshots.py, "
                   "tests/test_gpu_fullsize.py); its tree replay is serial host work, so its cost grows with the body count")
    return out


def bench_tiles(args, ctx, mgf_amd, scenes, configure, mode, scene_kind, rank, world_size, dist, torch, red_dev, barrier, refresh_every):
    halo = 1.0
    if scene_kind == "config4":
        total_tiles, (nx, ny, nz) = 8, (16, 128, 64)
    elif scene_kind == "config5_tiles":
        total_tiles, (nx, ny, nz) = 8, (8, 16, 64)
        halo = 2.0  # (a two-part body's fat half extent along x reaches ~1.5)
    else:
        total_tiles, (nx, ny, nz) = world_size, tuple(args.tile)
    per_rank = total_tiles // world_size
    first = rank * per_rank
    whole5 = None
    if scene_kind == "config5_tiles":
        whole5 = scenes.dumbbell_field(total_tiles * nx, ny, nz, iters=args.iters)
        tile_scenes = scenes.split_by_slabs(whole5, total_tiles, total_tiles * nx * 2.2 / 2.0)[first:first + per_rank]
    else:
        tile_scenes = [scenes.sphere_pile_tile(nx, ny, nz, first + k, total_tiles, iters=args.iters) for k in range(per_rank)]
    dt = float(tile_scenes[0]["dt"])
    transport = args.transport
    ranks_seen = None
    # The same scene's 8 tiles on ONE GPU of THIS box (rank 0's device, the other ranks idle at the barrier): what the N > 1 value is an
    # efficiency against - measured here and now, not read from profiles/ (an efficiency across boxes is not a same-node figure)
    one_gpu_here = None
    if world_size > 1 and scene_kind in ("config4", "config5_tiles") and transport == "native" and not args.no_one_gpu_reference and not args.standin:
        if rank == 0:
            one_gpu_here = _one_gpu_reference_here(args, scene_kind)
        barrier()
    if transport == "torch":
        if scene_kind != "weak" or per_rank != 1:
            raise SystemExit("--transport torch drives one tile per rank (--scene weak)")
        from mgf_amd.tiles import TiledWorld
        tw = TiledWorld(ctx, rank, world_size, nx, ny, nz, iters=args.iters, dist=dist, device=int(os.environ.get("MGF_BENCH_DEVICE", os.environ.get("LOCAL_RANK", "0"))),
                        host_staging=(args.backend == "gloo"), refresh_every=refresh_every, migrate=not args.no_migrate)
        configure(tw.world)
        step = lambda: [tw.step()]  # noqa: E731
    else:
        worlds = []
        for sc in tile_scenes:
            w = mgf_amd.World.from_scene(ctx, sc)
            w.set_tags(sc["tags"])
            configure(w)
            worlds.append(w)
        tiles = mgf_amd.Tiles(ctx, worlds, [sc["x_range"] for sc in tile_scenes], first_tile=first, n_tiles_total=total_tiles, halo=halo,
                              refresh_every=refresh_every, migrate=not args.no_migrate)
        if world_size > 1:
            # pre-flight: the RCCL communicator under the C-ABI comes up and sees every rank (rank 0's id travels by torch.distributed)
            uid = torch.zeros(128, dtype=torch.uint8, device=red_dev)
            if rank == 0:
                uid.copy_(torch.frombuffer(bytearray(mgf_amd.rccl_unique_id()), dtype=torch.uint8))
            dist.broadcast(uid, 0)
            tiles.connect(bytes(uid.cpu().numpy().tobytes()), rank, world_size)
            ranks_seen = tiles.preflight()
            if ranks_seen != world_size:
                raise SystemExit(f"RCCL pre-flight: {ranks_seen} ranks answered, {world_size} expected")
        step = lambda: tiles.step(dt, args.iters)  # noqa: E731
    warmup = OTHER_CONFIGS["config5"][2] if scene_kind == "config5_tiles" and args.warmup == 10 else args.warmup
    for _ in range(warmup):
        step()
    # the state the timed window starts from (mgf_world_clone of every tile): the instrumented replay below steps THE SAME ticks again
    snaps = [w.clone() for w in worlds] if transport == "native" else None
    XKEYS = ("exchange_bytes_out", "exchange_bytes_in", "exchange_bytes_local", "exchange_calls", "host_waits")
    x0 = {k: tiles.counter(k) for k in XKEYS} if transport == "native" else None
    barrier()
    t0 = time.perf_counter()
    ticks = [step() for _ in range(args.steps)]
    barrier()
    elapsed = time.perf_counter() - t0
    # what the neighbour exchanges cost in the timed window, per rank (mgf_tiles_counter; stream time between the events around every
    # exchange, the wait for the neighbouring rank included), and - outside the timed region - how deep bodies rest in each other
    # across tile faces against inside the tiles (the price of block-Jacobi across faces: DESIGN.md, tests/test_gpu_tiles_native.py)
    exchange = seam = None
    if transport == "native":
        mine = [float(tiles.counter(k) - x0[k]) / args.steps for k in XKEYS]
        rows = torch.zeros((world_size, len(XKEYS)), dtype=torch.float64, device=red_dev)
        rows[rank] = torch.tensor(mine, dtype=torch.float64, device=red_dev)
        if dist is not None:
            dist.all_reduce(rows, op=dist.ReduceOp.SUM)
        rows = rows.cpu().numpy()
        exchange = {"per_rank": [{"rank": r, "bytes_out_per_tick": int(rows[r][0]), "bytes_in_per_tick": int(rows[r][1]),
                                  "bytes_between_own_tiles_per_tick": int(rows[r][2]), "exchange_calls_per_tick": round(rows[r][3], 2),
                                  "host_waits_per_tick": round(rows[r][4], 2)} for r in range(world_size)],
                    "note": "bytes cross RANK faces (RCCL send/recv) unless said otherwise; exchange_us_per_tick = stream time between the HIP events around every "
                            "exchange (ghost bodies, ghost velocity refreshes, hand-overs), the wait for the neighbouring rank included - taken, like the "
                            "roofline's launches, in a replay of the timed ticks (option exchange_timing)"}
        def seam_now(worlds_now):
            if whole5 is not None:  # the bodies' sphere parts: centre = x + R(q) (p_sphere - x) of the initial pose
                cb5 = whole5["compound"]
                loc = cb5["comps"]["p"][0::2].astype(np.float64)
                m5 = np.asarray(cb5["comp_mass"], np.float64)
                cap_mid = cb5["comps"]["p"][1::2].astype(np.float64) + 0.5 * cb5["comps"]["d"][1::2].astype(np.float64)
                com = (loc * m5[0::2, None] + cap_mid * m5[1::2, None]) / (m5[0::2, None] + m5[1::2, None])
                loc = loc - com
                mine_x = []
                for k, w in enumerate(worlds_now):
                    st5, tg = w.state(), w.tags().astype(np.int64)
                    q = st5["q"].astype(np.float64)
                    sq, vq, l = q[:, 0:1], q[:, 1:4], loc[tg]
                    rot = l + 2.0 * np.cross(vq, np.cross(vq, l) + sq * l)
                    mine_x.append((first + k, (st5["x"].astype(np.float64) + rot).astype(np.float32)))
            else:
                mine_x = [(first + k, np.asarray(w.state()["x"], dtype=np.float32)) for k, w in enumerate(worlds_now)]
            if dist is not None:
                gathered = [None] * world_size
                dist.all_gather_object(gathered, mine_x)
                all_x = [t for g in gathered for t in g]
            else:
                all_x = mine_x
            return _seam_penetration(all_x) if rank == 0 else None
        try:
            seam = seam_now(worlds)
        except Exception as e:  # (a figure beside the measurement: never at the price of the line)
            seam = {"error": repr(e)}
    def tally(tt):
        units = cons = launches = 0
        kms = 0.0
        for tick in tt:
            for st in tick:
                ghost = int(st["n_ghost_constraints"]) if "n_ghost_constraints" in _keys(st) else 0
                c = int(st["n_constraints"]) - ghost / 2.0  # a constraint across a tile face exists on both tiles: half each
                cons += c
                units += c * args.iters
                launches += int(st["solver_kernel_launches"])
                kms += float(st["ms_solver_kernels"])
        return units, cons, launches, kms
    units, cons, launches, _kms = tally(ticks)
    # the dominant kernel's launches and the exchanges, timed with HIP events (ten events per tile-tick in the timed ticks themselves would
    # cost them 5-10 %): a REPLAY of the timed ticks - a second tile set over clones taken where the timed window started, its own
    # communicator - so that `roofline`, solver_kernel_ms_per_tile_tick and exchange_us_per_tick describe the window `value` was timed on
    replayed = "the timed ticks, replayed from clones of the tile worlds taken where the window starts"
    if transport == "native":
        del step
        del tiles
        rworlds = snaps
        for w in rworlds:
            configure(w)
            w.set_option("time_solver_kernels", 1)
        tiles = mgf_amd.Tiles(ctx, rworlds, [sc["x_range"] for sc in tile_scenes], first_tile=first, n_tiles_total=total_tiles, halo=halo,
                              refresh_every=refresh_every, migrate=not args.no_migrate)
        if world_size > 1:
            uid = torch.zeros(128, dtype=torch.uint8, device=red_dev)
            if rank == 0:
                uid.copy_(torch.frombuffer(bytearray(mgf_amd.rccl_unique_id()), dtype=torch.uint8))
            dist.broadcast(uid, 0)
            tiles.connect(bytes(uid.cpu().numpy().tobytes()), rank, world_size)
        tiles.set_option("exchange_timing", 1)
        ns0 = tiles.counter("exchange_ns")
        step = lambda: tiles.step(dt, args.iters)  # noqa: E731
    else:
        replayed = "the ticks behind the timed region (--transport torch: development driver)"
        tw.world.set_option("time_solver_kernels", 1)
    barrier()
    r_units, _rc, r_launches, kms = tally([step() for _ in range(args.steps)])
    replay_same = bool(abs(r_units - units) < 0.5)
    if transport == "native" and exchange is not None:
        xus = torch.zeros(world_size, dtype=torch.float64, device=red_dev)
        xus[rank] = (tiles.counter("exchange_ns") - ns0) / 1e3 / args.steps
        if dist is not None:
            dist.all_reduce(xus, op=dist.ReduceOp.SUM)
        for r, row in enumerate(exchange["per_rank"]):
            row["exchange_us_per_tick"] = round(float(xus[r].item()), 2)
    step_latency = None
    if transport == "native" and exchange is not None:
        # which step of the protocol costs what (the replay's HIP events around every exchange, by kind): per rank, p50 / p99 / max of a call
        kinds = ("bodies", "velocities", "handover")
        mine_l = [float(tiles.counter(f"exchange_{w}_{k}")) for k in kinds for w in ("calls", "p50_ns", "p99_ns", "max_ns", "mean_ns")]
        lat = torch.zeros((world_size, len(mine_l)), dtype=torch.float64, device=red_dev)
        lat[rank] = torch.tensor(mine_l, dtype=torch.float64, device=red_dev)
        if dist is not None:
            dist.all_reduce(lat, op=dist.ReduceOp.SUM)
        lat = lat.cpu().numpy()
        step_latency = {"unit": "us per exchange call (one grouped ncclSend/ncclRecv - or the device copies between a rank's own tiles - for all of a rank's faces), "
                                "stream time between HIP events, the wait for the neighbouring rank included; the instrumented replay of the timed ticks",
                        "per_rank": [{"rank": r, **{k: {"calls": int(lat[r][5 * i]), "p50": round(lat[r][5 * i + 1] / 1e3, 2), "p99": round(lat[r][5 * i + 2] / 1e3, 2),
                                                       "max": round(lat[r][5 * i + 3] / 1e3, 2), "mean": round(lat[r][5 * i + 4] / 1e3, 2)} for i, k in enumerate(kinds)}}
                                     for r in range(world_size)]}
        exchange["step_latency_us"] = step_latency
    # a second window where the pile has collapsed and come to rest (ticks 400 ..): twice the constraints, hand-overs all the time - the
    # window above is the falling pile (no settled tile window in any bench line)
    settled = None
    if transport == "native" and scene_kind in ("config4", "config5_tiles") and not args.no_settled_tiles:
        tiles.set_option("exchange_timing", 0)
        for w in rworlds:
            w.set_option("time_solver_kernels", 0)
        at = warmup + args.steps
        while at < 400:
            step(); at += 1
        barrier()
        ts0 = time.perf_counter()
        s_ticks = [step() for _ in range(args.steps)]
        barrier()
        s_el = time.perf_counter() - ts0
        s_units, s_cons, _sl, _sk = tally(s_ticks)
        if dist is not None:
            tt = torch.tensor([s_el], dtype=torch.float64, device=red_dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            s_el = float(tt.item())
            uu = torch.tensor([s_units, s_cons], dtype=torch.float64, device=red_dev)
            dist.all_reduce(uu, op=dist.ReduceOp.SUM)
            s_units, s_cons = float(uu[0].item()), float(uu[1].item())
        settled = {"value": s_units / s_el, "unit": "constraint-iters/s", "ms_per_step": s_el * 1e3 / args.steps, "steps": args.steps, "warmup": at,
                   "constraints_per_step": s_cons / args.steps, "tile_tick_ms": s_el * 1e3 / (args.steps * per_rank),
                   "note": "the same tile set stepped on to tick 400: the pile has collapsed and come to rest"}
        try:
            settled["seam_penetration"] = seam_now(rworlds)
        except Exception as e:
            settled["seam_penetration"] = {"error": repr(e)}
    rank_ms = torch.zeros(world_size, dtype=torch.float64, device=red_dev)
    rank_ms[rank] = elapsed * 1e3 / (args.steps * per_rank)  # (a rank's own wall clock between the two barriers, per tile-tick)
    if dist is not None:
        dist.all_reduce(rank_ms, op=dist.ReduceOp.SUM)
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        u = torch.tensor([units, cons], dtype=torch.float64, device=red_dev)
        dist.all_reduce(u, op=dist.ReduceOp.SUM)
        units_all, cons_all = float(u[0].item()), float(u[1].item())
    else:
        units_all, cons_all = float(units), float(cons)
    if rank != 0:
        return None
    n_total = nx * ny * nz * total_tiles
    one_gpu = _same_workload_one_gpu("config4" if scene_kind == "config4" else "config5") if scene_kind in ("config4", "config5_tiles") and world_size > 1 else None
    if one_gpu_here and "value" in one_gpu_here:  # (measured on this box a minute ago: that is the reference; the committed figure rides along)
        one_gpu_here["committed_figure_of_another_box"] = {k: one_gpu[k] for k in ("value", "ms_per_step", "source")} if one_gpu else None
        if one_gpu and one_gpu.get("undivided_world"):
            one_gpu_here["undivided_world"] = one_gpu["undivided_world"]
        one_gpu = one_gpu_here
    elif one_gpu is not None:
        one_gpu["measured_on_this_box"] = False
        one_gpu["why_not"] = (one_gpu_here or {}).get("error") or ("--no-one-gpu-reference" if args.no_one_gpu_reference else "stand-in transport / development driver")
    eff = None
    if one_gpu:
        v = units_all / elapsed
        eff = {"vs_8_tiles_on_one_gpu": round(v / (world_size * one_gpu["value"]), 4),
               "vs_undivided_world_on_one_gpu": round(v / (world_size * one_gpu["undivided_world"]["value"]), 4) if one_gpu.get("undivided_world") else None,
               "definition": "(this line's value / N) / (the same scene's value on ONE GPU): 1.0 = perfect strong scaling; the one-GPU figures are committed "
                             "measurements (same_workload_on_one_gpu.source), the 8-tile one through the same tile protocol"
                             + (" - the 8-tile figure MEASURED ON THIS BOX by rank 0 before the ranks connected" if one_gpu.get("measured_on_this_box") else "")}
    if scene_kind == "config5_tiles":
        name = (f"BASELINE config 5 (this build's own definition of a rigid body of two components - the reference has none, SURVEY H8): {n_total} bodies, each a "
                f"sphere (r = 0.5) + a capsule (|d| = 1, r = 0.3), {total_tiles * nx}x{ny}x{nz} lattice of pitch 2.2 in one open box, cut into {total_tiles} x-slab "
                f"tiles of {nx} lattice columns; ticks {warmup}..{warmup + args.steps}")
    else:
      name = (f"BASELINE config 4: {n_total} spheres ({total_tiles * nx}x{ny}x{nz} jittered lattice pile, r=0.5, seed 0x6D6766) in one open box, cut into "
            f"{total_tiles} x-slab tiles of {nx} lattice columns" if scene_kind == "config4" else
            f"{n_total} spheres: {total_tiles} x-slab tiles of {nx}x{ny}x{nz} side by side in one open box (weak scaling of BASELINE config 2's tile)")
    return {
        "metric": "contact_constraint_iters_per_sec", "value": units_all / elapsed, "unit": "constraint-iters/s", "n_gpus": world_size,
        "steps": args.steps, "warmup": warmup, "ms_per_step": elapsed * 1e3 / args.steps, "higher_is_better": True,
        "scaling": "strong" if scene_kind in ("config4", "config5_tiles") else "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic" if not args.standin else "synthetic; NOT A MEASUREMENT OF xGMI: the ranks exchanged over a stand-in transport bound in place of librccl "
                                                     f"({args.standin}), sharing GPUs as MGF_BENCH_DEVICE says - the multi-rank flow on one box, nothing else",
        "config": {"workload": f"{name}, dt=1/60, {args.iters} solver iters; {per_rank} tile(s) per GPU; ghost bodies once per tick, ghost velocities every "
                               f"{refresh_every} solver iterations, bodies handed to the tile that holds their centre"
                               + (" - DISABLED" if args.no_migrate else "") + f"; {SCENE_NOTE}",
                   "bodies_total": n_total, "tiles_total": total_tiles, "tiles_per_gpu": per_rank, "iters": args.iters, "dt": dt,
                   "constraint_order": "canonical inside a tile (i asc; terrain DFS; partners j<i asc); block-Jacobi across tile faces",
                   "parallelism": f"{world_size} GPU(s) x {per_rank} tile(s); exchange between a rank's tiles by device copies"
                                  + ("" if world_size == 1 else (", between ranks by RCCL send/recv over xGMI under the C-ABI (mgf_tiles_*)" if transport == "native"
                                                                 else f", between ranks by torch.distributed ({args.backend})"))},
        "transport": transport, "rccl_ranks_seen": ranks_seen, "rccl_lib": args.standin or ("librccl (dlopen under the C-ABI)" if world_size > 1 else None),
        "physics_steps_per_sec": args.steps / elapsed, "constraints_per_step": cons_all / args.steps,
        "solver_launches_per_step_rank0": launches / args.steps,
        # (a rank's tiles share one stream and their phases are enqueued interleaved: per-phase event spans of one tile include
        # the other tiles' work, so only the solver kernels' own event time is reported)
        "solver_kernel_ms_per_tile_tick_rank0": kms / (args.steps * per_rank),  # (of the instrumented replay of the timed ticks)
        "tile_tick_ms_rank0": elapsed * 1e3 / (args.steps * per_rank),
        "tile_tick_ms_per_rank": [round(float(v), 4) for v in rank_ms.cpu().numpy()],
        "roofline": _roofline(r_units, r_launches, kms, mode, f"{refresh_every} iteration(s) of one tile between ghost refreshes; a replay of the {args.steps} "
                              "timed ticks, HIP events around every launch", ("tiles", warmup, args.steps), workload=("config4" if scene_kind != "config5_tiles" else "config5") + "_tiles"),
        "instrumentation": "the timed ticks carry no HIP events; roofline, solver_kernel_ms_per_tile_tick and exchange_us_per_tick come from " + replayed
                           + " with the events on",
        "replay_did_the_timed_windows_work_rank0": replay_same,
        "same_workload_on_one_gpu": one_gpu, "efficiency_vs_same_workload_on_one_gpu": eff,
        "prediction": _tile_prediction(scene_kind, world_size, total_tiles, elapsed * 1e3 / args.steps, one_gpu, refresh_every, args.iters),
        "exchange": exchange, "seam_penetration": seam, "settled": settled,
    }


def _seam_penetration(tiles_x, radius=0.5):
    """Resting depth of touching spheres at the end of the run: pairs whose bodies live on DIFFERENT tiles (their constraint is solved
    block-Jacobi across the face, ghost velocities refreshed every R iterations) against pairs inside one tile (exact Gauss-Seidel)."""
    from scipy.spatial import cKDTree
    x = np.concatenate([t[1] for t in tiles_x]).astype(np.float64)
    tile = np.concatenate([np.full(len(t[1]), t[0], np.int32) for t in tiles_x])
    pairs = cKDTree(x).query_pairs(2.0 * radius, output_type="ndarray")
    if len(pairs) == 0:
        return {"pairs_touching": 0}
    depth = 2.0 * radius - np.linalg.norm(x[pairs[:, 0]] - x[pairs[:, 1]], axis=1)
    across = tile[pairs[:, 0]] != tile[pairs[:, 1]]

    def stat(d):
        return {"pairs": int(len(d)), "mean": float(d.mean()) if len(d) else None, "p99": float(np.percentile(d, 99)) if len(d) else None, "max": float(d.max()) if len(d) else None}
    return {"unit": "sphere radii x 2 (depth = 2r - distance of centres, r = 0.5)", "across_tile_faces": stat(depth[across]), "inside_tiles": stat(depth[~across]),
            "note": "state at the end of the timed window; a pile that has not come to rest yet shows little depth on either side"}


def _one_gpu_reference_here(args, scene_kind):
    """python bench.py --gpus 1 --scene <the same> in a process of its own on this rank's device (the environment of the launcher's ranks taken out)."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "ROLE_RANK", "LOCAL_WORLD_SIZE",
                                                           "ROLE_WORLD_SIZE", "TORCHELASTIC_RUN_ID", "MGF_BENCH_DEVICE") and not k.startswith("TORCHELASTIC")}
    env["MGF_BENCH_DEVICE"] = os.environ.get("MGF_BENCH_DEVICE", os.environ.get("LOCAL_RANK", "0"))
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--scene", scene_kind, "--no-cpu-baseline", "--no-settled-tiles", "--steps", str(args.steps),
           "--warmup", str(args.warmup), "--iters", str(args.iters)]
    if args.refresh_every is not None:
        cmd += ["--refresh-every", str(args.refresh_every)]
    for o in args.opt:
        cmd += ["--opt", o]
    t0 = time.perf_counter()
    try:
        out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
        d = json.loads(line)
        return {"value": d["value"], "ms_per_step": d["ms_per_step"], "steps": d["steps"], "warmup": d["warmup"], "tile_tick_ms": d.get("tile_tick_ms_rank0"),
                "measured_on_this_box": True, "took_s": round(time.perf_counter() - t0, 1),
                "source": "measured by rank 0 on its own device before the ranks connected: " + " ".join(cmd[1:])}
    except Exception as e:  # (a reference beside the measurement: never at the price of the line)
        return {"error": repr(e)[:300]}


# The N = 2 / 4 / 8 prediction of DESIGN.md section 8 for BASELINE config 4 (and 5) as 8 x-slab tiles, so that the first run on a real node is
# judged against numbers written down BEFORE it: per tick = (tiles per rank) x (one tile's tick on one GPU) + the exchange steps of the
# protocol, each a grouped ncclSend/ncclRecv at an ASSUMED point-to-point latency, + the rows across a rank face at an ASSUMED link rate.
TILE_PREDICTION_ASSUMPTIONS = {
    "p2p_latency_us": 20.0,        # one grouped ncclSend/ncclRecv of a few KB..MB between neighbouring GPUs, launch to completion on the stream (RCCL 2.26 over xGMI; never measured here)
    "link_GBps": 60.0,             # achieved one-way rate of one xGMI link for MB-sized rows (153.6 GB/s bidirectional per link on paper)
    "allreduce_us": 30.0,          # the tick's status agreement (4-byte all-reduce over N ranks)
    "exchange_steps_per_tick": {"ghost_bodies": 1, "ghost_velocities": "ceil(iters / R) - 1 (R = 4, 10 iterations: 2)", "handover": 0.4, "status_allreduce": 1},
    "bytes_per_rank_face_per_tick": {"config4": 3.4e6, "config5_tiles": 0.5e6},  # one direction, AT R = 2 WITH ROUND 5's RECORDS: ghost records (288 B) + 4 velocity refreshes (32 B each) per ghost (profiles/r05_config4_8tiles_1gpu_bench.json: 95 MB between the 14 face directions of 8 tiles); scaled below to the run's R and to round 6's records
    "ghost_record_bytes": {"config4": 160, "config5_tiles": 288},  # r06: a world of single-component bodies sends the first 40 floats of the 72
    "velocity_record_bytes": 24,                                    # r06: 6 floats (the two zeros stay at home)
}


def _tile_prediction(scene_kind, world_size, total_tiles, measured_ms, one_gpu, refresh_every=4, iters=10):
    if scene_kind not in ("config4", "config5_tiles") or not one_gpu or "ms_per_step" not in one_gpu:
        return None
    A = TILE_PREDICTION_ASSUMPTIONS
    tile_tick = one_gpu["ms_per_step"] / total_tiles  # one tile's tick where 8 share a GPU: kernels, back to back
    per = total_tiles // world_size
    steps = A["exchange_steps_per_tick"]
    refreshes = max(0, -(-int(iters) // max(1, int(refresh_every))) - 1)  # a velocity exchange between consecutive solver launches
    lat_us = (steps["ghost_bodies"] + refreshes + steps["handover"]) * A["p2p_latency_us"] + steps["status_allreduce"] * A["allreduce_us"]
    wire_us = A["bytes_per_rank_face_per_tick"][scene_kind] * (A["ghost_record_bytes"][scene_kind] + A["velocity_record_bytes"] * refreshes) / (288.0 + 32.0 * 4) / (A["link_GBps"] * 1e3)
    comm_ms = 0.0 if world_size == 1 else (lat_us + wire_us) / 1e3
    pred_ms = per * tile_tick + comm_ms
    return {"predicted_ms_per_step": round(pred_ms, 4), "predicted_efficiency_vs_8_tiles_on_one_gpu": round(one_gpu["ms_per_step"] / (world_size * pred_ms), 4),
            "measured_ms_per_step": round(measured_ms, 4), "measured_over_predicted": round(measured_ms / pred_ms, 3),
            "model": "tiles_per_rank x tile_tick_ms + [exchange steps x p2p latency + status all-reduce + bytes per rank face / link rate]; exchanges are on the critical "
                     "path (every phase waits for its rows); DESIGN.md section 8",
            "tile_tick_ms": round(tile_tick, 4), "tiles_per_rank": per, "comm_ms_per_tick": round(comm_ms, 4), "assumptions": A}


def _same_workload_one_gpu(cfg):
    """The N = 1 bench line is BASELINE config 2 (the contract's single-GPU workload); the N > 1 lines are strong-scaling slices of
    config 4 (or 5).  For a scaling figure against the SAME workload: that config's 8 tiles on ONE GPU, as measured and committed."""
    import glob

    def newest(pattern):  # (profiles are named per round: the latest committed one)
        found = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
        return found[-1] if found else None
    p = newest(f"r*_{cfg}_8tiles_1gpu_bench.json")
    try:
        d = json.load(open(p))
        out = {"value": d["value"], "ms_per_step": d["ms_per_step"], "steps": d["steps"], "warmup": d["warmup"],
               "source": f"profiles/{os.path.basename(p)} (python bench.py --gpus 1 --scene {cfg}{'_tiles' if cfg == 'config5' else ''}; a committed measurement, not taken in this run)"}
    except (OSError, KeyError, ValueError, TypeError):
        return None
    try:  # and the same scene as ONE world (exact canonical order, no seams; tools/config4_undivided.py)
        pu = newest(f"r*_{cfg}_undivided_1gpu.json")
        u = json.load(open(pu))
        out["undivided_world"] = {"value": u["value"], "ms_per_step": u["ms_per_step"], "steps": u["steps"], "warmup": u["warmup"],
                                  "source": f"profiles/{os.path.basename(pu)} (tools/config4_undivided.py)"}
    except (OSError, KeyError, ValueError, TypeError):
        pass
    return out


def _keys(st):
    return [f[0] for f in st._fields_] if hasattr(st, "_fields_") else list(st.keys())


_LIB_SHA = []


def _lib_sha256():
    """sha256 of the libmgf_hip.so this process loaded"""
    if not _LIB_SHA:
        import hashlib
        from mgf_amd import _capi
        try:
            _LIB_SHA.append(hashlib.sha256(open(_capi.lib_path(), "rb").read()).hexdigest())
        except OSError:
            _LIB_SHA.append(None)
    return _LIB_SHA[0]


def _pmc_traffic(mode, window, workload="config2"):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (profiles/pmc_*k_solve_flow*.json: one
    record, or a list of records - one per window the passes were taken on), only for the window this run timed AND only when the
    passes ran the library this run loaded (the record's lib_sha256); else (None, why)."""
    prefix = "pmc_" if workload == "config2" else f"pmc_{workload}_"
    base = os.path.join(ROOT, "profiles", f"{prefix}k_solve_flow{mode}")
    recs, found = [], False
    # (the settled pile is the same scene further on: `--warmup 400` as a run of its own and the default run's nested settled window
    # are the same ticks - a window is the ticks it skips and the ticks it times)
    for p in ((base + "_settled.json", base + ".json") if window and window[0] == "settled" else (base + ".json", base + "_settled.json")):
        if not os.path.exists(p):
            continue
        found = True
        try:
            d = json.load(open(p))
        except Exception:
            continue
        for r in (d if isinstance(d, list) else [d]):
            recs.append(r.get("window"))
            same_scene = r.get("window") and {r["window"][0], window[0]} <= {"transient", "settled"}
            if r.get("window") and (list(r["window"]) == list(window) or (same_scene and list(r["window"][1:]) == list(window[1:]))):
                if r.get("lib_sha256") != _lib_sha256():
                    return None, (f"{os.path.relpath(p, ROOT)} covers this window, but its passes ran another build of libmgf_hip.so "
                                  f"(sha256 {str(r.get('lib_sha256'))[:12]}... there, {str(_lib_sha256())[:12]}... here): no traffic figure for this build")
                return r.get("hbm_bytes_per_launch"), (f"{os.path.relpath(p, ROOT)} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, same window, "
                                                       f"same libmgf_hip.so: sha256 {_lib_sha256()[:12]}...)")
    if not found:
        return None, "no PMC pass committed for this kernel"
    return None, f"the committed PMC passes cover windows {recs}, this run {list(window)}"


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def run_config3_contact_rich(scenes, O, iters, gpu_ticks=150, cpu_ticks=3, budget_s=14.0):
    """BASELINE config 3 on one host core from a contact-rich state: the capsules need a couple of hundred ticks to arrive on the
    heightfield and on each other, which the CPU path cannot afford here - so the GPU carries the scene to tick `gpu_ticks` and the
    oracle is started from that state (x, q, v, omega, delta: everything World::step reads) for `cpu_ticks` ticks."""
    import mgf_amd
    scene = scenes.capsule_field(128, 32, 32, quads=158)
    dt = float(scene["dt"])
    ctx = mgf_amd.Context(0)
    g = mgf_amd.World.from_scene(ctx, scene)
    g.step_many(dt, iters, gpu_ticks)
    st = g.state()
    del g
    ctx.close()
    w = O.World(O.ORDER_DEMO)
    t = scene["terrain"]
    w.set_terrain(t["verts"], t["faces"], t["pos"])
    w.add_bodies(scene["comps"], scene["mass"], scene["restitution"], scene["friction"], scene["force"])
    w.set_state(x=st["x"], q=st["q"], v=st["v"], omega=st["omega"], delta=st["delta"])
    units, solve_s, done, cons = 0, 0.0, 0, 0
    t0 = time.perf_counter()
    for _ in range(cpu_ticks):
        s1 = w.step(dt, iters)
        units += s1.n_constraints * iters
        cons = s1.n_constraints
        solve_s += s1.t_solve
        done += 1
        if time.perf_counter() - t0 > budget_s:
            break
    el = time.perf_counter() - t0
    return {"constraint_iters_per_sec": units / el if el > 0 else None, "physics_steps_per_sec": done / el, "steps": done, "seconds": round(el, 2),
            "solve_phase_constraint_iters_per_sec": units / solve_s if solve_s > 0 else None, "constraints_last_tick": int(cons),
            "sample": f"ticks {gpu_ticks}..{gpu_ticks + done} of config 3, started from the GPU path's state at tick {gpu_ticks} (contact-rich), world.rs order"}


def cpu_baseline(iters):
    """The reference is single-threaded Rust that cannot be built here; its CPU path is timed as the oracle's C++ restatement
    ("port") on 1 host core, in the reference's own (world.rs) constraint order, on bounded samples of BASELINE configs 1-3."""
    from mgf_amd import scenes
    from oracle import oracle as O

    def run(scene, steps, budget_s):
        w = O.World(O.ORDER_DEMO)
        t = scene["terrain"]
        w.set_terrain(t["verts"], t["faces"], t["pos"])
        w.add_bodies(scene["comps"], scene["mass"], scene["restitution"], scene["friction"], scene["force"])
        if scene.get("v0") is not None:
            w.set_state(v=scene["v0"])
        units, solve_s, done = 0, 0.0, 0
        t0 = time.perf_counter()
        for _ in range(steps):
            st = w.step(float(scene["dt"]), iters)
            units += st.n_constraints * iters
            solve_s += st.t_solve
            done += 1
            if time.perf_counter() - t0 > budget_s:
                break
        el = time.perf_counter() - t0
        return {"constraint_iters_per_sec": units / el if el > 0 else None, "physics_steps_per_sec": done / el, "steps": done, "seconds": round(el, 2),
                "solve_phase_constraint_iters_per_sec": units / solve_s if solve_s > 0 else None}

    c2 = run(scenes.sphere_pile(64, 64, 64), 3, 12.0)
    c1 = run(scenes.balls_demo(8), 400, 6.0)
    c3 = run_config3_contact_rich(scenes, O, iters)
    return {"value": c2["constraint_iters_per_sec"], "unit": "constraint-iters/s", "cores": 1, "kind": "port",
            "sample": f"first {c2['steps']} ticks of config 2 (262144 spheres), world.rs order, {c2['seconds']} s of CPU work",
            "physics_steps_per_sec": c2["physics_steps_per_sec"], "solve_phase_constraint_iters_per_sec": c2["solve_phase_constraint_iters_per_sec"],
            "cpu_model": _cpu_model(), "nproc": os.cpu_count(),
            "config1_balls_512": c1, "config3_capsules_131072": c3,
            "config3_note": "capsule_field(128, 32, 32): pitch 2.6 over a 158 x 158-quad heightfield, as SURVEY 8d names it",
            "note": "C++ restatement of mgf (g++ -O2 -ffp-contract=off), not rustc output; the reference is single-threaded"}


if __name__ == "__main__":
    main()
