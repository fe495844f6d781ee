"""The tile protocol across GPUs: one process per device, ncclSend / ncclRecv over xGMI under the
C-ABI.  These tests run when the box has at least two devices (the development box has one: they skip there, and the single-device
half of the status agreement is tested below on any box) - so the first lease of a multi-GPU node proves the path by itself:
a pile drifting through the tiles, ghost exchange, velocity refreshes and hand-overs, bit-identical to the oracle's tiles; and a
rank that fails mid-tick takes the whole tick down on every rank instead of leaving its neighbours in a receive."""
import multiprocessing as mp

import numpy as np
import pytest

import mgf_amd
from mgf_amd import scenes
from tests.util import values_equal

pytestmark = pytest.mark.gpu
STATE_KEYS = ("x", "q", "v", "omega", "delta")


def _device_count():
    import torch
    return torch.cuda.device_count()


def _launch(n_ranks, total_tiles, dims, drift, ticks, fail_rank=-1, fail_tick=-1, timeout=600, shared_device=False, world_opts=None, real_lib=False, extra_env=None,
            allow_crash=False):
    """shared_device: every rank on device 0, the C-ABI bound to the stand-in transport of tests/fake_rccl (RCCL refuses two ranks on
    one device) - every process still runs the whole of mgf_tiles_step's multi-rank path."""
    from tests.mgpu_worker import run_rank
    lib = None
    if shared_device and not real_lib:
        from tests.fake_rccl.build import build
        lib = build()
    mpc = mp.get_context("spawn")
    uid_q, out_q = mpc.Queue(), mpc.Queue()
    procs = [mpc.Process(target=run_rank, args=(r, n_ranks, total_tiles, dims, drift, ticks, fail_rank, fail_tick, uid_q, out_q, 0 if shared_device else None, lib, world_opts, extra_env))
             for r in range(n_ranks)]
    for p in procs:
        p.start()
    results = {}
    try:
        for _ in range(n_ranks):
            res = out_q.get(timeout=timeout)   # a rank stuck in a receive shows up here as a timeout, not as a hung test run
            results[res["rank"]] = res
    finally:
        for p in procs:
            p.join(timeout=30)
            if p.is_alive():
                p.kill()
    for r in range(n_ranks):
        assert allow_crash or "crash" not in results[r], results[r].get("crash")
    return results


def _check_against_oracle_tiles(res, n_ranks, P, dims, drift, ticks, expect_moves=True):
    from mgf_amd.tiles import Tile, step_tiles_inprocess
    from tests.oracle_engine import OracleEngine
    if dims == "two_kinds":
        from tests.util import two_kinds_tile_scenes
        tile_scenes, kw = two_kinds_tile_scenes(P), dict(halo=2.0)
    else:
        tile_scenes, kw = [scenes.sphere_pile_tile(*dims, r, P, drift=drift) for r in range(P)], {}
    ot = [Tile(OracleEngine(sc), sc["x_range"], r, P, sc["dt"], sc["iters"], **kw) for r, sc in enumerate(tile_scenes)]
    for _ in range(ticks):
        step_tiles_inprocess(ot)
    got = {}
    out_b = in_b = 0
    for r in range(n_ranks):
        assert res[r]["ranks_seen"] == n_ranks and res[r]["failed_at"] is None, res[r]
        for t in res[r]["tiles"]:
            got[t["tile"]] = t
        out_b += res[r]["bytes_out"]; in_b += res[r]["bytes_in"]
        assert res[r]["bytes_out"] > 0 and res[r]["bytes_in"] > 0   # every rank has a neighbouring rank
    assert out_b == in_b   # what left one rank arrived on another
    assert sorted(got) == list(range(P))
    moved = 0
    for k in range(P):
        assert np.array_equal(got[k]["tags"], ot[k].e.tags()), f"tile {k}: body order"
        so = ot[k].e.state()
        for f in STATE_KEYS:
            assert values_equal(got[k][f], so[f]), f"tile {k}: {f}"
        assert got[k]["migrated_in"] == ot[k].n_migrated_in
        moved += got[k]["migrated_in"]
    assert moved > 0 or not expect_moves  # bodies did change owner across ranks


@pytest.mark.parametrize("n_ranks", [2, 4, 8])
def test_ranks_on_distinct_devices_match_the_oracle_tiles(n_ranks):
    if _device_count() < n_ranks:
        pytest.skip(f"needs {n_ranks} devices")
    P, dims, drift, ticks = 8, (4, 4, 5), (5.0, 0.0, 0.0), 40
    res = _launch(n_ranks, P, dims, drift, ticks)
    _check_against_oracle_tiles(res, n_ranks, P, dims, drift, ticks)


def test_real_librccl_with_two_ranks_on_one_device():
    """the multi-rank path has only ever met the REAL librccl with one rank (test_gpu_tiles_native.py); every run of two
    and more ranks on the one-GPU development box went over the stand-in transport.  This test gives the real library two ranks - separate
    processes - on the box's one device.  NCCL-family libraries refuse a communicator with the same device twice ("Duplicate GPU
    detected"); if this RCCL does, the refusal - its own words, from its debug log - is the skip reason and is written to
    gpurun_out/rccl_two_ranks_one_device.txt (committed as profiles/r06_rccl_two_ranks_one_device.txt), so that why the grouped
    ncclSend / ncclRecv of mgf_tiles_step cannot be exercised on one GPU is on file; if it accepts, the run must match the oracle's tiles."""
    import glob
    import os
    import tempfile
    tmp = tempfile.mkdtemp(prefix="mgf_rccl_")
    env = {"NCCL_DEBUG": "WARN", "NCCL_DEBUG_FILE": os.path.join(tmp, "rccl_%h_%p.log")}
    P, dims, drift, ticks = 4, (4, 4, 5), (5.0, 0.0, 0.0), 40
    res = _launch(2, P, dims, drift, ticks, shared_device=True, real_lib=True, extra_env=env, allow_crash=True, timeout=180)
    crashed = [res[r]["crash"] for r in res if "crash" in res[r]]
    if not crashed:
        _check_against_oracle_tiles(res, 2, P, dims, drift, ticks)
        return
    log = "".join(open(f).read() for f in sorted(glob.glob(os.path.join(tmp, "rccl_*.log"))))
    said = [ln.strip() for ln in log.splitlines() if "Duplicate GPU" in ln] or [ln.strip() for ln in log.splitlines() if "WARN" in ln]
    last = [c.strip().splitlines()[-1] for c in crashed]
    reason = (f"librccl refuses two ranks on one device: {said[0] if said else '(no WARN line in its debug log)'} | C-ABI: {last[0]}")
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "rccl_two_ranks_one_device.txt"), "w") as f:
        f.write("tests/test_gpu_multi_device.py::test_real_librccl_with_two_ranks_on_one_device\n" + reason + "\n\n-- every rank's last line --\n" + "\n".join(last)
                + "\n\n-- librccl's debug log (NCCL_DEBUG=WARN), the lines about the communicator --\n" + "\n".join(said) + "\n")
    assert any("Duplicate GPU" in ln for ln in said) or any("mgf status" in c for c in last), (said, last)   # (a refusal, not some other failure)
    pytest.skip(reason)


@pytest.mark.parametrize("P", [2, 4])
def test_ghost_records_of_two_widths_across_a_rank_face(P):
    """r06: a rank whose worlds hold no body of several components sends 40-float ghost records, its neighbour 72 - the receiver sizes its
    ncclRecv from the kinds word that travels with the counts.  Two ranks on one device over the stand-in transport (which fails a send met by
    a receive of another size): two-part bodies on rank 0, plain spheres on rank 1, driven into each other - the oracle's tiles, bit for bit."""
    res = _launch(2, P, "two_kinds", None, 90, shared_device=True)
    _check_against_oracle_tiles(res, 2, P, "two_kinds", None, 90)


# ---- several ranks on ONE device (any box): the whole multi-rank path of mgf_tiles_step, processes and all, over the stand-in transport
@pytest.mark.parametrize("n_ranks,P", [(2, 4), (2, 8), (4, 8)])
def test_ranks_sharing_one_device_match_the_oracle_tiles(n_ranks, P):
    """What the RCCL runs above would show on a multi-GPU box, minus the fabric: every send of the protocol is met by a receive of the
    same size in the same order on the neighbouring rank (the stand-in fails a mismatch and reports a deadlock instead of hanging),
    counts, ghosts, velocity refreshes and hand-overs cross RANK faces, and the result is the oracle tiles' bit for bit."""
    dims, drift, ticks = (4, 4, 5), (5.0, 0.0, 0.0), 40
    res = _launch(n_ranks, P, dims, drift, ticks, shared_device=True)
    _check_against_oracle_tiles(res, n_ranks, P, dims, drift, ticks)


def test_two_processes_sharing_one_device_unannounced_still_finish_every_tick():
    """... and when nobody told them: each process launches 256 workgroups as if the device were its own.  Whenever the two launches take
    part of the CUs each and wait for the rest, they give up after ~0.5 s; the ranks agree on it, put their bodies back and repeat the tick
    with the launch-per-frontier executor (mgf_tiles_step, option retry_lost_ticks) - slower, never an error, the same bits."""
    P, dims, drift, ticks = 2, (16, 128, 64), None, 4
    res = _launch(2, P, dims, drift, ticks, shared_device=True, timeout=900)
    for r in range(2):
        assert res[r]["failed_at"] is None, res[r]["error"]
    _check_against_oracle_tiles(res, 2, P, dims, drift, ticks, expect_moves=False)


def test_two_processes_sharing_one_device_at_config4_tile_size():
    """two processes on device 0, a 131 072-sphere tile of BASELINE config 4's shape each.  A persistent solver launch
    needs all its workgroups resident at once; two such launches of 256 workgroups from two processes can take half the CUs each and
    wait for the rest forever (they give up after ~0.5 s, and a tile set then reports the tick as lost).  Told that they share the
    device (option flow_max_blocks = half the CUs: 128 workgroups of 1024 bodies), both processes finish every tick, through the
    block-local solver, bit-identical to the oracle's tiles."""
    P, dims, drift, ticks = 2, (16, 128, 64), None, 6
    res = _launch(2, P, dims, drift, ticks, shared_device=True, world_opts={"flow_max_blocks": 128}, timeout=900)
    for r in range(2):
        assert res[r]["failed_at"] is None, res[r]["error"]
        assert res[r]["flow_blocks"] == [128] and res[r]["flow6_runs"] >= ticks - 1, {k: v for k, v in res[r].items() if k != "tiles"}
    _check_against_oracle_tiles(res, 2, P, dims, drift, ticks, expect_moves=False)


def test_a_failing_rank_takes_the_tick_down_on_every_rank_sharing_one_device():
    n_ranks = 3
    res = _launch(n_ranks, 6, (4, 4, 5), (5.0, 0.0, 0.0), 12, fail_rank=n_ranks - 1, fail_tick=5, timeout=300, shared_device=True)
    for r in range(n_ranks):
        assert res[r]["failed_at"] == 5, f"rank {r}: {res[r]}"   # nobody hangs, nobody goes on alone
    assert "halo" in res[n_ranks - 1]["error"]
    assert all("rank failed" in res[r]["error"] for r in range(n_ranks - 1))


def test_a_failing_rank_takes_the_tick_down_on_every_rank():
    if _device_count() < 2:
        pytest.skip("needs 2 devices")
    n_ranks = 4 if _device_count() >= 4 else 2
    res = _launch(n_ranks, 8 if n_ranks == 4 else 4, (4, 4, 5), (5.0, 0.0, 0.0), 12, fail_rank=n_ranks - 1, fail_tick=5, timeout=300)
    for r in range(n_ranks):
        assert res[r]["failed_at"] == 5, f"rank {r}: {res[r]}"   # nobody hangs, nobody goes on alone
    assert "halo" in res[n_ranks - 1]["error"]
    assert all("rank failed" in res[r]["error"] for r in range(n_ranks - 1))


# ---- the same agreement inside one process (any box) -------------------------------------------------------------------------
@pytest.fixture(scope="module")
def ctx():
    c = mgf_amd.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("connected", [False, True])
def test_a_failed_tick_is_reported_at_its_end_and_moves_nobody(ctx, connected):
    """Three tiles in one process; the set fails in the collide phase of tick 4 (mgf_tiles_set_option "test_fail_tick"): the call
    returns the failure after the tick's communication skeleton has run to its end - with a one-rank RCCL communicator too, whose
    status all-reduce then runs - and no body has changed owner in that tick."""
    P = 3
    tile_scenes = [scenes.sphere_pile_tile(4, 4, 5, r, P, drift=(5.0, 0.0, 0.0)) for r in range(P)]
    worlds = []
    for sc in tile_scenes:
        w = mgf_amd.World.from_scene(ctx, sc)
        w.set_tags(sc["tags"])
        worlds.append(w)
    T = mgf_amd.Tiles(ctx, worlds, [sc["x_range"] for sc in tile_scenes])
    if connected:
        T.connect(mgf_amd.rccl_unique_id(), 0, 1)
        assert T.preflight() == 1
    T.set_option("test_fail_tick", 4)
    T.set_option("exchange_timing", 1)
    dt, iters = float(tile_scenes[0]["dt"]), tile_scenes[0]["iters"]
    for _ in range(4):
        T.step(dt, iters)
    owned = [len(w) for w in worlds]
    moved = [T.migrated(r) for r in range(P)]
    with pytest.raises(mgf_amd.MgfError) as e:
        T.step(dt, iters)
    assert "halo" in str(e.value)
    assert [len(w) for w in worlds] == owned and [T.migrated(r) for r in range(P)] == moved
    with pytest.raises(mgf_amd.MgfError):
        T.set_option("no_such_option", 1)
    # the exchanges' counters: rows moved between this process's tiles only, five ticks, the events' time is there
    from mgf_amd.tiles import DEFAULT_REFRESH_EVERY as R
    per_tick = 1 + (-(-iters // R) - 1)  # the ghost bodies, then a velocity exchange between consecutive solver launches (R = 4: 4 + 4 + 2 iterations)
    assert T.counter("ticks") == 5 and T.counter("exchange_calls") >= 5 * per_tick and T.counter("exchange_ns") > 0
    assert T.counter("exchange_bytes_local") > 0 and T.counter("exchange_bytes_out") == 0 and T.counter("exchange_bytes_in") == 0
    # (r05: one wait per phase whatever the number of tiles - counts, read-backs, solver flags - plus the status agreement between ranks)
    assert 5 * 3 <= T.counter("host_waits") <= 5 * 5
    with pytest.raises(KeyError):
        T.counter("no_such_counter")


# ---- `python bench.py --gpus N` as the driver issues it: no torchrun around it, the command spawns its own ranks ----------------
def test_bench_gpus_2_spawns_its_own_ranks_and_prints_one_line():
    """the scaling bench is launched like the N = 1 bench (plain `python bench.py --gpus N ...`); on a one-GPU box the
    two ranks share device 0 over the stand-in transport, and the line says that it is no fabric measurement.  The roofline comes from a
    replay of the timed ticks (clones of the tile worlds), which must have done the timed window's work."""
    import json
    import os
    import subprocess
    import sys
    from tests.fake_rccl.build import build
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--scene", "config5", "--steps", "4", "--warmup", "12"]
    if _device_count() < 2:
        cmd += ["--rccl-lib", build()]
    p = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["value"] > 0 and d["scaling"] == "strong"
    assert d["rccl_ranks_seen"] == 2 and len(d["tile_tick_ms_per_rank"]) == 2
    assert d["replay_did_the_timed_windows_work_rank0"] is True
    assert d["roofline"] and d["roofline"]["avg_launch_us"] > 0 and 0 < d["roofline"]["launches_timed"] <= 4 * 4 * 5  # 4 ticks x 4 tiles x 5 launches (a tile without constraints launches nothing)
    if _device_count() < 2:
        assert "NOT A MEASUREMENT" in d["data"] and d["rccl_lib"].endswith(".so")
    else:
        assert d["data"] == "synthetic"
    # r06: what makes the first real run self-diagnosing - a settled window with its seam, per-kind latency of the exchange calls for every rank,
    # the one-GPU reference (measured on this box on a real node; the committed figure, saying so, over the stand-in), the written-down prediction
    assert d["settled"]["ms_per_step"] > 0 and d["settled"]["warmup"] >= 400 and "seam_penetration" in d["settled"]
    lat = d["exchange"]["step_latency_us"]["per_rank"]
    # (4 timed ticks: the ghost bodies once per tick, a velocity exchange between consecutive solver launches - R = 4: 4 + 4 + 2 iterations, two exchanges)
    assert len(lat) == 2 and all(r["bodies"]["calls"] == 4 and r["velocities"]["calls"] == 8 and r["velocities"]["p99"] >= r["velocities"]["p50"] > 0 for r in lat)
    one = d["same_workload_on_one_gpu"]
    assert one and ("measured_on_this_box" in one) and (one["measured_on_this_box"] or one.get("why_not"))
    pr = d["prediction"]
    assert pr and pr["tiles_per_rank"] == 4 and pr["predicted_ms_per_step"] > 0 and pr["assumptions"]["p2p_latency_us"] > 0 and pr["measured_over_predicted"] > 0


@pytest.mark.timeout(900)
def test_two_ranks_through_the_piles_collapse_equal_one_process():
    """BASELINE config 4 at full size for 170 ticks - through the collapse of the million-sphere pile, the state no bench window reaches
    (tools/soak_tiles.py found a device fault there): 2 ranks x 4 tiles on one device over the stand-in transport against the same 8 tiles
    in one process.  Thousands of bodies change owner per tick, many of them across the RANK face; the result must be the same bit for bit."""
    import mgf_amd
    P, dims, ticks = 8, (16, 128, 64), 170
    res = _launch(2, P, dims, None, ticks, shared_device=True, timeout=800, world_opts={"flow_max_blocks": 128})
    got = {}
    for r in (0, 1):
        assert res[r]["failed_at"] is None and res[r]["ticks_retried"] == 0, res[r]
        for t in res[r]["tiles"]:
            got[t["tile"]] = t
    ctx = mgf_amd.Context(0)
    scs = [scenes.sphere_pile_tile(*dims, r, P) for r in range(P)]
    worlds = []
    for sc in scs:
        w = mgf_amd.World.from_scene(ctx, sc)
        w.set_tags(sc["tags"])
        worlds.append(w)
    T = mgf_amd.Tiles(ctx, worlds, [sc["x_range"] for sc in scs])
    dt, it = float(scs[0]["dt"]), scs[0]["iters"]
    for _ in range(ticks):
        T.step(dt, it)
    assert sum(T.migrated(k) for k in range(P)) > 10000  # (the collapse is under way)
    for k, w in enumerate(worlds):
        assert np.array_equal(w.tags(), got[k]["tags"]), f"tile {k}: bodies / order differ"
        st = w.state()
        for f in STATE_KEYS:
            assert np.array_equal(st[f].view(np.uint32), got[k][f].view(np.uint32)), f"tile {k}: {f} differs"
    del T, worlds
    ctx.close()
