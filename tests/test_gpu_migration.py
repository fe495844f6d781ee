"""Bodies changing tile: mgf_world_select_tile / export_migrants / remove_bodies / import_migrants and
the tiles driver's hand-over, on the GPU against the oracle's tile mode - bit for bit, tile by tile."""
import numpy as np
import pytest

from tests.util import oracle_world, rel_err, values_equal

pytestmark = pytest.mark.gpu

DRIFT = (5.0, 0.0, 0.0)
STATE_KEYS = ("x", "q", "v", "omega", "delta")


@pytest.fixture(scope="module")
def ctx():
    import mgf_amd
    c = mgf_amd.Context(0)
    yield c
    c.close()


def _pair_of_tile_sets(ctx, scenes_per_tile, **kw):
    from mgf_amd.tiles import HipEngine, Tile
    from tests.oracle_engine import OracleEngine
    P = len(scenes_per_tile)
    gt = [Tile(HipEngine(ctx, sc, 0), sc["x_range"], r, P, sc["dt"], sc["iters"], **kw) for r, sc in enumerate(scenes_per_tile)]
    ot = [Tile(OracleEngine(sc), sc["x_range"], r, P, sc["dt"], sc["iters"], **kw) for r, sc in enumerate(scenes_per_tile)]
    return gt, ot


def _assert_tiles_equal(gt, ot, what):
    for r, (g, o) in enumerate(zip(gt, ot)):
        assert len(g.e.world) == len(o.e.w), f"{what}: tile {r} owns {len(g.e.world)} bodies, oracle {len(o.e.w)}"
        assert np.array_equal(g.e.tags(), o.e.tags()), f"{what}: tile {r} body order"
        sg, so = g.e.state(), o.e.state()
        for k in STATE_KEYS:
            assert values_equal(sg[k], so[k]), f"{what}: tile {r} {k}: rel err {rel_err(sg[k], so[k])}"


@pytest.mark.parametrize("P,refresh_every", [(2, 2), (3, 2), (4, 1)])
def test_drifting_pile_changes_tiles_like_the_oracle(ctx, P, refresh_every):
    from mgf_amd import scenes
    from mgf_amd.tiles import step_tiles_inprocess
    nx, ny, nz = 4, 4, 5
    gt, ot = _pair_of_tile_sets(ctx, [scenes.sphere_pile_tile(nx, ny, nz, r, P, drift=DRIFT) for r in range(P)], refresh_every=refresh_every)
    for tick in range(40):
        sg, so = step_tiles_inprocess(gt), step_tiles_inprocess(ot)
        for r in range(P):
            assert sg[r]["n_constraints"] == so[r]["n_constraints"], f"tick {tick} tile {r}"
        assert [t.n_migrated_in for t in gt] == [t.n_migrated_in for t in ot], f"tick {tick}"
        if tick % 8 == 7:
            _assert_tiles_equal(gt, ot, f"tick {tick}")
    assert sum(t.n_migrated_in for t in gt) >= ny * nz  # a lattice layer's worth of hand-overs
    assert gt[-1].n_migrated_in > 0 and gt[0].n_migrated_out > 0
    _assert_tiles_equal(gt, ot, "end")
    all_tags = np.sort(np.concatenate([t.e.tags() for t in gt]))
    assert np.array_equal(all_tags, np.arange(P * nx * ny * nz))


def _two_kind_scenes():
    """Tile 0 holds spheres only, tile 1 capsules only; the spheres roll to the right, so tile 1 first sees them as
    ghosts of a kind it does not own and then receives them."""
    from mgf_amd import scenes
    terrain = scenes.box_terrain(5.0, 6.0, (0.0, 0.0, 0.0))
    i, j, k = np.meshgrid(np.arange(3), np.arange(2), np.arange(4), indexing="ij")
    sc = np.stack([-0.55 - 1.05 * i.ravel(), 0.5 + 1.02 * j.ravel(), -1.6 + 1.07 * k.ravel()], axis=1).astype(np.float32)
    s0 = scenes._scene("spheres_left", scenes._spheres(sc, 0.5), terrain, v0=np.tile(np.float32([4.0, 0.0, 0.0]), (len(sc), 1)))
    i, j, k = np.meshgrid(np.arange(2), np.arange(2), np.arange(3), indexing="ij")
    cc = np.stack([0.9 + 1.3 * i.ravel(), 0.45 + 1.0 * j.ravel(), -1.5 + 1.5 * k.ravel()], axis=1).astype(np.float32)
    dirs = np.tile(np.float32([0.0, 0.0, 1.0]), (len(cc), 1))
    s1 = scenes._scene("capsules_right", scenes._capsules(cc, dirs, 0.35, 0.4), terrain, v0=np.zeros((len(cc), 3), np.float32))
    s0["x_range"], s1["x_range"] = (-5.0, 0.0), (0.0, 5.0)
    s0["tags"] = np.arange(len(sc), dtype=np.uint32)
    s1["tags"] = (100 + np.arange(len(cc))).astype(np.uint32)
    return [s0, s1]


def test_tiles_of_different_body_kinds(ctx):
    """The narrowphase dispatch is chosen on the host from the kinds a world holds: ghosts and arrivals of another
    kind must switch it to the mixed path (kind masks ride on the count message / are read from the arrivals)."""
    from mgf_amd.tiles import step_tiles_inprocess
    gt, ot = _pair_of_tile_sets(ctx, _two_kind_scenes(), halo=1.25)  # (the capsules' cube bounds reach 1.0 and more with their motion)
    assert gt[0].e.kinds() == 1 and gt[1].e.kinds() == 2
    cross = 0
    for tick in range(45):
        sg, so = step_tiles_inprocess(gt), step_tiles_inprocess(ot)
        for r in range(2):
            assert sg[r]["n_constraints"] == so[r]["n_constraints"], f"tick {tick} tile {r}"
        cross += sg[1]["n_constraints"] - sg[1]["n_terrain_constraints"]
    assert gt[1].n_migrated_in > 0 and cross > 0
    assert gt[0].e.kinds() == 3 and gt[1].e.kinds() == 3
    _assert_tiles_equal(gt, ot, "two kinds")


def test_migrant_round_trip_on_one_world(ctx):
    """export -> remove -> import on the same world: every device array of the moved bodies comes back verbatim (they
    are re-appended at the end), and the next ticks equal the oracle doing the same."""
    import torch
    import mgf_amd
    from mgf_amd import scenes
    from mgf_amd.tiles import MIGRANT_FLOATS
    sc = scenes.capsule_field_dense(5, 2, 5, sphere_fraction=0.5)
    dt, iters = float(sc["dt"]), sc["iters"]
    gw, ow = mgf_amd.World.from_scene(ctx, sc), oracle_world(sc)
    n = len(gw)
    tags = np.arange(n, dtype=np.uint32) * 3 + 1
    gw.set_tags(tags); ow.set_tags(tags)
    for _ in range(3):
        gw.step(dt, iters); ow.step(dt, iters)
    before = gw.state()
    ids = np.array([0, 3, 4, n // 2, n - 1], np.uint32)
    d_ids = torch.from_numpy(ids.astype(np.int32)).cuda()
    rec = torch.empty((len(ids), MIGRANT_FLOATS), dtype=torch.float32, device="cuda")
    gw.export_migrants(d_ids.data_ptr(), len(ids), rec.data_ptr())
    gw.remove_bodies(d_ids.data_ptr(), len(ids))
    assert len(gw) == n - len(ids)
    keep = np.setdiff1d(np.arange(n), ids)
    assert np.array_equal(gw.tags(), tags[keep])
    gw.import_migrants(rec.data_ptr(), len(ids))
    torch.cuda.synchronize()
    order = np.concatenate([keep, ids])
    assert np.array_equal(gw.tags(), tags[order])
    after = gw.state()
    for k in STATE_KEYS:
        assert np.array_equal(after[k], before[k][order]), k
    orec = ow.export_migrants(ids)
    ow.remove_bodies(ids)
    ow.import_migrants(orec)
    for _ in range(4):
        sg, so = gw.step(dt, iters), ow.step(dt, iters)
        assert sg.n_constraints == so.n_constraints
    g, o = gw.state(), ow.state()
    for k in STATE_KEYS:
        assert values_equal(g[k], o[k]), f"{k}: rel err {rel_err(g[k], o[k])}"


@pytest.mark.timeout(300)
def test_three_ranks_on_one_gpu_match_oracle_tiles(tmp_path):
    """step_tile over a real transport (3 gloo ranks, payloads staged through the host, every rank's tile on GPU 0),
    with bodies changing tile: each rank's result equals the oracle's in-process tile loop bit for bit."""
    import os
    import subprocess
    import sys
    from mgf_amd import scenes
    from mgf_amd.tiles import Tile, step_tiles_inprocess
    from tests.oracle_engine import OracleEngine
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ws, nx, ny, nz, ticks = 3, 4, 4, 5, 30
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", PYTHONPATH=root, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ws}", "--master-addr", "127.0.0.1",
           "--master-port", "29547", os.path.join(root, "tests", "tile_worker.py"), str(tmp_path), str(nx), str(ny), str(nz), str(ticks),
           str(DRIFT[0]), "hip"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    scs = [scenes.sphere_pile_tile(nx, ny, nz, k, ws, drift=DRIFT) for k in range(ws)]
    ot = [Tile(OracleEngine(sc), sc["x_range"], k, ws, sc["dt"], sc["iters"]) for k, sc in enumerate(scs)]
    ncons = [[] for _ in range(ws)]
    for _ in range(ticks):
        for k, s in enumerate(step_tiles_inprocess(ot)):
            ncons[k].append(s["n_constraints"])
    assert sum(t.n_migrated_in for t in ot) > 0
    for rank in range(ws):
        got = np.load(tmp_path / f"rank{rank}.npz")
        assert got["ncons"].tolist() == ncons[rank]
        assert np.array_equal(got["tags"], ot[rank].e.tags())
        want = ot[rank].e.state()
        for k in STATE_KEYS:
            assert values_equal(got[k], want[k]), f"rank {rank} {k}"


@pytest.mark.parametrize("n_gone", [3, 64, 65, 700])
def test_remove_bodies_short_and_long_lists_keep_the_others_in_order(ctx, n_gone):
    """The compaction behind a removal has two ways to the kept bodies' new slots - one launch for a handful of ids (a tile's hand-over),
    count / offsets / positions for a long list: both must leave exactly the kept bodies, in their old order, every array verbatim."""
    import torch
    import mgf_amd
    from mgf_amd import scenes
    gw = mgf_amd.World.from_scene(ctx, scenes.sphere_pile(10, 10, 10))
    n = len(gw)
    tags = np.arange(n, dtype=np.uint32) + 7
    gw.set_tags(tags)
    gw.step(1.0 / 60.0, 4)
    before = gw.state()
    rng = np.random.default_rng(n_gone)
    ids = rng.permutation(n)[:n_gone].astype(np.uint32)  # (any order)
    d = torch.from_numpy(ids.astype(np.int32)).cuda()
    gw.remove_bodies(d.data_ptr(), len(ids))
    keep = np.setdiff1d(np.arange(n), ids)
    assert len(gw) == len(keep) and np.array_equal(gw.tags(), tags[keep])
    after = gw.state()
    for k in STATE_KEYS:
        assert np.array_equal(after[k], before[k][keep]), k
    gw.step(1.0 / 60.0, 4)


def test_remove_bodies_rejects_bad_ids(ctx):
    import torch
    import mgf_amd
    from mgf_amd import scenes
    gw = mgf_amd.World.from_scene(ctx, scenes.sphere_pile(5, 5, 5))
    for bad in ([1, 1], [5, 400], list(range(70)) + [3], list(range(70)) + [400]):  # (short lists and long ones: two ways to the new slots)
        d = torch.tensor(bad, dtype=torch.int32, device="cuda")
        with pytest.raises(mgf_amd.MgfError):
            gw.remove_bodies(d.data_ptr(), len(bad))
    assert len(gw) == 125
    gw.step(1.0 / 60.0, 4)


@pytest.mark.parametrize("target", ["two_part_bodies", "ordinary_bodies"])
def test_arrivals_of_more_parts_than_the_world_has_slots_for_raise_them(ctx, target):
    """through the plain C API (no tile set announcing its neighbours' kinds, no option body_kinds) bodies of three and four
    parts arrive in a world that holds two-part bodies - or ordinary ones only, with no part arrays at all.  Nothing may be dropped on
    the way in: the world reads the records' part counts first and raises its slots."""
    import torch
    import mgf_amd
    from mgf_amd import scenes
    from mgf_amd.tiles import GHOST_FLOATS, MIGRANT_FLOATS
    jacks = scenes.jack_field(3, 2, 3)
    src = mgf_amd.World.from_scene(ctx, jacks)
    dst = mgf_amd.World.from_scene(ctx, scenes.dumbbell_field(3, 2, 3) if target == "two_part_bodies" else scenes.sphere_pile(4, 3, 4))
    dt, iters = float(jacks["dt"]), jacks["iters"]
    src.step(dt, iters); dst.step(dt, iters)
    assert src.counter("body_kinds") & 8 and not dst.counter("body_kinds") & 8
    ids = np.array([0, 5, len(src) - 1], np.uint32)
    d_ids = torch.from_numpy(ids.astype(np.int32)).cuda()
    rec = torch.empty((len(ids), MIGRANT_FLOATS), dtype=torch.float32, device="cuda")
    src.export_migrants(d_ids.data_ptr(), len(ids), rec.data_ptr())
    n0 = len(dst)
    dst.import_migrants(rec.data_ptr(), len(ids))
    torch.cuda.synchronize()
    assert len(dst) == n0 + len(ids) and dst.counter("body_kinds") & 12 == 12
    back = torch.empty_like(rec)
    d_new = torch.arange(n0, n0 + len(ids), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()  # (arange is a kernel on torch's stream; the library reads the ids on its own, non-blocking one: without this the export
    #                           raced it and now and then exported whatever ids the memory held - the suite's one flaky test of round 6)
    dst.export_migrants(d_new.data_ptr(), len(ids), back.data_ptr())
    torch.cuda.synchronize()
    assert torch.equal(rec.view(torch.int32), back.view(torch.int32))  # every part slot arrived
    # ... and as ghosts: a fresh world of the same kind takes the jacks' ghost records
    dst2 = mgf_amd.World.from_scene(ctx, scenes.dumbbell_field(3, 2, 3) if target == "two_part_bodies" else scenes.sphere_pile(4, 3, 4))
    grec = torch.empty((len(ids), GHOST_FLOATS), dtype=torch.float32, device="cuda")
    src.export_bodies(d_ids.data_ptr(), len(ids), grec.data_ptr())
    dst2.begin_tick(dt)
    dst2.import_ghosts(grec.data_ptr(), len(ids))
    assert dst2.ghost_len() == len(ids) and dst2.counter("body_kinds") & 12 == 12
    dst2.collide(dt)
    dst.step(dt, iters)
