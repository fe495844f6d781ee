"""The tile protocol under the C-ABI (mgf_tiles_*: SURVEY.md §8e) against the oracle's tile mode: several
x-slab tiles of one scene stepped in ONE process on one GPU (the exchange between them is then a device copy; between
processes it is RCCL - same code path apart from the transfer primitive), bit for bit, tile by tile.  Includes BASELINE
config 4 at full size: 1 048 576 spheres as 8 tiles of 16 x 128 x 64."""
import numpy as np
import pytest

import mgf_amd
from mgf_amd import scenes
from tests.util import values_equal

pytestmark = pytest.mark.gpu
STATE_KEYS = ("x", "q", "v", "omega", "delta")


@pytest.fixture(scope="module")
def ctx():
    c = mgf_amd.Context(0)
    yield c
    c.close()


def _native(ctx, tile_scenes, **kw):
    worlds = []
    for sc in tile_scenes:
        w = mgf_amd.World.from_scene(ctx, sc)
        w.set_tags(sc["tags"])
        worlds.append(w)
    return mgf_amd.Tiles(ctx, worlds, [sc["x_range"] for sc in tile_scenes], **kw), worlds


def _oracle_tiles(tile_scenes, **kw):
    from mgf_amd.tiles import Tile
    from tests.oracle_engine import OracleEngine
    P = len(tile_scenes)
    return [Tile(OracleEngine(sc), sc["x_range"], r, P, sc["dt"], sc["iters"], **kw) for r, sc in enumerate(tile_scenes)]


def _assert_equal(worlds, ot, what):
    for r, (w, o) in enumerate(zip(worlds, ot)):
        assert len(w) == len(o.e.w), f"{what}: tile {r} owns {len(w)} bodies, oracle {len(o.e.w)}"
        assert np.array_equal(w.tags(), o.e.tags()), f"{what}: tile {r} body order"
        sg, so = w.state(), o.e.state()
        for k in STATE_KEYS:
            assert values_equal(sg[k], so[k]), f"{what}: tile {r} {k}"


@pytest.mark.parametrize("P,refresh_every", [(2, 2), (3, 2), (4, 1), (3, 10)])
def test_native_tiles_match_the_oracle_tiles_with_hand_overs(ctx, P, refresh_every):
    """A pile drifting at 5 m/s through P tiles: ghost exchange, velocity refresh every R iterations, bodies changing owner."""
    from mgf_amd.tiles import step_tiles_inprocess
    nx, ny, nz = 4, 4, 5
    tile_scenes = [scenes.sphere_pile_tile(nx, ny, nz, r, P, drift=(5.0, 0.0, 0.0)) for r in range(P)]
    T, worlds = _native(ctx, tile_scenes, refresh_every=refresh_every)
    ot = _oracle_tiles(tile_scenes, refresh_every=refresh_every)
    dt, iters = float(tile_scenes[0]["dt"]), tile_scenes[0]["iters"]
    for tick in range(40):
        sg, so = T.step(dt, iters), step_tiles_inprocess(ot)
        for r in range(P):
            assert sg[r].n_constraints == so[r]["n_constraints"], f"tick {tick} tile {r}"
        assert [T.migrated(r) for r in range(P)] == [t.n_migrated_in for t in ot], f"tick {tick}"
        if tick % 8 == 7:
            _assert_equal(worlds, ot, f"tick {tick}")
    assert sum(T.migrated(r) for r in range(P)) >= ny * nz and T.migrated(P - 1) > 0 and T.migrated(0, incoming=False) > 0
    _assert_equal(worlds, ot, "end")
    assert np.array_equal(np.sort(np.concatenate([w.tags() for w in worlds])), np.arange(P * nx * ny * nz))


def test_native_tiles_equal_the_python_driver(ctx):
    """mgf_tiles_step and mgf_amd.tiles.step_tiles_inprocess are the same protocol: same bits, capsules over a heightfield
    included (terrain constraints stay with the owner, ghosts bring another body kind)."""
    from mgf_amd.tiles import HipEngine, Tile, step_tiles_inprocess
    P = 3
    tile_scenes = [scenes.sphere_pile_tile(5, 4, 4, r, P, drift=(-3.0, 0.0, 0.0)) for r in range(P)]
    T, worlds = _native(ctx, tile_scenes)
    py = [Tile(HipEngine(ctx, sc, 0), sc["x_range"], r, P, sc["dt"], sc["iters"]) for r, sc in enumerate(tile_scenes)]
    dt, iters = float(tile_scenes[0]["dt"]), tile_scenes[0]["iters"]
    for tick in range(30):
        sg, sp = T.step(dt, iters), step_tiles_inprocess(py)
        assert [int(s.n_constraints) for s in sg] == [int(s["n_constraints"]) for s in sp], tick
    for r in range(P):
        assert np.array_equal(worlds[r].tags(), py[r].e.tags())
        a, b = worlds[r].state(), py[r].e.state()
        for k in STATE_KEYS:
            assert np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)), (r, k)
    assert sum(T.migrated(r) for r in range(P)) > 0


def test_one_rank_communicator_and_preflight(ctx):
    """RCCL is loaded at run time and a communicator of one rank comes up on the box's GPU: the pre-flight sum sees 1 rank,
    and a connected set of tiles without remote neighbours steps like an unconnected one."""
    tile_scenes = [scenes.sphere_pile_tile(4, 4, 4, r, 2) for r in range(2)]
    A, wa = _native(ctx, tile_scenes)
    B, wb = _native(ctx, tile_scenes)
    assert A.preflight() == 0
    uid = mgf_amd.rccl_unique_id()
    assert len(uid) == 128 and any(uid)
    A.connect(uid, 0, 1)
    assert A.preflight() == 1
    dt, iters = float(tile_scenes[0]["dt"]), tile_scenes[0]["iters"]
    for _ in range(12):
        A.step(dt, iters)
        B.step(dt, iters)
    for x, y in zip(wa, wb):
        sx, sy = x.state(), y.state()
        for k in STATE_KEYS:
            assert np.array_equal(sx[k].view(np.uint32), sy[k].view(np.uint32)), k


def test_config4_one_million_spheres_as_eight_tiles(ctx):
    """BASELINE config 4 at full size on one GPU: 128 x 128 x 64 spheres cut into 8 x-slabs of 16 lattice columns, the first
    tick and a later, contact-rich tick (the oracle teacher-forced from the GPU's state) bit-identical to the oracle's 8 tiles."""
    from mgf_amd.tiles import step_tiles_inprocess
    P, nx, ny, nz = 8, 16, 128, 64
    tile_scenes = [scenes.sphere_pile_tile(nx, ny, nz, r, P) for r in range(P)]
    assert sum(len(sc["comps"]) for sc in tile_scenes) == 1048576
    T, worlds = _native(ctx, tile_scenes)
    ot = _oracle_tiles(tile_scenes)
    dt, iters = float(tile_scenes[0]["dt"]), tile_scenes[0]["iters"]
    sg, so = T.step(dt, iters), step_tiles_inprocess(ot)
    assert [int(s.n_constraints) for s in sg] == [int(s["n_constraints"]) for s in so]
    _assert_equal(worlds, ot, "first tick")
    for _ in range(14):
        sg = T.step(dt, iters)
    assert sum(int(s.n_constraints) for s in sg) > 1000000
    for w, o in zip(worlds, ot):  # no body has changed tile yet: the oracle tiles take the GPU's state as it is
        assert np.array_equal(w.tags(), o.e.tags())
        s = w.state()
        o.e.w.set_state(x=s["x"], q=s["q"], v=s["v"], omega=s["omega"], delta=s["delta"])
    sg, so = T.step(dt, iters), step_tiles_inprocess(ot)
    assert [int(s.n_constraints) for s in sg] == [int(s["n_constraints"]) for s in so]
    _assert_equal(worlds, ot, "later tick")
    print("config 4, 8 tiles: constraints per tile", [int(s.n_constraints) for s in sg])


def test_config5_65536_two_part_bodies_as_eight_tiles(ctx):
    """BASELINE config 5 as BASELINE.json states it - 65 536 bodies of a sphere and a capsule each, cut into 8 x-slab tiles - on one
    GPU (the exchange between a rank's own tiles is a device copy, between ranks RCCL: same code): the first tick, and a later
    contact-rich one (the oracle's tiles teacher-forced from the GPU's state), bit-identical to the oracle's 8 tiles; ghost records
    carry the bodies' parts across every face.  (This build's own definition of such a body: SURVEY 8f row 1.)"""
    from mgf_amd.tiles import step_tiles_inprocess
    P = 8
    sc = scenes.dumbbell_field(64, 16, 64)
    tile_scenes = scenes.split_by_slabs(sc, P, 64 * 2.2 / 2.0)
    assert sum(len(t["compound"]["offsets"]) - 1 for t in tile_scenes) == 65536
    T, worlds = _native(ctx, tile_scenes, halo=2.0)
    ot = _oracle_tiles(tile_scenes, halo=2.0)
    dt, iters = float(sc["dt"]), sc["iters"]
    sg, so = T.step(dt, iters), step_tiles_inprocess(ot)
    assert [int(s.n_constraints) for s in sg] == [int(s["n_constraints"]) for s in so]
    _assert_equal(worlds, ot, "first tick")
    for _ in range(62):
        sg = T.step(dt, iters)
    assert sum(int(s.n_constraints) for s in sg) > 10000 and sum(int(s.n_ghost_constraints) for s in sg) > 100
    # a few bodies have changed tile by now: the oracle's tiles are rebuilt from the GPU's - each tile's bodies in the GPU tile's order
    # (tags = indices into the undivided scene) - and take its state
    cb, off = sc["compound"], sc["compound"]["offsets"]
    rebuilt = []
    for w, t in zip(worlds, tile_scenes):
        tags = w.tags().astype(np.int64)
        parts = np.concatenate([np.arange(off[b], off[b + 1]) for b in tags])
        sub = dict(t)
        sub["compound"] = dict(comps=cb["comps"][parts], comp_mass=np.asarray(cb["comp_mass"], np.float32)[parts],
                               offsets=np.arange(0, 2 * len(tags) + 1, 2, dtype=np.int64), restitution=cb["restitution"][tags],
                               friction=cb["friction"][tags], force=cb["force"][tags])
        sub["v0"] = sc["v0"][tags]
        sub["tags"] = tags.astype(np.uint32)
        rebuilt.append(sub)
    ot = _oracle_tiles(rebuilt, halo=2.0)
    for w, o in zip(worlds, ot):
        assert np.array_equal(w.tags(), o.e.tags())
        s = w.state()
        o.e.w.set_state(x=s["x"], q=s["q"], v=s["v"], omega=s["omega"], delta=s["delta"])
    sg, so = T.step(dt, iters), step_tiles_inprocess(ot)
    assert [int(s.n_constraints) for s in sg] == [int(s["n_constraints"]) for s in so]
    _assert_equal(worlds, ot, "later tick")
    print("config 5, 8 tiles: constraints per tile", [int(s.n_constraints) for s in sg], "hand-overs", sum(T.migrated(r) for r in range(P)))


def test_halo_smaller_than_a_body_is_refused(ctx):
    """a body whose fat half extent exceeds the halo could touch a neighbour's body that was never exported.  Both
    drivers fail loudly instead of dropping the contact."""
    from mgf_amd.tiles import HipEngine, Tile, step_tiles_inprocess
    tile_scenes = [scenes.sphere_pile_tile(4, 4, 4, r, 2) for r in range(2)]
    T, _ = _native(ctx, tile_scenes, halo=0.5)  # spheres of radius 0.5 + fat margin 0.25 + motion: 0.75 and more
    with pytest.raises(mgf_amd.MgfError) as e:
        T.step(float(tile_scenes[0]["dt"]), 10)
    assert e.value.status == 6 and "halo" in str(e.value)
    py = [Tile(HipEngine(ctx, sc, 0), sc["x_range"], r, 2, sc["dt"], sc["iters"], halo=0.5) for r, sc in enumerate(tile_scenes)]
    with pytest.raises(ValueError):
        step_tiles_inprocess(py)


def _penetration(x, owner, radius=0.5):
    """mean overlap depth of touching sphere pairs: (pairs whose bodies have different owners, the others, #different)"""
    from scipy.spatial import cKDTree
    x = x.astype(np.float64)
    pairs = cKDTree(x).query_pairs(2 * radius, output_type="ndarray")
    pen = 2 * radius - np.linalg.norm(x[pairs[:, 0]] - x[pairs[:, 1]], axis=1)
    cross = owner[pairs[:, 0]] != owner[pairs[:, 1]]
    return float(pen[cross].mean()), float(pen[~cross].mean()), int(cross.sum())


def test_seam_quality_at_scale_against_the_undivided_world(ctx):
    """What the block-Jacobi coupling across a slab face costs physically, at a size the CPU test cannot reach ("only
    characterised on a 6x6x6 pile"): 4 tiles of 16 x 24 x 32 spheres (49 152 bodies) for 240 ticks with the ghost velocities
    refreshed every R = 1, 2, 10 iterations, against the same pile as ONE world in the exact Gauss-Seidel order.  Measured: mean
    resting penetration of touching pairs across a slab face vs inside a tile, and the same two sets of pairs in the undivided
    world (its "seam" = the pairs that straddle the planes where the tiles' faces would be)."""
    P, nx, ny, nz, ticks = 4, 16, 24, 32, 240
    tile_scenes = [scenes.sphere_pile_tile(nx, ny, nz, r, P) for r in range(P)]
    merged = dict(tile_scenes[0])
    for key in ("comps", "mass", "restitution", "friction", "force", "v0"):
        merged[key] = np.concatenate([s[key] for s in tile_scenes])
    one = mgf_amd.World.from_scene(ctx, merged)
    dt, iters = float(merged["dt"]), merged["iters"]
    for _ in range(ticks):
        one.step(dt, iters)
    x1 = one.state()["x"]
    edges = np.array([sc["x_range"][1] for sc in tile_scenes[:-1]])
    ref_seam, ref_in, ref_n = _penetration(x1, np.searchsorted(edges, x1[:, 0]))
    print(f"undivided world: pairs across the would-be faces {ref_seam:.4f} ({ref_n} pairs), the others {ref_in:.4f}")
    rows = {}
    for R in (1, 2, 10):
        T, worlds = _native(ctx, tile_scenes, refresh_every=R)
        for _ in range(ticks):
            T.step(dt, iters)
        x = np.concatenate([w.state()["x"][:len(w)] for w in worlds])
        owner = np.concatenate([np.full(len(w), r) for r, w in enumerate(worlds)])
        rows[R] = _penetration(x, owner)
        # how far the tiled pile is from the undivided one, body by body (tags are global ids)
        tags = np.concatenate([w.tags()[:len(w)] for w in worlds])
        dev = np.linalg.norm(x[np.argsort(tags)].astype(np.float64) - x1.astype(np.float64), axis=1)
        print(f"tiles, refresh every {R:2d} iterations: seam penetration {rows[R][0]:.4f} ({rows[R][2]} pairs), interior {rows[R][1]:.4f}; "
              f"distance from the undivided world: median {np.median(dev):.4f}, 99th percentile {np.percentile(dev, 99):.4f}")
    assert ref_n > 500 and rows[2][2] > 500
    assert rows[2][0] < 1.25 * rows[2][1] and rows[1][0] < 1.25 * rows[1][1]   # the default keeps the seam at the interior's level
    assert rows[2][1] < 1.1 * ref_in                                          # ... and the interior at the undivided world's


def test_a_tick_lost_to_a_solver_launch_that_gave_up_is_repeated(ctx):
    """Solver::solve has no failure mode (solver.rs:72-78), and a tile set's tick has none either when a persistent solver launch gives up
    under it (a device shared with another process): every tile's bodies go back to where the tick found them, the tiles back off to the
    launch-per-frontier executor and the tick - exchanges, ghost refreshes and all - is repeated; the set stays bit-identical to the
    oracle's tiles.  (The middle tile's launches are made to give up: option flow_spin_limit = 1, small blocks so that it has several.)"""
    from mgf_amd.tiles import step_tiles_inprocess
    P, nx, ny, nz = 3, 5, 4, 5
    tile_scenes = [scenes.sphere_pile_tile(nx, ny, nz, r, P, drift=(4.0, 0.0, 0.0)) for r in range(P)]
    T, worlds = _native(ctx, tile_scenes)
    for w in worlds:
        w.set_option("flow5_block", 16)
    worlds[1].set_option("flow_spin_limit", 1)
    ot = _oracle_tiles(tile_scenes)
    dt, iters = float(tile_scenes[0]["dt"]), tile_scenes[0]["iters"]
    for tick in range(40):
        sg, so = T.step(dt, iters), step_tiles_inprocess(ot)
        assert [int(s.n_constraints) for s in sg] == [int(s["n_constraints"]) for s in so], tick
        if tick % 10 == 9:
            _assert_equal(worlds, ot, f"tick {tick}")
    assert T.counter("ticks_retried") >= 1 and worlds[1].counter("solver_abort_fallbacks") >= 1
    assert T.counter("ticks") == 40 and sum(T.migrated(r) for r in range(P)) > 0
    # ... and with the repetition switched off such a tick is reported as lost, as before
    worlds[1].set_option("flow_spin_limit", 1)
    T.set_option("retry_lost_ticks", 0)
    with pytest.raises(mgf_amd.MgfError):
        for _ in range(80):
            T.step(dt, iters)


def test_ghost_records_of_two_widths_across_one_face(ctx):
    """r06: a world without bodies of several components sends 40-float ghost records, one with such bodies 72 (k_tiles.h); the receiver learns
    the width from the kinds its neighbour announces.  A tile of two-part bodies beside a tile of plain spheres, driven into each other: records
    of both widths cross the face in opposite directions, then bodies change owner and the plain tile's width changes in mid-run - against the
    oracle's tiles, bit for bit."""
    from mgf_amd.tiles import step_tiles_inprocess
    sc = scenes.dumbbell_field(3, 2, 4, n_plain=24)
    cb = sc["compound"]
    cb["comps"]["p"][:, 0] -= np.float32(cb["comps"]["p"][:, 0].max() + 1.2)      # every two-part body left of x = 0 ...
    n_plain = len(sc["comps"])
    sc["comps"]["p"][:, 0] = np.float32(0.8) + np.float32(0.9) * (np.arange(n_plain) % 4).astype(np.float32)   # ... every plain sphere right of it
    sc["comps"]["p"][:, 1] = np.float32(1.0) + np.float32(1.1) * (np.arange(n_plain) // 4).astype(np.float32)
    sc["comps"]["p"][:, 2] = np.float32(-2.0) + np.float32(1.3) * (np.arange(n_plain) % 3).astype(np.float32)
    sc["v0"][:n_plain] = np.float32([-2.0, 0.0, 0.0])
    sc["v0"][n_plain:] = np.float32([3.0, 0.0, 0.0])
    tile_scenes = scenes.split_by_slabs(sc, 2, 12.0)
    assert len(tile_scenes[0]["comps"]) == 0 and len(tile_scenes[0]["compound"]["offsets"]) - 1 == 24          # compound only
    assert len(tile_scenes[1]["comps"]) == n_plain and len(tile_scenes[1]["compound"]["offsets"]) - 1 == 0     # plain only
    T, worlds = _native(ctx, tile_scenes, halo=2.0)
    ot = _oracle_tiles(tile_scenes, halo=2.0)
    assert worlds[0].counter("body_kinds") & 4 and not worlds[1].counter("body_kinds") & 4
    dt, iters = float(sc["dt"]), sc["iters"]
    ghosts = moved = 0
    for tick in range(90):
        sg, so = T.step(dt, iters), step_tiles_inprocess(ot)
        assert [int(s.n_constraints) for s in sg] == [int(s["n_constraints"]) for s in so], tick
        ghosts += sum(int(s.n_ghost_constraints) for s in sg)
        if tick % 10 == 9:
            _assert_equal(worlds, ot, f"tick {tick}")
    moved = T.migrated(0) + T.migrated(1)
    assert ghosts > 20 and moved > 0, (ghosts, moved)   # (contacts across the face, and bodies that changed tile)
    assert worlds[1].counter("body_kinds") & 4           # (the plain tile has taken two-part bodies in: its records are 72 floats wide now)
    _assert_equal(worlds, ot, "end")
