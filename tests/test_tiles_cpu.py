"""Multi-tile (multi-GPU) host logic on CPU: the tile driver of mgf_amd/tiles.py with the oracle
engine, in one process and across two gloo ranks."""
import os
import subprocess
import sys

import numpy as np
import pytest

from mgf_amd import scenes
from mgf_amd.tiles import Tile, step_tiles_inprocess
from tests.oracle_engine import OracleEngine
from tests.util import oracle_world, rel_err

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def make_tiles(nx, ny, nz, P, engine=OracleEngine, drift=None, **kw):
    tiles = []
    for r in range(P):
        sc = scenes.sphere_pile_tile(nx, ny, nz, r, P, drift=drift)
        tiles.append(Tile(engine(sc), sc["x_range"], r, P, sc["dt"], sc["iters"], **kw))
    return tiles


def test_single_tile_equals_plain_world():
    sc = scenes.sphere_pile_tile(6, 5, 6, 0, 1)
    tile = Tile(OracleEngine(sc), sc["x_range"], 0, 1, sc["dt"], sc["iters"])
    ow = oracle_world(sc)
    for _ in range(5):
        step_tiles_inprocess([tile])
        ow.step(float(sc["dt"]), sc["iters"])
    a, b = tile.e.state(), ow.state()
    for k in a:
        assert np.array_equal(a[k], b[k]), k


def test_two_tiles_inprocess_ghost_protocol():
    tiles = make_tiles(6, 5, 6, 2)
    n_owned = [len(t.e.w) for t in tiles]
    for tick in range(8):
        stats = step_tiles_inprocess(tiles)
        assert all(len(t.e.w) == n for t, n in zip(tiles, n_owned))  # ghosts are not owned bodies
        assert all(s["n_constraints"] > 0 for s in stats)
    # each tile exported the bodies of its inner face and nothing else
    xl = tiles[0].e.state()["x"][tiles[0].e.ids[1], 0]
    assert len(xl) > 0 and xl.min() > tiles[0].x_hi - 2.5 and len(tiles[0].e.ids[0]) == 0
    xr = tiles[1].e.state()["x"][tiles[1].e.ids[0], 0]
    assert len(xr) > 0 and xr.max() < tiles[1].x_lo + 2.5 and len(tiles[1].e.ids[1]) == 0
    for t in tiles:
        s = t.e.state()
        assert np.isfinite(s["x"]).all() and np.isfinite(s["v"]).all()


def test_three_tiles_middle_tile_has_two_neighbours():
    """The middle slab exchanges with both sides: its ghosts come from two tiles and its boundary bodies go two ways."""
    tiles = make_tiles(4, 4, 4, 3)
    for tick in range(10):
        stats = step_tiles_inprocess(tiles)
        assert all(s["n_constraints"] > 0 for s in stats)
    mid = tiles[1]
    assert len(mid.e.ids[0]) > 0 and len(mid.e.ids[1]) > 0
    assert len(tiles[0].e.ids[0]) == 0 and len(tiles[2].e.ids[1]) == 0
    x = mid.e.state()["x"][:, 0]
    assert x[mid.e.ids[0]].max() < mid.x_lo + 2.5 and x[mid.e.ids[1]].min() > mid.x_hi - 2.5
    # a body near both faces of a narrow slab may be exported both ways; every export is an owned body
    assert set(mid.e.ids[0]) <= set(range(len(mid.e.w))) and set(mid.e.ids[1]) <= set(range(len(mid.e.w)))
    for t in tiles:
        s = t.e.state()
        assert np.isfinite(s["x"]).all() and np.isfinite(s["v"]).all()


def test_tiled_result_tracks_the_undivided_world():
    """Block-Jacobi coupling across the slab face is a different iteration than one global Gauss-Seidel;
    the deviation is measured and bounded, not hidden."""
    P, nx, ny, nz = 2, 5, 4, 5
    tiles = make_tiles(nx, ny, nz, P)
    # the same bodies in one world: concatenate the tiles' scenes
    scs = [scenes.sphere_pile_tile(nx, ny, nz, r, P) for r in range(P)]
    merged = dict(scs[0])
    for key in ("comps", "mass", "restitution", "friction", "force", "v0"):
        merged[key] = np.concatenate([s[key] for s in scs])
    ow = oracle_world(merged)
    for _ in range(10):
        step_tiles_inprocess(tiles)
        ow.step(float(merged["dt"]), merged["iters"])
    whole = ow.state()
    tiled_v = np.concatenate([t.e.state()["v"] for t in tiles])
    tiled_x = np.concatenate([t.e.state()["x"] for t in tiles])
    dv, dx = rel_err(tiled_v, whole["v"]), rel_err(tiled_x, whole["x"])
    print(f"tiled-vs-undivided deviation after 10 ticks: v {dv:.3e}, x {dx:.3e}")
    assert dx < 0.05 and dv < 1.0


def test_refresh_interval_chunks():
    sc = scenes.sphere_pile_tile(2, 2, 2, 0, 1)
    mk = lambda it, R: Tile(OracleEngine(sc), sc["x_range"], 0, 2, sc["dt"], it, refresh_every=R).chunks()  # noqa: E731
    assert mk(10, 1) == [1] * 10 and mk(10, 2) == [2] * 5 and mk(10, 3) == [3, 3, 3, 1] and mk(10, 10) == [10] and mk(10, 99) == [10]
    assert mk(0, 2) == []


def _seam_penetration(tiles, radius=0.5):
    """mean overlap depth of touching sphere pairs: (pairs across the slab face, pairs inside a tile)"""
    from scipy.spatial import cKDTree
    x = np.concatenate([t.e.state()["x"] for t in tiles]).astype(np.float64)
    owner = np.concatenate([np.full(len(t.e.state()["x"]), r) for r, t in enumerate(tiles)])
    pairs = cKDTree(x).query_pairs(2 * radius, output_type="ndarray")
    pen = 2 * radius - np.linalg.norm(x[pairs[:, 0]] - x[pairs[:, 1]], axis=1)
    cross = owner[pairs[:, 0]] != owner[pairs[:, 1]]
    return float(pen[cross].mean()), float(pen[~cross].mean()), int(cross.sum())


def test_seam_quality_vs_refresh_interval():
    """What the ghost refresh interval costs physically: resting penetration of the sphere pairs that straddle
    the slab face, against pairs inside a tile (both near the solver's 0.05 slop).  The default interval must
    keep the seam at the interior's level; the numbers are printed for DESIGN.md §7."""
    from mgf_amd.tiles import DEFAULT_REFRESH_EVERY
    rows = {}
    for R in sorted({1, DEFAULT_REFRESH_EVERY, 10}):
        tiles = make_tiles(6, 6, 6, 2, refresh_every=R)
        for _ in range(60):
            step_tiles_inprocess(tiles)
        rows[R] = _seam_penetration(tiles)
        print(f"refresh every {R:2d} iterations: seam penetration {rows[R][0]:.4f} ({rows[R][2]} pairs), interior {rows[R][1]:.4f}")
    seam, interior, n = rows[DEFAULT_REFRESH_EVERY]
    assert n >= 10 and seam < 1.25 * interior
    assert rows[1][0] < 1.25 * rows[1][1]


DRIFT = (5.0, 0.0, 0.0)  # 0.083 per tick to the right: whole lattice columns change tile within a few dozen ticks


def _by_tag(tiles, key):
    tags = np.concatenate([t.e.tags() for t in tiles])
    vals = np.concatenate([t.e.state()[key] for t in tiles])
    assert len(np.unique(tags)) == len(tags)
    return vals[np.argsort(tags)]


def test_migration_hands_bodies_to_the_tile_that_holds_them():
    """Bodies drifting to the right change owner when their centre crosses a slab face: nothing is lost or
    duplicated, every body is owned by the tile that holds it (one tick of lag at most), identity (tag)
    and state travel along, and the tiled run keeps tracking the undivided world."""
    P, nx, ny, nz, ticks = 3, 3, 3, 4, 36
    tiles = make_tiles(nx, ny, nz, P, drift=DRIFT)
    scs = [scenes.sphere_pile_tile(nx, ny, nz, r, P, drift=DRIFT) for r in range(P)]
    merged = dict(scs[0])
    for key in ("comps", "mass", "restitution", "friction", "force", "v0"):
        merged[key] = np.concatenate([s[key] for s in scs])
    ow = oracle_world(merged)
    n_total = P * nx * ny * nz
    lag = 0.0
    for tick in range(ticks):
        step_tiles_inprocess(tiles)
        ow.step(float(merged["dt"]), merged["iters"])
        assert sum(len(t.e.w) for t in tiles) == n_total
        for t in tiles:
            x = t.e.state()["x"][:, 0]
            if len(x):
                lo = t.x_lo if t.has_left else -np.inf
                hi = t.x_hi if t.has_right else np.inf
                lag = max(lag, float(np.max(np.maximum(lo - x, x - hi))))
    moved_out = [t.n_migrated_out for t in tiles]
    moved_in = [t.n_migrated_in for t in tiles]
    print(f"migrated out {moved_out} in {moved_in}, bodies per tile {[len(t.e.w) for t in tiles]}, worst lag {lag:.3f}")
    assert sum(moved_out) == sum(moved_in) >= nx * ny  # at least a lattice layer's worth changed owner
    assert moved_in[0] == 0 or moved_out[0] > 0        # the drift is to the right
    assert moved_in[1] > 0 and moved_out[1] > 0        # the middle tile both receives and hands over
    assert lag < 0.25                                  # a body is on the wrong side for at most the tick it crossed in
    assert np.array_equal(np.sort(np.concatenate([t.e.tags() for t in tiles])), np.arange(n_total))
    whole = ow.state()
    dx, dv = rel_err(_by_tag(tiles, "x"), whole["x"]), rel_err(_by_tag(tiles, "v"), whole["v"])
    # A pile hitting the box wall at 5 m/s is chaotic: the tiled and the undivided iteration drift apart smoothly with
    # or without hand-overs (x deviation 0.11 after 40 ticks without drift, 0.34 with); what a lost or duplicated
    # contact would show is bodies sinking into each other, so the overlap depth is the check.
    from scipy.spatial import cKDTree

    def worst_overlap(x):
        x = x.astype(np.float64)
        pairs = cKDTree(x).query_pairs(1.0, output_type="ndarray")
        return float((1.0 - np.linalg.norm(x[pairs[:, 0]] - x[pairs[:, 1]], axis=1)).max())
    ov_t, ov_w = worst_overlap(_by_tag(tiles, "x")), worst_overlap(whole["x"])
    print(f"with migration after {ticks} ticks: deviation x {dx:.3e} v {dv:.3e}; worst sphere overlap tiled {ov_t:.3f} undivided {ov_w:.3f}")
    assert dx < 0.6 and np.isfinite(dv)
    assert ov_t < 1.5 * ov_w + 0.03


def test_migrant_record_round_trip_is_exact():
    """export -> remove -> import on the same world restores every body array bit for bit (the order changes:
    removed bodies are re-appended), and the next ticks equal an untouched copy's, body by body."""
    sc = scenes.sphere_pile_tile(4, 3, 4, 0, 1)
    a, b = oracle_world(sc), oracle_world(sc)
    for w in (a, b):
        w.set_tags(np.arange(len(w), dtype=np.uint32))
        for _ in range(3):
            w.step(float(sc["dt"]), sc["iters"])
    ids = np.array([0, 5, 6, 17, len(a) - 1], np.uint32)
    rec = a.export_migrants(ids)
    a.remove_bodies(ids)
    assert len(a) == len(b) - len(ids)
    a.import_migrants(rec)
    order = np.argsort(a.tags())
    assert np.array_equal(a.tags()[order], b.tags())
    for k, v in b.state().items():
        assert np.array_equal(a.state()[k][order], v), k
    for _ in range(3):
        a.step(float(sc["dt"]), sc["iters"])
        b.step(float(sc["dt"]), sc["iters"])
    # constraint order changed with the body order, so the Gauss-Seidel result differs in the last bits only
    assert rel_err(a.state()["x"][order], b.state()["x"]) < 1e-3


@pytest.mark.timeout(300)
@pytest.mark.parametrize("ws,drift", [(2, None), (3, None), (3, DRIFT)])
def test_gloo_ranks_match_inprocess_tiles(tmp_path, ws, drift):
    """world_size 2 and 3 over gloo (real point-to-point transport; with 3 ranks the middle one talks to both sides)
    == the in-process tile loop, bit for bit - with bodies changing tile in the drift case."""
    nx, ny, nz, ticks = (5, 4, 5, 6) if drift is None else (3, 3, 4, 30)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", PYTHONPATH=ROOT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ws}", "--master-addr", "127.0.0.1",
           "--master-port", str(29515 + ws + (10 if drift else 0)), os.path.join(ROOT, "tests", "tile_worker.py"), str(tmp_path), str(nx), str(ny),
           str(nz), str(ticks), str(drift[0] if drift else 0.0)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    tiles = make_tiles(nx, ny, nz, ws, drift=drift)
    ncons = [[] for _ in range(ws)]
    for _ in range(ticks):
        for k, s in enumerate(step_tiles_inprocess(tiles)):
            ncons[k].append(s["n_constraints"])
    for rank in range(ws):
        got = np.load(tmp_path / f"rank{rank}.npz")
        want = tiles[rank].e.state()
        assert got["ncons"].tolist() == ncons[rank]
        assert np.array_equal(got["tags"], tiles[rank].e.tags())
        for k in want:
            assert np.array_equal(got[k], want[k]), f"rank {rank} {k}"
    if drift is not None:
        assert sum(t.n_migrated_in for t in tiles) > 0
