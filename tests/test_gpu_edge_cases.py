"""Edge cases of the tick, GPU vs oracle, bit for bit: empty and tiny worlds, no terrain, zero iterations, coincident
centres, a zero-length capsule, a fast body (time-of-impact branch), a body far outside the others' Morton range,
resting exactly on the floor."""
import numpy as np
import pytest

from mgf_amd import scenes
from tests.util import bits_equal, compare_constraints, oracle_world

pytestmark = pytest.mark.gpu


def _scene(centres, r=0.5, caps=None, terrain="box", v0=None, iters=10):
    comps = scenes._spheres(np.asarray(centres, np.float32), r)
    if caps is not None:  # rows (index, d3)
        for i, d in caps:
            comps["tag"][i] = 1
            comps["d"][i] = d
    t = scenes.box_terrain(12.0, 12.0, (0.0, 0.0, 0.0)) if terrain == "box" else None
    return scenes._scene("edge", comps, t, v0=v0, iters=iters)


def _run(ctx, scene, ticks, iters=None, expect_constraints=None):
    import mgf_amd
    dt = float(scene["dt"])
    iters = scene["iters"] if iters is None else iters
    ow, gw = oracle_world(scene), mgf_amd.World.from_scene(ctx, scene)
    seen = 0
    for step in range(ticks):
        so, sg = ow.step(dt, iters), gw.step(dt, iters)
        assert sg.n_constraints == so.n_constraints, f"step {step}: {sg.n_constraints} vs {so.n_constraints}"
        compare_constraints(gw.constraints(), ow.constraints(), check_impulse=True)
        seen += so.n_constraints
        g, o = gw.state(), ow.state()
        for k in ("x", "q", "v", "omega", "delta"):
            assert bits_equal(g[k], o[k]) or np.array_equal(np.float32(g[k]), np.float32(o[k])), f"step {step}: {k}"
    if expect_constraints is not None:
        assert (seen > 0) == expect_constraints
    return gw, ow


@pytest.fixture(scope="module")
def ctx():
    import mgf_amd
    c = mgf_amd.Context(0)
    yield c
    c.close()


def test_empty_world_steps(ctx):
    import mgf_amd
    w = mgf_amd.World(ctx)
    for _ in range(3):
        st = w.step(1.0 / 60.0, 10)
        assert st.n_constraints == 0 and st.n_bodies == 0
    assert len(w) == 0 and len(w.constraints()) == 0


def test_single_body_with_and_without_terrain(ctx):
    _run(ctx, _scene([(0, 0.45, 0)]), 30, expect_constraints=True)      # one sphere resting on the floor
    _run(ctx, _scene([(0, 5.0, 0)], terrain=None), 30, expect_constraints=False)  # free fall, nothing to hit


def test_two_bodies_no_terrain_head_on(ctx):
    _run(ctx, _scene([(-0.7, 0, 0), (0.7, 0, 0)], terrain=None, v0=[(3, 0, 0), (-3, 0, 0)]), 20, expect_constraints=True)


def test_zero_iterations_builds_constraints_but_leaves_velocities(ctx):
    gw, ow = _run(ctx, scenes.sphere_pile(5, 5, 5), 4, iters=0, expect_constraints=True)
    assert float(np.abs(gw.constraints()["normal_impulse"]).max()) == 0.0


def test_coincident_centres_take_the_reference_fallback_normal(ctx):
    # identical centres: collision.rs:1098-1106 picks -v/|v| (or NaN for v = 0); whatever it is, both sides agree bitwise
    _run(ctx, _scene([(0, 3, 0), (0, 3, 0), (0.2, 3, 0)], terrain=None, v0=[(1, 0, 0), (0, 0, 0), (0, 0, 0)]), 3, expect_constraints=True)


def test_zero_length_capsule_is_rejected_like_the_reference_panics(ctx):
    # Capsule with d = 0: from_arc of a zero axis makes the inertia tensor singular, `.invert().unwrap()` panics (physics.rs:212)
    import mgf_amd
    sc = _scene([(0, 0.6, 0), (1.5, 0.6, 0)], caps=[(1, (0, 0, 0))])
    with pytest.raises(ValueError):
        oracle_world(sc)
    with pytest.raises(mgf_amd.MgfError) as e:
        mgf_amd.World.from_scene(ctx, sc)
    assert e.value.status == 5  # MGF_ERR_SINGULAR
    _run(ctx, _scene([(0, 0.6, 0), (1.5, 0.6, 0), (0.7, 1.3, 0)], caps=[(1, (0, 1e-3, 0)), (2, (0.8, 0, 0))]), 25, expect_constraints=True)


def test_fast_body_hits_through_the_time_of_impact_branch(ctx):
    # 40 m/s * dt = 0.67 per tick: the sphere starts clear of the floor and the resting one, contact at 0 < t <= 1
    _run(ctx, _scene([(0, 0.5, 0), (0.3, 2.2, 0)], v0=[(0, 0, 0), (0, -40.0, 0)]), 6, expect_constraints=True)


def test_outlier_body_stretches_the_morton_range(ctx):
    # one body 500 units away: the other 216 share a handful of Morton cells; the grid broadphase must still be exact
    sc = scenes.sphere_pile(6, 6, 6)
    sc["comps"]["p"][0] = (500.0, 3.0, 0.0)
    gw, _ = _run(ctx, sc, 12, expect_constraints=True)
    assert gw.counter("row_overflows") >= 0


def test_mixed_radii_fall_back_to_the_tree_walk(ctx):
    # a sphere 12x larger than the rest spans many cells: the tick switches to the tree walk and stays exact
    rng = np.random.default_rng(4)
    c = rng.uniform(-5, 5, (700, 3)).astype(np.float32)  # enough bodies for a 1024-cell grid
    c[:, 1] = rng.uniform(0.5, 6, 700)
    sc = _scene(np.concatenate([c, [[0, 9.0, 0]]]))
    sc["comps"]["r"][-1] = 6.0
    gw, _ = _run(ctx, sc, 10, expect_constraints=True)
    assert gw.counter("grid_too_wide") == 1


def test_resting_exactly_at_contact_distance(ctx):
    # |dist| == r takes the resting branch (collision.rs:525): centre exactly one radius above the floor
    _run(ctx, _scene([(0, 0.5, 0), (2.0, 0.5, 0)]), 10, expect_constraints=True)


def test_solve_called_twice_and_set_constraints_empty(ctx):
    import mgf_amd
    sc = scenes.sphere_pile(6, 6, 6)
    ow, gw = oracle_world(sc), mgf_amd.World.from_scene(ctx, sc)
    dt = float(sc["dt"])
    ow.build_constraints(dt); gw.build_constraints(dt)
    for _ in range(2):  # Solver::solve twice on the same list: impulses keep accumulating
        ow.solve(3); gw.solve(3)
    compare_constraints(gw.constraints(), ow.constraints(), check_impulse=True)
    assert bits_equal(gw.state()["v"], ow.state()["v"])
    gw.set_constraints(gw.constraints()[:0])
    v = gw.state()["v"].copy()
    gw.solve(5)
    assert bits_equal(gw.state()["v"], v)


@pytest.mark.gpu
def test_device_scan_around_tile_boundaries():
    """The scan behind every list offset of the tick, on its own: sizes around tile and vector boundaries, many blocks,
    repeated calls, wrap-around sums."""
    import mgf_amd
    ctx = mgf_amd.Context(0)
    rng = np.random.default_rng(5)
    for n in (1, 2, 3, 4, 5, 1023, 1024, 1025, 4093, 4095, 4096, 4097, 8191, 8192, 8193, 262145, 1048577, 3000001):
        a = rng.integers(0, 50, n, dtype=np.uint32)
        want = np.concatenate([[0], np.cumsum(a[:-1], dtype=np.uint64)]).astype(np.uint32)
        for _ in range(3):
            assert np.array_equal(ctx.exclusive_scan(a), want), n
    big = np.full(70000, 0x00010001, np.uint32)  # sums wrap modulo 2^32 like the host's
    assert np.array_equal(ctx.exclusive_scan(big), (np.arange(70000, dtype=np.uint64) * 0x00010001).astype(np.uint32))
    ctx.close()
