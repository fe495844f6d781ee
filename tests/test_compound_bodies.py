"""Bodies of several components (BASELINE config 5).  The reference has no such body (physics.rs:200 takes one
Component), so the authority is this build's own definition, stated in RigidBodyVec::add_compound_body
(oracle/mgf_physics.hpp) and include/mgf_hip.h.  CPU part: the definition checked against hand-derived values and
against the ordinary-body path it must reduce to.  GPU part: the HIP path against the oracle, bit for bit."""
import numpy as np
import pytest

from oracle import oracle as O
from mgf_amd import scenes
from tests.util import compare_constraints, oracle_world, rel_err, values_equal


def _sphere(c, r):
    k = np.zeros(1, O.COMPONENT_DTYPE)
    k["tag"], k["p"], k["r"] = 0, c, r
    return k


def _world(terrain=True):
    w = O.World(O.ORDER_CANONICAL)
    if terrain:
        t = scenes.box_terrain(8.0, 10.0, (0.0, 0.0, 0.0))
        w.set_terrain(t["verts"], t["faces"], t["pos"])
    return w


def test_dumbbell_mass_centre_and_inertia():
    """Two spheres m = 1 and m = 3, r = 0.5, at x = -1 and x = +1: centre of mass at x = 0.5; about it the tensor is
    sum_k (0.4 m r^2 I + m (|c|^2 I - c c^T)) with c = (-1.5, 0, 0) and (0.5, 0, 0):
    Ixx = 0.4*0.25*(1+3) = 0.4, Iyy = Izz = 0.4 + 1*2.25 + 3*0.25 = 3.4."""
    w = _world(terrain=False)
    comps = np.concatenate([_sphere([-1, 2, 0], 0.5), _sphere([1, 2, 0], 0.5)])
    w.add_compound_bodies(comps, [1.0, 3.0], [0, 2], 0.3, 0.6, [0, -9.8, 0])
    s = w.state()
    assert np.allclose(s["x"][0], [0.5, 2.0, 0.0], atol=1e-6)
    info = w.body_info(0)
    assert abs(info["inv_mass"] - 0.25) < 1e-7
    inv = np.asarray(info["inv_moment"]).reshape(3, 3)
    assert np.allclose(np.diag(inv), [1 / 0.4, 1 / 3.4, 1 / 3.4], rtol=1e-6) and np.allclose(inv - np.diag(np.diag(inv)), 0, atol=1e-7)


def test_one_part_body_equals_the_ordinary_body():
    """A body of ONE sphere is the reference's body: same tensor, same collider, same contacts - the two worlds must stay
    bit-identical through a fall and a collision."""
    a, b = _world(), _world()
    cs = np.concatenate([_sphere([0.0, 1.2, 0.0], 0.5), _sphere([0.3, 2.4, 0.1], 0.5)])
    a.add_bodies(cs, 1.0, 0.3, 0.6, [0, -9.8, 0])
    b.add_compound_bodies(cs, [1.0, 1.0], [0, 1, 2], 0.3, 0.6, [0, -9.8, 0])
    for _ in range(80):
        sa, sb = a.step(1 / 60, 10), b.step(1 / 60, 10)
        assert sa.n_constraints == sb.n_constraints
    assert sa.n_constraints > 0
    for k, v in a.state().items():
        assert np.array_equal(v, b.state()[k]), k


def test_two_part_body_rests_on_both_parts():
    """A dumbbell lying on the floor touches it with both spheres: two terrain constraints on one body, no spin."""
    w = _world()
    comps = np.concatenate([_sphere([-0.6, 0.6, 0], 0.5), _sphere([0.6, 0.6, 0], 0.5)])
    w.add_compound_bodies(comps, [1.0, 1.0], [0, 2], 0.0, 0.6, [0, -9.8, 0])
    for _ in range(240):
        st = w.step(1 / 60, 10)
    s = w.state()
    assert st.n_terrain_constraints == 2 and st.n_constraints == 2
    assert 0.40 < s["x"][0, 1] < 0.52 and abs(s["v"][0]).max() < 0.25 and abs(s["omega"][0]).max() < 1e-3
    cons = w.constraints()
    assert cons["a"].tolist() == [0, 0] and cons["b"].tolist() == [-1, -1]
    assert np.allclose(np.sort(cons["ra"][:, 0]), [-0.6, 0.6], atol=1e-3)  # local points relative to the centre of mass


def test_body_pair_keeps_up_to_four_contacts_in_one_manifold():
    """Two dumbbells stacked crosswise... lying parallel, one on top of the other: both spheres of the upper touch a
    sphere of the lower - one ContactConstraint with two contacts (ContactPruner keeps points farther apart than
    sqrt(0.5)), exported as two consecutive rows with the same normal."""
    w = _world()
    low = np.concatenate([_sphere([-0.6, 0.5, 0], 0.5), _sphere([0.6, 0.5, 0], 0.5)])
    up = np.concatenate([_sphere([-0.6, 1.52, 0], 0.5), _sphere([0.6, 1.52, 0], 0.5)])
    w.add_compound_bodies(np.concatenate([low, up]), 1.0, [0, 2, 4], 0.0, 0.6, [0, -9.8, 0])
    for _ in range(60):
        st = w.step(1 / 60, 10)
    cons = w.constraints()
    pair = cons[(cons["a"] == 1) & (cons["b"] == 0)]
    assert len(pair) == 2 and np.array_equal(pair["normal"][0], pair["normal"][1])
    assert abs(pair["normal"][0][1]) > 0.99 and st.n_constraints == len(cons)


def test_config5_scene_runs_and_settles():
    sc = scenes.dumbbell_field(4, 2, 4, n_plain=6)
    w = oracle_world(sc)
    for _ in range(120):
        st = w.step(float(sc["dt"]), sc["iters"])
    s = w.state()
    assert np.isfinite(s["x"]).all() and s["x"][:, 1].min() > 0.2 and st.n_constraints > 30


def _tiled_setup(engine_factory, P, drift=0.0, jacks=False):
    from mgf_amd.tiles import Tile
    sc = scenes.jack_field(6, 2, 4) if jacks else scenes.dumbbell_field(6, 2, 4, n_plain=8)
    if drift:
        sc["v0"] = (sc["v0"] + np.float32([drift, 0.0, 0.0])).astype(np.float32)
    half = 6 * (2.6 if jacks else 2.2) / 2.0 + 2.0
    # (halo 2: a dumbbell's fat half extent along x reaches ~1.5 - the drivers refuse a halo smaller than that)
    return sc, [Tile(engine_factory(t), t["x_range"], r, P, t["dt"], t["iters"], halo=2.0) for r, t in enumerate(scenes.split_by_slabs(sc, P, half))]


def test_two_part_bodies_across_oracle_tiles():
    """Ghost and migrant records carry the parts: a drifting field of two-part bodies over three tiles keeps every body,
    hands bodies over, and stays close to the undivided world."""
    from mgf_amd.tiles import step_tiles_inprocess
    from tests.oracle_engine import OracleEngine
    sc, tiles = _tiled_setup(OracleEngine, 3, drift=3.0)
    ow = oracle_world(sc)
    n_total = len(ow)
    for _ in range(40):
        stats = step_tiles_inprocess(tiles)
        ow.step(float(sc["dt"]), sc["iters"])
        assert sum(len(t.e.w) for t in tiles) == n_total
    assert sum(t.n_migrated_in for t in tiles) > 0 and sum(s["n_constraints"] for s in stats) > 20
    tags = np.concatenate([t.e.tags() for t in tiles])
    assert np.array_equal(np.sort(tags), np.arange(n_total))
    x = np.concatenate([t.e.state()["x"] for t in tiles])[np.argsort(tags)]
    assert rel_err(x, ow.state()["x"]) < 0.2 and np.isfinite(x).all()


# ---- HIP path -------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def ctx():
    import mgf_amd
    c = mgf_amd.Context(0)
    yield c
    c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n_plain,mode", [(0, 6), (6, 6), (6, 5), (6, 0), (6, 1)])
def test_hip_dumbbell_field_equals_oracle(ctx, n_plain, mode):
    import mgf_amd
    sc = scenes.dumbbell_field(4, 2, 4, n_plain=n_plain)
    dt, iters = float(sc["dt"]), sc["iters"]
    gw, ow = mgf_amd.World.from_scene(ctx, sc), oracle_world(sc)
    gw.set_option("solver_mode", mode)
    multi = 0
    for tick in range(100):
        sg, so = gw.step(dt, iters), ow.step(dt, iters)
        assert (sg.n_constraints, sg.n_terrain_constraints, sg.n_pair_candidates) == (so.n_constraints, so.n_terrain_constraints, so.n_pair_candidates), f"tick {tick}"
        if tick % 2 == 1:  # (manifolds of several contacts come and go within a few ticks in this small scene)
            got, want = gw.constraints(), ow.constraints()
            compare_constraints(got, want, check_impulse=True)
            ab = list(zip(want["a"].tolist(), want["b"].tolist()))
            multi += sum(1 for k in range(1, len(ab)) if ab[k] == ab[k - 1] and ab[k][1] >= 0)
    assert so.n_constraints > 30 and multi > 0  # manifolds of several contacts occurred
    g, o = gw.state(), ow.state()
    for k in ("x", "q", "v", "omega", "delta"):
        assert values_equal(g[k], o[k]), f"{k}: rel err {rel_err(g[k], o[k])}"


@pytest.mark.gpu
def test_hip_larger_config5_scene(ctx):
    """8 x 3 x 8 two-part bodies + 30 spheres, 60 ticks: counts and state equal to the oracle's."""
    import mgf_amd
    sc = scenes.dumbbell_field(8, 3, 8, n_plain=30)
    dt, iters = float(sc["dt"]), sc["iters"]
    gw, ow = mgf_amd.World.from_scene(ctx, sc), oracle_world(sc)
    for tick in range(60):
        sg, so = gw.step(dt, iters), ow.step(dt, iters)
        assert sg.n_constraints == so.n_constraints and sg.n_pair_candidates == so.n_pair_candidates, f"tick {tick}"
    g, o = gw.state(), ow.state()
    for k in ("x", "q", "v", "omega"):
        assert values_equal(g[k], o[k]), f"{k}: rel err {rel_err(g[k], o[k])}"


@pytest.mark.gpu
@pytest.mark.parametrize("P,drift,jacks", [(2, 0.0, False), (3, 3.0, False), (2, 0.0, True), (3, 3.0, True)])
def test_hip_two_part_bodies_across_tiles_equal_oracle_tiles(ctx, P, drift, jacks):
    """The tile protocol with bodies of several parts (ghost records with parts, hand-overs with the part arrays, kind
    masks telling a tile that its neighbours hold such bodies): every tile bit-identical to the oracle's.  jacks: bodies of FOUR
    parts (r04: the tile records carry four part slots)."""
    from mgf_amd.tiles import HipEngine, step_tiles_inprocess
    from tests.oracle_engine import OracleEngine
    _, gt = _tiled_setup(lambda t: HipEngine(ctx, t, 0), P, drift, jacks)
    _, ot = _tiled_setup(OracleEngine, P, drift, jacks)
    for tick in range(40):
        sg, so = step_tiles_inprocess(gt), step_tiles_inprocess(ot)
        for r in range(P):
            assert sg[r]["n_constraints"] == so[r]["n_constraints"], f"tick {tick} tile {r}"
        assert [t.n_migrated_in for t in gt] == [t.n_migrated_in for t in ot]
    if drift:
        assert sum(t.n_migrated_in for t in gt) > 0
    assert sum(s["n_constraints"] for s in sg) > 20
    for r in range(P):
        assert np.array_equal(gt[r].e.tags(), ot[r].e.tags())
        g, o = gt[r].e.state(), ot[r].e.state()
        for k in ("x", "q", "v", "omega", "delta"):
            assert values_equal(g[k], o[k]), f"tile {r} {k}: rel err {rel_err(g[k], o[k])}"


@pytest.mark.gpu
def test_hip_bodies_of_more_than_thirty_two_parts_are_refused(ctx):
    import mgf_amd
    sc = scenes.dumbbell_field(6, 1, 3)
    gw = mgf_amd.World.from_scene(ctx, sc)
    gw.add_compound_bodies(sc["compound"]["comps"][:3], 1.0, [0, 3], 0.3, 0.6, [0, -9.8, 0])      # three parts: fine since round 3
    gw.add_compound_bodies(sc["compound"]["comps"][:5], 1.0, [0, 5], 0.3, 0.6, [0, -9.8, 0])      # five: fine since round 6 (tests/test_gpu_many_part_bodies.py)
    n = len(gw)
    with pytest.raises(mgf_amd.MgfError):
        gw.add_compound_bodies(sc["compound"]["comps"][:33], 1.0, [0, 33], 0.3, 0.6, [0, -9.8, 0])  # thirty-three: over the limit
    assert len(gw) == n  # (nothing was added)


# ---- bodies of FOUR components (round 3) ---------------------------------------------------------------------------------------
def test_oracle_four_part_body_mass_properties():
    """A jack: hub sphere + three orthogonal capsules through it.  Mass = sum; centre of mass = the hub (symmetric); the inertia
    tensor is the sum of the parts' own tensors about it, and - three equal capsules along an orthonormal frame - isotropic."""
    sc = scenes.jack_field(1, 1, 1)
    ow = oracle_world(sc)
    info = ow.body_info(0)
    assert abs(1.0 / info["inv_mass"] - 2.2) < 1e-6
    x = ow.state()["x"][0]
    assert np.allclose(x, sc["compound"]["comps"]["p"][0], atol=1e-6)
    im = np.asarray(ow.inv_moment()[0], np.float64).reshape(3, 3)
    assert np.allclose(im, im.T, atol=1e-6) and np.allclose(im, np.eye(3) * im[0, 0], atol=2e-4 * abs(im[0, 0]))


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [6, 5, 1, 0])
def test_hip_four_part_bodies_equal_oracle(ctx, mode):
    import mgf_amd
    sc = scenes.jack_field(4, 2, 4)
    dt, iters = float(sc["dt"]), sc["iters"]
    gw, ow = mgf_amd.World.from_scene(ctx, sc), oracle_world(sc)
    gw.set_option("solver_mode", mode)
    gw.set_option("resort_every", 3)
    longest = 0
    for tick in range(120):
        sg, so = gw.step(dt, iters), ow.step(dt, iters)
        assert (sg.n_constraints, sg.n_terrain_constraints, sg.n_pair_candidates) == (so.n_constraints, so.n_terrain_constraints, so.n_pair_candidates), f"tick {tick}"
        if tick % 3 == 2:
            got, want = gw.constraints(), ow.constraints()
            compare_constraints(got, want, check_impulse=True)
            ab = list(zip(want["a"].tolist(), want["b"].tolist()))
            run = 1
            for k in range(1, len(ab)):
                run = run + 1 if (ab[k] == ab[k - 1] and ab[k][1] >= 0) else 1
                longest = max(longest, run)
    assert so.n_constraints > 40 and longest >= 3  # manifolds of three and more contacts occurred (a two-part pair yields at most four part pairs)
    g, o = gw.state(), ow.state()
    for k in ("x", "q", "v", "omega", "delta"):
        assert values_equal(g[k], o[k]), f"{k}: rel err {rel_err(g[k], o[k])}"


@pytest.mark.gpu
def test_hip_worlds_mixing_one_two_and_four_parts(ctx):
    """Ordinary spheres, two-part and four-part bodies in one world (the four-part kernels then serve every pair), through a re-sorted
    store and a clone; the tile protocol takes such a world too (r04: four part slots in its records)."""
    import mgf_amd
    a, b = scenes.jack_field(3, 2, 3), scenes.dumbbell_field(3, 1, 3, n_plain=8)
    b["compound"]["comps"]["p"][:, 1] += 7.0
    b["comps"]["p"][:, 1] += 9.0
    gw, ow = mgf_amd.World.from_scene(ctx, b), oracle_world(b)
    ca = a["compound"]
    gw.add_compound_bodies(ca["comps"], ca["comp_mass"], ca["offsets"], ca["restitution"], ca["friction"], ca["force"])
    ow.add_compound_bodies(ca["comps"], ca["comp_mass"], ca["offsets"], ca["restitution"], ca["friction"], ca["force"])
    gw.set_option("resort_every", 2)
    dt, iters = float(a["dt"]), a["iters"]
    for tick in range(90):
        sg, so = gw.step(dt, iters), ow.step(dt, iters)
        assert sg.n_constraints == so.n_constraints and sg.n_pair_candidates == so.n_pair_candidates, f"tick {tick}"
        if tick == 40:
            gw = gw.clone()
    compare_constraints(gw.constraints(), ow.constraints(), check_impulse=True)
    g, o = gw.state(), ow.state()
    for k in ("x", "q", "v", "omega"):
        assert values_equal(g[k], o[k]), k
    T = mgf_amd.Tiles(ctx, [gw], [(-1e30, 1e30)], halo=2.5)  # one tile: steps like the world itself
    sg, so = T.step(dt, iters), ow.step(dt, iters)
    assert int(sg[0].n_constraints) == so.n_constraints
    g, o = gw.state(), ow.state()
    for k in ("x", "q", "v", "omega"):
        assert values_equal(g[k], o[k]), k


@pytest.mark.gpu
def test_hip_65536_four_part_bodies_match_the_oracle(ctx):
    """VERDICT r2 item 7's size: 65 536 bodies of four components.  The first tick and a later, contact-rich one (the oracle started
    from the GPU's state) bit for bit: constraint list in insertion order, counts, post-step state."""
    import mgf_amd
    sc = scenes.jack_field(64, 16, 64)
    dt, iters = float(sc["dt"]), sc["iters"]
    gw, ow = mgf_amd.World.from_scene(ctx, sc), oracle_world(sc)
    assert len(gw) == 65536
    sg, so = gw.step(dt, iters), ow.step(dt, iters)
    assert (sg.n_constraints, sg.n_pair_candidates) == (so.n_constraints, so.n_pair_candidates)
    gw.step_many(dt, iters, 100)
    s = gw.state()
    ow.set_state(x=s["x"], q=s["q"], v=s["v"], omega=s["omega"], delta=s["delta"])
    sg, so = gw.step(dt, iters), ow.step(dt, iters)
    # (the accepted-partner statistic depends on the persistent fat boxes, which are not part of the state handed over)
    assert sg.n_constraints == so.n_constraints > 50000
    compare_constraints(gw.constraints(), ow.constraints(), check_impulse=True)
    g, o = gw.state(), ow.state()
    for k in ("x", "q", "v", "omega", "delta"):
        assert values_equal(g[k], o[k]), k
    print(f"65 536 four-part bodies: {sg.n_constraints} constraints")
