"""Dynamic bodies of MORE than four components (round 6): up to 32 parts per body, kept in a pool beside the
bodies' four part slots (Bodies::xl0, k_bodies.h), a wave per candidate pair of bodies over its part pairs (k_narrow_pairs_big) and per
(body, face) over its parts (k_narrow_terrain_big).  The reference's Compound (compound.rs:232-352) is a static shape; a dynamic body of many
components is this build's definition, stated by the oracle (RigidBodyVec::add_compound_body; World::collide: every pair of parts in order
through ContactPruner) - the HIP path against it, bit for bit."""
import numpy as np
import pytest

from mgf_amd import scenes
from tests.util import compare_constraints, oracle_world, rel_err, values_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import mgf_amd
    c = mgf_amd.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("mode", [6, 1])
def test_sixteen_part_bodies_equal_the_oracle(ctx, mode):
    """caterpillars of sixteen components, a few bodies of three among them (the four-slot kind) and ordinary spheres on top: every tick's counts,
    every third tick's constraint list in insertion order with the solved impulses, through a re-sorted store and a clone; the state at the end"""
    import mgf_amd
    sc = scenes.caterpillar_field(4, 2, 4, n_plain=8, small_every=5)
    dt, iters = float(sc["dt"]), sc["iters"]
    gw, ow = mgf_amd.World.from_scene(ctx, sc), oracle_world(sc)
    gw.set_option("solver_mode", mode)
    gw.set_option("resort_every", 3)
    longest = terrain = 0
    for tick in range(150):
        sg, so = gw.step(dt, iters), ow.step(dt, iters)
        assert (sg.n_constraints, sg.n_terrain_constraints, sg.n_pair_candidates) == (so.n_constraints, so.n_terrain_constraints, so.n_pair_candidates), f"tick {tick}"
        terrain = max(terrain, int(sg.n_terrain_constraints))
        if tick % 3 == 2:
            got, want = gw.constraints(), ow.constraints()
            compare_constraints(got, want, check_impulse=True)
            ab = list(zip(want["a"].tolist(), want["b"].tolist()))
            run = 1
            for k in range(1, len(ab)):
                run = run + 1 if (ab[k] == ab[k - 1] and ab[k][1] >= 0) else 1
                longest = max(longest, run)
        if tick == 70:
            gw = gw.clone()
    assert so.n_constraints > 60 and terrain > 40 and longest >= 3, (so.n_constraints, terrain, longest)
    g, o = gw.state(), ow.state()
    for k in ("x", "q", "v", "omega", "delta"):
        assert values_equal(g[k], o[k]), f"{k}: rel err {rel_err(g[k], o[k])}"


def test_a_field_of_a_thousand_sixteen_part_bodies(ctx):
    """1 024 bodies of sixteen components (16 384 parts): the first tick and a later, contact-rich one - the oracle started from the GPU's state -
    bit for bit, and mgf_world_step_many against single steps"""
    import mgf_amd
    sc = scenes.caterpillar_field(16, 4, 16)
    dt, iters = float(sc["dt"]), sc["iters"]
    gw, ow = mgf_amd.World.from_scene(ctx, sc), oracle_world(sc)
    g2 = mgf_amd.World.from_scene(ctx, sc)
    sg, so = gw.step(dt, iters), ow.step(dt, iters)
    assert (sg.n_constraints, sg.n_pair_candidates) == (so.n_constraints, so.n_pair_candidates)
    gw.step_many(dt, iters, 120)
    for _ in range(121):
        g2.step(dt, iters)
    a, b = gw.state(), g2.state()
    for k in ("x", "q", "v", "omega"):
        assert np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)), k
    ow.set_state(x=a["x"], q=a["q"], v=a["v"], omega=a["omega"], delta=a["delta"])
    sg, so = gw.step(dt, iters), ow.step(dt, iters)
    assert sg.n_constraints == so.n_constraints > 3000, (sg.n_constraints, so.n_constraints)
    compare_constraints(gw.constraints(), ow.constraints(), check_impulse=True)
    g, o = gw.state(), ow.state()
    for k in ("x", "q", "v", "omega", "delta"):
        assert values_equal(g[k], o[k]), k


def test_limits_of_many_part_bodies(ctx):
    """what is not built is refused: such bodies in a tile set (beside static obstacles they run: tests/test_gpu_obstacles.py); two bodies that meet in
    more part pairs than the wave's list holds fail the tick with MGF_ERR_CAPACITY"""
    import mgf_amd
    sc = scenes.caterpillar_field(2, 1, 2)
    gw = mgf_amd.World.from_scene(ctx, sc)
    with pytest.raises(mgf_amd.MgfError):
        mgf_amd.Tiles(ctx, [gw], [(-1e30, 1e30)], halo=2.5)
    # thirty-two spheres of r = 0.4 in a 4 x 4 x 2 block, 0.5 apart, and the same block again shifted by a quarter of a radius: every
    # part of one touches several of the other - hundreds of part pairs in contact
    i, j, k = np.meshgrid(np.arange(4), np.arange(4), np.arange(2), indexing="ij")
    p = np.stack([i.ravel() * 0.5, 3.0 + j.ravel() * 0.5, k.ravel() * 0.5], axis=1).astype(np.float32)
    comps = np.zeros(64, dtype=sc["compound"]["comps"].dtype)
    comps["p"][:32] = p
    comps["p"][32:] = p + np.float32([0.1, 0.05, 0.1])
    comps["r"] = 0.4
    w2 = mgf_amd.World.from_scene(ctx, scenes.dumbbell_field(1, 1, 1))
    w2.add_compound_bodies(comps, 1.0, [0, 32, 64], 0.3, 0.6, [0, -9.8, 0])
    with pytest.raises(mgf_amd.MgfError) as e:
        w2.step(1.0 / 60.0, 10)
    assert "part pairs" in str(e.value)


def test_random_clumps_pressed_into_each_other(ctx):
    """the pruner under load: clumps of 5..32 random spheres and capsules, a few of them thrown into one another so that pairs of bodies meet in
    dozens of part pairs at several collision times - contacts merged, replaced and dropped by ContactPruner::push in the order the wave packed them
    (k_narrow_pairs_big) - one tick each, constraint lists and states against the oracle; a trial that exceeds the lists' capacity must say so"""
    import mgf_amd
    rng = np.random.default_rng(616)
    dtype = scenes.COMPONENT_DTYPE
    full = over = manifolds = 0
    for trial in range(80):
        nb = int(rng.integers(2, 5))
        comps, offsets, masses = [], [0], []
        centres = rng.normal(0.0, 0.9, (nb, 3)) + np.array([0.0, 6.0, 0.0])
        for b in range(nb):
            np_ = int(rng.integers(5, 33))
            k = np.zeros(np_, dtype)
            k["tag"] = (rng.random(np_) < 0.4).astype(np.int32)
            k["p"] = (centres[b] + rng.normal(0.0, 0.8, (np_, 3))).astype(np.float32)
            k["d"] = (rng.normal(0.0, 0.5, (np_, 3)) * (k["tag"][:, None] == 1)).astype(np.float32)
            k["r"] = rng.uniform(0.08, 0.3, np_).astype(np.float32)
            comps.append(k); offsets.append(offsets[-1] + np_); masses.append(rng.uniform(0.2, 1.0, np_).astype(np.float32))
        sc = scenes._scene(f"clumps_{trial}", np.zeros(0, dtype), scenes.box_terrain(12.0, 14.0, (0.0, 0.0, 0.0)))
        sc["compound"] = dict(comps=np.concatenate(comps), comp_mass=np.concatenate(masses), offsets=np.asarray(offsets, np.int64),
                              restitution=np.full(nb, 0.3, np.float32), friction=np.full(nb, 0.6, np.float32), force=np.tile(np.float32([0.0, -9.8, 0.0]), (nb, 1)))
        sc["v0"] = rng.normal(0.0, 3.0, (nb, 3)).astype(np.float32)
        gw, ow = mgf_amd.World.from_scene(ctx, sc), oracle_world(sc)
        dt, iters = float(sc["dt"]), sc["iters"]
        try:
            sg = gw.step(dt, iters)
        except mgf_amd.MgfError as e:
            assert "part pairs" in str(e)   # (more than 64 raw contacts or a manifold above 16: refused, not mangled)
            over += 1
            continue
        so = ow.step(dt, iters)
        assert (sg.n_constraints, sg.n_pair_candidates) == (so.n_constraints, so.n_pair_candidates), trial
        got, want = gw.constraints(), ow.constraints()
        compare_constraints(got, want, check_impulse=True)
        g, o = gw.state(), ow.state()
        for key in ("x", "q", "v", "omega"):
            assert values_equal(g[key], o[key]), (trial, key)
        full += 1
        ab = list(zip(want["a"].tolist(), want["b"].tolist()))
        manifolds += sum(1 for i in range(1, len(ab)) if ab[i] == ab[i - 1] and ab[i][1] >= 0)
    assert full >= 40 and manifolds > 100, (full, over, manifolds)


def test_a_sixteen_part_body_that_runs_away(ctx):
    """the wide-body list (WideSpec, k_pair_wide) knows boxes, not kinds: a caterpillar thrown through the field at 600 m/s - its fat box three times
    the others', which are five metres long themselves - and one falling far below the scene, against the oracle and against the same world with the list switched off"""
    import mgf_amd
    sc = scenes.caterpillar_field(4, 2, 4, n_plain=6)
    n_plain = len(sc["comps"])
    sc["v0"][n_plain + 3] = np.float32([600.0, 40.0, -200.0])     # (through the field: ten metres a tick)
    sc["v0"][n_plain + 7] = np.float32([0.0, -700.0, 0.0])
    cb = sc["compound"]
    lo, hi = int(cb["offsets"][7]), int(cb["offsets"][8])
    cb["comps"]["p"][lo:hi, 1] -= np.float32(1500.0)               # (far below the floor: it never comes back)
    dt, iters = float(sc["dt"]), sc["iters"]
    a, b, ow = mgf_amd.World.from_scene(ctx, sc), mgf_amd.World.from_scene(ctx, sc), oracle_world(sc)
    b.set_option("wide_list", 0)
    for tick in range(80):
        sa, sb, so = a.step(dt, iters), b.step(dt, iters), ow.step(dt, iters)
        assert (sa.n_constraints, sa.n_pair_candidates) == (so.n_constraints, so.n_pair_candidates) == (sb.n_constraints, sb.n_pair_candidates), tick
        if tick % 8 == 7:
            compare_constraints(a.constraints(), ow.constraints(), check_impulse=True)
    assert a.counter("wide_ticks") > 20 and b.counter("wide_ticks") == 0
    g, h, o = a.state(), b.state(), ow.state()
    for k in ("x", "q", "v", "omega"):
        assert values_equal(g[k], o[k]) and values_equal(h[k], o[k]), k
