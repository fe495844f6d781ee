"""BASELINE.json's full-size configurations on the GPU: a one-tick comparison with the oracle (the
oracle needs a few seconds per tick at this size) and size-independent properties of the tick:
determinism, constraint-list invariants, schedule coverage, finite state, momentum bookkeeping."""
import numpy as np
import pytest

from oracle import oracle as O
from tests.util import compare_constraints, oracle_world, rel_err, values_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import mgf_amd
    c = mgf_amd.Context(0)
    yield c
    c.close()


def _scene(name):
    from mgf_amd import scenes
    return scenes.sphere_pile(64, 64, 64) if name == "config2_spheres_262144" else scenes.capsule_field(128, 32, 32, quads=158, y0=0.9)


@pytest.mark.parametrize("name", ["config2_spheres_262144", "config3_capsules_131072_tris_49928"])
def test_full_size_tick_matches_oracle_and_properties(ctx, name):
    import mgf_amd
    scene = _scene(name)
    dt, iters = float(scene["dt"]), scene["iters"]
    n = len(scene["comps"])
    assert n == (262144 if "spheres" in name else 131072)
    if "capsules" in name:
        assert len(scene["terrain"]["faces"]) == 49928
    gw = mgf_amd.World.from_scene(ctx, scene)
    gw2 = mgf_amd.World.from_scene(ctx, scene)
    ow = oracle_world(scene)
    # --- tick 1 against the oracle: constraint list and post-step state
    ow.build_constraints(dt)
    st = gw.build_constraints(dt)
    oc, gc = ow.constraints(), gw.constraints()
    compare_constraints(gc, oc)
    ow.solve(iters)
    gw.solve(iters)
    g, o = gw.state(), ow.state()
    for k in ("x", "q", "v", "omega", "delta"):
        assert rel_err(g[k], o[k]) <= 1e-4, k
        assert values_equal(g[k], o[k]), f"{k} not bit-identical"
    # --- constraint-list invariants (size independent)
    a, b = gc["a"], gc["b"]
    assert (a >= 0).all() and (a < n).all() and (b < a).all()          # partner j < i (world.rs:266); static = -1
    assert np.all(np.diff(a) >= 0)                                      # insertion order: body i ascending
    same = a[1:] == a[:-1]
    bt, bn = b[:-1][same], b[1:][same]
    assert np.all((bt == -1) | (bn != -1))                              # terrain constraints precede partners
    both = (bt >= 0) & (bn >= 0)
    assert np.all(bn[both] > bt[both])                                  # partners ascending, no duplicates
    nrm = np.linalg.norm(gc["normal"].astype(np.float64), axis=1)
    assert np.allclose(nrm, 1.0, atol=1e-5)
    assert (gc["normal_impulse"] >= 0).all() and np.isfinite(gc["normal_mass"]).all()
    # body-body impulses are equal and opposite: the pair constraints cannot change total momentum.
    # (unit masses) total change of momentum = gravity + terrain constraints only
    # --- determinism: an independent world built from the same arrays gives the same bits
    for _ in range(3):
        s1 = gw2.step(dt, iters)
    for _ in range(2):
        gw.step(dt, iters)
    a1, a2 = gw.state(), gw2.state()
    for k in a1:
        assert np.array_equal(a1[k].view(np.uint32), a2[k].view(np.uint32)), f"non-deterministic {k}"
        assert np.isfinite(a1[k]).all()
    assert s1.n_constraints == gw.stats.n_constraints
    # --- a later, contact-rich tick: advance the GPU, teacher-force the oracle from its state, compare one tick
    for _ in range(40 if "capsules" in name else 15):
        gw.step(dt, iters)
    s = gw.state()
    ow.set_state(x=s["x"], q=s["q"], v=s["v"], omega=s["omega"], delta=s["delta"])
    ow.build_constraints(dt)
    st2 = gw.build_constraints(dt)
    compare_constraints(gw.constraints(), ow.constraints())
    ow.solve(iters)
    gw.solve(iters)
    g, o = gw.state(), ow.state()
    for k in ("x", "q", "v", "omega", "delta"):
        assert values_equal(g[k], o[k]), f"later tick: {k} not bit-identical (rel err {rel_err(g[k], o[k])})"
    print(f"{name}: later tick {st2.n_constraints} constraints ({st2.n_terrain_constraints} terrain)")
    print(f"{name}: {st.n_constraints} constraints, {gw.stats.n_levels} solver launches, "
          f"{st.n_pair_candidates} pair candidates, {st.n_terrain_candidates} terrain candidates")


def test_config2_full_size_in_the_references_own_order(ctx):
    """262 144 spheres with option constraint_order = demo: the host replays world.rs:233-291 (world BVH refit inside the
    loop, partners in BVH::query order) and the device builds and solves the constraints in THAT order - the first ticks
    bit-identical to the oracle run in world.rs order, i.e. comparable with the reference as the reference runs."""
    import mgf_amd
    from mgf_amd import scenes
    scene = scenes.sphere_pile(64, 64, 64)
    dt, iters = float(scene["dt"]), scene["iters"]
    gw, ow = mgf_amd.World.from_scene(ctx, scene), oracle_world(scene, order=O.ORDER_DEMO)
    gw.set_option("constraint_order", 1)
    for tick in range(2):
        ow.build_constraints(dt)
        st = gw.build_constraints(dt)
        got, want = gw.constraints(), ow.constraints()
        compare_constraints(got, want)
        if tick == 0:
            a, b = want["a"], want["b"]
            same = (a[1:] == a[:-1]) & (b[1:] >= 0) & (b[:-1] >= 0)
            assert np.any(b[1:][same] < b[:-1][same])  # the order really is not the canonical one (partners not ascending)
        ow.solve(iters)
        gw.solve(iters)
        g, o = gw.state(), ow.state()
        for k in ("x", "q", "v", "omega", "delta"):
            assert values_equal(g[k], o[k]), f"tick {tick}: {k} not bit-identical (rel err {rel_err(g[k], o[k])})"
    assert st.n_constraints > 400000


def test_config5_full_size_two_part_bodies(ctx):
    """BASELINE config 5 at size: 65 536 bodies of two components (sphere + capsule).  The reference has no such body (the
    definition is this build's); the HIP path is held to the oracle's statement of it: the first tick and
    a later, contact-rich tick (oracle teacher-forced from the GPU's state) bit for bit - constraint rows of the
    multi-contact manifolds included."""
    import mgf_amd
    from mgf_amd import scenes
    scene = scenes.dumbbell_field(64, 16, 64)
    dt, iters = float(scene["dt"]), scene["iters"]
    gw, ow = mgf_amd.World.from_scene(ctx, scene), oracle_world(scene)
    assert len(gw) == 65536 == len(ow)
    sg, so = gw.step(dt, iters), ow.step(dt, iters)
    assert (sg.n_constraints, sg.n_terrain_constraints, sg.n_pair_candidates) == (so.n_constraints, so.n_terrain_constraints, so.n_pair_candidates)
    g, o = gw.state(), ow.state()
    for k in ("x", "q", "v", "omega", "delta"):
        assert values_equal(g[k], o[k]), f"first tick: {k}"
    for _ in range(110):
        gw.step(dt, iters)
    s = gw.state()
    ow.set_state(x=s["x"], q=s["q"], v=s["v"], omega=s["omega"], delta=s["delta"])
    ow.build_constraints(dt)
    st = gw.build_constraints(dt)
    got, want = gw.constraints(), ow.constraints()
    compare_constraints(got, want)
    ab = np.stack([want["a"], want["b"]], axis=1)
    multi = int(np.sum((ab[1:] == ab[:-1]).all(axis=1) & (ab[1:, 1] >= 0)))
    assert st.n_constraints > 30000 and multi > 100, (st.n_constraints, multi)  # manifolds of several contacts are there
    ow.solve(iters)
    gw.solve(iters)
    g, o = gw.state(), ow.state()
    for k in ("x", "q", "v", "omega", "delta"):
        assert values_equal(g[k], o[k]), f"later tick: {k} (rel err {rel_err(g[k], o[k])})"
    print(f"config 5: {st.n_constraints} constraints ({st.n_terrain_constraints} terrain), {multi} rows continue a multi-contact manifold")


def test_pair_constraints_conserve_momentum(ctx):
    """Solve only the body-body constraints of a large pile (no gravity step, no terrain): total linear
    momentum is unchanged up to f32 accumulation (every impulse is applied +/- to the two bodies)."""
    import mgf_amd
    from mgf_amd import scenes
    scene = scenes.sphere_pile(40, 40, 40)
    scene = dict(scene, terrain=None)
    gw = mgf_amd.World.from_scene(ctx, scene)
    gw.build_constraints(float(scene["dt"]))
    assert gw.stats.n_constraints > 50000 and gw.stats.n_terrain_constraints == 0
    p0 = gw.state()["v"].astype(np.float64).sum(axis=0)
    gw.solve(10)
    p1 = gw.state()["v"].astype(np.float64).sum(axis=0)
    assert np.abs(p1 - p0).max() < 1e-2 * np.sqrt(len(gw)), (p0, p1)


@pytest.mark.parametrize("mode", [1, 4, 5, 6])
def test_dataflow_solver_full_size_stress(ctx, mode):
    """262 144 spheres, 40 ticks: the persistent dataflow solver and the launch-per-frontier solver must
    agree bit for bit (any stale cross-CU read would change bits somewhere in ~200 M constraint solves)."""
    import mgf_amd
    from mgf_amd import scenes
    scene = scenes.sphere_pile(64, 64, 64)
    dt, iters = float(scene["dt"]), scene["iters"]
    a, b = mgf_amd.World.from_scene(ctx, scene), mgf_amd.World.from_scene(ctx, scene)
    a.set_option("solver_mode", 0)
    b.set_option("solver_mode", mode)
    for w in (a, b):
        w.set_option("phase_timing", 1)  # (the solve phase's span, printed below)
    ms_a = ms_b = 0.0
    for step in range(40):
        sa, sb = a.step(dt, iters), b.step(dt, iters)
        ms_a += sa.ms_solve
        ms_b += sb.ms_solve
        assert sa.n_constraints == sb.n_constraints
        if step % 10 == 9:
            s1, s2 = a.state(), b.state()
            for k in ("v", "omega", "x"):
                assert np.array_equal(s1[k].view(np.uint32), s2[k].view(np.uint32)), f"step {step}: {k} differs"
    print(f"solve phase: launches {ms_a / 40:.3f} ms/tick, dataflow mode {mode} {ms_b / 40:.3f} ms/tick")


def test_dataflow_solver_settled_pile_at_full_size(ctx):
    """The settled pile (tick 260 on: ~1 M constraints, long chains, the LDS plan of mode 6 at its narrow margins) - the regime
    the 40-tick stress above never reaches.  Modes 1 (global dataflow) and 6 (block-local with channels, the default) carried
    there by mgf_world_step_many and then stepped side by side: bit-identical, and mode 6 really ran its own kernel."""
    import mgf_amd
    from mgf_amd import scenes
    scene = scenes.sphere_pile(64, 64, 64)
    dt, iters = float(scene["dt"]), scene["iters"]
    a, b = mgf_amd.World.from_scene(ctx, scene), mgf_amd.World.from_scene(ctx, scene)
    a.set_option("solver_mode", 1)
    b.set_option("solver_mode", 6)
    a.step_many(dt, iters, 260)
    b.step_many(dt, iters, 260)
    fb0 = b.counter("flow6_fallbacks")
    for step in range(24):
        sa, sb = a.step(dt, iters), b.step(dt, iters)
        assert sa.n_constraints == sb.n_constraints
    assert sb.n_constraints > 900000
    s1, s2 = a.state(), b.state()
    for k in ("x", "q", "v", "omega", "delta"):
        assert np.array_equal(s1[k].view(np.uint32), s2[k].view(np.uint32)), f"{k} differs"
    assert b.counter("flow6_fallbacks") - fb0 <= 2, (fb0, b.counter("flow6_fallbacks"), b.counter("flow6_fail_reason"))
    print(f"settled: {sb.n_constraints} constraints; mode 6 fallbacks {b.counter('flow6_fallbacks')} (reason {b.counter('flow6_fail_reason')}), "
          f"nimp in LDS on {b.counter('flow6_nimp_lds')} ticks, max slots {b.counter('flow6_max_slots')}")


def test_config2_front_ends_agree_over_a_hundred_ticks(ctx):
    """Round 5's front end of a world of spheres - Morton cells inside k_integrate over the previous tick's bounds, near-mesh records and
    the sphere-triangle tests by four lanes in the scatter's launch, constraint records straight from the rows (no candidate lists) - against
    the candidate-list path it replaced (options fused_contacts = 0, cells_in_integrate = 0), 262 144 spheres, ticks 0..100 and 400..420
    of the same pile: every count and every bit of the state."""
    import mgf_amd
    from mgf_amd import scenes
    scene = scenes.sphere_pile(64, 64, 64)
    dt, iters = float(scene["dt"]), scene["iters"]
    a, b = mgf_amd.World.from_scene(ctx, scene), mgf_amd.World.from_scene(ctx, scene)
    b.set_option("fused_contacts", 0)
    b.set_option("cells_in_integrate", 0)
    done = 0
    for ticks in (100, 300, 20):
        sa, sb = a.step_many(dt, iters, ticks), b.step_many(dt, iters, ticks)
        done += ticks
        for key in ("n_constraints", "n_terrain_constraints", "n_terrain_candidates", "n_pair_candidates", "n_refits"):
            assert [int(s[key]) for s in sa] == [int(s[key]) for s in sb], (done, key)
        xa, xb = a.state(), b.state()
        for k in ("x", "q", "v", "omega", "delta"):
            assert np.array_equal(xa[k].view(np.uint32), xb[k].view(np.uint32)), (done, k)
    assert int(sa[-1]["n_terrain_constraints"]) > 4000 and int(sa[-1]["n_constraints"]) > 900000
