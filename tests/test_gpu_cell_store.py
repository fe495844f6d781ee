"""The body store in an internal (cell) order (mgf_amd/csrc/host_perm.inc): the fused tick re-sorts the RigidBodyVec into the order
of its cell sort every few ticks; slots change, the Gauss-Seidel sequence - decided by the CALLER's body indices - does not.

Held here: a world that re-sorts every tick / every few ticks / never produces the same bits (state, constraint list in insertion
order, counts), equal to the oracle's; the boundary maps indices both ways (state, get / set, colliders, tags, constraints);
entry points that name bodies by index put the caller's order back; clones, added bodies, every solver mode and caller-set lists
work on a store that has been re-sorted."""
import numpy as np
import pytest

from oracle import oracle as O
from tests.util import compare_constraints, oracle_world, values_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import mgf_amd
    c = mgf_amd.Context(0)
    yield c
    c.close()


def _world(ctx, scene, resort, **opts):
    import mgf_amd
    w = mgf_amd.World.from_scene(ctx, scene)
    w.set_option("resort_every", resort)
    for k, v in opts.items():
        w.set_option(k, v)
    return w


def _same_state(a, b, what):
    for k in ("x", "q", "v", "omega", "delta"):
        assert np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)), f"{what}: {k} differs"


def _scenes():
    from mgf_amd import scenes
    return {
        "spheres": scenes.sphere_pile(12, 10, 12),
        "capsules_on_heightfield": scenes.capsule_field_dense(10, 4, 10, quads=12, y0=0.9),
        "mixed": scenes.capsule_field_dense(8, 4, 8, quads=10, y0=0.9, sphere_fraction=0.5),
        "two_part_bodies": scenes.dumbbell_field(6, 3, 6, n_plain=20),
    }


@pytest.mark.parametrize("name", ["spheres", "capsules_on_heightfield", "mixed", "two_part_bodies"])
def test_resorted_store_steps_bit_identically_and_matches_the_oracle(ctx, name):
    scene = _scenes()[name]
    dt, iters = float(scene["dt"]), scene["iters"]
    ref = _world(ctx, scene, 0)
    every1 = _world(ctx, scene, 1)
    every3 = _world(ctx, scene, 3)
    ow = oracle_world(scene)
    ticks = 36
    for t in range(ticks):
        for w in (ref, every1, every3):
            w.step(dt, iters)
        if t < 6 or t % 6 == 5:
            ow.step(dt, iters)
            so = ow.state()
            a, b, c = ref.state(), every1.state(), every3.state()
            _same_state(a, b, f"tick {t}, resort every tick")
            _same_state(a, c, f"tick {t}, resort every 3 ticks")
            for k in ("x", "q", "v", "omega"):
                assert values_equal(a[k], so[k]), f"tick {t}: {k} differs from the oracle"
            ca, cb, cc = ref.constraints(), every1.constraints(), every3.constraints()
            compare_constraints(cb, ca, check_impulse=True)
            compare_constraints(cc, ca, check_impulse=True)
            compare_constraints(ca, ow.constraints(), check_impulse=True)
            assert ref.stats.n_pair_candidates == every1.stats.n_pair_candidates == every3.stats.n_pair_candidates
        else:
            ow.step(dt, iters)
    assert ref.counter("store_permuted") == 0 and ref.counter("store_resorts") == 0
    assert every1.counter("store_permuted") == 1 and every1.counter("store_resorts") >= ticks - 2
    assert 8 <= every3.counter("store_resorts") <= 14


def test_step_many_with_a_stale_order(ctx):
    """Many ticks between two re-sorts: bodies drift out of the store's order, the tick's own cell sort still runs."""
    from mgf_amd import scenes
    scene = scenes.sphere_pile(14, 12, 14)
    dt, iters = float(scene["dt"]), scene["iters"]
    ref = _world(ctx, scene, 0)
    w = _world(ctx, scene, 25)
    for _ in range(3):
        ref.step_many(dt, iters, 30)
        w.step_many(dt, iters, 30)
        _same_state(ref.state(), w.state(), "step_many")
        compare_constraints(w.constraints(), ref.constraints(), check_impulse=True)
    assert 3 <= w.counter("store_resorts") <= 5


def test_boundary_maps_indices(ctx):
    from mgf_amd import scenes
    scene = scenes.capsule_field_dense(8, 3, 8, quads=10, y0=0.9, sphere_fraction=0.3)
    dt, iters = float(scene["dt"]), scene["iters"]
    n = len(scene["comps"])
    ref = _world(ctx, scene, 0)
    w = _world(ctx, scene, 1)
    tags = (np.arange(n, dtype=np.uint32) * 7 + 3).astype(np.uint32)
    ref.set_tags(tags)
    w.set_tags(tags)
    for _ in range(5):
        ref.step(dt, iters)
        w.step(dt, iters)
    assert w.counter("store_permuted") == 1
    assert np.array_equal(w.tags(), tags)                      # read through the map
    assert ref.colliders().tobytes() == w.colliders().tobytes()
    for i in (0, 1, n // 2, n - 1):
        va, ia = ref.get(i)
        vb, ib = w.get(i)
        assert bytes(va) == bytes(vb) and bytes(ia) == bytes(ib), f"get({i})"
    # set through the map, then step on: both worlds take the same kick on the same body
    for wd in (ref, w):
        wd.set(n // 3, (0.5, 2.0, -0.25), (0.1, 0.0, 0.3))
    for _ in range(4):
        ref.step(dt, iters)
        w.step(dt, iters)
    _same_state(ref.state(), w.state(), "after set")
    # write_state names bodies by the caller's indices and leaves the store in its internal order: rows go through slot_of
    s = w.state()
    v2 = s["v"].copy(); v2[::7] += np.float32(0.25)
    x2 = s["x"].copy(); x2[::11, 1] += np.float32(0.01)
    om2 = s["omega"].copy(); om2[::5, 2] -= np.float32(0.125)
    w.write_state(x=x2, v=v2, omega=om2)
    ref.write_state(x=x2, v=v2, omega=om2)
    assert w.counter("store_permuted") == 1
    _same_state(ref.state(), w.state(), "right after write_state")
    tags2 = (np.arange(n, dtype=np.uint32)[::-1] * 5 + 1).astype(np.uint32)   # ... and so does set_tags
    w.set_tags(tags2)
    assert w.counter("store_permuted") == 1 and np.array_equal(w.tags(), tags2)
    for _ in range(3):
        ref.step(dt, iters)
        w.step(dt, iters)
    assert w.counter("store_permuted") == 1
    _same_state(ref.state(), w.state(), "after write_state")


def test_library_boundary_on_a_resorted_store(ctx):
    """build_constraints / solve (the reference's own call sequence) run on whatever order the store is in; a caller's list
    (mgf_world_set_constraints) and a Solver handle name bodies by the caller's indices."""
    import mgf_amd
    from mgf_amd import scenes
    scene = scenes.sphere_pile(10, 8, 10)
    dt, iters = float(scene["dt"]), scene["iters"]
    ref = _world(ctx, scene, 0)
    w = _world(ctx, scene, 2)
    for _ in range(12):
        ref.step(dt, iters)
        w.step(dt, iters)
    assert w.counter("store_permuted") == 1
    for wd in (ref, w):
        wd.build_constraints(dt)
    assert w.counter("store_permuted") == 1  # (no index crossed the boundary)
    compare_constraints(w.constraints(), ref.constraints())
    lst = ref.constraints()
    for wd in (ref, w):
        wd.solve(iters)
    _same_state(ref.state(), w.state(), "build + solve")
    compare_constraints(w.constraints(), ref.constraints(), check_impulse=True)
    # the same list handed back by the caller: solved again on top, in both worlds
    for wd in (ref, w):
        wd.set_constraints(lst)
        wd.solve(3)
    assert w.counter("store_permuted") == 1  # (the list's body names are translated, the store stays as it is)
    _same_state(ref.state(), w.state(), "caller's list")
    compare_constraints(w.constraints(), ref.constraints(), check_impulse=True)
    sv = mgf_amd.Solver()
    sv.add_constraints(lst)
    for wd in (ref, w):
        wd.step(dt, iters)
        wd.step(dt, iters)
    assert w.counter("store_permuted") == 1
    sa, sb = mgf_amd.Solver(), mgf_amd.Solver()
    sa.add_constraints(lst)
    sb.add_constraints(lst)
    sa.solve(ref, 2)
    sb.solve(w, 2)
    _same_state(ref.state(), w.state(), "Solver handle")


@pytest.mark.parametrize("mode", [0, 1, 4, 5, 6])
def test_every_solver_mode_on_a_resorted_store(ctx, mode):
    from mgf_amd import scenes
    scene = scenes.sphere_pile(10, 10, 10)
    dt, iters = float(scene["dt"]), scene["iters"]
    ref = _world(ctx, scene, 0, solver_mode=1)
    w = _world(ctx, scene, 1, flow5_block=128)
    for _ in range(10):
        ref.step(dt, iters)
        w.step(dt, iters)
    assert w.counter("store_permuted") == 1
    w.set_option("solver_mode", mode)
    for t in range(8):
        ref.step(dt, iters)
        w.step(dt, iters)
        _same_state(ref.state(), w.state(), f"mode {mode}, tick {t}")
    if mode == 6:
        assert w.counter("flow6_fallbacks") == 0


def test_clone_and_added_bodies(ctx):
    from mgf_amd import scenes
    scene = scenes.sphere_pile(10, 8, 10)
    extra = scenes.sphere_pile(4, 2, 4, seed=99)
    extra["comps"]["p"][:, 1] += 14.0
    dt, iters = float(scene["dt"]), scene["iters"]
    ref = _world(ctx, scene, 0)
    w = _world(ctx, scene, 2)
    for _ in range(9):
        ref.step(dt, iters)
        w.step(dt, iters)
    c = w.clone()
    assert c.counter("store_permuted") == 1
    for _ in range(7):
        ref.step(dt, iters)
        w.step(dt, iters)
        c.step(dt, iters)
    _same_state(ref.state(), w.state(), "original")
    _same_state(ref.state(), c.state(), "clone")
    compare_constraints(c.constraints(), ref.constraints(), check_impulse=True)
    for wd in (ref, w):
        wd.add_bodies(extra["comps"], extra["mass"], extra["restitution"], extra["friction"], extra["force"])
    assert w.counter("store_permuted") == 0 and len(w) == len(scene["comps"]) + len(extra["comps"])
    for _ in range(8):
        ref.step(dt, iters)
        w.step(dt, iters)
    assert w.counter("store_permuted") == 1
    _same_state(ref.state(), w.state(), "after add_bodies")


def test_demo_order_keeps_the_callers_order(ctx):
    from mgf_amd import scenes
    scene = scenes.balls_demo(6)
    dt, iters = float(scene["dt"]), scene["iters"]
    w = _world(ctx, scene, 1)
    ow = oracle_world(scene, O.ORDER_DEMO)
    for _ in range(4):
        w.step(dt, iters)
    assert w.counter("store_permuted") == 1
    s = w.state()
    ow.set_state(x=s["x"], q=s["q"], v=s["v"], omega=s["omega"], delta=s["delta"])
    w.set_option("constraint_order", 1)
    assert w.counter("store_permuted") == 0
    for t in range(40):
        w.step(dt, iters)
        ow.step(dt, iters)
    a, b = w.state(), ow.state()
    for k in ("x", "q", "v", "omega"):
        assert values_equal(a[k], b[k]), k


def test_other_iteration_counts_in_the_fused_tick(ctx):
    """a fused tick (one synchronisation) whose iteration count differs from the one the block-local solver's
    channels were laid out for - the first tick with iters != 10, and iters raised between ticks - on a dense pile: a layout
    that does not fit re-runs the tick with the global solver; nothing may run on unbuilt links or be solved twice."""
    from mgf_amd import scenes
    scene = scenes.sphere_pile(24, 20, 24)
    dt = float(scene["dt"])
    ref = _world(ctx, scene, 0, solver_mode=1)
    w = _world(ctx, scene, -1)
    plan = [4] * 3 + [16] * 3 + [40] * 2 + [3] * 2 + [10] * 2
    for it in plan:
        ref.step(dt, it)
    k = 0
    while k < len(plan):  # the same plan in runs of equal counts through step_many (the pipelined tick)
        m = 1
        while k + m < len(plan) and plan[k + m] == plan[k]:
            m += 1
        w.step_many(dt, plan[k], m)
        k += m
    _same_state(ref.state(), w.state(), "changing iteration counts")
    # and through the synchronous path: the collide phase finished, then solves of different lengths
    ref.build_constraints(dt)
    w.build_constraints(dt)
    for it in (2, 7, 30):
        ref.solve(it)
        w.solve(it)
        _same_state(ref.state(), w.state(), f"solve({it}) after a finished collide phase")
    compare_constraints(w.constraints(), ref.constraints(), check_impulse=True)
