"""Bodies far larger - by their motion - than the rest (r06: WideSpec in k_bodies.h, k_pair_wide): a sphere that left the scene and falls at
220 m/s sweeps 3.7 m per tick, and the largest fat half extent of the scene is the reach of every query of the cell grid.  Such bodies are
kept out of the scene bounds and paired by a launch of their own: same constraints, same states - against the oracle, against the same world
with the list switched off - and the world's tick costs what it costs without them."""
import time

import numpy as np
import pytest

from tests.util import compare_constraints, oracle_world, values_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import mgf_amd
    c = mgf_amd.Context(0)
    yield c
    c.close()


def _same(x, y):
    return all(np.array_equal(x[k].view(np.uint32), y[k].view(np.uint32)) for k in ("x", "q", "v", "omega"))


def _with_runaways(scene, where):
    """the scene with some of its spheres far away and fast: [(body, position, velocity)]"""
    sc = dict(scene)
    comps, v0 = scene["comps"].copy(), scene["v0"].copy()
    for body, pos, vel in where:
        comps["p"][body] = np.float32(pos)
        v0[body] = np.float32(vel)
    sc["comps"], sc["v0"] = comps, v0
    return sc


@pytest.mark.parametrize("kind", ["spheres", "capsules", "two_part_bodies"])
def test_runaway_bodies_change_nothing_but_the_bounds(ctx, kind):
    import mgf_amd
    from mgf_amd import scenes
    # (two_part_bodies: plain spheres run away among bodies of two components - a tick with wide bodies there takes the list-based kernels instead
    # of the r06 front end, whose pair search writes manifolds k_pair_wide does not)
    base = (scenes.sphere_pile(16, 16, 16) if kind == "spheres" else scenes.capsule_field(8, 6, 8, quads=12, pitch=1.6) if kind == "capsules"
            else scenes.dumbbell_field(8, 5, 8, n_plain=40))
    n = len(base["comps"])
    # one far below and fast, one INSIDE the pile and fast (a wide body that meets others: partner and query), one fast pair side by side
    sc = _with_runaways(base, [(n - 1, (0.0, -2500.0, 0.0), (0.0, -220.0, 0.0)), (n // 2, (1.0, 6.0, 0.5), (150.0, 20.0, -90.0)),
                               (5, (40.0, 30.0, 0.0), (-200.0, 0.0, 0.0)), (9, (40.0, 30.8, 0.3), (-200.0, 0.0, 0.0))])
    dt, it = float(sc["dt"]), sc["iters"]
    a, b = mgf_amd.World.from_scene(ctx, sc), mgf_amd.World.from_scene(ctx, sc)
    b.set_option("wide_list", 0)
    ow = oracle_world(sc)
    for s in range(90):
        if s % 15 == 3:  # against the oracle from the same state
            st = a.state()
            ow.set_state(x=st["x"], q=st["q"], v=st["v"], omega=st["omega"], delta=st["delta"])
            ow.build_constraints(dt)
            sa, sb = a.build_constraints(dt), b.build_constraints(dt)
            compare_constraints(a.constraints(), ow.constraints())
            assert sa.n_pair_candidates == sb.n_pair_candidates, (s, sa.n_pair_candidates, sb.n_pair_candidates)
            ow.solve(it); a.solve(it); b.solve(it)
            g, o = a.state(), ow.state()
            for k in ("x", "q", "v", "omega"):
                assert values_equal(g[k], o[k]), (s, k)
        else:
            sa, sb = a.step(dt, it), b.step(dt, it)
        assert sa.n_constraints == sb.n_constraints and sa.n_pair_candidates == sb.n_pair_candidates, (s, sa.n_constraints, sb.n_constraints)
        assert _same(a.state(), b.state()), s
    assert a.counter("wide_ticks") > 60 and 1 <= a.counter("wide_bodies") <= 4 and a.counter("wide_overflows") == 0
    assert b.counter("wide_ticks") == 0


def test_many_fast_bodies_are_no_outliers(ctx):
    """a hundred bodies become fast at once: more than the list holds - the tick is run again without it, the reference is learnt anew"""
    import mgf_amd
    from mgf_amd import scenes
    base = scenes.sphere_pile(12, 12, 12)
    n = len(base["comps"])
    dt, it = float(base["dt"]), base["iters"]
    a, b = mgf_amd.World.from_scene(ctx, base), mgf_amd.World.from_scene(ctx, base)
    b.set_option("wide_list", 0)
    for _ in range(10):
        a.step(dt, it); b.step(dt, it)
    st = a.state()
    v = st["v"].copy()
    v[:100] = np.float32([120.0, 0.0, 0.0])
    for w in (a, b):
        w.write_state(x=st["x"], q=st["q"], v=v, omega=st["omega"])
    for s in range(30):
        sa, sb = a.step(dt, it), b.step(dt, it)
        assert sa.n_constraints == sb.n_constraints and _same(a.state(), b.state()), s
    assert a.counter("wide_overflows") >= 1


def test_a_runaway_does_not_slow_the_world_down(ctx):
    """one sphere at y = -2 500 falling at 220 m/s - within 1.1 x of the same world without it (it was 4 x)"""
    import mgf_amd
    from mgf_amd import scenes
    base = scenes.sphere_pile(48, 48, 48)
    n = len(base["comps"])
    sc = _with_runaways(base, [(n - 1, (0.0, -2500.0, 0.0), (0.0, -220.0, 0.0))])
    dt, it = float(base["dt"]), base["iters"]

    def ms_per_tick(scene, opts):
        w = mgf_amd.World.from_scene(ctx, scene)
        for k, v in opts.items():
            w.set_option(k, v)
        w.step_many(dt, it, 40)
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); w.step_many(dt, it, 40); best = min(best, (time.perf_counter() - t0) / 40)
        return best * 1e3, w
    t_plain, _ = ms_per_tick(base, {})
    t_run, w = ms_per_tick(sc, {})
    t_off, _ = ms_per_tick(sc, {"wide_list": 0})
    print(f"ms per tick: without the runaway {t_plain:.3f}, with it {t_run:.3f}, with it and the list off {t_off:.3f}")
    assert w.counter("wide_ticks") > 100 and w.counter("wide_bodies") == 1
    assert t_run <= 1.1 * t_plain + 0.02, (t_plain, t_run, t_off)
