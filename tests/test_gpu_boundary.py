"""The reference's library-level surface on its own, through the C-ABI, against the oracle:
ConstrainedSet::get / set (physics.rs:272-315), RigidBodyVec::integrate / complete_motion / colliders (physics.rs:222-269),
ContactConstraint::new for caller-built manifolds (solver.rs:101-191), a Solver handle (solver.rs:53-79), RigidBodyVec: Clone
(physics.rs:140).  Everything is compared bit for bit."""
import numpy as np
import pytest

import mgf_amd
from mgf_amd import scenes
from oracle import oracle as O
from tests.util import bits_equal, compare_constraints, oracle_world

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = mgf_amd.Context(0)
    yield c
    c.close()


def _pair(ctx, scene, ticks):
    gw, ow = mgf_amd.World.from_scene(ctx, scene), oracle_world(scene, O.ORDER_CANONICAL)
    dt, iters = float(scene["dt"]), scene["iters"]
    for _ in range(ticks):
        gw.step(dt, iters)
        ow.step(dt, iters)
    return gw, ow, dt


def _same_state(gw, ow, what=""):
    g, o = gw.state(), ow.state()
    for k in ("x", "q", "v", "omega", "delta"):
        assert bits_equal(g[k], o[k]), f"{what}: {k} differs"


def _info_equal(vel, info, want, what):
    got = dict(linear=np.array(vel.linear.tup(), np.float32), angular=np.array(vel.angular.tup(), np.float32),
               x=np.array(info.x.tup(), np.float32), restitution=np.float32(info.restitution), friction=np.float32(info.friction),
               inv_mass=np.float32(info.inv_mass), inv_moment=np.array(list(info.inv_moment), np.float32))
    for k, v in got.items():
        assert bits_equal(np.atleast_1d(v), np.atleast_1d(np.asarray(want[k], np.float32))), f"{what}: {k}: {v} vs {want[k]}"


@pytest.mark.parametrize("scene_fn", [lambda: scenes.sphere_pile(6, 6, 6), lambda: scenes.capsule_field_dense(6, 2, 6)])
def test_constrained_set_get_and_set(ctx, scene_fn):
    """ConstrainedSet::get returns (v, omega) + (x + delta, e, mu, 1/m, world I^-1) (physics.rs:273-288); Static gives
    zeros at its centre (:289-302); set scatters (v, omega) and ignores Static (:306-314)."""
    scene = scene_fn()
    gw, ow, dt = _pair(ctx, scene, 12)
    n = len(gw)
    rng = np.random.default_rng(3)
    for i in [0, n - 1] + rng.integers(0, n, 24).tolist():
        vel, info = gw.get(i)
        _info_equal(vel, info, ow.get(i), f"body {i}")
    vel, info = gw.get(static=((1.5, -10.0, 2.25), 0.375))
    _info_equal(vel, info, ow.get(static=((1.5, -10.0, 2.25), 0.375)), "static")
    with pytest.raises(mgf_amd.MgfError) as e:
        gw.get(n)
    assert e.value.status == 6
    for i in rng.integers(0, n, 8).tolist():
        lin, ang = rng.uniform(-3, 3, 3).astype(np.float32), rng.uniform(-3, 3, 3).astype(np.float32)
        gw.set(i, lin, ang)
        ow.set_velocity(i, lin, ang)
    _same_state(gw, ow, "after set")
    # the written velocities are what the next tick integrates
    gw.step(dt, scene["iters"])
    ow.step(dt, scene["iters"])
    _same_state(gw, ow, "tick after set")


@pytest.mark.parametrize("scene_fn", [lambda: scenes.sphere_pile(6, 6, 6), lambda: scenes.capsule_field_dense(6, 2, 6)])
def test_integrate_complete_motion_and_colliders_alone(ctx, scene_fn):
    """RigidBodyVec::integrate (physics.rs:222-253), ::complete_motion (:262-269) and ::colliders (:256) called one by one,
    the way a caller that owns its own tick would."""
    scene = scene_fn()
    gw, ow, dt = _pair(ctx, scene, 10)
    for rep in range(3):
        gw.complete_motion()
        ow.complete_motion()
        _same_state(gw, ow, f"complete_motion {rep}")
        gw.integrate(dt)
        ow.integrate(dt)
        _same_state(gw, ow, f"integrate {rep}")
        got = gw.colliders()
        comps, delta = ow.colliders()
        assert np.array_equal(got["tag"], comps["tag"])
        for f in ("p", "d", "r"):
            assert bits_equal(got[f], comps[f][:len(got)]), f"collider {f}"
        assert bits_equal(got["delta"], delta[:len(got)])
        # ConstrainedSet::get sees the advanced pose: x + delta of the new collider, the rotated inverse inertia
        i = 5 + rep
        vel, info = gw.get(i)
        _info_equal(vel, info, ow.get(i), f"get after integrate {rep}")
    # and the tick goes on from there identically
    gw.step(dt, scene["iters"])
    ow.step(dt, scene["iters"])
    _same_state(gw, ow, "tick after the hand-made one")


def _random_manifolds(rng, n_bodies, n, scale=0.6):
    m = np.zeros(n, mgf_amd.MANIFOLD_DTYPE)
    refs_a, refs_b = [], []
    for i in range(n):
        a = int(rng.integers(0, n_bodies))
        if i % 5 == 4:
            b = ((float(rng.uniform(-5, 5)), -10.0, float(rng.uniform(-5, 5))), float(rng.uniform(0, 1)))
        else:
            b = int(rng.integers(0, n_bodies - 1))
            b = b + 1 if b >= a else b
        refs_a.append(a)
        refs_b.append(b)
        nrm = rng.normal(size=3).astype(np.float32)
        nrm /= np.float32(np.linalg.norm(nrm))
        if i % 7 == 3:
            nrm *= np.float32(0.8)  # an un-renormalised mean normal, as Manifold::from(pruner) produces (manifold.rs:135-140)
        t = rng.normal(size=(2, 3)).astype(np.float32)  # the caller's tangents are used as given
        k = 1 + int(rng.integers(0, 3))
        m[i]["time"] = rng.uniform(0, 1)
        m[i]["normal"], m[i]["tangent"], m[i]["n_contacts"] = nrm, t, k
        m[i]["local_a"][:k] = rng.uniform(-scale, scale, (k, 3))
        m[i]["local_b"][:k] = rng.uniform(-scale, scale, (k, 3))
    return refs_a, refs_b, m


def _oracle_rows(ow, refs_a, refs_b, m, dt):
    rows = []
    for a, b, mf in zip(refs_a, refs_b, m):
        k = int(mf["n_contacts"])
        kw = dict(static_b=b) if isinstance(b, tuple) else {}
        rows.append(ow.constraint_new(a, None if isinstance(b, tuple) else b, mf["normal"], mf["tangent"], mf["local_a"][:k],
                                      mf["local_b"][:k], dt, **kw))
    return np.concatenate(rows)


def test_contact_constraint_new_for_caller_built_manifolds(ctx):
    """ContactConstraint::new(pool, a, b, manifold, dt) with manifolds the world did not make: dynamic and static obj_b,
    1-3 contacts each (flattened to consecutive rows), unnormalised normals, arbitrary tangents."""
    scene = scenes.sphere_pile(7, 7, 7)
    gw, ow, dt = _pair(ctx, scene, 15)
    rng = np.random.default_rng(11)
    refs_a, refs_b, m = _random_manifolds(rng, len(gw), 400)
    got = gw.constraints_new(refs_a, refs_b, m, dt)
    want = _oracle_rows(ow, refs_a, refs_b, m, dt)
    assert len(got) == int(m["n_contacts"].sum())
    compare_constraints(got, want, check_impulse=True)
    assert (got["b"] == -1).sum() >= 80 and np.abs(got["bias"]).max() > 0
    # the reference panics on a Static obj_a nowhere (it never passes one); here it is a status
    with pytest.raises(mgf_amd.MgfError) as e:
        gw.constraints_new([((0, 0, 0), 0.5)], [1], m[:1], dt)
    assert e.value.status == 4
    with pytest.raises(mgf_amd.MgfError) as e:
        gw.constraints_new([len(gw)], [1], m[:1], dt)
    assert e.value.status == 6
    assert len(gw.constraints_new([], [], m[:0], dt)) == 0


@pytest.mark.parametrize("mode", [6, 5, 1, 0])
def test_solver_handle_runs_a_callers_list_in_insertion_order(ctx, mode):
    """Solver::new / add_constraint / solve(rbv, iters) (solver.rs:59-78) with constraints made by mgf_constraints_new: the
    exact sequential result, state kept between two solve calls, len / clear."""
    scene = scenes.sphere_pile(7, 7, 7)
    gw, ow, dt = _pair(ctx, scene, 15)
    gw.set_option("solver_mode", mode)
    rng = np.random.default_rng(5)
    refs_a, refs_b, m = _random_manifolds(rng, len(gw), 600, scale=0.5)
    rows = gw.constraints_new(refs_a, refs_b, m, dt)
    s = mgf_amd.Solver()
    s.add_constraints(rows[:-1])
    s.add_constraint(rows[-1])
    assert len(s) == len(rows)
    s.solve(gw, 4)
    want = ow.solver_solve(_oracle_rows(ow, refs_a, refs_b, m, dt), 4)
    _same_state(gw, ow, "first solve")
    compare_constraints(s.constraints(), want, check_impulse=True)
    s.solve(gw, 3)  # the accumulated normal impulses carry over (ContactState lives in the Solver)
    want = ow.solver_solve(want, 3)
    _same_state(gw, ow, "second solve")
    compare_constraints(s.constraints(), want, check_impulse=True)
    s.clear()
    assert len(s) == 0
    s.solve(gw, 2)  # an empty Solver leaves the bodies alone
    _same_state(gw, ow, "empty solver")
    # the world goes on ticking with its own lists afterwards
    gw.step(dt, scene["iters"])
    ow.step(dt, scene["iters"])
    _same_state(gw, ow, "tick after the solver calls")


def test_world_clone_is_independent_and_steps_identically(ctx):
    for scene in (scenes.capsule_field_dense(8, 2, 8), scenes.sphere_pile(8, 8, 8)):
        dt, iters = float(scene["dt"]), scene["iters"]
        a = mgf_amd.World.from_scene(ctx, scene)
        for _ in range(8):
            a.step(dt, iters)
        b = a.clone()
        assert len(b) == len(a)
        sa, sb = a.state(), b.state()
        for k in sa:
            assert bits_equal(sa[k], sb[k]), k
        for _ in range(20):
            x, y = a.step(dt, iters), b.step(dt, iters)
            assert (x.n_constraints, x.n_terrain_constraints, x.n_refits) == (y.n_constraints, y.n_terrain_constraints, y.n_refits)
        assert a.stats.n_constraints > 0
        sa, sb = a.state(), b.state()
        for k in sa:
            assert bits_equal(sa[k], sb[k]), f"after 20 ticks: {k}"
        b.set(0, (5, 5, 5), (0, 0, 0))  # writing to the clone leaves the original alone
        assert not bits_equal(a.state()["v"], b.state()["v"])
        assert bits_equal(a.state()["v"], sa["v"])


def test_tick_without_events_and_the_two_read_back_paths(ctx):
    """The tick records no HIP event unless asked (option phase_timing: mgf_step_stats::ms_* read 0 otherwise), and its read-back goes
    by a kernel into pinned memory with a polled sequence word (readback_kernel, default) or by hipMemcpyAsync + an event: same bits,
    through step, step_many (pipelined: two read-backs in flight) and build_constraints / solve."""
    from mgf_amd import scenes
    scene = scenes.sphere_pile(10, 9, 10)
    dt, iters = float(scene["dt"]), scene["iters"]
    a, b = mgf_amd.World.from_scene(ctx, scene), mgf_amd.World.from_scene(ctx, scene)
    b.set_option("readback_kernel", 0)
    b.set_option("phase_timing", 1)
    for _ in range(5):
        sa, sb = a.step(dt, iters), b.step(dt, iters)
    assert sa.ms_total == 0.0 and sa.ms_solve == 0.0 and sa.ms_broadphase == 0.0
    assert sb.ms_total > 0.0 and sb.ms_solve > 0.0 and sb.ms_broadphase > 0.0
    pa, pb = a.step_many(dt, iters, 25), b.step_many(dt, iters, 25)
    assert [int(s.n_constraints) for s in pa] == [int(s.n_constraints) for s in pb] and int(pa[24].n_constraints) > 0
    assert all(float(s.ms_total) == 0.0 for s in pa) and all(float(s.ms_total) > 0.0 for s in pb)
    for w in (a, b):
        w.build_constraints(dt)
        w.solve(iters)
    sa, sb = a.state(), b.state()
    for k in ("x", "q", "v", "omega", "delta"):
        assert np.array_equal(sa[k].view(np.uint32), sb[k].view(np.uint32)), k
    a.set_option("phase_timing", 1)   # switched on in mid-run: figures from the next tick on, nothing read from unrecorded events
    s = a.step(dt, iters)
    assert s.ms_total > 0.0


@pytest.mark.parametrize("resorted", [False, True])
def test_read_and_write_state_take_any_subset_of_the_arrays(ctx, resorted):
    """mgf_world_read_state / write_state pack what was asked for on the device, in the caller's order, and move it through pinned memory in one
    copy: any subset of (x, q, v, omega, delta) - NULL for the rest - must give the same numbers as the full call, on a store in the caller's
    order and on a re-sorted one, and a write of a subset must leave the other arrays (and the records' other words) alone."""
    from mgf_amd._capi import load_library, _check
    sc = scenes.sphere_pile(9, 9, 9)
    w = mgf_amd.World.from_scene(ctx, sc)
    if resorted:
        w.set_option("resort_every", 4)
    dt, it = float(sc["dt"]), sc["iters"]
    for _ in range(9):
        w.step(dt, it)
    if resorted:
        assert w.counter("store_permuted") == 1
    full = w.state()
    n = len(w)
    lib = load_library()
    for keys in (("v",), ("x", "omega"), ("q", "delta"), ("x", "q", "v", "omega", "delta")):
        arr = {k: np.full((n, 4 if k == "q" else 3), np.nan, np.float32) for k in keys}
        p = [arr[k].ctypes.data if k in arr else None for k in ("x", "q", "v", "omega", "delta")]
        _check(lib.mgf_world_read_state(w._h, *p, n))
        for k in keys:
            assert np.array_equal(arr[k], full[k]), (keys, k)
    # a write of v alone: v changes, everything else - and the next ticks against a twin that was written in full - stays
    twin = w.clone()
    v2 = (full["v"] * np.float32(0.5)).astype(np.float32)
    w.write_state(v=v2)
    twin.write_state(x=full["x"], q=full["q"], v=v2, omega=full["omega"], delta=full["delta"])
    a, b = w.state(), twin.state()
    assert np.array_equal(a["v"], v2)
    for k in ("x", "q", "omega", "delta"):
        assert np.array_equal(a[k], full[k]) and np.array_equal(b[k], full[k]), k
    for _ in range(3):
        w.step(dt, it); twin.step(dt, it)
    a, b = w.state(), twin.state()
    for k in ("x", "q", "v", "omega", "delta"):
        assert np.array_equal(a[k], b[k]), k
    cols = w.colliders()
    assert np.array_equal(cols["delta"], a["delta"]) if "delta" in cols.dtype.names else True
