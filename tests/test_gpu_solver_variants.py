"""Every variant of the block-local solver launch (k_solve_flow6; round 4: records / foreign constants in LDS, four lanes per node,
the hybrid of quad and one-lane trips) against the launch-per-frontier executor (mode 0), bit for bit.  Solver::solve promises the
insertion order (solver.rs:72-78) and ContactConstraint::solve its arithmetic (solver.rs:203-252): whatever the LDS split and the
lanes per node, the velocities, the accumulated impulses and the positions they lead to are those of the sequential loop."""
import numpy as np
import pytest

import mgf_amd
from mgf_amd import scenes

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = mgf_amd.Context(0)
    yield c
    c.close()


VARIANTS = {
    "one lane per node": {"flow6_quad": 0},
    "quad trips only": {"flow6_quad": 1, "flow6_quad_max": 1000000},
    "hybrid, quad while <= 4 ready": {"flow6_quad": 1, "flow6_quad_max": 4},
    "no records in LDS": {"flow6_rec_lds": 0},
    "no foreign constants in LDS": {"flow6_foreign_lds": 0},
    "no constants in LDS": {"flow6_const_lds": 0},
    "records without constants": {"flow6_rec_lds": 2, "flow6_const_lds": 0},
    "no impulses in LDS": {"flow6_nimp_lds": 0, "flow6_rec_lds": 0},
    "quiet sweeps read 4 positions per channel on spec": {"flow6_spec": 4},
    "quiet sweeps read 8 positions on spec, one lane per node": {"flow6_spec": 8, "flow6_quad": 0},
    "one polling wave, the head only on spec": {"flow6_poll_waves": 1, "flow6_spec_wl": 0, "flow6_spec": 1},
    "three polling waves, 8 positions behind the hint": {"flow6_poll_waves": 3, "flow6_spec_wl": 8},
    "quad, nothing optional in LDS": {"flow6_quad_max": 1000000, "flow6_nimp_lds": 0, "flow6_rec_lds": 0, "flow6_const_lds": 0},
}
SCENES = {
    "spheres": lambda: scenes.sphere_pile(12, 10, 12),
    "capsules_and_spheres": lambda: scenes.capsule_field_dense(10, 4, 10, y0=2.0, sphere_fraction=0.3),
    "two_part_bodies": lambda: scenes.dumbbell_field(8, 3, 8, n_plain=20),
}


@pytest.mark.parametrize("scene_name", sorted(SCENES))
@pytest.mark.parametrize("variant", sorted(VARIANTS))
def test_flow6_variants_equal_launch_per_frontier(ctx, scene_name, variant):
    sc = SCENES[scene_name]()
    dt, iters = float(sc["dt"]), sc["iters"]
    a, b = mgf_amd.World.from_scene(ctx, sc), mgf_amd.World.from_scene(ctx, sc)
    a.set_option("solver_mode", 0)
    b.set_option("solver_mode", 6)
    b.set_option("flow5_block", 64)    # several blocks on a small scene: messages cross block faces
    b.set_option("resort_every", 8)    # (worlds this small are not re-sorted on their own)
    for k, v in VARIANTS[variant].items():
        b.set_option(k, v)
    for tick in range(70):
        sa, sb = a.step(dt, iters), b.step(dt, iters)
        assert sa.n_constraints == sb.n_constraints, tick
        if tick % 23 == 22:
            ca, cb = a.constraints(), b.constraints()
            assert np.array_equal(ca["normal_impulse"].view(np.uint32), cb["normal_impulse"].view(np.uint32)), f"tick {tick}: accumulated impulses"
    assert sb.n_constraints > 200
    assert b.counter("flow6_runs") >= 40 and b.counter("flow6_fallbacks") <= 2  # (the first ticks have no constraints: nothing to run)
    sa, sb = a.state(), b.state()
    for k in ("x", "q", "v", "omega"):
        assert np.array_equal(sa[k].view(np.uint32), sb[k].view(np.uint32)), k


def test_flow6_variants_with_odd_iteration_counts(ctx):
    """Launches of 1, 2, 3 and 7 iterations on the same list (a tile makes several per tick): every variant continues from the
    impulses the previous launch left in the records."""
    sc = scenes.sphere_pile(10, 8, 10)
    dt = float(sc["dt"])
    worlds = {}
    for name in ("reference", "one lane per node", "quad trips only", "no records in LDS"):
        w = mgf_amd.World.from_scene(ctx, sc)
        if name == "reference":
            w.set_option("solver_mode", 0)
        else:
            w.set_option("flow5_block", 64)
            for k, v in VARIANTS[name].items():
                w.set_option(k, v)
        worlds[name] = w
    for tick in range(45):
        for w in worlds.values():
            w.build_constraints(dt)
            for it in (1, 2, 3, 7):
                w.solve(it)
    ref = worlds["reference"].state()
    for name, w in worlds.items():
        s = w.state()
        for k in ("x", "v", "omega"):
            assert np.array_equal(ref[k].view(np.uint32), s[k].view(np.uint32)), (name, k)
    assert worlds["reference"].stats.n_constraints > 300
