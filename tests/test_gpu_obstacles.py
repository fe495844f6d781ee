"""Static Compounds as obstacles of the world beside the Mesh (round 3): every body's parts
against Compound::contacts (compound.rs:334-352) each tick, every contact a constraint against a Static body at the compound's
displacement.  The oracle states the definition (World::obstacles); the HIP path is held to it bit for bit: constraint list in
insertion order, counts, state - ordinary spheres and capsules, bodies of several parts, a world without a Mesh, a re-sorted store."""
import numpy as np
import pytest

import mgf_amd
from mgf_amd import scenes
from tests.util import compare_constraints, oracle_world, values_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = mgf_amd.Context(0)
    yield c
    c.close()


def _obstacles():
    """two compounds: a rotated ramp of three capsules with a ball on its end, and a ring of spheres"""
    a = np.zeros(4, scenes.COMPONENT_DTYPE)
    a["tag"] = [1, 1, 1, 0]
    a["p"] = [(-3.0, 0.4, -1.0), (-3.0, 0.4, 0.0), (-3.0, 0.4, 1.0), (3.2, 0.9, 0.0)]
    a["d"] = [(6.0, 1.0, 0.0), (6.0, 1.0, 0.0), (6.0, 1.0, 0.0), (0, 0, 0)]
    a["r"] = [0.35, 0.35, 0.35, 0.8]
    k = 10
    ang = np.linspace(0.0, 2.0 * np.pi, k, endpoint=False)
    b = np.zeros(k, scenes.COMPONENT_DTYPE)
    b["tag"] = 0
    b["p"] = np.stack([2.5 * np.cos(ang), np.full(k, 0.5), 2.5 * np.sin(ang)], axis=1)
    b["r"] = 0.55
    half = np.float32(np.sin(0.2)), np.float32(np.cos(0.2))
    return [(a, (0.4, 0.2, -0.3), (float(half[1]), 0.0, float(half[0]), 0.0)), (b, (-1.0, 0.0, 1.5), (1.0, 0.0, 0.0, 0.0))]


def _add(ctx, gw, ow):
    for comps, disp, rot in _obstacles():
        c = mgf_amd.Compound(ctx, comps)
        c.set_pose(disp, rot)
        gw.add_obstacle(c)
        ow.add_obstacle(comps, disp, rot)


SCENES = {
    "spheres": lambda: scenes.sphere_pile(8, 5, 8),
    "capsules_and_spheres": lambda: scenes.capsule_field_dense(7, 3, 7, y0=2.5, sphere_fraction=0.4),
    "bodies_of_several_parts": lambda: scenes.jack_field(4, 2, 4, y0=3.0),
    "bodies_of_sixteen_parts": lambda: scenes.caterpillar_field(3, 2, 3, small_every=4, y0=3.0),  # (r06: parts from the pool, five bits of part in an obstacle candidate)
}


@pytest.mark.parametrize("name", sorted(SCENES))
@pytest.mark.parametrize("mode", [6, 0])
def test_world_with_obstacles_equals_oracle(ctx, name, mode):
    sc = SCENES[name]()
    dt, iters = float(sc["dt"]), sc["iters"]
    gw, ow = mgf_amd.World.from_scene(ctx, sc), oracle_world(sc)
    _add(ctx, gw, ow)
    gw.set_option("solver_mode", mode)
    gw.set_option("resort_every", 4)
    seen = 0
    for tick in range(100):
        sg, so = gw.step(dt, iters), ow.step(dt, iters)
        assert (sg.n_constraints, sg.n_terrain_constraints, sg.n_pair_candidates) == (so.n_constraints, so.n_terrain_constraints, so.n_pair_candidates), f"tick {tick}"
        if tick % 10 == 9:
            compare_constraints(gw.constraints(), ow.constraints(), check_impulse=True)
    got = gw.constraints()
    seen = int((got["b"] < 0).sum())
    g, o = gw.state(), ow.state()
    for k in ("x", "q", "v", "omega", "delta"):
        assert values_equal(g[k], o[k]), k
    assert seen > 10  # constraints against static bodies exist (terrain and obstacles)


def test_obstacles_without_a_mesh_and_through_a_clone(ctx):
    """No terrain at all: bodies fall onto the compounds and past them; the clone carries the obstacles."""
    sc = scenes.sphere_pile(7, 4, 7)
    sc = dict(sc)
    sc["terrain"] = None
    dt, iters = float(sc["dt"]), sc["iters"]
    from oracle import oracle as O
    ow = O.World(O.ORDER_CANONICAL)
    ow.add_bodies(sc["comps"], sc["mass"], sc["restitution"], sc["friction"], sc["force"])
    ow.set_state(v=sc["v0"])
    gw = mgf_amd.World(ctx)
    gw.add_bodies(sc["comps"], sc["mass"], sc["restitution"], sc["friction"], sc["force"])
    gw.write_state(v=sc["v0"])
    _add(ctx, gw, ow)
    hits = 0
    for tick in range(80):
        sg, so = gw.step(dt, iters), ow.step(dt, iters)
        assert (sg.n_constraints, sg.n_terrain_constraints) == (so.n_constraints, so.n_terrain_constraints), f"tick {tick}"
        hits = max(hits, int(sg.n_terrain_constraints))
        if tick == 30:
            gw = gw.clone()
    assert hits > 5
    compare_constraints(gw.constraints(), ow.constraints(), check_impulse=True)
    g, o = gw.state(), ow.state()
    for k in ("x", "q", "v", "omega"):
        assert values_equal(g[k], o[k]), k


@pytest.mark.parametrize("name", ["spheres", "bodies_of_several_parts"])
def test_obstacles_on_the_exact_two_pass_candidate_path(ctx, name):
    """A tick whose partner row overflows is re-run on the exact count / fill path (k_candidates); a world with obstacles used to
    fail there with MGF_ERR_INVALID after the bodies had been integrated.  The obstacles' components are now counted and
    written on that path too (k_obstacle_candidates): option two_pass_candidates = 1 takes it every tick - same lists, same state."""
    sc = SCENES[name]()
    dt, iters = float(sc["dt"]), sc["iters"]
    gw, ow = mgf_amd.World.from_scene(ctx, sc), oracle_world(sc)
    _add(ctx, gw, ow)
    gw.set_option("two_pass_candidates", 1)
    for tick in range(60):
        sg, so = gw.step(dt, iters), ow.step(dt, iters)
        assert (sg.n_constraints, sg.n_terrain_constraints) == (so.n_constraints, so.n_terrain_constraints), f"tick {tick}"
        if tick % 20 == 19:
            compare_constraints(gw.constraints(), ow.constraints(), check_impulse=True)
    g, o = gw.state(), ow.state()
    for k in ("x", "q", "v", "omega"):
        assert values_equal(g[k], o[k]), k
    assert int((gw.constraints()["b"] < 0).sum()) > 10
