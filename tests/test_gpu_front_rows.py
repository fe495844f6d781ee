"""The list-free front end of worlds that are not spheres only (r06, mgf_amd/csrc/k_front_rows.h; option front_rows): capsules, mixed
worlds, bodies of two components.  Against the oracle (constraint lists and states, bit for bit), against the list-based kernels it
replaces (front_rows = 0: candidate lists, one narrowphase launch per shape-pair type), and with the cheap conservative reject ahead of
the body-triangle tests (comp_tri_far, dev_geom.h) checked against the reference's tests on every face it drops (front_rows_check)."""
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


def _same(x, y):
    return all(np.array_equal(x[k].view(np.uint32), y[k].view(np.uint32)) for k in ("x", "q", "v", "omega"))


def _scene(name):
    from mgf_amd import scenes
    if name == "capsules_face_grid":      # 288 faces: the face grid, k_near_list + k_terrain_near + k_terrain_tests
        return scenes.capsule_field(8, 6, 8, quads=12, pitch=1.6)
    if name == "capsules_rows":           # 32 faces: the rows and records of k_integrate's tail, k_terrain_contacts<1>
        return scenes.capsule_field(8, 6, 8, quads=4, pitch=1.6)
    if name == "mixed_face_grid":
        return scenes.capsule_field(10, 6, 10, quads=16, pitch=1.6, sphere_fraction=0.3)
    if name == "mixed_rows":
        return scenes.capsule_field(10, 6, 10, quads=5, pitch=1.6, sphere_fraction=0.3)
    if name == "two_part_bodies_and_spheres":
        return scenes.dumbbell_field(8, 5, 8, n_plain=40)
    if name == "two_part_bodies":
        return scenes.dumbbell_field(10, 4, 10)
    if name == "two_part_bodies_wide_floor":
        # a floor 240 m across under a small field: the bodies come to rest around the floor's DIAGONAL, the shared edge of its two
        # triangles, 170 m from its far end - where the reference's edge tests are f32 noise at the scale of a capsule's radius and
        # report contacts 0.31 from an edge for r = 0.3 (what comp_tri_far's reach has to allow for; found by BASELINE config 5)
        sc = scenes.dumbbell_field(12, 3, 12)
        sc["terrain"] = scenes.box_terrain(120.0, 20.0, (0.0, 0.0, 0.0))
        return sc
    raise KeyError(name)


SCENES = ["capsules_face_grid", "capsules_rows", "mixed_face_grid", "mixed_rows", "two_part_bodies_and_spheres", "two_part_bodies",
          "two_part_bodies_wide_floor"]


@pytest.mark.parametrize("name", SCENES)
def test_front_rows_matches_the_lists_and_the_oracle(ctx, name):
    import mgf_amd
    sc = _scene(name)
    dt, it = float(sc["dt"]), sc["iters"]
    a, b = mgf_amd.World.from_scene(ctx, sc), mgf_amd.World.from_scene(ctx, sc)
    b.set_option("front_rows", 0)
    a.set_option("front_rows_check", 1)
    assert a.counter("front_rows") == 1
    ow = oracle_world(sc)
    ticks = 300 if "wide_floor" in name else 200
    for s in range(ticks):
        if s % 40 == 0:  # against the oracle from the same state: the constraint list and the state after the solve
            st = a.state()
            ow.set_state(x=st["x"], q=st["q"], v=st["v"], omega=st["omega"], delta=st["delta"])
            ow.build_constraints(dt)
            sa, sb = a.build_constraints(dt), b.build_constraints(dt)
            compare_constraints(a.constraints(), ow.constraints())
            assert sa.n_pair_candidates == sb.n_pair_candidates and sa.n_terrain_candidates == sb.n_terrain_candidates, s
            ow.solve(it); a.solve(it); b.solve(it)
            g, o = a.state(), ow.state()
            for k in ("x", "q", "v", "omega"):
                assert values_equal(g[k], o[k]), (name, s, k)
        else:
            sa, sb = a.step(dt, it), b.step(dt, it)
        assert sa.n_constraints == sb.n_constraints and sa.n_terrain_constraints == sb.n_terrain_constraints, (name, s)
        assert _same(a.state(), b.state()), (name, s)
    assert sa.n_constraints > 200 and sa.n_terrain_constraints > 50, (sa.n_constraints, sa.n_terrain_constraints)
    assert a.counter("front_rows") == 1  # (never switched off on the way)


def test_front_rows_step_many_and_the_side_stream(ctx):
    """mgf_world_step_many (two ticks in flight) with the terrain kernels on the context's second stream, against single steps on one stream."""
    import mgf_amd
    sc = _scene("capsules_face_grid")
    dt, it = float(sc["dt"]), sc["iters"]
    a, b = mgf_amd.World.from_scene(ctx, sc), mgf_amd.World.from_scene(ctx, sc)
    b.set_option("side_stream", 0)
    for _ in range(6):
        many = a.step_many(dt, it, 30)
        singles = [int(b.step(dt, it).n_constraints) for _ in range(30)]  # (step returns the world's one stats record: its numbers are taken at once)
        assert [int(m["n_constraints"]) for m in many] == singles
        assert _same(a.state(), b.state())
    assert b.stats.n_terrain_constraints > 50


@pytest.mark.parametrize("config", ["config3", "config5"])
def test_front_rows_full_size(ctx, config):
    """BASELINE configs 3 and 5 at full size, 150 ticks into the pile: the front end with its reject checked against the list-based kernels."""
    import mgf_amd
    from mgf_amd import scenes
    sc = scenes.capsule_field(128, 32, 32, quads=158) if config == "config3" else scenes.dumbbell_field(64, 16, 64)
    dt, it = float(sc["dt"]), sc["iters"]
    a, b = mgf_amd.World.from_scene(ctx, sc), mgf_amd.World.from_scene(ctx, sc)
    b.set_option("front_rows", 0)
    a.set_option("front_rows_check", 1)
    for s in range(50, 151, 50):
        sa, sb = a.step_many(dt, it, 50), b.step_many(dt, it, 50)
        assert _same(a.state(), b.state()), (config, s)
        for key in ("n_constraints", "n_terrain_constraints", "n_pair_candidates", "n_terrain_candidates"):
            assert int(sa[49][key]) == int(sb[49][key]), (config, s, key)
    assert int(sa[49]["n_constraints"]) > 100000
    if config == "config3":
        assert 0 < a.counter("front_slots") <= a.counter("front_faces")
