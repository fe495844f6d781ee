"""k_pair_brick (the grid broadphase with an 8 x 8 x 8-cell box staged in LDS) against k_pair_grid (every look-up from global
memory) and against the oracle: the accepted partner set - and so the candidate statistic, the constraints and the state -
must not depend on which way a query was answered, including the queries the brick kernel hands to its one-lane global
path (regions that reach outside the box; boxes with more records than the LDS copy holds)."""
import numpy as np
import pytest

import mgf_amd
from mgf_amd import scenes
from tests.util import bits_equal, oracle_world

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = mgf_amd.Context(0)
    yield c
    c.close()


def _same(a, b, what):
    sa, sb = a.state(), b.state()
    for k in ("x", "q", "v", "omega", "delta"):
        assert bits_equal(sa[k], sb[k]), f"{what}: {k}"


SCENES = {
    "spheres": lambda: scenes.sphere_pile(14, 10, 14),                       # fused sphere test (k_pair_brick<true>)
    "capsules_and_spheres": lambda: scenes.capsule_field(8, 4, 8, pitch=1.15, sphere_fraction=0.5),  # partner rows (k_pair_brick<false>)
    "two_part_bodies": lambda: scenes.dumbbell_field(6, 4, 6, n_plain=40, pitch=1.5),
}


KERNELS = [1]  # option pair_brick: 1 = k_pair_brick (the cell walk in LDS)


@pytest.mark.parametrize("kern", KERNELS)
@pytest.mark.parametrize("name", sorted(SCENES))
def test_brick_and_grid_list_the_same_partners(ctx, name, kern):
    scene = SCENES[name]()
    dt, iters = float(scene["dt"]), scene["iters"]
    a, b = mgf_amd.World.from_scene(ctx, scene), mgf_amd.World.from_scene(ctx, scene)
    a.set_option("pair_brick", kern)
    b.set_option("pair_brick", 0)
    ow = oracle_world(scene)
    for tick in range(30):
        sa, sb, so = a.step(dt, iters), b.step(dt, iters), ow.step(dt, iters)
        assert (sa.n_pair_candidates, sa.n_constraints, sa.n_terrain_constraints) == (sb.n_pair_candidates, sb.n_constraints, sb.n_terrain_constraints), tick
        assert (sa.n_pair_candidates, sa.n_constraints) == (so.n_pair_candidates, so.n_constraints), tick
    assert sa.n_constraints > sa.n_terrain_constraints
    assert b.counter("pair_brick_slow_queries") == 0
    _same(a, b, name)


def _with_extra(scene, centres, radii):
    comps = np.zeros(len(scene["comps"]) + len(centres), scenes.COMPONENT_DTYPE)
    comps[:len(scene["comps"])] = scene["comps"]
    comps["p"][len(scene["comps"]):] = np.asarray(centres, np.float32)
    comps["r"][len(scene["comps"]):] = np.asarray(radii, np.float32)
    v0 = np.concatenate([scene["v0"], np.zeros((len(centres), 3), np.float32)])
    return scenes._scene(scene["name"] + "_extra", comps, scene["terrain"], v0=v0)


@pytest.mark.parametrize("kern", KERNELS)
def test_a_body_much_larger_than_a_cell_goes_through_global_memory(ctx, kern):
    """One sphere of radius 2 in a pile of radius-0.5 spheres: the largest fat half extent grows every query's region to 5-7 cells
    per axis; those that no longer fit the staged box are answered by the one-lane global path - same partners, same result as the oracle -
    and the world switches back to k_pair_grid for the ticks that follow."""
    base = scenes.sphere_pile(24, 10, 24)
    scene = _with_extra(base, [(0.0, 13.5, 0.0)], [2.0])
    dt, iters = float(scene["dt"]), scene["iters"]
    a, b = mgf_amd.World.from_scene(ctx, scene), mgf_amd.World.from_scene(ctx, scene)
    a.set_option("pair_brick", kern)
    b.set_option("pair_brick", 0)
    ow = oracle_world(scene)
    sa, sb, so = a.step(dt, iters), b.step(dt, iters), ow.step(dt, iters)
    assert a.counter("pair_brick_slow_queries") > len(a) // 8
    assert a.counter("pair_brick_off_ticks") > 0
    assert (sa.n_pair_candidates, sa.n_constraints) == (sb.n_pair_candidates, sb.n_constraints) == (so.n_pair_candidates, so.n_constraints)
    for tick in range(12):
        sa, sb, so = a.step(dt, iters), b.step(dt, iters), ow.step(dt, iters)
        assert (sa.n_pair_candidates, sa.n_constraints) == (sb.n_pair_candidates, sb.n_constraints) == (so.n_pair_candidates, so.n_constraints), tick
    assert a.counter("pair_brick_slow_queries") == 0  # (k_pair_grid is running)
    _same(a, b, "big body")
    so_state = ow.state()
    for k in ("x", "q", "v", "omega"):
        assert bits_equal(a.state()[k], so_state[k]), k


@pytest.mark.parametrize("kern", KERNELS)
def test_a_box_with_more_records_than_the_lds_copy_holds(ctx, kern):
    """A clump of 2500 small spheres and a few bodies far away (they stretch the scene bounds, so the cells are large and the
    clump sits in a handful of them): the bricks around the clump cannot stage their boxes and answer from global memory."""
    rng = np.random.default_rng(5)
    clump = rng.uniform(-3.0, 3.0, (2500, 3)).astype(np.float32) + np.float32([0, 6, 0])
    far = np.float32([[-60, 2, -60], [60, 2, 60], [60, 40, -60], [-60, 2, 60]])
    centres = np.concatenate([clump, far])
    radii = np.concatenate([np.full(2500, 0.1, np.float32), np.full(4, 0.5, np.float32)])
    perm = rng.permutation(len(centres))
    comps = np.zeros(len(centres), scenes.COMPONENT_DTYPE)
    comps["p"], comps["r"] = centres[perm], radii[perm]
    scene = scenes._scene("clump", comps, scenes.box_terrain(70.0, 50.0, (0, 0, 0)), v0=np.zeros((len(centres), 3), np.float32))
    dt, iters = float(scene["dt"]), 4
    a, b = mgf_amd.World.from_scene(ctx, scene), mgf_amd.World.from_scene(ctx, scene)
    a.set_option("pair_brick", kern)
    b.set_option("pair_brick", 0)
    ow = oracle_world(scene)
    for tick in range(3):
        sa, sb, so = a.step(dt, iters), b.step(dt, iters), ow.step(dt, iters)
        if tick == 0:
            assert a.counter("pair_brick_slow_queries") > 1000 and a.counter("pair_brick_off_ticks") > 0
        assert (sa.n_pair_candidates, sa.n_constraints) == (sb.n_pair_candidates, sb.n_constraints) == (so.n_pair_candidates, so.n_constraints), tick
    assert sa.n_pair_candidates > 5000 and sa.n_constraints > 100
    _same(a, b, "clump")


def test_a_list_capacity_miss_far_beyond_the_allocation_is_survived(ctx):
    """2500 heavily overlapping spheres: ~30 contacts per body on the first tick, seven times the constraint capacity a new
    world starts with.  The tick is re-run with grown lists; no kernel may touch the unwritten part in between
    (k_chain_rows used to)."""
    rng = np.random.default_rng(5)
    centres = rng.uniform(-3.0, 3.0, (2500, 3)).astype(np.float32) + np.float32([0, 6, 0])
    comps = np.zeros(len(centres), scenes.COMPONENT_DTYPE)
    comps["p"], comps["r"] = centres, 0.5
    scene = scenes._scene("overlapping", comps, scenes.box_terrain(20.0, 20.0, (0, 0, 0)), v0=np.zeros((len(centres), 3), np.float32))
    gw, ow = mgf_amd.World.from_scene(ctx, scene), oracle_world(scene)
    for tick in range(2):
        sg, so = gw.step(float(scene["dt"]), 2), ow.step(float(scene["dt"]), 2)
        assert (sg.n_pair_candidates, sg.n_constraints) == (so.n_pair_candidates, so.n_constraints)
    assert sg.n_constraints > 8 * len(gw)
    so_state = ow.state()
    for k in ("x", "q", "v", "omega"):
        assert bits_equal(gw.state()[k], so_state[k]), k
