"""Solver::solve has no failure mode (solver.rs:72-78): a persistent solver launch that gives up - not all of its workgroups resident: a
device shared with another process, a CU mask - must not leave the world half-solved.  The launches are made to give
up here (option flow_spin_limit = 1: every other workgroup returns at once, the others wait for it in vain); the world restores the pre-launch velocities and impulses, solves the list with the
launch-per-frontier executor, counts it, and steps on bit-identical to a world that never gave up."""
import numpy as np
import pytest

import mgf_amd
from mgf_amd import scenes
from tests.util import values_equal

pytestmark = pytest.mark.gpu
STATE_KEYS = ("x", "q", "v", "omega", "delta")


@pytest.fixture(scope="module")
def ctx():
    c = mgf_amd.Context(0)
    yield c
    c.close()


def _same(a, b, what):
    sa, sb = a.state(), b.state()
    for f in STATE_KEYS:
        assert values_equal(sa[f], sb[f]), f"{what}: {f}"


@pytest.mark.parametrize("driver", ["step", "step_many", "build_solve"])
def test_a_launch_that_gives_up_is_solved_again_bit_identically(ctx, driver):
    scene = scenes.sphere_pile(24, 12, 24)
    dt, iters = float(scene["dt"]), scene["iters"]
    ref, w = mgf_amd.World.from_scene(ctx, scene), mgf_amd.World.from_scene(ctx, scene)
    w.set_option("flow_spin_limit", 1)  # every other workgroup of a persistent launch plays "never became resident"; the others give up waiting for it
    ticks = 40
    if driver == "step_many":
        ref.step_many(dt, iters, ticks)
        w.step_many(dt, iters, ticks)
    else:
        for _ in range(ticks):
            if driver == "step":
                ref.step(dt, iters); w.step(dt, iters)
            else:
                ref.build_constraints(dt); ref.solve(iters)
                w.build_constraints(dt); w.solve(iters)
    assert w.counter("solver_abort_fallbacks") >= 1, "the wait limit never bit: the test tests nothing"
    assert ref.counter("solver_abort_fallbacks") == 0
    _same(w, ref, driver)
    # ... and once the limit is back to normal the world returns to the persistent launch (after its back-off) and stays identical
    w.set_option("flow_spin_limit", 0)
    n0 = w.counter("solver_abort_fallbacks")
    for _ in range(3):
        ref.step_many(dt, iters, 40); w.step_many(dt, iters, 40)
    assert w.counter("solver_abort_fallbacks") == n0
    _same(w, ref, driver + " (afterwards)")


def test_a_world_told_to_share_the_device_uses_fewer_workgroups_and_steps_identically(ctx):
    """option flow_max_blocks: the persistent launches of this world take at most that many workgroups (one per CU) - what two processes
    on one device set so that both their launches are resident together."""
    scene = scenes.sphere_pile(24, 12, 24)
    dt, iters = float(scene["dt"]), scene["iters"]
    ref, w = mgf_amd.World.from_scene(ctx, scene), mgf_amd.World.from_scene(ctx, scene)
    w.set_option("flow_max_blocks", 8)
    ref.step_many(dt, iters, 30); w.step_many(dt, iters, 30)
    assert w.counter("flow5_blocks") <= 8 < ref.counter("flow5_blocks")
    _same(w, ref, "flow_max_blocks")
    with pytest.raises(mgf_amd.MgfError):
        w.set_option("flow_max_blocks", -1)


def test_a_tick_rerun_behind_a_speculative_one_keeps_its_terrain_contacts(ctx):
    """mgf_world_step_many enqueues tick k + 1 before it has read tick k's counts; when tick k then fails a capacity check, tick k + 1 -
    a no-op by its guard - has still cleared the per-tick counters, the terrain rows' counts of tick k's k_integrate among them.  The
    re-run of tick k must list those rows again (it used to take them as still there: the pile lost its floor contacts for a tick)."""
    scene = scenes.sphere_pile(24, 12, 24)
    dt, iters = float(scene["dt"]), scene["iters"]
    ref, w = mgf_amd.World.from_scene(ctx, scene), mgf_amd.World.from_scene(ctx, scene)
    ref.step_many(dt, iters, 10); w.step_many(dt, iters, 10)
    for _ in range(3):
        w.set_option("list_capacity", 1024)  # the next tick's lists do not fit: it is re-run, with a speculative tick behind it
        r0 = w.counter("capacity_retries")
        sa, sb = ref.step_many(dt, iters, 6), w.step_many(dt, iters, 6)
        assert w.counter("capacity_retries") > r0
        assert [int(x["n_terrain_constraints"]) for x in sa] == [int(x["n_terrain_constraints"]) for x in sb]
        assert int(sa[0]["n_terrain_constraints"]) > 0
    _same(w, ref, "re-run behind a speculative tick")
