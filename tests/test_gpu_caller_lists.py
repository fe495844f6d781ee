"""A caller's constraint list (mgf_world_set_constraints, the Solver handle: solver.rs:53-79) on the block-local solver
(VERDICT r2 item 3): its block tables are built from the list's generic dependency links (k_flow6g_*), so the drop-in
Solver::solve runs the kernel the tick runs - same bits as the launch-per-frontier executor, and about as fast as the world's own solve."""
import numpy as np
import pytest

import mgf_amd
from mgf_amd import scenes
from tests.util import compare_constraints

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = mgf_amd.Context(0)
    yield c
    c.close()


def _same(a, b, what):
    for k in ("v", "omega"):
        assert np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)), f"{what}: {k}"


@pytest.mark.parametrize("scene_name", ["spheres", "mixed"])
@pytest.mark.parametrize("resort", [0, 1])
def test_callers_list_on_the_block_local_solver_equals_the_frontier_executor(ctx, scene_name, resort):
    scene = scenes.sphere_pile(12, 10, 12) if scene_name == "spheres" else scenes.capsule_field_dense(10, 4, 10, quads=12, y0=0.9, sphere_fraction=0.5)
    dt, iters = float(scene["dt"]), scene["iters"]
    a, b = mgf_amd.World.from_scene(ctx, scene), mgf_amd.World.from_scene(ctx, scene)
    for w in (a, b):
        w.set_option("resort_every", resort)
        w.set_option("flow5_block", 64)    # (several blocks on a scene this small: links across block faces, messages)
        for _ in range(25):
            w.step(dt, iters)
    a.set_option("solver_mode", 0)
    a.build_constraints(dt)
    b.build_constraints(dt)
    lst = a.constraints()
    assert len(lst) > 150
    # the same list in a shuffled insertion order: nothing of the tick's own structure (contiguous per-body ranges) is left
    rng = np.random.default_rng(5)
    shuffled = lst[rng.permutation(len(lst))]
    for it in (10, 3):
        a.set_constraints(shuffled)
        b.set_constraints(shuffled)
        runs0 = b.counter("flow6_runs")
        a.solve(it)
        b.solve(it)
        assert b.counter("flow6_runs") == runs0 + 1 and b.counter("flow6_fallbacks") == 0, "the block-local kernel did not take the caller's list"
        _same(a.state(), b.state(), f"{scene_name}, {it} iterations")
        compare_constraints(b.constraints(), a.constraints(), check_impulse=True)
    # through the Solver handle, impulses kept between the calls (solver.rs: &mut self)
    sa, sb = mgf_amd.Solver(), mgf_amd.Solver()
    sa.add_constraints(lst)
    sb.add_constraints(lst)
    for _ in range(2):
        sa.solve(a, 4)
        sb.solve(b, 4)
    _same(a.state(), b.state(), "Solver handle")
    compare_constraints(sb.constraints(), sa.constraints(), check_impulse=True)


def test_solver_handle_at_full_size_is_as_fast_as_the_worlds_own_solve(ctx):
    """262 144 spheres, ~0.5 M constraints: Solver::solve on the caller's copy of the tick's list against the world's own solve
    of that list (HIP events around the solve phase; the caller's list pays for its table build inside that span)."""
    scene = scenes.sphere_pile(64, 64, 64)
    dt, iters = float(scene["dt"]), scene["iters"]
    own, other = mgf_amd.World.from_scene(ctx, scene), mgf_amd.World.from_scene(ctx, scene)
    for w in (own, other):
        w.step_many(dt, iters, 40)
        w.build_constraints(dt)
    lst = own.constraints()
    assert len(lst) > 450000
    ms_own = []
    snap = own.clone()
    for _ in range(3):   # the world's own solve of its list (a fresh clone each time: same velocities)
        w = snap.clone()
        w.set_option("phase_timing", 1)   # (mgf_step_stats::ms_solve: HIP events around the solve phase)
        w.build_constraints(dt)
        ms_own.append(w.solve(iters).ms_solve)
    sv = mgf_amd.Solver()
    sv.add_constraints(lst)
    ms_list = []
    for _ in range(3):
        w = snap.clone()
        w.set_option("phase_timing", 1)
        w.build_constraints(dt)      # (the spatial order of the bodies: the caller's list is cut into blocks by it)
        sv.clear()
        sv.add_constraints(lst)
        runs0 = w.counter("flow6_runs")
        ms_list.append(sv.solve(w, iters).ms_solve)
        assert w.counter("flow6_runs") == runs0 + 1
    ref = snap.clone()            # the same list by the launch-per-frontier executor
    ref.build_constraints(dt)
    ref.set_option("solver_mode", 0)
    ref.set_constraints(lst)
    ref.solve(iters)
    _same(ref.state(), w.state(), "full size")
    print(f"own solve {min(ms_own):.3f} ms, Solver handle on the caller's list {min(ms_list):.3f} ms")
    # (alone on the device the ratio is 1.18-1.22 - 0.52-0.54 against 0.44-0.46 ms; behind tests that left a million-body tile set's
    # buffers around it has been seen at 1.5: the bound here only catches a list that fell back to the global dataflow launch, ~3x)
    assert min(ms_list) <= 1.6 * min(ms_own) + 0.05, (ms_own, ms_list)
