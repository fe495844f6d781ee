"""World::step golden snapshots of the reference's demo scene (tests/golden/world_snapshots.npz, made by
tests/golden/make_world_snapshots.py from the CPU oracle).  The CPU test pins the oracle
to the committed snapshots; the GPU test holds the HIP path to them bit for bit."""
import os

import numpy as np
import pytest

from mgf_amd import scenes
from oracle import oracle as O
from tests.golden.make_world_snapshots import FIELDS, SCENES, STEPS, state_digest
from tests.util import bits_equal, oracle_world, rel_err

_SNAP = np.load(os.path.join(os.path.dirname(__file__), "golden", "world_snapshots.npz"))


def _check(name, oname, k, st, n_constraints, exact=True):
    key = f"{name}/{oname}/{k}"
    assert n_constraints == int(_SNAP[key + "/n_constraints"]), key
    ids = _SNAP[key + "/ids"]
    for f in FIELDS:
        if exact:
            assert bits_equal(st[f][ids], _SNAP[key + "/" + f]), f"{key}/{f}"
    if exact:
        assert state_digest(st) == str(_SNAP[key + "/sha256"]), key


@pytest.mark.parametrize("oname,order", [("demo", O.ORDER_DEMO), ("canonical", O.ORDER_CANONICAL)])
@pytest.mark.parametrize("name", list(SCENES))
def test_oracle_reproduces_snapshots(name, oname, order):
    scene = scenes.balls_demo(**SCENES[name])
    w = oracle_world(scene, order=order)
    k = 0
    for target in STEPS:
        while k < target:
            st = w.step(float(scene["dt"]), scene["iters"])
            k += 1
        _check(name, oname, target, w.state(), int(st.n_constraints))


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(SCENES))
def test_hip_world_in_the_references_own_order(name):
    """Option constraint_order = demo: the host replays world.rs:233-291 on the reference-built world BVH (refits inside the
    loop, partners in BVH::query's order) and the device inserts the constraints in that order - the HIP result is then
    comparable with the reference AS THE REFERENCE RUNS: bit-identical to the oracle's world.rs-order snapshots."""
    import mgf_amd
    scene = scenes.balls_demo(**SCENES[name])
    ctx = mgf_amd.Context(0)
    w = mgf_amd.World.from_scene(ctx, scene)
    w.set_option("constraint_order", 1)
    k = 0
    for target in STEPS:
        while k < target:
            st = w.step(float(scene["dt"]), scene["iters"])
            k += 1
        _check(name, "demo", target, w.state(), int(st.n_constraints))
    # ... constraint by constraint, against the oracle run in the same order (a clone taken mid-run carries the tree along)
    ow = oracle_world(scene, order=O.ORDER_DEMO)
    for _ in range(STEPS[-1]):
        ow.step(float(scene["dt"]), scene["iters"])
    w2 = w.clone()
    for tick in range(5):
        so, sg, sc = ow.step(float(scene["dt"]), scene["iters"]), w.step(float(scene["dt"]), scene["iters"]), w2.step(float(scene["dt"]), scene["iters"])
        assert (sg.n_constraints, sg.n_pair_candidates) == (so.n_constraints, so.n_pair_candidates) == (sc.n_constraints, sc.n_pair_candidates)
        got, want = w.constraints(), ow.constraints()
        assert np.array_equal(got["a"], want["a"]) and np.array_equal(got["b"], want["b"])
        assert bits_equal(got["normal_impulse"], want["normal_impulse"])
    for f in FIELDS:
        assert bits_equal(w.state()[f], ow.state()[f]) and bits_equal(w2.state()[f], ow.state()[f]), f
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(SCENES))
def test_hip_world_reproduces_snapshots(name):
    import mgf_amd
    scene = scenes.balls_demo(**SCENES[name])
    ctx = mgf_amd.Context(0)
    w = mgf_amd.World.from_scene(ctx, scene)
    k = 0
    for target in STEPS:
        while k < target:
            st = w.step(float(scene["dt"]), scene["iters"])
            k += 1
        state = w.state()
        _check(name, "canonical", target, state, int(st.n_constraints))
        # world.rs insertion order: same constraint set, different Gauss-Seidel order - report the deviation
        key = f"{name}/demo/{target}"
        ids = _SNAP[key + "/ids"]
        assert int(st.n_constraints) == int(_SNAP[key + "/n_constraints"])
        dev = max(rel_err(state[f][ids], _SNAP[key + "/" + f]) for f in FIELDS)
        print(f"{name} step {target}: canonical-vs-demo order deviation {dev:.3e}")
    ctx.close()
