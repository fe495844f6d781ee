"""The C++ oracle vs an independent numpy restatement (tests/np_restatement.py) on the parts of the path
that no reference test pins: integrate / complete_motion, ContactConstraint::new,
ContactConstraint::solve, Solver::solve.  Bit-exact, tick by tick, on small sphere piles."""
import numpy as np
import pytest

from mgf_amd import scenes
from tests import np_restatement as NP
from tests.util import bits_equal, oracle_world


def _np_bodies(scene, ow):
    st = ow.state()
    body_I, _ = ow.inv_moment()
    n = len(ow)
    return NP.Bodies(st["x"], st["q"], st["v"], st["omega"], st["delta"], scene["force"] * scene["mass"][:, None],
                     1.0 / scene["mass"].astype(np.float32), body_I, scene["restitution"], scene["friction"]), n


def _arr(list_of_vec):
    return np.array(list_of_vec, np.float32)


@pytest.mark.parametrize("dims,ticks", [((4, 4, 4), 12), ((3, 5, 2), 20), ((6, 6, 6), 16)])
def test_dynamics_restatements_agree_bit_for_bit(dims, ticks):
    scene = scenes.sphere_pile(*dims)
    dt, iters = float(scene["dt"]), 6
    center = scene["terrain"]["pos"]
    ow = oracle_world(scene)
    nb, n = _np_bodies(scene, ow)
    total_constraints = 0
    for tick in range(ticks):
        # complete_motion + integrate
        ow.begin_tick(dt)
        nb.complete_motion()
        nb.integrate(dt)
        st = ow.state()
        _, world_I = ow.inv_moment()
        assert bits_equal(_arr(nb.x), st["x"]) and bits_equal(_arr(nb.q), st["q"]), f"tick {tick}: x/q after integrate"
        assert bits_equal(_arr(nb.v), st["v"]) and bits_equal(_arr(nb.delta), st["delta"]), f"tick {tick}: v/delta after integrate"
        assert bits_equal(np.stack([NP.mat_flat(m) for m in nb.inv_moment]), world_I), f"tick {tick}: world inverse inertia"
        # ContactConstraint::new from the oracle's contact list (normal + local points are the narrowphase's output)
        ow.collide(dt)
        oc = ow.constraints()
        cons = [NP.Constraint(nb, int(c["a"]), int(c["b"]), c["normal"], c["ra"], c["rb"], dt, center) for c in oc]
        total_constraints += len(cons)
        for k, (c, o) in enumerate(zip(cons, oc)):
            got = np.concatenate([_arr(c.t[0]), _arr(c.t[1]), [c.bias, c.normal_mass, c.tangent_mass[0], c.tangent_mass[1], c.friction]]).astype(np.float32)
            want = np.concatenate([o["t0"], o["t1"], [o["bias"], o["normal_mass"], o["tangent_mass0"], o["tangent_mass1"], o["friction"]]]).astype(np.float32)
            assert np.array_equal(got, want), f"tick {tick} constraint {k}: {got} vs {want}"
        # Solver::solve
        ow.solve(iters)
        NP.solver_solve(cons, nb, iters)
        st = ow.state()
        assert bits_equal(_arr(nb.v), st["v"]), f"tick {tick}: v after solve"
        assert bits_equal(_arr(nb.omega), st["omega"]), f"tick {tick}: omega after solve"
        oc = ow.constraints()
        assert np.array_equal(np.array([c.normal_impulse for c in cons], np.float32), oc["normal_impulse"]), f"tick {tick}: accumulated impulses"
    assert total_constraints > 50 * ticks // 4  # the pile is in contact: the comparison is not vacuous
