"""Closed-form cases for the parts of the path the reference never tests (solver.rs, manifold.rs,
physics.rs integrate): values derived by hand from the source text, checked on the oracle.  The
same scenes run on the HIP path in tests/test_gpu_parity.py::test_closed_form_cases_hip."""
import numpy as np
import pytest

from mgf_amd import scenes
from oracle import oracle as O

DT = np.float32(1.0 / 60.0)
G = np.float32(-9.8)


def floor_scene(centres, v0=None, gravity=(0.0, -9.8, 0.0), r=0.5):
    comps = np.zeros(len(centres), scenes.COMPONENT_DTYPE)
    comps["tag"] = 0
    comps["p"] = np.asarray(centres, np.float32)
    comps["r"] = r
    verts = np.array([(-50, 0, -50), (-50, 0, 50), (50, 0, 50), (50, 0, -50)], np.float32)
    faces = np.array([(0, 1, 3), (1, 2, 3)], np.uint32)  # world.rs:140-141 winding: normal +y
    n = len(comps)
    return dict(name="floor", comps=comps, terrain=dict(verts=verts, faces=faces, pos=np.zeros(3, np.float32)),
                dt=DT, iters=10, mass=np.ones(n, np.float32), restitution=np.full(n, 0.3, np.float32),
                friction=np.full(n, 0.6, np.float32), force=np.tile(np.asarray(gravity, np.float32), (n, 1)),
                v0=None if v0 is None else np.asarray(v0, np.float32))


def make(scene):
    w = O.World(O.ORDER_CANONICAL)
    t = scene["terrain"]
    if t is not None:
        w.set_terrain(t["verts"], t["faces"], t["pos"])
    w.add_bodies(scene["comps"], scene["mass"], scene["restitution"], scene["friction"], scene["force"])
    if scene["v0"] is not None:
        w.set_state(v=scene["v0"])
    return w


CASES = {}


def case(fn):
    CASES[fn.__name__] = fn
    return fn


@case
def resting_sphere_attractive_band(make_world):
    """Sphere exactly resting: pen = delta.y in (-slop, 0] -> bias < 0 -> clamped impulse 0 (Appendix A.3):
    the body keeps its gravity-integrated velocity."""
    w = make_world(floor_scene([(5, 0.5, 1)]))
    st = w.step(float(DT), 10)
    assert st.n_constraints == 1
    c = w.constraints()[0]
    vy = np.float32(G * np.float32(1.0) * DT)          # physics.rs:236
    delta_y = np.float32(vy * DT)                      # physics.rs:249
    assert c["b"] == -1 and tuple(c["normal"]) == (0.0, -1.0, 0.0)       # n points body -> terrain
    assert tuple(c["ra"]) == (0.0, -0.5, 0.0) and tuple(c["rb"]) == (5.0, 0.0, 1.0)  # contact point relative to mesh.center() = 0
    pen = -delta_y * np.float32(-1.0) * np.float32(-1.0)  # (cb - ca) . n = -(x.y + delta.y - 0.5) * -1 ... = delta_y
    bias = np.float32(np.float32(-0.2) / DT) * np.float32(np.float32(delta_y) + np.float32(0.05))
    assert c["bias"] == pytest.approx(float(bias), rel=1e-6) and c["bias"] < 0
    assert c["normal_mass"] == 1.0 and c["normal_impulse"] == 0.0
    assert c["tangent_mass0"] == pytest.approx(1.0 / 3.5, rel=1e-6)     # 1/(1 + 0.25 * 10)
    assert w.state()["v"][0, 1] == vy


@case
def deep_penetration_is_pushed_out(make_world):
    """pen <= -slop: post-solve normal velocity equals the Baumgarte bias (solver.rs:145-149, 237)."""
    w = make_world(floor_scene([(5, 0.3, 1)]))
    w.step(float(DT), 10)
    c = w.constraints()[0]
    assert c["bias"] > 0
    assert w.state()["v"][0, 1] == pytest.approx(float(c["bias"]), rel=1e-5)
    assert c["normal_impulse"] > 0


@case
def sliding_sphere_friction_is_unclamped(make_world):
    """Friction rows ignore the Coulomb bound (solver.rs:223-231): with ZERO normal impulse one iteration
    still removes all slip at the contact: v.x = 1 - 1/3.5, w.z = -10 * 0.5 / 3.5."""
    w = make_world(floor_scene([(5, 0.5, 1)], v0=[(1.0, 0.0, 0.0)]))
    w.step(float(DT), 1)
    s = w.state()
    assert w.constraints()[0]["normal_impulse"] == 0.0
    assert s["v"][0, 0] == pytest.approx(1.0 - 1.0 / 3.5, rel=1e-6)
    assert s["omega"][0, 2] == pytest.approx(-10.0 * 0.5 / 3.5, rel=1e-6)
    w2 = make_world(floor_scene([(5, 0.5, 1)], v0=[(1.0, 0.0, 0.0)]))
    w2.step(float(DT), 10)  # further iterations see zero slip
    assert w2.state()["v"][0, 0] == pytest.approx(1.0 - 1.0 / 3.5, rel=1e-5)


@case
def head_on_spheres_restitution(make_world):
    """rel_v < -1 at creation adds -e * rel_v to the bias (solver.rs:149-153); max(e_a, e_b) = 0.3."""
    sc = floor_scene([(-0.5, 100.0, 0.0), (0.5, 100.0, 0.0)], v0=[(2, 0, 0), (-2, 0, 0)], gravity=(0, 0, 0))
    w = make_world(sc)
    st = w.step(float(DT), 10)
    assert st.n_constraints == 1
    c = w.constraints()[0]
    assert (c["a"], c["b"]) == (1, 0)                                   # world.rs:266: partner j < i
    assert tuple(c["normal"]) == (-1.0, 0.0, 0.0)                       # from a (body 1) towards b (body 0)
    pen = np.float32(-4.0) * DT                                         # end-of-sweep overlap
    bias = np.float32(-12.0) * (pen + np.float32(0.05)) + np.float32(0.3 * 4.0)
    assert c["bias"] == pytest.approx(float(bias), rel=1e-4)
    v = w.state()["v"]
    assert v[0, 0] + v[1, 0] == pytest.approx(0.0, abs=1e-6)            # momentum
    assert v[1, 0] - v[0, 0] == pytest.approx(float(c["bias"]), rel=1e-5)  # separating speed = bias


@case
def free_fall_and_delayed_position_update(make_world):
    """x advances with the PRE-solve velocity one tick late (physics.rs:249,266; Appendix A.5)."""
    w = make_world(floor_scene([(5, 50.0, 1)]))
    w.step(float(DT), 10)
    s = w.state()
    vy1 = np.float32(G * DT)
    assert s["x"][0, 1] == np.float32(50.0) and s["v"][0, 1] == vy1 and s["delta"][0, 1] == np.float32(vy1 * DT)
    w.step(float(DT), 10)
    s = w.state()
    assert s["x"][0, 1] == np.float32(np.float32(50.0) + np.float32(vy1 * DT))
    assert s["v"][0, 1] == np.float32(vy1 + np.float32(G * DT))


@pytest.mark.parametrize("name", sorted(CASES))
def test_closed_form_cases_oracle(name):
    CASES[name](make)
