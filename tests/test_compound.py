"""mgf::Compound (compound.rs:230-352): the oracle against the reference's own test (compound.rs:360-389, vectors in
tests/golden/reference_known_answers.json), the HIP path (mgf_compound_*) against the oracle bit for bit."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from oracle import oracle as O

_G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_known_answers.json")))
F = np.float32


def _from_arc_normalized(src, dst):
    q = O.Quat()
    O.lib().mgfo_quat_from_arc(C.byref(O.vec3(src)), C.byref(O.vec3(dst)), C.byref(q))
    s, x, y, z = F(q.s), F(q.x), F(q.y), F(q.z)
    inv = F(1) / np.sqrt(s * s + (x * x + y * y + z * z))  # Quaternion::normalize = q * (1 / |q|)
    return (s * inv, x * inv, y * inv, z * inv)


def test_oracle_reproduces_the_reference_compound_test():
    case = _G["compound"][0]
    comp = O.Compound([O.component(O.SPHERE, tuple(c["c"]), (0, 0, 0), c["r"]) for c in case["components"]])
    ts = case["test_sphere"]
    sphere = O.shape(O.SPHERE, tuple(ts["c"]), ts["r"])
    for step in case["steps"]:
        rot = (1.0, 0.0, 0.0, 0.0) if step["rot"] == "identity" else _from_arc_normalized(*step["rot"]["from_arc"])
        comp.set_pose((0, 0, 0), rot)
        if "rhs" in step:
            r = step["rhs"]
            got = comp.contacts(O.shape(O.RECTANGLE, tuple(r["c"]), tuple(r["u0"]), tuple(r["u1"]), tuple(r["e"])), tuple(r["vel"]))
        else:
            got = comp.contacts(sphere, tuple(ts["vel"]))
        if step["expect"] == "no_contact":
            assert got == []
        elif step["expect"] == "some_contact":
            assert len(got) >= 1
        else:
            last = got[-1]  # Contacts::last_contact collision.rs:477-481
            eps = step["expect"]["epsilon"]
            assert abs(last["t"] - step["expect"]["t"]) <= eps * max(1.0, abs(step["expect"]["t"]))
            assert np.allclose(last["a"], step["expect"]["a"], atol=eps, rtol=eps)


def _random_compound(rng, n):
    comps = np.zeros(n, O.COMPONENT_DTYPE)
    comps["tag"] = rng.integers(0, 2, n)
    comps["p"] = rng.uniform(-6, 6, (n, 3))
    comps["d"] = rng.uniform(-1.5, 1.5, (n, 3))
    comps["d"][comps["tag"] == 0] = 0
    comps["r"] = rng.uniform(0.3, 1.2, n)
    return comps


def _random_quat(rng):
    q = rng.normal(size=4).astype(np.float32)
    return tuple(q / np.sqrt((q * q).sum(dtype=np.float32)))


def test_compound_bounds_and_ray_properties():
    rng = np.random.default_rng(3)
    comps = _random_compound(rng, 24)
    comp = O.Compound([O.component(int(c["tag"]), tuple(c["p"]), tuple(c["d"]), float(c["r"])) for c in comps])
    c0, r0 = comp.bounds()
    comp.set_pose((1.0, 2.0, 3.0), (1, 0, 0, 0))
    c1, r1 = comp.bounds()
    assert np.allclose(np.array(c1) - np.array(c0), [1, 2, 3]) and np.allclose(r0, r1)  # translation moves the box only
    # a ray that hits: the reported point lies on the ray at parameter t
    hit = comp.intersection((-30.0, 2.0, 3.0), (1.0, 0.0, 0.0))
    if hit is not None:
        p, t = hit
        assert np.allclose(p, np.array([-30.0, 2.0, 3.0]) + t * np.array([1.0, 0, 0]), atol=1e-4)
    assert comp.intersection((-30.0, 60.0, 3.0), (1.0, 0.0, 0.0)) is None


@pytest.mark.gpu
def test_hip_compound_matches_oracle_bitwise():
    import mgf_amd
    ctx = mgf_amd.Context(0)
    rng = np.random.default_rng(21)
    comps = _random_compound(rng, 40)
    oc = O.Compound([O.component(int(c["tag"]), tuple(c["p"]), tuple(c["d"]), float(c["r"])) for c in comps])
    gc = mgf_amd.Compound(ctx, comps.astype(mgf_amd.COMPONENT_DTYPE))
    n = 600
    moving = np.zeros(n, mgf_amd.MOVING_DTYPE)
    moving["tag"] = rng.integers(0, 2, n)
    moving["p"] = rng.uniform(-9, 9, (n, 3))
    moving["d"] = rng.uniform(-1, 1, (n, 3))
    moving["d"][moving["tag"] == 0] = 0
    moving["r"] = rng.uniform(0.2, 1.0, n)
    moving["delta"] = rng.uniform(-3, 3, (n, 3))
    parts = np.zeros(300, mgf_amd.PARTICLE_DTYPE)
    parts["p"] = rng.uniform(-14, 14, (300, 3))
    parts["d"] = (rng.uniform(-6, 6, (300, 3)) - parts["p"]) * rng.uniform(0.3, 2.0, (300, 1))  # aimed at the aggregate
    parts["dt"] = np.where(np.arange(300) % 3 == 0, 1.0, np.inf)
    total = hits = 0
    for pose in range(4):
        disp = tuple(rng.uniform(-2, 2, 3).astype(np.float32)) if pose else (0.0, 0.0, 0.0)
        rot = _random_quat(rng) if pose else (1.0, 0.0, 0.0, 0.0)
        oc.set_pose(disp, rot); gc.set_pose(disp, rot)
        assert np.array_equal(np.float32(gc.bounds()), np.float32(oc.bounds()))
        off, got = gc.contacts_many(moving)
        for i in range(n):
            m = moving[i]
            sh = O.shape(O.SPHERE, tuple(m["p"]), float(m["r"])) if m["tag"] == 0 else O.shape(O.CAPSULE, tuple(m["p"]), tuple(m["d"]), float(m["r"]))
            want = oc.contacts(sh, tuple(m["delta"]), cap=64)
            assert off[i + 1] - off[i] == len(want), (pose, i)
            for k, w in enumerate(want):
                g = got[off[i] + k]
                for f in ("a", "b", "n"):
                    assert np.array_equal(g[f], np.float32(w[f])), (pose, i, k, f, g[f], w[f])
                assert g["t"] == np.float32(w["t"])
            total += len(want)
        gi = gc.intersections(parts)
        for i in range(300):
            want = oc.intersection(tuple(parts["p"][i]), tuple(parts["d"][i]), float(parts["dt"][i]))
            assert (gi[i] is None) == (want is None), (pose, i)
            if want is not None:
                hits += 1
                assert np.array_equal(np.float32(gi[i][0]), np.float32(want[0])) and np.float32(gi[i][1]) == np.float32(want[1])
    assert total > 200 and hits > 50
    ctx.close()
