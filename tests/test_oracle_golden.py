"""Pin the CPU oracle against every known-answer vector the reference's own unit tests
hold for the hot path: collision.rs:1543-2268, bvh.rs:514-529,
bounds.rs:330-350, pool.rs:255-389, physics.rs:321-335, geom.rs:1154-1161."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle as O
from tests.golden_check import check_value, run_contacts_case

import json
import os

_G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_known_answers.json")))


def oracle_contacts(a, va, b, vb):
    return O.contacts(O.shape_from_dict(a), va, O.shape_from_dict(b), vb)


@pytest.mark.parametrize("case", _G["contacts"], ids=[c["id"] for c in _G["contacts"]])
def test_contacts_known_answers(case):
    run_contacts_case(case, oracle_contacts)


@pytest.mark.parametrize("case", _G["ray_capsule"], ids=[c["id"] for c in _G["ray_capsule"]])
def test_ray_capsule_known_answers(case):
    d = np.asarray(case["d"], np.float32)
    if case["normalize_d"]:
        # Vector3::normalize = v * (1/|v|) in f32
        mag = np.sqrt(np.float32(d[0] * d[0] + d[1] * d[1]) + np.float32(d[2] * d[2]), dtype=np.float32)
        d = (d * (np.float32(1.0) / mag)).astype(np.float32)
    cap = O.shape(O.CAPSULE, case["capsule"]["a"], case["capsule"]["d"], case["capsule"]["r"])
    ip = O.Vec3()
    t = C.c_float()
    hit = O.lib().mgfo_ray_capsule(C.byref(O.vec3(case["p"])), C.byref(O.vec3(d)), C.byref(cap), C.byref(ip), C.byref(t))
    assert hit == 1
    check_value(ip.tup(), case["expect_p"], case["id"] + ".p")
    if "expect_t" in case:
        check_value(t.value, case["expect_t"], case["id"] + ".t")
    if "expect_p_plus_dt" in case:
        p = np.asarray(case["p"], np.float32) + d * np.float32(t.value)
        check_value(p, case["expect_p_plus_dt"], case["id"] + ".p+d*t")


def test_sphere_tensor():
    m = _G["misc"]["sphere_tensor"]
    out = (C.c_float * 9)()
    comp = O.component(0, m["sphere"]["c"], [0, 0, 0], m["sphere"]["r"])
    O.lib().mgfo_tensor(C.byref(comp), m["mass"], out)
    assert [np.float32(x) for x in out] == [np.float32(x) for x in m["expect_cols"]]


def test_tri_closest_point():
    m = _G["misc"]["tri_closest_point"]
    tri = O.shape(O.TRIANGLE, m["tri"]["a"], m["tri"]["b"], m["tri"]["c"])
    out = O.Vec3()
    O.lib().mgfo_tri_closest_point(C.byref(tri), C.byref(O.vec3(m["to"])), C.byref(out))
    p = np.asarray(out.tup(), np.float32)
    assert float(p @ p) < m["expect_magnitude2_lt"]


def test_bvh_query():
    m = _G["misc"]["bvh_query"]
    bvh = O.Bvh()
    for s in m["spheres"]:
        bvh.insert(s["c"], [s["r"]] * 3, s["val"])
    found = 0
    for s in m["spheres"]:
        hits = bvh.query(s["c"], [s["r"]] * 3)
        assert hits == [s["val"]]
        found += len(hits)
    assert found == 3


def test_aabb_bounds():
    m = _G["misc"]["aabb"]
    mk = lambda d: O.Aabb(O.vec3(d["c"]), O.vec3(d["r"]))
    b1, b2, b3 = mk(m["b1"]), mk(m["b2"]), mk(m["b3"])
    comb = O.Aabb()
    L = O.lib()
    L.mgfo_aabb_combine(C.byref(b1), C.byref(b2), C.byref(comb))
    assert L.mgfo_aabb_overlaps(C.byref(b1), C.byref(b2))
    assert not L.mgfo_aabb_overlaps(C.byref(b1), C.byref(b3))
    assert not L.mgfo_aabb_contains(C.byref(b1), C.byref(b2))
    assert L.mgfo_aabb_contains(C.byref(comb), C.byref(b1))
    assert L.mgfo_aabb_contains(C.byref(comb), C.byref(b2))
    assert not L.mgfo_aabb_contains(C.byref(comb), C.byref(b3))


def _pool_vals(p):
    idx = (C.c_uint64 * 64)()
    val = (C.c_uint64 * 64)()
    n = O.lib().mgfo_pool_iter(p, idx, val, 64)
    return [val[i] for i in range(n)]


def test_pool_known_answers():
    m = _G["misc"]["pool"]
    L = O.lib()
    out = C.c_uint64()
    p = L.mgfo_pool_new()
    ids = [L.mgfo_pool_push(p, v) for v in m["manual"]["push"]]
    assert ids[0] == 0 and ids[3] == 3
    for r in m["manual"]["remove"]:
        assert L.mgfo_pool_remove(p, ids[r], C.byref(out)) == 0
    assert L.mgfo_pool_get(p, ids[0], C.byref(out)) == 0 and out.value == 0
    assert L.mgfo_pool_get(p, ids[3], C.byref(out)) == 0 and out.value == 3
    assert L.mgfo_pool_get(p, ids[1], C.byref(out)) == -1  # reference panics: "not occupied"
    assert _pool_vals(p) == m["manual"]["expect_iter_vals"]
    # LIFO reuse of freed slots (pool.rs:81-113): last removed id comes back first
    assert L.mgfo_pool_push(p, 9) == ids[2]
    assert L.mgfo_pool_push(p, 9) == ids[1]
    L.mgfo_pool_free(p)
    for sc in m["scenarios"]:
        p = L.mgfo_pool_new()
        for i in range(sc["n"]):
            L.mgfo_pool_push(p, i)
        assert _pool_vals(p) == list(range(sc["n"]))
        for r in sc["remove"]:
            assert L.mgfo_pool_remove(p, r, C.byref(out)) == 0
        assert _pool_vals(p) == sc["expect"]
        for r in sc["then_remove"]:
            assert L.mgfo_pool_remove(p, r, C.byref(out)) == 0
        assert _pool_vals(p) == sc["then_expect"]
        L.mgfo_pool_free(p)
