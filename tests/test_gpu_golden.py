"""The reference's known-answer vectors, run through the C-ABI on the GPU (same fixture file the
oracle is pinned against)."""
import json
import os

import numpy as np
import pytest

from tests.golden_check import check_value, run_contacts_case

pytestmark = pytest.mark.gpu
_G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_known_answers.json")))


def _on_path(case):
    return case["a"]["kind"] != "rectangle" and case["b"]["kind"] != "rectangle"


@pytest.fixture(scope="module")
def ctx():
    import mgf_amd
    c = mgf_amd.Context(0)
    yield c
    c.close()


_CASES = [c for c in _G["contacts"] if _on_path(c)]


@pytest.mark.parametrize("case", _CASES, ids=[c["id"] for c in _CASES])
def test_contacts_known_answers_hip(ctx, case):
    import mgf_amd
    run_contacts_case(case, lambda a, va, b, vb: mgf_amd.contacts(ctx, a, va, b, vb))


@pytest.mark.parametrize("case", _G["ray_capsule"], ids=[c["id"] for c in _G["ray_capsule"]])
def test_ray_capsule_known_answers_hip(ctx, case):
    import mgf_amd
    d = np.asarray(case["d"], np.float32)
    if case["normalize_d"]:
        mag = np.sqrt(np.float32(d[0] * d[0] + d[1] * d[1]) + np.float32(d[2] * d[2]), dtype=np.float32)
        d = (d * (np.float32(1.0) / mag)).astype(np.float32)
    r = mgf_amd.ray_capsule(ctx, case["p"], d, case["capsule"]["a"], case["capsule"]["d"], case["capsule"]["r"])
    assert r is not None
    ip, t = r
    check_value(ip, case["expect_p"], case["id"] + ".p")
    if "expect_t" in case:
        check_value(t, case["expect_t"], case["id"] + ".t")


def test_sphere_tensor_hip():
    import mgf_amd
    m = _G["misc"]["sphere_tensor"]
    got = mgf_amd.inertia_tensor(0, m["sphere"]["c"], [0, 0, 0], m["sphere"]["r"], m["mass"])
    assert [np.float32(x) for x in got] == [np.float32(x) for x in m["expect_cols"]]


def test_bvh_known_answers_hip(ctx):
    import mgf_amd
    m = _G["misc"]["bvh_query"]
    bvh = mgf_amd.Bvh(ctx)
    assert bvh.empty()
    with pytest.raises(mgf_amd.MgfError) as e:
        bvh.root()
    assert e.value.status == 1  # MGF_ERR_EMPTY (bvh.rs:265 panics)
    ids = [bvh.insert(s["c"], [s["r"]] * 3, s["val"]) for s in m["spheres"]]
    found = 0
    for s in m["spheres"]:
        hits = bvh.query(s["c"], [s["r"]] * 3)
        assert hits == [s["val"]]
        found += len(hits)
    assert found == 3
    assert bvh.get_leaf(ids[0]) == 1
    with pytest.raises(mgf_amd.MgfError) as e:
        bvh.get_leaf(bvh.root())
    assert e.value.status == 3  # NOT_LEAF (bvh.rs:274)
