"""Shared checker for the reference's known-answer vectors (tests/golden/*.json).

Used twice: against the CPU oracle (-m "not gpu") and against the HIP path through
the C-ABI (-m gpu).  `contacts_fn(a_dict, vel_a, b_dict, vel_b) -> list[dict(a,b,n,t)]`.
"""
import numpy as np

F32_EPS = np.float32(1.1920929e-07)


def relative_eq(a, b, eps):
    """approx::relative_eq for f32 (max_relative = f32::EPSILON)."""
    a = np.float32(a)
    b = np.float32(b)
    if a == b:
        return True
    if np.isinf(a) or np.isinf(b):
        return False
    d = abs(a - b)
    if d <= np.float32(eps):
        return True
    return d <= max(abs(a), abs(b)) * F32_EPS


def check_value(got, spec, what):
    mode = spec["mode"]
    if mode == "lt1m":
        assert (np.float32(1.0) - np.float32(got)) < np.float32(spec["eps"]), f"{what}: 1-{got} !< {spec['eps']}"
        return
    want = spec["value"]
    gv = np.atleast_1d(np.asarray(got, dtype=np.float32))
    wv = np.atleast_1d(np.asarray(want, dtype=np.float32))
    assert gv.shape == wv.shape, what
    for g, w in zip(gv, wv):
        if mode == "eq":
            assert g == w, f"{what}: got {gv!r} want exactly {wv!r}"
        elif mode == "rel":
            assert relative_eq(g, w, spec["eps"]), f"{what}: got {gv!r} want {wv!r} (eps {spec['eps']})"
        else:
            raise ValueError(mode)


def check_contact(c, spec, what):
    for field, s in spec.items():
        check_value(c[field], s, f"{what}.{field}")


def run_contacts_case(case, contacts_fn):
    cs = contacts_fn(case["a"], case["vel_a"], case["b"], case["vel_b"])
    cid = case["id"]
    if "returns" in case:
        assert (len(cs) > 0) == case["returns"], f"{cid}: returned {len(cs) > 0}"
    if "count" in case:
        assert len(cs) == case["count"], f"{cid}: {len(cs)} contacts, want {case['count']}: {cs}"
    if "every" in case:
        assert len(cs) > 0, cid
        for k, c in enumerate(cs):
            check_contact(c, case["every"], f"{cid}[{k}]")
    if "last" in case:
        assert len(cs) > 0, f"{cid}: no contact"
        check_contact(cs[-1], case["last"], f"{cid}[last]")
    if "index" in case:
        for k, spec in case["index"].items():
            assert len(cs) > int(k), f"{cid}: only {len(cs)} contacts"
            check_contact(cs[int(k)], spec, f"{cid}[{k}]")
