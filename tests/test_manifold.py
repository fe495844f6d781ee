"""ContactPruner::push + Manifold::from(pruner) (manifold.rs:42-148): hand-derived cases on the oracle, the batched HIP
entry point mgf_manifolds_from_contacts against the oracle bit for bit.  (The reference has no test for these: parity
unpinned, the authority is the source text.)"""
import numpy as np
import pytest

from oracle import oracle as O


def _lc(local_a, local_b, a, b, n, t):
    r = np.zeros(1, O.LOCAL_CONTACT_DTYPE)
    r["local_a"], r["local_b"], r["a"], r["b"], r["n"], r["t"] = local_a, local_b, a, b, n, t
    return r


def test_pruner_hand_cases():
    far = _lc((1, 0, 0), (-1, 0, 0), (0, 0, 0), (0, 0, 0), (0, 1, 0), 0.5)
    near_small = _lc((0.2, 0, 0), (-0.2, 0, 0), (0.3, 0, 0), (0.3, 0, 0), (0, 0, 1), 0.5)       # within sqrt(0.5) of `far`, closer to the centres
    near_big = _lc((2, 0, 0), (-2, 0, 0), (0.3, 0, 0), (0.3, 0, 0), (0, 0, 1), 0.5)            # same place, farther from the centres
    apart = _lc((0, 1, 0), (0, -1, 0), (3, 0, 0), (3, 0, 0), (1, 0, 0), 0.5)
    earlier = _lc((0, 0, 1), (0, 0, -1), (9, 9, 9), (9, 9, 9), (0, 0, 1), 0.25)
    later = _lc((0, 0, 1), (0, 0, -1), (5, 5, 5), (5, 5, 5), (0, 0, 1), 0.75)
    m = O.manifold_from_contacts(np.concatenate([far, near_small]))
    assert m["n"] == 1 and np.array_equal(m["pairs"][0], [1, 0, 0, -1, 0, 0]) and np.array_equal(m["normal"], [0, 1, 0])   # merged, old one kept
    m = O.manifold_from_contacts(np.concatenate([far, near_big]))
    assert m["n"] == 1 and np.array_equal(m["pairs"][0], [2, 0, 0, -2, 0, 0]) and np.array_equal(m["normal"], [0, 0, 1])   # merged, new one wins
    m = O.manifold_from_contacts(np.concatenate([far, apart, later]))
    assert m["n"] == 2 and m["time"] == np.float32(0.5) and np.array_equal(m["normal"], [0.5, 0.5, 0])                   # un-renormalised mean; later dropped
    m = O.manifold_from_contacts(np.concatenate([far, apart, earlier]))
    assert m["n"] == 1 and m["time"] == np.float32(0.25) and np.array_equal(m["pairs"][0], [0, 0, 1, 0, 0, -1])          # earlier time clears the set
    m = O.manifold_from_contacts(np.zeros(0, O.LOCAL_CONTACT_DTYPE))
    assert m["n"] == 0 and np.isnan(m["normal"]).all() and np.isinf(m["time"])                                            # 0 / 0
    # tangents are compute_basis of the mean normal
    m = O.manifold_from_contacts(far)
    assert np.array_equal(m["t0"], [0, 0, -1]) and np.array_equal(m["t1"], [-1, 0, 0])


@pytest.mark.gpu
def test_hip_manifolds_match_oracle_bitwise():
    import mgf_amd
    ctx = mgf_amd.Context(0)
    rng = np.random.default_rng(31)
    groups, offsets = [], [0]
    for g in range(3000):
        k = int(rng.integers(0, 9))
        lc = np.zeros(k, O.LOCAL_CONTACT_DTYPE)
        centre = rng.uniform(-2, 2, 3)
        lc["a"] = centre + rng.normal(size=(k, 3)) * rng.choice([0.2, 1.5])      # clustered or spread: merges and keeps
        lc["b"] = lc["a"] + rng.normal(size=(k, 3)) * 0.05
        lc["local_a"] = rng.uniform(-1, 1, (k, 3)); lc["local_b"] = rng.uniform(-1, 1, (k, 3))
        nrm = rng.normal(size=(k, 3)); lc["n"] = nrm / np.linalg.norm(nrm, axis=1, keepdims=True) if k else nrm
        lc["t"] = rng.choice([0.0, 0.25, 0.25 + 5e-7, 0.5], k)                    # ties inside and outside the 1e-6 window
        groups.append(lc); offsets.append(offsets[-1] + k)
    allc = np.concatenate(groups)
    try:
        got = mgf_amd.manifolds_from_contacts(ctx, offsets, allc)
        overflowed = False
    except mgf_amd.MgfError as e:
        assert e.status == 7
        overflowed = True
    assert not overflowed   # groups hold at most 8 contacts: MGF_MANIFOLD_CAP is never exceeded
    kept = 0
    for g, lc in enumerate(groups):
        want = O.manifold_from_contacts(lc)
        m = got[g]
        assert m["n_contacts"] == want["n"], g
        assert np.array_equal(m["time"].view(np.uint32), np.float32(want["time"]).view(np.uint32))
        for f, w in (("normal", want["normal"]), ("tangent", np.stack([want["t0"], want["t1"]]))):
            a, b = np.float32(m[f]), np.float32(w)
            if np.isnan(b).any():   # an empty group: 0 / 0 - NaN on both sides (payload and sign are not specified)
                assert np.array_equal(np.isnan(a), np.isnan(b)), (g, f, a, b)
            else:
                assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (g, f, a, b)
        for k in range(want["n"]):
            assert np.array_equal(np.concatenate([m["local_a"][k], m["local_b"][k]]), want["pairs"][k]), (g, k)
        kept += want["n"]
    assert kept > 3000
    ctx.close()
