"""Test helpers: build the same scene in the CPU oracle and in the HIP world."""
import numpy as np

from oracle import oracle as O


def oracle_world(scene, order=O.ORDER_CANONICAL):
    w = O.World(order)
    t = scene["terrain"]
    if t is not None:
        w.set_terrain(t["verts"], t["faces"], t["pos"])
    if len(scene["comps"]):
        w.add_bodies(scene["comps"], scene["mass"], scene["restitution"], scene["friction"], scene["force"])
    cb = scene.get("compound")
    if cb is not None:
        w.add_compound_bodies(cb["comps"], cb["comp_mass"], cb["offsets"], cb["restitution"], cb["friction"], cb["force"])
    if scene.get("v0") is not None:
        w.set_state(v=scene["v0"])
    return w


def rel_err(a, b):
    """max |a-b| / max(1, |b|) elementwise — the 1e-4 relative f32 bar of BASELINE.json."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    if a.size == 0:
        return 0.0
    return float(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b))))


def bits_equal(a, b):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


def values_equal(a, b):
    """f32 equality treating +0 == -0 (numerically identical results)."""
    return np.array_equal(np.asarray(a, np.float32), np.asarray(b, np.float32))


CONSTRAINT_FIELDS = ["normal", "t0", "t1", "ra", "rb", "bias", "normal_mass", "tangent_mass0", "tangent_mass1", "friction"]


def compare_constraints(got, want, check_impulse=False):
    assert len(got) == len(want), f"{len(got)} constraints, oracle has {len(want)}"
    assert np.array_equal(got["a"], want["a"]), "obj_a order differs"
    assert np.array_equal(got["b"], want["b"]), "obj_b order differs"
    fields = CONSTRAINT_FIELDS + (["normal_impulse"] if check_impulse else [])
    for f in fields:
        if not values_equal(got[f], want[f]):
            d = np.abs(got[f].astype(np.float64) - want[f].astype(np.float64))
            raise AssertionError(f"constraint field {f} differs: max abs diff {d.max()} at {np.unravel_index(d.argmax(), d.shape)}")


def two_kinds_tile_scenes(P):
    """tiles for the tests of ghost records of two widths (r06): every two-part body left of x = 0, every plain sphere right of it, driven into each
    other - cut into P x-slabs of [-12, 12) (P = 2: one kind per tile; P = 4: two tiles of two-part bodies, one of spheres, one empty)"""
    from mgf_amd import scenes
    sc = scenes.dumbbell_field(3, 2, 4, n_plain=24)
    cb = sc["compound"]
    cb["comps"]["p"][:, 0] -= np.float32(cb["comps"]["p"][:, 0].max() + 1.2)
    n_plain = len(sc["comps"])
    sc["comps"]["p"][:, 0] = np.float32(0.8) + np.float32(0.9) * (np.arange(n_plain) % 4).astype(np.float32)
    sc["comps"]["p"][:, 1] = np.float32(1.0) + np.float32(1.1) * (np.arange(n_plain) // 4).astype(np.float32)
    sc["comps"]["p"][:, 2] = np.float32(-2.0) + np.float32(1.3) * (np.arange(n_plain) % 3).astype(np.float32)
    sc["v0"][:n_plain] = np.float32([-2.0, 0.0, 0.0])
    sc["v0"][n_plain:] = np.float32([3.0, 0.0, 0.0])
    return scenes.split_by_slabs(sc, P, 12.0)
