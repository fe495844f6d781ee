"""Scene I/O: BVH<AABB, usize> and Mesh in the shape serde_json gives the reference's types (bvh.rs:29-47,
pool.rs:25-41, mesh.rs:31-37).  Host-side plumbing: everything but the last test runs without a GPU."""
import json
import os

import numpy as np
import pytest

import mgf_amd
from mgf_amd import scenes
from oracle import oracle as O

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _grow(rng, trees, n_ops):
    """the same random insert / remove sequence on every tree in `trees` (lib and oracle)"""
    ids = []
    for i in range(n_ops):
        c, r = rng.uniform(-20, 20, 3).astype(np.float32), rng.uniform(0.1, 2.0, 3).astype(np.float32)
        got = [t.insert(tuple(c), tuple(r), i) if isinstance(t, O.Bvh) else t.insert(c, r, i) for t in trees]
        assert len(set(int(g) for g in got)) == 1
        ids.append(int(got[0]))
        if i % 4 == 3:
            victim = ids.pop(int(rng.integers(len(ids))))
            for t in trees:
                t.remove(victim)
    return ids


def _f32_tree(v):
    """parsed JSON with every float squeezed to f32 (the oracle view holds exact f32 values already)"""
    if isinstance(v, dict):
        return {k: _f32_tree(x) for k, x in v.items()}
    if isinstance(v, list):
        return [_f32_tree(x) for x in v]
    if isinstance(v, float):
        return float(np.float32(v))
    return v


def test_fixture_text_round_trips_byte_for_byte():
    """A hand-written file in serde_json's text conventions (1.0, 0.25, -1.5e-7, null, externally tagged enums)."""
    text = open(os.path.join(GOLDEN, "bvh_two_leaves.json")).read()
    b = mgf_amd.Bvh.from_json(None, text)
    assert b.to_json() == text
    nodes, boxes = b.dump()
    assert nodes[2].tolist() == [1, 0, 0, 0, 0, 1] and nodes[1].tolist() == [1, -1, 2, 1, 9, 0]
    assert boxes[1].tolist() == [4.0, 0.5, float(np.float32(-1.5e-7)), 1.0, 0.25, 1.0]


def test_bvh_json_matches_the_oracles_tree_and_round_trips():
    rng = np.random.default_rng(17)
    lib_tree, ora_tree = mgf_amd.Bvh(None), O.Bvh()
    _grow(rng, [lib_tree, ora_tree], 120)
    text = lib_tree.to_json()
    assert _f32_tree(json.loads(text)) == _f32_tree(ora_tree.serde())       # entry for entry, free list included
    again = mgf_amd.Bvh.from_json(None, text)
    assert again.to_json() == text
    # the restored pool hands out the same slots: keep growing both and compare
    rng2a, rng2b = np.random.default_rng(99), np.random.default_rng(99)
    _grow(rng2a, [lib_tree], 40)
    _grow(rng2b, [again], 40)
    assert again.to_json() == lib_tree.to_json()


def test_empty_and_single_leaf_trees():
    b = mgf_amd.Bvh(None)
    assert b.to_json() == '{"root":0,"pool":{"len":0,"free_list":null,"entries":[]}}'
    assert mgf_amd.Bvh.from_json(None, b.to_json()).empty()
    b.insert((1, 2, 3), (1, 1, 1), 5)
    c = mgf_amd.Bvh.from_json(None, b.to_json())
    assert c.root() == b.root() == 0 and c.get_leaf(0) == 5


@pytest.mark.parametrize("text,why", [
    ("{\"root\":0,\"pool\":{\"len\":1,\"free_list\":null,\"entries\":[]}}", "len does not match"),
    ("{\"root\":3,\"pool\":{\"len\":1,\"free_list\":null,\"entries\":[{\"Occupied\":{\"height\":-1,\"parent\":0,\"bounds\":{\"c\":{\"x\":0,\"y\":0,\"z\":0},\"r\":{\"x\":1,\"y\":1,\"z\":1}},\"node_type\":{\"Leaf\":1}}}]}}", "root"),
    ("{\"root\":0,\"pool\":{\"len\":1,\"free_list\":null,\"entries\":[{\"Occupied\":{\"height\":0,\"parent\":0,\"bounds\":{\"c\":{\"x\":0,\"y\":0,\"z\":0},\"r\":{\"x\":1,\"y\":1,\"z\":1}},\"node_type\":{\"Parent\":[5,6]}}}]}}", "child link"),
    ("{\"root\":0,\"pool\":{\"len\":0,\"free_list\":null,\"entries\":[\"Oops\"]}}", "unit variant"),
    ("{\"root\":0,\"pool\":", "end of input"),
])
def test_damaged_files_are_rejected(text, why):
    with pytest.raises(mgf_amd.MgfError) as e:
        mgf_amd.Bvh.from_json(None, text)
    assert e.value.status == 6 and why in str(e.value)


def _occ(parent, kind, val, height=0):
    return {"Occupied": {"height": height, "parent": parent, "bounds": {"c": {"x": 0, "y": 0, "z": 0}, "r": {"x": 1, "y": 1, "z": 1}},
                         "node_type": {kind: val}}}


def _tree(root, entries, free_list=None, n_occupied=None):
    n = sum(1 for e in entries if isinstance(e, dict) and "Occupied" in e) if n_occupied is None else n_occupied
    return json.dumps({"root": root, "pool": {"len": n, "free_list": free_list, "entries": entries}})


@pytest.mark.parametrize("text,why", [
    # a descendant lists the root as its child and the root names it as its parent: a ring through the root
    (_tree(0, [_occ(1, "Parent", [1, 2]), _occ(0, "Parent", [0, 3]), _occ(0, "Leaf", 7, -1), _occ(1, "Leaf", 8, -1)]), "root is listed as a child"),
    # both children are the same entry
    (_tree(0, [_occ(0, "Parent", [1, 1]), _occ(0, "Leaf", 7, -1)]), "both children"),
    # a ring of two parents next to a sound tree: every link is mutual, nothing reaches it from the root
    (_tree(0, [_occ(0, "Leaf", 1, -1), _occ(2, "Parent", [2, 3]), _occ(1, "Parent", [1, 4]), _occ(1, "Leaf", 5, -1), _occ(2, "Leaf", 6, -1)]),
     "do not form one tree"),
    # free list: a ring of FreeListPtr entries / a chain that runs into an occupied entry
    (_tree(0, [_occ(0, "Leaf", 1, -1), {"FreeListPtr": {"next_free": 2}}, {"FreeListPtr": {"next_free": 1}}], free_list=1), "free list has a cycle"),
    (_tree(0, [_occ(0, "Leaf", 1, -1), {"FreeListPtr": {"next_free": 0}}], free_list=1), "occupied entry"),
])
def test_rings_and_broken_free_lists_are_rejected(text, why):
    """links that pass the pairwise checks but do not form a tree (host walks would never end)."""
    with pytest.raises(mgf_amd.MgfError) as e:
        mgf_amd.Bvh.from_json(None, text)
    assert e.value.status == 6 and why in str(e.value), str(e.value)


def _terrain_mesh(ctx, terrain):
    m = mgf_amd.Mesh(ctx)
    m.build(terrain["verts"], terrain["faces"])
    m.set_pos(terrain["pos"])
    return m


def test_mesh_json_round_trip_host_only():
    t = scenes.heightfield_terrain(12, 12, 30.0, 30.0, 0.2)
    m = _terrain_mesh(None, t)
    text = m.to_json()
    v = json.loads(text)
    assert list(v) == ["x", "verts", "faces", "bvh"] and len(v["verts"]) == len(t["verts"]) and len(v["faces"]) == len(t["faces"])
    assert v["faces"][5] == [int(k) for k in t["faces"][5]] and v["x"] == dict(x=float(t["pos"][0]), y=float(t["pos"][1]), z=float(t["pos"][2]))
    m2 = mgf_amd.Mesh.from_json(None, text)
    assert m2.to_json() == text


@pytest.mark.gpu
def test_world_steps_identically_on_a_deserialised_terrain():
    ctx = mgf_amd.Context(0)
    scene = scenes.capsule_field_dense(8, 2, 8)
    a = mgf_amd.World.from_scene(ctx, scene)
    b = mgf_amd.World(ctx)
    b.set_terrain(mgf_amd.Mesh.from_json(ctx, _terrain_mesh(ctx, scene["terrain"]).to_json()))
    b.add_bodies(scene["comps"], scene["mass"], scene["restitution"], scene["friction"], scene["force"])
    if scene.get("v0") is not None:
        b.write_state(v=scene["v0"])
    for _ in range(40):
        sa, sb = a.step(float(scene["dt"]), 10), b.step(float(scene["dt"]), 10)
        assert (sa.n_constraints, sa.n_terrain_constraints) == (sb.n_constraints, sb.n_terrain_constraints)
    assert sa.n_terrain_constraints > 0
    s1, s2 = a.state(), b.state()
    for k in s1:
        assert np.array_equal(s1[k].view(np.uint32), s2[k].view(np.uint32)), k
    ctx.close()


# ---- the geometry structs of geom.rs:31-357 (Sphere, Capsule, Triangle, Plane, Rectangle, Ray, Segment, AABB, Moving<T>) ----
GEOM_FIXTURES = [
    # hand-written in serde_json's text conventions: field order = declaration order, 1.0 not 1, exponents without "+"
    ("sphere", dict(kind="sphere", c=[1.0, -2.5, 0.125], r=0.5), '{"c":{"x":1.0,"y":-2.5,"z":0.125},"r":0.5}'),
    ("capsule", dict(kind="capsule", a=[-0.5, 0.0, 0.0], d=[1.0, 0.0, 0.0], r=1.0),
     '{"a":{"x":-0.5,"y":0.0,"z":0.0},"d":{"x":1.0,"y":0.0,"z":0.0},"r":1.0}'),
    ("triangle", dict(kind="triangle", a=[0.0, 0.0, 0.0], b=[1.0, 0.0, 0.0], c=[0.0, 0.0, 1e21]),
     '{"a":{"x":0.0,"y":0.0,"z":0.0},"b":{"x":1.0,"y":0.0,"z":0.0},"c":{"x":0.0,"y":0.0,"z":1e21}}'),
    ("plane", dict(kind="plane", n=[0.0, 1.0, 0.0], d=-10.0), '{"n":{"x":0.0,"y":1.0,"z":0.0},"d":-10.0}'),
    ("rectangle", dict(kind="rectangle", c=[0.0, -10.0, 0.0], u0=[1.0, 0.0, 0.0], u1=[0.0, 0.0, 1.0], e=[20.0, 20.0]),
     '{"c":{"x":0.0,"y":-10.0,"z":0.0},"u":[{"x":1.0,"y":0.0,"z":0.0},{"x":0.0,"y":0.0,"z":1.0}],"e":[20.0,20.0]}'),
    ("ray", dict(kind="ray", p=[0.0, 3.0, 0.0], d=[0.0, -1.0, 0.0]), '{"p":{"x":0.0,"y":3.0,"z":0.0},"d":{"x":0.0,"y":-1.0,"z":0.0}}'),
    ("segment", dict(kind="segment", a=[0.0, 0.0, 0.0], b=[0.0, 0.0, -1.5e-7]),
     '{"a":{"x":0.0,"y":0.0,"z":0.0},"b":{"x":0.0,"y":0.0,"z":-1.5e-7}}'),
    ("aabb", dict(kind="aabb", c=[4.0, 0.5, 0.0], r=[1.0, 0.25, 1.0]), '{"c":{"x":4.0,"y":0.5,"z":0.0},"r":{"x":1.0,"y":0.25,"z":1.0}}'),
]


@pytest.mark.parametrize("kind,shape,text", GEOM_FIXTURES)
def test_geom_structs_in_serde_json_shape(kind, shape, text):
    assert mgf_amd.geom_to_json(shape) == text
    back = mgf_amd.geom_from_json(kind, text)
    assert mgf_amd.geom_to_json(back) == text
    for k, v in shape.items():
        if k != "kind":
            assert np.array_equal(np.asarray(back[k], np.float32), np.asarray(v, np.float32)), k
    # serde reads a struct's fields in any order and skips unknown ones
    v = json.loads(text)
    shuffled = json.dumps(dict(list(reversed(list(v.items()))) + [("comment", [1, {"a": None}])]))
    assert mgf_amd.geom_to_json(mgf_amd.geom_from_json(kind, shuffled)) == text


def test_moving_is_a_two_element_sequence():
    """Moving<T>(pub T, pub Vector3<f32>) (geom.rs:356-357): a tuple struct serialises as an array."""
    text = '[{"c":{"x":0.0,"y":5.0,"z":0.0},"r":0.5},{"x":0.0,"y":-0.16333334,"z":0.0}]'
    sphere = dict(kind="sphere", c=[0.0, 5.0, 0.0], r=0.5)
    assert mgf_amd.geom_to_json(sphere, moving=[0.0, np.float32(-9.8) * np.float32(1 / 60), 0.0]) == text
    shape, vel = mgf_amd.geom_from_json("sphere", text, moving=True)
    assert shape["c"] == [0.0, 5.0, 0.0] and np.float32(vel[1]) == np.float32(-9.8) * np.float32(1 / 60)
    cap = '[{"a":{"x":-0.5,"y":0.0,"z":0.0},"d":{"x":1.0,"y":0.0,"z":0.0},"r":1.0},{"x":0.0,"y":0.0,"z":0.0}]'
    shape, vel = mgf_amd.geom_from_json("capsule", cap, moving=True)
    assert mgf_amd.geom_to_json(shape, moving=vel) == cap


def test_random_geom_floats_round_trip_exactly():
    """shortest round-trip float text: every f32 survives write -> read bit for bit"""
    rng = np.random.default_rng(5)
    for _ in range(200):
        vals = (rng.standard_normal(7) * 10.0 ** rng.integers(-6, 7, 7)).astype(np.float32)
        cap = dict(kind="capsule", a=vals[0:3].tolist(), d=vals[3:6].tolist(), r=float(vals[6]))
        back = mgf_amd.geom_from_json("capsule", mgf_amd.geom_to_json(cap))
        got = np.array(back["a"] + back["d"] + [back["r"]], np.float32)
        assert np.array_equal(got.view(np.uint32), vals.view(np.uint32))


@pytest.mark.parametrize("kind,text,why", [
    ("sphere", '{"c":{"x":0.0,"y":0.0,"z":0.0}}', "missing field `r`"),
    ("sphere", '{"c":{"x":0.0,"y":0.0,"z":0.0},"r":1.0,"r":2.0}', "duplicate field `r`"),
    ("sphere", '{"c":[0.0,0.0,0.0],"r":1.0}', "field `c` has the wrong shape"),
    ("capsule", '{"a":{"x":0.0,"y":0.0},"d":{"x":1.0,"y":0.0,"z":0.0},"r":1.0}', "field `a` has the wrong shape"),
    ("rectangle", '{"c":{"x":0.0,"y":0.0,"z":0.0},"u":[{"x":1.0,"y":0.0,"z":0.0}],"e":[1.0,1.0]}', "field `u` has the wrong shape"),
    ("plane", '[1.0]', "expected a map"),
    ("ray", '{"p":{"x":0.0,"y":0.0,"z":0.0},"d":{"x":0.0,"y":1.0,"z":0.0}} x', "trailing characters"),
    ("segment", '{"a":{"x":0.0,"y":0.0,"z":0.0},"b":', "end of input"),
])
def test_damaged_geom_text_is_rejected(kind, text, why):
    with pytest.raises(mgf_amd.MgfError) as e:
        mgf_amd.geom_from_json(kind, text)
    assert e.value.status == 6 and why in str(e.value), str(e.value)


def test_moving_needs_both_elements():
    with pytest.raises(mgf_amd.MgfError) as e:
        mgf_amd.geom_from_json("sphere", '[{"c":{"x":0.0,"y":0.0,"z":0.0},"r":1.0}]', moving=True)
    assert "Moving" in str(e.value)
