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
    scene = scenes.capsule_field(8, 2, 8)
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
