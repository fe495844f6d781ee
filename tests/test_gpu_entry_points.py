"""Entry points of include/mgf_hip.h that no other test reached ("exported but untested ABI"): the bulk and
callback forms of the BVH queries, the single-shot LocalContacts calls, Mesh::push_vert / push_face one by one, the
zero-copy device pointers and the ghost count.  Each against the oracle, bit for bit."""
import ctypes as C

import numpy as np
import pytest

import mgf_amd
from mgf_amd import scenes
from oracle import oracle as O
from tests.util import bits_equal, oracle_world, values_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = mgf_amd.Context(0)
    yield c
    c.close()


def _trees(ctx, rng, n):
    gb, ob = mgf_amd.Bvh(ctx), O.Bvh()
    for i in range(n):
        c, r = rng.uniform(-20, 20, 3).astype(np.float32), rng.uniform(0.2, 2.5, 3).astype(np.float32)
        assert gb.insert(c, r, 1000 + i) == ob.insert(tuple(c), tuple(r), 1000 + i)
    return gb, ob


def test_bvh_query_many_equals_the_callback_form_and_the_oracle(ctx):
    """BVH::query (bvh.rs:283-310): the bulk form lists every query's hits in the reference's DFS order."""
    rng = np.random.default_rng(21)
    gb, ob = _trees(ctx, rng, 300)
    boxes = np.concatenate([rng.uniform(-20, 20, (200, 3)), rng.uniform(0.5, 6, (200, 3))], axis=1).astype(np.float32)
    boxes[7, 3:] = 100.0  # everything
    boxes[8, :3], boxes[8, 3:] = 500.0, 0.1  # nothing
    off, vals = gb.query_many(boxes)
    assert off[0] == 0 and off[-1] == len(vals) and len(off) == len(boxes) + 1
    for k, bx in enumerate(boxes):
        want = ob.query(tuple(bx[:3]), tuple(bx[3:]))
        assert vals[off[k]:off[k + 1]].tolist() == want, f"query {k}"
        if k % 25 == 0:
            assert gb.query(bx[:3], bx[3:]) == want
    assert off[9] - off[8] == 0 and off[8] - off[7] == 300


def test_bvh_raytrace_callback_form(ctx):
    """BVH::raytrace (bvh.rs:345-369) one particle at a time through the callback, as the reference's signature has it."""
    rng = np.random.default_rng(22)
    gb, ob = _trees(ctx, rng, 200)
    n_hit = 0
    for k in range(120):
        p, d = rng.uniform(-25, 25, 3).astype(np.float32), rng.normal(size=3).astype(np.float32)
        dt = float("inf") if k % 2 == 0 else 1.0
        if dt == 1.0:
            d *= np.float32(30.0)
        got, want = gb.raytrace(p, d, dt), ob.raytrace(tuple(p), tuple(d), dt)
        assert [g[0] for g in got] == [w[0] for w in want]
        for g, w in zip(got, want):
            assert values_equal(g[1], w[1]) and values_equal([g[2]], [w[2]])
        n_hit += len(want)
    assert n_hit > 20


def test_single_shot_local_contacts_pair_and_mesh(ctx):
    """LocalContacts<Moving<Component>> for Moving<Component> (compound.rs:192-207) and LocalContacts<Mesh>
    (collision.rs:1490-1506 over mesh.rs:115-139) called once per pair, like world.rs:241,277 does."""
    rng = np.random.default_rng(23)
    n_hit = 0
    for k in range(300):
        ta, tb = int(rng.integers(0, 2)), int(rng.integers(0, 2))
        pa, pb = rng.uniform(-1.5, 1.5, 3).astype(np.float32), rng.uniform(-1.5, 1.5, 3).astype(np.float32)
        da = rng.uniform(-1, 1, 3).astype(np.float32) if ta else np.zeros(3, np.float32)
        db = rng.uniform(-1, 1, 3).astype(np.float32) if tb else np.zeros(3, np.float32)
        ra, rb = float(rng.uniform(0.3, 0.9)), float(rng.uniform(0.3, 0.9))
        va, vb = rng.uniform(-1, 1, 3).astype(np.float32), rng.uniform(-1, 1, 3).astype(np.float32)
        got = mgf_amd.local_contacts_pair(ctx, (ta, pa, da, ra, va), (tb, pb, db, rb, vb))
        want = O.local_contacts_pair(O.component(ta, pa, da, ra), va, O.component(tb, pb, db, rb), vb)
        assert len(got) == len(want), k
        for g, w in zip(got, want):
            for f in ("local_a", "local_b", "a", "b", "n"):
                assert values_equal(g[f], w[f]), (k, f)
            assert values_equal([g["t"]], [w["t"]])
        n_hit += len(want)
    assert n_hit > 30
    # body against the terrain mesh: the oracle world's own terrain contacts for a few resting capsules
    scene = scenes.capsule_field_dense(6, 2, 6)
    gw, ow = mgf_amd.World.from_scene(ctx, scene), oracle_world(scene)
    dt = float(scene["dt"])
    for _ in range(45):
        gw.step(dt, 10)
        ow.step(dt, 10)
    t = scene["terrain"]
    mesh = mgf_amd.Mesh(ctx)
    mesh.build(t["verts"], t["faces"])
    mesh.set_pos(t["pos"])
    col = gw.colliders()
    n_hit = 0
    for i in range(len(col)):
        c = col[i]
        got = mgf_amd.local_contacts_mesh(ctx, (int(c["tag"]), c["p"], c["d"], float(c["r"]), c["delta"]), mesh)
        want = ow.terrain_contacts(i)
        assert len(got) == len(want), i
        for g, w in zip(got, want):
            for f in ("local_a", "local_b", "a", "b", "n"):
                assert values_equal(g[f], w[f]), (i, f)
        n_hit += len(want)
    assert n_hit > 10


def test_mesh_pushed_vertex_by_vertex_equals_the_bulk_build(ctx):
    """Mesh::push_vert / push_face (mesh.rs:58-73): the face BVH grows by insert + balance per face, so the one-by-one
    mesh and the bulk one serialise to the same tree, and a world steps identically on either."""
    t = scenes.heightfield_terrain(6, 6, 12.0, 12.0, 0.2)
    a, b = mgf_amd.Mesh(ctx), mgf_amd.Mesh(ctx)
    a.build(t["verts"], t["faces"])
    for k, v in enumerate(t["verts"]):
        assert b.push_vert(v) == k
    for k, f in enumerate(t["faces"]):
        assert b.push_face(int(f[0]), int(f[1]), int(f[2])) == k
    a.set_pos(t["pos"])
    b.set_pos(t["pos"])
    assert a.to_json() == b.to_json()


def test_device_pointers_keep_their_meaning_across_ticks_of_a_large_world(ctx):
    """A world of >= 16 384 bodies keeps its store in an internal order that the fused tick re-sorts.  Raw pointers
    handed out by mgf_world_device_ptr are indexed by the CALLER's body index: while they are out the store stays in the caller's
    order, so a zero-copy consumer that steps the world and reads through a pointer it already holds sees body i in row i; the
    results are those of a world that never handed pointers out (bit for bit); releasing them lets the world re-sort again."""
    scene = scenes.sphere_pile(26, 26, 26)  # 17 576 bodies
    dt = float(scene["dt"])
    a, b = mgf_amd.World.from_scene(ctx, scene), mgf_amd.World.from_scene(ctx, scene)
    for _ in range(3):
        a.step(dt, 10); b.step(dt, 10)
    assert a.counter("store_permuted") == 1
    n = len(a)
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    px, nbytes = a.device_ptr("x")
    psr, _ = a.device_ptr("solver_rec")
    assert a.counter("store_permuted") == 0
    for k in range(4):  # (resort_every = 1 would re-sort at every one of these ticks)
        a.set_option("resort_every", 1)
        a.step(dt, 10); b.step(dt, 10)
        assert a.counter("store_permuted") == 0
        hx, hs = np.zeros((n, 4), np.float32), np.zeros((n, 16), np.float32)
        assert hip.hipMemcpy(hx.ctypes.data, C.c_void_p(px), nbytes, 2) == 0
        assert hip.hipMemcpy(hs.ctypes.data, C.c_void_p(psr), 64 * n, 2) == 0
        sb = b.state()
        assert bits_equal(hx[:, :3], sb["x"]) and bits_equal(hs[:, :3], sb["v"]) and bits_equal(hs[:, 3:6], sb["omega"])
    assert a.counter("device_ptrs_out") == 1
    a.release_device_ptrs()
    assert a.counter("device_ptrs_out") == 0
    a.step(dt, 10); b.step(dt, 10)
    a.step(dt, 10); b.step(dt, 10)
    assert a.counter("store_permuted") == 1
    sa, sb = a.state(), b.state()
    for f in ("x", "q", "v", "omega"):
        assert bits_equal(sa[f], sb[f])
    # a failed look-up pins nothing; pointers die with the arrays - adding bodies gives the re-sort back without a release
    with pytest.raises(mgf_amd.MgfError):
        a.device_ptr("no_such_array")
    assert a.counter("device_ptrs_out") == 0
    a.device_ptr("q")
    assert a.counter("device_ptrs_out") == 1
    extra = scenes.sphere_pile(2, 1, 2)
    extra["comps"]["p"][:, 1] += 40.0
    a.add_bodies(extra["comps"], extra["mass"], extra["restitution"], extra["friction"], extra["force"])
    assert a.counter("device_ptrs_out") == 0


def test_device_pointers_and_ghost_len(ctx):
    """mgf_world_device_ptr hands out the resident arrays (x, q, solver records, delta) for zero-copy exchange; what is read
    through them is the state the ordinary read-back returns.  mgf_world_ghost_len counts the tick's imported ghosts."""
    import torch
    scene = scenes.sphere_pile(5, 5, 5)
    w = mgf_amd.World.from_scene(ctx, scene)
    for _ in range(5):
        w.step(float(scene["dt"]), 10)
    st, n = w.state(), len(w)
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    for name, per in (("x", 4), ("q", 4), ("solver_rec", 16), ("delta", 4)):
        p, nbytes = w.device_ptr(name)
        assert p and nbytes == 4 * per * n
        host = np.zeros((n, per), np.float32)
        assert hip.hipMemcpy(host.ctypes.data, C.c_void_p(p), nbytes, 2) == 0  # hipMemcpyDeviceToHost
        if name == "x":
            assert bits_equal(host[:, :3], st["x"])
        elif name == "q":
            assert bits_equal(host, st["q"])
        elif name == "delta":
            assert bits_equal(host[:, :3], st["delta"])
        else:
            assert bits_equal(host[:, :3], st["v"]) and bits_equal(host[:, 3:6], st["omega"])
    with pytest.raises(mgf_amd.MgfError):
        w.device_ptr("nonsense")
    assert w.ghost_len() == 0
    # ghosts: export three bodies of a second world and import them here
    w2 = mgf_amd.World.from_scene(ctx, scenes.sphere_pile(4, 4, 4))
    w2.begin_tick(float(scene["dt"]))
    ids = torch.tensor([1, 5, 9], dtype=torch.int32, device="cuda")
    recs = torch.zeros((3, 72), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()  # (the fill runs on torch's stream, the library writes on its own)
    w2.export_bodies(ids.data_ptr(), 3, recs.data_ptr())
    w.begin_tick(float(scene["dt"]))
    w.import_ghosts(recs.data_ptr(), 3)
    assert w.ghost_len() == 3 and len(w) == n
    w.collide(float(scene["dt"]))
    w.solve_enqueue(2)
    w.finish()
    torch.cuda.synchronize()


def test_pair_contacts_at_the_edge_of_reach_survive_the_conservative_reject(ctx):
    """The narrowphase rejects a pair whose bounding spheres never come within reach during the tick before it runs the
    reference's tests (comp_pair_far, dev_geom.h).  That reject must never drop a contact: pairs built so that the shapes just
    touch at the very end of the sweep (t close to 1), graze each other sideways, or sit exactly at the distance where the
    bounding spheres meet - compared with the oracle, which has no such reject."""
    rng = np.random.default_rng(77)
    n_hit = n_late = 0
    for k in range(600):
        ta, tb = int(rng.integers(0, 2)), int(rng.integers(0, 2))
        ra, rb = float(rng.uniform(0.2, 0.9)), float(rng.uniform(0.2, 0.9))
        da = (rng.normal(size=3) * rng.uniform(0.1, 1.5)).astype(np.float32) if ta else np.zeros(3, np.float32)
        db = (rng.normal(size=3) * rng.uniform(0.1, 1.5)).astype(np.float32) if tb else np.zeros(3, np.float32)
        pa = rng.uniform(-50, 50, 3).astype(np.float32)
        # the reach of the two bounding spheres, and a relative motion along the line of centres that closes a gap of about its own length
        Ra, Rb = ra + 0.5 * float(np.linalg.norm(da)), rb + 0.5 * float(np.linalg.norm(db))
        u = rng.normal(size=3); u /= np.linalg.norm(u)
        speed = float(rng.uniform(0.0, 3.0))
        mode = k % 3
        if mode == 0:    # head-on: centres start a little inside / outside Ra + Rb + speed
            gap = (Ra + Rb + speed) * float(rng.uniform(0.9, 1.02))
            side = np.zeros(3)
        elif mode == 1:  # grazing: offset sideways by about the sum of the radii
            w = np.cross(u, rng.normal(size=3)); w /= np.linalg.norm(w)
            gap = speed * float(rng.uniform(0.2, 1.0))
            side = w * (ra + rb) * float(rng.uniform(0.8, 1.05))
        else:            # resting: no motion to speak of, surfaces a hair apart or overlapping
            speed = float(rng.uniform(0.0, 1e-3))
            gap = (ra + rb) * float(rng.uniform(0.7, 1.3))
            side = np.zeros(3)
        ma = pa + 0.5 * da                      # bounding-sphere centres
        mb = ma + u * gap + side
        pb = (mb - 0.5 * db).astype(np.float32)
        vb = (-u * speed * rng.uniform(0.5, 1.0)).astype(np.float32)
        va = (u * speed * rng.uniform(0.0, 0.5)).astype(np.float32)
        got = mgf_amd.local_contacts_pair(ctx, (ta, pa, da, ra, va), (tb, pb, db, rb, vb))
        want = O.local_contacts_pair(O.component(ta, pa, da, ra), va, O.component(tb, pb, db, rb), vb)
        assert len(got) == len(want), (k, mode, ta, tb)
        for g, w_ in zip(got, want):
            for f in ("local_a", "local_b", "a", "b", "n"):
                assert values_equal(g[f], w_[f]), (k, f)
            assert values_equal([g["t"]], [w_["t"]])
            n_late += 1 if w_["t"] > 0.8 else 0
        n_hit += len(want)
    assert n_hit > 100 and n_late > 10, (n_hit, n_late)


def test_handles_outlive_a_destroyed_context():
    """mgf_ctx_destroy only drops the creator's reference (r06: an interpreter's finalisation freed a context ahead of its worlds, and
    mgf_world_free read a deleted struct): a world and its mesh freed AFTER their context - straight through the C-ABI, the order a garbage
    collector may choose - and every other call on them refused."""
    lib = mgf_amd._capi.load_library()
    c = mgf_amd.Context(0)
    sc = scenes.capsule_field(6, 4, 6, quads=8, pitch=1.6)
    w = mgf_amd.World.from_scene(c, sc)
    m = mgf_amd.Mesh(c)
    for _ in range(5):
        w.step(float(sc["dt"]), sc["iters"])   # (the second stream and its events exist)
    n = len(w)
    h = c._h
    c._h = None                                # (the wrapper must not close the children first: that is the order under test)
    lib.mgf_ctx_destroy(h)
    with pytest.raises(mgf_amd.MgfError) as e:
        w.step(float(sc["dt"]), sc["iters"])
    assert "destroyed" in str(e.value)
    assert len(w) == n                         # (host-side queries still answer)
    del w, m                                   # mgf_world_free, mgf_mesh_free: the last one takes the context's streams along
    c2 = mgf_amd.Context(0)                    # ... and the device is as usable as before
    w2 = mgf_amd.World.from_scene(c2, sc)
    assert w2.step(float(sc["dt"]), sc["iters"]).n_constraints >= 0
    c2.close()
