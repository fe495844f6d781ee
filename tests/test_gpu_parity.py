"""HIP path vs CPU oracle on identical seeded inputs (bit-exact is the expectation; the contract
bar is 1e-4 relative f32, BASELINE.json)."""
import numpy as np
import pytest

from oracle import oracle as O
from tests.util import bits_equal, compare_constraints, oracle_world, rel_err, values_equal

pytestmark = pytest.mark.gpu
TOL = 1e-4  # north_star: post-step positions/velocities within 1e-4 rel f32


@pytest.fixture(scope="module")
def ctx():
    import mgf_amd
    c = mgf_amd.Context(0)
    yield c
    c.close()


def _rand_shape(rng, kind):
    c = rng.uniform(-2, 2, 3).astype(np.float32)
    if kind == "sphere":
        return dict(kind="sphere", c=c.tolist(), r=float(rng.uniform(0.3, 1.2)))
    d = rng.uniform(-1.5, 1.5, 3).astype(np.float32)
    return dict(kind="capsule", a=c.tolist(), d=d.tolist(), r=float(rng.uniform(0.3, 0.9)))


def _same_contacts(got, want, what):
    assert len(got) == len(want), f"{what}: {len(got)} contacts vs oracle {len(want)}\n{got}\n{want}"
    for g, w in zip(got, want):
        for f in ("a", "b", "n", "t"):
            assert values_equal(g[f], w[f]), f"{what}.{f}: {g[f]} vs {w[f]}"


@pytest.mark.parametrize("ka,kb", [("sphere", "sphere"), ("capsule", "sphere"), ("sphere", "capsule"), ("capsule", "capsule")])
def test_random_moving_pairs_bit_exact(ctx, ka, kb):
    import mgf_amd
    rng = np.random.default_rng(hash((ka, kb)) % 2 ** 32)
    probs = []
    for _ in range(3000):
        a, b = _rand_shape(rng, ka), _rand_shape(rng, kb)
        va = rng.uniform(-1, 1, 3).astype(np.float32).tolist()
        vb = rng.uniform(-1, 1, 3).astype(np.float32).tolist()
        probs.append((a, va, b, vb))
    # degenerate cases: coincident centres, zero relative velocity, parallel capsules
    a = _rand_shape(rng, ka)
    probs.append((a, [0, 0, 0], dict(a) if ka == kb else _rand_shape(rng, kb), [0, 0, 0]))
    probs.append((a, [0.1, 0, 0], dict(a) if ka == kb else _rand_shape(rng, kb), [0, 0.2, 0]))
    if ka == kb == "capsule":
        for off in ([0, 0.5, 0], [0, 3.0, 0], [2.0, 0.5, 0], [-3.0, 0.1, 0]):
            p = dict(kind="capsule", a=[0, 0, 0], d=[1, 0, 0], r=0.5)
            q = dict(kind="capsule", a=off, d=[1, 0, 0], r=0.4)
            probs.append((p, [0, 0, 0], q, [0, -1.5, 0]))
            probs.append((p, [0.3, 0, 0], q, [-0.5, -2.5, 0]))
    got = mgf_amd.contacts_batch(ctx, probs)
    nhit = 0
    for i, (p, g) in enumerate(zip(probs, got)):
        want = O.contacts(O.shape_from_dict(p[0]), p[1], O.shape_from_dict(p[2]), p[3])
        _same_contacts(g, want, f"{ka}-{kb}[{i}]")
        nhit += len(want)
    assert nhit > 100  # the sample really exercises contact paths


@pytest.mark.parametrize("kb", ["sphere", "capsule"])
def test_random_triangle_contacts_bit_exact(ctx, kb):
    import mgf_amd
    rng = np.random.default_rng(7 + len(kb))
    probs = []
    for _ in range(4000):
        tri = dict(kind="triangle", a=rng.uniform(-2, 2, 3).tolist(), b=rng.uniform(-2, 2, 3).tolist(), c=rng.uniform(-2, 2, 3).tolist())
        b = _rand_shape(rng, kb)
        probs.append((tri, None, b, rng.uniform(-1.5, 1.5, 3).astype(np.float32).tolist()))
    # axis-aligned floor triangles with capsules parallel to the face / to an edge (the exact-equality branch collision.rs:915)
    floor = dict(kind="triangle", a=[1, 1, 0], b=[0, 1, -1], c=[0, 1, 1])
    for x in np.linspace(-1.5, 1.5, 13):
        for dz in (-2.0, 2.0, -4.0):
            probs.append((floor, None, dict(kind="capsule", a=[float(x), 2.0, 1.0], d=[0, 0, dz], r=1.0), [0, -1, 0]))
            probs.append((floor, None, dict(kind="capsule", a=[float(x), 2.5, 0.3], d=[0.7, 0, dz], r=0.8), [0.1, -1.2, 0]))
    got = mgf_amd.contacts_batch(ctx, probs)
    nhit = 0
    for i, (p, g) in enumerate(zip(probs, got)):
        want = O.contacts(O.shape_from_dict(p[0]), None, O.shape_from_dict(p[2]), p[3])
        _same_contacts(g, want, f"tri-{kb}[{i}]")
        nhit += len(want)
    assert nhit > 100


def _sync_from_oracle(gw, ow):
    s = ow.state()
    gw.write_state(x=s["x"], q=s["q"], v=s["v"], omega=s["omega"], delta=s["delta"])


def _compare_state(gw, ow, what, exact=True):
    g, o = gw.state(), ow.state()
    for k in ("x", "q", "v", "omega", "delta"):
        e = rel_err(g[k], o[k])
        assert e <= TOL, f"{what}: {k} rel err {e}"
        if exact:
            assert values_equal(g[k], o[k]), f"{what}: {k} not bit-identical (rel err {e})"


@pytest.mark.parametrize("scene_name", ["balls8", "pile12", "pile16_noshuffle", "capsules", "mixed"])
def test_world_step_parity_teacher_forced(ctx, scene_name):
    """Per-step parity from identical snapshots: constraint list, then post-step state."""
    import mgf_amd
    from mgf_amd import scenes
    scene = {"balls8": lambda: scenes.balls_demo(8), "pile12": lambda: scenes.sphere_pile(12, 12, 12),
             "pile16_noshuffle": lambda: scenes.sphere_pile(16, 8, 16, shuffle=False),
             "capsules": lambda: scenes.capsule_field_dense(8, 3, 8),                       # Capsule-Capsule, Capsule-Triangle
             "mixed": lambda: scenes.capsule_field_dense(8, 3, 8, sphere_fraction=0.5),     # all four pair types + binning
             }[scene_name]()
    dt, iters = float(scene["dt"]), scene["iters"]
    ow = oracle_world(scene)
    gw = mgf_amd.World.from_scene(ctx, scene)
    gw.set_option("solver_mode", 0)  # launch-per-frontier path: n_levels is then the depth of the unrolled graph
    _compare_state(gw, ow, "initial")
    # let the oracle run the scene forward; at chosen steps teacher-force the GPU from the oracle snapshot
    checkpoints = {0, 1, 2, 5, 20, 60, 140, 141, 142, 170, 200} if scene_name == "balls8" else {0, 1, 2, 3, 10, 25, 40}
    if scene_name in ("capsules", "mixed"):
        checkpoints = {0, 10, 20, 30, 31, 45, 60, 75, 90, 120}
    last = max(checkpoints)
    total_constraints = 0
    for step in range(last + 1):
        if step in checkpoints:
            _sync_from_oracle(gw, ow)
            snap = ow.state()
            # constraint list parity (insertion order, every field)
            ow.build_constraints(dt)
            st = gw.build_constraints(dt)
            oc, gc = ow.constraints(), gw.constraints()
            compare_constraints(gc, oc)
            assert st.n_constraints == len(oc)
            total_constraints += len(oc)
            # finish the tick on both
            ow.solve(iters)
            gw.solve(iters)
            _compare_state(gw, ow, f"{scene_name} step {step}")
            compare_constraints(gw.constraints(), ow.constraints(), check_impulse=True)
            assert gw.stats.n_levels == ow.constraint_depth(iters)  # launches = depth of the unrolled graph
        else:
            ow.step(dt, iters)
    assert total_constraints > 0


def test_world_free_running_matches_oracle(ctx):
    """No teacher forcing: 60 consecutive ticks of the 512-ball demo stay bit-identical."""
    import mgf_amd
    from mgf_amd import scenes
    scene = scenes.sphere_pile(10, 10, 10)
    dt, iters = float(scene["dt"]), scene["iters"]
    ow = oracle_world(scene)
    gw = mgf_amd.World.from_scene(ctx, scene)
    for step in range(60):
        so = ow.step(dt, iters)
        sg = gw.step(dt, iters)
        assert sg.n_constraints == so.n_constraints, f"step {step}"
        assert sg.n_pair_candidates == so.n_pair_candidates, f"step {step}"
        assert sg.n_refits == so.n_refits, f"step {step}"
    _compare_state(gw, ow, "free-running 60 ticks")


def test_demo_order_deviation_is_reported(ctx):
    """The HIP path emits canonical order; the demo (world.rs) order differs only in the order of a
    body's partners.  Same constraint SET; the state deviation is reported, not hidden."""
    import mgf_amd
    from mgf_amd import scenes
    scene = scenes.sphere_pile(8, 8, 8)
    dt, iters = float(scene["dt"]), scene["iters"]
    od = oracle_world(scene, O.ORDER_DEMO)
    gw = mgf_amd.World.from_scene(ctx, scene)
    for _ in range(5):
        od.step(dt, iters)
    s = od.state()
    gw.write_state(x=s["x"], q=s["q"], v=s["v"], omega=s["omega"], delta=s["delta"])
    od.build_constraints(dt)
    gw.build_constraints(dt)
    oc, gc = od.constraints(), gw.constraints()
    key = lambda c: sorted(zip(c["a"].tolist(), c["b"].tolist()))
    assert key(oc) == key(gc)
    od.solve(iters)
    gw.solve(iters)
    dev = rel_err(gw.state()["v"], od.state()["v"])
    print(f"demo-vs-canonical order deviation after one solve: {dev:.3e}")
    assert dev < 0.5


def test_standalone_solver_on_user_constraints(ctx):
    """Solver::add_constraint (bulk) + solve on an arbitrary insertion order (not the world's)."""
    import mgf_amd
    from mgf_amd import scenes
    scene = scenes.sphere_pile(8, 8, 8)
    dt, iters = float(scene["dt"]), scene["iters"]
    ow = oracle_world(scene)
    gw = mgf_amd.World.from_scene(ctx, scene)
    for _ in range(3):
        ow.step(dt, iters)
    _sync_from_oracle(gw, ow)
    ow.build_constraints(dt)
    gw.build_constraints(dt)
    cons = gw.constraints()
    # the world's own list, re-submitted through the bulk Solver API
    gw.set_constraints(cons)
    gw.solve(iters)
    ow.solve(iters)
    _compare_state(gw, ow, "set_constraints round trip")


def _closed_form_names():
    from tests.test_oracle_solver import CASES
    return sorted(CASES)


@pytest.mark.parametrize("name", _closed_form_names())
def test_closed_form_cases_hip(ctx, name):
    """The hand-derived solver cases of tests/test_oracle_solver.py, on the HIP path."""
    import mgf_amd
    from tests.test_oracle_solver import CASES
    CASES[name](lambda scene: mgf_amd.World.from_scene(ctx, scene))


@pytest.mark.parametrize("refresh_every,P", [(1, 2), (2, 2), (3, 2), (10, 2), (2, 3), (1, 4)])
def test_tiled_worlds_match_oracle_tiles(ctx, refresh_every, P):
    """Two x-slab tiles (ghost export/import kernels, ghost filtering in the broadphase, ghost velocity
    refresh every R solver iterations) on the GPU vs the oracle's tile mode: bit-identical per tile."""
    import mgf_amd
    from mgf_amd import scenes
    from mgf_amd.tiles import HipEngine, Tile, step_tiles_inprocess
    from tests.oracle_engine import OracleEngine
    nx, ny, nz = (8, 6, 8) if P == 2 else (5, 5, 6)
    gt, ot = [], []
    for r in range(P):
        sc = scenes.sphere_pile_tile(nx, ny, nz, r, P)
        gt.append(Tile(HipEngine(ctx, sc, 0), sc["x_range"], r, P, sc["dt"], sc["iters"], refresh_every=refresh_every))
        ot.append(Tile(OracleEngine(sc), sc["x_range"], r, P, sc["dt"], sc["iters"], refresh_every=refresh_every))
    for tick in range(12):
        sg = step_tiles_inprocess(gt)
        so = step_tiles_inprocess(ot)
        for r in range(P):
            assert sg[r]["n_constraints"] == so[r]["n_constraints"], f"tick {tick} tile {r}"
            assert tuple(gt[r].e.counts) == (len(ot[r].e.ids[0]), len(ot[r].e.ids[1]))
    assert gt[0].e.counts[1] > 0 and gt[1].e.counts[0] > 0
    if P > 2:
        assert gt[1].e.counts[0] > 0 and gt[1].e.counts[1] > 0  # an interior tile exports both ways
    for r in range(P):
        g, o = gt[r].e.state(), ot[r].e.state()
        for k in ("x", "q", "v", "omega", "delta"):
            assert values_equal(g[k], o[k]), f"tile {r} {k}: rel err {rel_err(g[k], o[k])}"


def _crowded_scene(big_first=False):
    """One big sphere (last index, or first) touched by 80 small ones: more than the 32-entry candidate row."""
    from mgf_amd import scenes
    rng = np.random.default_rng(11)
    d = rng.normal(size=(80, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[:, 1] = np.abs(d[:, 1])
    centres = np.concatenate([(d * 3.45 + [0, 4.0, 0]).astype(np.float32), np.array([[0, 4.0, 0]], np.float32)])
    if big_first:
        centres = centres[::-1].copy()
    comps = np.zeros(len(centres), scenes.COMPONENT_DTYPE)
    comps["p"] = centres
    comps["r"] = 0.5
    comps["r"][0 if big_first else -1] = 3.0
    sc = scenes._scene("crowded", comps, scenes.box_terrain(12.0, 12.0, (0, 0, 0)), v0=rng.uniform(-0.2, 0.2, (len(comps), 3)))
    return sc


@pytest.mark.parametrize("force_two_pass", [0, 1])
def test_candidate_row_overflow_falls_back_exactly(ctx, force_two_pass):
    import mgf_amd
    scene = _crowded_scene()
    dt, iters = float(scene["dt"]), scene["iters"]
    ow = oracle_world(scene)
    gw = mgf_amd.World.from_scene(ctx, scene)
    gw.set_option("two_pass_candidates", force_two_pass)
    for step in range(6):
        so = ow.step(dt, iters)
        sg = gw.step(dt, iters)
        assert sg.n_constraints == so.n_constraints and sg.n_pair_candidates == so.n_pair_candidates
        compare_constraints(gw.constraints(), ow.constraints(), check_impulse=True)
    assert so.n_pair_candidates >= 60  # the big sphere alone has > 32 partners
    _compare_state(gw, ow, "crowded scene")


def test_fused_sphere_filter_equals_the_separate_narrowphase(ctx):
    """A world of spheres runs the sphere-sphere test inside the grid broadphase and lists contacts only; the candidate
    statistic, the constraints and the state must equal the path that lists every accepted partner."""
    import mgf_amd
    from mgf_amd import scenes
    scene = scenes.sphere_pile(12, 10, 12)
    dt, iters = float(scene["dt"]), scene["iters"]
    a, b = mgf_amd.World.from_scene(ctx, scene), mgf_amd.World.from_scene(ctx, scene)
    b.set_option("no_fused_narrowphase", 1)
    ow = oracle_world(scene)
    for step in range(25):
        sa, sb, so = a.step(dt, iters), b.step(dt, iters), ow.step(dt, iters)
        assert (sa.n_constraints, sa.n_pair_candidates, sa.n_terrain_candidates) == (sb.n_constraints, sb.n_pair_candidates, sb.n_terrain_candidates)
        assert sa.n_pair_candidates == so.n_pair_candidates and sa.n_constraints == so.n_constraints
    assert sa.n_pair_candidates > 4 * (sa.n_constraints - sa.n_terrain_constraints) > 0  # most accepted partners are not contacts
    compare_constraints(a.constraints(), b.constraints(), check_impulse=True)
    _compare_state(a, ow, "fused sphere filter")


def test_body_in_many_constraints_as_b_widens_its_row(ctx):
    """The big sphere has index 0, so it is body `b` of every contact with the 80 small ones: its row of `b`
    occurrences (16 ids to start with) overflows, the tick is re-run with wider rows, and the result is the oracle's."""
    import mgf_amd
    scene = _crowded_scene(big_first=True)
    dt, iters = float(scene["dt"]), scene["iters"]
    ow = oracle_world(scene)
    gw = mgf_amd.World.from_scene(ctx, scene)
    assert gw.counter("rev_row_capacity") == 16
    for step in range(6):
        so, sg = ow.step(dt, iters), gw.step(dt, iters)
        assert sg.n_constraints == so.n_constraints
        compare_constraints(gw.constraints(), ow.constraints(), check_impulse=True)
    assert so.n_constraints - so.n_terrain_constraints >= 60
    assert gw.counter("rev_row_capacity") >= 64
    _compare_state(gw, ow, "crowded scene, big sphere first")


@pytest.mark.parametrize("mode", [0, 1])
def test_list_capacity_miss_reruns_collide_exactly(ctx, mode):
    """The tick is enqueued with speculative list capacities; a miss must re-run the collide phase and
    leave exactly the result of a run that never missed (solver_mode 0 takes the read-back path)."""
    import mgf_amd
    from mgf_amd import scenes
    scene = scenes.sphere_pile(10, 10, 10)
    dt, iters = float(scene["dt"]), scene["iters"]
    a, b = mgf_amd.World.from_scene(ctx, scene), mgf_amd.World.from_scene(ctx, scene)
    a.set_option("solver_mode", mode); b.set_option("solver_mode", mode)
    for step in range(30):
        if step % 3 == 0:
            b.set_option("list_capacity", 7 + step)   # far too small once contacts exist
        sa, sb = a.step(dt, iters), b.step(dt, iters)
        assert (sa.n_constraints, sa.n_pair_candidates, sa.n_terrain_candidates) == (sb.n_constraints, sb.n_pair_candidates, sb.n_terrain_candidates)
    assert sa.n_constraints > 1000
    s1, s2 = a.state(), b.state()
    for k in s1:
        assert bits_equal(s1[k], s2[k]), k
    compare_constraints(a.constraints(), b.constraints(), check_impulse=True)


@pytest.mark.parametrize("scene_name", ["pile16", "capsules", "mixed"])
def test_grid_and_tree_broadphase_agree(ctx, scene_name):
    """Default broadphase = direct enumeration of Morton cells (k_pair_grid); option broadphase_tree walks the
    4-ary tree (k_pair_rows).  Same acceptance predicate, so the same candidates and the same tick."""
    import mgf_amd
    from mgf_amd import scenes
    scene = {"pile16": lambda: scenes.sphere_pile(16, 16, 16), "capsules": lambda: scenes.capsule_field_dense(10, 3, 10),
             "mixed": lambda: scenes.capsule_field_dense(8, 3, 8, sphere_fraction=0.5)}[scene_name]()
    a, b = mgf_amd.World.from_scene(ctx, scene), mgf_amd.World.from_scene(ctx, scene)
    b.set_option("broadphase_tree", 1)
    for _ in range(30):
        sa, sb = a.step(float(scene["dt"]), 10), b.step(float(scene["dt"]), 10)
        assert (sa.n_constraints, sa.n_pair_candidates, sa.n_terrain_candidates) == (sb.n_constraints, sb.n_pair_candidates, sb.n_terrain_candidates)
    assert sa.n_pair_candidates > 0
    s1, s2 = a.state(), b.state()
    for k in s1:
        assert bits_equal(s1[k], s2[k]), k


def test_terrain_face_grid_and_tree_walk_agree(ctx):
    """Terrain candidates: cell enumeration over the face boxes + DFS-rank sort (k_terrain_grid) vs the reference-order
    walk of the mesh BVH (k_terrain_rows): same faces in the same order, so the same tick."""
    import mgf_amd
    from mgf_amd import scenes
    scene = scenes.capsule_field_dense(24, 3, 24, quads=40)
    a, b = mgf_amd.World.from_scene(ctx, scene), mgf_amd.World.from_scene(ctx, scene)
    b.set_option("terrain_tree", 1)
    assert a.counter("terrain_grid") == 1 and b.counter("terrain_grid") == 0
    for _ in range(60):
        sa, sb = a.step(float(scene["dt"]), 10), b.step(float(scene["dt"]), 10)
        assert (sa.n_constraints, sa.n_terrain_constraints, sa.n_terrain_candidates) == (sb.n_constraints, sb.n_terrain_constraints, sb.n_terrain_candidates)
    assert sa.n_terrain_constraints > 100 and a.counter("terrain_grid") == 1
    compare_constraints(a.constraints(), b.constraints(), check_impulse=True)
    s1, s2 = a.state(), b.state()
    for k in s1:
        assert bits_equal(s1[k], s2[k]), k


def test_terrain_face_grid_on_a_mesh_that_is_not_flat(ctx):
    """The face grid's cells per axis follow the mesh (build_face_grid): the heightfield above is flat and gets one layer of cells; a
    craggy one - heights of +-6 over 0.9-wide quads, faces as tall as they are wide - gets cells along all three axes, and a slab of it
    offset far from the origin exercises the early return for bodies outside the mesh's bounds.  Grid and tree walk must agree, and
    both with the oracle's first ticks."""
    import mgf_amd
    from mgf_amd import scenes
    from tests.util import oracle_world
    scene = scenes.capsule_field_dense(20, 4, 20, quads=40, sphere_fraction=0.5)
    scene = dict(scene, terrain=scenes.heightfield_terrain(40, 40, 36.0, 36.0, 6.0, pos=(0.0, -7.0, 0.0)))
    a, b, o = mgf_amd.World.from_scene(ctx, scene), mgf_amd.World.from_scene(ctx, scene), oracle_world(scene)
    b.set_option("terrain_tree", 1)
    assert a.counter("terrain_grid") == 1 and b.counter("terrain_grid") == 0
    dt = float(scene["dt"])
    for tick in range(70):
        sa, sb = a.step(dt, 10), b.step(dt, 10)
        assert (sa.n_constraints, sa.n_terrain_constraints, sa.n_terrain_candidates) == (sb.n_constraints, sb.n_terrain_constraints, sb.n_terrain_candidates), tick
        if tick < 45:
            so = o.step(dt, 10)
            assert (sa.n_constraints, sa.n_terrain_constraints, sa.n_pair_candidates) == (so.n_constraints, so.n_terrain_constraints, so.n_pair_candidates), tick
        if tick == 44:
            g, w_ = a.state(), o.state()
            for k in ("x", "q", "v", "omega"):
                assert bits_equal(g[k], w_[k]), f"tick {tick}: {k} differs from the oracle"
    assert sa.n_terrain_constraints > 100 and a.counter("terrain_grid") == 1
    s1, s2 = a.state(), b.state()
    for k in s1:
        assert bits_equal(s1[k], s2[k]), k


def test_terrain_face_grid_with_cells_along_all_three_axes(ctx):
    """Two floors: a small heightfield 12 above a large one.  The mesh is tall compared with its faces, so the face grid cuts the
    vertical axis too (row-major cells, bits per axis from build_face_grid); bodies land on both floors.  Grid vs tree walk vs oracle."""
    import mgf_amd
    from mgf_amd import scenes
    from tests.util import oracle_world
    scene = scenes.capsule_field_dense(16, 3, 16, quads=30, sphere_fraction=0.5)
    lower = scenes.heightfield_terrain(30, 30, 32.0, 32.0, 0.2)
    upper = scenes.heightfield_terrain(12, 12, 12.0, 12.0, 0.2, seed=5)
    uv = upper["verts"].copy()
    uv[:, 1] += 12.0
    terrain = dict(verts=np.concatenate([lower["verts"], uv]).astype(np.float32),
                   faces=np.concatenate([lower["faces"], upper["faces"] + np.uint32(len(lower["verts"]))]).astype(np.uint32),
                   pos=np.float32([0.0, -14.0, 0.0]))
    scene = dict(scene, terrain=terrain)
    a, b, o = mgf_amd.World.from_scene(ctx, scene), mgf_amd.World.from_scene(ctx, scene), oracle_world(scene)
    b.set_option("terrain_tree", 1)
    assert a.counter("terrain_grid") == 1 and b.counter("terrain_grid") == 0
    dt = float(scene["dt"])
    for tick in range(130):
        sa, sb = a.step(dt, 10), b.step(dt, 10)
        assert (sa.n_constraints, sa.n_terrain_constraints, sa.n_terrain_candidates) == (sb.n_constraints, sb.n_terrain_constraints, sb.n_terrain_candidates), tick
        if tick < 100:
            so = o.step(dt, 10)
            assert (sa.n_constraints, sa.n_terrain_constraints) == (so.n_constraints, so.n_terrain_constraints), tick
    y = a.state()["x"][:, 1]
    assert (y > -4.0).sum() > 20 and (y < -10.0).sum() > 100, "bodies rest on both floors"
    assert sa.n_terrain_constraints > 100
    s1, s2 = a.state(), b.state()
    for k in s1:
        assert bits_equal(s1[k], s2[k]), k


def test_two_pass_and_row_paths_agree(ctx):
    import mgf_amd
    from mgf_amd import scenes
    scene = scenes.capsule_field_dense(8, 3, 8, sphere_fraction=0.5)
    a, b = mgf_amd.World.from_scene(ctx, scene), mgf_amd.World.from_scene(ctx, scene)
    b.set_option("two_pass_candidates", 1)
    for _ in range(40):
        sa, sb = a.step(float(scene["dt"]), 10), b.step(float(scene["dt"]), 10)
        assert (sa.n_constraints, sa.n_pair_candidates, sa.n_terrain_candidates) == (sb.n_constraints, sb.n_pair_candidates, sb.n_terrain_candidates)
    s1, s2 = a.state(), b.state()
    for k in s1:
        assert bits_equal(s1[k], s2[k]), k


def test_step_many_equals_single_steps(ctx):
    import mgf_amd
    from mgf_amd import scenes
    scene = scenes.sphere_pile(8, 8, 8)
    dt, iters = float(scene["dt"]), scene["iters"]
    a, b = mgf_amd.World.from_scene(ctx, scene), mgf_amd.World.from_scene(ctx, scene)
    per_tick = a.step_many(dt, iters, 25)
    singles = [b.step(dt, iters).n_constraints for _ in range(25)]
    assert [st.n_constraints for st in per_tick] == singles and singles[-1] > 100
    sa, sb = a.state(), b.state()
    for k in sa:
        assert bits_equal(sa[k], sb[k]), k


@pytest.mark.parametrize("scene_name", ["pile", "mixed", "dumbbells"])
def test_pipelined_step_many_with_capacity_misses(ctx, scene_name):
    """step_many enqueues tick k + 1 before it has read tick k back.  When tick k then fails a capacity check the device
    guard must have turned tick k + 1 into a no-op: the batch gives the statistics and the state of single steps, bit for
    bit, through forced misses (tiny list capacities before every batch) and with the pipeline switched off."""
    import mgf_amd
    from mgf_amd import scenes
    scene = {"pile": lambda: scenes.sphere_pile(10, 10, 10), "mixed": lambda: scenes.capsule_field_dense(8, 3, 8, sphere_fraction=0.5),
             "dumbbells": lambda: scenes.dumbbell_field(6, 4, 6, 60)}[scene_name]()
    dt, iters = float(scene["dt"]), scene["iters"]
    a, b, c = (mgf_amd.World.from_scene(ctx, scene) for _ in range(3))
    c.set_option("pipeline", 0)
    for batch in range(6):
        a.set_option("list_capacity", 5 + batch)  # the first tick of the batch misses; the tick enqueued behind it must not run
        many = a.step_many(dt, iters, 9)
        plain = c.step_many(dt, iters, 9)
        keys = ("n_constraints", "n_terrain_constraints", "n_pair_candidates", "n_terrain_candidates", "n_refits")
        singles = []
        for _ in range(9):
            st = b.step(dt, iters)
            singles.append({key: st[key] for key in keys})
        for key in keys:
            assert [st[key] for st in many] == [st[key] for st in singles] == [st[key] for st in plain], (batch, key)
        sa, sb, sc = a.state(), b.state(), c.state()
        for k in sa:
            assert bits_equal(sa[k], sb[k]) and bits_equal(sc[k], sb[k]), (batch, k)
    assert singles[-1]["n_constraints"] > 100
    assert a.counter("capacity_retries") >= 6


def test_terrain_rows_from_integrate_equal_the_separate_kernel(ctx):
    """mgf_world_step lists a body's terrain faces at the end of k_integrate (the bounds are in registers); the separate
    kernel remains for split ticks, re-runs and tiles.  Same rows, same tick - also when a row overflows and is widened."""
    import mgf_amd
    from mgf_amd import scenes
    scene = scenes.balls_demo(8)
    dt, iters = float(scene["dt"]), scene["iters"]
    a, b = mgf_amd.World.from_scene(ctx, scene), mgf_amd.World.from_scene(ctx, scene)
    b.set_option("no_fused_terrain_rows", 1)
    b.set_option("no_fused_scene_bounds", 1)  # likewise the scene bounds: gathered by k_integrate / by their own kernel
    for step in range(150):
        sa, sb = a.step(dt, iters), b.step(dt, iters)
        for key in ("n_constraints", "n_terrain_constraints", "n_terrain_candidates", "n_pair_candidates"):
            assert sa[key] == sb[key], (step, key)
    assert sa.n_terrain_constraints > 0
    xa, xb = a.state(), b.state()
    for k in xa:
        assert bits_equal(xa[k], xb[k]), k
    compare_constraints(a.constraints(), b.constraints(), check_impulse=True)


def test_body_added_between_begin_and_collide_is_seen(ctx):
    """A split tick (begin_tick, ..., collide) may change the bodies in between; whatever k_integrate gathered on the way
    (scene bounds, terrain rows) must then be dropped.  A sphere added outside the old bounds, touching the floor, gets
    its terrain constraint and its place in the grid exactly as in a world that never fuses."""
    import mgf_amd
    from mgf_amd import scenes
    scene = scenes.balls_demo(6)
    dt, iters = float(scene["dt"]), scene["iters"]
    a, b = mgf_amd.World.from_scene(ctx, scene), mgf_amd.World.from_scene(ctx, scene)
    b.set_option("no_fused_terrain_rows", 1)
    b.set_option("no_fused_scene_bounds", 1)
    extra = np.zeros(2, dtype=scene["comps"].dtype)
    extra["tag"] = 0
    extra["p"] = [(8.5, -9.5, 8.5), (8.5, -8.45, 8.5)]  # in a corner of the box, on the floor and on top of each other
    extra["r"] = 0.5
    for tick in range(40):
        for w in (a, b):
            w.begin_tick(dt)
            if tick == 5:
                w.add_bodies(extra, 1.0, 0.3, 0.6, (0.0, -9.8, 0.0))
        sa, sb = a.collide(dt), b.collide(dt)
        for key in ("n_bodies", "n_constraints", "n_terrain_constraints", "n_pair_candidates"):
            assert sa[key] == sb[key], (tick, key)
        if tick == 5:
            assert sa.n_terrain_constraints >= 1
        a.solve(iters); b.solve(iters)
    xa, xb = a.state(), b.state()
    for k in xa:
        assert bits_equal(xa[k], xb[k]), k


@pytest.mark.parametrize("mode", [5, 6])
def test_block_that_does_not_fit_falls_back_on_the_device(ctx, mode):
    """When a spatial block holds more constraints than its workgroup's LDS layout (forced here by a tiny test limit), a
    device flag turns k_solve_flow5 / k_solve_flow6 into a no-op and k_solve_flow does the work - enqueued behind it, no host round
    trip inside the tick (mode 5), or in a re-run of the tick that the read-back asks for (mode 6) - with the same result, and the
    counter says it happened."""
    import mgf_amd
    from mgf_amd import scenes
    scene = scenes.sphere_pile(10, 10, 10)
    dt, iters = float(scene["dt"]), scene["iters"]
    ow, gw = oracle_world(scene), mgf_amd.World.from_scene(ctx, scene)
    gw.set_option("solver_mode", mode)
    gw.set_option(f"flow{mode}_test_cap", 40)
    for step in range(40):
        so, sg = ow.step(dt, iters), gw.step(dt, iters)
        assert sg.n_constraints == so.n_constraints
    assert so.n_constraints > 500 and gw.counter(f"flow{mode}_fallbacks") > 10
    compare_constraints(gw.constraints(), ow.constraints(), check_impulse=True)
    _compare_state(gw, ow, "stand-by solver")
    gw.set_option(f"flow{mode}_test_cap", 0)  # back to the block-local kernel: same world, same results
    n0 = gw.counter(f"flow{mode}_fallbacks")
    for step in range(10):
        so, sg = ow.step(dt, iters), gw.step(dt, iters)
    assert gw.counter(f"flow{mode}_fallbacks") == n0
    _compare_state(gw, ow, "after the stand-by phase")


def test_mode6_tables_that_do_not_fit_rerun_the_tick_also_in_the_pipelined_loop(ctx):
    """Mode 6 builds its block tables inside the collide phase; when they do not fit (forced by a tiny limit) the tick's
    read-back says so (kFailFlow6), the solver launch has done nothing, and the tick is re-run with the global dataflow solver -
    in mgf_world_step_many the speculative tick behind it is turned into a no-op by the device-side guard first."""
    import mgf_amd
    from mgf_amd import scenes
    scene = scenes.sphere_pile(10, 10, 10)
    dt, iters = float(scene["dt"]), scene["iters"]
    ow, gw = oracle_world(scene), mgf_amd.World.from_scene(ctx, scene)
    gw.step_many(dt, iters, 6)
    for _ in range(6):
        ow.step(dt, iters)
    gw.set_option("flow6_test_cap", 40)
    got = [int(s.n_constraints) for s in gw.step_many(dt, iters, 24)]
    want = [int(ow.step(dt, iters).n_constraints) for _ in range(24)]
    assert got == want and gw.counter("flow6_fallbacks") >= 20 and gw.counter("flow6_fail_reason") & 2
    _compare_state(gw, ow, "re-run ticks")
    gw.set_option("flow6_test_cap", 0)
    n0 = gw.counter("flow6_fallbacks")
    got = [int(s.n_constraints) for s in gw.step_many(dt, iters, 10)]
    want = [int(ow.step(dt, iters).n_constraints) for _ in range(10)]
    assert got == want and gw.counter("flow6_fallbacks") == n0
    _compare_state(gw, ow, "after the re-run phase")


def test_mode6_chosen_after_the_collide_phase_keeps_its_stand_by(ctx):
    """A caller that builds the constraints in another solver mode and switches to 6 before Solver::solve: the tables are built at
    solve time, too late for a re-run of the collide phase, so the guarded stand-by launch (and its links) stay behind the solver."""
    import mgf_amd
    from mgf_amd import scenes
    scene = scenes.sphere_pile(9, 9, 9)
    dt, iters = float(scene["dt"]), scene["iters"]
    ow, gw = oracle_world(scene), mgf_amd.World.from_scene(ctx, scene)
    for cap in (0, 40):  # the block-local kernel itself, then its stand-by
        for _ in range(12):
            ow.step(dt, iters)
            gw.set_option("solver_mode", 1)
            gw.build_constraints(dt)
            gw.set_option("solver_mode", 6)
            gw.set_option("flow6_test_cap", cap)
            gw.solve(iters)
        _compare_state(gw, ow, f"mode 6 after the collide phase, test cap {cap}")
    assert gw.counter("flow6_fallbacks") >= 10


@pytest.mark.parametrize("mode", [1, 4, 5, 105, 6, 106, 206])
@pytest.mark.parametrize("scene_name", ["pile12", "mixed", "balls8"])
def test_dataflow_solver_matches_oracle(ctx, scene_name, mode):
    """solver_mode=1 (one persistent dataflow launch) must give the sequential Gauss-Seidel result too."""
    import mgf_amd
    from mgf_amd import scenes
    scene = {"pile12": lambda: scenes.sphere_pile(12, 12, 12), "mixed": lambda: scenes.capsule_field_dense(8, 3, 8, sphere_fraction=0.5),
             "balls8": lambda: scenes.balls_demo(8)}[scene_name]()
    dt, iters = float(scene["dt"]), scene["iters"]
    ow = oracle_world(scene)
    gw = mgf_amd.World.from_scene(ctx, scene)
    gw.set_option("solver_mode", mode % 100)
    if mode >= 100:  # block-local solvers with small blocks: many block faces on a small scene
        gw.set_option("flow5_block", 96 if mode < 200 else 40)
    if mode == 206:  # ... and hardly any foreign slots: ticks that need more run on the stand-by, the others on k_solve_flow6
        gw.set_option("flow6_fcap", 24)
    n_ticks = 160 if scene_name == "balls8" else 60
    for step in range(n_ticks):
        so = ow.step(dt, iters)
        sg = gw.step(dt, iters)
        assert sg.n_constraints == so.n_constraints, f"step {step}"
        if step % 20 == 0 or step == n_ticks - 1:
            compare_constraints(gw.constraints(), ow.constraints(), check_impulse=True)
            _compare_state(gw, ow, f"dataflow {scene_name} step {step}")
