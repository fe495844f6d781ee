"""BVH::raytrace (bvh.rs:345-369) and Intersects for Ray / Segment (collision.rs:169-373): the CPU oracle against
hand-derived answers and a brute-force scan; the HIP path against the oracle, bit for bit."""
import numpy as np
import pytest

from oracle import oracle as O

INF = float("inf")


def _rand_shapes(rng, n):
    """shape dicts + oracle shapes of mixed kinds"""
    dicts, oshapes = [], []
    for i in range(n):
        k = i % 4
        if k == 0:
            c, r = rng.uniform(-4, 4, 3), rng.uniform(0.3, 2.0)
            dicts.append(dict(kind="sphere", c=c, r=r)); oshapes.append(O.shape(O.SPHERE, tuple(c), r))
        elif k == 1:
            a, d, r = rng.uniform(-4, 4, 3), rng.uniform(-2, 2, 3), rng.uniform(0.3, 1.5)
            dicts.append(dict(kind="capsule", a=a, d=d, r=r)); oshapes.append(O.shape(O.CAPSULE, tuple(a), tuple(d), r))
        elif k == 2:
            a = rng.uniform(-4, 4, 3); b = a + rng.uniform(-3, 3, 3); c = a + rng.uniform(-3, 3, 3)
            dicts.append(dict(kind="triangle", a=a, b=b, c=c)); oshapes.append(O.shape(O.TRIANGLE, tuple(a), tuple(b), tuple(c)))
        else:
            nrm = rng.normal(size=3); nrm = (nrm / np.linalg.norm(nrm)).astype(np.float32); d = rng.uniform(-3, 3)
            dicts.append(dict(kind="plane", n=nrm, d=d)); oshapes.append(O.shape(O.PLANE, tuple(nrm), d))
    return dicts, oshapes


def _rand_particles(rng, n):
    p = rng.uniform(-8, 8, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32) * rng.uniform(0.2, 6.0, (n, 1)).astype(np.float32)
    d[::7, 1] = 0.0  # axis-parallel components exercise the |d| < epsilon slab branch
    dt = np.where(np.arange(n) % 3 == 0, np.float32(1.0), np.float32(np.inf)).astype(np.float32)
    return p, d, dt


def test_ray_aabb_hand_cases():
    # straight down onto the top face
    assert O.intersection_aabb((0, 5, 0), (0, -1, 0), INF, (0, 0, 0), (1, 1, 1)) == ((0.0, 1.0, 0.0), 4.0)
    # origin inside the box: t_min stays 0
    assert O.intersection_aabb((0.2, 0.1, 0), (1, 0, 0), INF, (0, 0, 0), (1, 1, 1)) == ((0.20000000298023224, 0.10000000149011612, 0.0), 0.0)
    # parallel to a slab and outside it
    assert O.intersection_aabb((0, 2, 0), (1, 0, 0), INF, (0, 0, 0), (1, 1, 1)) is None
    # a Segment (DT = 1) that stops short / reaches
    assert O.intersection_aabb((0, 5, 0), (0, -3, 0), 1.0, (0, 0, 0), (1, 1, 1)) is None
    assert O.intersection_aabb((0, 5, 0), (0, -8, 0), 1.0, (0, 0, 0), (1, 1, 1)) == ((0.0, 1.0, 0.0), 0.5)
    # pointing away: the slabs never overlap for t >= 0
    assert O.intersection_aabb((0, 5, 0), (0, 1, 0), INF, (0, 0, 0), (1, 1, 1)) is None


def test_ray_plane_and_triangle_hand_cases():
    pl = O.shape(O.PLANE, (0, 1, 0), 2.0)
    assert O.intersection((0, 5, 0), (0, -1, 0), INF, pl) == ((0.0, 2.0, 0.0), 3.0)
    assert O.intersection((0, 5, 0), (1, 0, 0), INF, pl) is None          # denom == 0
    assert O.intersection((0, 1, 0), (0, -1, 0), INF, pl) is None         # t <= 0
    tri = O.shape(O.TRIANGLE, (0, 0, 0), (1, 0, 0), (0, 0, 1))
    hit = O.intersection((0.25, 3, 0.25), (0, -1, 0), INF, tri)
    assert hit == ((0.25, 0.0, 0.25), 3.0)
    assert O.intersection((0.75, 3, 0.75), (0, -1, 0), INF, tri) is None  # plane hit outside the face (u + v >= 1)


def test_raytrace_matches_brute_force_over_leaves():
    rng = np.random.default_rng(5)
    b = O.Bvh()
    boxes = []
    for i in range(200):
        c, r = rng.uniform(-20, 20, 3).astype(np.float32), rng.uniform(0.2, 2.0, 3).astype(np.float32)
        b.insert(tuple(c), tuple(r), i)
        boxes.append((c, r))
    p, d, dt = _rand_particles(rng, 60)
    for k in range(60):
        got = b.raytrace(tuple(p[k]), tuple(d[k]), float(dt[k]))
        want = {}
        for i, (c, r) in enumerate(boxes):
            h = O.intersection_aabb(tuple(p[k]), tuple(d[k]), float(dt[k]), tuple(c), tuple(r))
            if h is not None:
                want[i] = h
        assert {v for v, _, _ in got} == set(want), k
        assert len(got) == len(want)
        for v, ip, t in got:
            assert (ip, t) == want[v]


@pytest.mark.gpu
def test_hip_intersections_match_oracle_bitwise():
    import mgf_amd
    ctx = mgf_amd.Context(0)
    rng = np.random.default_rng(9)
    n = 4000
    dicts, oshapes = _rand_shapes(rng, n)
    p, d, dt = _rand_particles(rng, n)
    parts = np.zeros(n, mgf_amd.PARTICLE_DTYPE)
    parts["p"], parts["d"], parts["dt"] = p, d, dt
    got = mgf_amd.intersections(ctx, parts, shapes=dicts)
    hits = 0
    for i in range(n):
        want = O.intersection(tuple(p[i]), tuple(d[i]), float(dt[i]), oshapes[i])
        assert (got[i] is None) == (want is None), i
        if want is not None:
            hits += 1
            assert np.array_equal(np.float32(got[i][0]), np.float32(want[0])) and np.float32(got[i][1]) == np.float32(want[1]), (i, got[i], want)
    assert hits > n // 10
    boxes = np.concatenate([rng.uniform(-6, 6, (n, 3)), rng.uniform(1.0, 5.0, (n, 3))], axis=1).astype(np.float32)
    got = mgf_amd.intersections(ctx, parts, boxes=boxes)
    hits = 0
    for i in range(n):
        want = O.intersection_aabb(tuple(p[i]), tuple(d[i]), float(dt[i]), tuple(boxes[i, :3]), tuple(boxes[i, 3:]))
        assert (got[i] is None) == (want is None), i
        if want is not None:
            hits += 1
            assert np.array_equal(np.float32(got[i][0]), np.float32(want[0])) and np.float32(got[i][1]) == np.float32(want[1])
    assert hits > n // 40
    ctx.close()


@pytest.mark.gpu
def test_hip_bvh_raytrace_matches_oracle_order_and_bits():
    import mgf_amd
    ctx = mgf_amd.Context(0)
    rng = np.random.default_rng(13)
    gb, ob = mgf_amd.Bvh(ctx), O.Bvh()
    ids = []
    for i in range(600):
        c, r = rng.uniform(-25, 25, 3).astype(np.float32), rng.uniform(0.2, 2.5, 3).astype(np.float32)
        ids.append((gb.insert(c, r, i), ob.insert(tuple(c), tuple(r), i)))
        if i % 9 == 8:  # removals reshape the tree (rotations, LIFO slot reuse) - both sides identically
            g_id, o_id = ids.pop(int(rng.integers(len(ids))))
            gb.remove(g_id); ob.remove(o_id)
    p, d, dt = _rand_particles(rng, 300)
    parts = np.zeros(300, mgf_amd.PARTICLE_DTYPE)
    parts["p"], parts["d"], parts["dt"] = p, d, dt
    off, vals, inter = gb.raytrace_many(parts)
    total = 0
    for k in range(300):
        want = ob.raytrace(tuple(p[k]), tuple(d[k]), float(dt[k]))
        lo, hi = off[k], off[k + 1]
        assert hi - lo == len(want), k
        assert list(vals[lo:hi]) == [v for v, _, _ in want], f"particle {k}: visiting order"
        for j, (_, ip, t) in enumerate(want):
            assert np.array_equal(inter["p"][lo + j], np.float32(ip)) and inter["t"][lo + j] == np.float32(t)
        total += len(want)
    assert total > 100
    ctx.close()
