"""Everything at once (round 6): worlds that mix what the rounds added one by one - spheres and capsules, bodies of 2, 3, 4, 7 and 16 components,
a box or a heightfield (few faces: rows; many: the face grid), static obstacles, a body that runs away (the wide list), a store that is
re-sorted, either block-local or global solver, single steps and step_many - drawn at random per seed and stepped against the oracle: the
counts of every tick, the constraint list in insertion order with its impulses every few ticks, the state at the end.  The features have their
own tests; this one is for what happens BETWEEN them (the r06 front end switching on and off, pools beside slots, lists beside rows)."""
import numpy as np
import pytest

from mgf_amd import scenes
from tests.util import compare_constraints, oracle_world, rel_err, values_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import mgf_amd
    c = mgf_amd.Context(0)
    yield c
    c.close()


def _clump(rng, centre, n_parts):
    k = np.zeros(n_parts, scenes.COMPONENT_DTYPE)
    k["tag"] = (rng.random(n_parts) < 0.4).astype(np.int32)
    spread = 0.35 + 0.12 * np.sqrt(n_parts)
    k["p"] = (centre + rng.normal(0.0, spread, (n_parts, 3))).astype(np.float32)
    k["d"] = (rng.normal(0.0, 0.45, (n_parts, 3)) * (k["tag"][:, None] == 1)).astype(np.float32)
    k["r"] = rng.uniform(0.12, 0.3, n_parts).astype(np.float32)
    return k


def _scene(seed):
    rng = np.random.default_rng(9000 + seed)
    terrain_kind = seed % 3
    if terrain_kind == 0:
        terrain = scenes.box_terrain(11.0, 30.0, (0.0, 0.0, 0.0))
    else:
        q = 4 if terrain_kind == 1 else 14          # 32 faces: the rows of k_integrate's tail; 392: the face grid
        terrain = scenes.heightfield_terrain(q, q, 26.0, 26.0, 0.25, seed=1000 + seed)
    n_plain = int(rng.integers(30, 70)) if seed % 5 else 0
    cells = [(i, j, k) for j in range(4) for i in range(6) for k in range(6)]
    rng.shuffle(cells)
    centres = np.array([((i - 2.5) * 2.1, 1.6 + 2.2 * j, (k - 2.5) * 2.1) for i, j, k in cells], np.float64) + rng.uniform(-0.2, 0.2, (len(cells), 3))
    pc = centres[:n_plain]
    plain = np.zeros(n_plain, scenes.COMPONENT_DTYPE)
    caps = rng.random(n_plain) < (0.0 if seed % 4 == 0 else 0.5)
    plain["tag"] = caps.astype(np.int32)
    d = rng.normal(0.0, 1.0, (n_plain, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    length = rng.uniform(0.4, 1.2, (n_plain, 1))
    plain["d"] = (d * length * caps[:, None]).astype(np.float32)
    plain["p"] = (pc - 0.5 * plain["d"]).astype(np.float32)
    plain["r"] = rng.uniform(0.25, 0.5, n_plain).astype(np.float32)
    sc = scenes._scene(f"kitchen_sink_{seed}", plain, terrain)
    sizes_pool = [[2], [2, 3, 4], [2, 4, 7, 16], [16, 16, 3], []][seed % 5]
    nb = (int(rng.integers(12, 30)) if n_plain else int(rng.integers(45, 70))) if sizes_pool else 0
    v0 = [rng.normal(0.0, 1.0, (n_plain, 3))]
    if nb:
        comps, offsets, masses = [], [0], []
        for b in range(nb):
            npart = int(rng.choice(sizes_pool))
            comps.append(_clump(rng, centres[n_plain + b], npart))
            offsets.append(offsets[-1] + npart); masses.append(rng.uniform(0.2, 1.0, npart).astype(np.float32))
        sc["compound"] = dict(comps=np.concatenate(comps), comp_mass=np.concatenate(masses), offsets=np.asarray(offsets, np.int64),
                              restitution=np.full(nb, 0.3, np.float32), friction=np.full(nb, 0.6, np.float32), force=np.tile(np.float32([0.0, -9.8, 0.0]), (nb, 1)))
        v0.append(rng.normal(0.0, 1.0, (nb, 3)))
    v0 = np.concatenate(v0).astype(np.float32)
    if seed % 3 == 1 and len(v0) > 4:
        v0[int(rng.integers(0, len(v0)))] = np.float32([260.0, 30.0, -90.0])   # a runaway: the wide list
    sc["v0"] = v0
    obstacles = []
    if seed % 4 == 2:
        a = np.zeros(3, scenes.COMPONENT_DTYPE)
        a["tag"] = [1, 0, 0]
        a["p"] = [(-4.0, 0.5, 0.0), (2.0, 0.8, 2.0), (-1.0, 0.6, -3.0)]
        a["d"] = [(8.0, 0.8, 0.0), (0, 0, 0), (0, 0, 0)]
        a["r"] = [0.4, 0.9, 0.7]
        obstacles.append((a, (0.3, 0.1, -0.2), (float(np.cos(0.15)), 0.0, float(np.sin(0.15)), 0.0)))
    return sc, obstacles, rng


@pytest.mark.parametrize("seed", range(15))
def test_mixed_worlds_against_the_oracle(ctx, seed):
    import mgf_amd
    sc, obstacles, rng = _scene(seed)
    dt, iters = float(sc["dt"]), sc["iters"]
    gw, ow = mgf_amd.World.from_scene(ctx, sc), oracle_world(sc)
    for comps, disp, rot in obstacles:
        c = mgf_amd.Compound(ctx, comps)
        c.set_pose(disp, rot)
        gw.add_obstacle(c)
        ow.add_obstacle(comps, disp, rot)
    gw.set_option("solver_mode", 6 if seed % 2 else 1)
    gw.set_option("resort_every", int(rng.integers(2, 9)))
    peak = 0
    tick = 0
    while tick < 130:
        if seed % 2 == 0 and tick % 20 == 10:      # a stretch of step_many: the oracle follows tick by tick
            many = gw.step_many(dt, iters, 6)
            for m in many:
                so = ow.step(dt, iters)
                assert (int(m["n_constraints"]), int(m["n_terrain_constraints"])) == (so.n_constraints, so.n_terrain_constraints), (seed, tick)
                tick += 1
            continue
        sg, so = gw.step(dt, iters), ow.step(dt, iters)
        assert (sg.n_constraints, sg.n_terrain_constraints, sg.n_pair_candidates) == (so.n_constraints, so.n_terrain_constraints, so.n_pair_candidates), (seed, tick)
        peak = max(peak, int(sg.n_constraints))
        if tick % 9 == 8:
            compare_constraints(gw.constraints(), ow.constraints(), check_impulse=True)
        if tick == 45 and seed % 3 == 0:
            gw = gw.clone()
        tick += 1
    assert peak > 30, peak
    g, o = gw.state(), ow.state()
    for k in ("x", "q", "v", "omega", "delta"):
        assert values_equal(g[k], o[k]), (seed, k, rel_err(g[k], o[k]))
