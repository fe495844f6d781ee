"""r06: the native tile driver against the oracle's tiles on random cuts of random mixed fields (two-part bodies and plain spheres; a seed in three has
one kind per side of a face): P in 2..4 tiles, R in 1..4, a drift that carries bodies across the faces - ghost records of both widths, hand-overs,
kinds that change in mid-run.  python tools/r06/tiles_fuzz.py <first seed> <last seed>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, mgf_amd
from mgf_amd import scenes
from mgf_amd.tiles import step_tiles_inprocess
from tests.test_gpu_tiles_native import _native, _oracle_tiles, _assert_equal
ctx = mgf_amd.Context(0)
a, b = int(sys.argv[1]), int(sys.argv[2])
bad = []
for seed in range(a, b):
    rng = np.random.default_rng(31000 + seed)
    P, R = int(rng.integers(2, 5)), int(rng.integers(1, 5))
    nx = int(rng.integers(4, 9))
    sc = scenes.dumbbell_field(nx, 2, 4, n_plain=int(rng.integers(0, 30)), seed=scenes.SEED + seed)
    n_plain = len(sc["comps"])
    half = nx * 2.2 / 2.0 + 2.0
    if seed % 3 == 0 and n_plain:   # plain spheres on the right only, two-part bodies on the left only
        cb = sc["compound"]
        cb["comps"]["p"][:, 0] -= np.float32(cb["comps"]["p"][:, 0].max() + 1.0)
        sc["comps"]["p"][:, 0] = np.float32(0.7) + np.float32(0.95) * (np.arange(n_plain) % 4).astype(np.float32)
        sc["comps"]["p"][:, 1] = np.float32(1.0) + np.float32(1.1) * (np.arange(n_plain) // 4).astype(np.float32)
    drift = np.float32([rng.uniform(-4.0, 4.0), 0.0, 0.0])
    sc["v0"] = (sc["v0"] + drift).astype(np.float32)
    tiles = scenes.split_by_slabs(sc, P, half)
    try:
        T, worlds = _native(ctx, tiles, halo=2.0, refresh_every=R)
        ot = _oracle_tiles(tiles, halo=2.0, refresh_every=R)
        dt, iters = float(sc["dt"]), sc["iters"]
        ghosts = 0
        for tick in range(70):
            sg, so = T.step(dt, iters), step_tiles_inprocess(ot)
            assert [int(s.n_constraints) for s in sg] == [int(s["n_constraints"]) for s in so], tick
            ghosts += sum(int(s.n_ghost_constraints) for s in sg)
            if tick % 10 == 9: _assert_equal(worlds, ot, f"tick {tick}")
        _assert_equal(worlds, ot, "end")
        print(f"seed {seed}: P {P} R {R} bodies {sum(len(w) for w in worlds)} ghost constraints {ghosts} hand-overs {sum(T.migrated(r) for r in range(P))}: bit-identical", flush=True)
    except AssertionError as e:
        bad.append(seed); print("seed", seed, "FAILED", str(e)[:200], flush=True)
print("seeds", a, "..", b, ":", len(bad), "failures", bad)
ctx.close()
