"""Long run of the tile set on one GPU: BASELINE config 4 as 8 x-slab tiles, as 4 tiles of twice the width and undivided - three cuts of
the SAME scene.  No oracle at this size for this long (tests/test_gpu_fullsize.py pins the first ticks bit for bit): what must hold over
hundreds of ticks is that no body is lost or doubled when it changes owner (the tags stay a permutation of the scene's), every state stays
finite, no tick is lost or repeated, and the tiled cuts - block-Jacobi across different faces, so not bit-identical - settle to the same pile
(height of the centre of mass) within a few per cent."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, mgf_amd
from mgf_amd import scenes
ticks = int(sys.argv[1]) if len(sys.argv) > 1 else 600
every = int(sys.argv[2]) if len(sys.argv) > 2 else 100
only = [int(a) for a in sys.argv[3].split(",")] if len(sys.argv) > 3 else [8, 4, 1]  # (1: the same scene undivided - exact Gauss-Seidel, the global solver)
ctx = mgf_amd.Context(0)
summ = {}
for P in only:
    nx = 128 // P
    scs = [scenes.sphere_pile_tile(nx, 128, 64, r, P) for r in range(P)]
    worlds = []
    for sc in scs:
        w = mgf_amd.World.from_scene(ctx, sc)
        w.set_tags(sc["tags"])
        worlds.append(w)
    T = mgf_amd.Tiles(ctx, worlds, [sc["x_range"] for sc in scs])
    dt, it = float(scs[0]["dt"]), scs[0]["iters"]
    n_total = sum(len(w) for w in worlds)
    t0 = time.time()
    for s in range(1, ticks + 1):
        T.step(dt, it)
        if s % every == 0:
            tags = np.concatenate([w.tags() for w in worlds])
            assert len(tags) == n_total and np.array_equal(np.sort(tags), np.arange(n_total, dtype=tags.dtype)), f"{P} tiles tick {s}: bodies lost or doubled"
            st = [w.state() for w in worlds]
            x = np.concatenate([a["x"] for a in st]); v = np.concatenate([a["v"] for a in st])
            assert np.isfinite(x).all() and np.isfinite(v).all(), f"{P} tiles tick {s}: a state is not finite"
            assert T.counter("ticks") == s and T.counter("ticks_retried") == 0
            moved = sum(T.migrated(k) for k in range(P))
            summ[(P, s)] = (float(x[:, 1].mean()), float((v.astype(np.float64) ** 2).sum() * 0.5))
            print(f"{P} tiles tick {s}: {n_total} bodies accounted for, {moved} hand-overs so far, mean height {summ[(P, s)][0]:.4f}, kinetic energy {summ[(P, s)][1]:.1f}, "
                  f"{(time.time() - t0) * 1e3 / s:.2f} ms per tick [{time.time() - t0:.0f} s]", flush=True)
    del T, worlds
# ---- BASELINE config 5 (65 536 bodies of a sphere and a capsule each) the same way: 8 and 4 slabs of one field, bodies of two parts crossing faces
whole = scenes.dumbbell_field(64, 16, 64)
for P in [p for p in only if p > 1]:
    scs = scenes.split_by_slabs(whole, P, 64 * 2.2 / 2.0)
    worlds = []
    for sc in scs:
        w = mgf_amd.World.from_scene(ctx, sc)
        w.set_tags(sc["tags"])
        worlds.append(w)
    T = mgf_amd.Tiles(ctx, worlds, [sc["x_range"] for sc in scs], halo=2.0)
    dt, it = float(scs[0]["dt"]), scs[0]["iters"]
    n_total = sum(len(w) for w in worlds)
    t0 = time.time()
    for s in range(1, ticks + 1):
        T.step(dt, it)
        if s % every == 0:
            tags = np.concatenate([w.tags() for w in worlds])
            assert len(tags) == n_total and len(np.unique(tags)) == n_total, f"config 5, {P} tiles tick {s}: bodies lost or doubled"
            st = [w.state() for w in worlds]
            x = np.concatenate([a["x"] for a in st]); v = np.concatenate([a["v"] for a in st])
            assert np.isfinite(x).all() and np.isfinite(v).all(), f"config 5, {P} tiles tick {s}: a state is not finite"
            assert T.counter("ticks") == s and T.counter("ticks_retried") == 0
            summ[("c5", P, s)] = float(x[:, 1].mean())
            print(f"config 5, {P} tiles tick {s}: {n_total} bodies accounted for, {sum(T.migrated(k) for k in range(P))} hand-overs so far, mean height {summ[('c5', P, s)]:.4f}, "
                  f"{(time.time() - t0) * 1e3 / s:.2f} ms per tick [{time.time() - t0:.0f} s]", flush=True)
    del T, worlds
cuts = [P for P in only if P > 1]  # (the undivided world is exact Gauss-Seidel - another algorithm than block-Jacobi across faces: ten iterations
                                    # compress a 128-layer pile differently; it is here for the invariants, not for the comparison)
if len(cuts) > 1:
    for s in range(every, ticks + 1, every):
        hs = [summ[(P, s)][0] for P in cuts]
        assert max(hs) - min(hs) <= 0.03 * abs(min(hs)), f"tick {s}: the cuts disagree on the pile's height ({hs})"
    print(f"the cuts into {cuts} tiles agree on the pile's mean height within 3 % at every mark")
    for s in range(every, ticks + 1, every):
        hs = [summ[("c5", P, s)] for P in cuts]
        assert max(hs) - min(hs) <= 0.03 * abs(min(hs)) + 0.02, f"config 5 tick {s}: the cuts disagree on the field's height ({hs})"
    print(f"config 5: likewise")
print("soak OK")
