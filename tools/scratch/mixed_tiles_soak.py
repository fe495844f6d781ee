"""a mixed field as tiles: two-part bodies and ordinary spheres on top (dumbbell_field(n_plain > 0)), 8 and 4 slabs, 400 ticks; and one process of 8
tiles against itself with the front-end / solver options turned plain - bit for bit"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, mgf_amd
from mgf_amd import scenes
ctx = mgf_amd.Context(0)
whole = scenes.dumbbell_field(48, 12, 48, n_plain=30000)
def run(P, opts, ticks=400, every=100):
    scs = scenes.split_by_slabs(whole, P, 48 * 2.2 / 2.0)
    worlds = []
    for sc in scs:
        w = mgf_amd.World.from_scene(ctx, sc); w.set_tags(sc["tags"])
        for k, v in opts.items(): w.set_option(k, v)
        worlds.append(w)
    T = mgf_amd.Tiles(ctx, worlds, [sc["x_range"] for sc in scs], halo=2.0)
    dt, it = float(scs[0]["dt"]), scs[0]["iters"]
    n_total = sum(len(w) for w in worlds)
    out = []
    for s in range(1, ticks + 1):
        T.step(dt, it)
        if s % every == 0:
            tags = np.concatenate([w.tags() for w in worlds])
            assert len(tags) == n_total and len(np.unique(tags)) == n_total
            st = [w.state() for w in worlds]
            x = np.concatenate([a["x"] for a in st])
            assert np.isfinite(x).all()
            order = np.argsort(tags)
            out.append((s, float(x[:, 1].mean()), np.concatenate([np.concatenate([a[k].ravel() for k in ("x", "q", "v", "omega")]) for a in st]).view(np.uint32).sum(dtype=np.uint64), sum(T.migrated(k) for k in range(P))))
            print(f"{P} tiles {opts} tick {s}: {n_total} bodies, hand-overs {out[-1][3]}, mean height {out[-1][1]:.4f}", flush=True)
    return out
a = run(8, {})
b = run(8, {"solver_mode": 1, "front_rows": 0, "wide_list": 0})
assert [r[2] for r in a] == [r[2] for r in b], "8 tiles: default against the global solver differ"
c = run(4, {})
for ra, rc in zip(a, c):
    assert abs(ra[1] - rc[1]) <= 0.03 * abs(rc[1]) + 0.02, (ra, rc)
print("OK: 8 tiles default == 8 tiles with the global solver (checksums), 8 and 4 tiles agree on the height")
