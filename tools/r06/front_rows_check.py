"""r06: the list-free front end of worlds that are not spheres only (option front_rows, k_front_rows.h) against the list-based kernels it replaces
(front_rows = 0) and, on small scenes, against the oracle: states bit for bit; then the tick times of BASELINE config 3 either way."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, mgf_amd
from mgf_amd import scenes
from oracle import oracle as O
from tests.util import oracle_world, compare_constraints, values_equal
ctx = mgf_amd.Context(0)
quick = "--quick" in sys.argv

def same_state(x, y):
    return all(np.array_equal(x[k].view(np.uint32), y[k].view(np.uint32)) for k in ("x", "q", "v", "omega"))

cases = [
    ("capsules 8x6x8 dense, heightfield 12x12 (face grid)", scenes.capsule_field(8, 6, 8, quads=12, pitch=1.6), 240),
    ("capsules 8x6x8 dense, heightfield 4x4 (32 faces: rows)", scenes.capsule_field(8, 6, 8, quads=4, pitch=1.6), 240),
    ("mixed 30% spheres 10x6x10, heightfield 16x16", scenes.capsule_field(10, 6, 10, quads=16, pitch=1.6, sphere_fraction=0.3), 240),
    ("mixed 30% spheres 10x6x10, heightfield 5x5 (rows)", scenes.capsule_field(10, 6, 10, quads=5, pitch=1.6, sphere_fraction=0.3), 240),
    ("two-part bodies 8x5x8 + 40 plain spheres over the box", scenes.dumbbell_field(8, 5, 8, n_plain=40), 300),
    ("two-part bodies 10x4x10 over the box", scenes.dumbbell_field(10, 4, 10), 240),
]
for name, sc, ticks in cases:
    dt, it = float(sc["dt"]), sc["iters"]
    a, b = mgf_amd.World.from_scene(ctx, sc), mgf_amd.World.from_scene(ctx, sc)
    b.set_option("front_rows", 0)
    a.set_option("front_rows_check", 1)  # (the faces the cheap reject drops are tested all the same: a contact among them is an error)
    ow = oracle_world(sc)
    for s in range(ticks):
        if s % 40 == 0:  # against the oracle from the same state: the constraint list and the state after the solve
            st = a.state()
            ow.set_state(x=st["x"], q=st["q"], v=st["v"], omega=st["omega"], delta=st["delta"])
            ow.build_constraints(dt); sa = a.build_constraints(dt); sb = b.build_constraints(dt)
            compare_constraints(a.constraints(), ow.constraints())
            assert sa.n_pair_candidates == sb.n_pair_candidates and sa.n_terrain_candidates == sb.n_terrain_candidates, (name, s, sa.n_pair_candidates, sb.n_pair_candidates, sa.n_terrain_candidates, sb.n_terrain_candidates)
            ow.solve(it); a.solve(it); b.solve(it)
            g, o = a.state(), ow.state()
            for k in ("x", "q", "v", "omega"):
                assert values_equal(g[k], o[k]), (name, s, k)
        else:
            sa, sb = a.step(dt, it), b.step(dt, it)
        assert sa.n_constraints == sb.n_constraints and sa.n_terrain_constraints == sb.n_terrain_constraints, (name, s, sa.n_constraints, sb.n_constraints)
        assert same_state(a.state(), b.state()), (name, s)
    print(f"{name}: {ticks} ticks bit-identical (front_rows 1 / 0 / oracle), {sa.n_constraints} constraints ({sa.n_terrain_constraints} terrain) at the end", flush=True)
    del a, b

if not quick and "--config5" in sys.argv:
    sc = scenes.dumbbell_field(64, 16, 64)
    dt, it = float(sc["dt"]), sc["iters"]
    a, b = mgf_amd.World.from_scene(ctx, sc), mgf_amd.World.from_scene(ctx, sc)
    b.set_option("front_rows", 0)
    for s in range(50, 401, 50):
        t0 = time.perf_counter(); sa = a.step_many(dt, it, 50); ta = (time.perf_counter() - t0) / 50
        t0 = time.perf_counter(); sb = b.step_many(dt, it, 50); tb = (time.perf_counter() - t0) / 50
        assert same_state(a.state(), b.state()), s
        assert int(sa[49]["n_constraints"]) == int(sb[49]["n_constraints"]) and int(sa[49]["n_pair_candidates"]) == int(sb[49]["n_pair_candidates"]) and int(sa[49]["n_terrain_candidates"]) == int(sb[49]["n_terrain_candidates"]), s
        print(f"config 5 tick {s}: bit-identical, {int(sa[49]['n_constraints'])} constraints ({int(sa[49]['n_terrain_constraints'])} terrain), {int(sa[49]['n_pair_candidates'])} accepted partners; ms/tick front_rows {ta*1e3:.3f} vs lists {tb*1e3:.3f}", flush=True)
    del a, b
if not quick and "--config5" not in sys.argv:
    sc = scenes.capsule_field(128, 32, 32, quads=158)
    dt, it = float(sc["dt"]), sc["iters"]
    a, b = mgf_amd.World.from_scene(ctx, sc), mgf_amd.World.from_scene(ctx, sc)
    b.set_option("front_rows", 0)
    if "--check" in sys.argv: a.set_option("front_rows_check", 1)
    for s in range(50, 401, 50):
        t0 = time.perf_counter(); sa = a.step_many(dt, it, 50); ta = (time.perf_counter() - t0) / 50
        t0 = time.perf_counter(); sb = b.step_many(dt, it, 50); tb = (time.perf_counter() - t0) / 50
        assert same_state(a.state(), b.state()), s
        assert int(sa[49]["n_constraints"]) == int(sb[49]["n_constraints"]) and int(sa[49]["n_pair_candidates"]) == int(sb[49]["n_pair_candidates"]) and int(sa[49]["n_terrain_candidates"]) == int(sb[49]["n_terrain_candidates"]), s
        print(f"   near bodies {a.counter('front_near')}, faces accepted {a.counter('front_faces')}, after the cheap reject {a.counter('front_slots')}")
        print(f"config 3 tick {s}: bit-identical, {int(sa[49]['n_constraints'])} constraints, {int(sa[49]['n_terrain_candidates'])} terrain candidates, {int(sa[49]['n_pair_candidates'])} accepted partners; ms/tick front_rows {ta*1e3:.3f} vs lists {tb*1e3:.3f}", flush=True)
print("OK")
