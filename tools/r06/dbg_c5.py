import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, mgf_amd
from mgf_amd import scenes
from tests.util import compare_constraints
ctx = mgf_amd.Context(0)
sc = scenes.dumbbell_field(64, 16, 64)
dt, it = float(sc["dt"]), sc["iters"]
a, b = mgf_amd.World.from_scene(ctx, sc), mgf_amd.World.from_scene(ctx, sc)
b.set_option("front_rows", 0)
a.step_many(dt, it, 100); b.step_many(dt, it, 100)
for s in range(100, 160):
    sa = a.build_constraints(dt); sb = b.build_constraints(dt)
    ca, cb = a.constraints(), b.constraints()
    if sa.n_constraints != sb.n_constraints:
        print("tick", s, "constraint counts differ", sa.n_constraints, sb.n_constraints, "terrain", sa.n_terrain_constraints, sb.n_terrain_constraints)
        na, nb = len(ca["a"]), len(cb["a"])
        # per body counts
        ba = np.bincount(ca["a"], minlength=len(sc["v0"])); bb = np.bincount(cb["a"], minlength=len(sc["v0"]))
        bad = np.nonzero(ba != bb)[0]
        print("bodies with different counts:", bad[:10], ba[bad[:10]], bb[bad[:10]])
        i = bad[0]
        print("A rows:", [(int(x), int(y)) for x, y in zip(ca["a"][ca["a"] == i], ca["b"][ca["a"] == i])])
        print("B rows:", [(int(x), int(y)) for x, y in zip(cb["a"][cb["a"] == i], cb["b"][cb["a"] == i])])
        break
    try:
        compare_constraints(ca, cb)
    except AssertionError as e:
        print("tick", s, "constraints differ:", str(e)[:300]); break
    a.solve(it); b.solve(it)
    x, y = a.state(), b.state()
    if not all(np.array_equal(x[k].view(np.uint32), y[k].view(np.uint32)) for k in ("x", "q", "v", "omega")):
        print("tick", s, "states differ after the solve (equal constraint lists)"); break
else:
    print("no difference up to tick 160")

# ---- the body the lists disagree about: its parts against every face, in float64
if 'bad' in dir():
    i = int(bad[0])
    st = a.state()
    x, q, v = st["x"][i].astype(np.float64), st["q"][i].astype(np.float64), st["v"][i].astype(np.float64)
    comps = sc["compound"]["comps"]; cm = sc["compound"]["comp_mass"]
    k0 = 2 * i
    c_s = comps["p"][k0].astype(np.float64); c_c = (comps["p"][k0 + 1] + 0.5 * comps["d"][k0 + 1]).astype(np.float64)
    com = (cm[k0] * c_s + cm[k0 + 1] * c_c) / (cm[k0] + cm[k0 + 1])
    s_, qx, qy, qz = q
    R = np.array([[1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - s_ * qz), 2 * (qx * qz + s_ * qy)],
                  [2 * (qx * qy + s_ * qz), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - s_ * qx)],
                  [2 * (qx * qz - s_ * qy), 2 * (qy * qz + s_ * qx), 1 - 2 * (qx * qx + qy * qy)]])
    delta = st["delta"][i].astype(np.float64)
    print("body", i, "x", x, "q", q, "v", v, "delta", delta)
    parts = [("sphere", x + R @ (c_s - com), np.zeros(3), 0.5), ("capsule", x + R @ (comps["p"][k0 + 1].astype(np.float64) - com), R @ comps["d"][k0 + 1].astype(np.float64), 0.3)]
    T = sc["terrain"]; V = T["verts"].astype(np.float64) + np.asarray(T["pos"], np.float64); F = T["faces"]
    def pt_tri(p, a_, b_, c_):
        ab, ac, ap = b_ - a_, c_ - a_, p - a_
        d1, d2 = ab @ ap, ac @ ap
        if d1 <= 0 and d2 <= 0: return a_
        bp = p - b_; d3, d4 = ab @ bp, ac @ bp
        if d3 >= 0 and d4 <= d3: return b_
        vc = d1 * d4 - d3 * d2
        if vc <= 0 and d1 >= 0 and d3 <= 0: return a_ + ab * (d1 / (d1 - d3))
        cp = p - c_; d5, d6 = ab @ cp, ac @ cp
        if d6 >= 0 and d5 <= d6: return c_
        vb = d5 * d2 - d1 * d6
        if vb <= 0 and d2 >= 0 and d6 <= 0: return a_ + ac * (d2 / (d2 - d6))
        va = d3 * d6 - d5 * d4
        if va <= 0 and (d4 - d3) >= 0 and (d5 - d6) >= 0: return b_ + (c_ - b_) * ((d4 - d3) / ((d4 - d3) + (d5 - d6)))
        den = 1.0 / (va + vb + vc)
        return a_ + ab * (vb * den) + ac * (vc * den)
    for name, p, d, r in parts:
        for f, (ia, ib, ic) in enumerate(F):
            best = min(np.linalg.norm(p + d * s - pt_tri(p + d * s, V[ia], V[ib], V[ic])) for s in np.linspace(0, 1, 201))
            lim = (r + np.linalg.norm(delta)) * 1.01 + 1e-3
            print(f"  {name} face {f}: distance of the axis {best:.5f}  lim {lim:.5f}  {'NEAR' if best <= lim else ''}")
