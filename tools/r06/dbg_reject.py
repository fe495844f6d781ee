import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, mgf_amd
from tests.test_gpu_tri_reject import _problems
ctx = mgf_amd.Context(0)
def pt_tri(P, a_, b_, c_):
    ab, ac, ap = b_ - a_, c_ - a_, P - a_
    d1, d2 = ab @ ap, ac @ ap
    if d1 <= 0 and d2 <= 0: return a_
    bp = P - b_; d3, d4 = ab @ bp, ac @ bp
    if d3 >= 0 and d4 <= d3: return b_
    vc = d1 * d4 - d3 * d2
    if vc <= 0 and d1 >= 0 and d3 <= 0: return a_ + ab * (d1 / (d1 - d3))
    cp = P - c_; d5, d6 = ab @ cp, ac @ cp
    if d6 >= 0 and d5 <= d6: return c_
    vb = d5 * d2 - d1 * d6
    if vb <= 0 and d2 >= 0 and d6 <= 0: return a_ + ac * (d2 / (d2 - d6))
    va = d3 * d6 - d5 * d4
    if va <= 0 and (d4 - d3) >= 0 and (d5 - d6) >= 0: return b_ + (c_ - b_) * ((d4 - d3) / ((d4 - d3) + (d5 - d6)))
    den = 1.0 / (va + vb + vc)
    return a_ + ab * (vb * den) + ac * (vc * den)
for tri_size in (0.05, 0.5, 3.0, 40.0, 400.0):
  rng = np.random.default_rng(int(tri_size * 1000) + 6)
  for near in (0.6, 1.0, 1.1, 1.6, 4.0):
    tag, p, d, r, delta, tris = _problems(rng, 200_000, tri_size, near)
    far, cnt = mgf_amd._capi.tri_reject_batch(ctx, tag, p, d, r, delta, tris)
    bad = np.nonzero((far == 1) & (cnt > 0))[0]
    print(tri_size, near, len(bad), "violations; tags", np.bincount(tag[bad], minlength=2), "counts", np.bincount(cnt[bad]), "dropped", int(far.sum()), "contacts", int((cnt > 0).sum()))
    for i in bad[:4]:
        P, D, T = p[i].astype(np.float64), d[i].astype(np.float64), tris[i].astype(np.float64)
        best = min(np.linalg.norm(P + D * s - pt_tri(P + D * s, T[0], T[1], T[2])) for s in np.linspace(0, 1, 401))
        n = np.cross(T[1] - T[0], T[2] - T[0]); n /= np.linalg.norm(n)
        d0, d1 = (P - T[0]) @ n, (P + D - T[0]) @ n
        print(f"tag {tag[i]} r {r[i]:.4f} |d| {np.linalg.norm(D):.3f} |delta| {np.linalg.norm(delta[i]):.4f} contacts {cnt[i]}: distance axis-triangle {best:.4f} (reach r+|delta| = {r[i] + np.linalg.norm(delta[i]):.4f}); plane distances of the ends {d0:.4f} {d1:.4f}; tri edge lengths {[round(float(np.linalg.norm(T[(k+1)%3]-T[k])),4) for k in range(3)]}")
