"""The gate of VERDICT r4 item 5 (ghost refresh inside the solver launch): what would a tile's solver cost if its ten iterations were ONE
launch?  A 131 072-sphere slab as a world of its own (no neighbours: nothing to wait for - the upper bound of what an in-launch refresh can
gain), the same ticks solved as five launches of two iterations (what a tile pays today) and as one launch of ten; solver kernel time by
HIP events, ticks 10..30 (falling) and 70..130 (compacting).  The two ways are the same sequential algorithm: the states must be equal."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401

import mgf_amd  # noqa: E402
from mgf_amd import scenes  # noqa: E402


def ticks(w, dt, n, launches, iters):
    ms, cons = 0.0, 0
    for _ in range(n):
        st = w.build_constraints(dt)
        cons += int(st.n_constraints)
        for _k in range(launches):
            ms += float(w.solve(iters).ms_solver_kernels)
    return ms * 1e3 / n, cons / n


ctx = mgf_amd.Context(0)
for dims in ((16, 128, 64), (32, 64, 64)):
    sc = scenes.sphere_pile(*dims)
    dt = float(sc["dt"])
    w = mgf_amd.World.from_scene(ctx, sc)
    done = 0
    for lo, hi in ((10, 30), (70, 130)):
        ticks(w, dt, lo - done, 5, 2)
        a, b = w.clone(), w.clone()
        for x in (a, b):
            x.set_option("time_solver_kernels", 1)
        us5, c5 = ticks(a, dt, hi - lo, 5, 2)
        us1, c1 = ticks(b, dt, hi - lo, 1, 10)
        sa, sb = a.state(), b.state()
        same = all(np.array_equal(sa[f], sb[f]) for f in ("x", "q", "v", "omega"))
        print(f"slab {dims} ticks {lo}..{hi}: constraints/tick {c5:.0f}; five launches of two {us5:.1f} us per tick ({us5 / 5:.1f} per launch), "
              f"one launch of ten {us1:.1f} us: {us5 / us1:.2f}x; states equal: {same}", flush=True)
        ticks(w, dt, hi - lo, 5, 2)
        done = hi
        del a, b
    del w
