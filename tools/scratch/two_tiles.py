"""Can two tile-sized solver launches share the chip?  Two 131 072-sphere slabs, each in a context (stream) of its own, solved the way a
tile is (five launches of two iterations per tick) from two host threads, against one of them alone - with the LDS split that lets two
blocks live on a CU (own constants only: < 80 KB per block) and with the default one."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, mgf_amd
from mgf_amd import scenes

def make(opts):
    ctx = mgf_amd.Context(0)
    sc = scenes.sphere_pile(16, 128, 64)
    w = mgf_amd.World.from_scene(ctx, sc)
    for k, v in opts.items():
        w.set_option(k, v)
    return ctx, w, float(sc["dt"])

def ticks(w, dt, n, out=None, k=0):
    t0 = time.perf_counter()
    for _ in range(n):
        w.build_constraints(dt)
        for _ in range(5):
            w.solve(2)
    if out is not None:
        out[k] = time.perf_counter() - t0

for name, opts in (("default split", {}), ("half-LDS split (CL 1, no foreign, 1300 slots)", {"flow6_foreign_lds": 0, "flow6_test_cap": 1300}),
                   ("no constants in LDS, 1300 slots", {"flow6_const_lds": 0, "flow6_test_cap": 1300})):
    a = make(opts); b = make(opts)
    ticks(a[1], a[2], 10); ticks(b[1], b[2], 10)
    t0 = time.perf_counter(); ticks(a[1], a[2], 20); one = time.perf_counter() - t0
    out = [0, 0]
    th = [threading.Thread(target=ticks, args=(x[1], x[2], 20, out, k)) for k, x in enumerate((a, b))]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    both = time.perf_counter() - t0
    print(f"{name}: one world 20 ticks {one*1e3/20:.3f} ms/tick; two worlds concurrently {both*1e3/20:.3f} ms per pair of ticks ({2*one/both:.2f}x the serial rate); "
          f"CL{a[1].counter('flow6_const_lds')} fallbacks {a[1].counter('flow6_fallbacks')}+{b[1].counter('flow6_fallbacks')}", flush=True)
    del a, b
