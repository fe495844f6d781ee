"""Long-run cross-check of the solver executors (development aid): the same 262 144-sphere scene stepped with solver
modes 0 (one launch per frontier), 1 (global dataflow), 5 (block-local dataflow) and 6 (block-local with message channels); velocities compared bit for bit."""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, mgf_amd
from mgf_amd import scenes
ticks = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
every = int(sys.argv[2]) if len(sys.argv) > 2 else 100
ctx = mgf_amd.Context(0)
sc = scenes.sphere_pile(64, 64, 64)
worlds = {}
for mode in (0, 1, 5, 6):
    w = mgf_amd.World.from_scene(ctx, sc)
    w.set_option('solver_mode', mode)
    worlds[mode] = w
t0 = time.time()
for s in range(1, ticks + 1):
    st = {m: w.step(float(sc['dt']), 10) for m, w in worlds.items()}
    assert len({int(x.n_constraints) for x in st.values()}) == 1, f"tick {s}: constraint counts differ"
    if s % every == 0 or s == ticks:
        ref = worlds[0].state()
        for m in (1, 5, 6):
            cur = worlds[m].state()
            for k in ("x", "q", "v", "omega"):
                assert np.array_equal(ref[k].view(np.uint32), cur[k].view(np.uint32)), f"tick {s}: mode {m} differs from mode 0 in {k}"
        print(f"tick {s}: {int(st[0].n_constraints)} constraints, modes 0/1/5/6 bit-identical; fallbacks {worlds[5].counter('flow5_fallbacks')} / {worlds[6].counter('flow6_fallbacks')} (reason {worlds[6].counter('flow6_fail_reason')}), "
              f"retries {worlds[5].counter('capacity_retries')}  [{time.time() - t0:.0f} s]", flush=True)
print("soak OK")
