"""Long-run cross-check of the pipelined mgf_world_step_many (development aid): the 262 144-sphere scene stepped in batches
with the pipeline on, with it off, and tick by tick; per-tick constraint counts and the state compared bit for bit."""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, mgf_amd
from mgf_amd import scenes
ticks = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 50
ctx = mgf_amd.Context(0)
sc = scenes.sphere_pile(64, 64, 64)
dt = float(sc['dt'])
a, b, c = (mgf_amd.World.from_scene(ctx, sc) for _ in range(3))
b.set_option('pipeline', 0)
t0 = time.time()
for s in range(batch, ticks + 1, batch):
    na = [st.n_constraints for st in a.step_many(dt, 10, batch)]
    nb = [st.n_constraints for st in b.step_many(dt, 10, batch)]
    nc = [int(c.step(dt, 10).n_constraints) for _ in range(batch)]
    assert na == nb == nc, f"ticks up to {s}: constraint counts differ"
    if s % (4 * batch) == 0 or s + batch > ticks:
        sa, sb, sc_ = a.state(), b.state(), c.state()
        for k in ("x", "q", "v", "omega"):
            assert np.array_equal(sa[k].view(np.uint32), sc_[k].view(np.uint32)) and np.array_equal(sb[k].view(np.uint32), sc_[k].view(np.uint32)), f"tick {s}: {k}"
        print(f"tick {s}: {na[-1]} constraints, pipelined / plain batches / single steps bit-identical; retries {a.counter('capacity_retries')}/"
              f"{b.counter('capacity_retries')}/{c.counter('capacity_retries')}  [{time.time() - t0:.0f} s]", flush=True)
print("soak OK")
