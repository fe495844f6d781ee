"""Development probe for DESIGN 9.1 (mirrored constraints): dumps the tick's constraint list (caller indices, insertion order) and the
body positions of the 64^3 pile at a tick, for tools/mirror_sim.py."""
import sys
sys.path.insert(0, '/root/repo')
import numpy as np, mgf_amd
from mgf_amd import scenes
tick = int(sys.argv[1]) if len(sys.argv) > 1 else 40
ctx = mgf_amd.Context(0)
sc = scenes.sphere_pile(64, 64, 64)
w = mgf_amd.World.from_scene(ctx, sc)
w.step_many(float(sc['dt']), 10, tick)
w.step(float(sc['dt']), 10)
c = w.constraints()
x = w.state()['x']
np.savez_compressed(f'/root/repo/gpurun_out/mirror_dump_{tick}.npz', a=c['a'].astype(np.int32), b=c['b'].astype(np.int32), x=x.astype(np.float32))
print('tick', tick, 'constraints', len(c))
