"""Development aid: the headline world stepped with no instrumentation at all (for rocprofv3 traces of the launch gaps)."""
import sys
sys.path.insert(0, '/root/repo')
import time, torch, mgf_amd
from mgf_amd import scenes
warm, steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10, int(sys.argv[2]) if len(sys.argv) > 2 else 60
ctx = mgf_amd.Context(0)
sc = scenes.sphere_pile(64, 64, 64)
w = mgf_amd.World.from_scene(ctx, sc)
dt = float(sc['dt'])
w.step_many(dt, 10, warm)
torch.cuda.synchronize(); t0 = time.perf_counter()
w.step_many(dt, 10, steps)
torch.cuda.synchronize(); print('ms/tick %.4f' % ((time.perf_counter() - t0) * 1e3 / steps))
