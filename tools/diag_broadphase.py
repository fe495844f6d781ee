import sys
sys.path.insert(0, '/root/repo')
import numpy as np, mgf_amd
from mgf_amd import scenes
ctx = mgf_amd.Context(0)
sc = scenes.sphere_pile(64, 64, 64)
w = mgf_amd.World.from_scene(ctx, sc)
w.set_option('phase_timing', 1)
w.set_option('debug_bvh', 1)
for s in range(90):
    st = w.step(float(sc['dt']), 10)
    if s % 8 == 0:
        x = w.state()['x']
        print(s, 'bp %.3f ms' % st.ms_broadphase, 'cand', st.n_pair_candidates, 'C', st.n_constraints, 'refits', st.n_refits,
              'x[%.1f %.1f] y[%.1f %.1f] z[%.1f %.1f]' % (x[:,0].min(), x[:,0].max(), x[:,1].min(), x[:,1].max(), x[:,2].min(), x[:,2].max()), flush=True)
