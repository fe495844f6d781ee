import sys
sys.path.insert(0, '/root/repo')
import numpy as np, mgf_amd
from mgf_amd import scenes
ctx = mgf_amd.Context(0)
sc = scenes.sphere_pile(64, 64, 64)
for mode, bpc, sl in [(0,0,0),(1,4,2),(3,1,2),(3,2,2),(3,4,2),(3,4,0),(3,6,2)]:
    w = mgf_amd.World.from_scene(ctx, sc)
    w.set_option('solver_mode', mode)
    if mode:
        w.set_option('flow_blocks_per_cu', bpc); w.set_option('flow_sleep', sl)
    ms=[]; tot=[]
    import time
    for s in range(40):
        t0=time.perf_counter(); st = w.step(float(sc['dt']), 10); el=time.perf_counter()-t0
        if s>=10: ms.append(st.ms_solve); tot.append(el*1e3)
    print(f"mode {mode} blocks/cu {bpc} sleep {sl}: solve {np.mean(ms):.3f} ms  tick {np.mean(tot):.3f} ms", flush=True)
    del w
