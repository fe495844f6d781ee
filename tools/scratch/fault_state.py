"""what a block of k_contacts_spheres looks like at the tick the pre-fix kernel faulted (undivided 1M pile, store in the caller's order)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, mgf_amd
from mgf_amd import scenes
ctx = mgf_amd.Context(0)
sc = scenes.sphere_pile(128, 128, 64)
w = mgf_amd.World.from_scene(ctx, sc)
w.set_option("resort_every", 0)
dt, it = float(sc["dt"]), sc["iters"]
stop = int(sys.argv[1]) if len(sys.argv) > 1 else 0
t = 0
while True:
    t += 1
    w.step(dt, it)
    if stop == 0:
        print("tick", t, flush=True)
        if t >= 200: break
        continue
    if t == stop:
        c = w.constraints()
        n = len(w)
        pair = c["b"] >= 0
        per_a = np.bincount(c["a"][pair], minlength=n)
        per_t = np.bincount(c["a"][~pair], minlength=n)
        blk = per_a[: (n // 256) * 256].reshape(-1, 256).sum(1)
        both = (per_a > 0) & (per_t > 0)
        print(f"tick {t}: {len(c)} constraints; partner contacts per body as a: max {per_a.max()}, bodies with > 12: {(per_a > 12).sum()}; terrain per body max {per_t.max()}, "
              f"bodies with both {both.sum()}; per block of 256: max {blk.max()}, blocks > 1024: {(blk > 1024).sum()}, > 2048: {(blk > 2048).sum()}, > 3072: {(blk > 3072).sum()}", flush=True)
        break
