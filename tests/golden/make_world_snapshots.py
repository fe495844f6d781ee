"""Generates tests/golden/world_snapshots.npz: World::step golden snapshots of the reference's demo scene
(mgf_demo/balls.rs:67-96 on the terrain of world.rs:118-150).

The reference cannot be built or run in this container (Rust), so the snapshots are
produced by the CPU oracle (oracle/, the line-by-line restatement pinned by the reference's known-answer
vectors and cross-checked by tests/np_restatement.py) - they pin the oracle and the HIP path against
regressions and against each other, not against rustc output.

    python tests/golden/make_world_snapshots.py          # rewrites the fixture

Content per scene (balls512: num=8, iters=10 = BASELINE config 1; balls1332: the unmodified demo,
num=11 plus the extra ball, iters=20) and per insertion order (demo = world.rs order, canonical = the
HIP path's order): state after k = 1, 2, 10, 60, 300 steps.  balls512 stores the full state; balls1332
stores 64 sampled bodies plus a SHA-256 of the full state bytes.
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mgf_amd import scenes  # noqa: E402
from oracle import oracle as O  # noqa: E402
from tests.util import oracle_world  # noqa: E402

STEPS = (1, 2, 10, 60, 300)
SCENES = {"balls512": dict(num=8, extra_ball=False, iters=10), "balls1332": dict(num=11, extra_ball=True, iters=20)}
FIELDS = ("x", "q", "v", "omega")


def state_digest(st):
    h = hashlib.sha256()
    for f in FIELDS:
        h.update(np.ascontiguousarray(st[f], np.float32).tobytes())
    return h.hexdigest()


def sample_ids(n):
    return np.unique(np.linspace(0, n - 1, 64).astype(np.int64))


def snapshots(name, order):
    scene = scenes.balls_demo(**SCENES[name])
    w = oracle_world(scene, order=order)
    out, k = {}, 0
    for target in STEPS:
        while k < target:
            st = w.step(float(scene["dt"]), scene["iters"])
            k += 1
        out[target] = (w.state(), int(st.n_constraints))
    return out


def main():
    data = {}
    for name in SCENES:
        for oname, order in (("demo", O.ORDER_DEMO), ("canonical", O.ORDER_CANONICAL)):
            for k, (st, nc) in snapshots(name, order).items():
                key = f"{name}/{oname}/{k}"
                data[key + "/n_constraints"] = np.int64(nc)
                data[key + "/sha256"] = np.array(state_digest(st))
                ids = np.arange(len(st["x"])) if name == "balls512" else sample_ids(len(st["x"]))
                data[key + "/ids"] = ids
                for f in FIELDS:
                    data[key + "/" + f] = st[f][ids]
    path = os.path.join(ROOT, "tests", "golden", "world_snapshots.npz")
    np.savez_compressed(path, **data)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
