"""BASELINE config 5 on one GPU (measured only: the reference has no such body, the definition is this build's):
65 536 bodies of two components each (sphere + capsule), `--plain` extra spheres; ms per tick after a warm-up."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401

import mgf_amd  # noqa: E402
from mgf_amd import scenes  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dims", type=int, nargs=3, default=[64, 16, 64])
    ap.add_argument("--plain", type=int, default=0)
    ap.add_argument("--warmup", type=int, default=80)
    ap.add_argument("--ticks", type=int, default=40)
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=VALUE")
    a = ap.parse_args()
    sc = scenes.dumbbell_field(*a.dims, n_plain=a.plain)
    ctx = mgf_amd.Context(0)
    w = mgf_amd.World.from_scene(ctx, sc)
    w.set_option('phase_timing', 1)
    dt, iters = float(sc["dt"]), sc["iters"]
    for kv in a.opt:
        key, val = kv.split("=")
        w.set_option(key, int(val))
    for _ in range(a.warmup):
        w.step(dt, iters)
    acc = dict(ms_integrate=0.0, ms_broadphase=0.0, ms_narrowphase=0.0, ms_setup=0.0, ms_solve=0.0, ms_total=0.0)
    for _ in range(a.ticks):
        st = w.step(dt, iters).as_dict()
        for k in acc:
            acc[k] += st[k]
    n = len(w)
    print(f"config 5: {n} bodies of two parts; last {a.ticks} ticks mean {acc['ms_total'] / a.ticks:.3f} ms/tick; constraints {st['n_constraints']} "
          f"(terrain {st['n_terrain_constraints']}), accepted pairs {st['n_pair_candidates']}")
    print({k: round(v / a.ticks, 3) for k, v in acc.items()})


if __name__ == "__main__":
    main()
