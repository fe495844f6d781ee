"""Bodies-per-block sweep of the block-local solver: for every scene and every block size, the tick's wall clock,
the solver kernel's time (HIP events around the launch, replay of the same ticks) and - from a traced tick in the middle of the
window - the critical path's hop mix.  Writes one JSON object per (scene, nb) line to gpurun_out/r04_block_sweep.jsonl.

    python tools/r04_block_sweep.py [--scenes config2,config3,config5,tile] [--nb 256,512,...] [--no-trace]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401

import mgf_amd  # noqa: E402
from mgf_amd import scenes  # noqa: E402
from tools import flowtrace_lib  # noqa: E402

SCENES = {
    # name: (builder, warm-up ticks, timed ticks, iterations per launch, launches per tick)
    "config2": (lambda: scenes.sphere_pile(64, 64, 64), 5, 20, 10, 1),
    "config2_settled": (lambda: scenes.sphere_pile(64, 64, 64), 400, 20, 10, 1),
    "config3": (lambda: scenes.capsule_field(128, 32, 32, quads=158), 150, 20, 10, 1),
    "config5": (lambda: scenes.dumbbell_field(64, 16, 64), 80, 20, 10, 1),
    # one x-slab of config 4 as a world of its own, solved the way a tile is: five launches of two iterations per tick
    "tile": (lambda: scenes.sphere_pile(16, 128, 64), 10, 20, 2, 5),
}


def configure(w, nb, opts):
    if nb:
        w.set_option("flow5_block", nb)
    for k, v in opts:
        w.set_option(k, v)


def run_ticks(w, dt, iters, launches, n):
    """n ticks; returns per-tick (constraints, solver kernel ms, launches)."""
    out = []
    if launches == 1:
        for st in w.step_many(dt, iters, n):
            out.append((int(st["n_constraints"]), float(st["ms_solver_kernels"]), int(st["solver_kernel_launches"])))
        return out
    for _ in range(n):
        st = w.build_constraints(dt)
        ms = 0.0
        for _k in range(launches):
            s2 = w.solve(iters)
            ms += float(s2.ms_solver_kernels)
        out.append((int(st.n_constraints), ms, launches))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", default="config3,config5,tile,config2")
    ap.add_argument("--nb", default="0,256,512,768,1024,1536,2047")
    ap.add_argument("--no-trace", action="store_true")
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=VALUE")
    ap.add_argument("--out", default="gpurun_out/r04_block_sweep.jsonl")
    a = ap.parse_args()
    opts = [(kv.split("=")[0], int(kv.split("=")[1])) for kv in a.opt]
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    ctx = mgf_amd.Context(0)
    ncu = 256
    with open(a.out, "a") as f:
        for name in a.scenes.split(","):
            build, warm, steps, iters, launches = SCENES[name]
            sc = build()
            dt = float(sc["dt"])
            seen_nb = set()
            for nb in [int(x) for x in a.nb.split(",")]:
                w = mgf_amd.World.from_scene(ctx, sc)
                n = len(w)
                need = (n + ncu - 1) // ncu
                eff = max(need, min(n, 256)) if nb == 0 else max(need, nb)
                if (eff, nb == 0) in seen_nb or (nb and (eff, True) in seen_nb and False):
                    continue
                seen_nb.add((eff, nb == 0))
                configure(w, nb, opts)
                run_ticks(w, dt, iters, launches, warm)
                snap = w.clone()
                row = {"scene": name, "bodies": n, "flow5_block": nb, "bodies_per_block": eff, "blocks": (n + eff - 1) // eff,
                       "warmup": warm, "steps": steps, "iters_per_launch": iters, "launches_per_tick": launches, "opts": dict(opts)}
                # wall clock, no events in the stream
                walls = []
                for _ in range(3):
                    x = snap.clone()
                    configure(x, nb, opts)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    run_ticks(x, dt, iters, launches, steps)
                    torch.cuda.synchronize()
                    walls.append((time.perf_counter() - t0) * 1e3 / steps)
                    row["flow6_runs"] = x.counter("flow6_runs")
                    row["flow6_fallbacks"] = x.counter("flow6_fallbacks")
                    del x
                row["tick_ms_wall"] = round(sorted(walls)[1], 4)
                # the solver kernel by HIP events
                x = snap.clone()
                configure(x, nb, opts)
                x.set_option("time_solver_kernels", 1)
                per = run_ticks(x, dt, iters, launches, steps)
                cons = sum(p[0] for p in per)
                kms = sum(p[1] for p in per)
                nl = sum(p[2] for p in per)
                row.update({"constraints_per_tick": round(cons / steps, 1), "solver_us_per_tick": round(kms * 1e3 / steps, 2),
                            "solver_us_per_launch": round(kms * 1e3 / max(nl, 1), 2),
                            "frac_of_hbm_roofline": round(cons * iters * launches * 288 / (kms * 1e-3) / 8e12, 4) if kms > 0 else None,
                            "flow6_const_lds": x.counter("flow6_const_lds"), "flow6_rec_lds": x.counter("flow6_rec_lds"), "flow6_nimp_lds": x.counter("flow6_nimp_lds"),
                            "flow6_max_slots": x.counter("flow6_max_slots"), "flow6_max_foreign": x.counter("flow6_max_foreign")})
                del x
                if not a.no_trace and row["flow6_fallbacks"] == 0:
                    x = snap.clone()
                    configure(x, nb, opts)
                    run_ticks(x, dt, iters, launches, steps // 2)
                    x.set_option("flow_trace", 1)
                    if launches == 1:
                        x.step(dt, iters)
                    else:
                        x.build_constraints(dt)
                        x.solve(iters)  # (the tick's first launch)
                    x.set_option("flow_trace", 0)
                    try:
                        row["trace"] = flowtrace_lib.analyse(flowtrace_lib.load(x.constraints()))
                    except Exception as e:  # noqa: BLE001
                        row["trace"] = {"error": repr(e)}
                    del x
                del snap, w
                f.write(json.dumps(row) + "\n")
                f.flush()
                t = row.get("trace", {})
                h = t.get("hops", {})
                print(f"{name:16s} nb {eff:5d} ({row['blocks']:3d} blocks): tick {row['tick_ms_wall']:.3f} ms, solver {row['solver_us_per_tick']:7.1f} us/tick "
                      f"({row['solver_us_per_launch']:.1f}/launch), cons {row['constraints_per_tick']:.0f}, frac {row['frac_of_hbm_roofline']}, "
                      f"fallbacks {row['flow6_fallbacks']} CL{row['flow6_const_lds']}NL{row['flow6_nimp_lds']}RL{row['flow6_rec_lds']}; hops in {h.get('in_block', {}).get('n')} x ({h.get('in_block', {}).get('handoff_us')}+{h.get('in_block', {}).get('service_us')}) "
                      f"cross {h.get('cross_block', {}).get('n')} x ({h.get('cross_block', {}).get('handoff_us')}+{h.get('cross_block', {}).get('service_us')}) "
                      f"span {t.get('span_us')} iter0 {t.get('iteration_complete_us', [None])[0]}", flush=True)


if __name__ == "__main__":
    main()
