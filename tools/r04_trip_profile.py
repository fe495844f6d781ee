"""Where a trip of a serving wave goes (MGF_F6_PROFILE build: tools/build_variant.sh prof -DMGF_F6_PROFILE=1; load it with MGF_AMD_LIB).
Shader clocks per trip of wave 0, averaged over the blocks of one traced launch: pop / operand loads / solve + store / release."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import mgf_amd  # noqa: E402
from tools.r04_block_sweep import SCENES, run_ticks  # noqa: E402
from tools import flowtrace_lib as FT  # noqa: E402

names = sys.argv[1].split(",")
opts = [kv.split("=") for kv in sys.argv[2].split(",")] if len(sys.argv) > 2 and sys.argv[2] else []
ctx = mgf_amd.Context(0)
for name in names:
    build, warm, steps, iters, launches = SCENES[name]
    sc = build()
    dt = float(sc["dt"])
    w = mgf_amd.World.from_scene(ctx, sc)
    for k, v in opts:
        w.set_option(k, int(v))
    run_ticks(w, dt, iters, launches, warm + steps // 2)
    w.set_option("time_solver_kernels", 1)
    per = run_ticks(w, dt, iters, launches, 4)
    w.set_option("time_solver_kernels", 0)
    w.set_option("flow_trace", 1)
    if launches == 1:
        w.step(dt, iters)
    else:
        w.build_constraints(dt); w.solve(iters)
    w.set_option("flow_trace", 0)
    p = np.fromfile(FT.POLL, dtype=np.uint64).reshape(-1, FT.WORDS).astype(np.float64)
    p = p[p[:, 17] > 0]
    trips = p[:, 17].sum()
    serving = np.mean(p[:, 11] - p[:, 10]) * 0.01
    tot = (p[:, 13] + p[:, 14] + p[:, 15] + p[:, 16]).sum() / trips
    print(f"{name:16s} {dict(opts)} solver {np.mean([x[1] for x in per]) * 1e3 / launches:7.1f} us/launch untraced; traced: serving phase {serving:.1f} us; wave 0: {trips / len(p):.1f} trips/block, "
          f"{p[:, 18].sum() / trips:.2f} nodes/trip; clocks/trip: pop {p[:, 13].sum() / trips:.0f} loads {p[:, 14].sum() / trips:.0f} (LDS part {p[:, 21].sum() / trips:.0f}) solve+store {p[:, 15].sum() / trips:.0f} "
          f"release {p[:, 16].sum() / trips:.0f} = {tot:.0f}; idle polls/block {p[:, 20].sum() / len(p):.0f} x {p[:, 19].sum() / max(p[:, 20].sum(), 1):.0f} clocks", flush=True)
