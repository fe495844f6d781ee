"""Development check of solver mode 6 (k_solve_flow6): bit-equality with the launch-per-frontier solver (mode 0) on small
scenes with small blocks, and with mode 5 on the full pile, plus timings."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
import mgf_amd
from mgf_amd import scenes

ctx = mgf_amd.Context(0)
OPTS = [kv.split("=") for kv in os.environ.get("MGF_F6_OPTS", "").split(",") if kv]


def same(a, b):
    sa, sb = a.state(), b.state()
    return all(np.array_equal(sa[k].view(np.uint32), sb[k].view(np.uint32)) for k in sa)


def run(scene, ticks, ref_mode, block=None, label=""):
    a, b = mgf_amd.World.from_scene(ctx, scene), mgf_amd.World.from_scene(ctx, scene)
    a.set_option("solver_mode", ref_mode); b.set_option("solver_mode", 6)
    for k, v in OPTS:
        b.set_option(k, int(v))
    if block:
        a.set_option("flow5_block", block); b.set_option("flow5_block", block)
    dt = float(scene["dt"])
    bad = None
    for k in range(ticks):
        sa, sb = a.step(dt, 10), b.step(dt, 10)
        if not same(a, b):
            bad = k
            break
    print(f"{label}: {len(a)} bodies, {ticks} ticks vs mode {ref_mode}: {'EQUAL' if bad is None else 'DIFFERS at tick %d' % bad}; "
          f"constraints {sb.n_constraints}, fallbacks {b.counter('flow6_fallbacks')}, max slots {b.counter('flow6_max_slots')} / cap {b.counter('flow6_slot_cap')}, "
          f"max foreign {b.counter('flow6_max_foreign')} / cap {b.counter('flow6_fcap')}", flush=True)
    return bad is None


ok = True
if len(sys.argv) < 2 or sys.argv[1] != "big":
    ok &= run(scenes.sphere_pile(12, 12, 12), 80, 0, block=64, label="pile 12^3, blocks of 64")
    ok &= run(scenes.sphere_pile(16, 16, 16), 60, 0, block=128, label="pile 16^3, blocks of 128")
    ok &= run(scenes.capsule_field_dense(16, 2, 16), 60, 0, block=64, label="capsules 16x2x16, blocks of 64")
    ok &= run(scenes.sphere_pile(20, 20, 20), 40, 1, label="pile 20^3, default blocks")
sc = scenes.sphere_pile(64, 64, 64)
ok &= run(sc, 30, 5, label="pile 64^3")
for mode in ((6,) if os.environ.get('MGF_F6_ONLY') else (5, 6)):
    w = mgf_amd.World.from_scene(ctx, sc)
    w.set_option('phase_timing', 1)
    w.set_option("solver_mode", mode)
    for k, v in OPTS:
        w.set_option(k, int(v))
    w.set_option("time_solver_kernels", 1)
    dt = float(sc["dt"])
    for _ in range(10):
        w.step(dt, 10)
    st = w.step_many(dt, 10, 60)
    ms = np.array([s.ms_solver_kernels for s in st]); tot = np.array([s.ms_total for s in st]); sol = np.array([s.ms_solve for s in st])
    cons = np.array([s.n_constraints for s in st])
    print(f"mode {mode}: ticks 10-70: solver kernel {ms.mean()*1e3:.1f} us, solve phase {sol.mean()*1e3:.1f} us, tick {tot.mean()*1e3:.1f} us; "
          f"frac {288 * 10 * cons.mean() / (ms.mean() * 1e-3) / 8e12:.3f}", flush=True)
    if len(sys.argv) > 2:
        for _ in range(int(sys.argv[2])):
            w.step(dt, 10)
        st = w.step_many(dt, 10, 40)
        ms = np.array([s.ms_solver_kernels for s in st]); cons = np.array([s.n_constraints for s in st])
        print(f"mode {mode}: after {sys.argv[2]} more ticks: solver kernel {ms.mean()*1e3:.1f} us, constraints {cons.mean():.0f}, frac {288 * 10 * cons.mean() / (ms.mean() * 1e-3) / 8e12:.3f}; "
              f"fallbacks {w.counter('flow6_fallbacks') if mode == 6 else w.counter('flow5_fallbacks')} reason {w.counter('flow6_fail_reason')} "
              f"slots {w.counter('flow6_max_slots')}/{w.counter('flow6_slot_cap')} foreign {w.counter('flow6_max_foreign')}/{w.counter('flow6_fcap')}", flush=True)
print("ALL EQUAL" if ok else "MISMATCH")
