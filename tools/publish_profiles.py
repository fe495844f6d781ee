#!/usr/bin/env python3
"""Copies one collection of tools/collect_profiles.sh (gpurun_out/<tag>_*) into profiles/ (tracked) and assembles the PMC records
bench.py's `roofline.traffic` reads: profiles/pmc_k_solve_flow6.json = the LIST of the falling-pile windows that were profiled
(bench.py's default window and the driver's own --warmup 5 --steps 20), _settled, and one file per other BASELINE config.
Usage: python tools/publish_profiles.py <tag>"""
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out")
dst = os.path.join(ROOT, "profiles")
keep = ("_bench.json", "_bench_under_trace.json", "_kernel_stats.txt", "_pmc_hbm_traffic.txt", "_bench_default.json", "_bench_driver_command.json",
        "_config4_8tiles_1gpu_bench.json", "_config4_undivided_1gpu.json", "_config5_8tiles_1gpu_bench.json", "_config5_undivided_1gpu.json",
        "_gputests.txt", "_soak_solver_modes.txt", "_soak_scenes.txt", "_soak_tiles.txt",
        "_refresh_every_2.json", "_refresh_every_3.json", "_refresh_every_4.json", "_refresh_every_5.json")
n = 0
for p in sorted(glob.glob(os.path.join(src, tag + "_*"))):
    if os.path.isfile(p) and p.endswith(keep) and os.path.getsize(p) > 0:
        shutil.copy(p, os.path.join(dst, os.path.basename(p)))
        n += 1


def rec(name):
    p = os.path.join(src, f"{tag}_{name}_pmc.json")
    return json.load(open(p)) if os.path.exists(p) else None


def put(name, obj):
    if obj:
        json.dump(obj, open(os.path.join(dst, name), "w"), indent=1)
        print("wrote", name)


put("pmc_k_solve_flow6.json", [r for r in (rec("transient"), rec("driver")) if r])
put("pmc_k_solve_flow6_settled.json", [r for r in (rec("settled"), rec("settled20")) if r])
put("pmc_config3_k_solve_flow6.json", [r for r in (rec("config3"), rec("config3_20")) if r])
put("pmc_config5_k_solve_flow6.json", [r for r in (rec("config5"), rec("config5_20")) if r])
for name in ("config4_tiles", "config5_tiles"):
    r = rec(name)
    if r:
        r["note"] = (f"rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `python bench.py --gpus 1 --scene {name.replace('config4_tiles', 'config4')} "
                     "--no-cpu-baseline` (8 x-slab tiles on one GPU, tools/collect_profiles.sh); average over the LAST 1440 launches of k_solve_flow6 = the 60 "
                     "instrumented ticks x 8 tiles x 3 launches of 4 + 4 + 2 iterations (R = 4); counter unit KiB; hbm_bytes_per_launch = 2*FETCH + WRITE "
                     "(gfx950 FETCH_SIZE reads 1/2 of 16-B/lane loads, MI355X_MICROARCH.md HBM section)")
        put(f"pmc_{name}_k_solve_flow6.json", [r])
print(n, "files copied under profiles/ with tag", tag)
