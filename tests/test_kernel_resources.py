"""What the build must keep true of the tick's kernels (read from the notes of libmgf_hip.so's gfx950 code object, no GPU needed): no scratch
memory - DESIGN.md says so - and no register spills in k_contacts_rows<true> (round 5's k_contacts_spheres), whose one device fault of round 5 went away with them
(EXPERIMENTS.md: the listing of a block's later windows, inlined a second time, spilled scalar registers inside nested branches)."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TICK_KERNELS = ("k_integrate", "k_tick_clear", "k_scan", "k_scatter_leaves", "k_pair_brick", "k_pair_grid", "k_terrain_contacts", "k_contacts_rows", "k_near_list", "k_terrain_near", "k_terrain_tests",
                "k_flow6_blocks", "k_flow6_links", "k_publish", "k_setup_pairs", "k_lists_spheres", "k_narrow_pairs<", "k_narrow_terrain<", "k_count_contacts",
                "k_rows_to_csr", "k_morton_count", "k_zero_many", "k_reset_step", "k_tile_select", "k_export_bodies", "k_import_ghosts", "k_export_vel",
                "k_import_ghost_vel", "k_compact_", "k_tick_snapshot", "k_pair_wide", "k_narrow_pairs_big", "k_narrow_terrain_big", "k_narrow_pairs_parts", "k_narrow_terrain_parts")


def _rows():
    if not os.path.exists(os.path.join(ROOT, "mgf_amd", "libmgf_hip.so")) or not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-readelf"):
        pytest.skip("needs the built library and the ROCm LLVM tools")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_resources.py")], capture_output=True, text=True, check=True).stdout
    rows = {}
    for line in out.splitlines()[1:]:
        m = re.match(r"(.{80}) +(\S+) +(\S+) +(\S+) +(\S+) +(\S+) +(\S+)$", line)
        if m:
            rows[m.group(1).strip()] = tuple(m.group(k) for k in range(2, 8))  # vgpr, sgpr, scratch, lds, sgpr spills, vgpr spills
    assert len(rows) > 100, out[:500]
    return rows


def test_no_kernel_of_the_tick_uses_scratch_memory():
    rows = _rows()
    bad = {k: v for k, v in rows.items() if any(k.startswith(p) for p in TICK_KERNELS) and not k.startswith("k_solve_flow6<true") and v[2] not in ("0", "?")}
    assert not bad, bad
    solver = {k: v for k, v in rows.items() if k.startswith("k_solve_flow6<false")}
    assert solver and all(v[2] == "0" and v[4] == "0" and v[5] == "0" for v in solver.values()), solver


def test_k_contacts_rows_of_spheres_spills_nothing():
    v = _rows()["k_contacts_rows<true>"]
    assert v[2] == "0" and v[4] == "0" and v[5] == "0", v


def _lane_mask_checker():
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_lane_masks", os.path.join(ROOT, "tools", "check_lane_masks.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_the_lane_mask_checker_finds_round_fives_fault():
    """The root cause of round 5's device fault (EXPERIMENTS.md, round 6): a wave-uniform boolean's lane mask re-derived inside a loop that lanes
    leave one by one, read behind the loop under an exec of other lanes.  The checker against the disassembly of that very kernel."""
    cl = _lane_mask_checker()
    body = []
    for line in open(os.path.join(ROOT, "tests", "golden", "r05_fault_isa_excerpt.txt")):
        if line.startswith("0x"):
            a, ins = line.rstrip("\n").split(": ", 1)
            body.append((int(a, 16), ins))
    found = cl.check(body)
    assert [(a, a2) for a, _, a2, _ in found] == [(0x4008C, 0x40220)], found
    # ... and not where the mask is derived AHEAD of the loop (what the first copy of the same listing did, and what every kernel does now)
    moved = [(a, ins) for a, ins in body if a != 0x4008C]
    k = next(i for i, (a, _) in enumerate(moved) if a == 0x40058)
    moved.insert(k, (0x40056, "v_cmp_ne_u32_e64 s[0:1], 1, v71"))
    assert cl.check(moved) == []


def test_no_kernel_reads_a_lane_mask_behind_the_divergent_loop_that_made_it():
    """... on every kernel of the library (the tick's, the tile protocol's, the single-shot entry points'): 0 such pairs."""
    lib = os.path.join(ROOT, "mgf_amd", "libmgf_hip.so")
    if not os.path.exists(lib) or not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"):
        pytest.skip("needs the built library and the ROCm LLVM tools")
    cl = _lane_mask_checker()
    bad, n = [], 0
    for name, body in cl.kernels(lib):
        n += 1
        bad += [(name, hex(a), ins, hex(a2), ins2) for a, ins, a2, ins2 in cl.check(body)]
    assert n > 300 and not bad, bad
