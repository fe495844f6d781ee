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
                "k_import_ghost_vel", "k_compact_", "k_tick_snapshot")


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
