"""A compiled C99 client of the boundary: tests/c_client/balls.c includes include/mgf_hip.h, is compiled by gcc
-std=c99 -pedantic and linked against libmgf_hip.so - the exact layer INTEGRATION.md's Rust `extern "C"` block binds (by-value mgf_vec3,
callbacks, opaque handles).  It builds BASELINE config 1 (mgf_demo/balls.rs:67-96 on the terrain of world.rs:118-150) call by call and
steps it 300 ticks; its raw f32 states must equal tests/golden/world_snapshots.npz bit for bit, in both constraint orders."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c_client", "balls.c")
LIBDIR = os.path.join(ROOT, "mgf_amd")


def _compile(tmp_path):
    exe = str(tmp_path / "balls")
    cmd = ["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-O1", "-I", os.path.join(ROOT, "include"), SRC, "-o", exe,
           "-L", LIBDIR, "-lmgf_hip", "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    return exe


def test_the_c_client_compiles_and_links_against_the_library(tmp_path):
    """(no GPU needed: the header is C99-clean as a client sees it and every symbol the client uses resolves)"""
    if not os.path.exists(os.path.join(LIBDIR, "libmgf_hip.so")):
        from mgf_amd import build
        build.build()
    exe = _compile(tmp_path)
    assert os.path.getsize(exe) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("order,name", [(0, "canonical"), (1, "demo")])
def test_the_c_client_reproduces_the_golden_snapshots(tmp_path, order, name):
    exe = _compile(tmp_path)
    out = str(tmp_path / "states.bin")
    p = subprocess.run([exe, out, "8", "10", str(order)], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr
    golden = np.load(os.path.join(ROOT, "tests", "golden", "world_snapshots.npz"))
    raw = np.fromfile(out, dtype=np.uint8)
    n, pos = 512, 0
    for k in (1, 2, 10, 60, 300):
        head = raw[pos:pos + 16].view(np.uint64); pos += 16
        assert int(head[0]) == k and int(head[1]) == int(golden[f"balls512/{name}/{k}/n_constraints"])
        for f, w in (("x", 3), ("q", 4), ("v", 3), ("omega", 3)):
            a = raw[pos:pos + 4 * w * n].view(np.float32).reshape(n, w); pos += 4 * w * n
            g = golden[f"balls512/{name}/{k}/{f}"]
            assert np.array_equal(a.view(np.uint32), g.view(np.uint32)), f"tick {k} {f}: max |diff| {np.max(np.abs(a - g))}"
    assert pos == len(raw)
    # the BVH built and queried through callbacks: the balls whose fat boxes meet the query box, by brute force
    words = p.stdout.split()
    rad, shift = np.float32(0.5), np.float32(1.25)
    i, j, kk = np.meshgrid(np.arange(8), np.arange(8), np.arange(8), indexing="ij")
    c = np.stack([i.ravel() * shift - np.float32(5.0), np.float32(10.0) + j.ravel() * shift + np.float32(10.0), kk.ravel() * shift - np.float32(5.0)], axis=1).astype(np.float32)
    qc, qr = np.array([0.0, c[0, 1] + 2.0, 0.0], np.float32), np.array([2.0, 1.0, 3.0], np.float32)
    hit = np.all(np.abs(c - qc) <= qr + np.float32(0.75), axis=1)
    assert int(words[1]) == int(hit.sum()) and int(words[3]) == int(np.nonzero(hit)[0].sum())
    assert float(words[-1]) == 1.0
