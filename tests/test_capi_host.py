"""CPU-side checks of the drop-in boundary and of the product's host logic (no GPU compute):
the library loads and exports every symbol include/mgf_hip.h declares; compute entry points fail
loudly without a device; the host-side BVH (insert / remove / balance / free-list reuse) is
structurally identical to the oracle's restatement of bvh.rs + pool.rs."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import mgf_amd
from mgf_amd import _capi, scenes
from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "mgf_hip.h")).read()
    return sorted(set(re.findall(r"MGF_API\s+[\w\s\*]+?\b(mgf_\w+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = mgf_amd.load_library()
    declared = _declared_symbols()
    assert len(declared) >= 45
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, missing
    assert sorted(_capi.SYMBOLS) == declared, "Python binding and header disagree"


def _wrappers_of(symbol, capi_src):
    """names of the Python functions / methods of mgf_amd/_capi.py whose body mentions `symbol` (a helper nested in a
    method counts for the method)"""
    names, stack = set(), []
    for line in capi_src.splitlines():
        m = re.match(r"(\s*)def (\w+)\(", line)
        if m:
            ind = len(m.group(1))
            stack = [(i, n) for i, n in stack if i < ind] + [(ind, m.group(2))]
        elif line.strip() and not line.startswith(" ") and not line.startswith(")"):
            stack = []
        if stack and re.search(r"\b%s\b" % re.escape(symbol), line):
            names.update(n for _, n in stack)
    return names


def test_every_export_is_called_by_some_test():
    """`mgf_world_get/set/integrate/complete_motion/read_colliders` were exported and never exercised.  Walk the
    header: every entry point must be reached from a test (or from smoke / bench, which the driver runs) - either named
    directly or through its wrapper in mgf_amd/_capi.py (a wrapper counts when a test calls it by name)."""
    capi_src = open(os.path.join(ROOT, "mgf_amd", "_capi.py")).read()
    callers = ""
    for d, names in ((os.path.join(ROOT, "tests"), None), (ROOT, ["__graft_entry__.py", "bench.py"]),
                     (os.path.join(ROOT, "mgf_amd"), ["tiles.py", "tiles_native.py"])):
        for f in sorted(names if names is not None else os.listdir(d)):
            path = os.path.join(d, f)
            if f.endswith(".py") and os.path.exists(path):
                callers += open(path).read() + "\n"
    # constructors / destructors / length are reached through the class, not by method name
    implicit = {"__init__": ("Context(", "Mesh(", "Bvh(", "World(", "Compound(", "Solver(", "World.from_scene("), "__del__": ("",),
                "close": ("ctx.close()", ".close()"), "__len__": ("len(",)}
    unreached = []
    for sym in _declared_symbols():
        if re.search(r"\b%s\b" % sym, callers):
            continue
        ws = _wrappers_of(sym, capi_src) - {"load_library"}  # (its signature table names every symbol)
        ok = False
        for w in ws:
            if w in implicit:
                ok = ok or any(tok in callers for tok in implicit[w])
            else:
                ok = ok or re.search(r"[\.\s\(]%s\(" % re.escape(w), callers) is not None
        if not ok:
            unreached.append((sym, sorted(ws)))
    assert not unreached, f"exported but never called by a test: {unreached}"


def test_pod_layouts_match_header():
    assert C.sizeof(_capi.Component) == 32 and C.sizeof(_capi.MovingComponent) == 44
    assert C.sizeof(_capi.Contact) == 40 and C.sizeof(_capi.LocalContact) == 64
    assert C.sizeof(_capi.Shape) == 52 and C.sizeof(_capi.BodyRef) == 24
    assert C.sizeof(_capi.Velocity) == 24 and C.sizeof(_capi.RigidBodyInfo) == 60
    assert C.sizeof(_capi.Params) == 20
    p = mgf_amd.default_params()
    assert (p.baumgarte, p.penetration_slop, p.persistent_threshold_sq, p.fat_margin) == (
        pytest.approx(0.2), pytest.approx(0.05), 0.5, 0.25)


def _no_gpu():
    try:
        import torch
        return not torch.cuda.is_available()
    except Exception:
        return True


@pytest.mark.skipif(not _no_gpu(), reason="needs a machine without a GPU")
def test_no_cpu_fallback():
    with pytest.raises(mgf_amd.MgfError) as e:
        mgf_amd.Context(0)
    assert e.value.status == _capi.ERR_HIP
    # a host-only tree exists, but querying it needs the device
    b = mgf_amd.Bvh(None)
    b.insert([0, 0, 0], [1, 1, 1], 7)
    with pytest.raises(mgf_amd.MgfError) as e:
        b.query([0, 0, 0], [1, 1, 1])
    assert e.value.status == _capi.ERR_HIP


def _compare_trees(hb, ob):
    gn, gb = hb.dump()
    on, obx = ob.dump()
    assert gn.shape == on.shape
    assert np.array_equal(gn, on), "node topology / heights / ids differ"
    used = gn[:, 0] == 1
    assert np.array_equal(gb[used].view(np.uint32), obx[used].view(np.uint32)), "node bounds differ"


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_host_bvh_matches_reference_restatement(seed):
    """Random insert/remove traffic (the World::step refit pattern, world.rs:235-238) on both trees."""
    rng = np.random.default_rng(seed)
    hb, ob = mgf_amd.Bvh(None), O.Bvh()
    live = {}
    for step in range(600):
        if live and rng.random() < 0.4:
            val = int(rng.choice(list(live)))
            hid, oid = live.pop(val)
            assert hid == oid
            hb.remove(hid)
            assert ob.remove(oid) == 0
        else:
            val = step
            c = rng.uniform(-20, 20, 3).astype(np.float32)
            r = rng.uniform(0.2, 2.0, 3).astype(np.float32)
            hid, oid = hb.insert(c, r, val), ob.insert(c, r, val)
            assert hid == oid  # LIFO slot reuse (pool.rs:81-113)
            live[val] = (hid, oid)
        if step % 50 == 0 and live:
            assert hb.root() == ob.root()
            _compare_trees(hb, ob)
    _compare_trees(hb, ob)
    # panics map to statuses
    with pytest.raises(mgf_amd.MgfError) as e:
        hb.remove(10 ** 6)
    assert e.value.status == _capi.ERR_NOT_OCCUPIED
    for hid, _ in list(live.values()):
        hb.remove(hid)
    assert hb.empty()
    with pytest.raises(mgf_amd.MgfError) as e:
        hb.root()
    assert e.value.status == _capi.ERR_EMPTY


def test_mesh_bvh_matches_reference_restatement():
    rng = np.random.default_rng(5)
    verts = rng.uniform(-10, 10, (300, 3)).astype(np.float32)
    faces = rng.integers(0, 300, (500, 3)).astype(np.uint32)
    m = mgf_amd.Mesh(None)
    m.build(verts, faces)
    m.set_pos([1.0, -2.0, 3.0])
    ow = O.World()
    ow.set_terrain(verts, faces, [1.0, -2.0, 3.0])
    gn, gb = m.bvh_dump()
    on, obx = ow.terrain_bvh().dump()
    assert np.array_equal(gn, on)
    assert np.array_equal(gb.view(np.uint32), obx.view(np.uint32))
    with pytest.raises(mgf_amd.MgfError):
        m.push_face(0, 1, 10 ** 6)  # out-of-bounds vertex (Vec index panic in mesh.rs:65-67)


@pytest.mark.parametrize("comp", [(0, [0.3, -1.0, 2.0], [0, 0, 0], 0.7), (1, [0.1, 0.2, 0.3], [0.5, -1.0, 0.25], 0.4),
                                  (1, [0, 0, 0], [0, 2.0, 0], 1.0), (1, [1, 1, 1], [0, -1.5, 0], 0.5)])
def test_inertia_tensor_matches_oracle(comp):
    tag, p, d, r = comp
    got = mgf_amd.inertia_tensor(tag, p, d, r, 2.5)
    out = (C.c_float * 9)()
    O.lib().mgfo_tensor(C.byref(O.component(tag, p, d, r)), 2.5, out)
    assert np.array_equal(np.asarray(got, np.float32).view(np.uint32), np.asarray(list(out), np.float32).view(np.uint32))


def test_scenes_are_deterministic():
    z = scenes.splitmix64(0x6D6766, 3)
    assert z.dtype == np.uint64 and len(set(z.tolist())) == 3
    assert np.array_equal(z, scenes.splitmix64(0x6D6766, 3))
    s1, s2 = scenes.sphere_pile(6, 5, 4), scenes.sphere_pile(6, 5, 4)
    assert np.array_equal(s1["comps"]["p"], s2["comps"]["p"]) and np.array_equal(s1["v0"], s2["v0"])
    assert len(s1["comps"]) == 120
    b = scenes.balls_demo(8)
    assert len(b["comps"]) == 512 and b["terrain"]["faces"].shape == (10, 3)
    # balls.rs:74-92 with num = 8: first body at (-5, 20, -5), pitch 1.25
    assert tuple(b["comps"]["p"][0]) == (-5.0, 20.0, -5.0) and tuple(b["comps"]["p"][1]) == (-5.0, 20.0, -3.75)
    full = scenes.balls_demo(11, extra_ball=True, iters=20)
    assert len(full["comps"]) == 1332 and tuple(full["comps"]["p"][-1]) == (0.0, 130.0, 0.0)


def test_integration_md_declares_every_export():
    """the Rust `extern "C"` block of INTEGRATION.md (what a maintainer of the reference would paste) against
    include/mgf_hip.h - the same set of entry points, with the same number of parameters each."""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r'^extern "C" \{\n(.*?)^\}', text, re.S | re.M)
    assert m, 'INTEGRATION.md has no extern "C" block'
    block = re.sub(r"//[^\n]*", "", m.group(1))
    rust = {}
    for fm in re.finditer(r"pub fn (mgf_\w+)\s*\((.*?)\)\s*(?:->\s*[^;]+)?;", block, re.S):
        args = fm.group(2).strip()
        depth, n = 0, 1 if args else 0   # top-level commas (a callback type `extern "C" fn(a, b)` holds commas of its own)
        for ch in args:
            depth += ch == "("
            depth -= ch == ")"
            n += ch == "," and depth == 0
        rust[fm.group(1)] = n
    header = open(os.path.join(ROOT, "include", "mgf_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    c = {}
    for fm in re.finditer(r"MGF_API\s+[\w\s\*]+?\b(mgf_\w+)\s*\((.*?)\)\s*;", header, re.S):
        args = fm.group(2).strip()
        if args == "void":
            c[fm.group(1)] = 0
            continue
        depth, n = 0, 1
        for ch in args:
            depth += ch == "("
            depth -= ch == ")"
            n += ch == "," and depth == 0
        c[fm.group(1)] = n
    assert set(c) == set(_declared_symbols()), "this test's header parser and _declared_symbols disagree"
    missing = sorted(set(c) - set(rust))
    extra = sorted(set(rust) - set(c))
    assert not missing, f"exports not declared in INTEGRATION.md: {missing}"
    assert not extra, f"INTEGRATION.md declares functions the header does not export: {extra}"
    wrong = {k: (rust[k], c[k]) for k in c if rust[k] != c[k]}
    assert not wrong, f"parameter counts differ (INTEGRATION.md, header): {wrong}"


def _c_structs(header):
    """typedef struct NAME { fields } NAME;  ->  {NAME: [(c type, field name, array length or None), ...]}"""
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    header = re.sub(r"//[^\n]*", "", header)
    out = {}
    for m in re.finditer(r"typedef\s+struct\s+(\w+)?\s*\{(.*?)\}\s*(\w+)\s*;", header, re.S):
        fields = []
        for decl in m.group(2).split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            ty, names = decl.rsplit(" ", 1)[0], decl.rsplit(" ", 1)[1]
            # `float a, b, c` style declarations: the type is everything before the first name
            parts = [p.strip() for p in decl.split(",")]
            first = parts[0]
            ty = first[:first.rfind(" ")].strip()
            for name in [first[first.rfind(" ") + 1:]] + parts[1:]:
                am = re.match(r"(\w+)\[(\w+)\]$", name)
                fields.append((ty, am.group(1), am.group(2)) if am else (ty, name, None))
        out[m.group(3)] = fields
    return out


def _rust_structs(text):
    """#[repr(C)] ... pub struct NAME { pub a: T, pub b: [T; N], ... }  ->  {NAME: [(rust type, field, array length or None), ...]}"""
    out = {}
    for m in re.finditer(r"#\[repr\(C\)\][^\n]*?pub struct (\w+)\s*\{(.*?)\n?\}", text, re.S):
        body = re.sub(r"//[^\n]*", "", m.group(2))
        fields = []
        for fm in re.finditer(r"pub (\w+)\s*:\s*(\[[^\]]+\]|[\w:\*\s]+?)\s*(?:,|$)", body, re.S):
            ty = " ".join(fm.group(2).split())
            am = re.match(r"\[\s*([\w:]+)\s*;\s*(\w+)\s*\]$", ty)
            fields.append((am.group(1), fm.group(1), am.group(2)) if am else (ty, fm.group(1), None))
        out[m.group(1)] = fields
    return out


_RUST_OF_C = {"float": "f32", "int32_t": "i32", "uint32_t": "u32", "int64_t": "i64", "uint64_t": "u64", "double": "f64", "uint8_t": "u8", "int": "i32"}


def test_integration_md_repr_c_structs_match_the_header_field_for_field():
    """the Rust shim of INTEGRATION.md can never be compiled here (no rustc), so the one check its `#[repr(C)]` structs
    can get is mechanical: every struct the header defines appears there with the same fields - names, order, types (f32 for float, i32
    for int32_t, nested mgf_* structs by name, arrays by length) - because a swapped pair of floats would compile on both sides and
    read garbage across the boundary."""
    c = _c_structs(open(os.path.join(ROOT, "include", "mgf_hip.h")).read())
    rust = _rust_structs(open(os.path.join(ROOT, "INTEGRATION.md")).read())
    assert len(c) >= 15, sorted(c)
    missing = sorted(set(c) - set(rust))
    assert not missing, f"structs of the header without a #[repr(C)] twin in INTEGRATION.md: {missing}"
    consts = dict(re.findall(r"#define\s+(MGF_\w+)\s+(\d+)", open(os.path.join(ROOT, "include", "mgf_hip.h")).read()))
    for name, cf in c.items():
        rf = rust[name]
        assert [f[1] for f in cf] == [f[1] for f in rf], f"{name}: field names / order differ: header {[f[1] for f in cf]} vs INTEGRATION.md {[f[1] for f in rf]}"
        for (cty, fname, clen), (rty, _rn, rlen) in zip(cf, rf):
            want = _RUST_OF_C.get(cty, cty)
            assert rty == want, f"{name}.{fname}: header type {cty} (Rust {want}) vs INTEGRATION.md {rty}"
            cl = consts.get(clen, clen) if clen else None
            rl = consts.get(rlen, rlen) if rlen else None
            assert cl == rl, f"{name}.{fname}: array length {clen} vs {rlen}"
