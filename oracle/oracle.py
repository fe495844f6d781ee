"""ORACLE — TEST INFRASTRUCTURE ONLY.

ctypes binding of oracle/libmgf_oracle.so (the CPU restatement of mgf's hot path).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.  Nothing under mgf_amd/ imports it; the product path fails loudly when the
HIP extension is missing instead of falling back to this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libmgf_oracle.so")


def build(force=False):
    """Compile the oracle with g++ (no GPU needed)."""
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".hpp", ".cpp"))]
    if (not force and os.path.exists(_LIB_PATH)
            and all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in srcs)):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B", "libmgf_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


class Vec3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]

    def tup(self):
        return (self.x, self.y, self.z)


class Quat(C.Structure):
    _fields_ = [("s", C.c_float), ("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]


class Aabb(C.Structure):
    _fields_ = [("c", Vec3), ("r", Vec3)]


class Contact(C.Structure):
    _fields_ = [("a", Vec3), ("b", Vec3), ("n", Vec3), ("t", C.c_float)]


class LocalContact(C.Structure):
    _fields_ = [("local_a", Vec3), ("local_b", Vec3), ("glob", Contact)]


class Shape(C.Structure):
    _fields_ = [("kind", C.c_int32), ("v", C.c_float * 12)]


class Component(C.Structure):
    _fields_ = [("tag", C.c_int32), ("p", Vec3), ("d", Vec3), ("r", C.c_float)]


class Stats(C.Structure):
    _fields_ = [("n_constraints", C.c_uint64), ("n_terrain_constraints", C.c_uint64),
                ("n_pair_candidates", C.c_uint64), ("n_refits", C.c_uint64),
                ("t_integrate", C.c_double), ("t_collide", C.c_double), ("t_solve", C.c_double)]


class Constraint(C.Structure):
    _fields_ = [("a", C.c_int32), ("b", C.c_int32), ("n_contacts", C.c_int32),
                ("normal", Vec3), ("t0", Vec3), ("t1", Vec3), ("ra", Vec3), ("rb", Vec3),
                ("bias", C.c_float), ("normal_mass", C.c_float), ("tangent_mass0", C.c_float),
                ("tangent_mass1", C.c_float), ("normal_impulse", C.c_float), ("friction", C.c_float)]


# numpy views of the PODs (same memory layout)
COMPONENT_DTYPE = np.dtype([("tag", "<i4"), ("p", "<f4", 3), ("d", "<f4", 3), ("r", "<f4")])
CONSTRAINT_DTYPE = np.dtype([("a", "<i4"), ("b", "<i4"), ("n_contacts", "<i4"),
                             ("normal", "<f4", 3), ("t0", "<f4", 3), ("t1", "<f4", 3),
                             ("ra", "<f4", 3), ("rb", "<f4", 3),
                             ("bias", "<f4"), ("normal_mass", "<f4"), ("tangent_mass0", "<f4"),
                             ("tangent_mass1", "<f4"), ("normal_impulse", "<f4"), ("friction", "<f4")])
assert COMPONENT_DTYPE.itemsize == C.sizeof(Component)
assert CONSTRAINT_DTYPE.itemsize == C.sizeof(Constraint)

SPHERE, CAPSULE, TRIANGLE, RECTANGLE, PLANE = 0, 1, 2, 3, 4
ORDER_DEMO, ORDER_CANONICAL = 0, 1

_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        P = C.POINTER
        L.mgfo_contacts.argtypes = [P(Shape), P(Vec3), P(Shape), P(Vec3), P(Contact), C.c_int]
        L.mgfo_contacts.restype = C.c_int
        L.mgfo_local_contacts_pair.argtypes = [P(Component), P(Vec3), P(Component), P(Vec3), P(LocalContact), C.c_int]
        L.mgfo_local_contacts_pair.restype = C.c_int
        for f in (L.mgfo_ray_capsule, L.mgfo_ray_sphere):
            f.argtypes = [P(Vec3), P(Vec3), P(Shape), P(Vec3), P(C.c_float)]
            f.restype = C.c_int
        L.mgfo_intersection.argtypes = [P(Vec3), P(Vec3), C.c_float, P(Shape), P(Vec3), P(C.c_float)]
        L.mgfo_intersection.restype = C.c_int
        L.mgfo_intersection_aabb.argtypes = [P(Vec3), P(Vec3), C.c_float, P(Aabb), P(Vec3), P(C.c_float)]
        L.mgfo_intersection_aabb.restype = C.c_int
        L.mgfo_bvh_raytrace.argtypes = [C.c_void_p, P(Vec3), P(Vec3), C.c_float, P(C.c_uint64), P(Vec3), P(C.c_float), C.c_int64]
        L.mgfo_bvh_raytrace.restype = C.c_int64
        L.mgfo_compound_new.argtypes = [P(Component), C.c_int64]
        L.mgfo_compound_new.restype = C.c_void_p
        L.mgfo_compound_free.argtypes = [C.c_void_p]
        L.mgfo_compound_set_pose.argtypes = [C.c_void_p, P(Vec3), P(Quat)]
        L.mgfo_world_add_obstacle.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, P(Vec3), P(Quat)]
        L.mgfo_compound_bounds.argtypes = [C.c_void_p, P(Aabb)]
        L.mgfo_compound_contacts.argtypes = [C.c_void_p, P(Shape), P(Vec3), P(Contact), C.c_int]
        L.mgfo_compound_contacts.restype = C.c_int
        L.mgfo_compound_intersection.argtypes = [C.c_void_p, P(Vec3), P(Vec3), C.c_float, P(Vec3), P(C.c_float)]
        L.mgfo_compound_intersection.restype = C.c_int
        L.mgfo_manifold_from_contacts.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, P(C.c_int32), C.c_void_p, C.c_int32]
        L.mgfo_tri_closest_point.argtypes = [P(Shape), P(Vec3), P(Vec3)]
        L.mgfo_compute_basis.argtypes = [P(Vec3), P(Vec3)]
        L.mgfo_quat_from_arc.argtypes = [P(Vec3), P(Vec3), P(Quat)]
        L.mgfo_rotate_vector.argtypes = [P(Quat), P(Vec3), P(Vec3)]
        L.mgfo_tensor.argtypes = [P(Component), C.c_float, P(C.c_float)]
        L.mgfo_component_bounds.argtypes = [P(Component), P(Vec3), P(Aabb)]
        L.mgfo_aabb_combine.argtypes = [P(Aabb), P(Aabb), P(Aabb)]
        L.mgfo_aabb_overlaps.argtypes = [P(Aabb), P(Aabb)]
        L.mgfo_aabb_overlaps.restype = C.c_int
        L.mgfo_aabb_contains.argtypes = [P(Aabb), P(Aabb)]
        L.mgfo_aabb_contains.restype = C.c_int
        L.mgfo_pool_new.restype = C.c_void_p
        L.mgfo_pool_free.argtypes = [C.c_void_p]
        L.mgfo_pool_push.argtypes = [C.c_void_p, C.c_uint64]
        L.mgfo_pool_push.restype = C.c_int64
        L.mgfo_pool_remove.argtypes = [C.c_void_p, C.c_uint64, P(C.c_uint64)]
        L.mgfo_pool_remove.restype = C.c_int
        L.mgfo_pool_get.argtypes = [C.c_void_p, C.c_uint64, P(C.c_uint64)]
        L.mgfo_pool_get.restype = C.c_int
        L.mgfo_pool_iter.argtypes = [C.c_void_p, P(C.c_uint64), P(C.c_uint64), C.c_int64]
        L.mgfo_pool_iter.restype = C.c_int64
        L.mgfo_bvh_new.restype = C.c_void_p
        L.mgfo_bvh_free.argtypes = [C.c_void_p]
        L.mgfo_bvh_insert.argtypes = [C.c_void_p, P(Aabb), C.c_uint64]
        L.mgfo_bvh_insert.restype = C.c_int64
        L.mgfo_bvh_remove.argtypes = [C.c_void_p, C.c_uint64]
        L.mgfo_bvh_remove.restype = C.c_int
        L.mgfo_bvh_root.argtypes = [C.c_void_p]
        L.mgfo_bvh_root.restype = C.c_int64
        L.mgfo_bvh_bounds.argtypes = [C.c_void_p, C.c_uint64, P(Aabb)]
        L.mgfo_bvh_bounds.restype = C.c_int
        L.mgfo_bvh_get_leaf.argtypes = [C.c_void_p, C.c_uint64, P(C.c_uint64)]
        L.mgfo_bvh_get_leaf.restype = C.c_int
        L.mgfo_bvh_query.argtypes = [C.c_void_p, P(Aabb), P(C.c_uint64), C.c_int64]
        L.mgfo_bvh_query.restype = C.c_int64
        L.mgfo_bvh_pool.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, P(C.c_int64), P(C.c_int64)]
        L.mgfo_bvh_pool.restype = C.c_int64
        L.mgfo_bvh_dump.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
        L.mgfo_bvh_dump.restype = C.c_int64
        L.mgfo_world_new.argtypes = [C.c_int]
        L.mgfo_world_new.restype = C.c_void_p
        L.mgfo_world_free.argtypes = [C.c_void_p]
        L.mgfo_world_set_params.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float]
        L.mgfo_world_set_terrain.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, P(Vec3)]
        L.mgfo_world_add_bodies.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.mgfo_world_add_bodies.restype = C.c_int64
        L.mgfo_world_add_compound_bodies.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.mgfo_world_add_compound_bodies.restype = C.c_int64
        L.mgfo_world_body_info.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
        L.mgfo_world_len.argtypes = [C.c_void_p]
        L.mgfo_world_len.restype = C.c_int64
        L.mgfo_world_step.argtypes = [C.c_void_p, C.c_float, C.c_int64, P(Stats)]
        L.mgfo_world_build_constraints.argtypes = [C.c_void_p, C.c_float, P(Stats)]
        L.mgfo_world_solve.argtypes = [C.c_void_p, C.c_int64]
        L.mgfo_world_begin_tick.argtypes = [C.c_void_p, C.c_float]
        L.mgfo_world_collide.argtypes = [C.c_void_p, C.c_float, P(Stats)]
        L.mgfo_world_select_boundary.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_int64, P(C.c_int64), P(C.c_int64)]
        L.mgfo_world_export_bodies.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        L.mgfo_world_select_migrants.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_int64, P(C.c_int64), P(C.c_int64)]
        L.mgfo_world_export_migrants.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        L.mgfo_world_remove_bodies.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        L.mgfo_world_import_migrants.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        L.mgfo_world_set_tags.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        L.mgfo_world_read_tags.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        L.mgfo_world_import_ghosts.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        L.mgfo_world_export_velocities.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        L.mgfo_world_import_ghost_velocities.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        L.mgfo_world_constraint_depth.argtypes = [C.c_void_p, C.c_uint32]
        L.mgfo_world_constraint_depth.restype = C.c_uint32
        L.mgfo_world_get_constraints.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        L.mgfo_world_get_constraints.restype = C.c_int64
        L.mgfo_world_get_state.argtypes = [C.c_void_p] + [C.c_void_p] * 5
        L.mgfo_world_set_state.argtypes = [C.c_void_p] + [C.c_void_p] * 5
        L.mgfo_world_get_colliders.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.mgfo_world_get_inv_moment.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.mgfo_world_set_velocity.argtypes = [C.c_void_p, C.c_int64, P(Vec3), P(Vec3)]
        L.mgfo_world_terrain_contacts.argtypes = [C.c_void_p, C.c_int64, P(LocalContact), C.c_int64]
        L.mgfo_world_terrain_contacts.restype = C.c_int64
        L.mgfo_world_terrain_bvh.argtypes = [C.c_void_p]
        L.mgfo_world_terrain_bvh.restype = C.c_void_p
        L.mgfo_world_bvh.argtypes = [C.c_void_p]
        L.mgfo_world_bvh.restype = C.c_void_p
        L.mgfo_world_integrate.argtypes = [C.c_void_p, C.c_float]
        L.mgfo_world_complete_motion.argtypes = [C.c_void_p]
        L.mgfo_world_get.argtypes = [C.c_void_p, C.c_int32, C.c_int64, P(Vec3), C.c_float, C.c_void_p]
        L.mgfo_constraint_new.argtypes = [C.c_void_p, C.c_int64, C.c_int64, P(Vec3), C.c_float, P(Vec3), C.c_void_p, C.c_int64,
                                          C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]
        L.mgfo_solver_solve.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64]
        _lib = L
    return _lib


def vec3(v):
    return Vec3(float(v[0]), float(v[1]), float(v[2]))


def shape(kind, *vals):
    s = Shape()
    s.kind = kind
    flat = []
    for v in vals:
        if isinstance(v, (int, float)):
            flat.append(float(v))
        else:
            flat.extend(float(x) for x in v)
    for i, x in enumerate(flat):
        s.v[i] = x
    return s


def shape_from_dict(d):
    k = d["kind"]
    if k == "sphere":
        return shape(SPHERE, d["c"], d["r"])
    if k == "capsule":
        return shape(CAPSULE, d["a"], d["d"], d["r"])
    if k == "triangle":
        return shape(TRIANGLE, d["a"], d["b"], d["c"])
    if k == "rectangle":
        return shape(RECTANGLE, d["c"], d["u0"], d["u1"], d["e"][0], d["e"][1])
    if k == "plane":
        return shape(PLANE, d["n"], d["d"])
    raise ValueError(k)


def contacts(a, vel_a, b, vel_b, cap=8):
    """Contacts::contacts of the reference for (a [moving at vel_a]) vs (b [moving at vel_b])."""
    out = (Contact * cap)()
    va = C.byref(vec3(vel_a)) if vel_a is not None else None
    vb = C.byref(vec3(vel_b)) if vel_b is not None else None
    n = lib().mgfo_contacts(C.byref(a), va, C.byref(b), vb, out, cap)
    if n < 0:
        raise ValueError("unsupported shape pair")
    return [dict(a=out[i].a.tup(), b=out[i].b.tup(), n=out[i].n.tup(), t=out[i].t) for i in range(min(n, cap))]


def intersection(p, d, dt, sh):
    """Intersects<shape> for a Ray (dt = inf) or a Segment (dt = 1, p = a, d = b - a): (point, t) or None."""
    ip, t = Vec3(), C.c_float()
    r = lib().mgfo_intersection(C.byref(vec3(p)), C.byref(vec3(d)), C.c_float(dt), C.byref(sh), C.byref(ip), C.byref(t))
    if r < 0:
        raise ValueError("unsupported shape")
    return (ip.tup(), t.value) if r else None


def intersection_aabb(p, d, dt, c, r):
    ip, t = Vec3(), C.c_float()
    box = Aabb(vec3(c), vec3(r))
    hit = lib().mgfo_intersection_aabb(C.byref(vec3(p)), C.byref(vec3(d)), C.c_float(dt), C.byref(box), C.byref(ip), C.byref(t))
    return (ip.tup(), t.value) if hit else None


LOCAL_CONTACT_DTYPE = np.dtype([("local_a", "<f4", 3), ("local_b", "<f4", 3), ("a", "<f4", 3), ("b", "<f4", 3), ("n", "<f4", 3), ("t", "<f4")])


def manifold_from_contacts(lcs, cap=64):
    """ContactPruner::push for every LocalContact (LOCAL_CONTACT_DTYPE rows, in order), then Manifold::from(pruner):
    dict(time, normal, t0, t1, pairs[(local_a, local_b)])."""
    lcs = np.ascontiguousarray(lcs, LOCAL_CONTACT_DTYPE)
    out = np.zeros(10, np.float32)
    pairs = np.zeros((cap, 6), np.float32)
    n = C.c_int32()
    lib().mgfo_manifold_from_contacts(lcs.ctypes.data, len(lcs), out.ctypes.data, C.byref(n), pairs.ctypes.data, cap)
    return dict(time=out[0], normal=out[1:4].copy(), t0=out[4:7].copy(), t1=out[7:10].copy(), pairs=pairs[:min(n.value, cap)].copy(), n=n.value)


def component(tag, p, d, r):
    return Component(tag, vec3(p), vec3(d), float(r))


def local_contacts_pair(ca, da, cb, db, cap=8):
    out = (LocalContact * cap)()
    n = lib().mgfo_local_contacts_pair(C.byref(ca), C.byref(vec3(da)), C.byref(cb), C.byref(vec3(db)), out, cap)
    return [dict(local_a=out[i].local_a.tup(), local_b=out[i].local_b.tup(), a=out[i].glob.a.tup(),
                 b=out[i].glob.b.tup(), n=out[i].glob.n.tup(), t=out[i].glob.t) for i in range(min(n, cap))]


class Compound:
    """mgf::Compound (compound.rs:230-352): components = list of component(tag, p, d, r)."""

    def __init__(self, comps):
        arr = (Component * len(comps))(*comps)
        self.h = lib().mgfo_compound_new(arr, len(comps))

    def __del__(self):
        if getattr(self, "h", None):
            lib().mgfo_compound_free(self.h)
            self.h = None

    def set_pose(self, disp, rot):
        """rot = (s, x, y, z)"""
        q = Quat(*[float(v) for v in rot])
        lib().mgfo_compound_set_pose(self.h, C.byref(vec3(disp)), C.byref(q))

    def bounds(self):
        b = Aabb()
        lib().mgfo_compound_bounds(self.h, C.byref(b))
        return b.c.tup(), b.r.tup()

    def contacts(self, sh, vel, cap=16):
        """compound.contacts(&Moving::sweep(shape, vel)) -> contacts in callback order"""
        out = (Contact * cap)()
        n = lib().mgfo_compound_contacts(self.h, C.byref(sh), C.byref(vec3(vel)), out, cap)
        if n < 0:
            raise ValueError("unsupported shape")
        return [dict(a=out[i].a.tup(), b=out[i].b.tup(), n=out[i].n.tup(), t=out[i].t) for i in range(min(n, cap))]

    def intersection(self, p, d, dt=float("inf")):
        ip, t = Vec3(), C.c_float()
        hit = lib().mgfo_compound_intersection(self.h, C.byref(vec3(p)), C.byref(vec3(d)), C.c_float(dt), C.byref(ip), C.byref(t))
        return (ip.tup(), t.value) if hit else None


class Bvh:
    """BVH<AABB, usize> of the reference (bvh.rs) — oracle copy."""

    def __init__(self, handle=None):
        self._own = handle is None
        self.h = lib().mgfo_bvh_new() if handle is None else handle

    def __del__(self):
        if getattr(self, "_own", False) and self.h:
            lib().mgfo_bvh_free(self.h)
            self.h = None

    @staticmethod
    def _aabb(c, r):
        return Aabb(vec3(c), vec3(r))

    def insert(self, c, r, val):
        return lib().mgfo_bvh_insert(self.h, C.byref(self._aabb(c, r)), val)

    def remove(self, node):
        return lib().mgfo_bvh_remove(self.h, node)

    def root(self):
        return lib().mgfo_bvh_root(self.h)

    def query(self, c, r, cap=4096):
        out = (C.c_uint64 * cap)()
        n = lib().mgfo_bvh_query(self.h, C.byref(self._aabb(c, r)), out, cap)
        return [out[i] for i in range(min(n, cap))]

    def raytrace(self, p, d, dt=float("inf"), cap=4096):
        """BVH::raytrace: [(value, point, t)] in the reference's visiting order."""
        vals = (C.c_uint64 * cap)()
        ips = (Vec3 * cap)()
        ts = (C.c_float * cap)()
        n = lib().mgfo_bvh_raytrace(self.h, C.byref(vec3(p)), C.byref(vec3(d)), C.c_float(dt), vals, ips, ts, cap)
        return [(vals[i], ips[i].tup(), ts[i]) for i in range(min(n, cap))]

    def serde(self):
        """The tree as the Python value serde_json would produce for BVH<AABB, usize> (bvh.rs:29-47, pool.rs:25-41)."""
        nodes, bounds = self.dump()
        n = len(nodes)
        state = np.zeros(max(n, 1), np.int32)
        nxt = np.zeros(max(n, 1), np.int64)
        fl, ln = C.c_int64(), C.c_int64()
        lib().mgfo_bvh_pool(self.h, state.ctypes.data, nxt.ctypes.data, n, C.byref(fl), C.byref(ln))
        entries = []
        v3d = lambda a: dict(x=float(a[0]), y=float(a[1]), z=float(a[2]))  # noqa: E731
        for i in range(n):
            if state[i] == 0:
                entries.append("FreeListEnd")
            elif state[i] == 1:
                entries.append({"FreeListPtr": {"next_free": int(nxt[i])}})
            else:
                _, h, parent, leaf, a, b = (int(v) for v in nodes[i])
                nt = {"Leaf": a} if leaf else {"Parent": [a, b]}
                entries.append({"Occupied": {"height": h, "parent": parent, "bounds": {"c": v3d(bounds[i][:3]), "r": v3d(bounds[i][3:])}, "node_type": nt}})
        root = int(lib().mgfo_bvh_root(self.h)) if ln.value else 0
        return {"root": root, "pool": {"len": int(ln.value), "free_list": None if fl.value < 0 else int(fl.value), "entries": entries}}

    def dump(self):
        n = lib().mgfo_bvh_dump(self.h, None, None, 0)
        nodes = np.zeros((max(n, 1), 6), dtype=np.int64)
        bounds = np.zeros((max(n, 1), 6), dtype=np.float32)
        lib().mgfo_bvh_dump(self.h, nodes.ctypes.data, bounds.ctypes.data, n)
        return nodes[:n], bounds[:n]


GHOST_FLOATS = 72  # World::kGhostFloats


class World:
    """World::step harness of mgf_demo/world.rs — oracle copy."""

    def __init__(self, order=ORDER_CANONICAL):
        self.h = lib().mgfo_world_new(order)
        self.stats = Stats()

    def __del__(self):
        if getattr(self, "h", None) and callable(lib):  # (at interpreter exit the module's globals may be gone already)
            lib().mgfo_world_free(self.h)
            self.h = None

    def set_terrain(self, verts, faces, pos):
        verts = np.ascontiguousarray(verts, dtype=np.float32).reshape(-1, 3)
        faces = np.ascontiguousarray(faces, dtype=np.uint32).reshape(-1, 3)
        lib().mgfo_world_set_terrain(self.h, verts.ctypes.data, len(verts), faces.ctypes.data, len(faces),
                                     C.byref(vec3(pos)))

    def add_obstacle(self, comps, disp=(0.0, 0.0, 0.0), rot=(1.0, 0.0, 0.0, 0.0)):
        """A static Compound (compound.rs:230-352) as an obstacle of the world beside the Mesh (World::obstacles)."""
        comps = np.ascontiguousarray(comps, dtype=COMPONENT_DTYPE)
        lib().mgfo_world_add_obstacle(self.h, comps.ctypes.data, len(comps), C.byref(vec3(disp)), C.byref(Quat(*[float(v) for v in rot])))

    def add_bodies(self, comps, mass, rest, fric, force):
        comps = np.ascontiguousarray(comps, dtype=COMPONENT_DTYPE)
        n = len(comps)
        mass = np.ascontiguousarray(np.broadcast_to(np.asarray(mass, np.float32), (n,)))
        rest = np.ascontiguousarray(np.broadcast_to(np.asarray(rest, np.float32), (n,)))
        fric = np.ascontiguousarray(np.broadcast_to(np.asarray(fric, np.float32), (n,)))
        force = np.ascontiguousarray(np.broadcast_to(np.asarray(force, np.float32), (n, 3)))
        r = lib().mgfo_world_add_bodies(self.h, comps.ctypes.data, n, mass.ctypes.data, rest.ctypes.data,
                                        fric.ctypes.data, force.ctypes.data)
        if r < 0:
            raise ValueError("singular inertia tensor (reference panics, physics.rs:212)")
        return r

    def add_compound_bodies(self, comps, comp_mass, offsets, rest, fric, force):
        """Bodies of several components (not in the reference; see RigidBodyVec::add_compound_body): body b is made of
        comps[offsets[b]:offsets[b + 1]] with masses comp_mass[...]."""
        comps = np.ascontiguousarray(comps, dtype=COMPONENT_DTYPE)
        offsets = np.ascontiguousarray(offsets, np.int64)
        n = len(offsets) - 1
        comp_mass = np.ascontiguousarray(np.broadcast_to(np.asarray(comp_mass, np.float32), (len(comps),)))
        rest = np.ascontiguousarray(np.broadcast_to(np.asarray(rest, np.float32), (n,)))
        fric = np.ascontiguousarray(np.broadcast_to(np.asarray(fric, np.float32), (n,)))
        force = np.ascontiguousarray(np.broadcast_to(np.asarray(force, np.float32), (n, 3)))
        r = lib().mgfo_world_add_compound_bodies(self.h, comps.ctypes.data, comp_mass.ctypes.data, offsets.ctypes.data, n,
                                                 rest.ctypes.data, fric.ctypes.data, force.ctypes.data)
        if r < 0:
            raise ValueError("empty body or singular inertia tensor")
        return r

    def body_info(self, i):
        out = np.zeros(10, np.float32)
        lib().mgfo_world_body_info(self.h, int(i), out.ctypes.data)
        return dict(inv_mass=float(out[0]), inv_moment=out[1:].copy())

    def __len__(self):
        return lib().mgfo_world_len(self.h)

    def step(self, dt, iters):
        lib().mgfo_world_step(self.h, dt, iters, C.byref(self.stats))
        return self.stats

    def build_constraints(self, dt):
        lib().mgfo_world_build_constraints(self.h, dt, C.byref(self.stats))
        return self.stats

    def solve(self, iters):
        lib().mgfo_world_solve(self.h, iters)

    # ---- tiling counterpart (numpy buffers) ----
    def begin_tick(self, dt):
        lib().mgfo_world_begin_tick(self.h, dt)

    def collide(self, dt):
        lib().mgfo_world_collide(self.h, dt, C.byref(self.stats))
        return self.stats

    def select_boundary(self, x_left, x_right):
        n = len(self)
        l = np.zeros(max(n, 1), np.uint32)
        r = np.zeros(max(n, 1), np.uint32)
        nl, nr = C.c_int64(), C.c_int64()
        lib().mgfo_world_select_boundary(self.h, x_left, x_right, l.ctypes.data, r.ctypes.data, n, C.byref(nl), C.byref(nr))
        return l[:nl.value].copy(), r[:nr.value].copy()

    def export_bodies(self, ids):
        ids = np.ascontiguousarray(ids, np.uint32)
        out = np.zeros((len(ids), GHOST_FLOATS), np.float32)
        lib().mgfo_world_export_bodies(self.h, ids.ctypes.data, len(ids), out.ctypes.data)
        return out

    def import_ghosts(self, recs):
        recs = np.ascontiguousarray(recs, np.float32).reshape(-1, GHOST_FLOATS)
        lib().mgfo_world_import_ghosts(self.h, recs.ctypes.data, len(recs))

    # ---- migration of owned bodies between tiles ----
    MIGRANT_FLOATS = 148

    def select_migrants(self, x_lo, x_hi):
        n = len(self)
        l = np.zeros(max(n, 1), np.uint32)
        r = np.zeros(max(n, 1), np.uint32)
        nl, nr = C.c_int64(), C.c_int64()
        lib().mgfo_world_select_migrants(self.h, x_lo, x_hi, l.ctypes.data, r.ctypes.data, n, C.byref(nl), C.byref(nr))
        return l[:nl.value].copy(), r[:nr.value].copy()

    def export_migrants(self, ids):
        ids = np.ascontiguousarray(ids, np.uint32)
        out = np.zeros((len(ids), self.MIGRANT_FLOATS), np.float32)
        lib().mgfo_world_export_migrants(self.h, ids.ctypes.data, len(ids), out.ctypes.data)
        return out

    def remove_bodies(self, ids):
        ids = np.ascontiguousarray(ids, np.uint32)
        lib().mgfo_world_remove_bodies(self.h, ids.ctypes.data, len(ids))

    def import_migrants(self, recs):
        recs = np.ascontiguousarray(recs, np.float32).reshape(-1, self.MIGRANT_FLOATS)
        lib().mgfo_world_import_migrants(self.h, recs.ctypes.data, len(recs))

    def set_tags(self, tags):
        tags = np.ascontiguousarray(tags, np.uint32)
        lib().mgfo_world_set_tags(self.h, tags.ctypes.data, len(tags))

    def tags(self):
        out = np.zeros(max(len(self), 1), np.uint32)
        lib().mgfo_world_read_tags(self.h, out.ctypes.data, len(out))
        return out[:len(self)].copy()

    def export_velocities(self, ids):
        ids = np.ascontiguousarray(ids, np.uint32)
        out = np.zeros((len(ids), 8), np.float32)
        lib().mgfo_world_export_velocities(self.h, ids.ctypes.data, len(ids), out.ctypes.data)
        return out

    def import_ghost_velocities(self, vel):
        vel = np.ascontiguousarray(vel, np.float32).reshape(-1, 8)
        lib().mgfo_world_import_ghost_velocities(self.h, vel.ctypes.data, len(vel))

    def constraint_depth(self, iters=1):
        return lib().mgfo_world_constraint_depth(self.h, iters)

    def constraints(self):
        n = lib().mgfo_world_get_constraints(self.h, None, 0)
        out = np.zeros(max(n, 1), dtype=CONSTRAINT_DTYPE)
        lib().mgfo_world_get_constraints(self.h, out.ctypes.data, n)
        return out[:n]

    def state(self):
        n = len(self)
        x = np.zeros((n, 3), np.float32)
        q = np.zeros((n, 4), np.float32)
        v = np.zeros((n, 3), np.float32)
        w = np.zeros((n, 3), np.float32)
        d = np.zeros((n, 3), np.float32)
        lib().mgfo_world_get_state(self.h, x.ctypes.data, q.ctypes.data, v.ctypes.data, w.ctypes.data, d.ctypes.data)
        return dict(x=x, q=q, v=v, omega=w, delta=d)

    def set_state(self, x=None, q=None, v=None, omega=None, delta=None):
        def p(a, k):
            if a is None:
                return None
            a = np.ascontiguousarray(a, dtype=np.float32).reshape(-1, k)
            keep.append(a)
            return a.ctypes.data
        keep = []
        lib().mgfo_world_set_state(self.h, p(x, 3), p(q, 4), p(v, 3), p(omega, 3), p(delta, 3))

    def colliders(self):
        n = len(self)
        comps = np.zeros(n, dtype=COMPONENT_DTYPE)
        d = np.zeros((n, 3), np.float32)
        lib().mgfo_world_get_colliders(self.h, comps.ctypes.data, d.ctypes.data)
        return comps, d

    def inv_moment(self):
        n = len(self)
        b = np.zeros((n, 9), np.float32)
        w = np.zeros((n, 9), np.float32)
        lib().mgfo_world_get_inv_moment(self.h, b.ctypes.data, w.ctypes.data)
        return b, w

    def set_velocity(self, i, lin, ang):
        lib().mgfo_world_set_velocity(self.h, i, C.byref(vec3(lin)), C.byref(vec3(ang)))

    # ---- the RigidBodyVec / ConstrainedSet / ContactConstraint / Solver surface on its own ----
    def integrate(self, dt):
        """RigidBodyVec::integrate physics.rs:222-253 alone"""
        lib().mgfo_world_integrate(self.h, dt)

    def complete_motion(self):
        """RigidBodyVec::complete_motion physics.rs:262-269 alone"""
        lib().mgfo_world_complete_motion(self.h)

    def get(self, index=None, static=None):
        """ConstrainedSet::get physics.rs:273-304 -> dict(linear, angular, x, restitution, friction, inv_mass, inv_moment[9])"""
        out = np.zeros(21, np.float32)
        if static is None:
            lib().mgfo_world_get(self.h, 0, int(index), C.byref(vec3((0, 0, 0))), 0.0, out.ctypes.data)
        else:
            lib().mgfo_world_get(self.h, 1, 0, C.byref(vec3(static[0])), float(static[1]), out.ctypes.data)
        return dict(linear=out[0:3].copy(), angular=out[3:6].copy(), x=out[6:9].copy(), restitution=out[9], friction=out[10],
                    inv_mass=out[11], inv_moment=out[12:21].copy())

    def constraint_new(self, a, b, normal, tangents, local_a, local_b, dt, static_b=None):
        """ContactConstraint::new solver.rs:101-191 for one caller-built manifold -> m flattened rows (CONSTRAINT_DTYPE).
        b = None with static_b = (center, friction) for RigidBodyRef::Static."""
        la = np.ascontiguousarray(local_a, np.float32).reshape(-1, 3)
        lb = np.ascontiguousarray(local_b, np.float32).reshape(-1, 3)
        tg = np.ascontiguousarray(tangents, np.float32).reshape(2, 3)
        out = np.zeros(len(la), CONSTRAINT_DTYPE)
        cb, fb = (static_b if static_b is not None else ((0, 0, 0), 0.0))
        lib().mgfo_constraint_new(self.h, int(a), -1 if b is None else int(b), C.byref(vec3(cb)), float(fb), C.byref(vec3(normal)),
                                  tg.ctypes.data, len(la), la.ctypes.data, lb.ctypes.data, dt, out.ctypes.data)
        return out

    def solver_solve(self, rows, iters):
        """Solver::new, add_constraint for every row in order, solve(iters) solver.rs:59-78 on this world's bodies;
        returns the rows with their normal_impulse after the call."""
        rows = np.ascontiguousarray(rows, CONSTRAINT_DTYPE).copy()
        lib().mgfo_solver_solve(self.h, rows.ctypes.data, len(rows), int(iters))
        return rows

    def terrain_contacts(self, i, cap=16):
        out = (LocalContact * cap)()
        n = lib().mgfo_world_terrain_contacts(self.h, i, out, cap)
        return [dict(local_a=out[k].local_a.tup(), local_b=out[k].local_b.tup(), a=out[k].glob.a.tup(),
                     b=out[k].glob.b.tup(), n=out[k].glob.n.tup(), t=out[k].glob.t) for k in range(min(n, cap))]

    def terrain_bvh(self):
        return Bvh(lib().mgfo_world_terrain_bvh(self.h))

    def world_bvh(self):
        return Bvh(lib().mgfo_world_bvh(self.h))
