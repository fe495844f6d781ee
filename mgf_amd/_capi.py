"""ctypes binding of include/mgf_hip.h.  Names follow the reference (mgf) API they replace."""
import ctypes as C
import os
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


def lib_path():
    # (MGF_AMD_LIB: a differently built libmgf_hip.so for an A/B experiment; never a different implementation)
    return os.environ.get("MGF_AMD_LIB") or os.path.join(_HERE, "libmgf_hip.so")


class MgfError(RuntimeError):
    """A non-OK mgf_status; `.status` holds the code (the reference panics in these cases)."""

    def __init__(self, status, msg):
        super().__init__(f"mgf status {status} ({_STATUS_NAMES.get(status, '?')}): {msg}")
        self.status = status


_STATUS_NAMES = {0: "OK", 1: "EMPTY", 2: "NOT_OCCUPIED", 3: "NOT_LEAF", 4: "STATIC_REF", 5: "SINGULAR", 6: "INVALID",
                 7: "CAPACITY", 8: "HIP", 9: "OOM"}
OK, ERR_EMPTY, ERR_NOT_OCCUPIED, ERR_NOT_LEAF, ERR_STATIC_REF, ERR_SINGULAR, ERR_INVALID, ERR_CAPACITY, ERR_HIP, ERR_OOM = range(10)
SPHERE, CAPSULE, TRIANGLE, RECTANGLE, PLANE, RAY, SEGMENT, AABB = 0, 1, 2, 3, 4, 5, 6, 7


class Vec3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]

    def tup(self):
        return (self.x, self.y, self.z)


class Quat(C.Structure):
    _fields_ = [("s", C.c_float), ("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]


class Aabb(C.Structure):
    _fields_ = [("c", Vec3), ("r", Vec3)]


class Component(C.Structure):
    _fields_ = [("tag", C.c_int32), ("p", Vec3), ("d", Vec3), ("r", C.c_float)]


class MovingComponent(C.Structure):
    _fields_ = [("shape", Component), ("delta", Vec3)]


class Shape(C.Structure):
    _fields_ = [("kind", C.c_int32), ("v", C.c_float * 12)]


class Contact(C.Structure):
    _fields_ = [("a", Vec3), ("b", Vec3), ("n", Vec3), ("t", C.c_float)]


class LocalContact(C.Structure):
    _fields_ = [("local_a", Vec3), ("local_b", Vec3), ("glob", Contact)]


class BodyRef(C.Structure):
    _fields_ = [("tag", C.c_int32), ("index", C.c_uint32), ("center", Vec3), ("friction", C.c_float)]


class Velocity(C.Structure):
    _fields_ = [("linear", Vec3), ("angular", Vec3)]


class RigidBodyInfo(C.Structure):
    _fields_ = [("x", Vec3), ("restitution", C.c_float), ("friction", C.c_float), ("inv_mass", C.c_float),
                ("inv_moment", C.c_float * 9)]


class Params(C.Structure):
    _fields_ = [("baumgarte", C.c_float), ("penetration_slop", C.c_float), ("persistent_threshold_sq", C.c_float),
                ("collision_epsilon", C.c_float), ("fat_margin", C.c_float)]


class StepStats(C.Structure):
    _fields_ = [("n_bodies", C.c_uint64), ("n_constraints", C.c_uint64), ("n_terrain_constraints", C.c_uint64),
                ("n_pair_candidates", C.c_uint64), ("n_terrain_candidates", C.c_uint64), ("n_refits", C.c_uint64),
                ("n_levels", C.c_uint32), ("iters", C.c_uint32),
                ("ms_integrate", C.c_float), ("ms_broadphase", C.c_float), ("ms_narrowphase", C.c_float),
                ("ms_setup", C.c_float), ("ms_solve", C.c_float), ("ms_total", C.c_float),
                ("solver_kernel_launches", C.c_uint64), ("ms_solver_kernels", C.c_float), ("n_ghost_constraints", C.c_uint64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}

    def __getitem__(self, key):  # a read-only mapping view: callers in a hot loop need not build the dict
        return getattr(self, key)


COMPONENT_DTYPE = np.dtype([("tag", "<i4"), ("p", "<f4", 3), ("d", "<f4", 3), ("r", "<f4")])
MOVING_DTYPE = np.dtype([("tag", "<i4"), ("p", "<f4", 3), ("d", "<f4", 3), ("r", "<f4"), ("delta", "<f4", 3)])
CONSTRAINT_DTYPE = np.dtype([("a", "<i4"), ("b", "<i4"), ("n_contacts", "<i4"),
                             ("normal", "<f4", 3), ("t0", "<f4", 3), ("t1", "<f4", 3), ("ra", "<f4", 3), ("rb", "<f4", 3),
                             ("bias", "<f4"), ("normal_mass", "<f4"), ("tangent_mass0", "<f4"), ("tangent_mass1", "<f4"),
                             ("normal_impulse", "<f4"), ("friction", "<f4")])
assert COMPONENT_DTYPE.itemsize == C.sizeof(Component) == 32
assert MOVING_DTYPE.itemsize == C.sizeof(MovingComponent) == 44
assert CONSTRAINT_DTYPE.itemsize == 96

# every symbol include/mgf_hip.h declares (tests check the library exports all of them)
SYMBOLS = [
    "mgf_ctx_create", "mgf_ctx_destroy", "mgf_ctx_set_stream", "mgf_last_error", "mgf_default_params", "mgf_version", "mgf_exclusive_scan_u32",
    "mgf_contacts", "mgf_contacts_batch", "mgf_tri_reject_batch", "mgf_local_contacts_pair", "mgf_ray_capsule", "mgf_inertia_tensor",
    "mgf_mesh_new", "mgf_mesh_free", "mgf_mesh_push_vert", "mgf_mesh_push_face", "mgf_mesh_set_pos", "mgf_mesh_build",
    "mgf_local_contacts_mesh",
    "mgf_bvh_new", "mgf_bvh_with_capacity", "mgf_bvh_free", "mgf_bvh_empty", "mgf_bvh_clear", "mgf_bvh_insert",
    "mgf_bvh_remove", "mgf_bvh_root", "mgf_bvh_get_leaf", "mgf_bvh_bounds", "mgf_bvh_query", "mgf_bvh_query_many",
    "mgf_bvh_raytrace", "mgf_bvh_raytrace_many", "mgf_intersections_batch",
    "mgf_compound_new", "mgf_compound_free", "mgf_compound_set_pose", "mgf_compound_bounds", "mgf_compound_contacts_many",
    "mgf_compound_intersections",
    "mgf_bvh_to_json", "mgf_bvh_from_json", "mgf_mesh_to_json", "mgf_mesh_from_json", "mgf_manifolds_from_contacts",
    "mgf_world_new", "mgf_world_free", "mgf_world_set_terrain", "mgf_world_add_bodies", "mgf_world_add_compound_bodies", "mgf_world_len",
    "mgf_world_step", "mgf_world_step_many", "mgf_world_build_constraints", "mgf_world_solve", "mgf_world_complete_motion",
    "mgf_world_integrate", "mgf_world_get", "mgf_world_set", "mgf_world_read_state", "mgf_world_write_state",
    "mgf_world_read_colliders", "mgf_world_read_constraints", "mgf_world_set_constraints", "mgf_world_set_option",
    "mgf_world_device_ptr",
    "mgf_world_release_device_ptrs",
    "mgf_world_begin_tick", "mgf_world_collide", "mgf_world_select_boundary", "mgf_world_export_bodies",
    "mgf_world_import_ghosts", "mgf_world_export_velocities", "mgf_world_import_ghost_velocities", "mgf_world_ghost_len",
    "mgf_world_select_tile", "mgf_world_export_migrants", "mgf_world_remove_bodies", "mgf_world_import_migrants",
    "mgf_world_set_tags", "mgf_world_read_tags",
    "mgf_world_solve_enqueue", "mgf_world_finish", "mgf_world_counter",
    "mgf_constraints_new", "mgf_solver_new", "mgf_solver_free", "mgf_solver_add_constraint", "mgf_solver_add_constraints",
    "mgf_solver_len", "mgf_solver_clear", "mgf_solver_read_constraints", "mgf_solver_solve", "mgf_world_clone",
    "mgf_geom_to_json", "mgf_geom_from_json",
    "mgf_tiles_create", "mgf_tiles_free", "mgf_rccl_unique_id", "mgf_rccl_allow_override", "mgf_tiles_connect", "mgf_tiles_preflight", "mgf_tiles_step",
    "mgf_tiles_migrated", "mgf_tiles_set_option", "mgf_world_add_obstacle", "mgf_tiles_counter",
]

_lib = None
HIT_FN = C.CFUNCTYPE(None, C.POINTER(C.c_uint64), C.c_void_p)
RAY_FN = C.CFUNCTYPE(None, C.POINTER(C.c_uint64), C.c_void_p, C.c_void_p)


def load_library():
    """dlopen mgf_amd/libmgf_hip.so.  Raises (never falls back) when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise MgfError(ERR_HIP, f"{path} is missing: build it with `python -m mgf_amd.build` (hipcc, gfx950). "
                                "mgf_amd has no CPU fallback.")
    L = C.CDLL(path)
    P, vp, i32, i64, u64, f32 = C.POINTER, C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_float
    sig = {
        "mgf_ctx_create": (i32, [C.c_int, P(vp)]),
        "mgf_ctx_destroy": (None, [vp]),
        "mgf_ctx_set_stream": (i32, [vp, vp]),
        "mgf_last_error": (C.c_char_p, []),
        "mgf_default_params": (Params, []),
        "mgf_version": (C.c_char_p, []),
        "mgf_exclusive_scan_u32": (i32, [vp, vp, i64, vp]),
        "mgf_contacts": (i32, [vp, P(Shape), P(Vec3), P(Shape), P(Vec3), P(Contact), i32, P(i32)]),
        "mgf_contacts_batch": (i32, [vp, i64, vp, vp, vp, vp, vp, vp, vp]),
        "mgf_tri_reject_batch": (i32, [vp, i64, vp, vp, vp, vp]),
        "mgf_local_contacts_pair": (i32, [vp, P(MovingComponent), P(MovingComponent), P(LocalContact), i32, P(i32)]),
        "mgf_ray_capsule": (i32, [vp, P(Vec3), P(Vec3), P(Shape), P(Vec3), P(f32), P(i32)]),
        "mgf_inertia_tensor": (i32, [P(Component), f32, P(f32)]),
        "mgf_mesh_new": (i32, [vp, P(vp)]),
        "mgf_mesh_free": (None, [vp]),
        "mgf_mesh_push_vert": (i32, [vp, Vec3, P(u64)]),
        "mgf_mesh_push_face": (i32, [vp, u64, u64, u64, P(u64)]),
        "mgf_mesh_set_pos": (i32, [vp, Vec3]),
        "mgf_mesh_build": (i32, [vp, vp, i64, vp, i64]),
        "mgf_mesh_bvh_view": (vp, [vp]),
        "mgf_local_contacts_mesh": (i32, [vp, P(MovingComponent), vp, P(LocalContact), i32, P(i32)]),
        "mgf_bvh_new": (i32, [vp, P(vp)]),
        "mgf_bvh_with_capacity": (i32, [vp, u64, P(vp)]),
        "mgf_bvh_free": (None, [vp]),
        "mgf_bvh_empty": (i32, [vp]),
        "mgf_bvh_clear": (i32, [vp]),
        "mgf_bvh_insert": (i32, [vp, P(Aabb), u64, P(u64)]),
        "mgf_bvh_remove": (i32, [vp, u64]),
        "mgf_bvh_root": (i32, [vp, P(u64)]),
        "mgf_bvh_get_leaf": (i32, [vp, u64, P(u64)]),
        "mgf_bvh_bounds": (i32, [vp, u64, P(Aabb)]),
        "mgf_bvh_query": (i32, [vp, P(Aabb), HIT_FN, vp]),
        "mgf_bvh_query_many": (i32, [vp, vp, i64, vp, vp, i64, P(i64)]),
        "mgf_bvh_raytrace": (i32, [vp, vp, RAY_FN, vp]),
        "mgf_bvh_raytrace_many": (i32, [vp, vp, i64, vp, vp, vp, i64, P(i64)]),
        "mgf_intersections_batch": (i32, [vp, i64, vp, vp, vp, vp, vp]),
        "mgf_compound_new": (i32, [vp, vp, i64, P(vp)]),
        "mgf_compound_free": (None, [vp]),
        "mgf_compound_set_pose": (i32, [vp, Vec3, Quat]),
        "mgf_compound_bounds": (i32, [vp, P(Aabb)]),
        "mgf_compound_contacts_many": (i32, [vp, vp, i64, vp, vp, i64, P(i64)]),
        "mgf_compound_intersections": (i32, [vp, vp, i64, vp, vp]),
        "mgf_bvh_to_json": (i32, [vp, vp, i64, P(i64)]),
        "mgf_bvh_from_json": (i32, [vp, C.c_char_p, i64, P(vp)]),
        "mgf_mesh_to_json": (i32, [vp, vp, i64, P(i64)]),
        "mgf_mesh_from_json": (i32, [vp, C.c_char_p, i64, P(vp)]),
        "mgf_manifolds_from_contacts": (i32, [vp, vp, i64, vp, vp, vp]),
        "mgf_bvh_dump": (i64, [vp, vp, vp, i64]),
        "mgf_world_new": (i32, [vp, P(Params), P(vp)]),
        "mgf_world_free": (None, [vp]),
        "mgf_world_set_terrain": (i32, [vp, vp]),
        "mgf_world_add_bodies": (i32, [vp, vp, i64, vp, vp, vp, vp, P(u64)]),
        "mgf_world_add_compound_bodies": (i32, [vp, vp, vp, vp, i64, vp, vp, vp, P(u64)]),
        "mgf_world_len": (i64, [vp]),
        "mgf_world_step": (i32, [vp, f32, i32, P(StepStats)]),
        "mgf_world_step_many": (i32, [vp, f32, i32, i64, vp]),
        "mgf_world_build_constraints": (i32, [vp, f32, P(StepStats)]),
        "mgf_world_solve": (i32, [vp, i32, P(StepStats)]),
        "mgf_world_complete_motion": (i32, [vp]),
        "mgf_world_integrate": (i32, [vp, f32]),
        "mgf_world_get": (i32, [vp, P(BodyRef), P(Velocity), P(RigidBodyInfo)]),
        "mgf_world_set": (i32, [vp, P(BodyRef), P(Velocity)]),
        "mgf_world_read_state": (i32, [vp, vp, vp, vp, vp, vp, i64]),
        "mgf_world_write_state": (i32, [vp, vp, vp, vp, vp, vp, i64]),
        "mgf_world_read_colliders": (i32, [vp, vp, i64]),
        "mgf_world_read_constraints": (i32, [vp, vp, i64, P(i64)]),
        "mgf_world_set_constraints": (i32, [vp, vp, i64]),
        "mgf_world_set_option": (i32, [vp, C.c_char_p, i64]),
        "mgf_world_device_ptr": (i32, [vp, C.c_char_p, P(vp), P(i64)]),
        "mgf_world_release_device_ptrs": (i32, [vp]),
        "mgf_world_begin_tick": (i32, [vp, f32]),
        "mgf_world_collide": (i32, [vp, f32, P(StepStats)]),
        "mgf_world_select_boundary": (i32, [vp, f32, f32, vp, vp, i64, P(i64), P(i64)]),
        "mgf_world_export_bodies": (i32, [vp, vp, i64, vp]),
        "mgf_world_import_ghosts": (i32, [vp, vp, i64]),
        "mgf_world_export_velocities": (i32, [vp, vp, i64, vp]),
        "mgf_world_import_ghost_velocities": (i32, [vp, vp, i64]),
        "mgf_world_ghost_len": (i64, [vp]),
        "mgf_world_select_tile": (i32, [vp, f32, f32, f32, f32, vp, vp, vp, i64, P(i64)]),
        "mgf_world_export_migrants": (i32, [vp, vp, i64, vp]),
        "mgf_world_remove_bodies": (i32, [vp, vp, i64]),
        "mgf_world_import_migrants": (i32, [vp, vp, i64]),
        "mgf_world_set_tags": (i32, [vp, vp, i64]),
        "mgf_world_read_tags": (i32, [vp, vp, i64]),
        "mgf_world_solve_enqueue": (i32, [vp, i32]),
        "mgf_world_finish": (i32, [vp, P(StepStats)]),
        "mgf_world_counter": (i32, [vp, C.c_char_p, P(i64)]),
        "mgf_constraints_new": (i32, [vp, vp, vp, vp, i64, f32, vp, i64, P(i64)]),
        "mgf_solver_new": (i32, [P(vp)]),
        "mgf_solver_free": (None, [vp]),
        "mgf_solver_add_constraint": (i32, [vp, vp]),
        "mgf_solver_add_constraints": (i32, [vp, vp, i64]),
        "mgf_solver_len": (i64, [vp]),
        "mgf_solver_clear": (i32, [vp]),
        "mgf_solver_read_constraints": (i32, [vp, vp, i64, P(i64)]),
        "mgf_solver_solve": (i32, [vp, vp, i32, P(StepStats)]),
        "mgf_world_clone": (i32, [vp, P(vp)]),
        "mgf_geom_to_json": (i32, [P(Shape), P(Vec3), vp, i64, P(i64)]),
        "mgf_geom_from_json": (i32, [i32, C.c_char_p, i64, P(Shape), P(Vec3)]),
        "mgf_tiles_create": (i32, [vp, i32, vp, vp, vp, i32, i32, f32, i32, i32, P(vp)]),
        "mgf_tiles_free": (None, [vp]),
        "mgf_rccl_unique_id": (i32, [vp]),
        "mgf_rccl_allow_override": (i32, [C.c_int32]),
        "mgf_tiles_connect": (i32, [vp, vp, i32, i32]),
        "mgf_tiles_preflight": (i32, [vp, P(i32)]),
        "mgf_tiles_step": (i32, [vp, f32, i32, vp]),
        "mgf_tiles_migrated": (i64, [vp, i32, i32]),
        "mgf_tiles_counter": (i64, [vp, C.c_char_p]),
        "mgf_tiles_set_option": (i32, [vp, C.c_char_p, i64]),
        "mgf_world_add_obstacle": (i32, [vp, vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def _check(status):
    if status != 0:
        raise MgfError(status, load_library().mgf_last_error().decode(errors="replace"))


def _v3(v):
    return Vec3(float(v[0]), float(v[1]), float(v[2]))


def default_params():
    return load_library().mgf_default_params()


def inertia_tensor(tag, p, d, r, mass):
    """Inertia::tensor (physics.rs:26-93): column-major 3x3 as a flat list of 9."""
    out = (C.c_float * 9)()
    _check(load_library().mgf_inertia_tensor(C.byref(Component(tag, _v3(p), _v3(d), float(r))), float(mass), out))
    return [out[i] for i in range(9)]


class Context:
    """One per GPU (device + HIP stream).  Raises MgfError(HIP) when no GPU is present."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        self._children = weakref.WeakSet()
        _check(load_library().mgf_ctx_create(int(device), C.byref(self._h)))

    def _adopt(self, obj):
        self._children.add(obj)

    def exclusive_scan(self, counts):
        """out[i] = sum(counts[:i]) on the device (the tick's own scan primitive)."""
        a = np.ascontiguousarray(counts, np.uint32)
        out = np.zeros(len(a), np.uint32)
        _check(load_library().mgf_exclusive_scan_u32(self._h, a.ctypes.data, len(a), out.ctypes.data))
        return out

    def set_stream(self, hip_stream):
        """Enqueue this context's work on a caller-owned hipStream_t (an int handle, e.g. torch's cuda_stream)."""
        _check(load_library().mgf_ctx_set_stream(self._h, C.c_void_p(int(hip_stream))))

    def close(self):
        """Destroy the context; handles created from it are released first.  (A handle that outlives it all the same - an interpreter's
        finalisation clears the weak references before it runs the finalisers, in any order - keeps the C side's struct and streams alive until
        it is freed: mgf_ctx_destroy only drops the creator's reference.)"""
        if getattr(self, "_h", None):
            for child in list(self._children):
                child.__del__()
            load_library().mgf_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _shape(d):
    s = Shape()
    k = d["kind"]
    if k == "sphere":
        s.kind, vals = SPHERE, list(d["c"]) + [d["r"]]
    elif k == "capsule":
        s.kind, vals = CAPSULE, list(d["a"]) + list(d["d"]) + [d["r"]]
    elif k == "triangle":
        s.kind, vals = TRIANGLE, list(d["a"]) + list(d["b"]) + list(d["c"])
    elif k == "plane":
        s.kind, vals = PLANE, list(d["n"]) + [d["d"]]
    elif k == "rectangle":
        s.kind, vals = RECTANGLE, list(d["c"]) + list(d["u0"]) + list(d["u1"]) + list(d["e"])
    elif k == "ray":
        s.kind, vals = RAY, list(d["p"]) + list(d["d"])
    elif k == "segment":
        s.kind, vals = SEGMENT, list(d["a"]) + list(d["b"])
    elif k == "aabb":
        s.kind, vals = AABB, list(d["c"]) + list(d["r"])
    else:
        raise ValueError(k)
    for i, x in enumerate(vals):
        s.v[i] = float(x)
    return s


_GEOM_FIELDS = {"sphere": (SPHERE, [("c", 3), ("r", 1)]), "capsule": (CAPSULE, [("a", 3), ("d", 3), ("r", 1)]),
                "triangle": (TRIANGLE, [("a", 3), ("b", 3), ("c", 3)]), "plane": (PLANE, [("n", 3), ("d", 1)]),
                "rectangle": (RECTANGLE, [("c", 3), ("u0", 3), ("u1", 3), ("e", 2)]), "ray": (RAY, [("p", 3), ("d", 3)]),
                "segment": (SEGMENT, [("a", 3), ("b", 3)]), "aabb": (AABB, [("c", 3), ("r", 3)])}


def geom_to_json(shape, moving=None):
    """serde_json text of a geom.rs struct given as a shape dict (kind + fields); moving = the Vector3 of Moving<T>."""
    n = C.c_int64()
    sh = _shape(shape)
    mv = C.byref(_v3(moving)) if moving is not None else None
    st = load_library().mgf_geom_to_json(C.byref(sh), mv, None, 0, C.byref(n))
    if st != ERR_CAPACITY:
        _check(st)
    buf = C.create_string_buffer(n.value + 1)
    _check(load_library().mgf_geom_to_json(C.byref(sh), mv, buf, n.value + 1, C.byref(n)))
    return buf.value.decode()


def geom_from_json(kind, text, moving=False):
    """the shape dict of `kind` ("sphere", "capsule", ...) read from serde_json text; moving=True reads Moving<T> and returns
    (shape, velocity)."""
    code, fields = _GEOM_FIELDS[kind]
    raw = text.encode()
    out, vel = Shape(), Vec3()
    _check(load_library().mgf_geom_from_json(code, raw, len(raw), C.byref(out), C.byref(vel) if moving else None))
    d, at = dict(kind=kind), 0
    for name, w in fields:
        d[name] = out.v[at] if w == 1 else [out.v[at + i] for i in range(w)]
        at += w
    return (d, vel.tup()) if moving else d


def _contact_dict(c):
    return dict(a=c.a.tup(), b=c.b.tup(), n=c.n.tup(), t=c.t)


def contacts(ctx, a, vel_a, b, vel_b, cap=4):
    """Contacts::contacts (collision.rs:471-482) on the GPU for shape dicts a, b."""
    out = (Contact * cap)()
    n = C.c_int32()
    va = C.byref(_v3(vel_a)) if vel_a is not None else None
    vb = C.byref(_v3(vel_b)) if vel_b is not None else None
    _check(load_library().mgf_contacts(ctx._h, C.byref(_shape(a)), va, C.byref(_shape(b)), vb, out, cap, C.byref(n)))
    return [_contact_dict(out[i]) for i in range(n.value)]


def contacts_batch(ctx, problems):
    """problems: list of (a, vel_a, b, vel_b).  Returns a list of contact lists."""
    n = len(problems)
    A = (Shape * n)(*[_shape(p[0]) for p in problems])
    B = (Shape * n)(*[_shape(p[2]) for p in problems])
    va = np.zeros((n, 3), np.float32)
    vb = np.zeros((n, 3), np.float32)
    hv = np.zeros(n, np.uint8)
    for i, p in enumerate(problems):
        if p[1] is not None:
            va[i] = p[1]
            hv[i] |= 1
        if p[3] is not None:
            vb[i] = p[3]
            hv[i] |= 2
    out = (Contact * (2 * n))()
    counts = np.zeros(n, np.int32)
    _check(load_library().mgf_contacts_batch(ctx._h, n, A, va.ctypes.data, B, vb.ctypes.data, hv.ctypes.data, out,
                                             counts.ctypes.data))
    return [[_contact_dict(out[2 * i + k]) for k in range(counts[i])] for i in range(n)]


def tri_reject_batch(ctx, tag, p, d, r, delta, tris):
    """n (moving component, triangle) problems as arrays - tag (n,), p, d, delta (n, 3), r (n,), tris (n, 3, 3) - through the front end's cheap
    reject and through the body-triangle tests: returns (far (n,) uint8, counts (n,) int32)  (mgf_tri_reject_batch)."""
    n = len(tag)
    rec = np.zeros(n, dtype=np.dtype([("tag", np.int32), ("p", np.float32, 3), ("d", np.float32, 3), ("r", np.float32), ("delta", np.float32, 3)]))
    rec["tag"], rec["p"], rec["d"], rec["r"], rec["delta"] = tag, p, d, r, delta
    assert rec.dtype.itemsize == C.sizeof(MovingComponent)
    t = np.ascontiguousarray(tris, dtype=np.float32).reshape(n, 9)
    far = np.zeros(n, np.uint8)
    counts = np.zeros(n, np.int32)
    _check(load_library().mgf_tri_reject_batch(ctx._h, n, rec.ctypes.data, t.ctypes.data, far.ctypes.data, counts.ctypes.data))
    return far, counts


def _moving(tag, p, d, r, delta):
    return MovingComponent(Component(int(tag), _v3(p), _v3(d), float(r)), _v3(delta))


def _local_dict(lc):
    return dict(local_a=lc.local_a.tup(), local_b=lc.local_b.tup(), a=lc.glob.a.tup(), b=lc.glob.b.tup(),
                n=lc.glob.n.tup(), t=lc.glob.t)


def local_contacts_pair(ctx, a, b, cap=4):
    """LocalContacts for two Moving<Component>s given as (tag, p, d, r, delta) tuples (compound.rs:192-207)."""
    out = (LocalContact * cap)()
    n = C.c_int32()
    _check(load_library().mgf_local_contacts_pair(ctx._h, C.byref(_moving(*a)), C.byref(_moving(*b)), out, cap, C.byref(n)))
    return [_local_dict(out[i]) for i in range(n.value)]


def local_contacts_mesh(ctx, body, mesh, cap=32):
    out = (LocalContact * cap)()
    n = C.c_int32()
    _check(load_library().mgf_local_contacts_mesh(ctx._h, C.byref(_moving(*body)), mesh._h, out, cap, C.byref(n)))
    return [_local_dict(out[i]) for i in range(n.value)]


PARTICLE_DTYPE = np.dtype([("p", "<f4", 3), ("d", "<f4", 3), ("dt", "<f4")])
INTERSECTION_DTYPE = np.dtype([("p", "<f4", 3), ("t", "<f4")])


def particles(rays=(), segments=()):
    """Particles (geom.rs:802-855) from rays [(p, d)] and segments [(a, b)]: a Segment is p = a, d = b - a, DT = 1."""
    out = np.zeros(len(rays) + len(segments), PARTICLE_DTYPE)
    for i, (p, d) in enumerate(rays):
        out[i] = (p, d, np.inf)
    for i, (a, b) in enumerate(segments):
        a32, b32 = np.asarray(a, np.float32), np.asarray(b, np.float32)
        out[len(rays) + i] = (a32, b32 - a32, 1.0)
    return out


def intersections(ctx, parts, shapes=None, boxes=None):
    """Intersects<Shape> (shape dicts) or Intersects<AABB> (rows c3 r3) for each particle -> [(point, t) or None]."""
    parts = np.ascontiguousarray(parts, PARTICLE_DTYPE)
    n = len(parts)
    out = np.zeros(max(n, 1), INTERSECTION_DTYPE)
    hit = np.zeros(max(n, 1), np.int32)
    sp = bp = None
    if shapes is not None:
        arr = (Shape * max(n, 1))(*[_shape(x) for x in shapes])
        sp = C.cast(arr, C.c_void_p)
    else:
        bx = np.ascontiguousarray(boxes, np.float32).reshape(-1, 6)
        bp = bx.ctypes.data
    _check(load_library().mgf_intersections_batch(ctx._h, n, parts.ctypes.data, sp, bp, out.ctypes.data, hit.ctypes.data))
    return [(tuple(float(v) for v in out[i]["p"]), float(out[i]["t"])) if hit[i] else None for i in range(n)]


def ray_capsule(ctx, p, d, cap_a, cap_d, cap_r):
    """Intersects<Capsule> for Ray (collision.rs:275-359) -> (point, t) or None."""
    ip = Vec3()
    t = C.c_float()
    hit = C.c_int32()
    s = _shape(dict(kind="capsule", a=cap_a, d=cap_d, r=cap_r))
    _check(load_library().mgf_ray_capsule(ctx._h, C.byref(_v3(p)), C.byref(_v3(d)), C.byref(s), C.byref(ip), C.byref(t), C.byref(hit)))
    return (ip.tup(), t.value) if hit.value else None


LOCAL_CONTACT_DTYPE = np.dtype([("local_a", "<f4", 3), ("local_b", "<f4", 3), ("a", "<f4", 3), ("b", "<f4", 3), ("n", "<f4", 3), ("t", "<f4")])
MANIFOLD_CAP = 8
MANIFOLD_DTYPE = np.dtype([("time", "<f4"), ("normal", "<f4", 3), ("tangent", "<f4", (2, 3)), ("n_contacts", "<i4"),
                           ("local_a", "<f4", (MANIFOLD_CAP, 3)), ("local_b", "<f4", (MANIFOLD_CAP, 3))])


def manifolds_from_contacts(ctx, offsets, contacts):
    """ContactPruner::push for each group's LocalContacts in order, then Manifold::from(pruner) (manifold.rs:42-148)."""
    offsets = np.ascontiguousarray(offsets, np.uint64)
    contacts = np.ascontiguousarray(contacts, LOCAL_CONTACT_DTYPE)
    n = len(offsets) - 1
    out = np.zeros(max(n, 1), MANIFOLD_DTYPE)
    _check(load_library().mgf_manifolds_from_contacts(ctx._h, None, n, offsets.ctypes.data, contacts.ctypes.data, out.ctypes.data))
    return out[:n]


def _to_json(fn, handle):
    n = C.c_int64()
    st = fn(handle, None, 0, C.byref(n))
    if st not in (OK, ERR_CAPACITY):
        _check(st)
    buf = C.create_string_buffer(n.value + 1)
    _check(fn(handle, buf, n.value + 1, C.byref(n)))
    return buf.value.decode()


CONTACT_DTYPE = np.dtype([("a", "<f4", 3), ("b", "<f4", 3), ("n", "<f4", 3), ("t", "<f4")])


class Compound:
    """mgf::Compound (compound.rs:230-352): comps = COMPONENT_DTYPE array (tag, p, d, r)."""

    def __init__(self, ctx, comps):
        self._h = C.c_void_p()
        self._ctx = ctx
        comps = np.ascontiguousarray(comps, COMPONENT_DTYPE)
        _check(load_library().mgf_compound_new(ctx._h if ctx is not None else None, comps.ctypes.data, len(comps), C.byref(self._h)))
        if ctx is not None:
            ctx._adopt(self)

    def __del__(self):
        if getattr(self, "_h", None):
            load_library().mgf_compound_free(self._h)
            self._h = None

    def set_pose(self, disp, rot):
        """rot = (s, x, y, z), assumed normalised"""
        _check(load_library().mgf_compound_set_pose(self._h, _v3(disp), Quat(*[float(v) for v in rot])))

    def bounds(self):
        b = Aabb()
        _check(load_library().mgf_compound_bounds(self._h, C.byref(b)))
        return b.c.tup(), b.r.tup()

    def contacts_many(self, moving):
        """moving = MOVING_DTYPE array of swept spheres / capsules -> (offsets, CONTACT_DTYPE array)"""
        moving = np.ascontiguousarray(moving, MOVING_DTYPE)
        n = len(moving)
        off = np.zeros(n + 1, np.uint64)
        total = C.c_int64()
        cap = max(4 * n, 16)
        while True:
            out = np.zeros(cap, CONTACT_DTYPE)
            st = load_library().mgf_compound_contacts_many(self._h, moving.ctypes.data, n, off.ctypes.data, out.ctypes.data, cap, C.byref(total))
            if st == ERR_CAPACITY and total.value > cap:
                cap = total.value
                continue
            _check(st)
            break
        return off.astype(np.int64), out[:total.value]

    def intersections(self, parts):
        parts = np.ascontiguousarray(parts, PARTICLE_DTYPE)
        n = len(parts)
        out = np.zeros(max(n, 1), INTERSECTION_DTYPE)
        hit = np.zeros(max(n, 1), np.int32)
        _check(load_library().mgf_compound_intersections(self._h, parts.ctypes.data, n, out.ctypes.data, hit.ctypes.data))
        return [(tuple(float(v) for v in out[i]["p"]), float(out[i]["t"])) if hit[i] else None for i in range(n)]


class Mesh:
    """mgf::Mesh (mesh.rs:32-73)."""

    def __init__(self, ctx):
        """ctx=None builds a host-only mesh (no device queries)."""
        self._ctx = ctx
        self._h = C.c_void_p()
        _check(load_library().mgf_mesh_new(ctx._h if ctx is not None else None, C.byref(self._h)))
        if ctx is not None:
            ctx._adopt(self)

    def __del__(self):
        if getattr(self, "_h", None):
            load_library().mgf_mesh_free(self._h)
            self._h = None

    def to_json(self):
        """serde_json shape of mgf::Mesh (mesh.rs:31-37)"""
        return _to_json(load_library().mgf_mesh_to_json, self._h)

    @classmethod
    def from_json(cls, ctx, text):
        self = cls.__new__(cls)
        self._ctx = ctx
        self._h = C.c_void_p()
        raw = text.encode()
        _check(load_library().mgf_mesh_from_json(ctx._h if ctx is not None else None, raw, len(raw), C.byref(self._h)))
        if ctx is not None:
            ctx._adopt(self)
        return self

    def push_vert(self, p):
        i = C.c_uint64()
        _check(load_library().mgf_mesh_push_vert(self._h, _v3(p), C.byref(i)))
        return i.value

    def push_face(self, a, b, c):
        i = C.c_uint64()
        _check(load_library().mgf_mesh_push_face(self._h, a, b, c, C.byref(i)))
        return i.value

    def set_pos(self, p):
        _check(load_library().mgf_mesh_set_pos(self._h, _v3(p)))

    def build(self, verts, faces):
        verts = np.ascontiguousarray(verts, np.float32).reshape(-1, 3)
        faces = np.ascontiguousarray(faces, np.uint32).reshape(-1, 3)
        _check(load_library().mgf_mesh_build(self._h, verts.ctypes.data, len(verts), faces.ctypes.data, len(faces)))

    def bvh_dump(self):
        view = load_library().mgf_mesh_bvh_view(self._h)
        return _bvh_dump(view)


def _bvh_dump(handle):
    L = load_library()
    n = L.mgf_bvh_dump(handle, None, None, 0)
    nodes = np.zeros((max(n, 1), 6), np.int64)
    boxes = np.zeros((max(n, 1), 6), np.float32)
    L.mgf_bvh_dump(handle, nodes.ctypes.data, boxes.ctypes.data, n)
    return nodes[:n], boxes[:n]


class Bvh:
    """mgf::BVH<AABB, usize> (bvh.rs:30-310)."""

    def __init__(self, ctx, capacity=None):
        self._ctx = ctx
        self._h = C.c_void_p()
        h = ctx._h if ctx is not None else None  # ctx=None: host-only tree (no device queries)
        if capacity is None:
            _check(load_library().mgf_bvh_new(h, C.byref(self._h)))
        else:
            _check(load_library().mgf_bvh_with_capacity(h, capacity, C.byref(self._h)))
        if ctx is not None:
            ctx._adopt(self)

    def __del__(self):
        if getattr(self, "_h", None):
            load_library().mgf_bvh_free(self._h)
            self._h = None

    def to_json(self):
        """serde_json shape of BVH<AABB, usize> (bvh.rs:29-47 over pool.rs:25-41)"""
        return _to_json(load_library().mgf_bvh_to_json, self._h)

    @classmethod
    def from_json(cls, ctx, text):
        self = cls.__new__(cls)
        self._ctx = ctx
        self._h = C.c_void_p()
        raw = text.encode()
        _check(load_library().mgf_bvh_from_json(ctx._h if ctx is not None else None, raw, len(raw), C.byref(self._h)))
        if ctx is not None:
            ctx._adopt(self)
        return self

    def empty(self):
        return bool(load_library().mgf_bvh_empty(self._h))

    def clear(self):
        _check(load_library().mgf_bvh_clear(self._h))

    def insert(self, c, r, val):
        i = C.c_uint64()
        _check(load_library().mgf_bvh_insert(self._h, C.byref(Aabb(_v3(c), _v3(r))), val, C.byref(i)))
        return i.value

    def remove(self, node):
        _check(load_library().mgf_bvh_remove(self._h, node))

    def root(self):
        i = C.c_uint64()
        _check(load_library().mgf_bvh_root(self._h, C.byref(i)))
        return i.value

    def get_leaf(self, node):
        v = C.c_uint64()
        _check(load_library().mgf_bvh_get_leaf(self._h, node, C.byref(v)))
        return v.value

    def bounds(self, node):
        a = Aabb()
        _check(load_library().mgf_bvh_bounds(self._h, node, C.byref(a)))
        return a.c.tup(), a.r.tup()

    def query(self, c, r):
        hits = []
        cb = HIT_FN(lambda pv, _u: hits.append(pv[0]))
        _check(load_library().mgf_bvh_query(self._h, C.byref(Aabb(_v3(c), _v3(r))), cb, None))
        return hits

    def query_many(self, boxes):
        boxes = np.ascontiguousarray(boxes, np.float32).reshape(-1, 6)
        n = len(boxes)
        off = np.zeros(n + 1, np.uint64)
        total = C.c_int64()
        cap = 1 << 16
        while True:
            vals = np.zeros(cap, np.uint64)
            st = load_library().mgf_bvh_query_many(self._h, boxes.ctypes.data, n, off.ctypes.data, vals.ctypes.data, cap, C.byref(total))
            if st == ERR_CAPACITY and total.value > cap:
                cap = total.value
                continue
            _check(st)
            break
        return off.astype(np.int64), vals[:total.value].astype(np.int64)

    def raytrace(self, p, d, dt=float("inf")):
        """BVH::raytrace (bvh.rs:345-369) for one particle through the callback form: [(value, point, t)] in the
        reference's visiting order."""
        part = np.zeros(1, PARTICLE_DTYPE)
        part["p"], part["d"], part["dt"] = p, d, dt
        hits = []

        def on_hit(pv, pi, _u):
            rec = np.ctypeslib.as_array(C.cast(pi, C.POINTER(C.c_float)), (4,))
            hits.append((int(pv[0]), (float(rec[0]), float(rec[1]), float(rec[2])), float(rec[3])))
        cb = RAY_FN(on_hit)
        _check(load_library().mgf_bvh_raytrace(self._h, part.ctypes.data, cb, None))
        return hits

    def raytrace_many(self, parts):
        """BVH::raytrace for each particle: (offsets, values, intersections with the leaf bounds)."""
        parts = np.ascontiguousarray(parts, PARTICLE_DTYPE)
        n = len(parts)
        off = np.zeros(n + 1, np.uint64)
        total = C.c_int64()
        cap = 1 << 16
        while True:
            vals = np.zeros(cap, np.uint64)
            inter = np.zeros(cap, INTERSECTION_DTYPE)
            st = load_library().mgf_bvh_raytrace_many(self._h, parts.ctypes.data, n, off.ctypes.data, vals.ctypes.data, inter.ctypes.data, cap,
                                                      C.byref(total))
            if st == ERR_CAPACITY and total.value > cap:
                cap = total.value
                continue
            _check(st)
            break
        return off.astype(np.int64), vals[:total.value].astype(np.int64), inter[:total.value]

    def dump(self):
        return _bvh_dump(self._h)


class Solver:
    """Solver<ContactConstraint> (solver.rs:53-79): an insertion-ordered constraint list that is solved on a World's
    RigidBodyVec; the constraints keep their accumulated impulses between solve calls."""

    def __init__(self):
        self._h = C.c_void_p()
        _check(load_library().mgf_solver_new(C.byref(self._h)))

    def __del__(self):
        if getattr(self, "_h", None):
            load_library().mgf_solver_free(self._h)
            self._h = None

    def add_constraint(self, row):
        row = np.ascontiguousarray(row, CONSTRAINT_DTYPE).reshape(1)
        _check(load_library().mgf_solver_add_constraint(self._h, row.ctypes.data))

    def add_constraints(self, rows):
        rows = np.ascontiguousarray(rows, CONSTRAINT_DTYPE)
        _check(load_library().mgf_solver_add_constraints(self._h, rows.ctypes.data, len(rows)))

    def __len__(self):
        return load_library().mgf_solver_len(self._h)

    def clear(self):
        _check(load_library().mgf_solver_clear(self._h))

    def constraints(self):
        n = len(self)
        out = np.zeros(max(n, 1), CONSTRAINT_DTYPE)
        _check(load_library().mgf_solver_read_constraints(self._h, out.ctypes.data, len(out), None))
        return out[:n]

    def solve(self, world, iters):
        _check(load_library().mgf_solver_solve(self._h, world._h, int(iters), C.byref(world.stats)))
        return world.stats


class World:
    """RigidBodyVec + Solver + terrain + broadphase resident on one GPU; `step` is
    mgf_demo/world.rs::World::step."""

    def __init__(self, ctx, params=None):
        self._ctx = ctx
        self._h = C.c_void_p()
        self.stats = StepStats()
        p = C.byref(params) if params is not None else None
        _check(load_library().mgf_world_new(ctx._h, p, C.byref(self._h)))
        ctx._adopt(self)

    def __del__(self):
        if getattr(self, "_h", None):
            load_library().mgf_world_free(self._h)
            self._h = None

    @classmethod
    def from_scene(cls, ctx, scene, params=None):
        w = cls(ctx, params)
        t = scene["terrain"]
        if t is not None:
            m = Mesh(ctx)
            m.build(t["verts"], t["faces"])
            m.set_pos(t["pos"])
            w.set_terrain(m)
        if len(scene["comps"]):
            w.add_bodies(scene["comps"], scene["mass"], scene["restitution"], scene["friction"], scene["force"])
        cb = scene.get("compound")  # bodies of several components, appended after the ordinary ones
        if cb is not None:
            w.add_compound_bodies(cb["comps"], cb["comp_mass"], cb["offsets"], cb["restitution"], cb["friction"], cb["force"])
        if scene.get("v0") is not None:
            w.write_state(v=scene["v0"])
        return w

    def set_terrain(self, mesh):
        _check(load_library().mgf_world_set_terrain(self._h, mesh._h if mesh is not None else None))

    def add_obstacle(self, compound):
        """A static Compound as an obstacle of the world beside the Mesh (mgf_world_add_obstacle; the compound is copied)."""
        _check(load_library().mgf_world_add_obstacle(self._h, compound._h))

    def add_bodies(self, comps, mass, restitution, friction, world_force):
        comps = np.ascontiguousarray(comps, dtype=COMPONENT_DTYPE)
        n = len(comps)
        mass = np.ascontiguousarray(np.broadcast_to(np.asarray(mass, np.float32), (n,)))
        rest = np.ascontiguousarray(np.broadcast_to(np.asarray(restitution, np.float32), (n,)))
        fric = np.ascontiguousarray(np.broadcast_to(np.asarray(friction, np.float32), (n,)))
        force = np.ascontiguousarray(np.broadcast_to(np.asarray(world_force, np.float32), (n, 3)))
        first = C.c_uint64()
        _check(load_library().mgf_world_add_bodies(self._h, comps.ctypes.data, n, mass.ctypes.data, rest.ctypes.data,
                                                   fric.ctypes.data, force.ctypes.data, C.byref(first)))
        return first.value

    def add_compound_bodies(self, comps, comp_mass, offsets, restitution, friction, world_force):
        """Bodies of several components (mgf_world_add_compound_bodies): body b = comps[offsets[b]:offsets[b + 1]]."""
        comps = np.ascontiguousarray(comps, dtype=COMPONENT_DTYPE)
        offsets = np.ascontiguousarray(offsets, np.int64)
        n = len(offsets) - 1
        cm = np.ascontiguousarray(np.broadcast_to(np.asarray(comp_mass, np.float32), (len(comps),)))
        rest = np.ascontiguousarray(np.broadcast_to(np.asarray(restitution, np.float32), (n,)))
        fric = np.ascontiguousarray(np.broadcast_to(np.asarray(friction, np.float32), (n,)))
        force = np.ascontiguousarray(np.broadcast_to(np.asarray(world_force, np.float32), (n, 3)))
        first = C.c_uint64()
        _check(load_library().mgf_world_add_compound_bodies(self._h, comps.ctypes.data, cm.ctypes.data, offsets.ctypes.data, n,
                                                            rest.ctypes.data, fric.ctypes.data, force.ctypes.data, C.byref(first)))
        return first.value

    def __len__(self):
        return load_library().mgf_world_len(self._h)

    def step(self, dt, iters):
        _check(load_library().mgf_world_step(self._h, float(dt), int(iters), C.byref(self.stats)))
        return self.stats

    def step_many(self, dt, iters, n):
        """n ticks in one call; returns the per-tick statistics (a ctypes array of StepStats)."""
        arr = (StepStats * int(n))()
        _check(load_library().mgf_world_step_many(self._h, float(dt), int(iters), int(n), arr))
        if n:
            self.stats = arr[int(n) - 1]
        return arr

    def build_constraints(self, dt):
        _check(load_library().mgf_world_build_constraints(self._h, float(dt), C.byref(self.stats)))
        return self.stats

    def solve(self, iters):
        _check(load_library().mgf_world_solve(self._h, int(iters), C.byref(self.stats)))
        return self.stats

    def complete_motion(self):
        _check(load_library().mgf_world_complete_motion(self._h))

    def integrate(self, dt):
        _check(load_library().mgf_world_integrate(self._h, float(dt)))

    def get(self, index=None, static=None):
        """ConstrainedSet::get: index for Dynamic(i), static=(center, friction) for Static."""
        ref = BodyRef(0, index, Vec3(), 0.0) if static is None else BodyRef(1, 0, _v3(static[0]), float(static[1]))
        vel, info = Velocity(), RigidBodyInfo()
        _check(load_library().mgf_world_get(self._h, C.byref(ref), C.byref(vel), C.byref(info)))
        return vel, info

    def set(self, index, linear, angular):
        ref = BodyRef(0, index, Vec3(), 0.0)
        vel = Velocity(_v3(linear), _v3(angular))
        _check(load_library().mgf_world_set(self._h, C.byref(ref), C.byref(vel)))

    def state(self):
        n = len(self)
        x = np.empty((n, 3), np.float32)  # (every element is written: mgf_world_read_state fills n bodies)
        q = np.empty((n, 4), np.float32)
        v = np.empty((n, 3), np.float32)
        w = np.empty((n, 3), np.float32)
        d = np.empty((n, 3), np.float32)
        _check(load_library().mgf_world_read_state(self._h, x.ctypes.data, q.ctypes.data, v.ctypes.data, w.ctypes.data,
                                                   d.ctypes.data, n))
        return dict(x=x, q=q, v=v, omega=w, delta=d)

    def write_state(self, x=None, q=None, v=None, omega=None, delta=None):
        keep = []

        def p(a, k):
            if a is None:
                return None
            a = np.ascontiguousarray(a, np.float32).reshape(-1, k)
            assert len(a) == len(self)
            keep.append(a)
            return a.ctypes.data
        _check(load_library().mgf_world_write_state(self._h, p(x, 3), p(q, 4), p(v, 3), p(omega, 3), p(delta, 3), len(self)))

    def colliders(self):
        out = np.zeros(len(self), MOVING_DTYPE)
        _check(load_library().mgf_world_read_colliders(self._h, out.ctypes.data, len(out)))
        return out

    def constraints(self):
        n = C.c_int64()
        _check(load_library().mgf_world_read_constraints(self._h, None, 0, C.byref(n)))
        out = np.zeros(max(n.value, 1), CONSTRAINT_DTYPE)
        _check(load_library().mgf_world_read_constraints(self._h, out.ctypes.data, len(out), C.byref(n)))
        return out[:n.value]

    def set_constraints(self, cons):
        cons = np.ascontiguousarray(cons, CONSTRAINT_DTYPE)
        _check(load_library().mgf_world_set_constraints(self._h, cons.ctypes.data, len(cons)))

    def clone(self):
        """RigidBodyVec: Clone (physics.rs:140) with the world around it: an independent world that steps identically."""
        w = World.__new__(World)
        w._ctx, w._h, w.stats = self._ctx, C.c_void_p(), StepStats()
        _check(load_library().mgf_world_clone(self._h, C.byref(w._h)))
        self._ctx._adopt(w)
        return w

    def constraints_new(self, refs_a, refs_b, manifolds, dt):
        """ContactConstraint::new (solver.rs:101-191) for caller-built manifolds: refs are body indices (obj_b: an index, or
        (center, friction) for RigidBodyRef::Static); manifolds a MANIFOLD_DTYPE array.  -> flattened CONSTRAINT_DTYPE rows."""
        manifolds = np.ascontiguousarray(manifolds, MANIFOLD_DTYPE)
        n = len(manifolds)
        ra = (BodyRef * max(n, 1))()
        rb = (BodyRef * max(n, 1))()
        for i in range(n):
            a, b = refs_a[i], refs_b[i]
            ra[i] = BodyRef(0, int(a), Vec3(), 0.0) if not isinstance(a, tuple) else BodyRef(1, 0, _v3(a[0]), float(a[1]))
            rb[i] = BodyRef(0, int(b), Vec3(), 0.0) if not isinstance(b, tuple) else BodyRef(1, 0, _v3(b[0]), float(b[1]))
        cap = int(manifolds["n_contacts"].sum()) if n else 0
        out = np.zeros(max(cap, 1), CONSTRAINT_DTYPE)
        cnt = C.c_int64()
        _check(load_library().mgf_constraints_new(self._h, ra, rb, manifolds.ctypes.data, n, float(dt), out.ctypes.data, cap, C.byref(cnt)))
        return out[:cnt.value]

    # ---- tiling (device pointers of the caller) ----
    def begin_tick(self, dt):
        _check(load_library().mgf_world_begin_tick(self._h, float(dt)))

    def collide(self, dt):
        _check(load_library().mgf_world_collide(self._h, float(dt), C.byref(self.stats)))
        return self.stats

    def select_boundary(self, x_left, x_right, ids_left_ptr, ids_right_ptr, cap):
        nl, nr = C.c_int64(), C.c_int64()
        _check(load_library().mgf_world_select_boundary(self._h, float(x_left), float(x_right), ids_left_ptr, ids_right_ptr,
                                                        int(cap), C.byref(nl), C.byref(nr)))
        return nl.value, nr.value

    def export_bodies(self, ids_ptr, n, dst_ptr):
        _check(load_library().mgf_world_export_bodies(self._h, ids_ptr, int(n), dst_ptr))

    def import_ghosts(self, src_ptr, n):
        _check(load_library().mgf_world_import_ghosts(self._h, src_ptr, int(n)))

    def export_velocities(self, ids_ptr, n, dst_ptr):
        _check(load_library().mgf_world_export_velocities(self._h, ids_ptr, int(n), dst_ptr))

    def import_ghost_velocities(self, src_ptr, n):
        _check(load_library().mgf_world_import_ghost_velocities(self._h, src_ptr, int(n)))

    def ghost_len(self):
        return load_library().mgf_world_ghost_len(self._h)

    # ---- migration between tiles (device pointers, like the ghost calls) ----
    def select_tile(self, x_left, x_right, x_lo, x_hi, ids_left_ptr, ids_right_ptr, ids_migrants_ptr, cap):
        """-> (n_boundary_left, n_boundary_right, n_migrants_left, n_migrants_right)"""
        counts = (C.c_int64 * 4)()
        _check(load_library().mgf_world_select_tile(self._h, float(x_left), float(x_right), float(x_lo), float(x_hi), ids_left_ptr,
                                                    ids_right_ptr, ids_migrants_ptr, int(cap), counts))
        return tuple(int(c) for c in counts)

    def export_migrants(self, ids_ptr, n, dst_ptr):
        _check(load_library().mgf_world_export_migrants(self._h, ids_ptr, int(n), dst_ptr))

    def remove_bodies(self, ids_ptr, n):
        _check(load_library().mgf_world_remove_bodies(self._h, ids_ptr, int(n)))

    def import_migrants(self, src_ptr, n):
        _check(load_library().mgf_world_import_migrants(self._h, src_ptr, int(n)))

    def set_tags(self, tags):
        tags = np.ascontiguousarray(tags, np.uint32)
        _check(load_library().mgf_world_set_tags(self._h, tags.ctypes.data, len(tags)))

    def tags(self):
        out = np.zeros(max(len(self), 1), np.uint32)
        _check(load_library().mgf_world_read_tags(self._h, out.ctypes.data, len(out)))
        return out[:len(self)].copy()

    def solve_enqueue(self, iters):
        _check(load_library().mgf_world_solve_enqueue(self._h, int(iters)))

    def finish(self):
        _check(load_library().mgf_world_finish(self._h, C.byref(self.stats)))
        return self.stats

    def counter(self, name):
        v = C.c_int64()
        _check(load_library().mgf_world_counter(self._h, name.encode(), C.byref(v)))
        return v.value

    def set_option(self, key, value):
        _check(load_library().mgf_world_set_option(self._h, key.encode(), int(value)))

    def device_ptr(self, name):
        p = C.c_void_p()
        nb = C.c_int64()
        _check(load_library().mgf_world_device_ptr(self._h, name.encode(), C.byref(p), C.byref(nb)))
        return p.value, nb.value

    def release_device_ptrs(self):
        _check(load_library().mgf_world_release_device_ptrs(self._h))


def rccl_allow_override(allow=True):
    """Let the environment variable MGF_RCCL_LIB name the collectives library (before the first RCCL call of the process)."""
    _check(load_library().mgf_rccl_allow_override(1 if allow else 0))


def rccl_unique_id():
    """128 bytes from ncclGetUniqueId (rank 0 calls this and hands the bytes to the other ranks)."""
    buf = (C.c_ubyte * 128)()
    _check(load_library().mgf_rccl_unique_id(buf))
    return bytes(buf)


class Tiles:
    """This process's x-slab tiles of one scene behind mgf_tiles_* (the whole tile protocol under the C-ABI; mgf_amd.tiles is
    the same protocol in Python).  worlds[i] owns the slab x_ranges[i]; the tiles are first_tile .. of n_tiles_total."""

    def __init__(self, ctx, worlds, x_ranges, first_tile=0, n_tiles_total=None, halo=1.0, refresh_every=4, migrate=True):  # (refresh_every: tiles.DEFAULT_REFRESH_EVERY)
        self._ctx, self.worlds = ctx, list(worlds)
        n = len(self.worlds)
        total = n if n_tiles_total is None else int(n_tiles_total)
        big = 3.0e38
        lo = (C.c_float * n)(*[float(max(min(r[0], big), -big)) for r in x_ranges])
        hi = (C.c_float * n)(*[float(max(min(r[1], big), -big)) for r in x_ranges])
        hs = (C.c_void_p * n)(*[w._h for w in self.worlds])
        self._h = C.c_void_p()
        _check(load_library().mgf_tiles_create(ctx._h, n, hs, lo, hi, int(first_tile), total, float(halo), int(refresh_every),
                                               1 if migrate else 0, C.byref(self._h)))
        ctx._adopt(self)

    def __del__(self):
        if getattr(self, "_h", None):
            load_library().mgf_tiles_free(self._h)
            self._h = None

    def connect(self, unique_id, rank, n_ranks):
        buf = (C.c_ubyte * 128).from_buffer_copy(unique_id)
        _check(load_library().mgf_tiles_connect(self._h, buf, int(rank), int(n_ranks)))

    def preflight(self):
        n = C.c_int32()
        _check(load_library().mgf_tiles_preflight(self._h, C.byref(n)))
        return n.value

    def step(self, dt, iters):
        arr = (StepStats * len(self.worlds))()
        _check(load_library().mgf_tiles_step(self._h, float(dt), int(iters), arr))
        return arr

    def migrated(self, tile, incoming=True):
        return load_library().mgf_tiles_migrated(self._h, int(tile), 1 if incoming else 0)

    def counter(self, key):
        """exchange_bytes_out / _in / _local, exchange_calls, exchange_ns, host_waits, ticks (mgf_tiles_counter)."""
        v = load_library().mgf_tiles_counter(self._h, key.encode())
        if v < 0:
            raise KeyError(key)
        return int(v)

    def set_option(self, key, value):
        _check(load_library().mgf_tiles_set_option(self._h, key.encode(), int(value)))
