"""mgf_amd — MI355X-native rigid-body step behind mgf's API.

Python host-side mirror of the reference's interface for the per-tick hot path
(RigidBodyVec / Solver / Contact / BVH<AABB> / Mesh / World::step), bound with ctypes to the
C-ABI in include/mgf_hip.h (mgf_amd/libmgf_hip.so, hand-written HIP kernels for gfx950).
There is no CPU fallback: importing works anywhere, but creating a Context without the built
library or without a GPU raises MgfError.
"""
from . import scenes  # noqa: F401
from ._capi import (MgfError, Context, Mesh, Bvh, World, Solver, Tiles, rccl_unique_id, rccl_allow_override, Compound, contacts, contacts_batch, local_contacts_pair,  # noqa: F401
                    local_contacts_mesh, ray_capsule, intersections, particles, manifolds_from_contacts, inertia_tensor, geom_to_json, geom_from_json, default_params, lib_path, load_library,
                    COMPONENT_DTYPE, CONSTRAINT_DTYPE, MOVING_DTYPE, PARTICLE_DTYPE, INTERSECTION_DTYPE, CONTACT_DTYPE, LOCAL_CONTACT_DTYPE, MANIFOLD_DTYPE)
