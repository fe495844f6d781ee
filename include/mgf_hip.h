/*
 * mgf_hip.h — C-ABI of the MI355X-native rigid-body step behind mgf's API.
 *
 * The reference (maplant/mgf) is a pure-Rust crate with no FFI; its boundary for the
 * per-tick hot path is the public Rust API re-exported at src/lib.rs:117-150 and the
 * tick assembled in mgf_demo/world.rs:227-294.  Every entry point below cites the
 * reference item it replaces.  A Rust `extern "C"` shim binds these
 * one-to-one and turns non-OK statuses back into the panics the reference raises.
 *
 * Conventions
 *  - plain C, no torch types; opaque handles; (ptr,len) slices borrowed for the call;
 *    outputs into caller buffers with capacity + out-count (MGF_ERR_CAPACITY on overflow,
 *    out-count still reports the number required);
 *  - every call is synchronous: the context's HIP stream is drained before return, so
 *    Rust `&mut self` semantics hold; one handle = one thread at a time (Send, not Sync);
 *  - one mgf_ctx per GPU; all arithmetic is IEEE f32 with no FMA contraction, in the
 *    reference's operation order;
 *  - there is NO CPU fallback: without a HIP device every compute entry point returns
 *    MGF_ERR_HIP.
 */
#ifndef MGF_HIP_H
#define MGF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MGF_API __attribute__((visibility("default")))

/* Rust panics on this path, mapped to status codes (SURVEY.md §8b "Errors"). */
typedef enum mgf_status {
  MGF_OK = 0,
  MGF_ERR_EMPTY = 1,        /* BVH::root on empty tree            bvh.rs:265          */
  MGF_ERR_NOT_OCCUPIED = 2, /* Pool index not occupied            pool.rs:111,160,170 */
  MGF_ERR_NOT_LEAF = 3,     /* BVH::get_leaf on a parent          bvh.rs:274          */
  MGF_ERR_STATIC_REF = 4,   /* RigidBodyRef::into() on Static     physics.rs:174      */
  MGF_ERR_SINGULAR = 5,     /* inertia tensor .invert().unwrap()  physics.rs:212      */
  MGF_ERR_INVALID = 6,      /* bad argument / radius assert       geom.rs:300,328     */
  MGF_ERR_CAPACITY = 7,     /* caller buffer too small                                 */
  MGF_ERR_HIP = 8,          /* HIP runtime failure or no device                        */
  MGF_ERR_OOM = 9
} mgf_status;

/* ---- POD mirrors of the reference's Copy types --------------------------------- */
typedef struct mgf_vec3 { float x, y, z; } mgf_vec3;               /* cgmath Vector3/Point3<f32> */
typedef struct mgf_quat { float s, x, y, z; } mgf_quat;            /* cgmath Quaternion<f32> {s, v} */
typedef struct mgf_aabb { mgf_vec3 c, r; } mgf_aabb;               /* geom.rs:257-260 centre + half extents */

/* Component (compound.rs:33): tag 0 = Sphere{c = p, r}; tag 1 = Capsule{a = p, d, r}. */
enum { MGF_SPHERE = 0, MGF_CAPSULE = 1, MGF_TRIANGLE = 2, MGF_RECTANGLE = 3, MGF_PLANE = 4,
       MGF_RAY = 5, MGF_SEGMENT = 6, MGF_AABB = 7 /* scene I/O only (mgf_geom_to_json): v = {p, d} / {a, b} / {c, r} */ };
typedef struct mgf_component { int32_t tag; mgf_vec3 p; mgf_vec3 d; float r; } mgf_component;
/* Moving<Component> (geom.rs:357): shape + per-step displacement. */
typedef struct mgf_moving_component { mgf_component shape; mgf_vec3 delta; } mgf_moving_component;

/* Generic shape operand for the single-shot narrowphase entry point:
 *   MGF_SPHERE   v = {c.xyz, r}            MGF_CAPSULE  v = {a.xyz, d.xyz, r}
 *   MGF_TRIANGLE v = {a.xyz, b.xyz, c.xyz} MGF_PLANE    v = {n.xyz, d}
 * (MGF_RECTANGLE is not on the hot path: MGF_ERR_INVALID.) */
typedef struct mgf_shape { int32_t kind; float v[12]; } mgf_shape;

typedef struct mgf_contact { mgf_vec3 a, b, n; float t; } mgf_contact;                 /* collision.rs:431-442 */
typedef struct mgf_local_contact { mgf_vec3 local_a, local_b; mgf_contact global; } mgf_local_contact; /* :1410-1419 */

/* RigidBodyRef (physics.rs:158-162): tag 0 Dynamic(index), tag 1 Static{center, friction}. */
typedef struct mgf_body_ref { int32_t tag; uint32_t index; mgf_vec3 center; float friction; } mgf_body_ref;
typedef struct mgf_velocity { mgf_vec3 linear, angular; } mgf_velocity;                /* physics.rs:133-137 */
typedef struct mgf_rigid_body_info {                                                   /* physics.rs:124-130 */
  mgf_vec3 x; float restitution, friction, inv_mass; float inv_moment[9]; /* column-major */
} mgf_rigid_body_info;

/* Compile-time trait constants of the reference, as run-time parameters. */
typedef struct mgf_params {
  float baumgarte;               /* solver.rs:278   0.2  */
  float penetration_slop;        /* solver.rs:277   0.05 */
  float persistent_threshold_sq; /* manifold.rs:38  0.5  */
  float collision_epsilon;       /* geom.rs:27      1e-6 (informational: baked into the kernels) */
  float fat_margin;              /* world.rs:181    0.25 */
} mgf_params;

/* One contact constraint as the solver holds it (solver.rs:82-93, 256-262), flattened for
 * the single-contact manifolds this path produces.  Read-back/debug and bulk-insert format. */
typedef struct mgf_constraint {
  int32_t a, b;                 /* body indices; b = -1 for RigidBodyRef::Static */
  int32_t n_contacts;
  mgf_vec3 normal, t0, t1, ra, rb;
  float bias, normal_mass, tangent_mass0, tangent_mass1, normal_impulse, friction;
} mgf_constraint;

typedef struct mgf_step_stats {
  uint64_t n_bodies;
  uint64_t n_constraints;          /* ContactConstraints handed to the Solver this tick            */
  uint64_t n_terrain_constraints;  /* of which body-vs-Mesh (one per terrain contact, world.rs:243) */
  uint64_t n_pair_candidates;      /* broadphase hits (j < i, tight_i overlaps fat_j)               */
  uint64_t n_terrain_candidates;   /* mesh-BVH face hits                                            */
  uint64_t n_refits;               /* bodies whose swept AABB left their fat AABB (world.rs:235)    */
  uint32_t n_levels;               /* depth of the order-preserving dependency DAG                  */
  uint32_t iters;
  float ms_integrate, ms_broadphase, ms_narrowphase, ms_setup, ms_solve, ms_total; /* HIP-event times of the phases: 0 unless option "phase_timing" is on (each event is a barrier packet: ~50 us of an idle GPU per tick together) */
  uint64_t solver_kernel_launches; /* number of solver kernel launches this tick     */
  float ms_solver_kernels;         /* sum of their HIP-event durations (0 if not timed) */
  uint64_t n_ghost_constraints;    /* of n_constraints: those whose obj_a is a ghost body of a neighbouring tile - a constraint across
                                      a tile face exists on both tiles, on each with the other tile's body as obj_a (0 without tiles) */
} mgf_step_stats;

/* Particle (geom.rs:802-855): a Ray { p, d } has dt = INFINITY; a Segment { a, b } is p = a, d = b - a, dt = 1. */
typedef struct { mgf_vec3 p; mgf_vec3 d; float dt; } mgf_particle;
/* Intersection collision.rs:151-158 */
typedef struct { mgf_vec3 p; float t; } mgf_intersection;

/* Manifold manifold.rs:112-118: time, normal (the UN-renormalised mean of the kept contacts' normals; NaN for an empty
 * group, as in the reference), tangent_vector[2] = compute_basis(normal), the kept (local_a, local_b) pairs. */
#define MGF_MANIFOLD_CAP 8
typedef struct {
  float time; mgf_vec3 normal; mgf_vec3 tangent[2]; int32_t n_contacts;
  mgf_vec3 local_a[MGF_MANIFOLD_CAP]; mgf_vec3 local_b[MGF_MANIFOLD_CAP];
} mgf_manifold;

typedef struct mgf_ctx mgf_ctx;
typedef struct mgf_mesh mgf_mesh;
typedef struct mgf_bvh mgf_bvh;
typedef struct mgf_world mgf_world;
typedef struct mgf_compound mgf_compound;
typedef struct mgf_solver mgf_solver;
typedef struct mgf_tiles mgf_tiles;

/* ---- context ---------------------------------------------------------------------- */
MGF_API mgf_status mgf_ctx_create(int device, mgf_ctx** out);
/* Waits for the context's stream and drops the creator's reference.  Handles made from the context (worlds, meshes, trees, compounds, tile sets)
 * hold references of their own: those still alive keep the struct and its streams until they are freed, in whatever order a garbage collector
 * or a scope frees them; every other call on such a handle fails with MGF_ERR_INVALID ("the context was destroyed"). */
MGF_API void mgf_ctx_destroy(mgf_ctx* ctx);
/* Enqueue all work of this context on a caller-owned hipStream_t (e.g. the stream the caller's RCCL
 * transfers are ordered on) instead of the context's own stream.  The caller keeps ownership. */
MGF_API mgf_status mgf_ctx_set_stream(mgf_ctx* ctx, void* hip_stream);
MGF_API const char* mgf_last_error(void);        /* thread-local message for the last non-OK status */
MGF_API mgf_params mgf_default_params(void);     /* DefaultContactConstraintParams / DefaultPruningParams */
MGF_API const char* mgf_version(void);
/* The device-wide exclusive scan the tick uses for its list offsets (csrc/prims.hip: rocPRIM), on host arrays:
 * out[i] = sum of in[0..i).  Exposed so that the primitive can be tested on its own. */
MGF_API mgf_status mgf_exclusive_scan_u32(mgf_ctx* ctx, const uint32_t* in, int64_t n, uint32_t* out);

/* ---- single-shot narrowphase (unit parity; runs the same device functions as the step) ---
 * Contacts::contacts (collision.rs:471-482) for `a` [moving by vel_a] vs `b` [moving by vel_b];
 * NULL velocity = static operand.  Dispatch follows the reference's trait resolution
 * (collision.rs:484-494, 521-1401).  *count = number of contacts emitted. */
MGF_API mgf_status mgf_contacts(mgf_ctx* ctx, const mgf_shape* a, const mgf_vec3* vel_a, const mgf_shape* b,
                                const mgf_vec3* vel_b, mgf_contact* out, int32_t cap, int32_t* count);
/* Batched form: n independent (a, vel_a, b, vel_b) problems; has_vel bit0 = a moving, bit1 = b moving.
 * out holds 2 slots per problem (no pair on this path emits more), counts[n]. */
MGF_API mgf_status mgf_contacts_batch(mgf_ctx* ctx, int64_t n, const mgf_shape* a, const mgf_vec3* vel_a,
                                      const mgf_shape* b, const mgf_vec3* vel_b, const uint8_t* has_vel,
                                      mgf_contact* out, int32_t* counts);
/* (r06, no reference counterpart: a test entry point) n independent (moving component, triangle) problems - tris: three vertices each - through the
 * cheap conservative reject the tick's front end runs ahead of the body-triangle tests AND through those tests (Contacts<Moving<Component>> for
 * Triangle, collision.rs:610-1086, compound.rs:180-190): far[i] = 1 if the reject drops the problem, counts[i] = contacts the tests report.  A
 * problem with far[i] = 1 and counts[i] > 0 would be a contact the tick loses; tests/test_gpu_tri_reject.py looks for one in millions. */
MGF_API mgf_status mgf_tri_reject_batch(mgf_ctx* ctx, int64_t n, const mgf_moving_component* bodies, const mgf_vec3* tris,
                                        uint8_t* far, int32_t* counts);
/* LocalContacts<Moving<Component>> for Moving<Component> (compound.rs:192-207). */
MGF_API mgf_status mgf_local_contacts_pair(mgf_ctx* ctx, const mgf_moving_component* a, const mgf_moving_component* b,
                                           mgf_local_contact* out, int32_t cap, int32_t* count);
/* Intersects<Capsule>/<Sphere> for Ray (collision.rs:249-359); *hit = 0/1. */
MGF_API mgf_status mgf_ray_capsule(mgf_ctx* ctx, const mgf_vec3* p, const mgf_vec3* d, const mgf_shape* capsule,
                                   mgf_vec3* ip, float* t, int32_t* hit);
/* Inertia::tensor (physics.rs:26-93), column-major 3x3. */
MGF_API mgf_status mgf_inertia_tensor(const mgf_component* c, float mass, float out9[9]);

/* ---- Mesh (mesh.rs:32-73, geom.rs:459): static triangle soup + BVH<AABB,usize> over faces ---- */
MGF_API mgf_status mgf_mesh_new(mgf_ctx* ctx, mgf_mesh** out);                        /* Mesh::new         mesh.rs:40 */
MGF_API void mgf_mesh_free(mgf_mesh* m);
MGF_API mgf_status mgf_mesh_push_vert(mgf_mesh* m, mgf_vec3 p, uint64_t* id);         /* Mesh::push_vert   mesh.rs:58 */
MGF_API mgf_status mgf_mesh_push_face(mgf_mesh* m, uint64_t a, uint64_t b, uint64_t c, uint64_t* id); /* mesh.rs:64 */
MGF_API mgf_status mgf_mesh_set_pos(mgf_mesh* m, mgf_vec3 p);                         /* Shape::set_pos    geom.rs:459 */
MGF_API mgf_status mgf_mesh_build(mgf_mesh* m, const mgf_vec3* verts, int64_t nverts, const uint32_t* faces,
                                  int64_t nfaces);                                    /* bulk push_vert/push_face */
/* LocalContacts<Mesh> for Moving<Component> (collision.rs:1490-1506 over mesh.rs:115-139),
 * contacts in mesh-BVH DFS order. */
MGF_API mgf_status mgf_local_contacts_mesh(mgf_ctx* ctx, const mgf_moving_component* body, const mgf_mesh* mesh,
                                           mgf_local_contact* out, int32_t cap, int32_t* count);

/* ---- BVH<AABB, usize> (bvh.rs:30-310): reference-faithful dynamic tree (insert/remove/balance
 * are inherently sequential and run on the host); queries traverse the uploaded tree on the GPU
 * with the reference's stack discipline, so hit order equals the reference's DFS order. ---- */
MGF_API mgf_status mgf_bvh_new(mgf_ctx* ctx, mgf_bvh** out);                          /* BVH::new            bvh.rs:88  */
MGF_API mgf_status mgf_bvh_with_capacity(mgf_ctx* ctx, uint64_t cap, mgf_bvh** out);  /* BVH::with_capacity  bvh.rs:96  */
MGF_API void mgf_bvh_free(mgf_bvh* b);
MGF_API int32_t mgf_bvh_empty(const mgf_bvh* b);                                      /* BVH::empty          bvh.rs:104 */
MGF_API mgf_status mgf_bvh_clear(mgf_bvh* b);                                         /* BVH::clear          bvh.rs:109 */
MGF_API mgf_status mgf_bvh_insert(mgf_bvh* b, const mgf_aabb* key, uint64_t val, uint64_t* id); /* bvh.rs:125 */
MGF_API mgf_status mgf_bvh_remove(mgf_bvh* b, uint64_t id);                           /* BVH::remove         bvh.rs:220 */
MGF_API mgf_status mgf_bvh_root(const mgf_bvh* b, uint64_t* id);                      /* BVH::root           bvh.rs:263 */
MGF_API mgf_status mgf_bvh_get_leaf(const mgf_bvh* b, uint64_t id, uint64_t* val);    /* BVH::get_leaf       bvh.rs:270 */
MGF_API mgf_status mgf_bvh_bounds(const mgf_bvh* b, uint64_t id, mgf_aabb* out);      /* Index<usize>        bvh.rs:483 */
typedef void (*mgf_bvh_hit_fn)(const uint64_t* val, void* user);
MGF_API mgf_status mgf_bvh_query(mgf_bvh* b, const mgf_aabb* arg, mgf_bvh_hit_fn cb, void* user); /* bvh.rs:283 */
/* Bulk query: n AABBs; hits of query q are out_vals[out_offsets[q] .. out_offsets[q+1]) in DFS order. */
MGF_API mgf_status mgf_bvh_query_many(mgf_bvh* b, const mgf_aabb* args, int64_t n, uint64_t* out_offsets /* n+1 */,
                                      uint64_t* out_vals, int64_t cap, int64_t* total);
/* BVH::raytrace (bvh.rs:345-369): every leaf whose bounds the particle intersects, with that intersection, in the
 * reference's visiting order; _many is the bulk form (CSR offsets per particle). */
typedef void (*mgf_bvh_ray_fn)(const uint64_t* value, const mgf_intersection* inter, void* user);
MGF_API mgf_status mgf_bvh_raytrace(mgf_bvh* b, const mgf_particle* arg, mgf_bvh_ray_fn cb, void* user);
MGF_API mgf_status mgf_bvh_raytrace_many(mgf_bvh* b, const mgf_particle* args, int64_t n, uint64_t* out_offsets, uint64_t* out_vals,
                                         mgf_intersection* out_inter, int64_t cap, int64_t* total);
/* Intersects<Sphere | Capsule | Triangle | Plane> (shapes != NULL) or Intersects<AABB> (boxes != NULL) for n
 * particle/target pairs (collision.rs:169-373); hit[i] = 1 and out[i] filled, or hit[i] = 0. */
MGF_API mgf_status mgf_intersections_batch(mgf_ctx* ctx, int64_t n, const mgf_particle* parts, const mgf_shape* shapes,
                                           const mgf_aabb* boxes, mgf_intersection* out, int32_t* hit);

/* ---- World: RigidBodyVec + Solver + broadphase + terrain, resident in HBM for the whole tick.
 * Replaces mgf_demo/world.rs: World::add_body :178-184 and World::step :227-294, built on
 * RigidBodyVec (physics.rs:141-315), ContactPruner/Manifold (manifold.rs), ContactConstraint and
 * Solver (solver.rs).  Constraint insertion order: body i ascending; for each i the terrain
 * contacts in mesh-BVH DFS order, then partners j < i ascending (DESIGN.md "constraint order"). ---- */
MGF_API mgf_status mgf_world_new(mgf_ctx* ctx, const mgf_params* params, mgf_world** out);
MGF_API void mgf_world_free(mgf_world* w);
MGF_API mgf_status mgf_world_set_terrain(mgf_world* w, const mgf_mesh* mesh);         /* copies; World.terrain */
/* A static Compound (compound.rs:230-352) as an obstacle of the world beside the Mesh (copied; its pose as set when it is added).
 * Every tick each owned body's parts go through Compound::contacts (:334-352) after the body's terrain contacts - obstacles in the
 * order they were added, the body's parts in order - and every contact becomes a constraint against
 * Static{center: the compound's displacement (Shape::center, :289-291), friction: 0}: world.rs:243-251 with the compound in the Mesh's
 * place (the reference's demo world holds a Mesh only; the oracle states the definition, World::obstacles).  At most 256 obstacles of at most 2^18 components each. */
MGF_API mgf_status mgf_world_add_obstacle(mgf_world* w, const mgf_compound* c);
/* World::add_body / RigidBodyVec::add_body (physics.rs:200-218), bulk; MGF_ERR_SINGULAR as the unwrap. */
MGF_API mgf_status mgf_world_add_bodies(mgf_world* w, const mgf_component* comps, int64_t n, const float* mass,
                                        const float* restitution, const float* friction, const mgf_vec3* world_force,
                                        uint64_t* first_id);
/* Bodies of several components (BASELINE config 5).  NOT in the reference - physics.rs:200 takes one Component - so the
 * definition is this build's (oracle: RigidBodyVec::add_compound_body): body b is made of comps[offsets[b] ..
 * offsets[b + 1]) (1..32 components, world coordinates at creation) with masses comp_mass[..]; mass = sum, x = centre of
 * mass, q = identity, inertia = sum of the components' tensors about the centre of mass (the reference's Inertia,
 * physics.rs:30-93); the parts are fixed in the body frame and rebuilt from (x, q) every tick like a single collider
 * (physics.rs:243-251).  Contacts: every pair of parts in order, the first body's outer (Contacts, compound.rs:180-190), local points
 * relative to the bodies' centres, ContactPruner + Manifold::from(pruner) (manifold.rs:72-148) - up to 16 contacts per pair of bodies,
 * each a consecutive single-contact constraint record with the manifold's normal (equivalent to solver.rs:219-248).
 * Bodies of up to 4 components keep their parts in four slots per body; ghost and migrant records carry those four, so such bodies cross
 * tiles like the others (kind bits 2 and 3 of "body_kinds").  Bodies of 5..32 components (r06) keep theirs in a pool of the
 * world; a wave takes a candidate pair of bodies and its lanes the part pairs (the reference's Compound, compound.rs:232-352, a STATIC
 * shape, walks a BVH over its components instead - one CPU thread's way of skipping distant parts).
 * LIMITS: a body of more than 32 components (or of none) is refused with MGF_ERR_INVALID and nothing is added; bodies of more than 4
 * components are refused in tile sets (MGF_ERR_INVALID: the tile records carry four part slots); a tick in which two bodies meet in more
 * than 64 part pairs, or in a manifold of more than 16 contacts, fails with MGF_ERR_CAPACITY. */
MGF_API mgf_status mgf_world_add_compound_bodies(mgf_world* w, const mgf_component* comps, const float* comp_mass,
                                                 const int64_t* offsets /* n + 1 */, int64_t n, const float* restitution,
                                                 const float* friction, const mgf_vec3* world_force, uint64_t* first_id);
MGF_API int64_t mgf_world_len(const mgf_world* w);
/* One tick (world.rs:227-294): complete_motion, integrate, broadphase, narrowphase,
 * ContactConstraint::new for every contact, Solver::solve(iters). */
MGF_API mgf_status mgf_world_step(mgf_world* w, float dt, int32_t iters, mgf_step_stats* stats);
/* n ticks back to back (World::step in a host loop); stats (optional) receives one record per tick.  Same results as n
 * calls of mgf_world_step; with a dataflow solver tick k + 1 is enqueued before tick k's counts are read back, and a
 * device-side guard makes it a no-op when tick k has to be re-run with larger lists (option "pipeline" [1]). */
MGF_API mgf_status mgf_world_step_many(mgf_world* w, float dt, int32_t iters, int64_t n, mgf_step_stats* stats);
/* Same tick split at the solver boundary (for parity tests of the constraint list). */
MGF_API mgf_status mgf_world_build_constraints(mgf_world* w, float dt, mgf_step_stats* stats);
MGF_API mgf_status mgf_world_solve(mgf_world* w, int32_t iters, mgf_step_stats* stats);   /* Solver::solve solver.rs:72 */
/* RigidBodyVec::{complete_motion, integrate} alone (physics.rs:262, 222). */
MGF_API mgf_status mgf_world_complete_motion(mgf_world* w);
MGF_API mgf_status mgf_world_integrate(mgf_world* w, float dt);
/* ConstrainedSet::get / set (physics.rs:272-315). */
MGF_API mgf_status mgf_world_get(mgf_world* w, const mgf_body_ref* r, mgf_velocity* vel, mgf_rigid_body_info* info);
MGF_API mgf_status mgf_world_set(mgf_world* w, const mgf_body_ref* r, const mgf_velocity* vel);
/* Bulk state access; any pointer may be NULL.  delta = collider[i].1 (Moving displacement).  What is asked for is packed on the device in
 * the caller's body order and crosses PCIe once, through pinned memory (all five arrays of 262 144 bodies: 0.9 ms either way). */
MGF_API mgf_status mgf_world_read_state(mgf_world* w, mgf_vec3* x, mgf_quat* q, mgf_vec3* v, mgf_vec3* omega,
                                        mgf_vec3* delta, int64_t cap);
MGF_API mgf_status mgf_world_write_state(mgf_world* w, const mgf_vec3* x, const mgf_quat* q, const mgf_vec3* v,
                                         const mgf_vec3* omega, const mgf_vec3* delta, int64_t n);
MGF_API mgf_status mgf_world_read_colliders(mgf_world* w, mgf_moving_component* out, int64_t cap); /* colliders() :256 */
/* The Solver's constraint list of the last tick, in insertion order. */
MGF_API mgf_status mgf_world_read_constraints(mgf_world* w, mgf_constraint* out, int64_t cap, int64_t* count);
/* Solver::add_constraint in bulk + solve on the resident RigidBodyVec (solver.rs:66-78):
 * replaces the tick's constraint list with `cons` (insertion order = array order). */
MGF_API mgf_status mgf_world_set_constraints(mgf_world* w, const mgf_constraint* cons, int64_t n);
/* ---- the library-level pieces World::step is assembled from, on their own (a caller that does its own collision
 * detection builds manifolds, constraints and a Solver exactly as mgf_demo/world.rs:243-251,279-291 does) ----
 * ContactConstraint::new(pool, obj_a, obj_b, manifold, dt) (solver.rs:101-191) for n caller-built manifolds on the world's
 * resident RigidBodyVec (mix of restitution / friction, bias, normal and tangent masses from ConstrainedSet::get of both
 * bodies).  refs_a[i] must be Dynamic (MGF_ERR_STATIC_REF otherwise: every call site of the reference passes one);
 * refs_b[i] Dynamic or Static{center, friction}.  The manifold's normal and tangent vectors are used as given.  A manifold
 * of m contacts yields m consecutive single-contact records that share its normal and tangents - equivalent under
 * ContactConstraint::solve (solver.rs:219-248: the contacts one after the other on the same velocities).  *count = records
 * written (MGF_ERR_CAPACITY if cap is smaller; *count still reports the number required). */
MGF_API mgf_status mgf_constraints_new(mgf_world* w, const mgf_body_ref* refs_a, const mgf_body_ref* refs_b, const mgf_manifold* manifolds,
                                       int64_t n, float dt, mgf_constraint* out, int64_t cap, int64_t* count);
/* Solver<ContactConstraint> (solver.rs:53-79): new / add_constraint (bulk form too) / solve(&mut rbv, iters) / len.  The
 * handle owns its insertion-ordered list; solve runs it on the world's RigidBodyVec in the exact sequential order, with
 * the tick's executors (the world's own tick list is replaced, as by mgf_world_set_constraints), and the constraints keep
 * their state (normal_impulse) between solve calls as the reference's do.  mgf_solver_clear is `Solver::new()` again
 * (world.rs:228 rebuilds the solver every tick). */
MGF_API mgf_status mgf_solver_new(mgf_solver** out);                                              /* Solver::new            solver.rs:59 */
MGF_API void mgf_solver_free(mgf_solver* s);
MGF_API mgf_status mgf_solver_add_constraint(mgf_solver* s, const mgf_constraint* c);             /* Solver::add_constraint solver.rs:66 */
MGF_API mgf_status mgf_solver_add_constraints(mgf_solver* s, const mgf_constraint* cons, int64_t n);
MGF_API int64_t mgf_solver_len(const mgf_solver* s);
MGF_API mgf_status mgf_solver_clear(mgf_solver* s);
MGF_API mgf_status mgf_solver_read_constraints(const mgf_solver* s, mgf_constraint* out, int64_t cap, int64_t* count);
MGF_API mgf_status mgf_solver_solve(mgf_solver* s, mgf_world* w, int32_t iters, mgf_step_stats* stats); /* Solver::solve solver.rs:72 */
/* RigidBodyVec: Clone (physics.rs:140), with the rest of the world it lives in (terrain copy, parameters, options): an
 * independent world on the same context that steps bit-identically.  Ghosts and the tick's lists are not copied. */
MGF_API mgf_status mgf_world_clone(mgf_world* src, mgf_world** out);
/* ContactPruner::new + push(contact) for every LocalContact of a group, in order, then Manifold::from(pruner)
 * (manifold.rs:42-148) for n groups at once: group i = contacts[offsets[i] .. offsets[i+1]).  params = NULL uses
 * DefaultPruningParams / COLLISION_EPSILON.  The reference's pruner is unbounded; a group that keeps more than
 * MGF_MANIFOLD_CAP contacts returns MGF_ERR_CAPACITY (its n_contacts still reports the count). */
MGF_API mgf_status mgf_manifolds_from_contacts(mgf_ctx* ctx, const mgf_params* params, int64_t n, const uint64_t* offsets,
                                               const mgf_local_contact* contacts, mgf_manifold* out);
/* ---- scene I/O: the serde_json shape of the reference's persistent types (bvh.rs:29-47, pool.rs:25-41, mesh.rs:31-37,
 * geom.rs:256-260; cgmath vectors as {"x","y","z"}).  *_to_json writes a NUL-terminated string; *len is its length
 * (MGF_ERR_CAPACITY if cap < *len + 1).  *_from_json rebuilds the identical tree - entry for entry, free list included -
 * so later inserts reuse the same slots as in the process that wrote the file. */
MGF_API mgf_status mgf_bvh_to_json(const mgf_bvh* b, char* buf, int64_t cap, int64_t* len);
MGF_API mgf_status mgf_bvh_from_json(mgf_ctx* ctx, const char* json, int64_t len, mgf_bvh** out);
MGF_API mgf_status mgf_mesh_to_json(const mgf_mesh* m, char* buf, int64_t cap, int64_t* len);
MGF_API mgf_status mgf_mesh_from_json(mgf_ctx* ctx, const char* json, int64_t len, mgf_mesh** out);
/* The geometry structs of geom.rs:31-357 in serde_json's shape: s->kind selects the struct - MGF_SPHERE {"c","r"},
 * MGF_CAPSULE {"a","d","r"}, MGF_TRIANGLE {"a","b","c"}, MGF_PLANE {"n","d"}, MGF_RECTANGLE {"c","u":[V,V],"e":[f,f]} (v = c, u0,
 * u1, e0, e1), MGF_RAY {"p","d"}, MGF_SEGMENT {"a","b"}, MGF_AABB {"c","r"}; V = {"x","y","z"}.  moving != NULL writes / reads
 * Moving<T>(T, Vector3<f32>), a tuple struct, i.e. [T, V] (geom.rs:356-357; `Component` itself derives no Serialize).
 * Reading follows serde's struct rules: fields in any order, unknown fields ignored, missing or duplicate field = error. */
MGF_API mgf_status mgf_geom_to_json(const mgf_shape* s, const mgf_vec3* moving, char* buf, int64_t cap, int64_t* len);
MGF_API mgf_status mgf_geom_from_json(int32_t kind, const char* json, int64_t len, mgf_shape* out, mgf_vec3* moving);
/* ---- Compound (compound.rs:230-352): a static aggregate of spheres and capsules with a pose and an internal BVH ----
 * mgf_compound_new = Compound::new (components inserted into the BVH in order); set_pose writes the pub fields
 * disp / rot (rot is assumed normalised, as in the reference); contacts_many = Contacts<RHS> for Compound with
 * RHS = Moving<Sphere> / Moving<Capsule> for n moving components at once (per rhs: the contacts in the order the
 * reference's callback receives them, CSR offsets); intersections = Intersects<Compound> for n particles;
 * bounds = BoundedBy<AABB> (MGF_ERR_EMPTY for an empty compound, bvh.rs:265). */
MGF_API mgf_status mgf_compound_new(mgf_ctx* ctx, const mgf_component* comps, int64_t n, mgf_compound** out);
MGF_API void mgf_compound_free(mgf_compound* c);
MGF_API mgf_status mgf_compound_set_pose(mgf_compound* c, mgf_vec3 disp, mgf_quat rot);
MGF_API mgf_status mgf_compound_bounds(const mgf_compound* c, mgf_aabb* out);
MGF_API mgf_status mgf_compound_contacts_many(mgf_compound* c, const mgf_moving_component* rhs, int64_t n, uint64_t* out_offsets,
                                              mgf_contact* out, int64_t cap, int64_t* total);
MGF_API mgf_status mgf_compound_intersections(mgf_compound* c, const mgf_particle* parts, int64_t n, mgf_intersection* out,
                                              int32_t* hit);
/* ---- spatial tiling across the GPUs of a node (one process per GPU) -----------
 * A tick on a tile is begin_tick -> [select_boundary, export_bodies -> neighbour -> import_ghosts]
 * -> collide -> iters x { solve(1) -> [export_velocities -> neighbour -> import_ghost_velocities] }.
 * Ghost bodies are local copies of a neighbour tile's boundary bodies; they collide with owned
 * bodies only (their terrain contacts and ghost-ghost pairs belong to their owner).  All buffers
 * below are DEVICE pointers owned by the caller (e.g. the exchange buffers handed to RCCL).
 * Ghost record: MGF_GHOST_FLOATS = 72 floats  x3 q4 v3 w3 delta3 | tag p3 d3 r | inv_mass I9 restitution friction | n_parts, 3 pad |
 * 4 x (p3 r d3 kind) world parts of a body of several components (zeros otherwise);
 * velocity record: 8 floats v3 w3 0 0. */
MGF_API mgf_status mgf_world_begin_tick(mgf_world* w, float dt);     /* complete_motion + integrate (world.rs:230-231) */
MGF_API mgf_status mgf_world_collide(mgf_world* w, float dt, mgf_step_stats* stats); /* world.rs:233-291 */
/* Owned bodies whose fat AABB reaches below x_left / above x_right, ascending ids. */
MGF_API mgf_status mgf_world_select_boundary(mgf_world* w, float x_left, float x_right, uint32_t* ids_left,
                                             uint32_t* ids_right, int64_t cap, int64_t* n_left, int64_t* n_right);
MGF_API mgf_status mgf_world_export_bodies(mgf_world* w, const uint32_t* ids, int64_t n, float* dst);
MGF_API mgf_status mgf_world_import_ghosts(mgf_world* w, const float* src, int64_t n_ghost);
MGF_API mgf_status mgf_world_export_velocities(mgf_world* w, const uint32_t* ids, int64_t n, float* dst);
MGF_API mgf_status mgf_world_import_ghost_velocities(mgf_world* w, const float* src, int64_t n_ghost);
MGF_API int64_t mgf_world_ghost_len(const mgf_world* w);
/* ---- migration: an owned body whose centre leaves its tile's slab [x_lo, x_hi) changes owner -------
 * mgf_world_select_tile = mgf_world_select_boundary + the migrants of this tick: counts[0..1] boundary bodies
 * (left, right), counts[2..3] bodies with centre.x < x_lo / >= x_hi; ids_migrants receives the left-goers then the
 * right-goers (each ascending), at most cap in total.  A tick without migrants costs one extra counting kernel and
 * no extra host wait.  The tiles driver (mgf_amd/tiles.py) moves the selected bodies at the END of the tick:
 * export_migrants (MGF_MIGRANT_FLOATS floats per body: the body's row of every device array, fat AABB and tag
 * included) -> neighbour -> remove_bodies on the old owner, import_migrants (append) on the new one.  Ids of the
 * remaining bodies shift down on removal; mgf_world_set_tags / read_tags give bodies an identity that survives. */
#define MGF_GHOST_FLOATS 72
#define MGF_MIGRANT_FLOATS 148
MGF_API mgf_status mgf_world_select_tile(mgf_world* w, float x_left, float x_right, float x_lo, float x_hi,
                                         uint32_t* ids_left, uint32_t* ids_right, uint32_t* ids_migrants, int64_t cap,
                                         int64_t* counts /* [4] */);
MGF_API mgf_status mgf_world_export_migrants(mgf_world* w, const uint32_t* ids, int64_t n, float* dst);
MGF_API mgf_status mgf_world_remove_bodies(mgf_world* w, const uint32_t* ids, int64_t n);  /* distinct ids, any order */
MGF_API mgf_status mgf_world_import_migrants(mgf_world* w, const float* src, int64_t n);
MGF_API mgf_status mgf_world_set_tags(mgf_world* w, const uint32_t* tags /* host, one per owned body */, int64_t n);
MGF_API mgf_status mgf_world_read_tags(mgf_world* w, uint32_t* tags /* host */, int64_t cap);
/* Stream-ordered variant of the loop above, for a driver that issues its RCCL transfers on the context's
 * stream (mgf_ctx_set_stream): with option "stream_ordered" = 1 begin_tick / select_boundary's scatter /
 * export_* / import_* only enqueue; mgf_world_solve_enqueue is Solver::solve without the read-back, and
 * mgf_world_finish synchronises once and reports the outcome (status, timings) of everything enqueued. */
MGF_API mgf_status mgf_world_solve_enqueue(mgf_world* w, int32_t iters);
MGF_API mgf_status mgf_world_finish(mgf_world* w, mgf_step_stats* stats);
/* ---- the whole tile protocol under the C-ABI (what a Rust host calls once per tick; mgf_amd/tiles.py is the same protocol in
 * Python, kept as the transport-agnostic reference driver of the CPU tests).  A process owns n_local consecutive x-slab
 * tiles [first_tile, first_tile + n_local) of n_tiles_total, each a mgf_world on the same context with its slab
 * [x_lo, x_hi).  mgf_tiles_step runs one tick of all of them: begin_tick + boundary / migrant selection of every tile with ONE
 * host wait, ghost bodies to the neighbours, collide of every tile enqueued before the first read-back is waited for,
 * iters / refresh_every x { Solver::solve(refresh_every); ghost velocities from their owners }, finish, hand-over of bodies
 * whose centre left their slab.  Between tiles of one process the exchange is a device copy; between processes (one per GPU)
 * it is RCCL point-to-point (ncclSend / ncclRecv in one group per exchange step, on the context's stream, over xGMI):
 * rank 0 calls mgf_rccl_unique_id, the host program hands the 128 bytes to the other ranks, every rank calls
 * mgf_tiles_connect(rank, n_ranks) - rank r's tile range follows rank r - 1's - and mgf_tiles_preflight sums a 1 from every
 * rank over the communicator (0 = not connected).  librccl is loaded at run time, on the first of these calls.
 * stats (optional) receives one record per local tile.  Results are bit-identical to mgf_amd/tiles.py and to the oracle's
 * tile mode, and independent of how the tiles are spread over processes.
 * (Between its own tiles and ranks mgf_tiles_step moves narrower records than the calls above hand to a caller: a world without bodies
 * of several components sends the first 40 floats of a ghost record - the receiver learns the width from the kinds its neighbour announces
 * with its counts - and velocity records are 6 floats, v3 w3.) */
MGF_API mgf_status mgf_tiles_create(mgf_ctx* ctx, int32_t n_local, mgf_world* const* worlds, const float* x_lo, const float* x_hi,
                                    int32_t first_tile, int32_t n_tiles_total, float halo, int32_t refresh_every, int32_t migrate,
                                    mgf_tiles** out);
MGF_API void mgf_tiles_free(mgf_tiles* t);
MGF_API mgf_status mgf_rccl_unique_id(void* id128);
/* Which library provides ncclSend / ncclRecv: librccl.so by default, whatever the environment says.  After
 * mgf_rccl_allow_override(1) - called by the host program before any other mgf_rccl_* / mgf_tiles_connect call - the environment
 * variable MGF_RCCL_LIB may name another one (a site's own RCCL build; the test-suite's stand-in transport). */
MGF_API mgf_status mgf_rccl_allow_override(int32_t allow);
MGF_API mgf_status mgf_tiles_connect(mgf_tiles* t, const void* id128, int32_t rank, int32_t n_ranks);
MGF_API mgf_status mgf_tiles_preflight(mgf_tiles* t, int32_t* n_ranks_seen);
MGF_API mgf_status mgf_tiles_step(mgf_tiles* t, float dt, int32_t iters, mgf_step_stats* stats /* n_local, or NULL */);
/* Options of a tile set.  "exchange_timing" [0]: HIP events around every neighbour exchange feed the counter "exchange_ns" (an event is
 * a barrier packet in the stream: off unless asked for).  "test_fail_tick" = the mgf_tiles_step call (0-based) in which this rank fails on purpose in its collide
 * phase: the protocol's status agreement is then observable (no rank hangs, every rank reports the tick as lost); -1 = never.
 * "retry_lost_ticks" [1]: a tick in which a persistent solver launch gave up on any rank (a device shared with another process) is repeated
 * on every rank - the tiles' owned bodies put back to where the tick found them, the launch-per-frontier executor for the next solves;
 * Solver::solve has no failure mode, solver.rs:72-78 - instead of being reported as lost (counter "ticks_retried"). */
MGF_API mgf_status mgf_tiles_set_option(mgf_tiles* t, const char* key, int64_t value);
MGF_API int64_t mgf_tiles_migrated(const mgf_tiles* t, int32_t tile, int32_t direction_in); /* bodies handed over so far */
/* What the neighbour exchanges of a tile set have cost since its creation (no reference counterpart: world.rs has one World): key =
   "exchange_bytes_out" / "exchange_bytes_in" (rows that crossed a face between RANKS), "exchange_bytes_local" (rows copied between this
   rank's own tiles), "exchange_calls", "exchange_ns" (stream time between the events around the exchanges, the wait for the neighbouring
   rank included; counted while mgf_tiles_set_option "exchange_timing" is 1) and, by kind of exchange, "exchange_{calls|mean_ns|p50_ns|p99_ns|max_ns}_{bodies|
   velocities|handover}" (the calls' own durations: which step of the protocol costs what on the links), "host_waits", "ticks", "ticks_retried"; -1 for an unknown key. */
MGF_API int64_t mgf_tiles_counter(const mgf_tiles* t, const char* key);
/* Options (development and test knobs; defaults in brackets): "time_solver_kernels" [0] HIP events around the
 * solver kernels; "solver_mode" [6] 1 = persistent dataflow launch, 0 = one launch per dependency frontier,
 * 4 = dataflow with out-of-order slots, 5 = block-local dataflow (velocities and counters of a spatial block in LDS),
 * 6 = block-local dataflow with message channels between the blocks (every body a block touches in LDS);
 * (modes 1, 4, 5, 6 are persistent launches whose workgroups wait for one another: they need the device's compute units to
 * themselves - one process per GPU, one context's stream at a time.  Where two processes' launches overlap on one device each
 * may hold part of the compute units; a launch then gives up after about half a second.  Solver::solve cannot fail (solver.rs:72-78), so
 * neither does the call: mgf_world_step / _step_many / _solve put the velocities (and accumulated impulses) back as the launch found
 * them and solve the list again with the launch-per-frontier executor - bit-identical - counted in "solver_abort_fallbacks"; a tile set
 * repeats the tick on every rank ("retry_lost_ticks").  MGF_ERR_HIP "dataflow solver gave up waiting" can still surface only where
 * nothing can be solved again from inside the call: "stream_ordered" = 1 (no wait inside the call), or a tile set with
 * "retry_lost_ticks" = 0 - the state is then that of a tick whose solve did not happen.  Processes that know they share a device say
 * so with "flow_max_blocks");
 * "constraint_order" [0] 1 = the reference's own insertion order, replayed on the host (world.rs:233-291);
 * "pair_brick" [1] (grid broadphase with an 8x8x8-cell box staged in LDS; 0 = every look-up from global memory);
 * "body_pack" [1] (the constraint setup reads collider, motion and info from the packed per-tick copy);
 * "grid_min_frac_pct" [50] (axes of the scene shorter than this percentage of the longest one are widened to it before
 * the Morton cells are laid over it: cells stay near-cubic in an x-slab tile);
 * "flow6_fcap", "flow6_const_lds", "flow6_poll_waves", "flow6_test_cap" (mode 6: foreign-body slots, constants in LDS,
 * polling waves, a test limit that forces the stand-by kernel); "two_pass_candidates" [0]; "broadphase_tree" [0]; "terrain_tree" [0]; "no_fused_narrowphase" [0] (a world of spheres only runs the sphere-sphere test inside the grid broadphase and lists contacts only; 1 = list every accepted partner); "stream_ordered" [0]; "phase_timing" [0] (HIP events at the tick's phase boundaries: mgf_step_stats::ms_*), "time_solver_kernels" [0] (events around the solver launches: ms_solver_kernels); "spin_wait" [1] (the tick's one wait polls an event instead of blocking); "pipeline" [1] (mgf_world_step_many enqueues the next tick before it waits for this one); "cell_fill" [16] (bodies per Morton cell, in eighths, beyond which the broadphase grid gets another level); "no_fused_terrain_rows" [0], "no_fused_scene_bounds" [0] (mgf_world_step and mgf_world_begin_tick list the terrain faces of a body and gather the scene bounds inside the integration kernel; 1 = always the separate kernels); "list_capacity";
 * "fused_contacts" [1] (a world of spheres over a small mesh: rows -> constraint records without candidate lists; 0 = the candidate-list kernels);
 * "front_rows" [1] (r06: a world of single-component bodies that are not all spheres - capsules, mixed - or of bodies of up to two components:
 * the pair search runs the pair test on the partners it accepts, the bodies near the mesh get their faces and the body-triangle test in
 * launches of their own, the constraint records are written from the rows; 0 = candidate lists and one narrowphase launch per shape-pair
 * type); "front_rows_check" [0] (tests: the faces the cheap conservative reject ahead of the body-triangle tests drops are tested all the
 * same - a contact among them is reported as an internal error); "side_stream" [1] (the terrain kernels of that front end run on a
 * second stream of the context beside the pair search; 0 = everything on the context's stream; 3 = the fork and the join without the second
 * stream: what the two events cost by themselves);
 * "wide_list" [1] (r06: the few bodies whose fat box is far larger than the rest's - a body that left the scene and falls at 200 m/s - are kept
 * out of the scene bounds and of the reach of every query of the cell grid, and paired by a launch of their own; the accepted set is the
 * reference's either way (bvh.rs:283-310); engaged by the host from the tick after such a body shows;
 * 0 = never);
 * "cells_in_integrate" [1] (the fused tick's k_integrate works out the bodies' Morton cells over the previous tick's scene bounds);
 * "flow_max_blocks" [0] (the persistent solver launches of this world take at most this many workgroups - one per CU; 0 = all CUs.  Processes that
 * share a device each take a part, so that their launches are resident together); "flow_spin_limit" [0] (tests: 1 = every other workgroup of a
 * persistent launch returns at once and the launch gives up - the world then restores its pre-launch velocities and solves the list with the
 * launch-per-frontier executor, counter "solver_abort_fallbacks");
 * "flow_blocks_per_cu"; "flow_sleep"; "flow_trace"; "debug_bvh"; "flow5_block", "flow5_slow_x2", "flow5_poller", "flow5_test_cap" (block-local solver: block size, wave split, polling wave, a test limit that forces the stand-by kernel); "body_kinds" (OR-in, bit0 sphere, bit1 capsule): the
 * kinds this world's ghosts may have - a tile whose own bodies are all of one kind must be told when a neighbour's are
 * not, because the narrowphase dispatch is chosen on the host (the tiles driver exchanges the masks with the counts). */
MGF_API mgf_status mgf_world_set_option(mgf_world* w, const char* key, int64_t value);
/* Diagnostics: how often a slow path was taken.  name in {"row_overflows", "capacity_retries", "flow5_fallbacks",
 * "grid_too_wide", "flow5_blocks", "flow5_class0|1|2", "terrain_grid", "terrain_row_capacity", "rev_row_capacity", "body_kinds",
 * "flow6_fallbacks", "flow6_fail_reason", "flow6_max_slots", "flow6_max_foreign", "pair_brick_slow_queries", "pair_brick_off_ticks",
 * "max_fat_half_extent_x_milli", "scene_rmax_milli_x|y|z", "scene_ext_milli_x|y|z", "grid_levels", "front_rows", "front_near", "front_faces",
 * "front_slots" (the list-free front end of the last tick: bodies near the mesh, faces accepted, faces that passed the cheap reject),
 * "wide_bodies" (listed in the last tick), "wide_ticks", "wide_overflows" (ticks run again because more bodies were wide than the list holds),
 * "flow6_skipped" (plans that declined the block-local solver because the last launch that did not fit says it still would not), "solver_abort_fallbacks" (Solver::solve calls whose persistent launch gave up and that were solved again from
 * the pre-launch state: Solver::solve has no failure mode, solver.rs:72-78), "device_ptrs_out" (1 while mgf_world_device_ptr's pointers pin
 * the store to the caller's order)}. */
MGF_API mgf_status mgf_world_counter(const mgf_world* w, const char* name, int64_t* out);
/* Raw device pointers of resident state for zero-copy exchange (multi-GPU halo): name in
 * {"x","q","solver_rec","delta"} (the pub fields `x`, `q` of RigidBodyVec physics.rs:142-154 and what ConstrainedSet::get returns,
 * :273-288), rows indexed by the caller's body index.  Valid until the next add_bodies / remove_bodies.  While pointers are out the
 * world keeps its store in the caller's order (the fused tick does not re-sort it into cell order: slower ticks, same results);
 * mgf_world_release_device_ptrs gives them back - after it the pointers must not be used. */
MGF_API mgf_status mgf_world_device_ptr(mgf_world* w, const char* name, void** ptr, int64_t* bytes);
MGF_API mgf_status mgf_world_release_device_ptrs(mgf_world* w);

#ifdef __cplusplus
}
#endif
#endif /* MGF_HIP_H */
