// ORACLE — TEST INFRASTRUCTURE ONLY (see mgf_math.hpp header).
// extern "C" surface over the CPU restatement, loaded with ctypes by tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg — never by mgf_amd/.
#include <cstdio>
#include <cstring>
#include <utility>

#include "mgf_compound.hpp"
#include "mgf_world.hpp"

using namespace mgfo;

extern "C" {

struct o_vec3 { float x, y, z; };
struct o_quat { float s, x, y, z; };
struct o_aabb { o_vec3 c, r; };
struct o_contact { o_vec3 a, b, n; float t; };
struct o_local_contact { o_vec3 local_a, local_b; o_contact global; };
// kind: 0 sphere {c[0..2], r[3]}; 1 capsule {a[0..2], d[3..5], r[6]};
//       2 triangle {a,b,c}; 3 rectangle {c, u0, u1, e0[9], e1[10]}; 4 plane {n, d[3]}
struct o_shape { int32_t kind; float v[12]; };
struct o_component { int32_t tag; o_vec3 p; o_vec3 d; float r; };  // tag 0 sphere (p=c), 1 capsule (p=a)
struct o_stats {
  uint64_t n_constraints, n_terrain_constraints, n_pair_candidates, n_refits;
  double t_integrate, t_collide, t_solve;
};
struct o_constraint {  // flattened single-contact view of a ContactConstraint (this path: exactly 1 contact)
  int32_t a, b;        // body indices; b = -1 for Static
  int32_t n_contacts;
  o_vec3 normal, t0, t1, ra, rb;
  float bias, normal_mass, tangent_mass0, tangent_mass1, normal_impulse, friction;
};

static inline V3 V(const o_vec3& a) { return v3(a.x, a.y, a.z); }
static inline o_vec3 O(const V3& a) { return o_vec3{a.x, a.y, a.z}; }
static inline o_contact OC(const Contact& c) { return o_contact{O(c.a), O(c.b), O(c.n), c.t}; }
static inline Sphere as_sphere(const o_shape& s) { return Sphere{v3(s.v[0], s.v[1], s.v[2]), s.v[3]}; }
static inline Capsule as_capsule(const o_shape& s) { return Capsule{v3(s.v[0], s.v[1], s.v[2]), v3(s.v[3], s.v[4], s.v[5]), s.v[6]}; }
static inline Triangle as_tri(const o_shape& s) { return Triangle{v3(s.v[0], s.v[1], s.v[2]), v3(s.v[3], s.v[4], s.v[5]), v3(s.v[6], s.v[7], s.v[8])}; }
static inline Rectangle as_rect(const o_shape& s) {
  return Rectangle{v3(s.v[0], s.v[1], s.v[2]), {v3(s.v[3], s.v[4], s.v[5]), v3(s.v[6], s.v[7], s.v[8])}, {s.v[9], s.v[10]}};
}
static inline Plane as_plane(const o_shape& s) { return Plane{v3(s.v[0], s.v[1], s.v[2]), s.v[3]}; }
static inline Component as_component(const o_component& c) {
  if (c.tag == 0) return component(Sphere{V(c.p), c.r});
  return component(Capsule{V(c.p), V(c.d), c.r});
}
static inline o_component from_component(const Component& k) {
  if (k.kind == COMP_SPHERE) return o_component{0, O(k.s.c), o_vec3{0, 0, 0}, k.s.r};
  return o_component{1, O(k.c.a), O(k.c.d), k.c.r};
}

struct Sink {
  o_contact* out; int cap; int n;
  void operator()(const Contact& c) { if (n < cap) out[n] = OC(c); ++n; }
};

// Generic single-shot Contacts::contacts.  vel_a / vel_b NULL = static operand.
// Returns number of contacts emitted (may exceed cap), or -1 if the pair is unsupported.
int mgfo_contacts(const o_shape* a, const o_vec3* vel_a, const o_shape* b, const o_vec3* vel_b, o_contact* out, int cap) {
  Sink sink{out, cap, 0};
  auto cb = [&](const Contact& c) { sink(c); };
  const int ka = a->kind, kb = b->kind;
  if (!vel_a && vel_b) {  // static receiver, moving argument
    V3 vb = V(*vel_b);
    if (kb == 0) {
      Moving<Sphere> m = sweep(as_sphere(*b), vb);
      if (ka == 0) contacts(as_sphere(*a), m, cb);
      else if (ka == 1) contacts(as_capsule(*a), m, cb);
      else if (ka == 2) contacts(as_tri(*a), m, cb);
      else if (ka == 3) contacts(as_rect(*a), m, cb);
      else if (ka == 4) contacts(as_plane(*a), m, cb);
      else return -1;
    } else if (kb == 1) {
      Moving<Capsule> m = sweep(as_capsule(*b), vb);
      if (ka == 0) contacts(as_sphere(*a), m, cb);
      else if (ka == 1) contacts(as_capsule(*a), m, cb);
      else if (ka == 2) contacts(as_tri(*a), m, cb);
      else if (ka == 3) contacts(as_rect(*a), m, cb);
      else if (ka == 4) contacts(as_plane(*a), m, cb);
      else return -1;
    } else return -1;
  } else if (vel_a && !vel_b) {  // moving receiver, static argument
    V3 va = V(*vel_a);
    if (ka == 0) {
      Moving<Sphere> m = sweep(as_sphere(*a), va);
      if (kb == 0) moving_contacts_static(m, as_sphere(*b), cb);
      else if (kb == 1) moving_contacts_static(m, as_capsule(*b), cb);
      else if (kb == 2) moving_contacts_poly(m, as_tri(*b), cb);
      else if (kb == 3) moving_contacts_poly(m, as_rect(*b), cb);
      else if (kb == 4) contacts(as_plane(*b), m, [&](const Contact& c) { sink(neg(c)); });
      else return -1;
    } else if (ka == 1) {
      Moving<Capsule> m = sweep(as_capsule(*a), va);
      if (kb == 0) moving_contacts_static(m, as_sphere(*b), cb);
      else if (kb == 1) moving_contacts_static(m, as_capsule(*b), cb);
      else if (kb == 2) moving_contacts_poly(m, as_tri(*b), cb);
      else if (kb == 3) moving_contacts_poly(m, as_rect(*b), cb);
      else if (kb == 4) contacts(as_plane(*b), m, [&](const Contact& c) { sink(neg(c)); });
      else return -1;
    } else return -1;
  } else if (vel_a && vel_b) {  // both moving (collision.rs:1387)
    V3 va = V(*vel_a), vb = V(*vel_b);
    if (ka == 0 && kb == 0) contacts(sweep(as_sphere(*a), va), sweep(as_sphere(*b), vb), cb);
    else if (ka == 0 && kb == 1) contacts(sweep(as_sphere(*a), va), sweep(as_capsule(*b), vb), cb);
    else if (ka == 1 && kb == 0) contacts(sweep(as_capsule(*a), va), sweep(as_sphere(*b), vb), cb);
    else if (ka == 1 && kb == 1) contacts(sweep(as_capsule(*a), va), sweep(as_capsule(*b), vb), cb);
    else return -1;
  } else return -1;
  return sink.n;
}

// Moving<Component>.local_contacts(&Moving<Component>) compound.rs:192-207
int mgfo_local_contacts_pair(const o_component* a, const o_vec3* delta_a, const o_component* b, const o_vec3* delta_b,
                             o_local_contact* out, int cap) {
  int n = 0;
  local_contacts(sweep(as_component(*a), V(*delta_a)), sweep(as_component(*b), V(*delta_b)), [&](const LocalContact& lc) {
    if (n < cap) out[n] = o_local_contact{O(lc.local_a), O(lc.local_b), OC(lc.global)};
    ++n;
  });
  return n;
}

int mgfo_ray_capsule(const o_vec3* p, const o_vec3* d, const o_shape* cap, o_vec3* ip, float* t) {
  Intersection i;
  if (!ray_capsule(Ray{V(*p), V(*d)}, as_capsule(*cap), &i)) return 0;
  *ip = O(i.p); *t = i.t;
  return 1;
}
int mgfo_ray_sphere(const o_vec3* p, const o_vec3* d, const o_shape* s, o_vec3* ip, float* t) {
  Intersection i;
  if (!ray_sphere(Ray{V(*p), V(*d)}, as_sphere(*s), &i)) return 0;
  *ip = O(i.p); *t = i.t;
  return 1;
}
// Intersects<Shape> for a Ray (dt = inf) or Segment (dt = 1; p = a, d = b - a) collision.rs:169-373.
// kinds: 0 sphere, 1 capsule, 2 triangle, 3 rectangle, 4 plane; returns 1 on hit, 0 on miss, -1 unsupported.
int mgfo_intersection(const o_vec3* p, const o_vec3* d, float dt, const o_shape* sh, o_vec3* ip, float* t) {
  Intersection i;
  Ray r{V(*p), V(*d)};
  bool hit;
  switch (sh->kind) {
    case 0: hit = ray_sphere(r, as_sphere(*sh), &i, dt); break;
    case 1: hit = ray_capsule(r, as_capsule(*sh), &i, dt); break;
    case 2: hit = ray_polygon(r, as_tri(*sh), &i, dt); break;
    case 3: hit = ray_polygon(r, as_rect(*sh), &i, dt); break;
    case 4: hit = ray_plane(r, as_plane(*sh), &i, dt); break;
    default: return -1;
  }
  if (!hit) return 0;
  *ip = O(i.p); *t = i.t;
  return 1;
}
int mgfo_intersection_aabb(const o_vec3* p, const o_vec3* d, float dt, const o_aabb* a, o_vec3* ip, float* t) {
  Intersection i;
  if (!ray_aabb(Ray{V(*p), V(*d)}, AABB{V(a->c), V(a->r)}, &i, dt)) return 0;
  *ip = O(i.p); *t = i.t;
  return 1;
}
// BVH::raytrace bvh.rs:345-369: values + intersections with the leaf bounds, in the reference's DFS order
int64_t mgfo_bvh_raytrace(void* b, const o_vec3* p, const o_vec3* d, float dt, uint64_t* vals, o_vec3* ips, float* ts, int64_t cap) {
  int64_t n = 0;
  Ray r{V(*p), V(*d)};
  ((BVH<size_t>*)b)->raytrace(
      [&](const AABB& bounds) { Intersection i; bool h = ray_aabb(r, bounds, &i, dt); return std::make_pair(h, i); },
      [&](size_t v, const Intersection& i) {
        if (n < cap) { vals[n] = v; ips[n] = O(i.p); ts[n] = i.t; }
        ++n;
      });
  return n;
}
// ---- Compound (compound.rs:230-352) ----
void* mgfo_compound_new(const o_component* comps, int64_t n) {
  std::vector<Component> v;
  for (int64_t i = 0; i < n; ++i) v.push_back(as_component(comps[i]));
  return new Compound(v);
}
void mgfo_compound_free(void* c) { delete (Compound*)c; }
void mgfo_compound_set_pose(void* c, const o_vec3* disp, const o_quat* rot) {
  ((Compound*)c)->disp = V(*disp);
  ((Compound*)c)->rot = Quat{rot->s, v3(rot->x, rot->y, rot->z)};
}
void mgfo_compound_bounds(void* c, o_aabb* out) { AABB b = ((Compound*)c)->bounds(); out->c = O(b.c); out->r = O(b.r); }
// compound.contacts(&Moving(shape, vel)); shape kinds 0 sphere, 1 capsule, 3 rectangle
int mgfo_compound_contacts(void* cp, const o_shape* sh, const o_vec3* vel, o_contact* out, int cap) {
  Compound* c = (Compound*)cp;
  int n = 0;
  auto cb = [&](const Contact& k) { if (n < cap) out[n] = OC(k); ++n; };
  switch (sh->kind) {
    case 0: c->contacts(sweep(as_sphere(*sh), V(*vel)), cb); break;
    case 1: c->contacts(sweep(as_capsule(*sh), V(*vel)), cb); break;
    case 3: c->contacts(sweep(as_rect(*sh), V(*vel)), cb); break;
    default: return -1;
  }
  return n;
}
int mgfo_compound_intersection(void* cp, const o_vec3* p, const o_vec3* d, float dt, o_vec3* ip, float* t) {
  Intersection i;
  if (!((Compound*)cp)->intersection(Ray{V(*p), V(*d)}, dt, &i)) return 0;
  *ip = O(i.p); *t = i.t;
  return 1;
}
// ContactPruner::push for every contact of a group, then Manifold::from(pruner) (manifold.rs:72-148).
// out: time, normal3, t0 3, t1 3 (10 floats), *n = number of contact pairs, pairs = (local_a3, local_b3) each, cap pairs.
void mgfo_manifold_from_contacts(const o_local_contact* lcs, int64_t n_in, float* out10, int32_t* n, float* pairs, int32_t cap) {
  ContactPruner pr;
  for (int64_t i = 0; i < n_in; ++i) pr.push(LocalContact{V(lcs[i].local_a), V(lcs[i].local_b), Contact{V(lcs[i].global.a), V(lcs[i].global.b), V(lcs[i].global.n), lcs[i].global.t}});
  Manifold m = manifold_from(pr);
  out10[0] = m.time;
  out10[1] = m.normal.x; out10[2] = m.normal.y; out10[3] = m.normal.z;
  for (int k = 0; k < 2; ++k) { out10[4 + 3 * k] = m.tangent_vector[k].x; out10[5 + 3 * k] = m.tangent_vector[k].y; out10[6 + 3 * k] = m.tangent_vector[k].z; }
  *n = (int32_t)m.len();
  for (size_t k = 0; k < m.len() && (int32_t)k < cap; ++k) {
    float* q = pairs + 6 * k;
    q[0] = m.contacts[k].a.x; q[1] = m.contacts[k].a.y; q[2] = m.contacts[k].a.z; q[3] = m.contacts[k].b.x; q[4] = m.contacts[k].b.y; q[5] = m.contacts[k].b.z;
  }
}
void mgfo_tri_closest_point(const o_shape* tri, const o_vec3* to, o_vec3* out) { *out = O(tri_closest_point(as_tri(*tri), V(*to))); }
void mgfo_compute_basis(const o_vec3* n, o_vec3* out2) { V3 b[2]; compute_basis(V(*n), b); out2[0] = O(b[0]); out2[1] = O(b[1]); }
void mgfo_quat_from_arc(const o_vec3* src, const o_vec3* dst, o_quat* out) { Quat q = quat_from_arc(V(*src), V(*dst)); *out = o_quat{q.s, q.v.x, q.v.y, q.v.z}; }
void mgfo_rotate_vector(const o_quat* q, const o_vec3* v, o_vec3* out) { *out = O(rotate_vector(Quat{q->s, v3(q->x, q->y, q->z)}, V(*v))); }
void mgfo_tensor(const o_component* c, float m, float* out9) {
  M3 t = tensor(as_component(*c), m);
  for (int k = 0; k < 3; ++k) { out9[3 * k] = t.c[k].x; out9[3 * k + 1] = t.c[k].y; out9[3 * k + 2] = t.c[k].z; }
}
void mgfo_component_bounds(const o_component* c, const o_vec3* delta, o_aabb* out) {
  AABB b = bounds(sweep(as_component(*c), V(*delta)));
  *out = o_aabb{O(b.c), O(b.r)};
}
void mgfo_aabb_combine(const o_aabb* a, const o_aabb* b, o_aabb* out) {
  AABB r = aabb_combine(AABB{V(a->c), V(a->r)}, AABB{V(b->c), V(b->r)});
  *out = o_aabb{O(r.c), O(r.r)};
}
int mgfo_aabb_overlaps(const o_aabb* a, const o_aabb* b) { return aabb_overlaps(AABB{V(a->c), V(a->r)}, AABB{V(b->c), V(b->r)}); }
int mgfo_aabb_contains(const o_aabb* a, const o_aabb* b) { return aabb_contains(AABB{V(a->c), V(a->r)}, AABB{V(b->c), V(b->r)}); }

// ---- Pool<usize> (pool.rs tests) -----------------------------------------
void* mgfo_pool_new() { return new Pool<size_t>(); }
void mgfo_pool_free(void* p) { delete (Pool<size_t>*)p; }
int64_t mgfo_pool_push(void* p, uint64_t v) { return (int64_t)((Pool<size_t>*)p)->push((size_t)v); }
int mgfo_pool_remove(void* p, uint64_t i, uint64_t* out) {
  try { *out = ((Pool<size_t>*)p)->remove((size_t)i); return 0; } catch (...) { return -1; }
}
int mgfo_pool_get(void* p, uint64_t i, uint64_t* out) {
  try { *out = (*(Pool<size_t>*)p)[(size_t)i]; return 0; } catch (...) { return -1; }
}
// iterate occupied entries in index order (pool.rs:223-230); returns count
int64_t mgfo_pool_iter(void* p, uint64_t* idx, uint64_t* val, int64_t cap) {
  Pool<size_t>& pool = *(Pool<size_t>*)p;
  int64_t n = 0;
  for (size_t i = 0; i < pool.entries.size(); ++i)
    if (pool.entries[i].st == Pool<size_t>::OCCUPIED) {
      if (n < cap) { idx[n] = i; val[n] = pool.entries[i].item; }
      ++n;
    }
  return n;
}

// ---- BVH<AABB, usize> ----------------------------------------------------
void* mgfo_bvh_new() { return new BVH<size_t>(); }
void mgfo_bvh_free(void* b) { delete (BVH<size_t>*)b; }
int64_t mgfo_bvh_insert(void* b, const o_aabb* k, uint64_t v) { return (int64_t)((BVH<size_t>*)b)->insert(AABB{V(k->c), V(k->r)}, (size_t)v); }
int mgfo_bvh_remove(void* b, uint64_t id) { try { ((BVH<size_t>*)b)->remove((size_t)id); return 0; } catch (...) { return -1; } }
int64_t mgfo_bvh_root(void* b) { try { return (int64_t)((BVH<size_t>*)b)->get_root(); } catch (...) { return -1; } }
int mgfo_bvh_bounds(void* b, uint64_t id, o_aabb* out) {
  try { const AABB& a = (*(BVH<size_t>*)b)[(size_t)id]; *out = o_aabb{O(a.c), O(a.r)}; return 0; } catch (...) { return -1; }
}
int mgfo_bvh_get_leaf(void* b, uint64_t id, uint64_t* out) {
  try { *out = ((BVH<size_t>*)b)->get_leaf((size_t)id); return 0; } catch (...) { return -1; }
}
int64_t mgfo_bvh_query(void* b, const o_aabb* q, uint64_t* out, int64_t cap) {
  int64_t n = 0;
  ((BVH<size_t>*)b)->query(AABB{V(q->c), V(q->r)}, [&](size_t v) { if (n < cap) out[n] = v; ++n; });
  return n;
}
// dump nodes for structural comparison: per pool slot {occupied, height, parent, is_leaf, leaf/child1, child2}
// Pool internals (pool.rs:25-41) for the scene I/O tests: per entry 0 FreeListEnd / 1 FreeListPtr / 2 Occupied and next_free;
// returns free_list (-1 = None) through *free_list and len through *len
int64_t mgfo_bvh_pool(void* bp, int32_t* state, int64_t* next_free, int64_t cap, int64_t* free_list, int64_t* len) {
  BVH<size_t>& b = *(BVH<size_t>*)bp;
  int64_t n = (int64_t)b.pool.entries.size();
  for (int64_t i = 0; i < n && i < cap; ++i) { state[i] = (int32_t)b.pool.entries[(size_t)i].st; next_free[i] = (int64_t)b.pool.entries[(size_t)i].next_free; }
  *free_list = b.pool.has_free ? (int64_t)b.pool.free_list : -1;
  *len = (int64_t)b.pool.len;
  return n;
}
int64_t mgfo_bvh_dump(void* bp, int64_t* out6, o_aabb* bounds_out, int64_t cap) {
  BVH<size_t>& b = *(BVH<size_t>*)bp;
  int64_t n = (int64_t)b.pool.entries.size();
  for (int64_t i = 0; i < n && i < cap; ++i) {
    int64_t* o = out6 + 6 * i;
    if (!b.pool.occupied((size_t)i)) { o[0] = 0; o[1] = o[2] = o[3] = o[4] = o[5] = 0; continue; }
    const auto& nd = b.pool[(size_t)i];
    o[0] = 1; o[1] = nd.height; o[2] = (int64_t)nd.parent; o[3] = nd.is_leaf;
    o[4] = nd.is_leaf ? (int64_t)nd.leaf : (int64_t)nd.child1;
    o[5] = nd.is_leaf ? 0 : (int64_t)nd.child2;
    if (bounds_out) bounds_out[i] = o_aabb{O(nd.bounds.c), O(nd.bounds.r)};
  }
  return n;
}

// ---- World ---------------------------------------------------------------
void* mgfo_world_new(int order_mode) { World* w = new World(); w->order_mode = order_mode; return w; }
void mgfo_world_free(void* w) { delete (World*)w; }
void mgfo_world_set_params(void* wp, float baumgarte, float slop, float fat_margin) {
  World* w = (World*)wp;
  w->params.baumgarte = baumgarte; w->params.penetration_slop = slop; w->fat_margin = fat_margin;
}
void mgfo_world_set_terrain(void* wp, const o_vec3* verts, int64_t nverts, const uint32_t* faces, int64_t nfaces, const o_vec3* pos) {
  World* w = (World*)wp;
  w->terrain = Mesh();
  for (int64_t i = 0; i < nverts; ++i) w->terrain.push_vert(V(verts[i]));
  for (int64_t i = 0; i < nfaces; ++i) w->terrain.push_face(faces[3 * i], faces[3 * i + 1], faces[3 * i + 2]);
  w->terrain.set_pos(V(*pos));
}
// a static Compound as an obstacle of the world (World::obstacles)
void mgfo_world_add_obstacle(void* wp, const o_component* comps, int64_t n, const o_vec3* disp, const o_quat* rot) {
  World* w = (World*)wp;
  std::vector<Component> cs;
  for (int64_t k = 0; k < n; ++k) cs.push_back(as_component(comps[k]));
  Compound c(cs);
  c.disp = V(*disp);
  c.rot = Quat{rot->s, v3(rot->x, rot->y, rot->z)};
  w->obstacles.push_back(c);
}
int64_t mgfo_world_add_bodies(void* wp, const o_component* comps, int64_t n, const float* mass, const float* rest,
                              const float* fric, const o_vec3* force) {
  World* w = (World*)wp;
  for (int64_t i = 0; i < n; ++i) {
    size_t id;
    if (!w->add_body(as_component(comps[i]), mass[i], rest[i], fric[i], V(force[i]), &id)) return -1;
  }
  return (int64_t)w->bodies.len();
}
// bodies of several components: components of body b are comps[offsets[b] .. offsets[b + 1]) with masses comp_mass[..]
int64_t mgfo_world_add_compound_bodies(void* wp, const o_component* comps, const float* comp_mass, const int64_t* offsets, int64_t n,
                                       const float* rest, const float* fric, const o_vec3* force) {
  World* w = (World*)wp;
  for (int64_t b = 0; b < n; ++b) {
    std::vector<Component> cs;
    for (int64_t k = offsets[b]; k < offsets[b + 1]; ++k) cs.push_back(as_component(comps[k]));
    size_t id;
    if (!w->add_compound_body(cs.data(), comp_mass + offsets[b], cs.size(), rest[b], fric[b], V(force[b]), &id)) return -1;
  }
  return (int64_t)w->bodies.len();
}
// inv_mass, then inv_moment_body column by column (10 floats)
void mgfo_world_body_info(void* wp, int64_t i, float* out) {
  const RigidBodyVec& b = ((World*)wp)->bodies;
  out[0] = b.inv_mass[(size_t)i];
  for (int c = 0; c < 3; ++c) { const V3 col = b.inv_moment_body[(size_t)i].c[c]; out[1 + 3 * c] = col.x; out[2 + 3 * c] = col.y; out[3 + 3 * c] = col.z; }
}
int64_t mgfo_world_len(void* wp) { return (int64_t)((World*)wp)->n_owned; }
void mgfo_world_step(void* wp, float dt, int64_t iters, o_stats* st) {
  World* w = (World*)wp;
  w->step(dt, (size_t)iters);
  if (st) {
    const StepStats& s = w->stats;
    *st = o_stats{s.n_constraints, s.n_terrain_constraints, s.n_pair_candidates, s.n_refits, s.t_integrate, s.t_collide, s.t_solve};
  }
}
// complete_motion + integrate + constraint creation only (no solve): lets tests
// compare the constraint list itself.
void mgfo_world_build_constraints(void* wp, float dt, o_stats* st) {
  World* w = (World*)wp;
  w->build_constraints(dt);
  if (st) {
    const StepStats& s = w->stats;
    *st = o_stats{s.n_constraints, s.n_terrain_constraints, s.n_pair_candidates, s.n_refits, s.t_integrate, s.t_collide, s.t_solve};
  }
}
// ---- tiling counterpart of mgf_world_{begin_tick,collide,select_boundary,export_*,import_*} ----
void mgfo_world_begin_tick(void* wp, float dt) { ((World*)wp)->begin_tick(dt); }
void mgfo_world_collide(void* wp, float dt, o_stats* st) {
  World* w = (World*)wp;
  w->collide(dt);
  if (st) {
    const StepStats& s = w->stats;
    *st = o_stats{s.n_constraints, s.n_terrain_constraints, s.n_pair_candidates, s.n_refits, s.t_integrate, s.t_collide, s.t_solve};
  }
}
int64_t mgfo_world_owned_len(void* wp) { return (int64_t)((World*)wp)->n_owned; }
void mgfo_world_select_boundary(void* wp, float x_left, float x_right, uint32_t* ids_l, uint32_t* ids_r, int64_t cap, int64_t* nl, int64_t* nr) {
  std::vector<uint32_t> l, r;
  ((World*)wp)->select_boundary(x_left, x_right, &l, &r);
  *nl = (int64_t)l.size(); *nr = (int64_t)r.size();
  for (size_t i = 0; i < l.size() && (int64_t)i < cap; ++i) ids_l[i] = l[i];
  for (size_t i = 0; i < r.size() && (int64_t)i < cap; ++i) ids_r[i] = r[i];
}
void mgfo_world_export_bodies(void* wp, const uint32_t* ids, int64_t n, float* out) {
  for (int64_t i = 0; i < n; ++i) ((World*)wp)->export_body(ids[i], out + World::kGhostFloats * i);
}
void mgfo_world_import_ghosts(void* wp, const float* in, int64_t n) {
  World* w = (World*)wp;
  w->drop_ghosts();
  for (int64_t i = 0; i < n; ++i) w->add_ghost(in + World::kGhostFloats * i);
}
void mgfo_world_select_migrants(void* wp, float x_lo, float x_hi, uint32_t* ids_l, uint32_t* ids_r, int64_t cap, int64_t* nl, int64_t* nr) {
  std::vector<uint32_t> l, r;
  ((World*)wp)->select_migrants(x_lo, x_hi, &l, &r);
  *nl = (int64_t)l.size(); *nr = (int64_t)r.size();
  for (size_t i = 0; i < l.size() && (int64_t)i < cap; ++i) ids_l[i] = l[i];
  for (size_t i = 0; i < r.size() && (int64_t)i < cap; ++i) ids_r[i] = r[i];
}
int64_t mgfo_migrant_floats() { return World::kMigrantFloats; }
void mgfo_world_export_migrants(void* wp, const uint32_t* ids, int64_t n, float* out) {
  for (int64_t i = 0; i < n; ++i) ((World*)wp)->export_migrant(ids[i], out + World::kMigrantFloats * i);
}
void mgfo_world_remove_bodies(void* wp, const uint32_t* ids, int64_t n) { ((World*)wp)->remove_bodies(ids, (size_t)n); }
void mgfo_world_import_migrants(void* wp, const float* in, int64_t n) {
  for (int64_t i = 0; i < n; ++i) ((World*)wp)->import_migrant(in + World::kMigrantFloats * i);
}
void mgfo_world_set_tags(void* wp, const uint32_t* t, int64_t n) {
  World* w = (World*)wp;
  for (int64_t i = 0; i < n && (size_t)i < w->n_owned; ++i) w->tags[i] = t[i];
}
void mgfo_world_read_tags(void* wp, uint32_t* t, int64_t cap) {
  World* w = (World*)wp;
  for (size_t i = 0; i < w->n_owned && (int64_t)i < cap; ++i) t[i] = w->tags[i];
}
void mgfo_world_export_velocities(void* wp, const uint32_t* ids, int64_t n, float* out) {
  World* w = (World*)wp;
  for (int64_t i = 0; i < n; ++i) {
    const V3 v = w->bodies.v[ids[i]], o = w->bodies.omega[ids[i]];
    float* p = out + 8 * i;
    p[0] = v.x; p[1] = v.y; p[2] = v.z; p[3] = o.x; p[4] = o.y; p[5] = o.z; p[6] = p[7] = 0.0f;
  }
}
void mgfo_world_import_ghost_velocities(void* wp, const float* in, int64_t n) {
  World* w = (World*)wp;
  for (int64_t i = 0; i < n; ++i) {
    const float* p = in + 8 * i;
    w->bodies.v[w->n_owned + i] = v3(p[0], p[1], p[2]);
    w->bodies.omega[w->n_owned + i] = v3(p[3], p[4], p[5]);
  }
}
void mgfo_world_solve(void* wp, int64_t iters) { World* w = (World*)wp; w->solver.solve(w->bodies, (size_t)iters); }
uint32_t mgfo_world_constraint_depth(void* wp, uint32_t iters) { return ((World*)wp)->constraint_depth(iters); }
// One row per CONTACT: a constraint with m contacts (bodies of several parts only) gives m consecutive rows that share
// its bodies, normal, tangents and friction - the flattened form the HIP path stores (see k_setup_pairs).
int64_t mgfo_world_get_constraints(void* wp, o_constraint* out, int64_t cap) {
  World* w = (World*)wp;
  int64_t row = 0;
  for (const ContactConstraint& c : w->solver.constraints) {
    for (size_t k = 0; k < c.states.size(); ++k, ++row) {
      if (row >= cap) continue;
      o_constraint o;
      std::memset(&o, 0, sizeof(o));
      o.a = c.obj_a.is_static ? -1 : (int32_t)c.obj_a.index;
      o.b = c.obj_b.is_static ? -1 : (int32_t)c.obj_b.index;
      o.n_contacts = 1;
      o.normal = O(c.manifold.normal);
      o.t0 = O(c.manifold.tangent_vector[0]);
      o.t1 = O(c.manifold.tangent_vector[1]);
      o.friction = c.friction;
      o.ra = O(c.manifold.contacts[k].a);
      o.rb = O(c.manifold.contacts[k].b);
      const ContactState& st = c.states[k];
      o.bias = st.bias; o.normal_mass = st.normal_mass; o.tangent_mass0 = st.tangent_mass[0]; o.tangent_mass1 = st.tangent_mass[1];
      o.normal_impulse = st.normal_impulse;
      out[row] = o;
    }
  }
  return row;
}
// State access: x, q, v, omega, delta (collider.1) — the snapshot SURVEY §5 calls for.
void mgfo_world_get_state(void* wp, o_vec3* x, o_quat* q, o_vec3* v, o_vec3* omega, o_vec3* delta) {
  World* w = (World*)wp;
  const RigidBodyVec& b = w->bodies;
  for (size_t i = 0; i < w->n_owned; ++i) {
    if (x) x[i] = O(b.x[i]);
    if (q) q[i] = o_quat{b.q[i].s, b.q[i].v.x, b.q[i].v.y, b.q[i].v.z};
    if (v) v[i] = O(b.v[i]);
    if (omega) omega[i] = O(b.omega[i]);
    if (delta) delta[i] = O(b.collider[i].vel);
  }
}
void mgfo_world_set_state(void* wp, const o_vec3* x, const o_quat* q, const o_vec3* v, const o_vec3* omega, const o_vec3* delta) {
  World* w = (World*)wp;
  RigidBodyVec& b = w->bodies;
  for (size_t i = 0; i < w->n_owned; ++i) {
    if (x) b.x[i] = V(x[i]);
    if (q) b.q[i] = Quat{q[i].s, v3(q[i].x, q[i].y, q[i].z)};
    if (v) b.v[i] = V(v[i]);
    if (omega) b.omega[i] = V(omega[i]);
    if (delta) b.collider[i].vel = V(delta[i]);
  }
}
void mgfo_world_get_colliders(void* wp, o_component* comps, o_vec3* delta) {
  World* w = (World*)wp;
  for (size_t i = 0; i < w->bodies.len(); ++i) {
    comps[i] = from_component(w->bodies.collider[i].shape);
    if (delta) delta[i] = O(w->bodies.collider[i].vel);
  }
}
void mgfo_world_get_inv_moment(void* wp, float* out9_body, float* out9_world) {
  World* w = (World*)wp;
  for (size_t i = 0; i < w->bodies.len(); ++i)
    for (int k = 0; k < 3; ++k) {
      const V3 cb = w->bodies.inv_moment_body[i].c[k], cw = w->bodies.inv_moment[i].c[k];
      if (out9_body) { out9_body[9 * i + 3 * k] = cb.x; out9_body[9 * i + 3 * k + 1] = cb.y; out9_body[9 * i + 3 * k + 2] = cb.z; }
      if (out9_world) { out9_world[9 * i + 3 * k] = cw.x; out9_world[9 * i + 3 * k + 1] = cw.y; out9_world[9 * i + 3 * k + 2] = cw.z; }
    }
}
// Direct ConstrainedSet::set (lib.rs doc-test :74-75)
void mgfo_world_set_velocity(void* wp, int64_t i, const o_vec3* lin, const o_vec3* ang) {
  World* w = (World*)wp;
  w->bodies.set(dynamic_ref((size_t)i), Velocity{V(*lin), V(*ang)});
}
// Body-vs-terrain local contacts for body i in the current collider state (mesh DFS order)
int64_t mgfo_world_terrain_contacts(void* wp, int64_t i, o_local_contact* out, int64_t cap) {
  World* w = (World*)wp;
  int64_t n = 0;
  local_contacts(w->bodies.collider[(size_t)i], w->terrain, [&](const LocalContact& lc) {
    if (n < cap) out[n] = o_local_contact{O(lc.local_a), O(lc.local_b), OC(lc.global)};
    ++n;
  });
  return n;
}
void* mgfo_world_terrain_bvh(void* wp) { return &((World*)wp)->terrain.bvh; }
void* mgfo_world_bvh(void* wp) { return &((World*)wp)->bvh; }

// ---- the RigidBodyVec / ConstrainedSet / ContactConstraint / Solver surface on its own (boundary tests) ----
// RigidBodyVec::integrate physics.rs:222 and ::complete_motion :262, each alone
void mgfo_world_integrate(void* wp, float dt) { ((World*)wp)->bodies.integrate(dt); }
void mgfo_world_complete_motion(void* wp) { ((World*)wp)->bodies.complete_motion(); }
// ConstrainedSet::get physics.rs:273-304: out = linear3 angular3 | x3 restitution friction inv_mass | inv_moment9 (column-major)
void mgfo_world_get(void* wp, int32_t is_static, int64_t index, const o_vec3* center, float friction, float* out) {
  const RigidBodyVec& b = ((World*)wp)->bodies;
  RigidBodyRef r = is_static ? static_ref(V(*center), friction) : dynamic_ref((size_t)index);
  Velocity vel; RigidBodyInfo info;
  b.get(r, &vel, &info);
  out[0] = vel.linear.x; out[1] = vel.linear.y; out[2] = vel.linear.z; out[3] = vel.angular.x; out[4] = vel.angular.y; out[5] = vel.angular.z;
  out[6] = info.x.x; out[7] = info.x.y; out[8] = info.x.z; out[9] = info.restitution; out[10] = info.friction; out[11] = info.inv_mass;
  for (int c = 0; c < 3; ++c) { out[12 + 3 * c] = info.inv_moment.c[c].x; out[13 + 3 * c] = info.inv_moment.c[c].y; out[14 + 3 * c] = info.inv_moment.c[c].z; }
}
// ContactConstraint::new solver.rs:101-191 for ONE caller-built Manifold (normal, tangent_vector[2], m contact pairs) on the
// world's RigidBodyVec; obj_a Dynamic(index_a), obj_b Dynamic(index_b) or Static{center_b, friction_b} (index_b < 0).
// Writes m flattened rows (as mgfo_world_get_constraints does).
void mgfo_constraint_new(void* wp, int64_t index_a, int64_t index_b, const o_vec3* center_b, float friction_b, const o_vec3* normal,
                         const o_vec3* tangents /* 2 */, int64_t m, const o_vec3* local_a, const o_vec3* local_b, float dt,
                         o_constraint* out) {
  const RigidBodyVec& b = ((World*)wp)->bodies;
  Manifold mf;
  mf.time = 0.0f; mf.normal = V(*normal); mf.tangent_vector[0] = V(tangents[0]); mf.tangent_vector[1] = V(tangents[1]);
  for (int64_t k = 0; k < m; ++k) mf.contacts.push_back(ContactPair{V(local_a[k]), V(local_b[k])});
  RigidBodyRef ra = dynamic_ref((size_t)index_a);
  RigidBodyRef rb = index_b < 0 ? static_ref(V(*center_b), friction_b) : dynamic_ref((size_t)index_b);
  ContactConstraint c = ContactConstraint::make(b, ra, rb, mf, dt, ((World*)wp)->params);
  for (size_t k = 0; k < c.states.size(); ++k) {
    o_constraint o;
    std::memset(&o, 0, sizeof(o));
    o.a = (int32_t)index_a; o.b = index_b < 0 ? -1 : (int32_t)index_b; o.n_contacts = 1;
    o.normal = O(c.manifold.normal); o.t0 = O(c.manifold.tangent_vector[0]); o.t1 = O(c.manifold.tangent_vector[1]);
    o.friction = c.friction; o.ra = O(c.manifold.contacts[k].a); o.rb = O(c.manifold.contacts[k].b);
    const ContactState& st = c.states[k];
    o.bias = st.bias; o.normal_mass = st.normal_mass; o.tangent_mass0 = st.tangent_mass[0]; o.tangent_mass1 = st.tangent_mass[1];
    o.normal_impulse = st.normal_impulse;
    out[k] = o;
  }
}
// Solver::new + add_constraint x n + solve(iters) solver.rs:59-78 on the world's RigidBodyVec, from flattened single-contact
// rows (a Static obj_b's velocity and inverse masses are zero whatever its centre, physics.rs:289-302, so the rows need not
// carry it); normal_impulse is read from and written back to the rows (the Solver owns its constraints' state).
void mgfo_solver_solve(void* wp, o_constraint* rows, int64_t n, int64_t iters) {
  RigidBodyVec& b = ((World*)wp)->bodies;
  Solver s;
  for (int64_t i = 0; i < n; ++i) {
    const o_constraint& o = rows[i];
    ContactConstraint c;
    c.obj_a = dynamic_ref((size_t)o.a);
    c.obj_b = o.b < 0 ? static_ref(v3(0, 0, 0), 0.0f) : dynamic_ref((size_t)o.b);
    c.manifold.time = 0.0f; c.manifold.normal = V(o.normal); c.manifold.tangent_vector[0] = V(o.t0); c.manifold.tangent_vector[1] = V(o.t1);
    c.manifold.contacts.push_back(ContactPair{V(o.ra), V(o.rb)});
    c.friction = o.friction;
    ContactState st;
    st.bias = o.bias; st.normal_mass = o.normal_mass; st.normal_impulse = o.normal_impulse;
    st.tangent_mass[0] = o.tangent_mass0; st.tangent_mass[1] = o.tangent_mass1; st.tangent_impulse[0] = st.tangent_impulse[1] = 0.0f;
    c.states.push_back(st);
    s.add_constraint(std::move(c));
  }
  s.solve(b, (size_t)iters);
  for (int64_t i = 0; i < n; ++i) rows[i].normal_impulse = s.constraints[(size_t)i].states[0].normal_impulse;
}

}  // extern "C"
