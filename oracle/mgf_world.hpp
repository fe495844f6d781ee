// ORACLE — TEST INFRASTRUCTURE ONLY (see mgf_math.hpp header).
// CPU restatement of
//   src/mesh.rs:32-139        Mesh (static triangle soup + BVH over faces) and Mesh::contacts
//   src/collision.rs:1490-1506 LocalContacts<Mesh> for Moving<Component>
//   mgf_demo/world.rs:178-184  World::add_body (fat margin 0.25)
//   mgf_demo/world.rs:227-294  World::step — the tick
// Constraint insertion order modes:
//   ORDER_DEMO      — exactly world.rs: for each body i, terrain contacts in mesh-BVH
//                     DFS order, then partners j<i in world-BVH DFS order (the BVH is
//                     mutated inside the loop, world.rs:235-238).
//   ORDER_CANONICAL — same terrain order; partners j<i sorted ascending.  The set of
//                     constraints is identical (the narrowphase decides), only the
//                     order among body i's partners differs.  This is the order the
//                     HIP path emits.
// PARITY STATUS: World::step has no reference test ("parity unpinned").
#pragma once
#include <algorithm>
#include <array>
#include <chrono>

#include "mgf_bvh.hpp"
#include "mgf_compound.hpp"
#include "mgf_physics.hpp"

namespace mgfo {

struct Mesh {  // mesh.rs:32-37
  V3 x = v3(0, 0, 0);
  std::vector<V3> verts;
  std::vector<std::array<size_t, 3>> faces;
  BVH<size_t> bvh;

  size_t push_vert(V3 p) { verts.push_back(p); return verts.size() - 1; }  // mesh.rs:58-62
  size_t push_face(size_t a, size_t b, size_t c) {                         // mesh.rs:64-73
    Triangle tri{verts[a], verts[b], verts[c]};
    size_t index = faces.size();
    faces.push_back({a, b, c});
    bvh.insert(bounds(tri), index);
    return index;
  }
  V3 center() const { return x; }                      // mesh.rs:89-91
  void set_pos(V3 p) { V3 disp = p - center(); x += disp; }  // geom.rs:459-462 + mesh.rs:76-80

  // the face loop of Mesh::contacts (mesh.rs:115-139) with the body's test left to the caller: f(triangle at x)
  template <class F>
  void for_faces(const AABB& query, F&& f) const {
    bvh.query(query - x, [&](size_t face_index) {
      const auto& fc = faces[face_index];
      f(Triangle{verts[fc[0]] + x, verts[fc[1]] + x, verts[fc[2]] + x});
    });
  }
  // Contacts<RHS> for Mesh, RHS = Moving<Component>  mesh.rs:115-139
  template <class F>
  bool contacts(const Moving<Component>& rhs, F&& cb) const {
    bool collided = false;
    bvh.query(bounds(rhs) - x, [&](size_t face_index) {
      const auto& f = faces[face_index];
      V3 a = verts[f[0]] + x, b = verts[f[1]] + x, c = verts[f[2]] + x;
      Triangle tri{a, b, c};
      mgfo::contacts(rhs, tri, [&](const Contact& k) {
        collided = true;
        cb(Contact{k.b, k.a, -k.n, k.t});
      });
    });
    return collided;
  }
};

// LocalContacts<Arg = Mesh> for Moving<Recv = Component>  collision.rs:1490-1506
template <class F>
static inline bool local_contacts(const Moving<Component>& self, const Mesh& rhs, F&& cb) {
  return rhs.contacts(self, [&](const Contact& c) {
    V3 a_c = center(self.shape) + self.vel * c.t;
    V3 b_c = rhs.center();
    cb(LocalContact{c.b + -a_c, c.a + -b_c, neg(c)});
  });
}

enum OrderMode : int { ORDER_DEMO = 0, ORDER_CANONICAL = 1 };

struct StepStats {
  uint64_t n_constraints = 0;
  uint64_t n_terrain_constraints = 0;
  uint64_t n_pair_candidates = 0;  // broadphase hits with j < i
  uint64_t n_refits = 0;
  double t_integrate = 0, t_collide = 0, t_solve = 0;  // seconds
};

struct World {
  RigidBodyVec bodies;
  std::vector<size_t> bvh_ids;
  BVH<size_t> bvh;
  Mesh terrain;
  // Static Compounds as obstacles of the world beside the Mesh (round 3).  world.rs holds one Mesh; a Compound
  // (compound.rs:230-352) is the crate's other static aggregate and its Contacts<RHS> (:334-352) serves a moving Sphere or Capsule
  // the way Mesh::contacts does - so the harness treats it the same way: after a body's terrain contacts, every obstacle in
  // insertion order, the body's parts in order, `obstacle.contacts(&Moving(part, v))`; every contact its own constraint against
  // Static{center: obstacle.center() (= its displacement, compound.rs:289-291), friction: 0} (world.rs:243-251 with the obstacle in the Mesh's place).
  std::vector<Compound> obstacles;
  int order_mode = ORDER_DEMO;
  float fat_margin = 0.25f;  // world.rs:181,237
  ContactConstraintParams params;
  Solver solver;  // last step's solver (kept for inspection)
  StepStats stats;
  // Spatial tiling (not in the reference, which is single-process): bodies [n_owned, len) are ghosts,
  // local copies of a neighbouring tile's boundary bodies.  They collide with owned bodies only;
  // their terrain contacts and ghost-ghost pairs belong to their owner tile.  Mirrors the HIP path's
  // mgf_world_import_ghosts & co. so the tiled algorithm has an exact CPU counterpart.
  size_t n_owned = 0;
  std::vector<uint32_t> tags;  // caller-defined identity per owned body (travels with a migrant)

  // world.rs:178-184
  bool add_body(const Component& col, float mass, float rest, float fric, V3 world_force, size_t* id_out) {
    size_t id;
    if (!bodies.add_body(col, mass, rest, fric, world_force, &id)) return false;
    AABB b = bounds(bodies.collider[id]);
    size_t bvh_id = bvh.insert(b + fat_margin, id);
    bvh_ids.push_back(bvh_id);
    n_owned = bodies.len();
    tags.resize(n_owned, 0u);
    bodies.sync_parts();
    if (id_out) *id_out = id;
    return true;
  }
  // a body of several components (see RigidBodyVec::add_compound_body); single-process worlds only
  bool add_compound_body(const Component* comps, const float* masses, size_t k, float rest, float fric, V3 world_force, size_t* id_out) {
    size_t id;
    if (!bodies.add_compound_body(comps, masses, k, rest, fric, world_force, &id)) return false;
    bvh_ids.push_back(bvh.insert(body_bounds(id) + fat_margin, id));
    n_owned = bodies.len();
    tags.resize(n_owned, 0u);
    if (id_out) *id_out = id;
    return true;
  }
  // swept bounds of a body: its collider's, or the union over its parts
  AABB body_bounds(size_t i) const {
    AABB b = bounds(bodies.part(i, 0));
    for (size_t k = 1; k < bodies.n_parts(i); ++k) b = aabb_combine(b, bounds(bodies.part(i, k)));
    return b;
  }

  void drop_ghosts() {
    RigidBodyVec& b = bodies;
    for (size_t i = n_owned; i < bvh_ids.size(); ++i) bvh.remove(bvh_ids[i]);
    bvh_ids.resize(n_owned);
    b.x.resize(n_owned); b.q.resize(n_owned); b.v.resize(n_owned); b.omega.resize(n_owned); b.force.resize(n_owned);
    b.torque.resize(n_owned); b.restitution.resize(n_owned); b.friction.resize(n_owned); b.inv_mass.resize(n_owned);
    b.inv_moment_body.resize(n_owned, m3_zero()); b.inv_moment.resize(n_owned, m3_zero());
    b.constructor.resize(n_owned); b.collider.resize(n_owned);
    b.sync_parts();
    tags.resize(n_owned, 0u);
  }
  // record (kGhostFloats = 72): x3 q4 v3 w3 delta3 | tag p3 d3 r | inv_mass I9 restitution friction | n_parts, 3 pad |
  // 4 x (p3 r d3 kind) world parts of a body of several components (zeros for an ordinary body)
  static constexpr int kGhostFloats = 72;
  static void put_part(float* o, const Component& c) {
    uint32_t kind = (uint32_t)c.kind;
    if (c.kind == COMP_SPHERE) { o[0] = c.s.c.x; o[1] = c.s.c.y; o[2] = c.s.c.z; o[3] = c.s.r; o[4] = o[5] = o[6] = 0.0f; }
    else { o[0] = c.c.a.x; o[1] = c.c.a.y; o[2] = c.c.a.z; o[3] = c.c.r; o[4] = c.c.d.x; o[5] = c.c.d.y; o[6] = c.c.d.z; }
    std::memcpy(&o[7], &kind, 4);
  }
  static Component get_part(const float* o) {
    uint32_t kind;
    std::memcpy(&kind, &o[7], 4);
    if (kind == (uint32_t)COMP_SPHERE) return component(Sphere{v3(o[0], o[1], o[2]), o[3]});
    return component(Capsule{v3(o[0], o[1], o[2]), v3(o[4], o[5], o[6]), o[3]});
  }
  void add_ghost(const float* o) {
    RigidBodyVec& b = bodies;
    Component k;
    uint32_t tag;
    std::memcpy(&tag, &o[16], 4);
    if (tag == 0u) k = component(Sphere{v3(o[17], o[18], o[19]), o[23]});
    else k = component(Capsule{v3(o[17], o[18], o[19]), v3(o[20], o[21], o[22]), o[23]});
    V3 delta = v3(o[13], o[14], o[15]);
    b.x.push_back(v3(o[0], o[1], o[2]));
    b.q.push_back(Quat{o[3], v3(o[4], o[5], o[6])});
    b.v.push_back(v3(o[7], o[8], o[9]));
    b.omega.push_back(v3(o[10], o[11], o[12]));
    b.force.push_back(v3(0, 0, 0)); b.torque.push_back(v3(0, 0, 0));
    b.restitution.push_back(o[34]); b.friction.push_back(o[35]); b.inv_mass.push_back(o[24]);
    M3 I = m3_new(o[25], o[26], o[27], o[28], o[29], o[30], o[31], o[32], o[33]);
    b.inv_moment_body.push_back(m3_zero()); b.inv_moment.push_back(I);
    uint32_t n_parts;
    std::memcpy(&n_parts, &o[36], 4);
    b.constructor.push_back(ComponentConstructor{n_parts ? RigidBodyVec::KIND_COMPOUND : k.kind, o[23], 0.0f});
    b.collider.push_back(sweep(k, delta));
    b.sync_parts();
    size_t id = b.len() - 1;
    for (uint32_t pk = 0; pk < n_parts; ++pk) {
      Component wc = get_part(o + 40 + 8 * pk);
      b.parts[id].push_back(sweep(wc, delta));
      b.parts_local[id].push_back(wc);  // a ghost is never integrated here: only the count matters
    }
    bvh_ids.push_back(bvh.insert(body_bounds(id) + fat_margin, id));
  }
  void export_body(size_t i, float* o) const {
    const RigidBodyVec& b = bodies;
    const Moving<Component>& c = b.collider[i];
    o[0] = b.x[i].x; o[1] = b.x[i].y; o[2] = b.x[i].z;
    o[3] = b.q[i].s; o[4] = b.q[i].v.x; o[5] = b.q[i].v.y; o[6] = b.q[i].v.z;
    o[7] = b.v[i].x; o[8] = b.v[i].y; o[9] = b.v[i].z;
    o[10] = b.omega[i].x; o[11] = b.omega[i].y; o[12] = b.omega[i].z;
    o[13] = c.vel.x; o[14] = c.vel.y; o[15] = c.vel.z;
    uint32_t tag = (uint32_t)c.shape.kind;
    std::memcpy(&o[16], &tag, 4);
    if (c.shape.kind == COMP_SPHERE) { o[17] = c.shape.s.c.x; o[18] = c.shape.s.c.y; o[19] = c.shape.s.c.z; o[20] = o[21] = o[22] = 0.0f; o[23] = c.shape.s.r; }
    else { o[17] = c.shape.c.a.x; o[18] = c.shape.c.a.y; o[19] = c.shape.c.a.z; o[20] = c.shape.c.d.x; o[21] = c.shape.c.d.y; o[22] = c.shape.c.d.z; o[23] = c.shape.c.r; }
    o[24] = b.inv_mass[i];
    for (int k = 0; k < 3; ++k) { o[25 + 3 * k] = b.inv_moment[i].c[k].x; o[26 + 3 * k] = b.inv_moment[i].c[k].y; o[27 + 3 * k] = b.inv_moment[i].c[k].z; }
    o[34] = b.restitution[i]; o[35] = b.friction[i];
    for (int k = 36; k < kGhostFloats; ++k) o[k] = 0.0f;
    const uint32_t n_parts = b.constructor[i].kind == RigidBodyVec::KIND_COMPOUND ? (uint32_t)b.parts[i].size() : 0u;
    std::memcpy(&o[36], &n_parts, 4);
    for (uint32_t pk = 0; pk < n_parts; ++pk) put_part(o + 40 + 8 * pk, b.parts[i][pk].shape);
  }
  // owned bodies whose fat box reaches below x_left / above x_right (ascending ids)
  void select_boundary(float x_left, float x_right, std::vector<uint32_t>* left, std::vector<uint32_t>* right) const {
    left->clear(); right->clear();
    for (size_t i = 0; i < n_owned; ++i) {
      const AABB& fb = bvh[bvh_ids[i]];
      if (fb.c.x - fb.r.x < x_left) left->push_back((uint32_t)i);
      if (fb.c.x + fb.r.x > x_right) right->push_back((uint32_t)i);
    }
  }

  // ---- migration between tiles (not in the reference): an owned body whose centre leaves the slab
  // [x_lo, x_hi) is handed to the neighbouring tile with its whole state, persistent fat box included.
  static constexpr int kMigrantFloats = 148;  // the ghost record (72) | force3 inv_moment_body9 constructor3 fat box6 tag, 2 pad | 4 local parts | padding (the HIP path's record is 148 floats)
  void select_migrants(float x_lo, float x_hi, std::vector<uint32_t>* left, std::vector<uint32_t>* right) const {
    left->clear(); right->clear();
    for (size_t i = 0; i < n_owned; ++i) {
      const float cx = bodies.x[i].x;
      if (cx < x_lo) left->push_back((uint32_t)i);
      else if (cx >= x_hi) right->push_back((uint32_t)i);
    }
  }
  void export_migrant(size_t i, float* o) const {
    const RigidBodyVec& b = bodies;
    for (int k = 0; k < kMigrantFloats; ++k) o[k] = 0.0f;
    export_body(i, o);  // the ghost record, world parts included
    float* e = o + kGhostFloats;
    e[0] = b.force[i].x; e[1] = b.force[i].y; e[2] = b.force[i].z;
    for (int k = 0; k < 3; ++k) { e[3 + 3 * k] = b.inv_moment_body[i].c[k].x; e[4 + 3 * k] = b.inv_moment_body[i].c[k].y; e[5 + 3 * k] = b.inv_moment_body[i].c[k].z; }
    uint32_t kind = (uint32_t)b.constructor[i].kind;
    std::memcpy(&e[12], &kind, 4);
    e[13] = b.constructor[i].r; e[14] = b.constructor[i].half_h;
    const AABB& fb = bvh[bvh_ids[i]];
    e[15] = fb.c.x; e[16] = fb.c.y; e[17] = fb.c.z; e[18] = fb.r.x; e[19] = fb.r.y; e[20] = fb.r.z;
    std::memcpy(&e[21], &tags[i], 4);
    if (b.constructor[i].kind == RigidBodyVec::KIND_COMPOUND)
      for (size_t pk = 0; pk < b.parts_local[i].size(); ++pk) put_part(e + 24 + 8 * pk, b.parts_local[i][pk]);
  }
  // ids ascending, all owned; the remaining bodies keep their relative order (ghosts are dropped first)
  void remove_bodies(const uint32_t* ids, size_t m) {
    drop_ghosts();
    if (m == 0) return;
    RigidBodyVec& b = bodies;
    size_t k = 0, out = 0;
    for (size_t i = 0; i < n_owned; ++i) {
      if (k < m && ids[k] == i) { bvh.remove(bvh_ids[i]); ++k; continue; }
      if (out != i) {
        b.x[out] = b.x[i]; b.q[out] = b.q[i]; b.v[out] = b.v[i]; b.omega[out] = b.omega[i]; b.force[out] = b.force[i];
        b.torque[out] = b.torque[i]; b.restitution[out] = b.restitution[i]; b.friction[out] = b.friction[i]; b.inv_mass[out] = b.inv_mass[i];
        b.inv_moment_body[out] = b.inv_moment_body[i]; b.inv_moment[out] = b.inv_moment[i];
        b.constructor[out] = b.constructor[i]; b.collider[out] = b.collider[i];
        b.parts_local[out] = b.parts_local[i]; b.parts[out] = b.parts[i];
        bvh_ids[out] = bvh_ids[i];
        tags[out] = tags[i];
        bvh.pool[bvh_ids[out]].leaf = out;
      }
      ++out;
    }
    n_owned = out;
    bvh_ids.resize(out);  // the tail holds stale copies of moved entries, not ghosts
    drop_ghosts();        // truncates every body array to n_owned
  }
  void import_migrant(const float* o) {
    drop_ghosts();
    RigidBodyVec& b = bodies;
    const float* e = o + kGhostFloats;
    Component k;
    uint32_t tag;
    std::memcpy(&tag, &o[16], 4);
    if (tag == 0u) k = component(Sphere{v3(o[17], o[18], o[19]), o[23]});
    else k = component(Capsule{v3(o[17], o[18], o[19]), v3(o[20], o[21], o[22]), o[23]});
    const V3 delta = v3(o[13], o[14], o[15]);
    b.x.push_back(v3(o[0], o[1], o[2]));
    b.q.push_back(Quat{o[3], v3(o[4], o[5], o[6])});
    b.v.push_back(v3(o[7], o[8], o[9]));
    b.omega.push_back(v3(o[10], o[11], o[12]));
    b.force.push_back(v3(e[0], e[1], e[2])); b.torque.push_back(v3(0, 0, 0));
    b.restitution.push_back(o[34]); b.friction.push_back(o[35]); b.inv_mass.push_back(o[24]);
    b.inv_moment.push_back(m3_new(o[25], o[26], o[27], o[28], o[29], o[30], o[31], o[32], o[33]));
    b.inv_moment_body.push_back(m3_new(e[3], e[4], e[5], e[6], e[7], e[8], e[9], e[10], e[11]));
    uint32_t kind;
    std::memcpy(&kind, &e[12], 4);
    b.constructor.push_back(ComponentConstructor{(int)kind, e[13], e[14]});
    b.collider.push_back(sweep(k, delta));
    b.sync_parts();
    size_t id = b.len() - 1;
    uint32_t n_parts;
    std::memcpy(&n_parts, &o[36], 4);
    for (uint32_t pk = 0; pk < n_parts; ++pk) {
      b.parts[id].push_back(sweep(get_part(o + 40 + 8 * pk), delta));
      b.parts_local[id].push_back(get_part(e + 24 + 8 * pk));
    }
    bvh_ids.push_back(bvh.insert(AABB{v3(e[15], e[16], e[17]), v3(e[18], e[19], e[20])}, id));
    n_owned = b.len();
    uint32_t tg;
    std::memcpy(&tg, &e[21], 4);
    tags.push_back(tg);
  }

  // world.rs:228-231
  void begin_tick(float dt) {
    using clk = std::chrono::steady_clock;
    drop_ghosts();
    solver = Solver();
    stats = StepStats();
    auto t0 = clk::now();
    bodies.complete_motion();
    bodies.integrate(dt);
    stats.t_integrate = std::chrono::duration<double>(clk::now() - t0).count();
  }
  // world.rs:227-294 up to (not including) solver.solve
  void build_constraints(float dt) {
    begin_tick(dt);
    collide(dt);
  }
  // world.rs:233-291
  void collide(float dt) {
    using clk = std::chrono::steady_clock;
    auto t1 = clk::now();
    std::vector<size_t> hits;
    const size_t n = bodies.len();
    for (size_t i = 0; i < n; ++i) {
      // the body's centre and motion (for a body of several parts: its centre of mass; `collider` is then a carrier)
      const Moving<Component> collider = bodies.collider[i];
      const V3 ci = center(collider.shape), vi = collider.vel;
      const size_t np_i = bodies.n_parts(i);
      AABB b = body_bounds(i);
      if (i < n_owned) {
        if (!aabb_contains(bvh[bvh_ids[i]], b)) {
          bvh.remove(bvh_ids[i]);
          bvh_ids[i] = bvh.insert(b + fat_margin, i);
          stats.n_refits++;
        }
        // Mesh::contacts (mesh.rs:115-139) + LocalContacts (collision.rs:1490-1506): faces in DFS order, per face the
        // body's parts in order; every contact is its own constraint (world.rs:243-251)
        terrain.for_faces(b, [&](const Triangle& tri) {
          for (size_t pa = 0; pa < np_i; ++pa) {
            mgfo::contacts(bodies.part(i, pa), tri, [&](const Contact& k) {
              const Contact c{k.b, k.a, -k.n, k.t};  // the mesh-side view (mesh.rs:131-134)
              const LocalContact lc{c.b + -(ci + vi * c.t), c.a + -terrain.center(), neg(c)};
              solver.add_constraint(ContactConstraint::make(bodies, dynamic_ref(i), static_ref(terrain.center(), 0.0f),
                                                            manifold_from(lc), dt, params));
              stats.n_terrain_constraints++;
            });
          }
        });
        for (const Compound& ob : obstacles) {
          const V3 oc = ob.disp;  // Shape::center for Compound
          for (size_t pa = 0; pa < np_i; ++pa) {
            const Moving<Component> part = bodies.part(i, pa);
            auto emit = [&](const Contact& c) {  // c: the obstacle-side view (a on the obstacle, b on the body), as Mesh::contacts hands it out
              const LocalContact lc{c.b + -(ci + vi * c.t), c.a + -oc, neg(c)};
              solver.add_constraint(ContactConstraint::make(bodies, dynamic_ref(i), static_ref(oc, 0.0f), manifold_from(lc), dt, params));
              stats.n_terrain_constraints++;
            };
            if (part.shape.kind == COMP_SPHERE) ob.contacts(sweep(part.shape.s, part.vel), emit);
            else ob.contacts(sweep(part.shape.c, part.vel), emit);
          }
        }
      }
      if (i == 0) continue;
      auto on_hit = [&](size_t j) {
        ContactPruner pruner;
        // LocalContacts (compound.rs:192-207) over every pair of parts, local points relative to the bodies' centres
        const V3 cj = center(bodies.collider[j].shape), vj = bodies.collider[j].vel;
        for (size_t pa = 0; pa < np_i; ++pa)
          for (size_t pb = 0; pb < bodies.n_parts(j); ++pb)
            mgfo::contacts(bodies.part(i, pa), bodies.part(j, pb), [&](const Contact& c) {
              pruner.push(LocalContact{c.a + -(ci + vi * c.t), c.b + -(cj + vj * c.t), c});
            });
        Manifold manifold = manifold_from(pruner);
        if (manifold.len() == 0) return;
        solver.add_constraint(ContactConstraint::make(bodies, dynamic_ref(i), dynamic_ref(j), manifold, dt, params));
      };
      if (order_mode == ORDER_DEMO) {
        bvh.query(b, [&](size_t j) {
          if (j >= i || j >= n_owned) return;
          stats.n_pair_candidates++;
          on_hit(j);
        });
      } else {
        hits.clear();
        bvh.query(b, [&](size_t j) { if (j < i && j < n_owned) hits.push_back(j); });
        std::sort(hits.begin(), hits.end());
        stats.n_pair_candidates += hits.size();
        for (size_t j : hits) on_hit(j);
      }
    }
    auto t2 = clk::now();
    stats.n_constraints = 0;  // counted per contact (= per constraint on the reference's path, where every manifold has one)
    for (const ContactConstraint& c : solver.constraints) stats.n_constraints += c.states.size();
    stats.t_collide = std::chrono::duration<double>(t2 - t1).count();
  }

  void step(float dt, size_t iters) {
    using clk = std::chrono::steady_clock;
    build_constraints(dt);
    auto t2 = clk::now();
    solver.solve(bodies, iters);  // world.rs:293 (the demo passes 20)
    auto t3 = clk::now();
    stats.t_solve = std::chrono::duration<double>(t3 - t2).count();
  }

  // Depth of the order-preserving dependency graph of `iters` solver iterations over the current
  // constraint list (level = 1 + max level of the previous constraint touching either body, carried
  // across iterations).  iters = 1 gives the per-iteration depth.  Analysis only; not a reference function.
  uint32_t constraint_depth(uint32_t iters = 1) const {
    std::vector<uint32_t> last(bodies.len(), 0);
    uint32_t depth = 0;
    for (uint32_t it = 0; it < iters; ++it)
      for (const ContactConstraint& c : solver.constraints) {
        uint32_t la = c.obj_a.is_static ? 0 : last[c.obj_a.index];
        uint32_t lb = c.obj_b.is_static ? 0 : last[c.obj_b.index];
        uint32_t l = 1 + std::max(la, lb);
        if (!c.obj_a.is_static) last[c.obj_a.index] = l;
        if (!c.obj_b.is_static) last[c.obj_b.index] = l;
        depth = std::max(depth, l);
      }
    return depth;
  }
};

}  // namespace mgfo
