// ORACLE — TEST INFRASTRUCTURE ONLY (see mgf_math.hpp header).
// CPU restatement of
//   src/physics.rs : Inertia :26-93, RigidBodyInfo/Velocity :124-137,
//                    RigidBodyVec :141-270, ConstrainedSet impl :272-315
//   src/manifold.rs: ContactPruner :42-108, Manifold :112-164
//   src/solver.rs  : Solver :53-79, ContactConstraint::new :101-191,
//                    ContactConstraint::solve :203-252, ContactState :256-262,
//                    params :265-279, clamp :281-289
// PARITY STATUS: the reference holds NO test for any of these apart
// from the sphere tensor (physics.rs:321-335): "parity unpinned" — the authority
// is the source text; cross-checked by closed-form cases in tests/test_oracle_solver.py.
#pragma once
#include <vector>

#include "mgf_collision.hpp"

namespace mgfo {

// smallvec 0.6 `SmallVec<[T; N]>`: inline storage for N items, heap spill beyond.
// Containers only, no arithmetic (reference Cargo.toml:21).
template <class T, int N>
struct SmallVec {
  T inl[N];
  std::vector<T> spill;
  size_t n = 0;
  size_t size() const { return n; }
  bool empty() const { return n == 0; }
  void clear() { n = 0; spill.clear(); }
  void push_back(const T& t) {
    if (n < (size_t)N) inl[n] = t;
    else spill.push_back(t);
    ++n;
  }
  T& operator[](size_t i) { return i < (size_t)N ? inl[i] : spill[i - N]; }
  const T& operator[](size_t i) const { return i < (size_t)N ? inl[i] : spill[i - N]; }
};

// Inertia physics.rs:30-93
static inline M3 tensor(const Sphere& s, float m) {
  float i = 0.4f * m * s.r * s.r;
  M3 im = m3_new(i, 0, 0, 0, i, 0, 0, 0, i);
  V3 disp = s.c;
  M3 outer = m3_from_cols(disp * disp.x, disp * disp.y, disp * disp.z);
  return im + m * (m3_one() * dot(disp, disp) - outer);
}
static inline M3 tensor(const Capsule& c, float m) {
  float h = magnitude(c.d);
  float r = c.r;
  float mh = m * 2.0f * r / (4.0f * r + 3.0f * h);
  float mc = m * h / (4.0f / 3.0f * r + h);
  float ic_x = 1.0f / 12.0f * mc * (3.0f * r * r + h * h);
  float ic_y = 0.5f * mc * r * r;
  float ic_z = ic_x;
  float is_x = mh * (3.0f * r + 2.0f * h) / 4.0f * h;
  float is_y = 4.0f / 5.0f * mh * r * r;
  float is_z = is_x;
  float i_x = ic_x + is_x, i_y = ic_y + is_y, i_z = ic_z + is_z;
  V3 dst = c.d;
  V3 src = v3(0.0f, 1.0f, 0.0f) * h;
  M3 rot = m3_from_quat(quat_from_arc(src, dst));
  M3 i = rot * m3_new(i_x, 0, 0, 0, i_y, 0, 0, 0, i_z) * transpose(rot);
  V3 disp = center(c);
  M3 outer = m3_from_cols(disp * disp.x, disp * disp.y, disp * disp.z);
  return i + m * (m3_one() * dot(disp, disp) - outer);
}
static inline M3 tensor(const Component& k, float m) { return k.kind == COMP_SPHERE ? tensor(k.s, m) : tensor(k.c, m); }

struct Velocity { V3 linear; V3 angular; };
struct RigidBodyInfo { V3 x; float restitution; float friction; float inv_mass; M3 inv_moment; };

// RigidBodyRef physics.rs:158-177
struct RigidBodyRef {
  bool is_static;
  size_t index;
  V3 center;
  float friction;
};
static inline RigidBodyRef dynamic_ref(size_t i) { return RigidBodyRef{false, i, v3(0, 0, 0), 0.0f}; }
static inline RigidBodyRef static_ref(V3 c, float friction) { return RigidBodyRef{true, 0, c, friction}; }

// RigidBodyVec physics.rs:141-270
struct RigidBodyVec {
  std::vector<V3> x;
  std::vector<Quat> q;
  std::vector<V3> v, omega, force, torque;
  std::vector<float> restitution, friction, inv_mass;
  std::vector<M3> inv_moment_body, inv_moment;
  std::vector<ComponentConstructor> constructor;
  std::vector<Moving<Component>> collider;
  // Bodies made of several components (BASELINE config 5).  NOT in the reference: physics.rs:200 takes one Component.
  // Definition used by this build (and mirrored by the HIP path): the components are fixed in the body frame, relative
  // to the centre of mass; mass = sum, tensor = sum of the components' tensors about the centre of mass (the reference's
  // own Inertia, physics.rs:30-93); the swept parts are rebuilt from (x, q) every tick like the reference rebuilds a
  // single collider (physics.rs:243-251).  `collider[i]` of such a body is a radius-0 sphere at x: it carries the
  // centre and the motion (complete_motion, ConstrainedSet::get) and never collides.  Empty for ordinary bodies.
  std::vector<std::vector<Component>> parts_local;
  std::vector<std::vector<Moving<Component>>> parts;
  static constexpr int KIND_COMPOUND = 2;

  size_t len() const { return x.size(); }
  void sync_parts() { parts_local.resize(x.size()); parts.resize(x.size()); }
  size_t n_parts(size_t i) const { return i < parts_local.size() && !parts_local[i].empty() ? parts_local[i].size() : 1; }
  const Moving<Component>& part(size_t i, size_t k) const {
    return i < parts_local.size() && !parts_local[i].empty() ? parts[i][k] : collider[i];
  }
  static Component part_world(const Component& l, V3 px, Quat pq) {
    if (l.kind == COMP_SPHERE) return component(Sphere{px + rotate_vector(pq, l.s.c), l.s.r});
    return component(Capsule{px + rotate_vector(pq, l.c.a), rotate_vector(pq, l.c.d), l.c.r});
  }
  bool add_compound_body(const Component* comps, const float* masses, size_t k, float rest, float fric, V3 world_force, size_t* id_out) {
    if (k == 0) return false;
    size_t id = x.size();
    float total = 0.0f;
    V3 acc = v3(0, 0, 0);
    for (size_t c = 0; c < k; ++c) { total += masses[c]; acc = acc + center(comps[c]) * masses[c]; }
    V3 com = acc / total;
    M3 t = m3_zero();
    for (size_t c = 0; c < k; ++c) t = t + tensor(comps[c] - com, masses[c]);
    M3 inv;
    if (!invert(t, &inv)) return false;
    x.push_back(com);
    q.push_back(quat_one());
    v.push_back(v3(0, 0, 0));
    omega.push_back(v3(0, 0, 0));
    force.push_back(world_force * total);
    torque.push_back(v3(0, 0, 0));
    restitution.push_back(rest);
    friction.push_back(fric);
    inv_mass.push_back(1.0f / total);
    inv_moment_body.push_back(inv);
    inv_moment.push_back(inv);
    constructor.push_back(ComponentConstructor{KIND_COMPOUND, 0.0f, 0.0f});
    collider.push_back(sweep(component(Sphere{com, 0.0f}), v3(0, 0, 0)));
    sync_parts();
    for (size_t c = 0; c < k; ++c) {
      parts_local[id].push_back(comps[c] - com);
      parts[id].push_back(sweep(comps[c], v3(0, 0, 0)));
    }
    if (id_out) *id_out = id;
    return true;
  }

  // physics.rs:200-218.  Returns false where the reference's `.unwrap()` panics.
  bool add_body(const Component& col, float mass, float rest, float fric, V3 world_force, size_t* id_out) {
    size_t id = x.size();
    V3 px; Quat pq; ComponentConstructor cons;
    deconstruct(col, &px, &pq, &cons);
    M3 inv;
    if (!invert(tensor(col - px, mass), &inv)) return false;
    x.push_back(px);
    q.push_back(pq);
    v.push_back(v3(0, 0, 0));
    omega.push_back(v3(0, 0, 0));
    force.push_back(world_force * mass);
    torque.push_back(v3(0, 0, 0));
    restitution.push_back(rest);
    friction.push_back(fric);
    inv_mass.push_back(1.0f / mass);
    inv_moment_body.push_back(inv);
    inv_moment.push_back(inv);
    constructor.push_back(cons);
    collider.push_back(sweep(col, v3(0, 0, 0)));
    if (id_out) *id_out = id;
    return true;
  }

  // physics.rs:222-253 — five separate loops, kept separate.
  void integrate(float dt) {
    size_t n = x.size();
    for (size_t i = 0; i < n; ++i)
      q[i] = normalize(q[i] + quat_from_sv(0.0f, omega[i] * dt) * 0.5f * q[i]);
    for (size_t i = 0; i < n; ++i) {
      M3 r = m3_from_quat(q[i]);
      inv_moment[i] = r * inv_moment_body[i] * transpose(r);
    }
    for (size_t i = 0; i < n; ++i) v[i] += force[i] * inv_mass[i] * dt;
    for (size_t i = 0; i < n; ++i) omega[i] += inv_moment[i] * torque[i] * dt;
    for (size_t i = 0; i < n; ++i) {
      if (constructor[i].kind == KIND_COMPOUND) {
        collider[i] = sweep(component(Sphere{x[i], 0.0f}), v[i] * dt);
        for (size_t k = 0; k < parts_local[i].size(); ++k) parts[i][k] = sweep(part_world(parts_local[i][k], x[i], q[i]), v[i] * dt);
      } else {
        collider[i] = sweep(construct(constructor[i], x[i], q[i]), v[i] * dt);
      }
    }
  }

  // physics.rs:262-269
  void complete_motion() {
    for (size_t i = 0; i < x.size(); ++i) x[i] += collider[i].vel;
  }

  // ConstrainedSet physics.rs:272-315
  void get(const RigidBodyRef& r, Velocity* vel, RigidBodyInfo* info) const {
    if (!r.is_static) {
      size_t i = r.index;
      *vel = Velocity{v[i], omega[i]};
      *info = RigidBodyInfo{x[i] + collider[i].vel, restitution[i], friction[i], inv_mass[i], inv_moment[i]};
    } else {
      *vel = Velocity{v3(0, 0, 0), v3(0, 0, 0)};
      *info = RigidBodyInfo{r.center, 0.0f, r.friction, 0.0f, m3_zero()};
    }
  }
  void set(const RigidBodyRef& r, const Velocity& vel) {
    if (!r.is_static) { v[r.index] = vel.linear; omega[r.index] = vel.angular; }
  }
};

// ---------------------------------------------------------------------------
// manifold.rs
// ---------------------------------------------------------------------------
static constexpr float PERSISTENT_THRESHOLD_SQ = 0.5f;  // manifold.rs:38

struct ContactPruner {  // manifold.rs:42-108
  float min_col_time = F32_INF;
  SmallVec<LocalContact, 4> contacts;
  void push(const LocalContact& nc) {
    if (nc.global.t < min_col_time - COLLISION_EPSILON) {
      contacts.clear();
      contacts.push_back(nc);
      min_col_time = nc.global.t;
      return;
    } else if (nc.global.t > min_col_time + COLLISION_EPSILON) {
      return;
    }
    for (size_t oi = 0; oi < contacts.size(); ++oi) {
      LocalContact& old = contacts[oi];
      V3 ra = nc.global.a - old.global.a;
      V3 rb = nc.global.b - old.global.b;
      if (magnitude2(ra) <= PERSISTENT_THRESHOLD_SQ || magnitude2(rb) <= PERSISTENT_THRESHOLD_SQ) {
        float prev_dist = magnitude2(old.local_a) + magnitude2(old.local_b);
        float new_dist = magnitude2(nc.local_a) + magnitude2(nc.local_b);
        if (prev_dist < new_dist) old = nc;
        return;
      }
    }
    contacts.push_back(nc);
  }
  void clear() { min_col_time = F32_INF; contacts.clear(); }
};

struct ContactPair { V3 a, b; };
struct Manifold {  // manifold.rs:112-118 (SmallVec<[..;4]> -> inline 4 + count; spill is never hit on this path)
  float time;
  V3 normal;
  V3 tangent_vector[2];
  SmallVec<ContactPair, 4> contacts;
  size_t len() const { return contacts.size(); }
};
static inline Manifold manifold_from(const LocalContact& lc) {  // manifold.rs:120-128
  Manifold m;
  m.time = lc.global.t;
  m.normal = lc.global.n;
  compute_basis(lc.global.n, m.tangent_vector);
  m.contacts.push_back(ContactPair{lc.local_a, lc.local_b});
  return m;
}
static inline Manifold manifold_from(const ContactPruner& pr) {  // manifold.rs:131-148
  Manifold m;
  V3 sum = v3(0, 0, 0);
  for (size_t k = 0; k < pr.contacts.size(); ++k) {
    const LocalContact& lc = pr.contacts[k];
    m.contacts.push_back(ContactPair{lc.local_a, lc.local_b});
    sum = sum + lc.global.n;
  }
  V3 avg_normal = sum / (float)pr.contacts.size();
  m.time = pr.min_col_time;
  m.normal = avg_normal;
  compute_basis(avg_normal, m.tangent_vector);
  return m;
}

// ---------------------------------------------------------------------------
// solver.rs
// ---------------------------------------------------------------------------
struct ContactConstraintParams { float penetration_slop = 0.05f; float baumgarte = 0.2f; };  // solver.rs:276-279

static inline float solver_clamp(float n, float mn, float mx) {  // solver.rs:281-289
  if (n < mn) return mn;
  if (n > mx) return mx;
  return n;
}

struct ContactState {  // solver.rs:256-262
  float bias, normal_mass, normal_impulse;
  float tangent_mass[2];
  float tangent_impulse[2];
};

struct ContactConstraint {  // solver.rs:82-93
  RigidBodyRef obj_a, obj_b;
  Manifold manifold;
  float friction;
  SmallVec<ContactState, 4> states;

  // solver.rs:101-191
  template <class Set>
  static ContactConstraint make(const Set& pool, const RigidBodyRef& obj_a, const RigidBodyRef& obj_b,
                                const Manifold& manifold, float dt,
                                const ContactConstraintParams& P = ContactConstraintParams()) {
    Velocity va_, vb_;
    RigidBodyInfo ia, ib;
    pool.get(obj_a, &va_, &ia);
    pool.get(obj_b, &vb_, &ib);
    V3 va = va_.linear, oa = va_.angular, vb = vb_.linear, ob = vb_.angular;
    V3 xa = ia.x, xb = ib.x;
    float inv_mass_a = ia.inv_mass, inv_mass_b = ib.inv_mass;
    const M3& inv_moment_a = ia.inv_moment;
    const M3& inv_moment_b = ib.inv_moment;
    float restitution = fmaxf_rs(ia.restitution, ib.restitution);
    float friction = std::sqrt(ia.friction * ib.friction);
    ContactConstraint cc;
    cc.obj_a = obj_a; cc.obj_b = obj_b; cc.manifold = manifold; cc.friction = friction;
    for (size_t k0 = 0; k0 < manifold.contacts.size(); ++k0) {
      const ContactPair& cp = manifold.contacts[k0];
      V3 ra = cp.a, rb = cp.b;
      V3 ca = ra + xa, cb = rb + xb;
      V3 ra_cn = cross(ra, manifold.normal);
      V3 rb_cn = cross(rb, manifold.normal);
      float pen = dot(cb - ca, manifold.normal);
      V3 dv = vb + cross(ob, rb) - va - cross(oa, ra);
      float rel_v = dot(dv, manifold.normal);
      float bias = -P.baumgarte / dt * (pen > 0.0f ? 0.0f : pen + P.penetration_slop) +
                   (rel_v < -1.0f ? -restitution * rel_v : 0.0f);
      float normal_mass = 1.0f / (inv_mass_a + dot(ra_cn, inv_moment_a * ra_cn) + inv_mass_b +
                                  dot(rb_cn, inv_moment_b * rb_cn));
      ContactState st;
      for (int k = 0; k < 2; ++k) {
        V3 ra_ct = cross(ra, manifold.tangent_vector[k]);
        V3 rb_ct = cross(rb, manifold.tangent_vector[k]);
        st.tangent_mass[k] = 1.0f / (inv_mass_a + dot(ra_ct, inv_moment_a * ra_ct) + inv_mass_b +
                                     dot(rb_ct, inv_moment_b * rb_ct));
      }
      st.bias = bias;
      st.normal_mass = normal_mass;
      st.normal_impulse = 0.0f;
      st.tangent_impulse[0] = st.tangent_impulse[1] = 0.0f;
      cc.states.push_back(st);
    }
    return cc;
  }

  // solver.rs:203-252 — reproduces the reference's quirks (SURVEY Appendix A 1,2).
  template <class Set>
  void solve(Set& pool) {
    Velocity va_, vb_;
    RigidBodyInfo ia, ib;
    pool.get(obj_a, &va_, &ia);
    pool.get(obj_b, &vb_, &ib);
    V3 va = va_.linear, oa = va_.angular, vb = vb_.linear, ob = vb_.angular;
    float inv_mass_a = ia.inv_mass, inv_mass_b = ib.inv_mass;
    const M3& inv_moment_a = ia.inv_moment;
    const M3& inv_moment_b = ib.inv_moment;
    for (size_t i = 0; i < states.size(); ++i) {
      ContactState& cs = states[i];
      V3 ra = manifold.contacts[i].a, rb = manifold.contacts[i].b;
      V3 dv = vb + cross(ob, rb) - va - cross(oa, ra);
      for (int k = 0; k < 2; ++k) {
        float lambda = -dot(dv, manifold.tangent_vector[k]) * cs.tangent_mass[k];
        float max_lambda = friction * cs.normal_impulse;
        float prev_impulse = cs.tangent_impulse[k];
        cs.tangent_impulse[k] = solver_clamp(-max_lambda, max_lambda, prev_impulse + lambda);
        V3 impulse = manifold.tangent_vector[k] * lambda;
        va -= impulse * inv_mass_a;
        oa -= inv_moment_a * cross(ra, impulse);
        vb += impulse * inv_mass_b;
        ob += inv_moment_b * cross(rb, impulse);
      }
      V3 dv2 = vb + cross(ob, rb) - va - cross(oa, ra);
      float vn = dot(dv2, manifold.normal);
      float lambda = cs.normal_mass * (-vn + cs.bias);
      float prev_impulse = cs.normal_impulse;
      cs.normal_impulse = fmaxf_rs(prev_impulse + lambda, 0.0f);
      lambda = cs.normal_impulse - prev_impulse;
      V3 impulse = manifold.normal * lambda;
      va -= impulse * inv_mass_a;
      oa -= inv_moment_a * cross(ra, impulse);
      vb += impulse * inv_mass_b;
      ob += inv_moment_b * cross(rb, impulse);
    }
    pool.set(obj_a, Velocity{va, oa});
    pool.set(obj_b, Velocity{vb, ob});
  }
};

struct Solver {  // solver.rs:53-79
  std::vector<ContactConstraint> constraints;
  void add_constraint(ContactConstraint&& c) { constraints.push_back(std::move(c)); }
  template <class Set>
  void solve(Set& cs, size_t iters) {
    for (size_t it = 0; it < iters; ++it)
      for (ContactConstraint& c : constraints) c.solve(cs);
  }
  size_t len() const { return constraints.size(); }
};

}  // namespace mgfo
