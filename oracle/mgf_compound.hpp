// ORACLE — TEST INFRASTRUCTURE ONLY.  CPU restatement of mgf::Compound (compound.rs:230-352): an aggregate of
// Spheres and Capsules with a displacement, a rotation and an internal BVH<AABB, Component>, together with
// Contacts<RHS> for Compound (:334-352), Intersects<Compound> for a particle (:309-332), BoundedBy<AABB> (:274-278)
// and the Volumetric helpers it calls (geom.rs:927-1015).  Pinned by the reference's own test (compound.rs:360-389):
// tests/golden/reference_known_answers.json "compound".
#pragma once
#include <vector>

#include "mgf_bvh.hpp"
#include "mgf_collision.hpp"

namespace mgfo {

// Rotation3::rotate_point(p) = rotate_vector on the point's coordinates (cgmath)
static inline V3 rotate_point(Quat q, V3 p) { return rotate_vector(q, p); }

// Volumetric::rotate for AABB geom.rs:940-985: bounds of the eight rotated corners, min/max nested right to left
static inline AABB aabb_rotate(const AABB& b, Quat rot) {
  V3 vx = rotate_vector(rot, v3(b.r.x, 0.0f, 0.0f)), vy = rotate_vector(rot, v3(0.0f, b.r.y, 0.0f)), vz = rotate_vector(rot, v3(0.0f, 0.0f, b.r.z));
  V3 p[8] = {b.c + (vx + vy + vz), b.c + (vx + vy - vz), b.c + (vx - vy + vz), b.c + (vx - vy - vz),
             b.c + (-vx + vy + vz), b.c + (-vx + vy - vz), b.c + (-vx - vy + vz), b.c + (-vx - vy - vz)};
  V3 lower, upper;
  for (int k = 0; k < 3; ++k) {
    float lo = idx(p[7], k), hi = idx(p[7], k);
    for (int e = 6; e >= 0; --e) { lo = fminf_rs(idx(p[e], k), lo); hi = fmaxf_rs(idx(p[e], k), hi); }  // p1.min(p2.min(...p8))
    if (k == 0) { lower.x = lo; upper.x = hi; } else if (k == 1) { lower.y = lo; upper.y = hi; } else { lower.z = lo; upper.z = hi; }
  }
  return AABB{(upper + lower) / 2.0f, (upper - lower) / 2.0f};
}
// Volumetric::rotate for Sphere (no-op) / Capsule (about its centre) geom.rs:999-1015, for Component compound.rs:54-61
static inline Component comp_rotate(Component k, Quat r) {
  if (k.kind == COMP_CAPSULE) {
    V3 ctr = center(k.c);
    k.c = Capsule{ctr + rotate_vector(r, k.c.a - ctr), rotate_vector(r, k.c.d), k.c.r};
  }
  return k;
}
// Volumetric::rotate_about geom.rs:932-937
static inline Component comp_rotate_about(Component k, Quat r, V3 p) {
  V3 ctr = center(k);
  set_pos(k, p + rotate_vector(r, ctr - p));
  return comp_rotate(k, r);
}
// Intersects<Component> compound.rs:150-157
static inline bool ray_component(const Ray& ray, const Component& k, Intersection* out, float dt) {
  return k.kind == COMP_SPHERE ? ray_sphere(ray, k.s, out, dt) : ray_capsule(ray, k.c, out, dt);
}
// Contacts<Moving<Component>> for Sphere / Capsule / Rectangle (impl_component_collision! compound.rs:159-178)
template <class Recv, class F>
static inline bool contacts(const Recv& self, const Moving<Component>& rhs, F&& cb) {
  if (rhs.shape.kind == COMP_SPHERE) return contacts(self, sweep(rhs.shape.s, rhs.vel), cb);
  return contacts(self, sweep(rhs.shape.c, rhs.vel), cb);
}

struct Compound {
  V3 disp{0.0f, 0.0f, 0.0f};
  Quat rot = quat_one();
  std::vector<size_t> shapes;
  BVH<Component> bvh;

  explicit Compound(const std::vector<Component>& comps) {  // compound.rs:244-257
    for (const Component& k : comps) shapes.push_back(bvh.insert(mgfo::bounds(k), k));
  }
  AABB bounds() const { return aabb_rotate(bvh[bvh.get_root()], rot) + disp; }  // :274-278

  // Contacts<RHS> for Compound :334-352 with RHS = Moving<Shape>; `rhs.contacts(&shape)` is the :1368 wrapper
  template <class Shape, class F>
  bool contacts(const Moving<Shape>& rhs, F&& cb) const {
    Quat conj_rot = conjugate(rot);
    AABB rhs_bounds = aabb_rotate(mgfo::bounds(rhs), conj_rot);
    V3 rhs_center = rhs_bounds.c;
    V3 bounds_disp = rotate_point(conj_rot, rhs_center + -disp) + disp;
    rhs_bounds.c = bounds_disp;  // set_pos
    bool collided = false;
    bvh.query(rhs_bounds, [&](const Component& comp) {
      Component shape = comp_rotate_about(comp, rot, v3(0.0f, 0.0f, 0.0f)) + disp;
      Moving<Component> rhs_moving = sweep(shape, -rhs.vel);
      mgfo::contacts(rhs.shape, rhs_moving, [&](const Contact& c) {
        V3 d = rhs.vel * c.t;
        collided = true;
        cb(neg(Contact{c.a + d, c.b + d, c.n, c.t}));
      });
    });
    return collided;
  }

  // Intersects<Compound> for a particle :309-332 (the BVH is traced with a Ray whatever the particle's DT)
  bool intersection(const Ray& part, float dt, Intersection* out) const {
    Quat conj_rot = conjugate(rot);
    Ray r{rotate_point(conj_rot, part.p + -disp) + disp, rotate_vector(conj_rot, part.d)};
    bool have = false;
    Intersection result{};
    bvh.raytrace(
        [&](const AABB& b) { Intersection i; bool h = ray_aabb(r, b, &i, F32_INF); return std::make_pair(h, i); },
        [&](const Component& comp, const Intersection& bi) {
          if (bi.t > dt) return;
          Component shape = comp_rotate(comp, rot) + disp;
          Intersection inter;
          if (ray_component(part, shape, &inter, dt)) {
            if (have && inter.t > result.t) return;
            result = inter; have = true;
          }
        });
    if (have) *out = result;
    return have;
  }
};

}  // namespace mgfo
