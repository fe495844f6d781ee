// ORACLE — TEST INFRASTRUCTURE ONLY (see mgf_math.hpp header).
// CPU restatement of the reference's narrowphase, src/collision.rs:
//   Contains<Point3> for Triangle :85-100, Rectangle :102-112
//   Ray ∩ Sphere :249-273, Ray ∩ Capsule :275-359
//   Contact :431-456, commute_contacts! :484-494
//   Plane–Moving<Sphere> :521-553, Plane–Moving<Capsule> :555-605
//   Polygon–Moving<Sphere> :610-659, seg_2d_intersect :667-688,
//   Polygon–Moving<Capsule> :693-1086
//   Sphere–Moving<Sphere> :1089-1141, Capsule–Moving<Sphere> :1145-1203,
//   Capsule–Moving<Capsule> :1205-1356
//   Moving wrappers :1368-1401, LocalContact :1410-1432
// and of src/compound.rs: Component :33-52, dispatch :159-207,
// ComponentConstructor :211-228, and src/bitset.rs:40-55 (u64 bit set).
// Pinned by the reference's own known-answer tests collision.rs:1543-2268
// (transcribed in tests/golden/collision_vectors.py).
#pragma once
#include "mgf_geom.hpp"

namespace mgfo {

template <class T>
struct Moving {
  T shape;
  V3 vel;
};
template <class T>
static inline Moving<T> sweep(const T& s, V3 v) { return Moving<T>{s, v}; }

struct Intersection { V3 p; float t; };
struct Contact { V3 a; V3 b; V3 n; float t; };
static inline Contact neg(const Contact& c) { return Contact{c.b, c.a, -c.n, c.t}; }  // collision.rs:444-456
struct LocalContact { V3 local_a; V3 local_b; Contact global; };

// Contains<Point3> for Triangle collision.rs:85-100
static inline bool contains(const Triangle& t, V3 p) {
  V3 v = p - t.a, ac = t.c - t.a, ab = t.b - t.a;
  float dot1 = dot(ac, ac), dot2 = dot(ac, ab), dot3 = dot(ac, v), dot4 = dot(ab, ab), dot5 = dot(ab, v);
  float invd = 1.0f / (dot1 * dot4 - dot2 * dot2);
  float u = (dot4 * dot3 - dot2 * dot5) * invd;
  float vv = (dot1 * dot5 - dot2 * dot3) * invd;
  return u >= 0.0f && vv >= 0.0f && (u + vv) < 1.0f;
}
// Contains<Point3> for Rectangle collision.rs:102-112
static inline bool contains(const Rectangle& r, V3 p) {
  V3 n = cross(r.u[0], r.u[1]);
  return relative_eq(dot(p, n), dot(n, r.c), COLLISION_EPSILON) && std::fabs(dot(p, r.u[0])) <= r.e[0] &&
         std::fabs(dot(p, r.u[1])) <= r.e[1];
}

// Ray ∩ Sphere collision.rs:249-273 (Ray::DT = inf so the `t > DT` test never fires)
static inline bool ray_sphere(const Ray& ray, const Sphere& s, Intersection* out, float dt = F32_INF) {  // dt = Particle::DT (Ray inf, Segment 1)
  V3 p = ray.p, d = ray.d;
  V3 m = p - s.c;
  float a = magnitude2(d), b = dot(m, d), c = magnitude2(m) - s.r * s.r;
  if (c > 0.0f && b > 0.0f) return false;
  float discr = b * b - a * c;
  if (discr < 0.0f) return false;
  float t = fmaxf_rs((-b - std::sqrt(discr)) / a, 0.0f);
  if (t > dt) return false;
  *out = Intersection{p + t * d, t};
  return true;
}

// Ray ∩ Capsule collision.rs:275-359
static inline bool ray_capsule(const Ray& ray, const Capsule& cap, Intersection* out, float dt = F32_INF) {
  V3 p = ray.p, d = ray.d;
  V3 m = p - cap.a;
  float md = dot(m, cap.d), nd = dot(d, cap.d), dd = dot(cap.d, cap.d);
  float nn = magnitude2(d), mn = dot(m, d);
  float a = dd * nn - nd * nd;
  float k = magnitude2(m) - cap.r * cap.r;
  if (std::fabs(a) < COLLISION_EPSILON) {
    float b, c;
    if (md < 0.0f) { b = mn; c = k; }
    else if (md > dd) {
      V3 m2 = p - (cap.a + cap.d);
      b = dot(m2, d); c = magnitude2(m2) - cap.r * cap.r;
    } else {
      return false;  // "Already colliding"
    }
    if (c > 0.0f && b > 0.0f) return false;
    float discr = b * b - nn * c;
    if (discr < 0.0f) return false;
    float t = fmaxf_rs((-b - std::sqrt(discr)) / nn, 0.0f);
    if (t > dt) return false;
    *out = Intersection{p + t * d, t};
    return true;
  }
  float c = dd * k - md * md;
  float b = dd * mn - nd * md;
  float discr = b * b - a * c;
  if (discr < 0.0f) return false;
  float t = (-b - std::sqrt(discr)) / a;
  if (t < 0.0f) return false;
  if (md + t * nd < 0.0f) {
    if (mn > 0.0f && k > 0.0f) return false;
    float discr2 = mn * mn - nn * k;
    if (discr2 < 0.0f) return false;
    t = fmaxf_rs((-mn - std::sqrt(discr2)) / nn, 0.0f);
  } else if (md + t * nd > dd) {
    V3 m2 = p - (cap.a + cap.d);
    float b2 = dot(m2, d);
    float c2 = magnitude2(m2) - cap.r * cap.r;
    if (c2 > 0.0f && b2 > 0.0f) return false;
    float discr2 = b2 * b2 - nn * c2;
    if (discr2 < 0.0f) return false;
    t = fmaxf_rs((-b2 - std::sqrt(discr2)) / nn, 0.0f);
  }
  if (t > dt) return false;
  *out = Intersection{p + t * d, t};
  return true;
}

// Intersects<Plane> collision.rs:169-184
static inline bool ray_plane(const Ray& ray, const Plane& pl, Intersection* out, float dt = F32_INF) {
  float denom = dot(pl.n, ray.d);
  if (denom == 0.0f) return false;
  float t = (pl.d - dot(pl.n, ray.p)) / denom;
  if (t <= 0.0f || t > dt) return false;
  *out = Intersection{ray.p + ray.d * t, t};
  return true;
}
// Intersects<Poly> collision.rs:186-200 (Triangle, Rectangle)
template <class Poly>
static inline bool ray_polygon(const Ray& ray, const Poly& poly, Intersection* out, float dt = F32_INF) {
  Intersection i;
  if (ray_plane(ray, to_plane(poly), &i, dt) && contains(poly, i.p)) { *out = i; return true; }
  return false;
}
// Intersects<AABB> collision.rs:202-236
static inline bool ray_aabb(const Ray& ray, const AABB& a, Intersection* out, float dt = F32_INF) {
  float t_min = 0.0f, t_max = F32_INF;
  for (int dim = 0; dim < 3; ++dim) {
    float pd = idx(ray.p, dim), dd = idx(ray.d, dim), ac = idx(a.c, dim), ar = idx(a.r, dim);
    if (std::fabs(dd) < COLLISION_EPSILON) {
      if (std::fabs(pd - ac) > ar) return false;
    } else {
      float ood = 1.0f / dd;
      float t1 = (ac - ar - pd) * ood;
      float t2 = (ac + ar - pd) * ood;
      if (t1 > t2) { t_min = fmaxf_rs(t_min, t2); t_max = fminf_rs(t_max, t1); }
      else { t_min = fmaxf_rs(t_min, t1); t_max = fminf_rs(t_max, t2); }
      if (t_min > t_max) return false;
    }
  }
  if (t_min > dt) return false;
  *out = Intersection{ray.p + ray.d * t_min, t_min};
  return true;
}

// ---------------------------------------------------------------------------
// Plane–Moving<Sphere> collision.rs:521-553
// ---------------------------------------------------------------------------
template <class F>
static inline bool contacts(const Plane& pl, const Moving<Sphere>& sphere, F&& cb) {
  const Sphere& s = sphere.shape;
  V3 v = sphere.vel;
  float dist = dot(pl.n, s.c) - pl.d;
  if (std::fabs(dist) <= s.r) {
    cb(Contact{s.c + -pl.n * dist, s.c + -pl.n * s.r, pl.n, 0.0f});
    return true;
  }
  float denom = dot(pl.n, v);
  if (denom * dist >= 0.0f) return false;
  float r = dist > 0.0f ? s.r : -s.r;
  float t = (r - dist) / denom;
  if (t <= 1.0f) {
    V3 q = s.c + t * v - r * pl.n;
    cb(Contact{q, q, pl.n, t});
    return true;
  }
  return false;
}

// Plane–Moving<Capsule> collision.rs:555-605
template <class F>
static inline bool contacts(const Plane& pl, const Moving<Capsule>& capsule, F&& cb) {
  const Capsule& c = capsule.shape;
  V3 v = capsule.vel;
  float denom = dot(pl.n, normalize(c.d));
  V3 ctr;
  if (std::fabs(denom) < COLLISION_EPSILON) {
    ctr = c.a + c.d * 0.5f;
  } else {
    float t = (pl.d - dot(pl.n, c.a)) / denom;
    if (t > 1.0f) ctr = c.a + c.d;
    else if (t < 0.0f) ctr = c.a;
    else {
      V3 q = c.a + c.d * t;
      float dist = dot(pl.n, c.a) - pl.d;
      V3 base = dist < 0.0f ? c.a : (c.a + c.d);
      cb(Contact{q, base + -pl.n * c.r, pl.n, 0.0f});
      return true;
    }
  }
  Moving<Sphere> ms = sweep(Sphere{ctr, c.r}, v);
  return contacts(pl, ms, cb);
}

// Contacts::last_contact collision.rs:477-481
template <class A, class B>
static inline bool last_contact(const A& a, const B& b, Contact* out) {
  bool any = false;
  contacts(a, b, [&](const Contact& c) { *out = c; any = true; });
  return any;
}

// ---------------------------------------------------------------------------
// Polygon–Moving<Sphere> collision.rs:610-659
// ---------------------------------------------------------------------------
template <class Poly, class F>
static inline bool poly_contacts_sphere(const Poly& poly, const Moving<Sphere>& sphere, F&& cb) {
  const Sphere& s = sphere.shape;
  V3 v = sphere.vel;
  bool collision = false;
  Plane p = to_plane(poly);
  contacts(p, sphere, [&](const Contact& contact) {
    if (contains(poly, contact.a)) {
      collision = true;
      cb(contact);
      return;
    }
    float first_t = F32_INF;
    V3 tri_p = v3(0, 0, 0);
    if (magnitude2(v) == 0.0f) return;
    Ray ray{s.c, v};
    for (int edge_i = 0; edge_i < num_vertices(poly); ++edge_i) {
      int a, b;
      edge(poly, edge_i, &a, &b);
      V3 v1 = vertex(poly, a), v2 = vertex(poly, b);
      Capsule c{v1, v2 - v1, s.r};
      Intersection i;
      if (ray_capsule(ray, c, &i)) {
        if (i.t <= 1.0f && i.t < first_t) {
          first_t = i.t;
          tri_p = seg_closest_point(Segment{v1, v2}, i.p);
        }
      }
    }
    if (first_t != F32_INF) {
      collision = true;
      cb(Contact{tri_p, tri_p, p.n, first_t});
    }
  });
  return collision;
}
template <class F>
static inline bool contacts(const Triangle& t, const Moving<Sphere>& s, F&& cb) { return poly_contacts_sphere(t, s, cb); }
template <class F>
static inline bool contacts(const Rectangle& t, const Moving<Sphere>& s, F&& cb) { return poly_contacts_sphere(t, s, cb); }

// seg_2d_intersect collision.rs:667-688
static inline float signed_2d_tri_area(V2 a, V2 b, V2 c) { return (a.x - c.x) * (b.y - c.y) - (a.y - c.y) * (b.x - c.x); }
static inline bool seg_2d_intersect(V2 a, V2 b, V2 c, V2 d, V2* p, float* t_out) {
  float a1 = signed_2d_tri_area(a, b, d);
  float a2 = signed_2d_tri_area(a, b, c);
  if (a1 * a2 <= 0.0f) {
    float a3 = signed_2d_tri_area(c, d, a);
    float a4 = a3 + a2 - a1;
    if (a3 * a4 <= 0.0f) {
      float t = a3 / (a3 - a4);
      *p = a + t * (b - a);
      *t_out = t;
      return true;
    }
  }
  return false;
}

// ---------------------------------------------------------------------------
// Polygon–Moving<Capsule> collision.rs:693-1086
// ---------------------------------------------------------------------------
template <class Poly, class F>
static inline bool poly_contacts_capsule(const Poly& poly, const Moving<Capsule>& capsule, F&& cb) {
  const Capsule c = capsule.shape;
  const V3 v = capsule.vel;
  const Plane p = to_plane(poly);
  const int NV = num_vertices(poly);
  // :698-719 already colliding
  float denom = dot(p.n, normalize(c.d));
  if (std::fabs(denom) > COLLISION_EPSILON) {
    float t = (p.d - dot(p.n, c.a)) / denom;
    if (t <= 1.0f && t >= 0.0f) {
      V3 q = c.a + c.d * t;
      if (contains(poly, q)) {
        V3 base = (dot(p.n, c.a) - p.d < 0.0f) ? c.a : (c.a + c.d);
        cb(Contact{q, base + -p.n * c.r, p.n, 0.0f});
        return true;
      }
    }
  }
  // :723-764
  Moving<Sphere> start_sphere = sweep(Sphere{c.a, c.r}, v);
  Moving<Sphere> end_sphere = sweep(Sphere{c.a + c.d, c.r}, v);
  bool found = false;
  Contact fc{};
  V3 fdir = v3(0, 0, 0);
  bool fchecked = false;
  {
    Contact c1, c2;
    if (last_contact(p, start_sphere, &c1)) {
      if (last_contact(p, end_sphere, &c2)) {
        if (c2.t < c1.t) {
          found = true; fc = c2; fdir = -c.d; fchecked = false;
        } else {
          if (c2.t == 0.0f) {
            bool contains_1 = contains(poly, c1.a);
            bool contains_2 = contains(poly, c2.a);
            if (contains_1 && contains_2) {
              cb(c2);
              cb(c1);
              return true;
            } else if (contains_1) {
              found = true; fc = c1; fdir = c.d; fchecked = true;
            } else if (contains_2) {
              found = true; fc = c2; fdir = -c.d; fchecked = true;
            } else {
              found = false;
            }
          } else {
            found = true; fc = c1; fdir = c.d; fchecked = false;
          }
        }
      } else {
        found = true; fc = c1; fdir = c.d; fchecked = false;
      }
    } else if (last_contact(p, end_sphere, &c1)) {
      found = true; fc = c1; fdir = -c.d; fchecked = false;
    }
  }
  // :767-890
  if (found) {
    const Contact contact = fc;
    const V3 dir = fdir;
    V3 silhouette_v = dir - p.n * dot(dir, p.n) / magnitude2(p.n);
    V3 n_xy = v3(0.0f, 0.0f, 1.0f);
    Quat plane_rot = quat_from_arc(p.n, n_xy);
    V2 silhouette_a = truncate(rotate_vector(plane_rot, contact.a + -p.n * p.d));
    V2 silhouette_b = truncate(rotate_vector(plane_rot, contact.a + silhouette_v - p.n * p.d));
    if (fchecked || contains(poly, contact.a)) {
      cb(contact);
      if (std::fabs(dot(dir, p.n)) >= COLLISION_EPSILON) return true;
      float t_max = 0.0f;
      for (int edge_i = 0; edge_i < NV; ++edge_i) {
        int a, b;
        edge(poly, edge_i, &a, &b);
        V2 edge_a = truncate(rotate_vector(plane_rot, vertex(poly, a) - p.n * p.d));
        V2 edge_b = truncate(rotate_vector(plane_rot, vertex(poly, b) - p.n * p.d));
        V2 ip; float t;
        if (seg_2d_intersect(silhouette_a, silhouette_b, edge_a, edge_b, &ip, &t)) {
          if (t_max < t) t_max = t;
        }
      }
      float t_max2 = (t_max == 0.0f) ? 1.0f : t_max;
      V3 q = contact.a + silhouette_v * t_max2;
      cb(Contact{q, q, p.n, contact.t});
      return true;
    }
    if (contact.t > 0.0f && std::fabs(dot(dir, p.n)) < COLLISION_EPSILON) {
      float t_min = F32_INF, t_max = 0.0f;
      bool found2 = false;
      for (int edge_i = 0; edge_i < NV; ++edge_i) {
        int a, b;
        edge(poly, edge_i, &a, &b);
        V2 edge_a = truncate(rotate_vector(plane_rot, vertex(poly, a) - p.n * p.d));
        V2 edge_b = truncate(rotate_vector(plane_rot, vertex(poly, b) - p.n * p.d));
        V2 ip; float t;
        if (seg_2d_intersect(silhouette_a, silhouette_b, edge_a, edge_b, &ip, &t)) {
          found2 = true;
          if (t_min > t) t_min = t;
          if (t_max < t) t_max = t;
        }
      }
      if (found2) {
        float t_max2 = (t_max == 0.0f) ? 1.0f : t_max;
        V3 q = contact.a + silhouette_v * t_min;
        float t = contact.t;
        cb(Contact{q, q, p.n, t});
        q = contact.a + silhouette_v * t_max2;
        cb(Contact{q, q, p.n, t});
        return true;
      }
    }
  }
  // :898-971 Minkowski-sum fallback, parallel edges
  if (NV > 64) return false;
  uint64_t parallel_edge_vert = 0;  // bitset.rs:40-55 on u64
  float best_par_t = F32_INF;
  V3 best_par_a = v3(0, 0, 0), best_par_b = v3(0, 0, 0);
  for (int edge_i = 0; edge_i < NV; ++edge_i) {
    int a, b;
    edge(poly, edge_i, &a, &b);
    V3 edge_a = vertex(poly, a), edge_b = vertex(poly, b);
    V3 ab = edge_b - edge_a;
    float ab_cd = dot(ab, c.d);
    if (std::fabs(ab_cd) != magnitude(c.d) * magnitude(ab)) continue;  // exact test :915
    parallel_edge_vert |= (1ull << a);
    parallel_edge_vert |= (1ull << b);
    Ray ray{c.a, v};
    if (ab_cd < 0.0f) { V3 tmp = edge_a; edge_a = edge_b; edge_b = tmp; }
    Capsule edge_sum{edge_a, edge_b - edge_a, c.r};
    float m_edge = magnitude2(ab);
    Intersection inter;
    if (ray_capsule(ray, edge_sum, &inter)) {
      if (inter.t > fminf_rs(best_par_t, 1.0f)) continue;
      V3 tri_p = seg_closest_point(Segment{edge_a, edge_b}, inter.p);
      float m_proj = magnitude2((tri_p + c.d) - edge_a);
      float c_t = (m_proj > m_edge) ? (m_proj - m_edge) / (m_proj - magnitude2(tri_p - edge_a)) : 1.0f;
      V3 q = tri_p + c.d * c_t;
      best_par_t = inter.t; best_par_a = tri_p; best_par_b = q;
    } else if (ray_capsule(ray, Capsule{edge_a, -c.d, c.r}, &inter)) {
      if (inter.t > fminf_rs(best_par_t, 1.0f)) continue;
      V3 d = inter.p - edge_a;
      float capsule_t = -dot(d, c.d) / magnitude2(c.d);
      V3 tri_p = seg_closest_point(Segment{edge_a, edge_a + -c.d}, inter.p);
      V3 pa = tri_p + c.d * capsule_t;
      float m_proj = magnitude2((tri_p + c.d) - edge_a);
      V3 pb = (m_proj > m_edge) ? edge_b : (tri_p + c.d);
      best_par_t = inter.t; best_par_a = pa; best_par_b = pb;
    }
  }
  // :973-1060 edge quads + vertex capsules
  float best_sum_t = F32_INF;
  V3 best_sum_p = v3(0, 0, 0);
  for (int edge_i = 0; edge_i < NV; ++edge_i) {
    int a, b;
    edge(poly, edge_i, &a, &b);
    bool a_on_parallel_edge = (parallel_edge_vert >> a) & 1ull;
    bool b_on_parallel_edge = (parallel_edge_vert >> b) & 1ull;
    if (a_on_parallel_edge && b_on_parallel_edge) continue;
    V3 edge_a = vertex(poly, a), edge_b = vertex(poly, b);
    Triangle tris[2] = {Triangle{edge_a + -c.d, edge_a, edge_b}, Triangle{edge_a + -c.d, edge_b, edge_b + -c.d}};
    Plane p2 = to_plane(tris[1]);
    Sphere s{c.a, c.r};
    contacts(p2, sweep(s, v), [&](const Contact& contact) {
      if (best_sum_t > contact.t && (contains(tris[0], contact.a) || contains(tris[1], contact.b))) {
        V3 d = contact.a - edge_a;
        float capsule_t = -dot(d, c.d) / magnitude2(c.d);
        best_sum_t = contact.t;
        best_sum_p = contact.a + c.d * capsule_t;
      } else {
        Ray ray{c.a, v};
        Intersection inter;
        Capsule bottom_edge{edge_a, edge_b - edge_a, c.r};
        if (ray_capsule(ray, bottom_edge, &inter)) {
          if (inter.t <= 1.0f && inter.t <= best_sum_t) {
            V3 q = seg_closest_point(Segment{edge_a, edge_b}, inter.p);
            best_sum_t = inter.t; best_sum_p = q;
          }
        }
        Capsule top_edge{edge_a + -c.d, edge_b - edge_a, c.r};
        if (ray_capsule(ray, top_edge, &inter)) {
          if (inter.t <= 1.0f && inter.t <= best_sum_t) {
            V3 plane_p = inter.p + c.d;
            V3 q = seg_closest_point(Segment{edge_a, edge_b}, plane_p);
            best_sum_t = inter.t; best_sum_p = q;
          }
        }
        const V3 verts[2] = {edge_a, edge_b};
        const bool par[2] = {a_on_parallel_edge, b_on_parallel_edge};
        for (int k = 0; k < 2; ++k) {
          if (par[k]) continue;
          Capsule cap{verts[k], -c.d, c.r};
          if (ray_capsule(ray, cap, &inter)) {
            if (inter.t <= 1.0f && inter.t <= best_sum_t) {
              best_sum_t = inter.t; best_sum_p = verts[k];
            }
          }
        }
      }
    });
  }
  // :1061-1085  (n: p.n — the polygon's plane; the inner `p` at :993 is scoped to the loop)
  if (best_sum_t < best_par_t) {
    cb(Contact{best_sum_p, best_sum_p, p.n, best_sum_t});
  } else if (best_par_t != F32_INF) {
    cb(Contact{best_par_a, best_par_a, p.n, best_par_t});
    cb(Contact{best_par_b, best_par_b, p.n, best_par_t});
  } else {
    return false;
  }
  return true;
}
template <class F>
static inline bool contacts(const Triangle& t, const Moving<Capsule>& s, F&& cb) { return poly_contacts_capsule(t, s, cb); }
template <class F>
static inline bool contacts(const Rectangle& t, const Moving<Capsule>& s, F&& cb) { return poly_contacts_capsule(t, s, cb); }

// ---------------------------------------------------------------------------
// Sphere–Moving<Sphere> collision.rs:1089-1141
// ---------------------------------------------------------------------------
template <class F>
static inline bool contacts(const Sphere& self, const Moving<Sphere>& sphere, F&& cb) {
  const Sphere& s = sphere.shape;
  V3 v = sphere.vel;
  float r = self.r + s.r;
  V3 d = s.c - self.c;
  float len = magnitude2(d);
  if (len <= r * r) {
    V3 n;
    if (len == 0.0f) {
      if (is_zero(v)) return false;
      n = -normalize(v);
    } else {
      n = d / std::sqrt(len);
    }
    cb(Contact{self.c + n * self.r, s.c + -n * s.r, n, 0.0f});
    return true;
  }
  float l = magnitude2(v);
  if (l == 0.0f) return false;
  Ray ray{self.c, -v};
  Intersection inter;
  if (ray_sphere(ray, Sphere{s.c, r}, &inter)) {
    if (inter.t <= 1.0f) {
      V3 end_c = s.c + v * inter.t;
      V3 ba = normalize(end_c - self.c);
      V3 a = self.c + ba * self.r;
      cb(Contact{a, a, ba, inter.t});
      return true;
    }
  }
  return false;
}

// Capsule–Moving<Sphere> collision.rs:1145-1203
template <class F>
static inline bool contacts(const Capsule& self, const Moving<Sphere>& sphere, F&& cb) {
  const Sphere& s = sphere.shape;
  V3 v = sphere.vel;
  float r = self.r + s.r;
  V3 closest_pt = seg_closest_point(Segment{self.a, self.a + self.d}, s.c);
  V3 d = s.c - closest_pt;
  float len = magnitude2(d);
  if (len <= r * r) {
    V3 n;
    if (len == 0.0f) {
      if (is_zero(v)) return false;
      n = -normalize(v);
    } else {
      n = d / std::sqrt(len);
    }
    cb(Contact{closest_pt + n * self.r, s.c + -n * s.r, n, 0.0f});
    return true;
  }
  float l = magnitude2(v);
  if (l == 0.0f) return false;
  Ray ray{s.c, v};
  Intersection inter;
  if (ray_capsule(ray, Capsule{self.a, self.d, s.r + self.r}, &inter)) {
    if (inter.t <= 1.0f) {
      V3 b = s.c + v * inter.t;
      V3 a = seg_closest_point(Segment{self.a, self.a + self.d}, b);
      V3 ba = normalize(b - a);
      V3 q = a + ba * self.r;
      cb(Contact{q, q, ba, inter.t});
      return true;
    }
  }
  return false;
}

// Capsule–Moving<Capsule> collision.rs:1205-1356
template <class F>
static inline bool contacts(const Capsule& self, const Moving<Capsule>& capsule, F&& cb) {
  const Capsule c = capsule.shape;
  const V3 v = capsule.vel;
  Segment self_seg{self.a, self.a + self.d};
  V3 p1, p2, tmp;
  {
    V3 p, e;
    if (closest_pts_seg(self_seg, Segment{c.a, c.a + v}, &p, &tmp)) {
      if (closest_pts_seg(self_seg, Segment{c.a + c.d, c.a + c.d + v}, &e, &tmp)) {
        p1 = p; p2 = e;
      } else {
        return false;
      }
    } else {
      p1 = self.a; p2 = self.a + self.d;
    }
  }
  Segment clipped{p1, p2};
  {
    V3 q;
    if (closest_pts_seg(clipped, Segment{c.a, c.a + c.d}, &q, &tmp)) {
      Sphere ss{q, self.r};
      return contacts(ss, capsule, cb);  // -> Sphere–Moving<Capsule> via commute (declared below)
    }
  }
  // Parallel capsules :1234-1355
  float d_mag2 = magnitude2(self.d);
  float t1 = dot(c.a - self.a, self.d) / d_mag2;
  float t2 = dot(c.a + c.d - self.a, self.d) / d_mag2;
  float t_min, t_max;
  V3 c_a, c_d;
  if (t1 < t2) { t_min = t1; t_max = t2; c_a = c.a; c_d = c.d; }
  else { t_min = t2; t_max = t1; c_a = c.a + c.d; c_d = -c.d; }
  V3 h = self.a - (c_a + c_d * (-t_min / (t_max - t_min)));
  float h_len = magnitude(h);
  if (h_len <= self.r + c.r) {
    if (t_max <= 0.0f) return contacts(self, sweep(Sphere{c_a + c_d, c.r}, v), cb);
    if (t_min >= 1.0f) return contacts(self, sweep(Sphere{c_a, c.r}, v), cb);
    float s_t = (clampf(t_min, 0.0f, 1.0f) + clampf(t_max, 0.0f, 1.0f)) * 0.5f;
    float o_t = (s_t - t_min) / (t_max - t_min);
    V3 a_c = self.a + self.d * s_t;
    V3 b_c = c_a + c_d * o_t;
    V3 ab = b_c - a_c;
    V3 n;
    if (is_zero(ab)) {
      if (is_zero(v)) return false;
      n = -normalize(v);
    } else {
      n = normalize(b_c - a_c);
    }
    cb(Contact{a_c + n * self.r, b_c + -n * c.r, n, 0.0f});
    return true;
  }
  float h_rat = (h_len - self.r - c.r) / h_len;
  float v_comp = dot(v, h) / (h_len * h_len);
  if (v_comp < h_rat) return false;
  float coll_t = h_rat / v_comp;
  V3 v_travel = v * coll_t;
  float axis_t_delta = dot(v_travel, self.d) / d_mag2;
  t_min = t_min + axis_t_delta;
  t_max = t_max + axis_t_delta;
  if (t_max <= 0.0f) return contacts(self, sweep(Sphere{c_a + c_d, c.r}, v), cb);
  if (t_min >= 1.0f) return contacts(self, sweep(Sphere{c_a, c.r}, v), cb);
  float s_t = (clampf(t_min, 0.0f, 1.0f) + clampf(t_max, 0.0f, 1.0f)) * 0.5f;
  float o_t = (s_t - t_min) / (t_max - t_min);
  V3 a_c = self.a + self.d * s_t;
  V3 b_c = c_a + c_d * o_t + v_travel;
  V3 ab = b_c - a_c;
  V3 n;
  if (is_zero(ab)) {
    if (is_zero(v)) return false;
    n = -normalize(v);
  } else {
    n = normalize(b_c - a_c);
  }
  cb(Contact{a_c + n * self.r, b_c + -n * c.r, n, coll_t});
  return true;
}

// Moving<Recv>.contacts(&Arg) collision.rs:1368-1382 — rhs static, self moving.
template <class Recv, class Arg, class F>
static inline bool moving_contacts_static(const Moving<Recv>& self, const Arg& rhs, F&& cb) {
  Moving<Arg> rhs_moving = sweep(rhs, -self.vel);
  return contacts(self.shape, rhs_moving, [&](const Contact& c) {
    V3 d = self.vel * c.t;
    cb(Contact{c.a + d, c.b + d, c.n, c.t});
  });
}
// commute_contacts!{ Sphere, Moving<Capsule> } collision.rs:1143:
//   Sphere.contacts(&Moving<Capsule>) = rhs.contacts(self, |c| cb(-c)), where
//   Moving<Capsule>.contacts(&Sphere) is the :1368 wrapper over Capsule–Moving<Sphere>.
template <class F>
static inline bool contacts(const Sphere& self, const Moving<Capsule>& rhs, F&& cb) {
  return moving_contacts_static(rhs, self, [&](const Contact& c) { cb(neg(c)); });
}

// Moving<Recv>.contacts(&Moving<Arg>) collision.rs:1387-1401
template <class Recv, class Arg, class F>
static inline bool contacts(const Moving<Recv>& self, const Moving<Arg>& rhs, F&& cb) {
  V3 v_a = self.vel;
  return contacts(self.shape, sweep(rhs.shape, rhs.vel - v_a), [&](const Contact& c) {
    cb(Contact{c.a + v_a * c.t, c.b + v_a * c.t, c.n, c.t});
  });
}
// commute_contacts!{ Moving<Sphere>/Moving<Capsule>, Triangle/Rectangle } collision.rs:661-664
template <class S, class Poly, class F>
static inline bool moving_contacts_poly(const Moving<S>& self, const Poly& poly, F&& cb) {
  return contacts(poly, self, [&](const Contact& c) { cb(neg(c)); });
}

// ---------------------------------------------------------------------------
// Component (compound.rs:33-228)
// ---------------------------------------------------------------------------
enum ComponentKind : int { COMP_SPHERE = 0, COMP_CAPSULE = 1 };
struct Component {
  int kind;
  Sphere s;   // valid when kind == COMP_SPHERE
  Capsule c;  // valid when kind == COMP_CAPSULE
};
static inline Component component(const Sphere& s) { Component k{}; k.kind = COMP_SPHERE; k.s = s; return k; }
static inline Component component(const Capsule& c) { Component k{}; k.kind = COMP_CAPSULE; k.c = c; return k; }
static inline V3 center(const Component& k) { return k.kind == COMP_SPHERE ? center(k.s) : center(k.c); }
static inline AABB bounds(const Component& k) { return k.kind == COMP_SPHERE ? bounds(k.s) : bounds(k.c); }
static inline Component operator+(Component k, V3 v) { if (k.kind == COMP_SPHERE) k.s = k.s + v; else k.c = k.c + v; return k; }
static inline Component operator-(Component k, V3 v) { if (k.kind == COMP_SPHERE) k.s = k.s - v; else k.c = k.c - v; return k; }
// Shape::set_pos geom.rs:459-462
static inline void set_pos(Component& k, V3 p) { V3 disp = p - center(k); k = k + disp; }

struct ComponentConstructor { int kind; float r; float half_h; };
// Component::deconstruct compound.rs:42-52
static inline void deconstruct(const Component& k, V3* x, Quat* q, ComponentConstructor* cons) {
  if (k.kind == COMP_SPHERE) {
    *x = k.s.c; *q = quat_one(); *cons = ComponentConstructor{COMP_SPHERE, k.s.r, 0.0f};
  } else {
    float h = magnitude(k.c.d);
    *q = quat_from_arc(v3(0.0f, 1.0f, 0.0f) * h, k.c.d);
    *x = k.c.a + k.c.d * 0.5f;
    *cons = ComponentConstructor{COMP_CAPSULE, k.c.r, h * 0.5f};
  }
}
// ComponentConstructor::construct compound.rs:219-227
static inline Component construct(const ComponentConstructor& cons, V3 p, Quat rot) {
  if (cons.kind == COMP_SPHERE) return component(Sphere{p, cons.r});
  V3 d = rotate_vector(rot, v3(0.0f, 1.0f, 0.0f) * cons.half_h);
  return component(Capsule{p + -d, d * 2.0f, cons.r});
}

// BoundedBy<AABB> for Moving<T> bounds.rs:60-68
template <class T>
static inline AABB bounds(const Moving<T>& m) {
  AABB s_bounds = bounds(m.shape);
  AABB e_bounds = s_bounds + m.vel;
  return aabb_combine(s_bounds, e_bounds);
}

// Contacts<RHS> for Moving<Component> compound.rs:180-190, with
// RHS = Moving<Component> resolving recursively through the same impl and
// finally through collision.rs:1387 (see DESIGN.md "trait resolution").
template <class F>
static inline bool contacts(const Moving<Component>& self, const Moving<Component>& rhs, F&& cb) {
  // self.0 match -> rhs.contacts(&Moving(shape_self, self.1), |c| cb(-c))
  // rhs.0 match  -> Moving(shape_self).contacts(&Moving(shape_rhs, rhs.1), |c| cb'(-c))
  auto inner = [&](const auto& self_shape) {
    Moving<std::decay_t<decltype(self_shape)>> ms = sweep(self_shape, self.vel);
    auto cb1 = [&](const Contact& c) { cb(neg(c)); };
    auto cb2 = [&](const Contact& c) { cb1(neg(c)); };
    if (rhs.shape.kind == COMP_SPHERE) return contacts(ms, sweep(rhs.shape.s, rhs.vel), cb2);
    return contacts(ms, sweep(rhs.shape.c, rhs.vel), cb2);
  };
  if (self.shape.kind == COMP_SPHERE) return inner(self.shape.s);
  return inner(self.shape.c);
}
// Moving<Component>.contacts(&Triangle) compound.rs:180-190 (RHS = Triangle)
template <class F>
static inline bool contacts(const Moving<Component>& self, const Triangle& tri, F&& cb) {
  auto cb1 = [&](const Contact& c) { cb(neg(c)); };
  if (self.shape.kind == COMP_SPHERE) return contacts(tri, sweep(self.shape.s, self.vel), cb1);
  return contacts(tri, sweep(self.shape.c, self.vel), cb1);
}

// LocalContacts<Moving<Component>> for Moving<Component> compound.rs:192-207
template <class F>
static inline bool local_contacts(const Moving<Component>& self, const Moving<Component>& rhs, F&& cb) {
  return contacts(self, rhs, [&](const Contact& c) {
    cb(LocalContact{c.a + -(center(self.shape) + self.vel * c.t), c.b + -(center(rhs.shape) + rhs.vel * c.t), c});
  });
}

}  // namespace mgfo
