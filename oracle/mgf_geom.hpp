// ORACLE — TEST INFRASTRUCTURE ONLY (see mgf_math.hpp header).
// CPU restatement of the reference's shapes and bounds:
//   src/geom.rs  : Plane :32-58, Ray :63-86, Segment :91-124 & closest_point :590-603,
//                  Triangle :128-192, Rectangle :216-246 & vertex/edge :903-923,
//                  AABB :257-267, Sphere :290-306, Capsule :316-352, Moving :357-395,
//                  closest_pts_seg :408-444, Polygon for Triangle :889-901,
//                  compute_basis :1138-1145, COLLISION_EPSILON :27
//   src/bounds.rs: Moving bounds :60-68, AABB + f32 :91-98, combine :113-130,
//                  surface_area :132-134, Triangle/Sphere/Capsule -> AABB :137-188
#pragma once
#include "mgf_math.hpp"

namespace mgfo {

static constexpr float COLLISION_EPSILON = 0.000001f;  // geom.rs:27

static inline float clampf(float n, float mn, float mx) {  // geom.rs:398, collision.rs:1358
  if (n < mn) return mn;
  if (n > mx) return mx;
  return n;
}

struct Plane { V3 n; float d; };
struct Ray { V3 p; V3 d; };
struct Segment { V3 a; V3 b; };
struct Triangle { V3 a, b, c; };
struct Rectangle { V3 c; V3 u[2]; float e[2]; };
struct AABB { V3 c; V3 r; };
struct Sphere { V3 c; float r; };
struct Capsule { V3 a; V3 d; float r; };

// Plane::from((a,b,c)) geom.rs:49-58
static inline Plane plane_from_points(V3 a, V3 b, V3 c) {
  V3 n = normalize(cross(b - a, c - a));
  return Plane{n, dot(n, a)};
}
static inline Plane to_plane(const Triangle& t) { return plane_from_points(t.a, t.b, t.c); }  // geom.rs:182-192
static inline Plane to_plane(const Rectangle& r) {  // geom.rs:240-246
  V3 n = cross(r.u[1], r.u[0]);
  return Plane{n, dot(n, r.c)};
}

// Segment::closest_point geom.rs:590-603
static inline V3 seg_closest_point(const Segment& s, V3 to) {
  V3 ab = s.b - s.a;
  float t = dot(ab, to - s.a);
  if (t <= 0.0f) return s.a;
  float denom = dot(ab, ab);
  if (t >= denom) return s.b;
  return s.a + ab * (t / denom);
}

// Triangle::closest_point geom.rs:643-688 (pinned by geom.rs:1154-1161)
static inline V3 tri_closest_point(const Triangle& t, V3 to) {
  V3 ab = t.b - t.a, ac = t.c - t.a, ap = to - t.a;
  float d1 = dot(ab, ap), d2 = dot(ac, ap);
  if (d1 <= 0.0f && d2 <= 0.0f) return t.a;
  V3 bp = to - t.b;
  float d3 = dot(ab, bp), d4 = dot(ac, bp);
  if (d3 >= 0.0f && d4 <= d3) return t.b;
  float vc = d1 * d4 - d3 * d2;
  if (vc <= 0.0f && d1 >= 0.0f && d3 <= 0.0f) {
    float v = d1 / (d1 - d3);
    return t.a + ab * v;
  }
  V3 cp = to - t.c;
  float d5 = dot(ab, cp), d6 = dot(ac, cp);
  if (d6 >= 0.0f && d5 <= d6) return t.c;
  float vb = d5 * d2 - d1 * d6;
  if (vb <= 0.0f && d2 >= 0.0f && d6 <= 0.0f) {
    float w = d2 / (d2 - d6);
    return t.a + ac * w;
  }
  float va = d3 * d6 - d5 * d4;
  if (va <= 0.0f && (d4 - d3) >= 0.0f && (d5 - d6) >= 0.0f) {
    float w = (d4 - d3) / ((d4 - d3) + (d5 - d6));
    return t.b + (t.c - t.b) * w;
  }
  float denom = 1.0f / (va + vb + vc);
  float v = vb * denom, w = vc * denom;
  return t.a + ab * v + ac * w;
}

// closest_pts_seg geom.rs:408-444.  Returns false for the `None` (parallel) case.
static inline bool closest_pts_seg(const Segment& s1, const Segment& s2, V3* p1, V3* p2) {
  V3 d1 = s1.b - s1.a, d2 = s2.b - s2.a;
  float a = magnitude2(d1), e = magnitude2(d2);
  V3 r = s1.a - s2.a;
  float f = dot(d2, r);
  float s, t;
  if (a <= COLLISION_EPSILON) {
    if (e <= COLLISION_EPSILON) { s = 0.5f; t = 0.5f; }
    else { s = 0.5f; t = clampf(f / e, 0.0f, 1.0f); }
  } else {
    float c = dot(d1, r);
    if (e <= COLLISION_EPSILON) {
      s = clampf(-c / a, 0.0f, 1.0f); t = 0.0f;
    } else {
      float b = dot(d1, d2);
      float denom = a * e - b * b;
      if (denom != 0.0f) s = clampf((b * f - c * e) / denom, 0.0f, 1.0f);
      else return false;
      float tt = b * s + f;
      if (tt < 0.0f) { s = clampf(-c / a, 0.0f, 1.0f); t = 0.0f; }
      else if (tt > e) { s = clampf((b - c) / a, 0.0f, 1.0f); t = 1.0f; }
      else { t = tt / e; }
    }
  }
  *p1 = s1.a + d1 * s;
  *p2 = s2.a + d2 * t;
  return true;
}

// Shape::center
static inline V3 center(const Sphere& s) { return s.c; }                 // geom.rs:747
static inline V3 center(const Capsule& c) { return c.a + c.d * 0.5f; }   // geom.rs:787
static inline V3 center(const Triangle& t) { return (t.a + t.b + t.c) / 3.0f; }  // geom.rs:639
static inline V3 center(const Rectangle& r) { return r.c; }
// Shape + Vector3 (impl_shape_reqs! geom.rs:468-497, Capsule :758-784, Triangle :606-636)
static inline Sphere operator+(Sphere s, V3 v) { s.c = s.c + v; return s; }
static inline Sphere operator-(Sphere s, V3 v) { s.c = s.c + -v; return s; }
static inline Capsule operator+(Capsule c, V3 v) { c.a = c.a + v; return c; }
static inline Capsule operator-(Capsule c, V3 v) { c.a = c.a + -v; return c; }
static inline AABB operator+(AABB b, V3 v) { b.c = b.c + v; return b; }
static inline AABB operator-(AABB b, V3 v) { b.c = b.c + -v; return b; }
static inline AABB operator+(AABB b, float s) { b.r = b.r + v3(s, s, s); return b; }  // bounds.rs:91-98

// Polygon trait geom.rs:869-923
struct TrianglePoly {
  static constexpr int NUM_VERTICES = 3;
};
static inline int num_vertices(const Triangle&) { return 3; }
static inline int num_vertices(const Rectangle&) { return 4; }
static inline V3 vertex(const Triangle& t, int i) { return i == 0 ? t.a : (i == 1 ? t.b : t.c); }
static inline void edge(const Triangle&, int i, int* a, int* b) {
  static const int E[3][2] = {{0, 1}, {1, 2}, {2, 0}};
  *a = E[i][0]; *b = E[i][1];
}
static inline V3 vertex(const Rectangle& r, int i) {
  switch (i) {
    case 0: return r.c + r.u[0] * r.e[0] + r.u[1] * r.e[1];
    case 1: return r.c + r.u[0] * r.e[0] + -r.u[1] * r.e[1];
    case 2: return r.c + -r.u[0] * r.e[0] + -r.u[1] * r.e[1];
    default: return r.c + -r.u[0] * r.e[0] + r.u[1] * r.e[1];
  }
}
static inline void edge(const Rectangle&, int i, int* a, int* b) {
  static const int E[4][2] = {{0, 1}, {1, 2}, {2, 3}, {3, 0}};
  *a = E[i][0]; *b = E[i][1];
}

// compute_basis geom.rs:1138-1145
static inline void compute_basis(V3 n, V3 out[2]) {
  V3 b = (std::fabs(n.x) >= 0.57735f) ? v3(n.y, -n.x, 0.0f) : v3(0.0f, n.z, -n.y);
  b = normalize(b);
  out[0] = b;
  out[1] = cross(n, b);
}

// ---------------------------------------------------------------------------
// Bounds (src/bounds.rs)
// ---------------------------------------------------------------------------
static inline float fminf_rs(float a, float b) { return std::fmin(a, b); }  // f32::min
static inline float fmaxf_rs(float a, float b) { return std::fmax(a, b); }  // f32::max

struct BoundsAssert { bool failed = false; };

// Bound::combine for AABB bounds.rs:113-130.  The reference asserts r >= 0.
static inline AABB aabb_combine(const AABB& a, const AABB& b) {
  V3 lower = v3(fminf_rs(a.c.x - a.r.x, b.c.x - b.r.x), fminf_rs(a.c.y - a.r.y, b.c.y - b.r.y),
                fminf_rs(a.c.z - a.r.z, b.c.z - b.r.z));
  V3 upper = v3(fmaxf_rs(a.c.x + a.r.x, b.c.x + b.r.x), fmaxf_rs(a.c.y + a.r.y, b.c.y + b.r.y),
                fmaxf_rs(a.c.z + a.r.z, b.c.z + b.r.z));
  V3 r = (upper - lower) / 2.0f;
  V3 c = (upper + lower) / 2.0f;
  return AABB{c, r};
}
static inline float aabb_surface_area(const AABB& a) {  // bounds.rs:132-134
  return a.r.x * a.r.y + a.r.y * a.r.z + a.r.z * a.r.x;
}
static inline AABB bounds(const Triangle& t) {  // bounds.rs:137-154
  V3 c = (t.a + t.b + t.c) / 3.0f;
  float d0 = fmaxf_rs(std::fabs(t.a.x - c.x), fmaxf_rs(std::fabs(t.b.x - c.x), std::fabs(t.c.x - c.x)));
  float d1 = fmaxf_rs(std::fabs(t.a.y - c.y), fmaxf_rs(std::fabs(t.b.y - c.y), std::fabs(t.c.y - c.y)));
  float d2 = fmaxf_rs(std::fabs(t.a.z - c.z), fmaxf_rs(std::fabs(t.b.z - c.z), std::fabs(t.c.z - c.z)));
  return AABB{c, v3(d0, d1, d2)};
}
static inline AABB bounds(const Rectangle& q) {  // bounds.rs:156-168
  V3 p1 = q.c + q.u[0] * q.e[0], p2 = q.c + q.u[1] * q.e[1];
  return AABB{q.c, v3(fmaxf_rs(std::fabs(p1.x - q.c.x), std::fabs(p2.x - q.c.x)), fmaxf_rs(std::fabs(p1.y - q.c.y), std::fabs(p2.y - q.c.y)),
                      fmaxf_rs(std::fabs(p1.z - q.c.z), std::fabs(p2.z - q.c.z)))};
}
static inline AABB bounds(const Sphere& s) { return AABB{s.c, v3(s.r, s.r, s.r)}; }  // bounds.rs:170-177
static inline AABB bounds(const Capsule& c) {  // bounds.rs:179-188
  float r = c.r + magnitude(c.d) * 0.5f;
  return AABB{c.a + c.d * 0.5f, v3(r, r, r)};
}
static inline AABB bounds(const AABB& a) { return a; }  // bounds.rs:54-58

// Overlaps<AABB> for AABB collision.rs:22-29
static inline bool aabb_overlaps(const AABB& a, const AABB& b) {
  return std::fabs(a.c.x - b.c.x) <= (a.r.x + b.r.x) && std::fabs(a.c.y - b.c.y) <= (a.r.y + b.r.y) &&
         std::fabs(a.c.z - b.c.z) <= (a.r.z + b.r.z);
}
// Contains<Point3> for AABB collision.rs:114-120
static inline bool aabb_contains_point(const AABB& a, V3 p) {
  return std::fabs(a.c.x - p.x) <= a.r.x && std::fabs(a.c.y - p.y) <= a.r.y && std::fabs(a.c.z - p.z) <= a.r.z;
}
// Contains<AABB> for AABB collision.rs:129-135
static inline bool aabb_contains(const AABB& a, const AABB& rhs) {
  V3 rhs_max = rhs.c + rhs.r;
  V3 rhs_min = rhs.c + -rhs.r;
  return aabb_contains_point(a, rhs_max) && aabb_contains_point(a, rhs_min);
}

}  // namespace mgfo
