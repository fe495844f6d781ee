// Device narrowphase for the mgf hot path (gfx950): exact swept contact generation for
// Sphere / Capsule / Triangle / Plane.  One function per reference routine, value-returning
// (no closures, no recursion, at most 2 contacts) so each lane keeps its state in VGPRs.
//
// Reference routines followed (src/collision.rs unless noted):
//   Triangle::contains :85-100        Ray∩Sphere :249-273         Ray∩Capsule :275-359
//   Plane–Moving<Sphere> :521-553     Plane–Moving<Capsule> :555-605
//   Triangle–Moving<Sphere> :610-659  seg_2d_intersect :667-688   Triangle–Moving<Capsule> :693-1086
//   Sphere–Moving<Sphere> :1089-1141  Capsule–Moving<Sphere> :1145-1203
//   Capsule–Moving<Capsule> :1205-1356  Moving wrappers :1368-1401  commute_contacts! :484-494,1143
//   geom.rs: Segment::closest_point :590-603, closest_pts_seg :408-444, Plane::from :49-58
//   compound.rs: Component dispatch :159-207, ComponentConstructor::construct :219-227
//   bounds.rs: swept bounds :60-68, combine :113-130, Sphere/Capsule/Triangle AABB :137-188
#pragma once
#include "dev_math.h"

namespace mgf {

struct Sphere { V3 c; float r; };
struct Capsule { V3 a; V3 d; float r; };
struct Triangle { V3 a, b, c; };
struct Plane { V3 n; float d; };
struct Box { V3 c; V3 r; };  // AABB: centre + half extents (geom.rs:257)
struct Contact { V3 a, b, n; float t; };
struct LocalContact { V3 la, lb; Contact g; };

enum : int { KIND_SPHERE = 0, KIND_CAPSULE = 1 };
struct Comp { int kind; V3 p; V3 d; float r; };  // Component: sphere{c=p,r} | capsule{a=p,d,r}

HD Contact mkc(V3 a, V3 b, V3 n, float t) { Contact c; c.a = a; c.b = b; c.n = n; c.t = t; return c; }
HD Contact neg(Contact c) { return mkc(c.b, c.a, -c.n, c.t); }  // :444-456
HD float clampf(float n, float mn, float mx) { return n < mn ? mn : (n > mx ? mx : n); }  // :1358, geom.rs:398

HD Sphere mks(V3 c, float r) { Sphere s; s.c = c; s.r = r; return s; }
HD Capsule mkcap(V3 a, V3 d, float r) { Capsule c; c.a = a; c.d = d; c.r = r; return c; }
HD Triangle mkt(V3 a, V3 b, V3 c) { Triangle t; t.a = a; t.b = b; t.c = c; return t; }

HD V3 comp_center(const Comp& k) { return k.kind == KIND_SPHERE ? k.p : (k.p + k.d * 0.5f); }  // geom.rs:747,787

// ---- bounds (bounds.rs) ------------------------------------------------------------
HD Box box_combine(const Box& a, const Box& b) {  // :113-130
  V3 lo = mk3(fmin_rs(a.c.x - a.r.x, b.c.x - b.r.x), fmin_rs(a.c.y - a.r.y, b.c.y - b.r.y), fmin_rs(a.c.z - a.r.z, b.c.z - b.r.z));
  V3 hi = mk3(fmax_rs(a.c.x + a.r.x, b.c.x + b.r.x), fmax_rs(a.c.y + a.r.y, b.c.y + b.r.y), fmax_rs(a.c.z + a.r.z, b.c.z + b.r.z));
  Box o; o.r = (hi - lo) / 2.0f; o.c = (hi + lo) / 2.0f; return o;
}
HD float box_area(const Box& a) { return a.r.x * a.r.y + a.r.y * a.r.z + a.r.z * a.r.x; }  // :132-134
HD Box comp_bounds(const Comp& k) {  // :170-188
  Box o;
  if (k.kind == KIND_SPHERE) { o.c = k.p; o.r = mk3(k.r, k.r, k.r); }
  else { float r = k.r + mag(k.d) * 0.5f; o.c = k.p + k.d * 0.5f; o.r = mk3(r, r, r); }
  return o;
}
HD Box swept_bounds(const Comp& k, V3 delta) {  // :60-68
  Box s = comp_bounds(k);
  Box e = s; e.c = s.c + delta;
  return box_combine(s, e);
}
HD Box tri_bounds(const Triangle& t) {  // :137-154
  V3 c = (t.a + t.b + t.c) / 3.0f;
  Box o; o.c = c;
  o.r = mk3(fmax_rs(fabs_rs(t.a.x - c.x), fmax_rs(fabs_rs(t.b.x - c.x), fabs_rs(t.c.x - c.x))),
            fmax_rs(fabs_rs(t.a.y - c.y), fmax_rs(fabs_rs(t.b.y - c.y), fabs_rs(t.c.y - c.y))),
            fmax_rs(fabs_rs(t.a.z - c.z), fmax_rs(fabs_rs(t.b.z - c.z), fabs_rs(t.c.z - c.z))));
  return o;
}
HD bool box_overlaps(const Box& a, const Box& b) {  // collision.rs:22-29
  return fabs_rs(a.c.x - b.c.x) <= (a.r.x + b.r.x) && fabs_rs(a.c.y - b.c.y) <= (a.r.y + b.r.y) &&
         fabs_rs(a.c.z - b.c.z) <= (a.r.z + b.r.z);
}
HD bool box_contains_pt(const Box& a, V3 p) {  // :114-120
  return fabs_rs(a.c.x - p.x) <= a.r.x && fabs_rs(a.c.y - p.y) <= a.r.y && fabs_rs(a.c.z - p.z) <= a.r.z;
}
HD bool box_contains(const Box& a, const Box& rhs) {  // :129-135
  return box_contains_pt(a, rhs.c + rhs.r) && box_contains_pt(a, rhs.c + -rhs.r);
}
// ComponentConstructor::construct compound.rs:219-227
HD Comp construct(int kind, float r, float half_h, V3 p, Quat rot) {
  Comp k; k.kind = kind; k.r = r;
  if (kind == KIND_SPHERE) { k.p = p; k.d = mk3(0.0f, 0.0f, 0.0f); }
  else { V3 d = rotate(rot, mk3(0.0f, 1.0f, 0.0f) * half_h); k.p = p + -d; k.d = d * 2.0f; }
  return k;
}

// ---- helpers ------------------------------------------------------------------------
HD Plane plane_from(V3 a, V3 b, V3 c) {  // geom.rs:49-58
  Plane p; p.n = normalize(cross(b - a, c - a)); p.d = dot(p.n, a); return p;
}
HD V3 seg_closest(V3 sa, V3 sb, V3 to) {  // geom.rs:590-603
  V3 ab = sb - sa;
  float t = dot(ab, to - sa);
  if (t <= 0.0f) return sa;
  float denom = dot(ab, ab);
  if (t >= denom) return sb;
  return sa + ab * (t / denom);
}
HD bool tri_contains(const Triangle& t, V3 p) {  // :85-100
  V3 v = p - t.a, ac = t.c - t.a, ab = t.b - t.a;
  float d1 = dot(ac, ac), d2 = dot(ac, ab), d3 = dot(ac, v), d4 = dot(ab, ab), d5 = dot(ab, v);
  float invd = 1.0f / (d1 * d4 - d2 * d2);
  float u = (d4 * d3 - d2 * d5) * invd;
  float w = (d1 * d5 - d2 * d3) * invd;
  return u >= 0.0f && w >= 0.0f && (u + w) < 1.0f;
}
HD V3 tri_vertex(const Triangle& t, int i) { return i == 0 ? t.a : (i == 1 ? t.b : t.c); }

// closest_pts_seg geom.rs:408-444; false = None.  Only the first point is used on this path.
__device__ inline bool closest_pts_seg_first(V3 a1, V3 b1, V3 a2, V3 b2, V3* p1) {
  V3 d1 = b1 - a1, d2 = b2 - a2;
  float a = mag2(d1), e = mag2(d2);
  V3 r = a1 - a2;
  float f = dot(d2, r);
  float s;
  if (a <= kCollisionEps) {
    s = 0.5f;
  } else {
    float c = dot(d1, r);
    if (e <= kCollisionEps) {
      s = clampf(-c / a, 0.0f, 1.0f);
    } else {
      float b = dot(d1, d2);
      float denom = a * e - b * b;
      if (denom != 0.0f) s = clampf((b * f - c * e) / denom, 0.0f, 1.0f);
      else return false;
      float tt = b * s + f;
      if (tt < 0.0f) s = clampf(-c / a, 0.0f, 1.0f);
      else if (tt > e) s = clampf((b - c) / a, 0.0f, 1.0f);
    }
  }
  *p1 = a1 + d1 * s;
  return true;
}

// ---- rays ---------------------------------------------------------------------------
__device__ inline bool ray_sphere(V3 p, V3 d, const Sphere& s, V3* ip, float* tout, float dt = kInf) {  // :249-273; dt = Particle::DT
  V3 m = p - s.c;
  float a = mag2(d), b = dot(m, d), c = mag2(m) - s.r * s.r;
  if (c > 0.0f && b > 0.0f) return false;
  float discr = b * b - a * c;
  if (discr < 0.0f) return false;
  float t = fmax_rs((-b - __builtin_sqrtf(discr)) / a, 0.0f);
  if (t > dt) return false;
  *ip = p + t * d; *tout = t;
  return true;
}

__device__ inline bool ray_capsule(V3 p, V3 d, const Capsule& cap, V3* ip, float* tout, float dt = kInf) {  // :275-359
  V3 m = p - cap.a;
  float md = dot(m, cap.d), nd = dot(d, cap.d), dd = dot(cap.d, cap.d);
  float nn = mag2(d), mn = dot(m, d);
  float a = dd * nn - nd * nd;
  float k = mag2(m) - cap.r * cap.r;
  float t;
  if (fabs_rs(a) < kCollisionEps) {
    float b, c;
    if (md < 0.0f) { b = mn; c = k; }
    else if (md > dd) { V3 m2 = p - (cap.a + cap.d); b = dot(m2, d); c = mag2(m2) - cap.r * cap.r; }
    else return false;  // "Already colliding"
    if (c > 0.0f && b > 0.0f) return false;
    float discr = b * b - nn * c;
    if (discr < 0.0f) return false;
    t = fmax_rs((-b - __builtin_sqrtf(discr)) / nn, 0.0f);
    if (t > dt) return false;
    *ip = p + t * d; *tout = t;
    return true;
  }
  float c = dd * k - md * md;
  float b = dd * mn - nd * md;
  float discr = b * b - a * c;
  if (discr < 0.0f) return false;
  t = (-b - __builtin_sqrtf(discr)) / a;
  if (t < 0.0f) return false;
  if (md + t * nd < 0.0f) {
    if (mn > 0.0f && k > 0.0f) return false;
    float d2 = mn * mn - nn * k;
    if (d2 < 0.0f) return false;
    t = fmax_rs((-mn - __builtin_sqrtf(d2)) / nn, 0.0f);
  } else if (md + t * nd > dd) {
    V3 m2 = p - (cap.a + cap.d);
    float b2 = dot(m2, d), c2 = mag2(m2) - cap.r * cap.r;
    if (c2 > 0.0f && b2 > 0.0f) return false;
    float d2 = b2 * b2 - nn * c2;
    if (d2 < 0.0f) return false;
    t = fmax_rs((-b2 - __builtin_sqrtf(d2)) / nn, 0.0f);
  }
  if (t > dt) return false;
  *ip = p + t * d; *tout = t;
  return true;
}

// Intersects<Plane> :169-184, Intersects<Triangle> :186-200, Intersects<AABB> :202-236
__device__ inline bool ray_plane(V3 p, V3 d, const Plane& pl, V3* ip, float* tout, float dt = kInf) {
  float denom = dot(pl.n, d);
  if (denom == 0.0f) return false;
  float t = (pl.d - dot(pl.n, p)) / denom;
  if (t <= 0.0f || t > dt) return false;
  *ip = p + d * t; *tout = t;
  return true;
}
__device__ inline bool ray_triangle(V3 p, V3 d, const Triangle& tri, V3* ip, float* tout, float dt = kInf) {
  V3 q; float t;
  if (ray_plane(p, d, plane_from(tri.a, tri.b, tri.c), &q, &t, dt) && tri_contains(tri, q)) { *ip = q; *tout = t; return true; }
  return false;
}
__device__ inline bool ray_box(V3 p, V3 d, const Box& a, V3* ip, float* tout, float dt = kInf) {
  float t_min = 0.0f, t_max = kInf;
#pragma unroll
  for (int dim = 0; dim < 3; ++dim) {
    float pd = at(p, dim), dd = at(d, dim), ac = at(a.c, dim), ar = at(a.r, dim);
    if (fabs_rs(dd) < kCollisionEps) {
      if (fabs_rs(pd - ac) > ar) return false;
    } else {
      float ood = 1.0f / dd;
      float t1 = (ac - ar - pd) * ood;
      float t2 = (ac + ar - pd) * ood;
      if (t1 > t2) { t_min = fmax_rs(t_min, t2); t_max = fmin_rs(t_max, t1); }
      else { t_min = fmax_rs(t_min, t1); t_max = fmin_rs(t_max, t2); }
      if (t_min > t_max) return false;
    }
  }
  if (t_min > dt) return false;
  *ip = p + d * t_min; *tout = t_min;
  return true;
}

// ---- plane --------------------------------------------------------------------------
__device__ inline bool plane_msphere(const Plane& pl, const Sphere& s, V3 v, Contact* out) {  // :521-553
  float dist = dot(pl.n, s.c) - pl.d;
  if (fabs_rs(dist) <= s.r) {
    *out = mkc(s.c + -pl.n * dist, s.c + -pl.n * s.r, pl.n, 0.0f);
    return true;
  }
  float denom = dot(pl.n, v);
  if (denom * dist >= 0.0f) return false;
  float r = dist > 0.0f ? s.r : -s.r;
  float t = (r - dist) / denom;
  if (t <= 1.0f) {
    V3 q = s.c + t * v - r * pl.n;
    *out = mkc(q, q, pl.n, t);
    return true;
  }
  return false;
}

__device__ inline bool plane_mcapsule(const Plane& pl, const Capsule& c, V3 v, Contact* out) {  // :555-605
  float denom = dot(pl.n, normalize(c.d));
  V3 ctr;
  if (fabs_rs(denom) < kCollisionEps) {
    ctr = c.a + c.d * 0.5f;
  } else {
    float t = (pl.d - dot(pl.n, c.a)) / denom;
    if (t > 1.0f) ctr = c.a + c.d;
    else if (t < 0.0f) ctr = c.a;
    else {
      V3 q = c.a + c.d * t;
      float dist = dot(pl.n, c.a) - pl.d;
      V3 base = dist < 0.0f ? c.a : (c.a + c.d);
      *out = mkc(q, base + -pl.n * c.r, pl.n, 0.0f);
      return true;
    }
  }
  return plane_msphere(pl, mks(ctr, c.r), v, out);
}

// ---- triangle vs moving sphere :610-659 -----------------------------------------------
__device__ inline bool tri_msphere(const Triangle& tri, const Sphere& s, V3 v, Contact* out) {
  Plane p = plane_from(tri.a, tri.b, tri.c);
  Contact contact;
  if (!plane_msphere(p, s, v, &contact)) return false;
  if (tri_contains(tri, contact.a)) { *out = contact; return true; }
  float first_t = kInf;
  V3 tri_p = mk3(0.0f, 0.0f, 0.0f);
  if (mag2(v) == 0.0f) return false;
#pragma unroll 1
  for (int e = 0; e < 3; ++e) {
    V3 v1 = tri_vertex(tri, e), v2 = tri_vertex(tri, e == 2 ? 0 : e + 1);
    V3 ip; float it;
    if (ray_capsule(s.c, v, mkcap(v1, v2 - v1, s.r), &ip, &it)) {
      if (it <= 1.0f && it < first_t) { first_t = it; tri_p = seg_closest(v1, v2, ip); }
    }
  }
  if (first_t != kInf) { *out = mkc(tri_p, tri_p, p.n, first_t); return true; }
  return false;
}

// The same test by FOUR LANES that hold the same triangle, sphere and motion (k_terrain_contacts): the plane contact and the containment
// test by all of them, then lanes 0..2 an edge's ray-capsule test each instead of one lane running the three in a row (a sphere
// resting on the floor's other triangle goes all the way through them: ~5 us of one lane's arithmetic); the loop's choice - the
// smallest t <= 1, the earliest edge among equals - is replayed from the lanes' results.  Same operations on the same values: same bits.
// `e` = the lane's index in its group of four, `base` = the wave lane of the group's lane 0.
__device__ inline bool tri_msphere_x4(const Triangle& tri, const Sphere& s, V3 v, Contact* out, int e, int base) {
  Plane p = plane_from(tri.a, tri.b, tri.c);
  Contact contact;
  if (!plane_msphere(p, s, v, &contact)) return false;
  if (tri_contains(tri, contact.a)) { *out = contact; return true; }
  if (mag2(v) == 0.0f) return false;
  // (every lane of the group is here or none: the branches above depend on what the four share)
  bool ok = false;
  float it = kInf;
  V3 cp = mk3(0.0f, 0.0f, 0.0f);
  if (e < 3) {
    const V3 v1 = e == 0 ? tri.a : (e == 1 ? tri.b : tri.c), v2 = e == 0 ? tri.b : (e == 1 ? tri.c : tri.a);
    V3 ip;
    if (ray_capsule(s.c, v, mkcap(v1, v2 - v1, s.r), &ip, &it)) { ok = it <= 1.0f; if (ok) cp = seg_closest(v1, v2, ip); }
  }
  float first_t = kInf;
  V3 tri_p = mk3(0.0f, 0.0f, 0.0f);
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const bool ok_k = __shfl(ok ? 1 : 0, base + k) != 0;
    const float it_k = __shfl(it, base + k);
    const V3 cp_k = mk3(__shfl(cp.x, base + k), __shfl(cp.y, base + k), __shfl(cp.z, base + k));
    if (ok_k && it_k < first_t) { first_t = it_k; tri_p = cp_k; }
  }
  if (first_t != kInf) { *out = mkc(tri_p, tri_p, p.n, first_t); return true; }
  return false;
}

// seg_2d_intersect :667-688 (only t is used by the callers)
HD float area2d(V2 a, V2 b, V2 c) { return (a.x - c.x) * (b.y - c.y) - (a.y - c.y) * (b.x - c.x); }
HD bool seg2d(V2 a, V2 b, V2 c, V2 d, float* t) {
  float a1 = area2d(a, b, d), a2 = area2d(a, b, c);
  if (a1 * a2 <= 0.0f) {
    float a3 = area2d(c, d, a);
    float a4 = a3 + a2 - a1;
    if (a3 * a4 <= 0.0f) { *t = a3 / (a3 - a4); return true; }
  }
  return false;
}

// ---- triangle vs moving capsule :693-1086; returns number of contacts (0..2) ------------
// (the two contacts leave through references to two separate objects, not through an array: with `Contact out[2]` the compiler kept the
// pair in scratch memory - stored at every exit, loaded again by the caller: 96 bytes per lane in k_narrow_terrain<1>)
__device__ inline int tri_mcapsule(const Triangle& tri, const Capsule& c, V3 v, Contact& out0, Contact& out1) {
  const Plane p = plane_from(tri.a, tri.b, tri.c);
  // :698-719 capsule axis already crosses the face
  {
    float denom = dot(p.n, normalize(c.d));
    if (fabs_rs(denom) > kCollisionEps) {
      float t = (p.d - dot(p.n, c.a)) / denom;
      if (t <= 1.0f && t >= 0.0f) {
        V3 q = c.a + c.d * t;
        if (tri_contains(tri, q)) {
          V3 base = (dot(p.n, c.a) - p.d < 0.0f) ? c.a : (c.a + c.d);
          out0 = mkc(q, base + -p.n * c.r, p.n, 0.0f);
          out1 = out0;  // (every exit writes BOTH results - the second is not looked at when one contact is returned: with exits that wrote
                        // one or the other the compiler merged their stores through a selected pointer and kept the pair in scratch memory)
          return 1;
        }
      }
    }
  }
  // :723-764 plane contacts of the two end spheres
  bool found = false, checked = false;
  Contact fc = mkc(mk3(0, 0, 0), mk3(0, 0, 0), mk3(0, 0, 0), 0.0f);
  V3 dir = mk3(0.0f, 0.0f, 0.0f);
  {
    Contact c1, c2;
    bool h1 = plane_msphere(p, mks(c.a, c.r), v, &c1);
    if (h1) {
      bool h2 = plane_msphere(p, mks(c.a + c.d, c.r), v, &c2);
      if (h2) {
        if (c2.t < c1.t) { found = true; fc = c2; dir = -c.d; }
        else if (c2.t == 0.0f) {
          bool in1 = tri_contains(tri, c1.a), in2 = tri_contains(tri, c2.a);
          if (in1 && in2) { out0 = c2; out1 = c1; return 2; }
          else if (in1) { found = true; fc = c1; dir = c.d; checked = true; }
          else if (in2) { found = true; fc = c2; dir = -c.d; checked = true; }
        } else { found = true; fc = c1; dir = c.d; }
      } else { found = true; fc = c1; dir = c.d; }
    } else if (plane_msphere(p, mks(c.a + c.d, c.r), v, &c1)) {
      found = true; fc = c1; dir = -c.d;
    }
  }
  // :767-890 silhouette clipping in the plane rotated onto x-y
  if (found) {
    V3 sil_v = dir - p.n * dot(dir, p.n) / mag2(p.n);
    Quat rot = quat_from_arc(p.n, mk3(0.0f, 0.0f, 1.0f));
    V2 sa = xy(rotate(rot, fc.a + -p.n * p.d));
    V2 sb = xy(rotate(rot, fc.a + sil_v - p.n * p.d));
    bool inside = checked || tri_contains(tri, fc.a);
    bool parallel = fabs_rs(dot(dir, p.n)) < kCollisionEps;
    if (inside && !parallel) { out0 = fc; out1 = fc; return 1; }
    if (inside || (fc.t > 0.0f && parallel)) {
      float t_min = kInf, t_max = 0.0f;
      bool hit = false;
#pragma unroll 1
      for (int e = 0; e < 3; ++e) {
        V2 ea = xy(rotate(rot, tri_vertex(tri, e) - p.n * p.d));
        V2 eb = xy(rotate(rot, tri_vertex(tri, e == 2 ? 0 : e + 1) - p.n * p.d));
        float t;
        if (seg2d(sa, sb, ea, eb, &t)) {
          hit = true;
          if (t_min > t) t_min = t;
          if (t_max < t) t_max = t;
        }
      }
      float t_max2 = (t_max == 0.0f) ? 1.0f : t_max;
      if (inside) {  // :808-839 second contact for a face-parallel capsule
        V3 q = fc.a + sil_v * t_max2;
        out0 = fc;
        out1 = mkc(q, q, p.n, fc.t);
        return 2;
      }
      if (hit) {  // :847-888
        V3 q0 = fc.a + sil_v * t_min;
        V3 q1 = fc.a + sil_v * t_max2;
        out0 = mkc(q0, q0, p.n, fc.t);
        out1 = mkc(q1, q1, p.n, fc.t);
        return 2;
      }
    }
  }
  // :901-971 Minkowski-sum fallback: edges parallel to the capsule axis (exact test :915)
  uint32_t par_vert = 0;  // bitset.rs:40-55 on the 3 vertex bits
  float best_par_t = kInf;
  V3 best_par_a = mk3(0, 0, 0), best_par_b = mk3(0, 0, 0);
#pragma unroll 1
  for (int e = 0; e < 3; ++e) {
    int ia = e, ib = (e == 2 ? 0 : e + 1);
    V3 ea = tri_vertex(tri, ia), eb = tri_vertex(tri, ib);
    V3 ab = eb - ea;
    float ab_cd = dot(ab, c.d);
    if (fabs_rs(ab_cd) != mag(c.d) * mag(ab)) continue;
    par_vert |= (1u << ia) | (1u << ib);
    if (ab_cd < 0.0f) { V3 tmp = ea; ea = eb; eb = tmp; }
    float m_edge = mag2(ab);
    V3 ip; float it;
    if (ray_capsule(c.a, v, mkcap(ea, eb - ea, c.r), &ip, &it)) {
      if (it > fmin_rs(best_par_t, 1.0f)) continue;
      V3 tp = seg_closest(ea, eb, ip);
      float m_proj = mag2((tp + c.d) - ea);
      float c_t = (m_proj > m_edge) ? (m_proj - m_edge) / (m_proj - mag2(tp - ea)) : 1.0f;
      best_par_t = it; best_par_a = tp; best_par_b = tp + c.d * c_t;
    } else if (ray_capsule(c.a, v, mkcap(ea, -c.d, c.r), &ip, &it)) {
      if (it > fmin_rs(best_par_t, 1.0f)) continue;
      V3 d = ip - ea;
      float cap_t = -dot(d, c.d) / mag2(c.d);
      V3 tp = seg_closest(ea, ea + -c.d, ip);
      float m_proj = mag2((tp + c.d) - ea);
      best_par_t = it; best_par_a = tp + c.d * cap_t; best_par_b = (m_proj > m_edge) ? eb : (tp + c.d);
    }
  }
  // :973-1060 edge quads + vertex capsules
  float best_sum_t = kInf;
  V3 best_sum_p = mk3(0, 0, 0);
#pragma unroll 1
  for (int e = 0; e < 3; ++e) {
    int ia = e, ib = (e == 2 ? 0 : e + 1);
    bool a_par = (par_vert >> ia) & 1u, b_par = (par_vert >> ib) & 1u;
    if (a_par && b_par) continue;
    V3 ea = tri_vertex(tri, ia), eb = tri_vertex(tri, ib);
    Triangle q0 = mkt(ea + -c.d, ea, eb), q1 = mkt(ea + -c.d, eb, eb + -c.d);
    Plane p2 = plane_from(q1.a, q1.b, q1.c);
    Contact k;
    if (!plane_msphere(p2, mks(c.a, c.r), v, &k)) continue;
    if (best_sum_t > k.t && (tri_contains(q0, k.a) || tri_contains(q1, k.b))) {
      V3 d = k.a - ea;
      float cap_t = -dot(d, c.d) / mag2(c.d);
      best_sum_t = k.t; best_sum_p = k.a + c.d * cap_t;
    } else {
      V3 ip; float it;
      if (ray_capsule(c.a, v, mkcap(ea, eb - ea, c.r), &ip, &it)) {
        if (it <= 1.0f && it <= best_sum_t) { best_sum_t = it; best_sum_p = seg_closest(ea, eb, ip); }
      }
      if (ray_capsule(c.a, v, mkcap(ea + -c.d, eb - ea, c.r), &ip, &it)) {
        if (it <= 1.0f && it <= best_sum_t) { best_sum_t = it; best_sum_p = seg_closest(ea, eb, ip + c.d); }
      }
      if (!a_par && ray_capsule(c.a, v, mkcap(ea, -c.d, c.r), &ip, &it)) {
        if (it <= 1.0f && it <= best_sum_t) { best_sum_t = it; best_sum_p = ea; }
      }
      if (!b_par && ray_capsule(c.a, v, mkcap(eb, -c.d, c.r), &ip, &it)) {
        if (it <= 1.0f && it <= best_sum_t) { best_sum_t = it; best_sum_p = eb; }
      }
    }
  }
  // :1061-1085
  if (best_sum_t < best_par_t) { out0 = mkc(best_sum_p, best_sum_p, p.n, best_sum_t); out1 = out0; return 1; }
  if (best_par_t != kInf) {
    out0 = mkc(best_par_a, best_par_a, p.n, best_par_t);
    out1 = mkc(best_par_b, best_par_b, p.n, best_par_t);
    return 2;
  }
  out0 = out1 = mkc(mk3(0, 0, 0), mk3(0, 0, 0), mk3(0, 0, 0), 0.0f);
  return 0;
}

// ---- sphere / capsule pairs ---------------------------------------------------------
__device__ inline bool sphere_msphere(const Sphere& self, const Sphere& s, V3 v, Contact* out) {  // :1089-1141
  float r = self.r + s.r;
  V3 d = s.c - self.c;
  float len = mag2(d);
  if (len <= r * r) {
    V3 n;
    if (len == 0.0f) { if (is_zero(v)) return false; n = -normalize(v); }
    else n = d / __builtin_sqrtf(len);
    *out = mkc(self.c + n * self.r, s.c + -n * s.r, n, 0.0f);
    return true;
  }
  if (mag2(v) == 0.0f) return false;
  V3 ip; float it;
  if (ray_sphere(self.c, -v, mks(s.c, r), &ip, &it)) {
    if (it <= 1.0f) {
      V3 end_c = s.c + v * it;
      V3 ba = normalize(end_c - self.c);
      V3 a = self.c + ba * self.r;
      *out = mkc(a, a, ba, it);
      return true;
    }
  }
  return false;
}

__device__ inline bool capsule_msphere(const Capsule& self, const Sphere& s, V3 v, Contact* out) {  // :1145-1203
  float r = self.r + s.r;
  V3 cp = seg_closest(self.a, self.a + self.d, s.c);
  V3 d = s.c - cp;
  float len = mag2(d);
  if (len <= r * r) {
    V3 n;
    if (len == 0.0f) { if (is_zero(v)) return false; n = -normalize(v); }
    else n = d / __builtin_sqrtf(len);
    *out = mkc(cp + n * self.r, s.c + -n * s.r, n, 0.0f);
    return true;
  }
  if (mag2(v) == 0.0f) return false;
  V3 ip; float it;
  if (ray_capsule(s.c, v, mkcap(self.a, self.d, s.r + self.r), &ip, &it)) {
    if (it <= 1.0f) {
      V3 b = s.c + v * it;
      V3 a = seg_closest(self.a, self.a + self.d, b);
      V3 ba = normalize(b - a);
      V3 q = a + ba * self.r;
      *out = mkc(q, q, ba, it);
      return true;
    }
  }
  return false;
}

// Sphere.contacts(&Moving<Capsule>) = commute (:1143) over Moving<Capsule>.contacts(&Sphere) (:1368-1382)
__device__ inline bool sphere_mcapsule(const Sphere& self, const Capsule& c, V3 v, Contact* out) {
  Contact k;
  if (!capsule_msphere(c, self, -v, &k)) return false;
  V3 d = v * k.t;
  *out = neg(mkc(k.a + d, k.b + d, k.n, k.t));
  return true;
}

__device__ inline bool capsule_mcapsule(const Capsule& self, const Capsule& c, V3 v, Contact* out) {  // :1205-1356
  V3 sa = self.a, sb = self.a + self.d;
  V3 p1, p2;
  {
    V3 p, e;
    if (closest_pts_seg_first(sa, sb, c.a, c.a + v, &p)) {
      if (closest_pts_seg_first(sa, sb, c.a + c.d, c.a + c.d + v, &e)) { p1 = p; p2 = e; }
      else return false;
    } else { p1 = sa; p2 = sb; }
  }
  {
    V3 q;
    if (closest_pts_seg_first(p1, p2, c.a, c.a + c.d, &q)) return sphere_mcapsule(mks(q, self.r), c, v, out);
  }
  float d_mag2 = mag2(self.d);
  float t1 = dot(c.a - self.a, self.d) / d_mag2;
  float t2 = dot(c.a + c.d - self.a, self.d) / d_mag2;
  float t_min, t_max;
  V3 c_a, c_d;
  if (t1 < t2) { t_min = t1; t_max = t2; c_a = c.a; c_d = c.d; }
  else { t_min = t2; t_max = t1; c_a = c.a + c.d; c_d = -c.d; }
  V3 h = self.a - (c_a + c_d * (-t_min / (t_max - t_min)));
  float h_len = mag(h);
  V3 v_travel = mk3(0.0f, 0.0f, 0.0f);
  float coll_t = 0.0f;
  bool touching = h_len <= self.r + c.r;
  if (!touching) {
    float h_rat = (h_len - self.r - c.r) / h_len;
    float v_comp = dot(v, h) / (h_len * h_len);
    if (v_comp < h_rat) return false;
    coll_t = h_rat / v_comp;
    v_travel = v * coll_t;
    float axis_t_delta = dot(v_travel, self.d) / d_mag2;
    t_min = t_min + axis_t_delta;
    t_max = t_max + axis_t_delta;
  }
  if (t_max <= 0.0f) return capsule_msphere(self, mks(c_a + c_d, c.r), v, out);
  if (t_min >= 1.0f) return capsule_msphere(self, mks(c_a, c.r), v, out);
  float s_t = (clampf(t_min, 0.0f, 1.0f) + clampf(t_max, 0.0f, 1.0f)) * 0.5f;
  float o_t = (s_t - t_min) / (t_max - t_min);
  V3 a_c = self.a + self.d * s_t;
  V3 b_c = touching ? (c_a + c_d * o_t) : (c_a + c_d * o_t + v_travel);
  V3 ab = b_c - a_c;
  V3 n;
  if (is_zero(ab)) { if (is_zero(v)) return false; n = -normalize(v); }
  else n = normalize(b_c - a_c);
  *out = mkc(a_c + n * self.r, b_c + -n * c.r, n, touching ? 0.0f : coll_t);
  return true;
}

// ---- Moving<Component> pairs --------------------------------------------------------
// Moving<Component>.contacts(&Moving<Component>) (compound.rs:180-190 twice, then
// collision.rs:1387-1401): A's shape static, B sweeping at vB - vA, result shifted by vA*t;
// the two negations of compound.rs:186-187 cancel.  At most one contact.
// A cheap conservative reject before the reference's tests: every contact they report is a touching of the two shapes at some t
// in [0, 1] of B's relative motion (t is clamped at 0 or rejected below it: collision.rs:249-359, 1089-1356), so the shapes'
// bounding spheres come within reach of each other during the tick - with a margin of 1 % and a millimetre for rounding.  Nine
// candidates in ten of a pile of capsules or two-part bodies end here.
__device__ __forceinline__ bool comp_pair_far(const Comp& A, V3 vA, const Comp& B, V3 vB) {
  const bool sa = A.kind == KIND_SPHERE, sb = B.kind == KIND_SPHERE;
  const V3 ma = sa ? A.p : A.p + A.d * 0.5f, mb = sb ? B.p : B.p + B.d * 0.5f;
  const float ra = sa ? A.r : A.r + 0.5f * mag(A.d), rb = sb ? B.r : B.r + 0.5f * mag(B.d);
  const float lim = (ra + rb + mag(vB - vA)) * 1.01f + 1e-3f;
  const V3 dd = mb - ma;
  return dot(dd, dd) > lim * lim;
}
__device__ inline bool comp_pair_contact(const Comp& A, V3 vA, const Comp& B, V3 vB, Contact* out) {
  V3 vr = vB - vA;
  // (not for two spheres: their own test is as cheap, and the kernels that fuse it into the pair search reject on their own)
  if ((A.kind != KIND_SPHERE || B.kind != KIND_SPHERE) && comp_pair_far(A, vA, B, vB)) return false;
  Contact k;
  bool hit;
  if (A.kind == KIND_SPHERE) {
    if (B.kind == KIND_SPHERE) hit = sphere_msphere(mks(A.p, A.r), mks(B.p, B.r), vr, &k);
    else hit = sphere_mcapsule(mks(A.p, A.r), mkcap(B.p, B.d, B.r), vr, &k);
  } else {
    if (B.kind == KIND_SPHERE) hit = capsule_msphere(mkcap(A.p, A.d, A.r), mks(B.p, B.r), vr, &k);
    else hit = capsule_mcapsule(mkcap(A.p, A.d, A.r), mkcap(B.p, B.d, B.r), vr, &k);
  }
  if (!hit) return false;
  *out = mkc(k.a + vA * k.t, k.b + vA * k.t, k.n, k.t);
  return true;
}
// LocalContacts<Moving<Component>> for Moving<Component> compound.rs:192-207
__device__ inline bool comp_pair_local(const Comp& A, V3 vA, const Comp& B, V3 vB, LocalContact* out) {
  Contact c;
  if (!comp_pair_contact(A, vA, B, vB, &c)) return false;
  out->la = c.a + -(comp_center(A) + vA * c.t);
  out->lb = c.b + -(comp_center(B) + vB * c.t);
  out->g = c;
  return true;
}
// Body vs one terrain face.  mesh.rs:115-139 (tri rebuilt at verts + x; a/b swapped, n negated)
// over compound.rs:180-190 (negated once) gives the Mesh-side contact k = raw Triangle contact;
// collision.rs:1490-1506 then forms the LocalContact with global = -k.
// `centre`: the body's centre the local point is taken from (the component's own for an ordinary body; the centre of
// mass for a part of a body of several components).
__device__ inline int comp_tri_local_at(const Comp& A, V3 vA, const Triangle& tri, V3 mesh_center, V3 centre, LocalContact out[2]) {
  Contact raw0, raw1;
  raw0 = raw1 = mkc(mk3(0, 0, 0), mk3(0, 0, 0), mk3(0, 0, 0), 0.0f);
  int n;
  if (A.kind == KIND_SPHERE) n = tri_msphere(tri, mks(A.p, A.r), vA, &raw0) ? 1 : 0;
  else n = tri_mcapsule(tri, mkcap(A.p, A.d, A.r), vA, raw0, raw1);
  // Mesh::contacts callback value: a on the mesh, b on the body, n = face normal
  if (n > 0) { const V3 a_c = centre + vA * raw0.t; out[0].la = raw0.b + -a_c; out[0].lb = raw0.a + -mesh_center; out[0].g = neg(raw0); }
  if (n > 1) { const V3 a_c = centre + vA * raw1.t; out[1].la = raw1.b + -a_c; out[1].lb = raw1.a + -mesh_center; out[1].g = neg(raw1); }
  return n;
}
__device__ inline int comp_tri_local(const Comp& A, V3 vA, const Triangle& tri, V3 mesh_center, LocalContact out[2]) {
  return comp_tri_local_at(A, vA, tri, mesh_center, comp_center(A), out);
}

// A cheap conservative reject before the reference's body-triangle tests (collision.rs:610-1086), as comp_pair_far is for pairs.  Every
// contact those tests report - the axis crossing the face (:698-719), the end spheres' plane contacts inside the face (:723-764), the
// silhouette clipped against the edges (:767-890), the Minkowski faces, edge and vertex capsules (:901-1060), the sphere's plane contact
// and edge capsules (:610-659) - is a touching, at some t in [0, 1] of the body's motion, of the body's axis (a sphere: its centre)
// inflated by r with a point OF THE TRIANGLE.  The triangle lies on the inner side of the plane through each of its edges perpendicular
// to the face, and in the face's plane: a body whose axis end points are both further than r + |v| (1 % and a millimetre for rounding)
// beyond one of those four planes, on the same side, never touches it.  The candidates that fail here are the ones that would have
// walked through every branch of tri_mcapsule to report nothing.
__device__ __forceinline__ bool comp_tri_far(const Comp& A, V3 vA, const Triangle& tri) {
  // A capsule that does not move at all is never dropped: with v = 0 the reference's ray-capsule test (collision.rs:275-359) takes its
  // parallel branch and, where the ray's origin lies beyond an end of the edge, evaluates (-b - sqrt(discr)) / |v|^2 = 0 / 0 -> max(NaN, 0) = 0:
  // "a hit at t = 0" whatever the distance - a static capsule reports a contact with a triangle it is a radius and a half away from
  // (tests/test_gpu_tri_reject.py found it; a sphere is guarded: tri_msphere returns at |v| = 0, collision.rs:640).
  if (A.kind != KIND_SPHERE && mag2(vA) == 0.0f) return false;
  const V3 p0 = A.p, p1 = A.kind == KIND_SPHERE ? A.p : A.p + A.d;
  // (... of the reference's ARITHMETIC, which is f32 relative to the far end of an edge: ray_capsule forms |m|^2 |D|^2 - (m.D)^2 with m from
  // the edge's start - 200 m away on the floor of config 5's box - and whether a capsule 0.31 from the floor's diagonal touches it comes out
  // of the last bits (it did: tools/r06/dbg_c5.py).  The reach below grows with the square of the distance to the face's farthest vertex:
  // nothing beside a 2 m triangle, 0.7 instead of 0.3 beside a 200 m one.)
  const float far2 = fmax_rs(fmax_rs(mag2(p0 - tri.a), mag2(p0 - tri.b)), mag2(p0 - tri.c)) + mag2(p1 - p0);
  const float lim0 = (A.r + mag(vA)) * 1.01f + 1e-3f;
  const float lim = __builtin_sqrtf(lim0 * lim0 + 1e-5f * far2);
  const V3 e0 = tri.b - tri.a, e1 = tri.c - tri.b, e2 = tri.a - tri.c;
  const V3 n = cross(e0, tri.c - tri.a);
  const float nn = dot(n, n);
  if (!(nn > 0.0f)) return false;  // (a degenerate face: the reference's tests decide)
  {
    // ... and a needle or a sliver (an angle under 2 degrees): tri_contains' determinant |ab|^2 |ac|^2 - (ab.ac)^2 is then the rounding of its two
    // terms and the reference finds points well outside the face "inside" - its tests decide (the fuzz found contacts 6 m from a 800 m sliver)
    const float a0 = dot(e0, e0), a1 = dot(e1, e1), a2 = dot(e2, e2);
    if (!(nn >= 1e-3f * fmax_rs(fmax_rs(a0 * a1, a1 * a2), a2 * a0))) return false;
  }
  // The reference's first test of a capsule (collision.rs:698-719, "the axis already crosses the face") finds the crossing at t = (d - n.a) /
  // (n . normalize(axis)) - a LENGTH along the axis - and accepts 0 <= t <= 1 as if t were the parameter: a capsule shorter than 1 that points
  // at the face from 0.2 above it "crosses" it where its axis, carried on to length 1, would (the contact: c.a + c.d t, over the face's inside).
  // The tests on the face's plane below therefore take the axis carried on to length 1 (and 2 % for rounding) where it is shorter.
  V3 p1x = p1;
  if (A.kind != KIND_SPHERE) {
    const float len2 = mag2(A.d);
    if (len2 < 1.02f * 1.02f && len2 > 0.0f) p1x = p0 + A.d * (1.02f / __builtin_sqrtf(len2));
  }
  {
    const float l = lim * __builtin_sqrtf(nn);
    const float d0 = dot(p0 - tri.a, n), d1 = dot(p1x - tri.a, n);
    if ((d0 > l && d1 > l) || (d0 < -l && d1 < -l)) return true;
  }
  // cross(edge, n) points away from the triangle whatever its winding (n turns with it)
  const V3 m0 = cross(e0, n), m1 = cross(e1, n), m2 = cross(e2, n);
  const float l0 = lim * __builtin_sqrtf(dot(m0, m0)), l1 = lim * __builtin_sqrtf(dot(m1, m1)), l2 = lim * __builtin_sqrtf(dot(m2, m2));
  if (dot(p0 - tri.a, m0) > l0 && dot(p1 - tri.a, m0) > l0) return true;
  if (dot(p0 - tri.b, m1) > l1 && dot(p1 - tri.b, m1) > l1) return true;
  if (dot(p0 - tri.c, m2) > l2 && dot(p1 - tri.c, m2) > l2) return true;
  // ... and the distance itself, where the axis stays on one side of the face: the closest pair of a segment and a triangle it does not
  // cross is an end point over the face's inside or a pair of points of the segment and an edge.  (A body that lies on the NEXT face, half a
  // radius from this one's edge, passes the four planes above and touches nothing here: two candidates in three of a capsule field at rest.)
  {
    const float d0 = dot(p0 - tri.a, n), d1 = dot(p1 - tri.a, n), d1x = dot(p1x - tri.a, n);
    if (!((d0 > 0.0f && d1x > 0.0f) || (d0 < 0.0f && d1x < 0.0f))) return false;  // (the axis - carried on - meets the plane: the reference's tests decide)
    const float lim2 = lim * lim;
    // an end point whose projection lies inside the face (or on its rim: then an edge is as near) is |d| / |n| from it
    const bool in0 = dot(p0 - tri.a, m0) <= 0.0f && dot(p0 - tri.b, m1) <= 0.0f && dot(p0 - tri.c, m2) <= 0.0f;
    const bool in1 = dot(p1 - tri.a, m0) <= 0.0f && dot(p1 - tri.b, m1) <= 0.0f && dot(p1 - tri.c, m2) <= 0.0f;
    if (in0 && d0 * d0 <= lim2 * nn) return false;
    if (in1 && d1 * d1 <= lim2 * nn) return false;
    const V3 ax = p1 - p0;
    float best = kInf;
    bool sure = true;
#pragma unroll
    for (int k = 0; k < 3; ++k) {  // segment - segment (Ericson, Real-Time Collision Detection 5.1.9), the edges one after the other
      const V3 q = k == 0 ? tri.a : (k == 1 ? tri.b : tri.c), e = k == 0 ? e0 : (k == 1 ? e1 : e2);
      const V3 r = p0 - q;
      const float a = dot(ax, ax), ee = dot(e, e), f = dot(e, r);
      float sa = 0.0f, tb = 0.0f;
      if (!(ee > 0.0f)) { sure = false; continue; }
      if (a > 0.0f) {
        const float c = dot(ax, r), b = dot(ax, e), den = a * ee - b * b;
        if (!(den > 1e-3f * a * ee)) { sure = false; continue; }  // (all but parallel: the closest pair is ill-conditioned - no verdict from here)
        sa = clampf((b * f - c * ee) / den, 0.0f, 1.0f);
        tb = (b * sa + f) / ee;
        if (tb < 0.0f) { tb = 0.0f; sa = clampf(-c / a, 0.0f, 1.0f); }
        else if (tb > 1.0f) { tb = 1.0f; sa = clampf((b - c) / a, 0.0f, 1.0f); }
      } else {
        tb = clampf(f / ee, 0.0f, 1.0f);
      }
      const V3 dd = (p0 + ax * sa) - (q + e * tb);
      best = fmin_rs(best, dot(dd, dd));
    }
    return sure && best > lim2;
  }
}

// compute_basis geom.rs:1138-1145
HD void compute_basis(V3 n, V3* t0, V3* t1) {
  V3 b = (fabs_rs(n.x) >= 0.57735f) ? mk3(n.y, -n.x, 0.0f) : mk3(0.0f, n.z, -n.y);
  b = normalize(b);
  *t0 = b;
  *t1 = cross(n, b);
}

}  // namespace mgf
