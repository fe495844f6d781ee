// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the shipped product.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
// anything under oracle/.  The product (mgf_amd/) never includes this file.
//
// CPU restatement of the arithmetic of the third-party crate `cgmath = "0.17"`
// (reference Cargo.toml:20) as it is used on mgf's hot path.  cgmath's source is
// NOT vendored under /root/reference, so these formulas are restated from the
// published cgmath 0.17.0 algorithms; the call sites that anchor them are cited
// per function.  PARITY STATUS: quaternion integration, Matrix3::from(quat),
// Matrix3::invert and from_arc are "parity unpinned" except where a reference
// unit test goes through them (collision.rs:2012-2103 via from_arc/rotate_vector,
// exact-f32 asserts at collision.rs:2077 and :2103).
//
// Build rule: compile with -ffp-contract=off and without -ffast-math so every
// f32 operation rounds exactly as the Rust original (Rust never fuses a*b+c).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>

namespace mgfo {

static constexpr float F32_INF = std::numeric_limits<float>::infinity();
static constexpr float F32_EPS = 1.1920929e-7f;  // f32::EPSILON

// ---------------------------------------------------------------------------
// Vector3<f32> / Point3<f32>.  cgmath keeps two types; every operation mgf uses
// on them is the same element-wise arithmetic, so one struct serves both.
// ---------------------------------------------------------------------------
struct V3 {
  float x, y, z;
};
static inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
static inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
static inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
static inline V3 operator*(float s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
static inline V3 operator/(V3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
static inline V3& operator+=(V3& a, V3 b) { a = a + b; return a; }
static inline V3& operator-=(V3& a, V3 b) { a = a - b; return a; }
static inline bool operator==(V3 a, V3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
// InnerSpace::dot for Vector3 = mul_element_wise(...).sum() = (x + y) + z.
static inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline V3 cross(V3 a, V3 b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
static inline float magnitude2(V3 a) { return dot(a, a); }
static inline float magnitude(V3 a) { return std::sqrt(dot(a, a)); }
// InnerSpace::normalize = normalize_to(1) = self * (1 / magnitude).
static inline V3 normalize(V3 a) { return a * (1.0f / magnitude(a)); }
// Zero::is_zero = (*self == zero()) — exact comparison.
static inline bool is_zero(V3 a) { return a.x == 0.0f && a.y == 0.0f && a.z == 0.0f; }
static inline float idx(V3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

struct V2 {
  float x, y;
};
static inline V2 operator+(V2 a, V2 b) { return {a.x + b.x, a.y + b.y}; }
static inline V2 operator-(V2 a, V2 b) { return {a.x - b.x, a.y - b.y}; }
static inline V2 operator*(float s, V2 a) { return {s * a.x, s * a.y}; }
static inline V2 truncate(V3 a) { return {a.x, a.y}; }

// ---------------------------------------------------------------------------
// approx 0.3 `ulps_eq!` / `relative_eq!` for f32 (defaults: epsilon = f32::EPSILON,
// max_ulps = 4, max_relative = f32::EPSILON).  Used by cgmath's from_arc and by
// mgf's Plane/Rectangle::contains (collision.rs:81,108).
// ---------------------------------------------------------------------------
static inline bool ulps_eq(float a, float b, float epsilon = F32_EPS, uint32_t max_ulps = 4) {
  if (std::fabs(a - b) <= epsilon) return true;
  if (std::signbit(a) != std::signbit(b)) return false;
  int32_t ia, ib;
  std::memcpy(&ia, &a, 4);
  std::memcpy(&ib, &b, 4);
  int64_t d = (int64_t)ia - (int64_t)ib;
  if (d < 0) d = -d;
  return (uint64_t)d <= max_ulps;
}
static inline bool relative_eq(float a, float b, float epsilon, float max_relative = F32_EPS) {
  if (a == b) return true;
  if (std::isinf(a) || std::isinf(b)) return false;
  float abs_diff = std::fabs(a - b);
  if (abs_diff <= epsilon) return true;
  float aa = std::fabs(a), ab = std::fabs(b);
  float largest = ab > aa ? ab : aa;
  return abs_diff <= largest * max_relative;
}

// ---------------------------------------------------------------------------
// Quaternion<f32> { s, v }.
// ---------------------------------------------------------------------------
struct Quat {
  float s;
  V3 v;
};
static inline Quat quat_from_sv(float s, V3 v) { return Quat{s, v}; }
static inline Quat quat_one() { return Quat{1.0f, {0.0f, 0.0f, 0.0f}}; }
static inline Quat operator+(Quat a, Quat b) { return {a.s + b.s, a.v + b.v}; }
static inline Quat operator*(Quat a, float f) { return {a.s * f, a.v * f}; }
// Quaternion * Quaternion (cgmath quaternion.rs, non-SIMD path).
static inline Quat operator*(Quat l, Quat r) {
  return Quat{
      l.s * r.s - l.v.x * r.v.x - l.v.y * r.v.y - l.v.z * r.v.z,
      {l.s * r.v.x + l.v.x * r.s + l.v.y * r.v.z - l.v.z * r.v.y,
       l.s * r.v.y + l.v.y * r.s + l.v.z * r.v.x - l.v.x * r.v.z,
       l.s * r.v.z + l.v.z * r.s + l.v.x * r.v.y - l.v.y * r.v.x}};
}
// InnerSpace for Quaternion: dot = s*s' + v.dot(v').
static inline float dot(Quat a, Quat b) { return a.s * b.s + dot(a.v, b.v); }
static inline float magnitude(Quat a) { return std::sqrt(dot(a, a)); }
static inline Quat normalize(Quat a) { return a * (1.0f / magnitude(a)); }
// Rotation::rotate_vector = Quaternion * Vector3:
//   tmp = v.cross(rhs) + rhs * s;  (v.cross(tmp) * 2) + rhs
static inline V3 rotate_vector(Quat q, V3 r) {
  V3 tmp = cross(q.v, r) + (r * q.s);
  return (cross(q.v, tmp) * 2.0f) + r;
}
static inline Quat conjugate(Quat q) { return {q.s, -q.v}; }

// Rotation3::from_axis_angle(axis, Rad(angle)): (s, c) = sin_cos(angle * 0.5).
static inline Quat quat_from_axis_angle(V3 axis, float angle_rad) {
  float h = angle_rad * 0.5f;
  float s = std::sin(h), c = std::cos(h);
  return quat_from_sv(c, axis * s);
}

// Quaternion::from_arc(src, dst, None) (cgmath quaternion.rs).
// Call sites: physics.rs:70, compound.rs:48, collision.rs:782.
static inline Quat quat_from_arc(V3 src, V3 dst) {
  float mag_avg = std::sqrt(magnitude2(src) * magnitude2(dst));
  float d = dot(src, dst);
  if (ulps_eq(d, mag_avg)) {
    return quat_one();
  } else if (ulps_eq(d, -mag_avg)) {
    V3 v = cross(v3(1.0f, 0.0f, 0.0f), src);
    if (ulps_eq(v.x, 0.0f) && ulps_eq(v.y, 0.0f) && ulps_eq(v.z, 0.0f)) {
      v = cross(v3(0.0f, 1.0f, 0.0f), src);
    }
    V3 axis = normalize(v);
    return quat_from_axis_angle(axis, 3.14159265358979323846f);  // Rad::turn_div_2()
  } else {
    return normalize(quat_from_sv(mag_avg + d, cross(src, dst)));
  }
}

// ---------------------------------------------------------------------------
// Matrix3<f32>, column-major: c[col] is a column vector.
// ---------------------------------------------------------------------------
struct M3 {
  V3 c[3];
};
// Matrix3::new(c0r0, c0r1, c0r2, c1r0, ...)
static inline M3 m3_new(float c0r0, float c0r1, float c0r2, float c1r0, float c1r1, float c1r2,
                        float c2r0, float c2r1, float c2r2) {
  return M3{{{c0r0, c0r1, c0r2}, {c1r0, c1r1, c1r2}, {c2r0, c2r1, c2r2}}};
}
static inline M3 m3_from_cols(V3 a, V3 b, V3 c) { return M3{{a, b, c}}; }
static inline M3 m3_zero() { return m3_new(0, 0, 0, 0, 0, 0, 0, 0, 0); }
static inline M3 m3_one() { return m3_new(1, 0, 0, 0, 1, 0, 0, 0, 1); }
static inline V3 m3_row(const M3& m, int r) { return {idx(m.c[0], r), idx(m.c[1], r), idx(m.c[2], r)}; }
// Matrix3 * Vector3 = (row(r) . v): (m[0][r]*x + m[1][r]*y) + m[2][r]*z.
static inline V3 operator*(const M3& m, V3 v) {
  return {dot(m3_row(m, 0), v), dot(m3_row(m, 1), v), dot(m3_row(m, 2), v)};
}
// Matrix3 * Matrix3: element (r,c) = lhs.row(r).dot(rhs[c]).
static inline M3 operator*(const M3& l, const M3& r) { return M3{{l * r.c[0], l * r.c[1], l * r.c[2]}}; }
static inline M3 operator*(const M3& m, float s) { return M3{{m.c[0] * s, m.c[1] * s, m.c[2] * s}}; }
static inline M3 operator*(float s, const M3& m) { return M3{{s * m.c[0], s * m.c[1], s * m.c[2]}}; }
static inline M3 operator+(const M3& a, const M3& b) { return M3{{a.c[0] + b.c[0], a.c[1] + b.c[1], a.c[2] + b.c[2]}}; }
static inline M3 operator-(const M3& a, const M3& b) { return M3{{a.c[0] - b.c[0], a.c[1] - b.c[1], a.c[2] - b.c[2]}}; }
static inline M3 transpose(const M3& m) { return M3{{m3_row(m, 0), m3_row(m, 1), m3_row(m, 2)}}; }
// Matrix3::from(Quaternion) (cgmath quaternion.rs From<Quaternion> for Matrix3).
// Call site: physics.rs:231.
static inline M3 m3_from_quat(Quat q) {
  float x2 = q.v.x + q.v.x, y2 = q.v.y + q.v.y, z2 = q.v.z + q.v.z;
  float xx2 = x2 * q.v.x, xy2 = x2 * q.v.y, xz2 = x2 * q.v.z;
  float yy2 = y2 * q.v.y, yz2 = y2 * q.v.z, zz2 = z2 * q.v.z;
  float sy2 = y2 * q.s, sz2 = z2 * q.s, sx2 = x2 * q.s;
  return m3_new(1.0f - yy2 - zz2, xy2 + sz2, xz2 - sy2,
                xy2 - sz2, 1.0f - xx2 - zz2, yz2 + sx2,
                xz2 + sy2, yz2 - sx2, 1.0f - xx2 - yy2);
}
// SquareMatrix::determinant / invert for Matrix3.  Call site: physics.rs:212.
static inline float determinant(const M3& m) {
  return m.c[0].x * (m.c[1].y * m.c[2].z - m.c[2].y * m.c[1].z) -
         m.c[1].x * (m.c[0].y * m.c[2].z - m.c[2].y * m.c[0].z) +
         m.c[2].x * (m.c[0].y * m.c[1].z - m.c[1].y * m.c[0].z);
}
static inline bool invert(const M3& m, M3* out) {
  float det = determinant(m);
  if (det == 0.0f) return false;
  *out = transpose(m3_from_cols(cross(m.c[1], m.c[2]) / det, cross(m.c[2], m.c[0]) / det,
                                cross(m.c[0], m.c[1]) / det));
  return true;
}

}  // namespace mgfo
