// Device/host f32 math for the mgf hot path on gfx950.
//
// Every function reproduces, operation for operation, the cgmath 0.17 arithmetic the
// reference calls (reference Cargo.toml:20; call sites cited per function).  The file is
// compiled with -ffp-contract=off so hipcc never fuses a*b+c: results are bit-identical
// to the Rust original's IEEE f32 sequence.  HD functions also run on the host for the
// setup-time work (add_body tensors).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define HD __host__ __device__ __forceinline__

namespace mgf {

constexpr float kInf = __builtin_huge_valf();
constexpr float kF32Eps = 1.1920929e-7f;
constexpr float kCollisionEps = 0.000001f;  // geom.rs:27

struct V3 { float x, y, z; };
HD V3 mk3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
HD V3 operator+(V3 a, V3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
HD V3 operator-(V3 a, V3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
HD V3 operator-(V3 a) { return mk3(-a.x, -a.y, -a.z); }
HD V3 operator*(V3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
HD V3 operator*(float s, V3 a) { return mk3(s * a.x, s * a.y, s * a.z); }
HD V3 operator/(V3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }
// cgmath InnerSpace::dot = (x*x' + y*y') + z*z'
HD float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
HD V3 cross(V3 a, V3 b) { return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
HD float mag2(V3 a) { return dot(a, a); }
HD float mag(V3 a) { return __builtin_sqrtf(dot(a, a)); }
HD V3 normalize(V3 a) { return a * (1.0f / mag(a)); }  // normalize_to(1): v * (1/|v|)
HD bool is_zero(V3 a) { return a.x == 0.0f && a.y == 0.0f && a.z == 0.0f; }
HD float at(V3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }
HD float fmax_rs(float a, float b) { return __builtin_fmaxf(a, b); }  // f32::max
HD float fmin_rs(float a, float b) { return __builtin_fminf(a, b); }  // f32::min
HD float fabs_rs(float a) { return __builtin_fabsf(a); }

struct V2 { float x, y; };
HD V2 mk2(float x, float y) { V2 r; r.x = x; r.y = y; return r; }
HD V2 operator+(V2 a, V2 b) { return mk2(a.x + b.x, a.y + b.y); }
HD V2 operator-(V2 a, V2 b) { return mk2(a.x - b.x, a.y - b.y); }
HD V2 operator*(float s, V2 a) { return mk2(s * a.x, s * a.y); }
HD V2 xy(V3 a) { return mk2(a.x, a.y); }

// approx::ulps_eq for f32 (epsilon = f32::EPSILON, max_ulps = 4) as cgmath's from_arc uses it.
HD bool ulps_eq(float a, float b) {
  if (fabs_rs(a - b) <= kF32Eps) return true;
  if (__builtin_signbit(a) != __builtin_signbit(b)) return false;
  int32_t ia = __builtin_bit_cast(int32_t, a), ib = __builtin_bit_cast(int32_t, b);
  int64_t d = (int64_t)ia - (int64_t)ib;
  if (d < 0) d = -d;
  return d <= 4;
}

struct Quat { float s; V3 v; };
HD Quat mkq(float s, V3 v) { Quat q; q.s = s; q.v = v; return q; }
HD Quat operator+(Quat a, Quat b) { return mkq(a.s + b.s, a.v + b.v); }
HD Quat operator*(Quat a, float f) { return mkq(a.s * f, a.v * f); }
HD Quat operator*(Quat l, Quat r) {  // cgmath Quaternion * Quaternion (scalar path)
  return mkq(l.s * r.s - l.v.x * r.v.x - l.v.y * r.v.y - l.v.z * r.v.z,
             mk3(l.s * r.v.x + l.v.x * r.s + l.v.y * r.v.z - l.v.z * r.v.y,
                 l.s * r.v.y + l.v.y * r.s + l.v.z * r.v.x - l.v.x * r.v.z,
                 l.s * r.v.z + l.v.z * r.s + l.v.x * r.v.y - l.v.y * r.v.x));
}
HD float dot(Quat a, Quat b) { return a.s * b.s + dot(a.v, b.v); }
HD Quat normalize(Quat a) { return a * (1.0f / __builtin_sqrtf(dot(a, a))); }
// Rotation::rotate_vector (Quaternion * Vector3): tmp = v x r + r*s; (v x tmp)*2 + r
HD V3 rotate(Quat q, V3 r) {
  V3 tmp = cross(q.v, r) + (r * q.s);
  return (cross(q.v, tmp) * 2.0f) + r;
}
// Quaternion::from_arc(src, dst, None) — physics.rs:70, compound.rs:48, collision.rs:782.
HD Quat quat_from_arc(V3 src, V3 dst) {
  float mag_avg = __builtin_sqrtf(mag2(src) * mag2(dst));
  float d = dot(src, dst);
  if (ulps_eq(d, mag_avg)) return mkq(1.0f, mk3(0.0f, 0.0f, 0.0f));
  if (ulps_eq(d, -mag_avg)) {
    V3 v = cross(mk3(1.0f, 0.0f, 0.0f), src);
    if (ulps_eq(v.x, 0.0f) && ulps_eq(v.y, 0.0f) && ulps_eq(v.z, 0.0f)) v = cross(mk3(0.0f, 1.0f, 0.0f), src);
    V3 axis = normalize(v);
    // from_axis_angle(axis, Rad(pi)): (sin, cos)(pi/2 in f32) = (1.0, -4.371139e-8)
    return mkq(-4.371139e-8f, axis * 1.0f);
  }
  return normalize(mkq(mag_avg + d, cross(src, dst)));
}

struct M3 { V3 c[3]; };  // column-major like cgmath Matrix3
HD M3 m3_cols(V3 a, V3 b, V3 c) { M3 m; m.c[0] = a; m.c[1] = b; m.c[2] = c; return m; }
HD M3 m3_diag(float a, float b, float c) { return m3_cols(mk3(a, 0, 0), mk3(0, b, 0), mk3(0, 0, c)); }
HD V3 m3_row(const M3& m, int r) { return mk3(at(m.c[0], r), at(m.c[1], r), at(m.c[2], r)); }
HD V3 operator*(const M3& m, V3 v) {  // row(r) . v
  return mk3(m.c[0].x * v.x + m.c[1].x * v.y + m.c[2].x * v.z, m.c[0].y * v.x + m.c[1].y * v.y + m.c[2].y * v.z,
             m.c[0].z * v.x + m.c[1].z * v.y + m.c[2].z * v.z);
}
HD M3 operator*(const M3& l, const M3& r) { return m3_cols(l * r.c[0], l * r.c[1], l * r.c[2]); }
HD M3 operator*(const M3& m, float s) { return m3_cols(m.c[0] * s, m.c[1] * s, m.c[2] * s); }
HD M3 operator*(float s, const M3& m) { return m3_cols(s * m.c[0], s * m.c[1], s * m.c[2]); }
HD M3 operator+(const M3& a, const M3& b) { return m3_cols(a.c[0] + b.c[0], a.c[1] + b.c[1], a.c[2] + b.c[2]); }
HD M3 operator-(const M3& a, const M3& b) { return m3_cols(a.c[0] - b.c[0], a.c[1] - b.c[1], a.c[2] - b.c[2]); }
HD M3 transpose(const M3& m) { return m3_cols(m3_row(m, 0), m3_row(m, 1), m3_row(m, 2)); }
HD M3 m3_from_quat(Quat q) {  // Matrix3::from(Quaternion) — physics.rs:231
  float x2 = q.v.x + q.v.x, y2 = q.v.y + q.v.y, z2 = q.v.z + q.v.z;
  float xx2 = x2 * q.v.x, xy2 = x2 * q.v.y, xz2 = x2 * q.v.z;
  float yy2 = y2 * q.v.y, yz2 = y2 * q.v.z, zz2 = z2 * q.v.z;
  float sy2 = y2 * q.s, sz2 = z2 * q.s, sx2 = x2 * q.s;
  return m3_cols(mk3(1.0f - yy2 - zz2, xy2 + sz2, xz2 - sy2), mk3(xy2 - sz2, 1.0f - xx2 - zz2, yz2 + sx2),
                 mk3(xz2 + sy2, yz2 - sx2, 1.0f - xx2 - yy2));
}
HD float determinant(const M3& m) {
  return m.c[0].x * (m.c[1].y * m.c[2].z - m.c[2].y * m.c[1].z) - m.c[1].x * (m.c[0].y * m.c[2].z - m.c[2].y * m.c[0].z) +
         m.c[2].x * (m.c[0].y * m.c[1].z - m.c[1].y * m.c[0].z);
}
HD bool invert(const M3& m, M3* out) {  // SquareMatrix::invert — physics.rs:212
  float det = determinant(m);
  if (det == 0.0f) return false;
  *out = transpose(m3_cols(cross(m.c[1], m.c[2]) / det, cross(m.c[2], m.c[0]) / det, cross(m.c[0], m.c[1]) / det));
  return true;
}

}  // namespace mgf
