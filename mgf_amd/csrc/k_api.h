// Single-shot and batch entry points behind the C-ABI: golden-vector shapes, Compound, raytrace, manifolds.  (Part of the kernel set described in kernels.h.)
#pragma once
#include "k_tiles.h"

namespace mgf {

// ------------------------------------------------------------------------------------------
// Single-shot entry points (golden-vector parity through the C-ABI): one lane per problem.
// ------------------------------------------------------------------------------------------
struct ShapeIn { int kind; float v[12]; };
struct ContactOut { float a[3], b[3], n[3], t; };
__device__ __forceinline__ ContactOut to_out(const Contact& c) {
  ContactOut o; st3(o.a, c.a); st3(o.b, c.b); st3(o.n, c.n); o.t = c.t; return o;
}

// Contacts::contacts for (a [moving]) vs (b [moving]); mirrors the reference's trait resolution.
__device__ inline int contacts_dispatch(const ShapeIn& a, bool ma, V3 va, const ShapeIn& b, bool mb, V3 vb, Contact out[2]) {
  auto S = [](const ShapeIn& s) { return mks(mk3(s.v[0], s.v[1], s.v[2]), s.v[3]); };
  auto Cp = [](const ShapeIn& s) { return mkcap(mk3(s.v[0], s.v[1], s.v[2]), mk3(s.v[3], s.v[4], s.v[5]), s.v[6]); };
  auto Tr = [](const ShapeIn& s) { return mkt(mk3(s.v[0], s.v[1], s.v[2]), mk3(s.v[3], s.v[4], s.v[5]), mk3(s.v[6], s.v[7], s.v[8])); };
  auto Pl = [](const ShapeIn& s) { Plane p; p.n = mk3(s.v[0], s.v[1], s.v[2]); p.d = s.v[3]; return p; };
  const int ka = a.kind, kb = b.kind;
  if (!ma && mb) {  // static receiver, moving argument
    if (kb == MGF_SPHERE) {
      if (ka == MGF_SPHERE) return sphere_msphere(S(a), S(b), vb, out) ? 1 : 0;
      if (ka == MGF_CAPSULE) return capsule_msphere(Cp(a), S(b), vb, out) ? 1 : 0;
      if (ka == MGF_TRIANGLE) return tri_msphere(Tr(a), S(b), vb, out) ? 1 : 0;
      if (ka == MGF_PLANE) return plane_msphere(Pl(a), S(b), vb, out) ? 1 : 0;
    } else if (kb == MGF_CAPSULE) {
      if (ka == MGF_SPHERE) return sphere_mcapsule(S(a), Cp(b), vb, out) ? 1 : 0;
      if (ka == MGF_CAPSULE) return capsule_mcapsule(Cp(a), Cp(b), vb, out) ? 1 : 0;
      if (ka == MGF_TRIANGLE) return tri_mcapsule(Tr(a), Cp(b), vb, out[0], out[1]);
      if (ka == MGF_PLANE) return plane_mcapsule(Pl(a), Cp(b), vb, out) ? 1 : 0;
    }
    return -1;
  }
  if (ma && !mb) {  // moving receiver, static argument
    if (kb == MGF_TRIANGLE || kb == MGF_PLANE) {  // commute_contacts! :607-608, :661-664
      int n = contacts_dispatch(b, false, mk3(0, 0, 0), a, true, va, out);
      for (int k = 0; k < n; ++k) out[k] = neg(out[k]);
      return n;
    }
    // collision.rs:1368-1382: rhs sweeps at -self.vel, result shifted by self.vel * t
    int n = contacts_dispatch(a, false, mk3(0, 0, 0), b, true, -va, out);
    for (int k = 0; k < n; ++k) { V3 d = va * out[k].t; out[k] = mkc(out[k].a + d, out[k].b + d, out[k].n, out[k].t); }
    return n;
  }
  if (ma && mb) {  // collision.rs:1387-1401
    int n = contacts_dispatch(a, false, mk3(0, 0, 0), b, true, vb - va, out);
    for (int k = 0; k < n; ++k) out[k] = mkc(out[k].a + va * out[k].t, out[k].b + va * out[k].t, out[k].n, out[k].t);
    return n;
  }
  return -1;
}

__global__ void k_contacts_batch(int64_t n, const ShapeIn* a, const float* va, const ShapeIn* b, const float* vb,
                                 const uint8_t* has_vel, ContactOut* out, int32_t* counts) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  Contact c[2];
  bool ma = has_vel[t] & 1, mb = has_vel[t] & 2;
  int m = contacts_dispatch(a[t], ma, ld3(va + 3 * t), b[t], mb, ld3(vb + 3 * t), c);
  counts[t] = m;
  for (int k = 0; k < m && k < 2; ++k) out[2 * t + k] = to_out(c[k]);
}

// Intersects<Shape> for a particle (Ray: dt = inf; Segment: p = a, d = b - a, dt = 1) collision.rs:169-373
struct ParticleIn { float p[3], d[3], dt; };
struct InterOut { float p[3], t; };
__device__ inline int intersection_dispatch(const ParticleIn& q, const ShapeIn& s, V3* ip, float* t) {
  V3 p = ld3(q.p), d = ld3(q.d);
  switch (s.kind) {
    case MGF_SPHERE: return ray_sphere(p, d, mks(mk3(s.v[0], s.v[1], s.v[2]), s.v[3]), ip, t, q.dt) ? 1 : 0;
    case MGF_CAPSULE: return ray_capsule(p, d, mkcap(mk3(s.v[0], s.v[1], s.v[2]), mk3(s.v[3], s.v[4], s.v[5]), s.v[6]), ip, t, q.dt) ? 1 : 0;
    case MGF_TRIANGLE: return ray_triangle(p, d, mkt(mk3(s.v[0], s.v[1], s.v[2]), mk3(s.v[3], s.v[4], s.v[5]), mk3(s.v[6], s.v[7], s.v[8])), ip, t, q.dt) ? 1 : 0;
    case MGF_PLANE: { Plane pl; pl.n = mk3(s.v[0], s.v[1], s.v[2]); pl.d = s.v[3]; return ray_plane(p, d, pl, ip, t, q.dt) ? 1 : 0; }
    default: return -1;
  }
}
__global__ void k_intersections_batch(int64_t n, const ParticleIn* parts, const ShapeIn* shapes, const float* boxes /* or */, InterOut* out,
                                      int32_t* hit) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  V3 ip = mk3(0, 0, 0); float t = 0.0f;
  int h;
  if (boxes) { Box b; b.c = ld3(boxes + 6 * i); b.r = ld3(boxes + 6 * i + 3); h = ray_box(ld3(parts[i].p), ld3(parts[i].d), b, &ip, &t, parts[i].dt) ? 1 : 0; }
  else h = intersection_dispatch(parts[i], shapes[i], &ip, &t);
  hit[i] = h;
  if (h == 1) { st3(out[i].p, ip); out[i].t = t; }
}
// BVH::raytrace bvh.rs:345-369 over a flattened reference-built tree, reference visiting order.
template <bool FILL>
__global__ __launch_bounds__(kBlock) void k_bvh_raytrace(TerrainDev M, const ParticleIn* parts, int64_t n, uint32_t* cnt, const uint32_t* off,
                                                         uint32_t* vals, InterOut* inters) {
  int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  V3 p = ld3(parts[i].p), d = ld3(parts[i].d);
  const float dt = parts[i].dt;
  uint32_t m = 0, base = FILL ? off[i] : 0;
  uint32_t stack[kStack];
  int sp = 0;
  stack[sp++] = M.root;
  while (sp > 0) {
    uint32_t top = stack[--sp];
    const float4* raw = reinterpret_cast<const float4*>(&M.nodes[top]);
    float4 n0 = raw[0], n1 = raw[1];
    Box nb; nb.c = xyz(n0); nb.r = xyz(n1);
    V3 ip; float t;
    if (ray_box(p, d, nb, &ip, &t, dt)) {
      uint32_t w0 = f2u(n0.w), w1 = f2u(n1.w);
      if (w0 & 0x80000000u) {
        if (FILL) { vals[base + m] = w0 & 0x7FFFFFFFu; st3(inters[base + m].p, ip); inters[base + m].t = t; }
        ++m;
      } else if (sp + 2 <= kStack) { stack[sp++] = w0; stack[sp++] = w1; }
      else if (M.err) *M.err = 1u;
    }
  }
  if (!FILL) cnt[i] = m;
}

struct LocalOut { float la[3], lb[3]; ContactOut g; };
// ContactPruner::push (manifold.rs:72-102) for each LocalContact of a group in order, then Manifold::from(pruner)
// (:131-148): earliest-time contacts only (+-1e-6), points closer than sqrt(0.5) to a kept one merge (the one farther
// from the centres stays), normal = un-renormalised mean (NaN for an empty group, as in the reference).
constexpr int kManifoldCap = 8;  // the reference's SmallVec spills beyond 4 and never stops; groups that keep more raise `overflow`
struct ManifoldOut { float time; float normal[3]; float t0[3]; float t1[3]; int32_t n; float la[kManifoldCap][3]; float lb[kManifoldCap][3]; };
__global__ __launch_bounds__(kBlock) void k_manifolds(int64_t n, const unsigned long long* off, const LocalOut* lcs, float threshold_sq, float eps,
                                                       ManifoldOut* out, uint32_t* overflow) {
  int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  float min_t = kInf;
  int cnt = 0;
  LocalOut keep[kManifoldCap];
  for (unsigned long long e = off[i]; e < off[i + 1]; ++e) {
    LocalOut nc = lcs[e];
    if (nc.g.t < min_t - eps) { cnt = 1; keep[0] = nc; min_t = nc.g.t; continue; }
    if (nc.g.t > min_t + eps) continue;
    bool merged = false;
    for (int k = 0; k < cnt && !merged; ++k) {
      V3 ra = ld3(nc.g.a) - ld3(keep[k].g.a), rb = ld3(nc.g.b) - ld3(keep[k].g.b);
      if (mag2(ra) <= threshold_sq || mag2(rb) <= threshold_sq) {
        float prev = mag2(ld3(keep[k].la)) + mag2(ld3(keep[k].lb)), cur = mag2(ld3(nc.la)) + mag2(ld3(nc.lb));
        if (prev < cur) keep[k] = nc;
        merged = true;
      }
    }
    if (merged) continue;
    if (cnt < kManifoldCap) keep[cnt] = nc; else *overflow = 1u;
    ++cnt;
  }
  ManifoldOut m;
  V3 sum = mk3(0.0f, 0.0f, 0.0f);
  int stored = cnt < kManifoldCap ? cnt : kManifoldCap;
  for (int k = 0; k < stored; ++k) {
    sum = sum + ld3(keep[k].g.n);
    for (int c = 0; c < 3; ++c) { m.la[k][c] = keep[k].la[c]; m.lb[k][c] = keep[k].lb[c]; }
  }
  for (int k = stored; k < kManifoldCap; ++k) for (int c = 0; c < 3; ++c) { m.la[k][c] = 0.0f; m.lb[k][c] = 0.0f; }
  V3 avg = sum / (float)cnt;
  V3 t0, t1;
  compute_basis(avg, &t0, &t1);
  m.time = min_t; st3(m.normal, avg); st3(m.t0, t0); st3(m.t1, t1); m.n = cnt;
  out[i] = m;
}

struct MovingIn { int tag; float p[3], d[3], r; float delta[3]; };
// The cheap conservative reject ahead of the body-triangle tests (comp_tri_far, dev_geom.h) beside the tests themselves, for batches of
// independent (moving component, triangle) problems: far[i] = the reject's verdict, counts[i] = the contacts the reference's tests report
// (collision.rs:610-1086 through Mesh::contacts' frame) - a problem with far[i] and counts[i] > 0 would be a contact the tick's r06 front
// end loses (tests/test_gpu_tri_reject.py fuzzes it).
__global__ void k_tri_reject_batch(int64_t n, const MovingIn* bodies, const float* tris, uint8_t* far, int32_t* counts) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const MovingIn m = bodies[t];
  Comp A; A.kind = m.tag; A.p = ld3(m.p); A.d = m.tag == KIND_SPHERE ? mk3(0.0f, 0.0f, 0.0f) : ld3(m.d); A.r = m.r;
  const V3 vA = ld3(m.delta);
  const Triangle tri = mkt(ld3(tris + 9 * t), ld3(tris + 9 * t + 3), ld3(tris + 9 * t + 6));
  LocalContact lc[2];
  far[t] = comp_tri_far(A, vA, tri) ? 1 : 0;
  counts[t] = comp_tri_local(A, vA, tri, mk3(0.0f, 0.0f, 0.0f), lc);
}
// ---- Compound (compound.rs:230-352): components + internal reference-built BVH + pose -------------------
struct CompIn { int tag; float p[3], d[3], r; };
struct CompoundDev {
  TerrainDev tree;       // flattened BVH<AABB, Component>; leaf value = component index
  const CompIn* comps;
  float disp[3];
  float rot[4];          // s, x, y, z
};
__device__ __forceinline__ Comp to_comp(const CompIn& m) { Comp k; k.kind = m.tag; k.p = ld3(m.p); k.d = ld3(m.d); k.r = m.r; return k; }
// Volumetric::rotate for AABB geom.rs:940-985
HD Box box_rotate(const Box& b, Quat rot) {
  V3 vx = rotate(rot, mk3(b.r.x, 0.0f, 0.0f)), vy = rotate(rot, mk3(0.0f, b.r.y, 0.0f)), vz = rotate(rot, mk3(0.0f, 0.0f, b.r.z));
  V3 p[8] = {b.c + (vx + vy + vz), b.c + (vx + vy - vz), b.c + (vx - vy + vz), b.c + (vx - vy - vz),
             b.c + (-vx + vy + vz), b.c + (-vx + vy - vz), b.c + (-vx - vy + vz), b.c + (-vx - vy - vz)};
  V3 lo = p[7], hi = p[7];
#pragma unroll
  for (int e = 6; e >= 0; --e) {  // p1.min(p2.min(... p8)): nested right to left
    lo = mk3(fmin_rs(p[e].x, lo.x), fmin_rs(p[e].y, lo.y), fmin_rs(p[e].z, lo.z));
    hi = mk3(fmax_rs(p[e].x, hi.x), fmax_rs(p[e].y, hi.y), fmax_rs(p[e].z, hi.z));
  }
  Box o; o.r = (hi - lo) / 2.0f; o.c = (hi + lo) / 2.0f;
  return o;
}
// Volumetric::rotate for Component (sphere: no-op; capsule: about its centre) geom.rs:999-1015
__device__ inline Comp comp_rotate(Comp k, Quat r) {
  if (k.kind == KIND_CAPSULE) { V3 ctr = comp_center(k); k.p = ctr + rotate(r, k.p - ctr); k.d = rotate(r, k.d); }
  return k;
}
// Volumetric::rotate_about geom.rs:932-937 (set_pos moves the centre)
__device__ inline Comp comp_rotate_about(Comp k, Quat r, V3 p) {
  V3 ctr = comp_center(k);
  V3 disp = (p + rotate(r, ctr - p)) - ctr;
  k.p = k.p + disp;
  return comp_rotate(k, r);
}
__device__ inline ShapeIn comp_shape(const Comp& k) {
  ShapeIn s; s.kind = k.kind == KIND_SPHERE ? MGF_SPHERE : MGF_CAPSULE;
  for (int e = 0; e < 12; ++e) s.v[e] = 0.0f;
  if (k.kind == KIND_SPHERE) { st3(s.v, k.p); s.v[3] = k.r; }
  else { st3(s.v, k.p); st3(s.v + 3, k.d); s.v[6] = k.r; }
  return s;
}
// Contacts<RHS> for Compound compound.rs:334-352, RHS = Moving<Sphere | Capsule>: one thread per rhs, contacts in BVH
// query order (count pass / fill pass).
template <bool FILL>
__global__ __launch_bounds__(kBlock) void k_compound_contacts(CompoundDev D, const MovingIn* rhs, int64_t n, uint32_t* cnt, const uint32_t* off,
                                                              ContactOut* out) {
  int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  Comp R; R.kind = rhs[i].tag; R.p = ld3(rhs[i].p); R.d = ld3(rhs[i].d); R.r = rhs[i].r;
  V3 vel = ld3(rhs[i].delta), disp = ld3(D.disp);
  Quat rot = mkq(D.rot[0], mk3(D.rot[1], D.rot[2], D.rot[3]));
  Quat conj = mkq(rot.s, -rot.v);
  Box rb = box_rotate(swept_bounds(R, vel), conj);
  rb.c = rotate(conj, rb.c + -disp) + disp;
  ShapeIn rs = comp_shape(R);
  uint32_t m = 0, base = FILL ? off[i] : 0;
  terrain_traverse(D.tree, rb, [&](uint32_t ci) {
    Comp shape = comp_rotate_about(to_comp(D.comps[ci]), rot, mk3(0.0f, 0.0f, 0.0f));
    shape.p = shape.p + disp;
    Contact c[2];
    int k = contacts_dispatch(rs, true, vel, comp_shape(shape), false, mk3(0, 0, 0), c);  // Moving<Recv>.contacts(&Arg) :1368-1382
    for (int e = 0; e < k; ++e) {
      if (FILL) out[base + m] = to_out(neg(c[e]));
      ++m;
    }
  });
  if (!FILL) cnt[i] = m;
}
// Intersects<Compound> for a particle compound.rs:309-332
__global__ __launch_bounds__(kBlock) void k_compound_intersections(CompoundDev D, const ParticleIn* parts, int64_t n, InterOut* out, int32_t* hit) {
  int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  V3 pp = ld3(parts[i].p), pd = ld3(parts[i].d), disp = ld3(D.disp);
  const float dt = parts[i].dt;
  Quat rot = mkq(D.rot[0], mk3(D.rot[1], D.rot[2], D.rot[3]));
  Quat conj = mkq(rot.s, -rot.v);
  V3 rp = rotate(conj, pp + -disp) + disp, rd = rotate(conj, pd);
  bool have = false;
  V3 best_p = mk3(0, 0, 0); float best_t = 0.0f;
  uint32_t stack[kStack];
  int sp = 0;
  if (D.tree.n_nodes) stack[sp++] = D.tree.root;
  while (sp > 0) {
    uint32_t top = stack[--sp];
    const float4* raw = reinterpret_cast<const float4*>(&D.tree.nodes[top]);
    float4 n0 = raw[0], n1 = raw[1];
    Box nb; nb.c = xyz(n0); nb.r = xyz(n1);
    V3 ip; float t;
    if (ray_box(rp, rd, nb, &ip, &t, kInf)) {  // the BVH is traced with a Ray (DT = inf), :315-316
      uint32_t w0 = f2u(n0.w), w1 = f2u(n1.w);
      if (w0 & 0x80000000u) {
        if (!(t > dt)) {
          Comp shape = comp_rotate(to_comp(D.comps[w0 & 0x7FFFFFFFu]), rot);
          shape.p = shape.p + disp;
          ParticleIn q = parts[i];
          V3 sip; float st;
          if (intersection_dispatch(q, comp_shape(shape), &sip, &st) == 1 && !(have && st > best_t)) { best_p = sip; best_t = st; have = true; }
        }
      } else if (sp + 2 <= kStack) { stack[sp++] = w0; stack[sp++] = w1; }
      else if (D.tree.err) *D.tree.err = 1u;
    }
  }
  hit[i] = have ? 1 : 0;
  if (have) { st3(out[i].p, best_p); out[i].t = best_t; }
}

// ---- static Compounds as obstacles of the world (round 3; the oracle's World::obstacles) -------------------------------------
// A candidate of the "terrain" lists is a mesh face or - flagged - a component of an obstacle met by one part of the body:
// kObstacleFlag | obstacle << 23 | part << 18 | component (r06: five bits of part - bodies of up to 32 components - and eighteen of component).  Appended to a body's terrain row behind its faces: obstacles in insertion
// order, the body's parts in order, the components in the order Compound::contacts visits them (compound.rs:334-352: its BVH
// queried with the part's bounds turned into the obstacle's frame).
constexpr uint32_t kObstacleFlag = 0x80000000u, kObstacleMax = 256u, kObstacleCompMax = 1u << 18, kObstaclePartShift = 18u;
static_assert(kBigParts <= 32, "five bits of part in an obstacle candidate");
__global__ __launch_bounds__(kBlock) void k_obstacle_rows(Bodies B, uint32_t n_owned, const CompoundDev* obs, uint32_t n_obs, uint32_t cap_row, uint32_t* rows_t,
                                                          uint32_t* t_cnt, uint32_t* overflow) {
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n_owned) return;
  const uint32_t pc = B.pcount ? B.pcount[i] : 0u;
  const int np = pc ? (int)pc : 1;
  const V3 vel = xyz(B.delta[i]);
  uint32_t nt = t_cnt[i];
  uint32_t* row = rows_t + (size_t)i * cap_row;
  for (uint32_t k = 0; k < n_obs; ++k) {
    const CompoundDev D = obs[k];
    const V3 disp = ld3(D.disp);
    const Quat rot = mkq(D.rot[0], mk3(D.rot[1], D.rot[2], D.rot[3])), conj = mkq(rot.s, -rot.v);
    for (int pa = 0; pa < np; ++pa) {
      Comp part; V3 ci;
      (void)load_part(B, i, (uint32_t)pa, &part, &ci);
      Box rb = box_rotate(swept_bounds(part, vel), conj);  // compound.rs:340-344
      rb.c = rotate(conj, rb.c + -disp) + disp;
      terrain_traverse(D.tree, rb, [&](uint32_t comp) {
        if (nt < cap_row) row[nt] = kObstacleFlag | (k << 23) | ((uint32_t)pa << kObstaclePartShift) | comp;
        ++nt;
      });
    }
  }
  t_cnt[i] = nt;
  if (nt > cap_row) atomicOr(overflow, 2u);
}
// The same candidates on the exact count / fill path of the candidate search (k_candidates: the tick that follows a row overflow).
// FILL = false: adds a body's obstacle hits to its count, behind k_candidates<false>; FILL = true: writes them behind the body's
// faces, in the order k_obstacle_rows lists them (t_cnt[i] = faces + hits from the counting pass), behind k_candidates<true>.
template <bool FILL>
__global__ __launch_bounds__(kBlock) void k_obstacle_candidates(Bodies B, uint32_t n_owned, const CompoundDev* obs, uint32_t n_obs, uint32_t* t_cnt,
                                                                const uint32_t* t_off, uint32_t* t_cand, uint32_t* t_owner, const StepCounts* sc) {
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n_owned) return;
  if (FILL && sc->fail) return;
  const uint32_t pc = B.pcount ? B.pcount[i] : 0u;
  const int np = pc ? (int)pc : 1;
  const V3 vel = xyz(B.delta[i]);
  auto walk = [&](auto&& hit) {
    for (uint32_t k = 0; k < n_obs; ++k) {
      const CompoundDev D = obs[k];
      const V3 disp = ld3(D.disp);
      const Quat rot = mkq(D.rot[0], mk3(D.rot[1], D.rot[2], D.rot[3])), conj = mkq(rot.s, -rot.v);
      for (int pa = 0; pa < np; ++pa) {
        Comp part; V3 ci;
        (void)load_part(B, i, (uint32_t)pa, &part, &ci);
        Box rb = box_rotate(swept_bounds(part, vel), conj);  // compound.rs:340-344
        rb.c = rotate(conj, rb.c + -disp) + disp;
        terrain_traverse(D.tree, rb, [&](uint32_t comp) { hit(kObstacleFlag | (k << 23) | ((uint32_t)pa << kObstaclePartShift) | comp); });
      }
    }
  };
  uint32_t no = 0;
  walk([&](uint32_t) { ++no; });
  if (!FILL) { t_cnt[i] += no; return; }
  uint32_t at = t_off[i] + (t_cnt[i] - no);
  walk([&](uint32_t f) { t_cand[at] = f; t_owner[at] = i; ++at; });
}
// The flagged candidates' contacts: Moving<part>.contacts(&component) through the :1368-1382 wrapper (what Compound::contacts calls, and
// negates), then LocalContacts (collision.rs:1490-1506) with the obstacle in the Mesh's place: the record Manifold::from(lc) and
// ContactConstraint::new consume.  k_narrow_terrain* leave these candidates alone.
__global__ __launch_bounds__(kBlock) void k_narrow_obstacles(Bodies B, const CompoundDev* obs, const uint32_t* m_ptr, const uint32_t* t_owner, const uint32_t* t_cand,
                                                             uint32_t* t_nc, NContact* t_out, uint32_t stride) {
  const uint32_t p = blockIdx.x * kBlock + threadIdx.x;
  if (p >= *m_ptr) return;
  const uint32_t f = t_cand[p];
  if (!(f & kObstacleFlag)) return;
  const uint32_t i = t_owner[p], k = (f >> 23) & 0xFFu, pa = (f >> kObstaclePartShift) & 31u, ci_ = f & (kObstacleCompMax - 1u);
  Comp part;
  V3 centre;
  (void)load_part(B, i, pa, &part, &centre);
  const V3 vel = xyz(B.delta[i]);
  const CompoundDev D = obs[k];
  const Quat rot = mkq(D.rot[0], mk3(D.rot[1], D.rot[2], D.rot[3]));
  Comp shape = comp_rotate_about(to_comp(D.comps[ci_]), rot, mk3(0.0f, 0.0f, 0.0f));
  shape.p = shape.p + ld3(D.disp);
  Contact c[2];
  const int m = contacts_dispatch(comp_shape(part), true, vel, comp_shape(shape), false, mk3(0, 0, 0), c);
  const V3 oc = ld3(D.disp);  // Shape::center for Compound compound.rs:289-291
  uint32_t cnt = 0;
  for (int e = 0; e < m && e < 2; ++e) {
    // (Compound::contacts hands out -c: a on the obstacle; LocalContacts then takes b - the body side - relative to the body's
    // centre at the contact time, a relative to the obstacle's, and negates back)
    NContact o;
    o.la = mk4(c[e].a + -(centre + vel * c[e].t), c[e].t);
    o.lb = mk4(c[e].b + -oc, 0.0f);
    o.n = mk4(c[e].n, 0.0f);
    t_out[(size_t)stride * p + cnt++] = o;
  }
  t_nc[p] = cnt;
}

__device__ __forceinline__ Comp to_comp(const MovingIn& m) {
  Comp k; k.kind = m.tag; k.p = ld3(m.p); k.d = ld3(m.d); k.r = m.r; return k;
}
__global__ void k_local_pair(MovingIn a, MovingIn b, LocalOut* out, int32_t* count) {
  LocalContact lc;
  bool hit = comp_pair_local(to_comp(a), ld3(a.delta), to_comp(b), ld3(b.delta), &lc);
  *count = hit ? 1 : 0;
  if (hit) { st3(out->la, lc.la); st3(out->lb, lc.lb); out->g = to_out(lc.g); }
}
// Moving<Component>.local_contacts(&Mesh): mesh-BVH DFS order, up to 2 contacts per face.
__global__ void k_local_mesh(MovingIn a, TerrainDev M, LocalOut* out, int32_t cap, int32_t* count) {
  Comp A = to_comp(a);
  V3 vA = ld3(a.delta);
  V3 mx = mk3(M.x[0], M.x[1], M.x[2]);
  Box q = swept_bounds(A, vA);
  q.c = q.c + -mx;
  int n = 0;
  terrain_traverse(M, q, [&](uint32_t f) {
    uint4 fi = M.faces[f];
    Triangle tri = mkt(xyz(M.verts[fi.x]) + mx, xyz(M.verts[fi.y]) + mx, xyz(M.verts[fi.z]) + mx);
    LocalContact lc[2];
    int m = comp_tri_local(A, vA, tri, mx, lc);
    for (int k = 0; k < m; ++k) {
      if (n < cap) { st3(out[n].la, lc[k].la); st3(out[n].lb, lc[k].lb); out[n].g = to_out(lc[k].g); }
      ++n;
    }
  });
  *count = n;
}
__global__ void k_ray_capsule(V3 p, V3 d, Capsule cap, float* out4, int32_t* hit) {
  V3 ip; float t;
  bool h = ray_capsule(p, d, cap, &ip, &t);
  *hit = h ? 1 : 0;
  if (h) { out4[0] = ip.x; out4[1] = ip.y; out4[2] = ip.z; out4[3] = t; }
}
// BVH::query for many AABBs against a flattened host tree (reference DFS order).
template <bool FILL>
__global__ __launch_bounds__(kBlock) void k_bvh_query(TerrainDev M, const float* boxes /* 6 per query */, int64_t n, uint32_t* cnt,
                                                      const uint32_t* off, uint32_t* vals) {
  int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (t >= n) return;
  Box q; q.c = ld3(boxes + 6 * t); q.r = ld3(boxes + 6 * t + 3);
  uint32_t m = 0, base = FILL ? off[t] : 0;
  terrain_traverse(M, q, [&](uint32_t v) { if (FILL) vals[base + m] = v; ++m; });
  if (!FILL) cnt[t] = m;
}

// mgf_world_read_state's device half: the caller's body e sits in slot slot_of[e] (null: e itself); what was asked for of its x, q, v, omega,
// delta packed in the caller's order - 3 / 4 / 3 / 3 / 3 floats a body - so that ONE copy into pinned memory brings exactly the bytes the
// caller gets (r05: four pageable copies of whole float4 arrays and a gather on the host took 16-20 ms for 262 144 bodies)
__global__ __launch_bounds__(kBlock) void k_pack_state(Bodies B, uint32_t n, const uint32_t* slot_of, float* x, float* q, float* v, float* om, float* d) {
  const uint32_t e = blockIdx.x * kBlock + threadIdx.x;
  if (e >= n) return;
  const size_t i = slot_of ? slot_of[e] : e;
  if (x) { const float4 a = B.x[i]; x[3 * (size_t)e] = a.x; x[3 * (size_t)e + 1] = a.y; x[3 * (size_t)e + 2] = a.z; }
  if (q) reinterpret_cast<float4*>(q)[e] = B.q[i];
  if (d) { const float4 a = B.delta[i]; d[3 * (size_t)e] = a.x; d[3 * (size_t)e + 1] = a.y; d[3 * (size_t)e + 2] = a.z; }
  if (v || om) {
    const float4 s0 = B.srec[4 * i], s1 = B.srec[4 * i + 1];
    if (v) { v[3 * (size_t)e] = s0.x; v[3 * (size_t)e + 1] = s0.y; v[3 * (size_t)e + 2] = s0.z; }
    if (om) { om[3 * (size_t)e] = s0.w; om[3 * (size_t)e + 1] = s1.x; om[3 * (size_t)e + 2] = s1.y; }
  }
}

// ... and mgf_world_write_state's: the caller's packed arrays (null: not given) into the slots, the other words of a record - inverse mass,
// inertia, delta's friction word - left as they are
__global__ __launch_bounds__(kBlock) void k_unpack_state(Bodies B, uint32_t n, const uint32_t* slot_of, const float* x, const float* q, const float* v, const float* om,
                                                         const float* d) {
  const uint32_t e = blockIdx.x * kBlock + threadIdx.x;
  if (e >= n) return;
  const size_t i = slot_of ? slot_of[e] : e;
  if (x) B.x[i] = make_float4(x[3 * (size_t)e], x[3 * (size_t)e + 1], x[3 * (size_t)e + 2], 0.0f);
  if (q) B.q[i] = reinterpret_cast<const float4*>(q)[e];
  if (d) { const float fw = B.delta[i].w; B.delta[i] = make_float4(d[3 * (size_t)e], d[3 * (size_t)e + 1], d[3 * (size_t)e + 2], fw); }
  if (v || om) {
    float4 s0 = B.srec[4 * i];
    if (v) { s0.x = v[3 * (size_t)e]; s0.y = v[3 * (size_t)e + 1]; s0.z = v[3 * (size_t)e + 2]; }
    if (om) {
      s0.w = om[3 * (size_t)e];
      float2* s1 = reinterpret_cast<float2*>(&B.srec[4 * i + 1]);
      *s1 = make_float2(om[3 * (size_t)e + 1], om[3 * (size_t)e + 2]);
    }
    B.srec[4 * i] = s0;
  }
}

// mgf_world_read_colliders' device half: Moving<Component> of the caller's body e as the 11 words of mgf_moving_component (tag, p, d, r; delta)
__global__ __launch_bounds__(kBlock) void k_pack_colliders(Bodies B, uint32_t n, const uint32_t* slot_of, uint32_t* out) {
  const uint32_t e = blockIdx.x * kBlock + threadIdx.x;
  if (e >= n) return;
  const size_t i = slot_of ? slot_of[e] : e;
  const float4 a = B.col0[i], b = B.col1[i], d = B.delta[i];
  uint32_t* o = out + 11 * (size_t)e;
  o[0] = f2u(b.w);
  o[1] = f2u(a.x); o[2] = f2u(a.y); o[3] = f2u(a.z);
  o[4] = f2u(b.x); o[5] = f2u(b.y); o[6] = f2u(b.z);
  o[7] = f2u(a.w);
  o[8] = f2u(d.x); o[9] = f2u(d.y); o[10] = f2u(d.z);
}

}  // namespace mgf
