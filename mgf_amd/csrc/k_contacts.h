// Narrowphase per shape-pair type, contact numbering, ContactConstraint records and their setup.  (Part of the kernel set described in kernels.h.)
#pragma once
#include "k_broadphase.h"

namespace mgf {

// ------------------------------------------------------------------------------------------
// Narrowphase, one kernel per shape-pair type.  Output per candidate: contact count and the
// LocalContact reduced to what Manifold/ContactConstraint::new consume (local_a, local_b, n).
// ------------------------------------------------------------------------------------------
struct NContact { float4 la, lb, n; };  // la.xyz + t, lb.xyz, n.xyz
// an entry of a body's row of the constraints it takes part in as `b` (k_chain_rows / k_flow6_links, k_links.h): the constraint's id,
// body a's slot and body a's order id
struct RevEnt { uint32_t c, a, oid, pad; };
static_assert(sizeof(RevEnt) == 16, "a row entry is one 16-byte word");


// work = nullptr: dense over [0, m); else the m candidate ids of this pair type.
template <int KA, int KB>
__global__ __launch_bounds__(kBlock) void k_narrow_pairs(Bodies B, const uint32_t* work, const uint32_t* m_ptr, const uint32_t* p_owner,
                                                         const uint32_t* p_cand, uint32_t* p_nc, NContact* p_out) {
  // Candidates whose bounding spheres never come within reach (comp_pair_far: most of a pile's) leave first, and the block's
  // survivors are packed into its first lanes: a lane that leaves early saves nothing while its wave runs the pair test.
  __shared__ uint32_t s_list[kBlock];
  __shared__ uint32_t s_n;
  if (threadIdx.x == 0) s_n = 0;
  __syncthreads();
  {
    const uint32_t t = blockIdx.x * kBlock + threadIdx.x;
    bool live = false;
    uint32_t p0 = 0;
    if (t < *m_ptr) {
      p0 = work ? work[t] : t;
      V3 v0, v1;
      Comp A0 = load_comp_moving(B, p_owner[p0], &v0), B0 = load_comp_moving(B, p_cand[p0], &v1);
      A0.kind = KA; B0.kind = KB;
      live = !comp_pair_far(A0, v0, B0, v1);
      if (!live) p_nc[p0] = 0u;
    }
    const unsigned long long mask = __ballot(live);
    const int lane = threadIdx.x & 63;
    uint32_t base = 0;
    if (lane == 0 && mask) base = atomicAdd(&s_n, (uint32_t)__popcll(mask));
    base = __shfl(base, 0);
    if (live) s_list[base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = p0;
  }
  __syncthreads();
  if (threadIdx.x >= s_n) return;
  const uint32_t p = s_list[threadIdx.x];
  uint32_t i = p_owner[p], j = p_cand[p];
  V3 vA, vB;
  Comp A = load_comp_moving(B, i, &vA), Bc = load_comp_moving(B, j, &vB);
  A.kind = KA; Bc.kind = KB;  // compile-time dispatch: the list holds only this pair type
  LocalContact lc;
  bool hit = comp_pair_local(A, vA, Bc, vB, &lc);
  p_nc[p] = hit ? 1u : 0u;
  if (hit) {
    // ContactPruner::push on an empty pruner keeps the contact (manifold.rs:73-79);
    // Manifold::from(pruner): normal = (0 + n) / 1 (manifold.rs:135-140)
    V3 nrm = (mk3(0.0f, 0.0f, 0.0f) + lc.g.n) / 1.0f;
    NContact o; o.la = mk4(lc.la, lc.g.t); o.lb = mk4(lc.lb, 0.0f); o.n = mk4(nrm, 0.0f);
    p_out[p] = o;
  }
}

template <int KA>
__global__ __launch_bounds__(kBlock) void k_narrow_terrain(Bodies B, TerrainDev M, const uint32_t* work, const uint32_t* m_ptr,
                                                           const uint32_t* t_owner, const uint32_t* t_cand, uint32_t* t_nc,
                                                           NContact* t_out /* 2 per candidate */) {
  uint32_t t = blockIdx.x * kBlock + threadIdx.x;
  if (t >= *m_ptr) return;
  uint32_t p = work ? work[t] : t;
  uint32_t i = t_owner[p], f = t_cand[p];
  if (f & 0x80000000u) return;  // a component of a static obstacle, not a face: k_narrow_obstacles (k_api.h)
  V3 vA;
  Comp A = load_comp_moving(B, i, &vA);
  A.kind = KA;
  uint4 fi = M.faces[f];
  V3 mx = mk3(M.x[0], M.x[1], M.x[2]);
  Triangle tri = mkt(xyz(M.verts[fi.x]) + mx, xyz(M.verts[fi.y]) + mx, xyz(M.verts[fi.z]) + mx);  // mesh.rs:122-126
  LocalContact lc[2];
  int nc = comp_tri_local(A, vA, tri, mx, lc);
  t_nc[p] = (uint32_t)nc;
#pragma unroll
  for (int k = 0; k < 2; ++k) {  // (constant indices: the two contacts stay in registers)
    if (k < nc) {
      NContact o; o.la = mk4(lc[k].la, lc[k].g.t); o.lb = mk4(lc[k].lb, 0.0f); o.n = mk4(lc[k].g.n, 0.0f);  // Manifold::from(lc) manifold.rs:120-128
      t_out[2 * p + k] = o;
    }
  }
}

// ---- worlds with bodies of several parts (BASELINE config 5; not in the reference) -------------------------------
// The same steps over every pair of parts: Contacts (compound.rs:180-190) per part pair in order (parts of i outer),
// local points relative to the BODIES' centres (LocalContacts, compound.rs:192-207), ContactPruner::push
// (manifold.rs:72-102), Manifold::from(pruner) (:131-148).  An ordinary body is a body of one part (its collider), so
// these kernels serve mixed worlds too.  At most kMaxParts^2 = 4 contacts per pair (every part pair emits <= 1).
constexpr float kPersistentThresholdSq = 0.5f;  // manifold.rs:38
// MP = the most parts any body of the world has (2 or 4: the kernels are instantiated for both, so that a world of two-part bodies
// does not carry the registers of sixteen part pairs): a pair of bodies yields at most MP * MP contacts, a (body, face) 2 * MP.
// One part of a body: part k of a body of several components, or the body's own collider (k = 0 of an ordinary body); false when
// the body has no such part.  *centre = the centre the local contact points are taken from (load_parts).
__device__ __forceinline__ bool load_part(const Bodies& B, uint32_t i, uint32_t k, Comp* out, V3* centre) {
  const uint32_t pc = B.pcount ? B.pcount[i] : 0u;
  if (pc == 0) { if (k) return false; *out = load_comp(B, i); *centre = comp_center(*out); return true; }
  if (k >= pc) return false;
  float4 a, b;
  world_part(B, i, k, pc, a, b);
  out->kind = (int)f2u(b.w); out->p = xyz(a); out->r = a.w; out->d = xyz(b);
  *centre = xyz(B.col0[i]);
  return true;
}
// The block's 256 candidates in three steps (round 3; one thread used to walk its candidate's MP * MP part pairs by itself, the parts
// in arrays it indexed at run time - scratch memory - and a block of mostly culled candidates kept one wave busy for all of them):
//   1. cull + pack, as k_narrow_pairs: candidates whose tight boxes do not meet leave;
//   2. one ITEM per (part pair, surviving candidate), part-pair-major - consecutive lanes run the same (a, b) of different
//      candidates, which is the same pair of shapes in a world of equal bodies - through Contacts (compound.rs:180-190); the raw
//      contact goes to the candidate's own output slot (a, b);
//   3. a thread per surviving candidate: ContactPruner::push over its raw contacts in order (parts of i outer) and Manifold::from.
template <int MP>
__global__ __launch_bounds__(kBlock) void k_narrow_pairs_parts(Bodies B, const uint32_t* m_ptr, const uint32_t* p_owner, const uint32_t* p_cand,
                                                               uint32_t* p_nc, NContact* p_out /* MP * MP per candidate */) {
  constexpr int kPairContacts = MP * MP;
  // The candidate was accepted on i's tight box against j's FAT box (bvh.rs:297), which is the larger by the margin and by every
  // tick since j's last refit.  A contact is a touching of two parts somewhere inside both bodies' TIGHT swept boxes of this
  // tick, so if those do not overlap (a millimetre and 1e-5 of the coordinates allowed for rounding) no part pair reports
  // anything.  Three candidates in four leave here, before their parts are read.
  __shared__ uint32_t s_list[kBlock];
  __shared__ uint32_t s_n;
  if (threadIdx.x == 0) s_n = 0;
  __syncthreads();
  {
    const uint32_t p0 = blockIdx.x * kBlock + threadIdx.x;
    bool live = false;
    if (p0 < *m_ptr) {
      const uint32_t i0 = p_owner[p0], j0 = p_cand[p0];
      const float4 ca = B.tb_c[i0], ra = B.tb_r[i0], cb = B.tb_c[j0], rb = B.tb_r[j0];
      const float slack = 1e-3f + 1e-5f * (fabs_rs(ca.x) + fabs_rs(ca.y) + fabs_rs(ca.z) + fabs_rs(cb.x) + fabs_rs(cb.y) + fabs_rs(cb.z));
      live = !(fabs_rs(ca.x - cb.x) > ra.x + rb.x + slack || fabs_rs(ca.y - cb.y) > ra.y + rb.y + slack || fabs_rs(ca.z - cb.z) > ra.z + rb.z + slack);
      if (!live) p_nc[p0] = 0u;
    }
    const unsigned long long mask = __ballot(live);
    const int lane = threadIdx.x & 63;
    uint32_t base = 0;
    if (lane == 0 && mask) base = atomicAdd(&s_n, (uint32_t)__popcll(mask));
    base = __shfl(base, 0);
    if (live) s_list[base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = p0;
  }
  __syncthreads();
  const uint32_t live_n = s_n;
  if (live_n == 0) return;
  // 2. the part pairs (raw slot: a.xyz, t | b.xyz, hit | n.xyz).  First the cheap conservative test of k_narrow_pairs on every item
  // (the parts' bounding spheres never come within reach during the tick: of a pair of bodies that touch, one part pair in MP * MP
  // does), survivors packed again - whole waves have to leave for the leaving to save anything - then Contacts on those.
  __shared__ uint16_t s_item[kBlock * kPairContacts];
  __shared__ uint32_t s_items;
  if (threadIdx.x == 0) s_items = 0;
  __syncthreads();
  // (the two parts as VALUES: with `load_part(.., &Pa, ..)` and `load_part(.., &Pb, ..)` the compiler ran both through one piece of code that
  // stored through a selected pointer - the parts' 16 bytes of (d, r) in scratch memory)
  auto part_of = [&](uint32_t i, uint32_t k, bool& ok) -> Comp {
    Comp c; c.kind = KIND_SPHERE; c.p = mk3(0, 0, 0); c.r = 0.0f; c.d = mk3(0, 0, 0);
    const uint32_t pc = B.pcount ? B.pcount[i] : 0u;
    ok = pc == 0u ? k == 0u : k < pc;
    if (ok) {
      float4 a, b;
      if (pc == 0u) { a = B.col0[i]; b = B.col1[i]; }
      else { a = B.wp0[kMaxParts * i + k]; b = B.wp1[kMaxParts * i + k]; }
      c.kind = (int)f2u(b.w); c.p = xyz(a); c.r = a.w; c.d = xyz(b);
    }
    return c;
  };
  auto item_parts = [&](uint32_t w, uint32_t& p, Comp& Pa, Comp& Pb, V3& vA, V3& vB) -> bool {
    const uint32_t ab = w / live_n;
    p = s_list[w - ab * live_n];
    const uint32_t a = ab / (uint32_t)MP, b = ab - a * (uint32_t)MP;
    const uint32_t i = p_owner[p], j = p_cand[p];
    bool oka, okb;
    Pa = part_of(i, a, oka); Pb = part_of(j, b, okb);
    if (!oka || !okb) return false;
    vA = xyz(B.delta[i]); vB = xyz(B.delta[j]);
    return true;
  };
  const uint32_t n_items = live_n * (uint32_t)kPairContacts;
  for (uint32_t w0 = 0; w0 < n_items; w0 += kBlock) {  // (whole waves walk the loop together: the ballots below see every lane)
    const uint32_t w = w0 + threadIdx.x;
    bool near = false;
    if (w < n_items) {
      uint32_t p; Comp Pa, Pb; V3 vA, vB;
      const bool have = item_parts(w, p, Pa, Pb, vA, vB);
      near = have && !comp_pair_far(Pa, vA, Pb, vB);
      if (!near) { NContact z; z.la = z.lb = z.n = make_float4(0, 0, 0, 0); p_out[(size_t)kPairContacts * p + w / live_n] = z; }
    }
    const unsigned long long mask = __ballot(near);
    const int lane = threadIdx.x & 63;
    uint32_t base = 0;
    if (lane == 0 && mask) base = atomicAdd(&s_items, (uint32_t)__popcll(mask));
    base = __shfl(base, 0);
    if (near) s_item[base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = (uint16_t)w;
  }
  __syncthreads();
  const uint32_t n_near = s_items;
  for (uint32_t k = threadIdx.x; k < n_near; k += kBlock) {
    const uint32_t w = s_item[k];
    uint32_t p; Comp Pa, Pb; V3 vA, vB;
    (void)item_parts(w, p, Pa, Pb, vA, vB);
    Contact c;
    NContact raw;
    raw.la = raw.lb = raw.n = make_float4(0, 0, 0, 0);
    if (comp_pair_contact(Pa, vA, Pb, vB, &c)) { raw.la = mk4(c.a, c.t); raw.lb = mk4(c.b, 1.0f); raw.n = mk4(c.n, 0.0f); }
    p_out[(size_t)kPairContacts * p + w / live_n] = raw;
  }
  __syncthreads();  // (a block's stores are visible to the block behind its barrier)
  // 3. the pruner, a thread per surviving candidate
  if (threadIdx.x >= live_n) return;
  const uint32_t p = s_list[threadIdx.x];
  const uint32_t i = p_owner[p], j = p_cand[p];
  const uint32_t pci = B.pcount ? B.pcount[i] : 0u, pcj = B.pcount ? B.pcount[j] : 0u;
  const int na = pci ? (int)min(pci, (uint32_t)MP) : 1, nb = pcj ? (int)min(pcj, (uint32_t)MP) : 1;
  const V3 ci = pci ? xyz(B.col0[i]) : comp_center(load_comp(B, i)), cj = pcj ? xyz(B.col0[j]) : comp_center(load_comp(B, j));
  const V3 vA = xyz(B.delta[i]), vB = xyz(B.delta[j]);
  // ContactPruner::push manifold.rs:72-102 over the raw contacts in the candidate's own output slots.  What the pruner keeps is a list
  // of SLOT NUMBERS (four bits each in one register pair) - a kept contact is read again from its slot when a later one is compared
  // with it - instead of up to MP * MP LocalContacts of 64 bytes in scratch memory (r04: 288 -> 0 bytes per lane for MP = 2).
  static_assert(kPairContacts <= 16, "four bits per kept slot");
  const NContact* mine = p_out + (size_t)kPairContacts * p;
  auto local_of = [&](const NContact& raw) -> LocalContact {
    Contact c; c.a = xyz(raw.la); c.b = xyz(raw.lb); c.n = xyz(raw.n); c.t = raw.la.w;
    LocalContact nc; nc.la = c.a + -(ci + vA * c.t); nc.lb = c.b + -(cj + vB * c.t); nc.g = c;
    return nc;
  };
  float min_t = kInf;
  int cnt = 0;
  unsigned long long keep = 0ull;  // keep[k] = (keep >> 4k) & 15: the slot of the k-th kept contact
  auto kept = [&](int k) -> uint32_t { return (uint32_t)(keep >> (4 * k)) & 15u; };
  auto set_kept = [&](int k, uint32_t slot) { keep = (keep & ~(15ull << (4 * k))) | ((unsigned long long)slot << (4 * k)); };
  for (int a = 0; a < na; ++a) {
    for (int b = 0; b < nb; ++b) {
      const uint32_t slot = (uint32_t)(a * MP + b);
      const NContact raw = mine[slot];
      if (raw.lb.w == 0.0f) continue;
      const LocalContact nc = local_of(raw);
      if (nc.g.t < min_t - kCollisionEps) { cnt = 1; set_kept(0, slot); min_t = nc.g.t; continue; }
      if (nc.g.t > min_t + kCollisionEps) continue;
      bool merged = false;
      for (int k = 0; k < cnt && !merged; ++k) {
        const LocalContact kc = local_of(mine[kept(k)]);
        V3 ra = nc.g.a - kc.g.a, rb = nc.g.b - kc.g.b;
        if (mag2(ra) <= kPersistentThresholdSq || mag2(rb) <= kPersistentThresholdSq) {
          float prev = mag2(kc.la) + mag2(kc.lb), cur = mag2(nc.la) + mag2(nc.lb);
          if (prev < cur) set_kept(k, slot);
          merged = true;
        }
      }
      if (!merged) set_kept(cnt++, slot);  // cnt <= na * nb <= kPairContacts
    }
  }
  p_nc[p] = (uint32_t)cnt;
  if (cnt == 0) return;
  V3 sum = mk3(0.0f, 0.0f, 0.0f);
  for (int k = 0; k < cnt; ++k) sum = sum + xyz(mine[kept(k)].n);
  const V3 avg = sum / (float)cnt;
  // (the k-th kept contact sits in a slot >= k - it was found at or after the k-th raw contact - and the kept slots are distinct: writing
  // output slot k never overwrites a raw contact that a later k still has to read)
  for (int k = 0; k < cnt; ++k) {
    const LocalContact kc = local_of(mine[kept(k)]);
    NContact o; o.la = mk4(kc.la, min_t); o.lb = mk4(kc.lb, 0.0f); o.n = mk4(avg, 0.0f);
    p_out[(size_t)kPairContacts * p + k] = o;
  }
}
template <int MP>
__global__ __launch_bounds__(kBlock) void k_narrow_terrain_parts(Bodies B, TerrainDev M, const uint32_t* m_ptr, const uint32_t* t_owner,
                                                                 const uint32_t* t_cand, uint32_t* t_nc,
                                                                 NContact* t_out /* 2 * MP per candidate */) {
  constexpr int kTerrainContacts = 2 * MP;  // per (body, face): a capsule part emits up to 2
  __shared__ uint8_t s_nc[MP][kBlock];
  const uint32_t m = *m_ptr, p_lo = blockIdx.x * kBlock;
  if (p_lo >= m) return;
  const uint32_t here = min((uint32_t)kBlock, m - p_lo);
  const V3 mx = mk3(M.x[0], M.x[1], M.x[2]);
  for (uint32_t w = threadIdx.x; w < here * (uint32_t)MP; w += kBlock) {
    const uint32_t a = w / here, q = w - a * here, p = p_lo + q;
    const uint32_t i = t_owner[p], f = t_cand[p];
    int nc = 0;
    Comp Pa;
    V3 ci;
    if (!(f & 0x80000000u) && load_part(B, i, a, &Pa, &ci)) {  // (a flagged candidate is a component of a static obstacle: k_narrow_obstacles)
      const V3 vA = xyz(B.delta[i]);
      const uint4 fi = M.faces[f];
      const Triangle tri = mkt(xyz(M.verts[fi.x]) + mx, xyz(M.verts[fi.y]) + mx, xyz(M.verts[fi.z]) + mx);  // mesh.rs:122-126
      LocalContact lc[2];
      nc = comp_tri_local_at(Pa, vA, tri, mx, ci, lc);
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        if (k < nc) {
          NContact o; o.la = mk4(lc[k].la, lc[k].g.t); o.lb = mk4(lc[k].lb, 0.0f); o.n = mk4(lc[k].g.n, 0.0f);
          t_out[(size_t)kTerrainContacts * p + 2u * a + (uint32_t)k] = o;
        }
      }
    }
    s_nc[a][q] = (uint8_t)nc;
  }
  __syncthreads();
  const uint32_t q = threadIdx.x;
  if (q >= here) return;
  const uint32_t p = p_lo + q;
  if (t_cand[p] & 0x80000000u) return;
  uint32_t cnt = 0;
#pragma unroll
  for (int a = 0; a < MP; ++a) {
    const uint32_t nc = s_nc[a][q];
    for (uint32_t k = 0; k < nc; ++k) {  // (packing moves a contact to a slot at or before its own: read, then write)
      const uint32_t src = 2u * (uint32_t)a + k;
      if (src != cnt) { const NContact o = t_out[(size_t)kTerrainContacts * p + src]; t_out[(size_t)kTerrainContacts * p + cnt] = o; }
      ++cnt;
    }
  }
  t_nc[p] = cnt;
}

// ---- (r06) worlds with a body of MORE than kMaxParts components (up to kBigParts; SURVEY 8f-1: the reference's Compound over any number of
// components, compound.rs:232-352, is a static shape with a BVH inside; a dynamic body of many parts is this build's definition, stated by the
// oracle: every pair of parts in order - parts of i outer - through Contacts, ContactPruner::push, Manifold::from).  A body's internal tree is
// how one CPU thread avoids part pairs that are far apart; here a WAVE takes a candidate pair of bodies and its lanes the part pairs, sixty-four
// at a time in the oracle's order: the bounding-sphere reject (comp_pair_far) ends most of them, the contacts of the rest are packed in lane
// order - which is the order the pruner has to see - into a short list in LDS, and lane 0 runs the pruner over it.  Output as
// k_narrow_pairs_parts<kMaxParts>: up to 16 contacts per candidate.  More than kBigRaw raw contacts or a manifold of more than 16: *over.
constexpr uint32_t kBigRaw = 64;   // raw contacts of one pair of bodies the wave's list holds
constexpr uint32_t kBigKeep = 16;  // contacts of one manifold (the stride of p_out)
static_assert(kBigParts <= 64, "k_narrow_terrain_big: a lane per part");
__global__ __launch_bounds__(kBlock) void k_narrow_pairs_big(Bodies B, const uint32_t* m_ptr, const uint32_t* p_owner, const uint32_t* p_cand, uint32_t* p_nc,
                                                             NContact* p_out /* kBigKeep per candidate */, uint32_t* over) {
  __shared__ NContact s_raw[kBlock / 64][kBigRaw];
  __shared__ uint32_t s_keep[kBlock / 64][kBigKeep];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t m = *m_ptr, nwaves = gridDim.x * (kBlock / 64);
  for (uint32_t p = blockIdx.x * (kBlock / 64) + (uint32_t)wv; p < m; p += nwaves) {
    const uint32_t i = p_owner[p], j = p_cand[p];
    {  // the tight boxes of this tick (k_narrow_pairs_parts' first step)
      const float4 ca = B.tb_c[i], ra = B.tb_r[i], cb = B.tb_c[j], rb = B.tb_r[j];
      const float slack = 1e-3f + 1e-5f * (fabs_rs(ca.x) + fabs_rs(ca.y) + fabs_rs(ca.z) + fabs_rs(cb.x) + fabs_rs(cb.y) + fabs_rs(cb.z));
      if (fabs_rs(ca.x - cb.x) > ra.x + rb.x + slack || fabs_rs(ca.y - cb.y) > ra.y + rb.y + slack || fabs_rs(ca.z - cb.z) > ra.z + rb.z + slack) {
        if (lane == 0) p_nc[p] = 0u;
        continue;
      }
    }
    const uint32_t pci = B.pcount[i], pcj = B.pcount[j];
    const uint32_t na = pci ? pci : 1u, nb = pcj ? pcj : 1u, total = na * nb;
    const V3 vA = xyz(B.delta[i]), vB = xyz(B.delta[j]);
    uint32_t nraw = 0;
    for (uint32_t w0 = 0; w0 < total; w0 += 64u) {
      const uint32_t w = w0 + (uint32_t)lane;
      bool hit = false;
      NContact raw;
      raw.la = raw.lb = raw.n = make_float4(0, 0, 0, 0);
      if (w < total) {
        const uint32_t a = w / nb, b = w - a * nb;
        float4 a0, a1, b0, b1;
        if (pci) world_part(B, i, a, pci, a0, a1); else { a0 = B.col0[i]; a1 = B.col1[i]; }
        if (pcj) world_part(B, j, b, pcj, b0, b1); else { b0 = B.col0[j]; b1 = B.col1[j]; }
        Comp Pa, Pb;
        Pa.kind = (int)f2u(a1.w); Pa.p = xyz(a0); Pa.r = a0.w; Pa.d = xyz(a1);
        Pb.kind = (int)f2u(b1.w); Pb.p = xyz(b0); Pb.r = b0.w; Pb.d = xyz(b1);
        Contact c;
        if (!comp_pair_far(Pa, vA, Pb, vB) && comp_pair_contact(Pa, vA, Pb, vB, &c)) {
          hit = true;
          raw.la = mk4(c.a, c.t); raw.lb = mk4(c.b, 1.0f); raw.n = mk4(c.n, 0.0f);
        }
      }
      const unsigned long long mask = __ballot(hit);
      const uint32_t at = nraw + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
      if (hit && at < kBigRaw) s_raw[wv][at] = raw;
      nraw += (uint32_t)__popcll(mask);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // (a wave's LDS accesses are served in order; it reads only what it wrote)
    if (lane != 0) continue;
    if (nraw > kBigRaw) { atomicOr(over, 1u); p_nc[p] = 0u; continue; }
    // ContactPruner::push (manifold.rs:72-102) over the raw contacts in order, Manifold::from(pruner) (:131-148): k_narrow_pairs_parts' third
    // step with the kept contacts' list positions in LDS
    const V3 ci = pci ? xyz(B.col0[i]) : comp_center(load_comp(B, i)), cj = pcj ? xyz(B.col0[j]) : comp_center(load_comp(B, j));
    auto local_of = [&](const NContact& r) -> LocalContact {
      Contact c; c.a = xyz(r.la); c.b = xyz(r.lb); c.n = xyz(r.n); c.t = r.la.w;
      LocalContact nc; nc.la = c.a + -(ci + vA * c.t); nc.lb = c.b + -(cj + vB * c.t); nc.g = c;
      return nc;
    };
    float min_t = kInf;
    uint32_t cnt = 0;
    bool too_many = false;
    for (uint32_t e = 0; e < nraw; ++e) {
      const LocalContact nc = local_of(s_raw[wv][e]);
      if (nc.g.t < min_t - kCollisionEps) { cnt = 1; s_keep[wv][0] = e; min_t = nc.g.t; continue; }
      if (nc.g.t > min_t + kCollisionEps) continue;
      bool merged = false;
      for (uint32_t k = 0; k < cnt && !merged; ++k) {
        const LocalContact kc = local_of(s_raw[wv][s_keep[wv][k]]);
        const V3 ra = nc.g.a - kc.g.a, rb = nc.g.b - kc.g.b;
        if (mag2(ra) <= kPersistentThresholdSq || mag2(rb) <= kPersistentThresholdSq) {
          const float prev = mag2(kc.la) + mag2(kc.lb), cur = mag2(nc.la) + mag2(nc.lb);
          if (prev < cur) s_keep[wv][k] = e;
          merged = true;
        }
      }
      if (!merged) { if (cnt < kBigKeep) s_keep[wv][cnt++] = e; else too_many = true; }
    }
    if (too_many) { atomicOr(over, 2u); p_nc[p] = 0u; continue; }
    p_nc[p] = cnt;
    if (cnt == 0) continue;
    V3 sum = mk3(0.0f, 0.0f, 0.0f);
    for (uint32_t k = 0; k < cnt; ++k) sum = sum + xyz(s_raw[wv][s_keep[wv][k]].n);
    const V3 avg = sum / (float)cnt;
    for (uint32_t k = 0; k < cnt; ++k) {
      const LocalContact kc = local_of(s_raw[wv][s_keep[wv][k]]);
      NContact o; o.la = mk4(kc.la, min_t); o.lb = mk4(kc.lb, 0.0f); o.n = mk4(avg, 0.0f);
      p_out[(size_t)kBigKeep * p + k] = o;
    }
  }
}
// ... and against the faces of the mesh: a wave per (body, face) candidate, a lane per part; every contact is its own constraint, in the
// order of the parts (k_narrow_terrain_parts' output, `stride` = 2 x the most parts of any body of the world).  A part whose reach ends
// short of the face (comp_tri_far) is not walked through the reference's tests.
__global__ __launch_bounds__(kBlock) void k_narrow_terrain_big(Bodies B, TerrainDev M, const uint32_t* m_ptr, const uint32_t* t_owner, const uint32_t* t_cand,
                                                               uint32_t* t_nc, NContact* t_out, uint32_t stride) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t m = *m_ptr, nwaves = gridDim.x * (kBlock / 64);
  const V3 mx = mk3(M.x[0], M.x[1], M.x[2]);
  for (uint32_t p = blockIdx.x * (kBlock / 64) + (uint32_t)wv; p < m; p += nwaves) {
    const uint32_t i = t_owner[p], f = t_cand[p];
    if (f & 0x80000000u) continue;  // (a component of a static obstacle: k_narrow_obstacles)
    const uint32_t pc = B.pcount[i], np = pc ? pc : 1u;
    int nc = 0;
    LocalContact lc[2];
    if ((uint32_t)lane < np) {
      Comp Pa;
      V3 ci;
      (void)load_part(B, i, (uint32_t)lane, &Pa, &ci);
      const V3 vA = xyz(B.delta[i]);
      const uint4 fi = M.faces[f];
      const Triangle tri = mkt(xyz(M.verts[fi.x]) + mx, xyz(M.verts[fi.y]) + mx, xyz(M.verts[fi.z]) + mx);  // mesh.rs:122-126
      if (!comp_tri_far(Pa, vA, tri)) nc = comp_tri_local_at(Pa, vA, tri, mx, ci, lc);
    }
    const unsigned long long m1 = __ballot(nc >= 1), m2 = __ballot(nc == 2), lt = (1ull << lane) - 1ull;
    const uint32_t at = (uint32_t)__popcll(m1 & lt) + (uint32_t)__popcll(m2 & lt);
#pragma unroll
    for (int k = 0; k < 2; ++k) {  // (unrolled: an index the compiler does not know keeps the pair in scratch memory)
      if (k < nc) {
        NContact o; o.la = mk4(lc[k].la, lc[k].g.t); o.lb = mk4(lc[k].lb, 0.0f); o.n = mk4(lc[k].g.n, 0.0f);
        t_out[(size_t)stride * p + at + (uint32_t)k] = o;
      }
    }
    if (lane == 0) t_nc[p] = (uint32_t)__popcll(m1) + (uint32_t)__popcll(m2);
  }
}

// Bin candidate ids by pair type (only launched for scenes that mix spheres and capsules).
__global__ __launch_bounds__(kBlock) void k_bin_pairs(Bodies B, const uint32_t* m_ptr, uint32_t stride, const uint32_t* p_owner,
                                                      const uint32_t* p_cand, uint32_t* lists /* 4 x stride */, uint32_t* counts /* 4 */) {
  uint32_t p = blockIdx.x * kBlock + threadIdx.x;
  const uint32_t m = *m_ptr;
  int type = -1;
  if (p < m) type = (int)(f2u(B.col1[p_owner[p]].w) * 2u + f2u(B.col1[p_cand[p]].w));
  for (int ty = 0; ty < 4; ++ty) {  // wave-aggregated append: one atomic per wave per type
    unsigned long long mask = __ballot(type == ty);
    if (mask == 0) continue;
    uint32_t base = 0;
    int lane = threadIdx.x & 63;
    int leader = __ffsll((long long)mask) - 1;
    if (lane == leader) base = atomicAdd(&counts[ty], (uint32_t)__popcll(mask));
    base = __shfl(base, leader);
    if (type == ty) lists[(size_t)ty * stride + base + __popcll(mask & ((1ull << lane) - 1ull))] = p;
  }
}
__global__ __launch_bounds__(kBlock) void k_bin_terrain(Bodies B, const uint32_t* m_ptr, uint32_t stride, const uint32_t* t_owner,
                                                        uint32_t* lists /* 2 x stride */, uint32_t* counts /* 2 */) {
  uint32_t p = blockIdx.x * kBlock + threadIdx.x;
  const uint32_t m = *m_ptr;
  int type = -1;
  if (p < m) type = (int)f2u(B.col1[t_owner[p]].w);
  for (int ty = 0; ty < 2; ++ty) {
    unsigned long long mask = __ballot(type == ty);
    if (mask == 0) continue;
    uint32_t base = 0;
    int lane = threadIdx.x & 63;
    int leader = __ffsll((long long)mask) - 1;
    if (lane == leader) base = atomicAdd(&counts[ty], (uint32_t)__popcll(mask));
    base = __shfl(base, leader);
    if (type == ty) lists[(size_t)ty * stride + base + __popcll(mask & ((1ull << lane) - 1ull))] = p;
  }
}

// Per body: number of constraints it inserts (terrain contacts first, then partners) and the
// running offset of each candidate inside the body's block.
__global__ __launch_bounds__(kBlock) void k_count_contacts(StepCounts* sc, uint32_t n, const uint32_t* t_off, const uint32_t* p_off,
                                                           const uint32_t* t_nc, const uint32_t* p_nc, const uint32_t* p_cand, uint32_t* t_pre,
                                                           uint32_t* p_pre, uint32_t* cnt, int keep_order, uint32_t* tcn, const uint32_t* ext) {
  constexpr int kHitCap = 12;  // a sphere touches at most 12 equal ones
  __shared__ uint32_t s_j[kHitCap][kBlock], s_p[kHitCap][kBlock];
  const int tid = threadIdx.x;
  __shared__ uint32_t s_ct;
  uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (tid == 0) s_ct = 0;
  __syncthreads();
  bool active = i < n;
  if (active && sc->fail) { cnt[i] = 0; tcn[i] = 0; active = false; }
  if (active) {
  uint32_t run = 0;
  for (uint32_t p = t_off[i]; p < t_off[i + 1]; ++p) { t_pre[p] = run; run += t_nc[p]; }
  if (run) atomicAdd(&s_ct, run);  // only the total is needed: one global atomic per block
  // partner contacts are numbered in ascending partner order - by the partners' order ids - (the canonical insertion order); the candidate list itself
  // is in discovery order, and only a few of its ~10 entries are contacts (at most one per partner): collect them, then
  // rank them among themselves
  const uint32_t lo = p_off[i], hi = p_off[i + 1];
  uint32_t h = 0, total = 0;
  for (uint32_t base = lo; base < hi; base += 4) {  // four counts per round trip
    uint32_t nc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) nc[k] = base + k < hi ? (p_nc ? p_nc[base + k] : 1u) : 0u;  // p_nc == nullptr: the list holds contacts only
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (nc[k]) {
        if (h < (uint32_t)kHitCap) { s_j[h][tid] = order_id(ext, p_cand[base + k]); s_p[h][tid] = (base + k) | (nc[k] << 27); }
        ++h;
        total += nc[k];
      }
    }
  }
  if (h <= (uint32_t)kHitCap) {
    for (uint32_t a = 0; a < h; ++a) {
      const uint32_t j = s_j[a][tid];
      uint32_t before = 0;
      // a partner's contacts (1, or up to 4 for bodies of several parts); keep_order: the list IS the insertion order (world.rs order)
      for (uint32_t q = 0; q < h; ++q) before += (keep_order ? q < a : s_j[q][tid] < j) ? (s_p[q][tid] >> 27) : 0u;
      p_pre[s_p[a][tid] & 0x07FFFFFFu] = run + before;  // (5 bits of contact count - up to 16 for bodies of four parts - above a 27-bit list position)
    }
  } else {  // a crowded body: the same by rescanning its list
    for (uint32_t p = lo; p < hi; ++p) {
      if (p_nc && p_nc[p] == 0) continue;
      const uint32_t j = order_id(ext, p_cand[p]);
      uint32_t before = 0;
      for (uint32_t q = lo; q < hi; ++q) before += (keep_order ? q < p : order_id(ext, p_cand[q]) < j) ? (p_nc ? p_nc[q] : 1u) : 0u;
      p_pre[p] = run + before;
    }
  }
  cnt[i] = run + total;
  tcn[i] = run;  // the body's terrain constraints come first in its range (k_chain_rows)
  }
  __syncthreads();
  if (tid == 0 && s_ct) atomicAdd(&sc->ct_sum, s_ct);
}

// A world of spheres over a small mesh (the headline workload; the fused broadphase has listed CONTACTS, the terrain rows hold face
// ids in the mesh's DFS order): k_rows_to_csr, k_narrow_terrain<0> and k_count_contacts in one launch, a thread per body (round 3:
// two launches less per tick).  The body's terrain row goes to the candidate list and through the sphere-triangle test on the way;
// its partner row goes to the list and is ranked by the partners' order ids; the body's constraint count follows.
__global__ __launch_bounds__(kBlock) void k_lists_spheres(Bodies B, TerrainDev M, StepCounts* sc, uint32_t n, uint32_t cap_row_t, const uint32_t* rows_t,
                                                          const uint32_t* rows_p, const uint32_t* t_off, const uint32_t* p_off, uint32_t* t_cand,
                                                          uint32_t* t_owner, uint32_t* t_nc, NContact* t_out /* 2 per candidate */, uint32_t* t_pre,
                                                          uint32_t* p_cand, uint32_t* p_owner, uint32_t* p_pre, uint32_t* cnt, uint32_t* tcn,
                                                          const uint32_t* ext) {
  __shared__ uint32_t s_ct;
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (threadIdx.x == 0) s_ct = 0;
  __syncthreads();
  bool active = i < n;
  if (active && sc->fail) { cnt[i] = 0; tcn[i] = 0; active = false; }
  if (active) {
    const uint32_t tb = t_off[i], nt = t_off[i + 1] - tb, pb = p_off[i], np = p_off[i + 1] - pb;
    if (nt > cap_row_t || np > (uint32_t)kRowCap) { cnt[i] = 0; tcn[i] = 0; }  // (an overflowed row: the flag is up, the host re-runs the phase)
    else {
      uint32_t run = 0;
      if (nt) {  // Mesh::contacts' faces, in its order; every contact its own constraint (world.rs:243-251)
        V3 vA;
        Comp A = load_comp_moving(B, i, &vA);
        A.kind = KIND_SPHERE;
        const V3 mx = mk3(M.x[0], M.x[1], M.x[2]);
        const uint32_t* rt = rows_t + (size_t)i * cap_row_t;
        for (uint32_t a = 0; a < nt; ++a) {
          const uint32_t f = rt[a], p = tb + a;
          t_cand[p] = f; t_owner[p] = i;
          const uint4 fi = M.faces[f];
          Triangle tri = mkt(xyz(M.verts[fi.x]) + mx, xyz(M.verts[fi.y]) + mx, xyz(M.verts[fi.z]) + mx);  // mesh.rs:122-126
          LocalContact lc[2];
          const int nc = comp_tri_local(A, vA, tri, mx, lc);
          t_nc[p] = (uint32_t)nc;
          t_pre[p] = run;
#pragma unroll
          for (int k = 0; k < 2; ++k)
            if (k < nc) { NContact o; o.la = mk4(lc[k].la, lc[k].g.t); o.lb = mk4(lc[k].lb, 0.0f); o.n = mk4(lc[k].g.n, 0.0f); t_out[2 * p + k] = o; }  // Manifold::from(lc) manifold.rs:120-128
          run += (uint32_t)nc;
        }
        if (run) atomicAdd(&s_ct, run);
      }
      // the partners (all of them contacts, one each): to the list in discovery order, numbered in ascending order id - the canonical
      // insertion order
      const uint4* rp = reinterpret_cast<const uint4*>(rows_p + (size_t)i * kRowCap);
      uint32_t o_id[12], slot[12];  // (a sphere touches at most 12 equal ones; more: the rescan below)
      for (uint32_t a = 0; a < np; a += 4) {
        const uint4 v = rp[a >> 2];
        const uint32_t e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (a + k < np) { p_cand[pb + a + k] = e[k]; p_owner[pb + a + k] = i; }
      }
      if (np <= 12u) {
#pragma unroll
        for (int a = 0; a < 12; ++a) { slot[a] = (uint32_t)a < np ? rows_p[(size_t)i * kRowCap + a] : 0u; o_id[a] = (uint32_t)a < np ? order_id(ext, slot[a]) : 0xFFFFFFFFu; }
#pragma unroll
        for (int a = 0; a < 12; ++a) {
          if ((uint32_t)a < np) {
            uint32_t before = 0;
#pragma unroll
            for (int q = 0; q < 12; ++q) before += o_id[q] < o_id[a] ? 1u : 0u;
            p_pre[pb + a] = run + before;
          }
        }
      } else {
        for (uint32_t a = 0; a < np; ++a) {
          const uint32_t oa = order_id(ext, rows_p[(size_t)i * kRowCap + a]);
          uint32_t before = 0;
          for (uint32_t q = 0; q < np; ++q) before += order_id(ext, rows_p[(size_t)i * kRowCap + q]) < oa ? 1u : 0u;
          p_pre[pb + a] = run + before;
        }
      }
      cnt[i] = run + np;
      tcn[i] = run;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0 && s_ct) atomicAdd(&sc->ct_sum, s_ct);
}

// ------------------------------------------------------------------------------------------
// ContactConstraint (solver.rs:82-93, 256-262), single contact.  96-byte record.
// ------------------------------------------------------------------------------------------
struct CRec {
  uint32_t a, b;       // body indices; b = kNone for RigidBodyRef::Static
  float n[3], t0[3], t1[3], ra[3], rb[3];
  float bias, nmass, tmass0, tmass1;
  uint32_t pad0;
  float nimp;          // ContactState::normal_impulse            (word 22: 8-byte aligned with round)
  uint32_t round;      // solver iterations already applied in the current Solver::solve call (launch-per-frontier mode)
  uint32_t pad1;
  uint32_t indeg;      // predecessors still pending for the next round (atomics; launch-per-frontier mode)
  uint32_t pad2;
  float friction;      // dead state in the reference (solver.rs:223-226), kept for read-back
  uint32_t pad3[4];
};
// Dependency links live outside the records, in compact arrays (ConsLinks): building them touches 4-16 bytes per
// constraint instead of a 128-byte line.
struct ConsLinks {
  uint2* ab;           // (a, b) of every constraint
  uint2* succ;         // successor words on body a / body b (see k_chain)
  uint8_t* pred;       // pred[2c + role] = 1 if the constraint has a predecessor on that body inside one iteration
};
static_assert(sizeof(CRec) == 128, "CRec is one 128-byte line");

struct BodyDyn { V3 v, w; float im; M3 I; };
__device__ __forceinline__ BodyDyn load_dyn(const float4* srec, uint32_t i) {
  float4 s0 = srec[4 * i], s1 = srec[4 * i + 1], s2 = srec[4 * i + 2], s3 = srec[4 * i + 3];
  BodyDyn d;
  d.v = mk3(s0.x, s0.y, s0.z); d.w = mk3(s0.w, s1.x, s1.y); d.im = s1.z;
  d.I = m3_cols(mk3(s1.w, s2.x, s2.y), mk3(s2.z, s2.w, s3.x), mk3(s3.y, s3.z, s3.w));
  return d;
}
__device__ __forceinline__ BodyDyn static_dyn() {  // physics.rs:289-302
  BodyDyn d; d.v = mk3(0, 0, 0); d.w = mk3(0, 0, 0); d.im = 0.0f;
  d.I = m3_cols(mk3(0, 0, 0), mk3(0, 0, 0), mk3(0, 0, 0));
  return d;
}
__device__ __forceinline__ void st3(float* p, V3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }
__device__ __forceinline__ V3 ld3(const float* p) { return mk3(p[0], p[1], p[2]); }

// ContactConstraint::new solver.rs:101-191 for one contact.
// (the manifold's tangent_vector is the caller's: Manifold::from computes it with compute_basis, a caller-built Manifold
// may hold anything - ContactConstraint::new only reads the fields)
__device__ __forceinline__ CRec make_constraint_basis(uint32_t ia, uint32_t ib, const BodyDyn& A, V3 xa, float rest_a, float fric_a,
                                                      const BodyDyn& Bd, V3 xb, float rest_b, float fric_b, V3 normal, V3 t0, V3 t1,
                                                      V3 ra, V3 rb, float dt, float baumgarte, float slop) {
  CRec c;
  c.a = ia; c.b = ib;
  float restitution = fmax_rs(rest_a, rest_b);
  c.friction = __builtin_sqrtf(fric_a * fric_b);
  V3 ca = ra + xa, cb = rb + xb;
  V3 ra_cn = cross(ra, normal), rb_cn = cross(rb, normal);
  float pen = dot(cb - ca, normal);
  V3 dv = Bd.v + cross(Bd.w, rb) - A.v - cross(A.w, ra);
  float rel_v = dot(dv, normal);
  float bias = -baumgarte / dt * (pen > 0.0f ? 0.0f : pen + slop) + (rel_v < -1.0f ? -restitution * rel_v : 0.0f);
  c.nmass = 1.0f / (A.im + dot(ra_cn, A.I * ra_cn) + Bd.im + dot(rb_cn, Bd.I * rb_cn));
  V3 ra_ct = cross(ra, t0), rb_ct = cross(rb, t0);
  c.tmass0 = 1.0f / (A.im + dot(ra_ct, A.I * ra_ct) + Bd.im + dot(rb_ct, Bd.I * rb_ct));
  ra_ct = cross(ra, t1); rb_ct = cross(rb, t1);
  c.tmass1 = 1.0f / (A.im + dot(ra_ct, A.I * ra_ct) + Bd.im + dot(rb_ct, Bd.I * rb_ct));
  c.bias = bias;
  c.nimp = 0.0f;
  c.round = 0; c.indeg = 0; c.pad0 = c.pad1 = c.pad2 = 0;
  c.pad3[0] = c.pad3[1] = c.pad3[2] = c.pad3[3] = 0;
  st3(c.n, normal); st3(c.t0, t0); st3(c.t1, t1); st3(c.ra, ra); st3(c.rb, rb);
  return c;
}
__device__ __forceinline__ CRec make_constraint(uint32_t ia, uint32_t ib, const BodyDyn& A, V3 xa, float rest_a, float fric_a,
                                                const BodyDyn& Bd, V3 xb, float rest_b, float fric_b, V3 normal, V3 ra, V3 rb,
                                                float dt, float baumgarte, float slop) {
  V3 t0, t1;
  compute_basis(normal, &t0, &t1);  // manifold.rs:125,144
  return make_constraint_basis(ia, ib, A, xa, rest_a, fric_a, Bd, xb, rest_b, fric_b, normal, t0, t1, ra, rb, dt, baumgarte, slop);
}

__device__ __forceinline__ void store_crec(CRec* dst, const CRec& c) {
  const float4* s = reinterpret_cast<const float4*>(&c);
  float4* d = reinterpret_cast<float4*>(dst);
#pragma unroll
  for (int k = 0; k < 8; ++k) d[k] = s[k];
}
// the whole 128-byte line
__device__ __forceinline__ CRec load_crec(const CRec* src) {
  CRec c;
  const float4* s = reinterpret_cast<const float4*>(src);
  float4* d = reinterpret_cast<float4*>(&c);
#pragma unroll
  for (int k = 0; k < 8; ++k) d[k] = s[k];
  return c;
}
static_assert(offsetof(CRec, nimp) == 88 && offsetof(CRec, round) == 92 && offsetof(CRec, indeg) == 100, "CRec layout");
// what ContactConstraint::solve reads: the first 96 bytes (through nimp / round)
__device__ __forceinline__ CRec load_crec_solve(const CRec* src) {
  CRec c;
  const float4* s = reinterpret_cast<const float4*>(src);
  float4* d = reinterpret_cast<float4*>(&c);
#pragma unroll
  for (int k = 0; k < 6; ++k) d[k] = s[k];
  d[6] = make_float4(0, 0, 0, 0); d[7] = make_float4(0, 0, 0, 0);
  return c;
}

// What ContactConstraint::new needs of a body besides srec: collider word 0 (sphere centre, radius), delta (motion, friction),
// einfo (x + delta, restitution).  From the packed copy (one sector) when the host vouches for it, else from the arrays.
struct BodyPack { float4 c0, dl, ei; };
__device__ __forceinline__ BodyPack load_pack(const Bodies& B, uint32_t i, bool with_collider) {
  BodyPack P;
  if (B.bpk) { P.c0 = B.bpk[4 * i]; P.dl = B.bpk[4 * i + 1]; P.ei = B.bpk[4 * i + 2]; }
  else { P.c0 = with_collider ? B.col0[i] : make_float4(0, 0, 0, 0); P.dl = B.delta[i]; P.ei = B.einfo[i]; }
  return P;
}

// The terrain side of the constraint setup (one thread per terrain candidate): runs as the first blocks of k_setup_pairs' launch.
struct TerrainSetup {
  TerrainDev M;
  const uint32_t *t_owner, *t_nc, *t_pre;
  const NContact* t_in;
  uint32_t in_stride;
  uint32_t blocks;  // blocks of the launch that work on terrain candidates (0: the world has no terrain)
  const uint32_t* t_cand;       // with obstacles: the candidates (a flagged one is a component of obstacle (f >> 23) & 255) ...
  const float4* obs_center;     // ... and the obstacles' centres (their displacements): the Static body's centre of such a constraint
};
__device__ __forceinline__ void setup_terrain_one(const Bodies& B, const TerrainSetup& T, const StepCounts* sc, uint32_t p, const uint32_t* base, float dt,
                                                  float baumgarte, float slop, CRec* cons, uint2* ab) {
  if (p >= sc->Mt) return;
  uint32_t nc = T.t_nc[p];
  if (nc == 0) return;
  uint32_t i = T.t_owner[p];
  BodyDyn A = load_dyn(B.srec, i), S = static_dyn();
  const BodyPack Pa = load_pack(B, i, false);
  float4 ea = Pa.ei;
  V3 center = mk3(T.M.x[0], T.M.x[1], T.M.x[2]);  // Static{ center: terrain.center(), friction: 0.0 } world.rs:247
  if (T.obs_center) { const uint32_t f = T.t_cand[p]; if (f & 0x80000000u) center = xyz(T.obs_center[(f >> 23) & 0xFFu]); }
  for (uint32_t k = 0; k < nc; ++k) {
    NContact in = T.t_in[(size_t)T.in_stride * p + k];
    CRec r = make_constraint(i, kNone, A, xyz(ea), ea.w, Pa.dl.w, S, center, 0.0f, 0.0f, xyz(in.n), xyz(in.la), xyz(in.lb), dt,
                             baumgarte, slop);
    store_crec(&cons[base[i] + T.t_pre[p] + k], r);
    ab[base[i] + T.t_pre[p] + k] = make_uint2(i, kNone);
  }
}

// SPHERES = true: a world of spheres whose broadphase ran the sphere-sphere test itself (k_pair_grid<true>): the list
// holds contacts only, one per pair, and the contact is computed here from the colliders (no k_narrow_pairs pass, no
// NContact round trip through memory).  `flag` is raised if the two evaluations of the same test ever disagreed.
template <bool SPHERES>
__global__ __launch_bounds__(kBlock) void k_setup_pairs(Bodies B, const StepCounts* sc, const uint32_t* p_owner, const uint32_t* p_cand,
                                                        const uint32_t* p_nc, const uint32_t* p_pre, const NContact* p_in,
                                                        const uint32_t* base, float dt, float baumgarte, float slop,
                                                        CRec* cons, uint2* ab, uint32_t* degb, RevEnt* rev, uint32_t rev_cap,
                                                        uint32_t* rev_flag, uint32_t in_stride, uint32_t* flag, TerrainSetup TS, const uint32_t* ext) {
  if (blockIdx.x < TS.blocks) {  // the terrain candidates' constraints (k_setup_terrain's work, without a launch of its own)
    setup_terrain_one(B, TS, sc, blockIdx.x * kBlock + threadIdx.x, base, dt, baumgarte, slop, cons, ab);
    return;
  }
  uint32_t p = (blockIdx.x - TS.blocks) * kBlock + threadIdx.x;
  if (SPHERES) {
    // One contact per listed pair.  The 128-byte records leave through LDS: a lane storing its own record issues eight
    // 16-byte stores 128 bytes apart from its neighbours' (64 partial lines per instruction); handed round, consecutive lanes
    // store consecutive words of the same records (8 full lines per instruction).  A record's id is looked up per word, so
    // ids need not be contiguous (a body's terrain constraints sit between its neighbours' pair constraints).
    __shared__ float4 s_w[kBlock / 64][7 * 65];  // (the record's eighth 16-byte word is padding: not written)
    __shared__ uint32_t s_c[kBlock / 64][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t c = kNone;
    if (p < sc->Mp) {
      const uint32_t i = p_owner[p], j = p_cand[p];
      const BodyPack Pa = load_pack(B, i, true), Pb = load_pack(B, j, true);
      Comp Ca, Cb;  // as k_narrow_pairs<0, 0>
      Ca.p = xyz(Pa.c0); Ca.r = Pa.c0.w; Ca.d = mk3(0.0f, 0.0f, 0.0f); Ca.kind = KIND_SPHERE;
      Cb.p = xyz(Pb.c0); Cb.r = Pb.c0.w; Cb.d = mk3(0.0f, 0.0f, 0.0f); Cb.kind = KIND_SPHERE;
      LocalContact lc;
      if (!comp_pair_local(Ca, xyz(Pa.dl), Cb, xyz(Pb.dl), &lc)) {
        *flag = 1u;
      } else {
        const V3 nrm = (mk3(0.0f, 0.0f, 0.0f) + lc.g.n) / 1.0f;  // Manifold::from(pruner) of one contact (manifold.rs:135-140)
        const BodyDyn A = load_dyn(B.srec, i), Bd = load_dyn(B.srec, j);
        c = base[i] + p_pre[p];
        const CRec r = make_constraint(i, j, A, xyz(Pa.ei), Pa.ei.w, Pa.dl.w, Bd, xyz(Pb.ei), Pb.ei.w, Pb.dl.w, nrm, lc.la, lc.lb, dt, baumgarte, slop);
        const float4* rw = reinterpret_cast<const float4*>(&r);
#pragma unroll
        for (int k = 0; k < 7; ++k) s_w[wv][k * 65 + lane] = rw[k];
        ab[c] = make_uint2(i, j);
        // body j's row of the constraints it takes part in as `b` (k_chain_rows); as `a` a body owns a contiguous id range
        const uint32_t pos = atomicAdd(&degb[j], 1u);
        if (pos < rev_cap) rev[(size_t)j * rev_cap + pos] = RevEnt{c, i, order_id(ext, i), 0u};
        else *rev_flag = 1u;
      }
    }
    s_c[wv][lane] = c;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // (a wave's LDS accesses are served in order; the wave reads only what it wrote)
    float4* out = reinterpret_cast<float4*>(cons);
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int rec = it * 8 + (lane >> 3), k = lane & 7;
      const uint32_t cr = s_c[wv][rec];
      if (cr != kNone && k < 7) out[(size_t)cr * 8 + k] = s_w[wv][k * 65 + rec];
    }
    return;
  }
  if (p >= sc->Mp) return;
  const uint32_t nc = p_nc[p];
  if (nc == 0) return;
  uint32_t i = p_owner[p], j = p_cand[p];
  BodyPack Pa = load_pack(B, i, false), Pb = load_pack(B, j, false);
  BodyDyn A = load_dyn(B.srec, i), Bd = load_dyn(B.srec, j);
  float4 ea = Pa.ei, eb = Pb.ei;
  // A manifold of m contacts (bodies of several parts only) becomes m consecutive single-contact records that share
  // its normal and tangents: ContactConstraint::solve (solver.rs:219-248) handles the contacts of a constraint one after
  // the other on the same velocities, which is exactly what consecutive records do.
  for (uint32_t q = 0; q < nc; ++q) {
    const uint32_t c = base[i] + p_pre[p] + q;
    NContact k = p_in[(size_t)in_stride * p + q];
    CRec r = make_constraint(i, j, A, xyz(ea), ea.w, Pa.dl.w, Bd, xyz(eb), eb.w, Pb.dl.w, xyz(k.n), xyz(k.la), xyz(k.lb),
                             dt, baumgarte, slop);
    store_crec(&cons[c], r);
    ab[c] = make_uint2(i, j);
    // body j's row of the constraints it takes part in as `b` (k_chain_rows); as `a` a body owns a contiguous id range
    uint32_t pos = atomicAdd(&degb[j], 1u);
    if (pos < rev_cap) rev[(size_t)j * rev_cap + pos] = RevEnt{c, i, order_id(ext, i), 0u};
    else *rev_flag = 1u;
  }
}

// A world of spheres over a small mesh (round 5): from the rows to the constraint records without candidate lists.
//   k_terrain_contacts   a thread per body with terrain faces: the sphere-triangle test on its row (Mesh::contacts' faces, in its
//                        order), the contacts parked in the terrain list's slots (handed out by an atomic counter), tcn = their number;
//   k_scan<1> (+ add)    base = exclusive prefix of p_cnt + tcn: the partner rows hold contacts only, so that IS the body's constraint
//                        count; its last thread writes the tick's StepCounts (caps_contacts);
//   k_contacts_spheres   a block per 256 consecutive bodies: ContactConstraint::new for the terrain contacts by their bodies' threads
//                        and for the partner contacts as a list in LDS in canonical order (ascending order id inside a body), dealt
//                        evenly to the threads.
// What this replaces - k_scan<2> (row offsets), k_lists_spheres, k_scan<1>, k_setup_pairs<true> - built t_cand / p_cand / ..._pre /
// ..._owner lists that nobody downstream reads.
#ifndef MGF_CS_ENT_CAP
#define MGF_CS_ENT_CAP 1536
#endif
constexpr uint32_t kCsEntCap = MGF_CS_ENT_CAP;  // partner contacts of a block staged per pass (a settled pile's block has ~900-1 100: with 1 024 most of its blocks listed a second window - r06: k_contacts_rows 93 -> 81 us there)
constexpr uint32_t kCsSumStride = 32; // words between the two global counters (their own cache lines)
#ifndef MGF_TC_LANES
#define MGF_TC_LANES 4
#endif
constexpr int kTcLanes = MGF_TC_LANES;  // lanes per body in k_terrain_contacts: a face each (a body near the box's floor or walls lists 1-4)
constexpr uint32_t kTcMesh = 64;        // faces / vertices of a mesh staged in LDS
// The bodies that list a face (near_list, compacted by k_integrate's tail), kTcLanes lanes each: a scan over all bodies - most of them
// nowhere near the mesh - took 21 us for what is 4 dependent round trips and a few hundred instructions per body.  tcn / tpos of the
// other bodies are zero (the tick's clearing launch).
struct TerrainContacts {
  TerrainDev M;
  const float4* near_list; const uint32_t* near_cnt;
  uint32_t n_faces, n_verts, cap_row_t, cap_t;
  const uint32_t* rows_t;
  uint32_t* sums;      // [0] candidates = slot allocator, [kCsSumStride] contacts
  NContact* t_out;     // 2 per slot; .lb.w of the first = the face's contact count
  uint32_t *tcn, *tpos;
  const uint32_t* guard;
  const float4* col1;  // GEN: the bodies' collider word 1 (d.xyz, kind) - the records of k_integrate's tail carry word 0 and the motion
  const uint32_t* pcount; const float4 *wp0, *wp1;  // GEN = 2: the parts of bodies of several components (Bodies)
  uint32_t check; uint32_t* flag;  // tests (option front_rows_check): the faces comp_tri_far drops are tested all the same; a contact among them raises *flag
};
// GEN (r06): bodies of any single-component kind - a lane per FACE through the body-triangle test (collision.rs:610-1086) instead of a face by
// the group's four lanes (tri_msphere_x4 is the sphere's); the parked slots are the same.
// GEN = 2: bodies of up to two components - every part against the face (k_narrow_terrain_parts<2>), up to four contacts per slot (t_out: 4 per slot).
template <int GEN>
__device__ __forceinline__ void terrain_contacts_job(const TerrainContacts& A, uint32_t job_block, uint32_t job_blocks) {
  const TerrainDev& M = A.M;
  const float4* near_list = A.near_list; const uint32_t n_faces = A.n_faces, n_verts = A.n_verts, cap_row_t = A.cap_row_t, cap_t = A.cap_t;
  const uint32_t* rows_t = A.rows_t; uint32_t* sums = A.sums; NContact* t_out = A.t_out; uint32_t* tcn = A.tcn; uint32_t* tpos = A.tpos;
  __shared__ uint4 s_face[kTcMesh];
  __shared__ float4 s_vert[kTcMesh];
  const uint32_t L = *A.guard ? 0u : *A.near_cnt;
  const uint32_t per_pass = job_blocks * (uint32_t)(kBlock / kTcLanes);
  if (job_block * (uint32_t)(kBlock / kTcLanes) >= L) return;
  const bool staged = n_faces <= kTcMesh && n_verts <= kTcMesh;
  if (staged) {
    for (uint32_t e = threadIdx.x; e < n_faces; e += kBlock) s_face[e] = M.faces[e];
    for (uint32_t e = threadIdx.x; e < n_verts; e += kBlock) s_vert[e] = M.verts[e];
    __syncthreads();
  }
  const int lane = threadIdx.x & 63;
  const uint32_t sub = threadIdx.x % (uint32_t)kTcLanes;
  const V3 mx = mk3(M.x[0], M.x[1], M.x[2]);
  // (the loop's trips are the same for every lane of a wave: the shuffles below need the whole wave)
  for (uint32_t e0 = job_block * (uint32_t)(kBlock / kTcLanes) + ((uint32_t)threadIdx.x & ~63u) / (uint32_t)kTcLanes; e0 < L; e0 += per_pass) {
    const uint32_t e = e0 + (uint32_t)lane / (uint32_t)kTcLanes;
    uint32_t i = 0, nt = 0, pk0 = 0, pk1 = 0;
    float4 r0 = make_float4(0, 0, 0, 0), r1 = r0;
    if (e < L) {
      r0 = near_list[3 * (size_t)e]; r1 = near_list[3 * (size_t)e + 1];
      const float4 r2 = near_list[3 * (size_t)e + 2];
      i = f2u(r1.w); nt = f2u(r2.x); pk0 = f2u(r2.y); pk1 = f2u(r2.z);
    }
    if (nt > cap_row_t) nt = 0;  // (an overflowed row: the flag is up, the host re-runs the phase)
    // the wave's slots with ONE atomic
    uint32_t run = 0, tp = 0;
    {
      const uint32_t mine = sub == 0u ? nt : 0u;
      uint32_t inc = mine;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { const uint32_t u = __shfl_up(inc, o); if (lane >= o) inc += u; }
      uint32_t wave_base = 0;
      if (lane == 63 && inc) wave_base = atomicAdd(&sums[0], inc);
      wave_base = __shfl(wave_base, 63);
      tp = __shfl(wave_base + inc - mine, lane & ~(kTcLanes - 1));
    }
    if (GEN == 2) {
      if (nt) {
        const V3 vA = xyz(r1);
        const uint32_t pc = A.pcount ? A.pcount[i] : 0u;
        Comp P0, P1;
        V3 centre;
        int np = 1;
        P1.kind = KIND_SPHERE; P1.p = mk3(0, 0, 0); P1.d = mk3(0, 0, 0); P1.r = 0.0f;
        if (pc == 0u) { const float4 c1 = A.col1[i]; P0.p = xyz(r0); P0.r = r0.w; P0.d = xyz(c1); P0.kind = (int)f2u(c1.w); centre = comp_center(P0); }
        else {
          centre = xyz(r0); np = (int)min(pc, 2u);
          const float4 a0 = A.wp0[kMaxParts * (size_t)i], b0 = A.wp1[kMaxParts * (size_t)i];
          P0.kind = (int)f2u(b0.w); P0.p = xyz(a0); P0.r = a0.w; P0.d = xyz(b0);
          if (pc > 1u) { const float4 a1 = A.wp0[kMaxParts * (size_t)i + 1], b1 = A.wp1[kMaxParts * (size_t)i + 1]; P1.kind = (int)f2u(b1.w); P1.p = xyz(a1); P1.r = a1.w; P1.d = xyz(b1); }
        }
        const uint32_t* rt = rows_t + (size_t)i * cap_row_t;
        const bool bytes = pk0 != 0xFFFFFFFFu || pk1 != 0xFFFFFFFFu;
        for (uint32_t a = sub; a < nt; a += (uint32_t)kTcLanes) {  // the body's faces dealt to the group's lanes
          const uint32_t f = bytes ? ((a < 4u ? pk0 >> (8u * a) : pk1 >> (8u * (a - 4u))) & 255u) : rt[a];
          const uint4 fi = staged ? s_face[f] : M.faces[f];
          const Triangle tri = staged ? mkt(xyz(s_vert[fi.x]) + mx, xyz(s_vert[fi.y]) + mx, xyz(s_vert[fi.z]) + mx)
                                      : mkt(xyz(M.verts[fi.x]) + mx, xyz(M.verts[fi.y]) + mx, xyz(M.verts[fi.z]) + mx);  // mesh.rs:122-126
          LocalContact l0[2], l1[2];
          const bool f0 = comp_tri_far(P0, vA, tri), f1 = np >= 2 && comp_tri_far(P1, vA, tri);
          const int n0 = (f0 && !A.check) ? 0 : comp_tri_local_at(P0, vA, tri, mx, centre, l0);
          const int n1 = (np < 2 || (f1 && !A.check)) ? 0 : comp_tri_local_at(P1, vA, tri, mx, centre, l1);
          if (A.check && ((f0 && n0) || (f1 && n1))) *A.flag = 1u;  // (tests: the cheap reject dropped a face that reports a contact)
          const int nc = n0 + n1;
          if (tp + a < cap_t) {
            // the contacts packed in order, the parts one after the other (k_narrow_terrain_parts); the first record carries the count
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              if (k == 0 || k < nc) {
                NContact o;
                o.la = make_float4(0, 0, 0, 0); o.lb = make_float4(0, 0, 0, 0.0f); o.n = make_float4(0, 0, 0, 0);
                if (k < nc) {
                  const int q = k - n0;  // (k >= n0: contact q of the second part)
                  const LocalContact c = k < n0 ? (k == 0 ? l0[0] : l0[1]) : (q == 0 ? l1[0] : l1[1]);
                  o.la = mk4(c.la, c.g.t); o.lb = mk4(c.lb, 0.0f); o.n = mk4(c.g.n, 0.0f);  // Manifold::from(lc) manifold.rs:120-128
                }
                if (k == 0) o.lb.w = u2f((uint32_t)nc);
                t_out[4 * (size_t)(tp + a) + k] = o;
              }
            }
          }
          run += (uint32_t)nc;
        }
      }
      run += (uint32_t)__shfl_xor((int)run, 1); run += (uint32_t)__shfl_xor((int)run, 2);
    } else if (GEN == 1) {
      if (nt) {
        const V3 vA = xyz(r1);
        const float4 c1 = A.col1[i];
        Comp Ca; Ca.p = xyz(r0); Ca.r = r0.w; Ca.d = xyz(c1); Ca.kind = (int)f2u(c1.w);
        const uint32_t* rt = rows_t + (size_t)i * cap_row_t;
        const bool bytes = pk0 != 0xFFFFFFFFu || pk1 != 0xFFFFFFFFu;
        for (uint32_t a = sub; a < nt; a += (uint32_t)kTcLanes) {  // the body's faces dealt to the group's lanes
          const uint32_t f = bytes ? ((a < 4u ? pk0 >> (8u * a) : pk1 >> (8u * (a - 4u))) & 255u) : rt[a];
          const uint4 fi = staged ? s_face[f] : M.faces[f];
          const Triangle tri = staged ? mkt(xyz(s_vert[fi.x]) + mx, xyz(s_vert[fi.y]) + mx, xyz(s_vert[fi.z]) + mx)
                                      : mkt(xyz(M.verts[fi.x]) + mx, xyz(M.verts[fi.y]) + mx, xyz(M.verts[fi.z]) + mx);  // mesh.rs:122-126
          LocalContact lc[2];
          const bool ff = comp_tri_far(Ca, vA, tri);
          const int nc = (ff && !A.check) ? 0 : comp_tri_local(Ca, vA, tri, mx, lc);
          if (A.check && ff && nc) *A.flag = 1u;  // (tests: the cheap reject dropped a face that reports a contact)
          if (tp + a < cap_t) {
            NContact o;
            o.la = make_float4(0, 0, 0, 0); o.lb = make_float4(0, 0, 0, u2f(0u)); o.n = make_float4(0, 0, 0, 0);
            if (nc > 0) { o.la = mk4(lc[0].la, lc[0].g.t); o.lb = mk4(lc[0].lb, u2f((uint32_t)nc)); o.n = mk4(lc[0].g.n, 0.0f); }  // Manifold::from(lc) manifold.rs:120-128
            t_out[2 * (size_t)(tp + a)] = o;
            if (nc > 1) { o.la = mk4(lc[1].la, lc[1].g.t); o.lb = mk4(lc[1].lb, 0.0f); o.n = mk4(lc[1].g.n, 0.0f); t_out[2 * (size_t)(tp + a) + 1] = o; }
          }
          run += (uint32_t)nc;
        }
      }
      // (the group's total: every lane of the wave is here)
      run += (uint32_t)__shfl_xor((int)run, 1); run += (uint32_t)__shfl_xor((int)run, 2);
    } else if (nt) {
      const V3 vA = xyz(r1);
      Comp Ca; Ca.p = xyz(r0); Ca.r = r0.w; Ca.d = mk3(0.0f, 0.0f, 0.0f); Ca.kind = KIND_SPHERE;
      const uint32_t* rt = rows_t + (size_t)i * cap_row_t;
      const bool bytes = pk0 != 0xFFFFFFFFu || pk1 != 0xFFFFFFFFu;  // (both all-ones: faces above 255 or more than eight - read the row)
      static_assert(kTcLanes == 4, "tri_msphere_x4: a face by four lanes");
      for (uint32_t a = 0; a < nt; ++a) {  // the body's faces one after the other, each by the group's four lanes together
        const uint32_t f = bytes ? ((a < 4u ? pk0 >> (8u * a) : pk1 >> (8u * (a - 4u))) & 255u) : rt[a];
        const uint4 fi = staged ? s_face[f] : M.faces[f];
        Triangle tri = staged ? mkt(xyz(s_vert[fi.x]) + mx, xyz(s_vert[fi.y]) + mx, xyz(s_vert[fi.z]) + mx)
                              : mkt(xyz(M.verts[fi.x]) + mx, xyz(M.verts[fi.y]) + mx, xyz(M.verts[fi.z]) + mx);  // mesh.rs:122-126
        // comp_tri_local for a sphere, the triangle test by the four lanes (tri_msphere_x4)
        Contact raw = mkc(mk3(0, 0, 0), mk3(0, 0, 0), mk3(0, 0, 0), 0.0f);
        const uint32_t nc = tri_msphere_x4(tri, mks(Ca.p, Ca.r), vA, &raw, (int)sub, lane & ~(kTcLanes - 1)) ? 1u : 0u;
        if (sub == 0u && tp + a < cap_t) {
          NContact o;
          o.la = make_float4(0, 0, 0, 0); o.lb = make_float4(0, 0, 0, u2f(0u)); o.n = make_float4(0, 0, 0, 0);
          if (nc) {  // Mesh::contacts callback value: a on the mesh, b on the body, n = face normal; Manifold::from(lc) manifold.rs:120-128
            const V3 a_c = comp_center(Ca) + vA * raw.t;
            const Contact g = neg(raw);
            o.la = mk4(raw.b + -a_c, g.t); o.lb = mk4(raw.a + -mx, u2f(1u)); o.n = mk4(g.n, 0.0f);
          }
          t_out[2 * (size_t)(tp + a)] = o;
        }
        run += nc;
      }
    }
    if (e < L && sub == 0u) {
      tcn[i] = run;  // the body's terrain constraints come first in its range (k_chain_rows)
      tpos[i] = tp;
    }
    uint32_t wrun = sub == 0u ? run : 0u;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) wrun += __shfl_xor(wrun, o);
    if (lane == 0 && wrun) atomicAdd(&sums[kCsSumStride], wrun);
  }
}
template <int GEN>
__global__ __launch_bounds__(kBlock) void k_terrain_contacts(TerrainContacts A) { terrain_contacts_job<GEN>(A, blockIdx.x, gridDim.x); }
// ... as the last `tc_blocks` blocks of the leaf scatter's launch (both follow k_integrate, neither needs the other: the few waves of the
// sphere-triangle tests - a resting sphere against the floor's other triangle runs three ray-capsule tests, ~15 us of one lane's
// arithmetic - hide behind the streaming kernel instead of taking a launch of their own)
template <int GEN>
__global__ __launch_bounds__(kBlock) void k_scatter_leaves_tc(Lbvh T, const float4* fb_c, const float4* fb_r, const uint32_t* cell_of,
                                                              const uint32_t* rank, uint32_t* brank, const float4* col0, const float4* delta, const float4* tb_c,
                                                              const float4* tb_r, const SceneBounds* sb, float pad_abs, float min_frac, TerrainContacts A, uint32_t tc_blocks,
                                                              const SceneBounds* box, float3 wide_limit, uint32_t n_owned) {
  if (blockIdx.x < tc_blocks) { terrain_contacts_job<GEN>(A, blockIdx.x, tc_blocks); return; }  // (first: they take longest)
  scatter_leaf((blockIdx.x - tc_blocks) * kBlock + threadIdx.x, T, fb_c, fb_r, cell_of, rank, brank, col0, delta, tb_c, tb_r, sb, pad_abs, min_frac, box, wide_limit, n_owned);
}
struct ContactsSpheres {
  const StepCounts* sc;
  uint32_t n, cap_row_t, cap_c, cap_t;
  const uint32_t *rows_p, *t_cnt, *p_cnt, *base, *tcn, *tpos;
  const NContact* t_out;
  float dt, baumgarte, slop;
  CRec* cons; uint2* ab; uint32_t* degb; RevEnt* rev; uint32_t rev_cap; uint32_t* rev_flag; uint32_t* flag;
  const uint32_t* ext;
  // (r06, k_terrain_near's slots) the terrain contacts' constraints by the first `t_blocks` workgroups of the launch, a lane per SLOT, instead of
  // by their bodies' threads one after the other (a block of bodies on the floor held the launch up: 256 threads with six slots each, the rest idle)
  uint32_t t_blocks, region_cap, regions, cnt_stride;
  const uint32_t *slot_body, *slot_cnt;
};
// a body's partner row -> the block's list in LDS, canonical order (entries of the window [w0, w0 + kCsEntCap))
__device__ __forceinline__ void cs_list_row(const uint32_t* rp, uint32_t np, const uint32_t* ext, uint32_t first, uint32_t w0, uint32_t owner, uint32_t* s_j, uint16_t* s_b) {
  if (np <= 12u) {  // (a sphere touches at most 12 equal ones)
    uint32_t pj[12], o_id[12];
#pragma unroll
    for (int a = 0; a < 12; a += 4) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if ((uint32_t)a < np) v = *reinterpret_cast<const uint4*>(rp + a);
      pj[a] = v.x; pj[a + 1] = v.y; pj[a + 2] = v.z; pj[a + 3] = v.w;
    }
#pragma unroll
    for (int a = 0; a < 12; ++a) o_id[a] = (uint32_t)a < np ? order_id(ext, pj[a]) : 0xFFFFFFFFu;
#pragma unroll
    for (int a = 0; a < 12; ++a) {
      uint32_t before = 0;
#pragma unroll
      for (int q = 0; q < 12; ++q) before += o_id[q] < o_id[a] ? 1u : 0u;
      const uint32_t pos = first + before - w0;
      if ((uint32_t)a < np && pos < kCsEntCap) { s_j[pos] = pj[a]; s_b[pos] = (uint16_t)owner; }
    }
  } else {
    for (uint32_t a = 0; a < np; ++a) {
      const uint32_t j = rp[a], oa = order_id(ext, j);
      uint32_t before = 0;
      for (uint32_t q = 0; q < np; ++q) before += order_id(ext, rp[q]) < oa ? 1u : 0u;
      const uint32_t pos = first + before - w0;
      if (pos < kCsEntCap) { s_j[pos] = j; s_b[pos] = (uint16_t)owner; }
    }
  }
}
// ... for the windows behind the first (a block with more than kCsEntCap partner contacts: bodies pressed into each other - a collapsing
// pile): the PLAIN loops, out of line.  (r05: with the twelve-entry network inlined a second time inside the window loop the kernel faulted
// on a null-based address the first time a tile of the collapsing million-sphere pile reached a second window - tools/soak_tiles.py, tick 156 -
// while small worlds in the same state passed; never root-caused.  r05 moved the second copy out of line; r06 takes the network
// out of the later windows altogether - the plain loops were measured sound there in round 5 - and sizes the first window so that a pile at
// rest does not need a second one: the path the fault was on is no longer the common one.  tests/test_gpu_contacts_dense.py, the tile soaks.)
__device__ __attribute__((noinline)) void cs_list_row_again(const uint32_t* rp, uint32_t np, const uint32_t* ext, uint32_t first, uint32_t w0, uint32_t owner,
                                                            uint32_t* s_j, uint16_t* s_b) {
  for (uint32_t a = 0; a < np; ++a) {
    const uint32_t j = rp[a], oa = order_id(ext, j);
    uint32_t before = 0;
    for (uint32_t q = 0; q < np; ++q) before += order_id(ext, rp[q]) < oa ? 1u : 0u;
    const uint32_t pos = first + before - w0;
    if (pos < kCsEntCap) { s_j[pos] = j; s_b[pos] = (uint16_t)owner; }
  }
}
// the collider of a body of the rows from its packed copy: SPH = a world of spheres only (word 1 is not read)
template <bool SPH>
__device__ __forceinline__ Comp pack_comp(const Bodies& B, uint32_t i, const BodyPack& P) {
  Comp X; X.p = xyz(P.c0); X.r = P.c0.w; X.d = mk3(0.0f, 0.0f, 0.0f); X.kind = KIND_SPHERE;
  if (!SPH) { const float4 c1 = B.bpk ? B.bpk[4 * (size_t)i + 3] : B.col1[i]; X.d = xyz(c1); X.kind = (int)f2u(c1.w); }
  return X;
}
// SPH = false (r06): the rows of a world of any single-component kinds (k_pair_grid_n lists contacts only, k_terrain_near / k_terrain_contacts<true>
// park the terrain contacts): the same kernel with the colliders' kinds read from the bodies.
template <bool SPH>
__global__ __launch_bounds__(kBlock) void k_contacts_rows(Bodies B, TerrainDev M, ContactsSpheres A) {
  __shared__ float4 s_w[kBlock / 64][7 * 65];  // a wave's records on their way out (see k_setup_pairs)
  __shared__ uint32_t s_c[kBlock / 64][64];
  __shared__ uint32_t s_j[kCsEntCap];           // the pass's partner contacts in canonical order: partner ...
  __shared__ uint16_t s_b[kCsEntCap];           // ... and owner (the block's body)
  __shared__ uint32_t s_cbase[kBlock];          // body -> id of its first partner constraint minus its first entry's position
  __shared__ uint32_t s_wave[kBlock / 64];
  if (A.sc->fail) return;  // (the scan's closing thread found a flag up or a capacity exceeded: the host re-runs the phase)
  if (blockIdx.x < A.t_blocks) {  // ---- ContactConstraint::new for the terrain contacts (world.rs:243-251), a lane per slot
    const uint32_t p = blockIdx.x * (uint32_t)kBlock + threadIdx.x, region = p / A.region_cap;
    if (region >= A.regions || p - region * A.region_cap >= min(A.slot_cnt[region * A.cnt_stride], A.region_cap)) return;
    const NContact in0 = A.t_out[2 * (size_t)p];
    const uint32_t nc = f2u(in0.lb.w);
    if (nc == 0) return;
    const uint32_t i = A.slot_body[p], tp = A.tpos[i];
    uint32_t c = A.base[i];
    for (uint32_t q = tp; q < p; ++q) c += f2u(A.t_out[2 * (size_t)q].lb.w);  // the contacts of the body's earlier faces (its slots are a run, in Mesh::contacts' order)
    const V3 mx = mk3(M.x[0], M.x[1], M.x[2]);
    const BodyDyn Ad = load_dyn(B.srec, i), S = static_dyn();
    const BodyPack Pa = load_pack(B, i, false);
    for (uint32_t k = 0; k < nc; ++k, ++c) {
      const NContact in = k == 0 ? in0 : A.t_out[2 * (size_t)p + k];
      // Static{ center: terrain.center(), friction: 0.0 } world.rs:247
      const CRec r = make_constraint(i, kNone, Ad, xyz(Pa.ei), Pa.ei.w, Pa.dl.w, S, mx, 0.0f, 0.0f, xyz(in.n), xyz(in.la), xyz(in.lb), A.dt, A.baumgarte, A.slop);
      store_crec(&A.cons[c], r);
      A.ab[c] = make_uint2(i, kNone);
    }
    return;
  }
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const uint32_t i0 = (blockIdx.x - A.t_blocks) * (uint32_t)kBlock, i = i0 + (uint32_t)t;
  uint32_t np = 0, run = 0, base_i = 0;
  if (i < A.n) { np = A.p_cnt[i]; run = A.tcn[i]; base_i = A.base[i]; }
  // ---- where the body's partner contacts start in the block's list
  uint32_t inc_p = np;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(inc_p, o); if (lane >= o) inc_p += v; }
  if (lane == 63) s_wave[wv] = inc_p;
  __syncthreads();
  uint32_t before_p = 0, totalp = 0;
  for (int k = 0; k < kBlock / 64; ++k) { const uint32_t v = s_wave[k]; if (k < wv) before_p += v; totalp += v; }
  const uint32_t excl_p = before_p + inc_p - np;
  const uint32_t* rp = A.rows_p + (size_t)i * kRowCap;
  if (np) cs_list_row(rp, np, A.ext, excl_p, 0u, (uint32_t)t, s_j, s_b);
  s_cbase[t] = base_i + run - excl_p;
  // ---- ContactConstraint::new for the terrain contacts (world.rs:243-251): the body's own thread (setup_terrain_one's work)
  if (run && A.t_blocks == 0u) {
    const uint32_t nt = A.t_cnt[i], tp = A.tpos[i];
    const V3 mx = mk3(M.x[0], M.x[1], M.x[2]);
    const BodyDyn Ad = load_dyn(B.srec, i), S = static_dyn();
    const BodyPack Pa = load_pack(B, i, false);
    uint32_t c = base_i;
    for (uint32_t a = 0; a < nt; ++a) {
      const NContact in0 = A.t_out[2 * (size_t)(tp + a)];
      const uint32_t nc = f2u(in0.lb.w);
      for (uint32_t k = 0; k < nc; ++k, ++c) {
        const NContact in = k == 0 ? in0 : A.t_out[2 * (size_t)(tp + a) + k];
        // Static{ center: terrain.center(), friction: 0.0 } world.rs:247
        const CRec r = make_constraint(i, kNone, Ad, xyz(Pa.ei), Pa.ei.w, Pa.dl.w, S, mx, 0.0f, 0.0f, xyz(in.n), xyz(in.la), xyz(in.lb), A.dt, A.baumgarte, A.slop);
        store_crec(&A.cons[c], r);
        A.ab[c] = make_uint2(i, kNone);
      }
    }
  }
  // ---- ... and for the partner contacts: the block's list, kCsEntCap entries per pass, an entry per thread
  float4* out = reinterpret_cast<float4*>(A.cons);
  __syncthreads();  // (the list's first pass, s_cbase)
  for (uint32_t w0 = 0; w0 < totalp; w0 += kCsEntCap) {
    if (w0) {  // (rare: a block with more contacts than a pass holds lists the next window)
      __syncthreads();
      if (np) cs_list_row_again(rp, np, A.ext, excl_p, w0, (uint32_t)t, s_j, s_b);
      __syncthreads();
    }
    const uint32_t m = min(totalp - w0, kCsEntCap);
    for (uint32_t e0 = 0; e0 < m; e0 += (uint32_t)kBlock) {  // (the same trips for every wave of the block)
      const uint32_t e = e0 + (uint32_t)t;
      uint32_t c = kNone;
      if (e < m) {
        const uint32_t b = s_b[e], j = s_j[e], ia = i0 + b;
        const BodyPack Pa = load_pack(B, ia, true), Pb = load_pack(B, j, true);
        const Comp Xa = pack_comp<SPH>(B, ia, Pa), Xb = pack_comp<SPH>(B, j, Pb);  // as k_narrow_pairs<KA, KB>
        LocalContact lc;
        if (!comp_pair_local(Xa, xyz(Pa.dl), Xb, xyz(Pb.dl), &lc)) {
          *A.flag = 1u;  // the broadphase's pair test and this one disagree about a contact
        } else {
          const V3 nrm = (mk3(0.0f, 0.0f, 0.0f) + lc.g.n) / 1.0f;  // Manifold::from(pruner) of one contact (manifold.rs:135-140)
          const BodyDyn Ad = load_dyn(B.srec, ia), Bd = load_dyn(B.srec, j);
          c = s_cbase[b] + w0 + e;
          const CRec r = make_constraint(ia, j, Ad, xyz(Pa.ei), Pa.ei.w, Pa.dl.w, Bd, xyz(Pb.ei), Pb.ei.w, Pb.dl.w, nrm, lc.la, lc.lb, A.dt, A.baumgarte, A.slop);
          const float4* rw = reinterpret_cast<const float4*>(&r);
#pragma unroll
          for (int k = 0; k < 7; ++k) s_w[wv][k * 65 + lane] = rw[k];
          A.ab[c] = make_uint2(ia, j);
          // body j's row of the constraints it takes part in as `b` (k_chain_rows); as `a` a body owns a contiguous id range
          const uint32_t pos = atomicAdd(&A.degb[j], 1u);
          if (pos < A.rev_cap) A.rev[(size_t)j * A.rev_cap + pos] = RevEnt{c, ia, order_id(A.ext, ia), 0u};
          else *A.rev_flag = 1u;
        }
      }
      s_c[wv][lane] = c;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // (a wave's LDS accesses are served in order; the wave reads only what it wrote)
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int rec = it * 8 + (lane >> 3), k = lane & 7;
        const uint32_t cr = s_c[wv][rec];
        if (cr != kNone && k < 7) out[(size_t)cr * 8 + k] = s_w[wv][k * 65 + rec];
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
  }
}

// ContactConstraint::new (solver.rs:101-191) for caller-built manifolds on the resident RigidBodyVec (mgf_constraints_new):
// one thread per (manifold, contact) row.  obj_a is Dynamic; obj_b Dynamic or Static{center, friction} (b == kNone).
struct ManifoldRow { uint32_t a, b; float cb[3], fric_b; float n[3], t0[3], t1[3], la[3], lb[3]; };
__global__ __launch_bounds__(kBlock) void k_constraints_new(Bodies B, const ManifoldRow* rows, uint32_t m, float dt, float baumgarte, float slop,
                                                            CRec* out) {
  uint32_t r = blockIdx.x * kBlock + threadIdx.x;
  if (r >= m) return;
  const ManifoldRow R = rows[r];
  BodyDyn A = load_dyn(B.srec, R.a);
  float4 ea = B.einfo[R.a];
  BodyDyn Bd = static_dyn();
  V3 xb = ld3(R.cb);
  float rest_b = 0.0f, fric_b = R.fric_b;  // physics.rs:289-302
  if (R.b != kNone) {
    Bd = load_dyn(B.srec, R.b);
    float4 eb = B.einfo[R.b];
    xb = xyz(eb); rest_b = eb.w; fric_b = B.delta[R.b].w;
  }
  CRec c = make_constraint_basis(R.a, R.b, A, xyz(ea), ea.w, B.delta[R.a].w, Bd, xb, rest_b, fric_b, ld3(R.n), ld3(R.t0), ld3(R.t1),
                                 ld3(R.la), ld3(R.lb), dt, baumgarte, slop);
  store_crec(&out[r], c);
}

}  // namespace mgf
