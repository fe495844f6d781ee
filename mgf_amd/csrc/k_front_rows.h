// The list-free front end for worlds that are NOT spheres only (r06): capsules, mixed worlds of single-component bodies.
// (Part of the kernel set described in kernels.h.)
//
// What the spheres got in round 5 - rows that hold CONTACTS, a scan of the bodies' constraint counts, records written straight from the
// rows (k_contacts_rows) - for every single-component kind.  The tick of such a world used to build candidate lists in CSR form
// (k_scan<2>, k_rows_to_csr), run one narrowphase launch per shape-pair type over them, count (k_count_contacts), scan and set up
// (k_setup_pairs): on BASELINE config 3 (131 072 capsules over a 49 928-triangle heightfield) 165 us of list plumbing that nothing
// downstream reads.  Here:
//   k_near_list        the bodies whose tight box can reach a face of the static mesh at all (Mesh::contacts returns at the root of its
//                      tree for the others, bvh.rs:283-297): a compact list - one body in sixteen of a pile lies on the floor;
//   k_terrain_near<L>  L lanes per listed body: the face grid's cells (k_terrain_grid's search, the work dealt by FACE RECORD, not by
//                      cell), the hits sorted into the reference's DFS order in LDS, then a lane per hit through the body-triangle test
//                      (collision.rs:610-1086); the contacts are parked in the terrain list's slots exactly as k_terrain_contacts parks
//                      the spheres' (tcn / tpos / t_cnt per body);
//   k_pair_grid_n      k_pair_grid's search; the accepted partners pass the conservative bounding-sphere test by the query's own eight
//                      lanes, the survivors of the block's 64 queries are POOLED and go through the pair test (collision.rs:1089-1356)
//                      a lane each - the branchy capsule code runs in one or two full waves per block instead of in every wave with a
//                      lane or two active; the rows hold contacts only;
//   k_scan<1> (+ tcn)  base = exclusive prefix of p_cnt + tcn, the tick's counts in its epilogue (caps_contacts);
//   k_contacts_rows<false>  (k_contacts.h) ContactConstraint::new from the rows, the pair test evaluated once more for the contacts.
// Results are those of the list-based kernels bit for bit: the same device functions on the same inputs, the same insertion order
// (terrain contacts in Mesh::contacts' order, then partners by ascending order id).
#pragma once
#include "k_contacts.h"

namespace mgf {

// ---- bodies near the static mesh -------------------------------------------------------------------------------------------------
// (the test k_terrain_grid opens with: a query that ends before the first face box or starts behind the last one on any axis)
__global__ __launch_bounds__(kBlock) void k_near_list(const float4* tb_c, const float4* tb_r, uint32_t n_owned, TerrainDev M, const SceneBounds* gsb,
                                                      float pad_abs, uint32_t* near_ids, uint32_t* near_cnt, const uint32_t* guard) {
  __shared__ uint32_t s_n, s_base;
  if (*guard) return;
  if (threadIdx.x == 0) s_n = 0;
  __syncthreads();
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  bool near = false;
  if (i < n_owned) {
    const V3 qc = xyz(tb_c[i]) + -mk3(M.x[0], M.x[1], M.x[2]), qr = xyz(tb_r[i]);
    const float pad = pad_abs + 1e-5f * (fabs_rs(qc.x) + fabs_rs(qc.y) + fabs_rs(qc.z) + qr.x + qr.y + qr.z);
    near = true;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float lo = ord_f(gsb->lo[k]), hi = ord_f(gsb->hi[k]), rm = ord_f(gsb->rmax[k]);
      const float a = at(qc, k) - at(qr, k) - rm - pad, b = at(qc, k) + at(qr, k) + rm + pad;
      near = near && !(b < lo || a > hi);
    }
  }
  const unsigned long long mask = __ballot(near);
  const int lane = threadIdx.x & 63;
  uint32_t wbase = 0;
  if (lane == 0 && mask) wbase = atomicAdd(&s_n, (uint32_t)__popcll(mask));
  wbase = __shfl(wbase, 0);
  __syncthreads();
  if (threadIdx.x == 0 && s_n) s_base = atomicAdd(near_cnt, s_n);
  __syncthreads();
  if (near) near_ids[s_base + wbase + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = i;
}

// ---- terrain search + body-triangle narrowphase of the listed bodies ----------------------------------------------------------------
constexpr uint32_t kTnHitCap = 128;  // faces a body's query may accept (more: the flag goes up and the host takes the list-based kernels)
// The slots of the tick are handed out from kTnRegions counters, a block uses the one of its XCD (blockIdx & 7): one counter for all
// cost the launch 13 ns per workgroup (returning atomics on one word serialise) - 2 500 of them - however short the blocks' own work.
constexpr uint32_t kTnRegions = 8, kTnCntStride = 32;  // (words between counters: a cache line)
constexpr uint32_t kTnFarBit = 0x40000000u;  // a hit that comp_tri_far rejects (DFS ranks are below 2^30)
struct TerrainNear {
  TerrainDev M; FaceGrid G;
  const uint32_t* near_ids; const uint32_t* near_cnt;
  const uint32_t* face_of_rank;
  float pad_abs; uint32_t cap_t;
  uint32_t* cnt;       // kTnRegions slot counters and as many partial counts of the accepted faces (World::step's terrain candidates), a cache line each
  uint32_t region_cap; // slots per region
  uint32_t *slot_body, *slot_rank;  // per slot: the body and the face's DFS rank (k_terrain_tests)
  uint32_t *tpos, *t_cnt;
  uint32_t *overflow, *too_wide;
  const uint32_t* guard;
  uint32_t check;      // tests: 1 = the rejected hits get slots too, marked kTnFarBit: k_terrain_tests raises a flag if one of them reports a contact
};
template <int L>
__global__ __launch_bounds__(kBlock) void k_terrain_near(Bodies B, TerrainNear A) {
  static_assert(L == 16 || L == 32 || L == 64, "a group is a power-of-two part of a wave");
  constexpr int NG = kBlock / L;
  constexpr int kPer = (int)kTnHitCap / L;  // hits per lane when a group's list is full
  __shared__ uint32_t s_hits[NG][kTnHitCap], s_surv[NG][kTnHitCap];
  __shared__ uint32_t s_p0[NG][L], s_pre[NG][L + 1];
  __shared__ uint32_t s_cnt[NG], s_scnt[NG], s_tp[NG];
  const uint32_t Ltot = *A.guard ? 0u : *A.near_cnt;
  const int t = threadIdx.x, g = t / L, s = t % L;
  const V3 mx = mk3(A.M.x[0], A.M.x[1], A.M.x[2]);
  const uint32_t nb[3] = {A.G.bits.x, A.G.bits.y, A.G.bits.z};
  // (blockIdx & 7 - the block's XCD - picks its region: the stride keeps a block in its region over its trips)
  for (uint32_t e0 = blockIdx.x * (uint32_t)NG; e0 < Ltot; e0 += gridDim.x * (uint32_t)NG) {  // (the same trips for every thread of the block)
    if (t < NG) { s_cnt[t] = 0u; s_scnt[t] = 0u; }
    __syncthreads();
    const uint32_t e = e0 + (uint32_t)g;
    const bool live = e < Ltot;
    const uint32_t i = live ? A.near_ids[e] : 0u;
    V3 vA = mk3(0, 0, 0);
    Comp Ac; Ac.kind = KIND_SPHERE; Ac.p = mk3(0, 0, 0); Ac.d = mk3(0, 0, 0); Ac.r = 0.0f;
    if (live) {
      Ac = load_comp_moving(B, i, &vA);
      // ---- Mesh::contacts' BVH::query by cell enumeration (k_terrain_grid), a lane per face record
      Box q; q.c = xyz(B.tb_c[i]) + -mx; q.r = xyz(B.tb_r[i]);
      const float mag_q = fabs_rs(q.c.x) + fabs_rs(q.c.y) + fabs_rs(q.c.z) + q.r.x + q.r.y + q.r.z;
      const float pad = A.pad_abs + 1e-5f * mag_q;
      uint32_t ca[3], d[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float lo = ord_f(A.G.sb->lo[k]), hi = ord_f(A.G.sb->hi[k]), rm = ord_f(A.G.sb->rmax[k]);
        const float a = at(q.c, k) - at(q.r, k) - rm - pad, b = at(q.c, k) + at(q.r, k) + rm + pad;
        const uint32_t c0 = morton_quant(a, lo, hi) >> (10u - nb[k]), c1 = morton_quant(b, lo, hi) >> (10u - nb[k]);
        ca[k] = c0; d[k] = c1 - c0 + 1u;
      }
      const uint32_t ncell = d[0] * d[1] * d[2];
      if (ncell > kGridMaxCells) {
        if (s == 0) *A.too_wide = 1u;
      } else {
        for (uint32_t cb = 0; cb < ncell; cb += (uint32_t)L) {  // (the group's lanes move together)
          const uint32_t idx = cb + (uint32_t)s;
          uint32_t p0 = 0, cnt = 0;
          if (idx < ncell) {
            const uint32_t cz = idx % d[2], tt = idx / d[2];
            const uint32_t cy = tt % d[1], cx = tt / d[1];
            const uint32_t cell = face_cell(ca[0] + cx, ca[1] + cy, ca[2] + cz, A.G.bits);
            p0 = A.G.T.cell_lo[cell]; cnt = A.G.T.cell_lo[cell + 1] - p0;
          }
          uint32_t inc = cnt;
#pragma unroll
          for (int o = 1; o < L; o <<= 1) { const uint32_t u = __shfl_up(inc, o, L); if (s >= o) inc += u; }
          const uint32_t total = __shfl(inc, L - 1, L);
          s_p0[g][s] = p0; s_pre[g][s + 1] = inc;
          if (s == 0) s_pre[g][0] = 0u;
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // (a wave's LDS accesses are served in order; the group reads what it wrote)
          for (uint32_t m = (uint32_t)s; m < total; m += (uint32_t)L) {
            int c = 0;  // the cell of record m: the last c with s_pre[c] <= m
#pragma unroll
            for (int step = L / 2; step >= 1; step >>= 1) if (c + step < L && s_pre[g][c + step] <= m) c += step;
            const uint32_t p = s_p0[g][c] + (m - s_pre[g][c]);
            const LeafRec lr = A.G.T.leaves[p];
            const uint32_t face = f2u(lr.c.w);
            Box fb; fb.c = xyz(lr.c); fb.r = xyz(lr.r);
            if (!box_overlaps(q, fb)) continue;  // the reference's acceptance test at the leaf (bvh.rs:297)
            bool hit = true;
            // a hit within rounding distance of not overlapping: the reference only reaches a leaf through its ancestors (see k_terrain_grid)
            const float gap = fmin_rs(fmin_rs(q.r.x + fb.r.x - fabs_rs(q.c.x - fb.c.x), q.r.y + fb.r.y - fabs_rs(q.c.y - fb.c.y)),
                                      q.r.z + fb.r.z - fabs_rs(q.c.z - fb.c.z));
            const float tol = 1e-4f * (mag_q + fabs_rs(fb.c.x) + fabs_rs(fb.c.y) + fabs_rs(fb.c.z) + fb.r.x + fb.r.y + fb.r.z);
            if (!(gap > tol)) {
              uint32_t node = A.G.leaf_of_face[face];
              while (node != A.M.root) {
                node = A.G.parent[node];
                const float4* raw = reinterpret_cast<const float4*>(&A.M.nodes[node]);
                Box nbx; nbx.c = xyz(raw[0]); nbx.r = xyz(raw[1]);
                if (!box_overlaps(q, nbx)) { hit = false; break; }
                const float ga = fmin_rs(fmin_rs(q.r.x + nbx.r.x - fabs_rs(q.c.x - nbx.c.x), q.r.y + nbx.r.y - fabs_rs(q.c.y - nbx.c.y)),
                                         q.r.z + nbx.r.z - fabs_rs(q.c.z - nbx.c.z));
                const float ta = 1e-4f * (mag_q + fabs_rs(nbx.c.x) + fabs_rs(nbx.c.y) + fabs_rs(nbx.c.z) + nbx.r.x + nbx.r.y + nbx.r.z);
                if (ga > ta) break;
              }
            }
            if (hit) {
              const uint32_t pos = atomicAdd(&s_cnt[g], 1u);
              if (pos < kTnHitCap) s_hits[g][pos] = A.G.rank_of_face[face];
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // (the next chunk overwrites the tables)
        }
      }
    }
    __syncthreads();
    // ---- the cheap reject on every hit (comp_tri_far), by the group's own lanes
    uint32_t H = s_cnt[g];
    if (H > kTnHitCap) { if (s == 0) atomicOr(A.overflow, 2u); H = 0u; }
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
      const uint32_t a = (uint32_t)(u * L + s);
      if (a < H) {
        const uint32_t rank = s_hits[g][a];
        const uint4 fi = A.M.faces[A.face_of_rank[rank]];
        const Triangle tri = mkt(xyz(A.M.verts[fi.x]) + mx, xyz(A.M.verts[fi.y]) + mx, xyz(A.M.verts[fi.z]) + mx);  // mesh.rs:122-126
        if (comp_tri_far(Ac, vA, tri)) s_hits[g][a] = rank | kTnFarBit;
      }
    }
    __syncthreads();
    // ---- the survivors into the reference's callback order (DFS ranks are distinct: a hit's place = the smaller ranks among them)
    const uint32_t keep_mask = A.check ? 0u : kTnFarBit;  // (check: the rejected hits stay in the list, marked)
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
      const uint32_t a = (uint32_t)(u * L + s);
      if (a < H) {
        const uint32_t w = s_hits[g][a];
        if (!(w & keep_mask)) {
          const uint32_t r = w & ~kTnFarBit;
          uint32_t before = 0;
          for (uint32_t b = 0; b < H; ++b) { const uint32_t o = s_hits[g][b]; before += (!(o & keep_mask) && (o & ~kTnFarBit) < r) ? 1u : 0u; }
          s_surv[g][before] = w;
          atomicAdd(&s_scnt[g], 1u);
        }
      }
    }
    __syncthreads();
    if (t == 0) {  // the block's slots with ONE atomic (and the accepted faces counted with another)
      uint32_t total = 0, cand = 0;
      for (int k = 0; k < NG; ++k) { s_tp[k] = total; total += s_scnt[k]; cand += s_cnt[k] > kTnHitCap ? 0u : s_cnt[k]; }
      const uint32_t region = blockIdx.x & (kTnRegions - 1u);
      uint32_t base = total ? atomicAdd(&A.cnt[region * kTnCntStride], total) : 0u;
      if (base + total > A.region_cap) { atomicOr(A.overflow, 4u); base = A.region_cap; }  // (the tick is run again with more room: nothing of this block is written)
      if (cand) atomicAdd(&A.cnt[(kTnRegions + region) * kTnCntStride], cand);
      for (int k = 0; k < NG; ++k) s_tp[k] += region * A.region_cap + base;
    }
    __syncthreads();
    // ---- every survivor has its slot - in its body's run, in the reference's order - whether it will report a contact or not
    const uint32_t S = s_scnt[g], tp = s_tp[g];
    const uint32_t region_end = ((blockIdx.x & (kTnRegions - 1u)) + 1u) * A.region_cap;
    for (uint32_t a = (uint32_t)s; a < S; a += (uint32_t)L) {
      if (tp + a < region_end) { A.slot_body[tp + a] = i; A.slot_rank[tp + a] = s_surv[g][a]; }
    }
    if (live && s == 0) { A.tpos[i] = tp; A.t_cnt[i] = S; }  // (tcn: k_terrain_tests adds the contacts up; the tick's clearing launch zeroed it)
  }
}
// ... and the body-triangle test (collision.rs:610-1086), a lane per slot: whole waves of it, as many as there are slots - nothing else
// in the launch, so that it is as short as ONE walk through tri_mcapsule's branches.
struct TerrainTests {
  TerrainDev M;
  const uint32_t* face_of_rank;
  const uint32_t *slot_body, *slot_rank, *cnt;  // cnt: the regions' slot counters (k_terrain_near)
  uint32_t region_cap;
  NContact* t_out;     // 2 per slot; .lb.w of the first = the face's contact count
  uint32_t* tcn;       // += the body's terrain contacts
  uint32_t* sum_ct;    // += the tick's
  uint32_t* flag;      // check mode: a slot marked kTnFarBit reported a contact
  const uint32_t* overflow;  // bit 2: a region ran full - slots were reserved and not written; the host runs the tick again with more room
};
__global__ __launch_bounds__(kBlock) void k_terrain_tests(Bodies B, TerrainTests A) {
  __shared__ uint32_t s_con;
  if (*A.overflow & 4u) return;
  // (a region's slots are its first ones: the blocks behind them leave at once)
  const uint32_t p = blockIdx.x * (uint32_t)kBlock + threadIdx.x;
  const uint32_t r_first = (blockIdx.x * (uint32_t)kBlock) / A.region_cap, r_last = min((blockIdx.x * (uint32_t)kBlock + (uint32_t)kBlock - 1u) / A.region_cap, kTnRegions - 1u);
  bool any = false;
  for (uint32_t r = r_first; r <= r_last && r < kTnRegions; ++r) {
    const uint32_t lo = max(r * A.region_cap, blockIdx.x * (uint32_t)kBlock);
    any = any || lo - r * A.region_cap < min(A.cnt[r * kTnCntStride], A.region_cap);
  }
  if (!any) return;
  if (threadIdx.x == 0) s_con = 0u;
  __syncthreads();
  const uint32_t region = p / A.region_cap;
  int nc = 0;
  if (region < kTnRegions && p - region * A.region_cap < min(A.cnt[region * kTnCntStride], A.region_cap)) {
    const uint32_t i = A.slot_body[p], w = A.slot_rank[p];
    const V3 mx = mk3(A.M.x[0], A.M.x[1], A.M.x[2]);
    V3 vA;
    const Comp Ac = load_comp_moving(B, i, &vA);
    const uint4 fi = A.M.faces[A.face_of_rank[w & ~kTnFarBit]];
    const Triangle tri = mkt(xyz(A.M.verts[fi.x]) + mx, xyz(A.M.verts[fi.y]) + mx, xyz(A.M.verts[fi.z]) + mx);  // mesh.rs:122-126
    LocalContact lc[2];
    nc = comp_tri_local(Ac, vA, tri, mx, lc);
    if ((w & kTnFarBit) && nc) *A.flag = 1u;  // (check mode only: the reject was wrong)
    NContact o;
    o.la = make_float4(0, 0, 0, 0); o.lb = make_float4(0, 0, 0, u2f(0u)); o.n = make_float4(0, 0, 0, 0);
    if (nc > 0) { o.la = mk4(lc[0].la, lc[0].g.t); o.lb = mk4(lc[0].lb, u2f((uint32_t)nc)); o.n = mk4(lc[0].g.n, 0.0f); }  // Manifold::from(lc) manifold.rs:120-128
    A.t_out[2 * (size_t)p] = o;
    if (nc > 1) { o.la = mk4(lc[1].la, lc[1].g.t); o.lb = mk4(lc[1].lb, 0.0f); o.n = mk4(lc[1].g.n, 0.0f); A.t_out[2 * (size_t)p + 1] = o; }
    if (nc) atomicAdd(&A.tcn[i], (uint32_t)nc);  // the body's terrain constraints come first in its range (k_chain_rows)
  }
  uint32_t v = (uint32_t)nc;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  if ((threadIdx.x & 63) == 0 && v) atomicAdd(&s_con, v);
  __syncthreads();
  if (threadIdx.x == 0 && s_con) atomicAdd(A.sum_ct, s_con);
}

// ---- pair search with the narrowphase of any single-component pair in it ---------------------------------------------------------------
// (pair_query_cells<false>'s first phase with the accepted partners staged in LDS)
template <class Src>
__device__ __forceinline__ uint32_t pair_query_accept(const Src& S, const Box& q, uint32_t oi, uint32_t n_owned, const uint32_t* ca, const uint32_t* d,
                                                      const uint32_t* nb, int shift, int sub, int gbase, uint32_t* acc) {
  const uint32_t ncell = d[0] * d[1] * d[2];
  const uint32_t m2 = ((1u << 20) + d[2] - 1u) / d[2], m1 = ((1u << 20) + d[1] - 1u) / d[1];
  uint32_t np = 0;
  for (uint32_t cb = 0; cb < ncell; cb += kCoopLanes) {
    const uint32_t idx = cb + (uint32_t)sub;
    uint32_t p0 = 0, p1 = 0;
    if (idx < ncell) {
      const uint32_t t = (idx * m2) >> 20, cz = idx - t * d[2];
      const uint32_t cx = (t * m1) >> 20, cy = t - cx * d[1];
      const uint32_t cc3[3] = {ca[0] + cx, ca[1] + cy, ca[2] + cz};
      const uint32_t code = (expand10(cc3[0] << (10u - nb[0])) << 2) | (expand10(cc3[1] << (10u - nb[1])) << 1) | expand10(cc3[2] << (10u - nb[2]));
      S.range(code >> shift, cc3, p0, p1);
    }
    for (;;) {
      const bool more = p0 < p1;
      const unsigned long long mb = __ballot(more);
      if (((uint32_t)(mb >> gbase) & 255u) == 0u) break;
      bool hit = false;
      uint32_t j = 0;
      if (more) {
        float4 lc, lr;
        S.leaf(p0, lc, lr);
        j = f2u(lc.w);
        if (f2u(lr.w) < oi && j < n_owned) {  // world.rs:266; ghost-ghost skipped
          Box fb; fb.c = xyz(lc); fb.r = xyz(lr);
          hit = box_overlaps(q, fb);  // the reference's own acceptance test (bvh.rs:297)
        }
        ++p0;
      }
      const uint32_t gm = (uint32_t)(__ballot(hit) >> gbase) & 255u;
      if (hit) {
        const uint32_t slot = np + __popc(gm & ((1u << sub) - 1u));
        if (slot < (uint32_t)kRowCap) acc[slot] = j;
      }
      np += __popc(gm);
    }
  }
  return np;
}
constexpr uint32_t kPnQueries = kCoopBlock / kCoopLanes;  // 64 queries per block

// ---- bodies of up to two components (BASELINE config 5; not in the reference: DESIGN.md) ---------------------------------------------
// The manifold of a pair of bodies, all in registers: k_narrow_pairs_parts' three steps for ONE candidate - Contacts (compound.rs:180-190)
// per part pair in order (parts of i outer), local points relative to the bodies' centres (LocalContacts, compound.rs:192-207),
// ContactPruner::push (manifold.rs:72-102), Manifold::from(pruner) (:131-148) - the same operations on the same values.  An ordinary
// body is a body of one part (its collider).  Returns the contacts kept (0..4); la / lb = their local points, *normal = the manifold's.
// The manifold of a pair of bodies in k_narrow_pairs_parts' steps, the raw contacts of the part pairs staged in LDS (a lane per part pair,
// then a lane per pair of bodies for the pruner): one call site of the pair test, no per-lane arrays (a lane that ran the four part pairs by
// itself kept them in scratch memory, and a block's few hundred pairs filled a quarter of its lanes).
constexpr uint32_t kPpRound = 128;  // pairs of bodies per round of a block (4 raw slots of 48 bytes each: 24 KB)
// one part pair (slot = 2 a + b, parts of i outer) -> its raw slot: a.xyz, t | b.xyz, hit | n.xyz
__device__ __forceinline__ void parts_item(const Bodies& B, uint32_t i, uint32_t j, uint32_t slot, NContact* out) {
  NContact raw;
  raw.la = raw.lb = raw.n = make_float4(0, 0, 0, 0);
  const uint32_t a = slot >> 1, b = slot & 1u;
  const uint32_t pci = B.pcount ? B.pcount[i] : 0u, pcj = B.pcount ? B.pcount[j] : 0u;
  if ((pci == 0u ? a == 0u : a < pci) && (pcj == 0u ? b == 0u : b < pcj)) {
    float4 a0, a1, b0, b1;
    if (pci == 0u) { a0 = B.col0[i]; a1 = B.col1[i]; } else { a0 = B.wp0[kMaxParts * (size_t)i + a]; a1 = B.wp1[kMaxParts * (size_t)i + a]; }
    if (pcj == 0u) { b0 = B.col0[j]; b1 = B.col1[j]; } else { b0 = B.wp0[kMaxParts * (size_t)j + b]; b1 = B.wp1[kMaxParts * (size_t)j + b]; }
    Comp Pa, Pb;
    Pa.kind = (int)f2u(a1.w); Pa.p = xyz(a0); Pa.r = a0.w; Pa.d = xyz(a1);
    Pb.kind = (int)f2u(b1.w); Pb.p = xyz(b0); Pb.r = b0.w; Pb.d = xyz(b1);
    const V3 vA = xyz(B.delta[i]), vB = xyz(B.delta[j]);
    Contact c;
    if (!comp_pair_far(Pa, vA, Pb, vB) && comp_pair_contact(Pa, vA, Pb, vB, &c)) { raw.la = mk4(c.a, c.t); raw.lb = mk4(c.b, 1.0f); raw.n = mk4(c.n, 0.0f); }
  }
  *out = raw;
}
// ContactPruner::push (manifold.rs:72-102) over the four raw slots of a pair, in order: the contacts kept (their slots, two bits each, in
// *keep), the earliest time, the manifold's normal (Manifold::from(pruner), :131-148).  `mine` is LDS: indexed at run time.
struct PairCentres { V3 ci, cj, vA, vB; };
__device__ __forceinline__ PairCentres pair_centres(const Bodies& B, uint32_t i, uint32_t j) {
  PairCentres P;
  const uint32_t pci = B.pcount ? B.pcount[i] : 0u, pcj = B.pcount ? B.pcount[j] : 0u;
  P.ci = pci ? xyz(B.col0[i]) : comp_center(load_comp(B, i)); P.cj = pcj ? xyz(B.col0[j]) : comp_center(load_comp(B, j));
  P.vA = xyz(B.delta[i]); P.vB = xyz(B.delta[j]);
  return P;
}
__device__ __forceinline__ LocalContact parts_local(const PairCentres& P, const NContact& raw) {
  Contact c; c.a = xyz(raw.la); c.b = xyz(raw.lb); c.n = xyz(raw.n); c.t = raw.la.w;
  LocalContact nc; nc.la = c.a + -(P.ci + P.vA * c.t); nc.lb = c.b + -(P.cj + P.vB * c.t); nc.g = c;
  return nc;
}
__device__ __forceinline__ int parts_prune(const PairCentres& P, const NContact* mine, uint32_t* keep_out, float* min_t_out, V3* normal) {
  float min_t = kInf;
  int cnt = 0;
  uint32_t keep = 0u;  // (keep >> 2k) & 3: the slot of the k-th kept contact
  for (uint32_t slot = 0; slot < 4u; ++slot) {
    const NContact raw = mine[slot];
    if (raw.lb.w == 0.0f) continue;
    const LocalContact nc = parts_local(P, raw);
    if (nc.g.t < min_t - kCollisionEps) { cnt = 1; keep = slot; min_t = nc.g.t; continue; }
    if (nc.g.t > min_t + kCollisionEps) continue;
    bool merged = false;
    for (int k = 0; k < cnt && !merged; ++k) {
      const LocalContact kc = parts_local(P, mine[(keep >> (2 * k)) & 3u]);
      const V3 ra = nc.g.a - kc.g.a, rb = nc.g.b - kc.g.b;
      if (mag2(ra) <= kPersistentThresholdSq || mag2(rb) <= kPersistentThresholdSq) {
        const float prev = mag2(kc.la) + mag2(kc.lb), cur = mag2(nc.la) + mag2(nc.lb);
        if (prev < cur) keep = (keep & ~(3u << (2 * k))) | (slot << (2 * k));
        merged = true;
      }
    }
    if (!merged) { keep = (keep & ~(3u << (2 * cnt))) | (slot << (2 * cnt)); ++cnt; }
  }
  if (cnt) {
    V3 sum = mk3(0.0f, 0.0f, 0.0f);
    for (int k = 0; k < cnt; ++k) sum = sum + xyz(mine[(keep >> (2 * k)) & 3u].n);
    *normal = sum / (float)cnt;
  }
  *keep_out = keep; *min_t_out = min_t;
  return cnt;
}

// PARTS: the world holds bodies of up to two components - the accepted partners whose tight boxes meet are pooled (k_narrow_pairs_parts'
// first step), the pool goes through pair_manifold2, a row entry is partner | contacts << 27 and p_cnt the body's CONTACTS (p_ent its entries).
template <bool PARTS>
__global__ __launch_bounds__(kCoopBlock) void k_pair_grid_n(Bodies B, uint32_t n, uint32_t n_owned, Lbvh T, const SceneBounds* sb, float pad_abs, uint32_t* rows_p,
                                                            uint32_t* p_cnt, uint32_t* overflow, uint32_t* too_wide, uint32_t* pair_stat, float min_frac, uint32_t* p_ent) {
  __shared__ uint32_t s_acc[kPnQueries][kRowCap];   // accepted partners of a query (body slots)
  __shared__ uint32_t s_ncon[PARTS ? kPnQueries : 1];
  __shared__ NContact s_raw[PARTS ? 4 * kPpRound : 1];  // the raw contacts of a round's pairs, four slots each
  __shared__ uint32_t s_pool[kPnQueries * kRowCap]; // the block's partners that may touch: partner slot | query << 26
  __shared__ uint32_t s_qi[kPnQueries], s_np[kPnQueries];
  __shared__ uint32_t s_pool_n, s_sum;
  const int lane = threadIdx.x & 63;
  const int sub = lane & 7;
  const uint32_t qg = threadIdx.x >> 3;
  const uint32_t kq = xcd_logical_block_coop() * kPnQueries + qg;
  const bool live = kq < n;  // whole groups are live or not
  const uint32_t i = live ? T.sidx[kq] : 0u;
  if (threadIdx.x == 0) { s_pool_n = 0u; s_sum = 0u; }
  if (sub == 0) { s_qi[qg] = i; s_np[qg] = 0u; if (PARTS) s_ncon[qg] = 0u; }
  __syncthreads();
  uint32_t n_accepted = 0;
  const uint32_t oi = live ? order_id(T.ext, i) : 0u;
  if (live && oi != 0 && T.n >= 2) {  // world.rs:256
    Box q;
    uint32_t ca[3], d[3];
    const uint32_t P = 2u * T.levels;
    const uint32_t nb[3] = {(P + 2u) / 3u, (P + 1u) / 3u, P / 3u};  // prefix bits per axis (x is the most significant)
    if (T.ltb) {  // the query in cell order, its cells worked out by k_scatter_leaves
      const float4 qc = T.ltb[2 * kq], qr = T.ltb[2 * kq + 1];
      q.c = xyz(qc); q.r = xyz(qr);
      const uint32_t ra = f2u(qc.w), rd = f2u(qr.w);
      ca[0] = ra & 1023u; ca[1] = (ra >> 10) & 1023u; ca[2] = ra >> 20;
      d[0] = rd & 1023u; d[1] = (rd >> 10) & 1023u; d[2] = rd >> 20;
    } else {
      q.c = xyz(B.tb_c[i]); q.r = xyz(B.tb_r[i]);
      const float pad = pad_abs + 1e-5f * (fabs_rs(q.c.x) + fabs_rs(q.c.y) + fabs_rs(q.c.z) + q.r.x + q.r.y + q.r.z);
      pair_query_region(q.c, q.r, pad, sb, nb, ca, d, min_frac);
    }
    if (d[0] * d[1] * d[2] > kGridMaxCells) {
      if (sub == 0) *too_wide = 1u;
    } else {
#if defined(MGF_PN_ACCEPT) && MGF_PN_ACCEPT == 0
      PairSrcGlobal S; S.T = T;
      n_accepted = pair_query_accept(S, q, oi, n_owned, ca, d, nb, kMortonBits - (int)P, sub, lane & ~7, s_acc[qg]);
#else
      // (the search of k_pair_brick's slow path: a lane walks its z-columns of cells on its own, four leaf records in flight, hits appended
      // through an LDS counter - no ballot per record: in the dense part of a scene a cell holds ten bodies)
      BrickSrcGlobal S; S.T = T; S.nb[0] = nb[0]; S.nb[1] = nb[1]; S.nb[2] = nb[2]; S.shift = kMortonBits - (int)P;
      Comp A0; A0.kind = KIND_SPHERE; A0.p = mk3(0, 0, 0); A0.d = mk3(0, 0, 0); A0.r = 0.0f;
      brick_query<false>(S, q, A0, mk3(0, 0, 0), oi, n_owned, ca, d, (uint32_t)sub, (uint32_t)kCoopLanes, s_acc[qg], s_acc[qg], &s_np[qg]);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      n_accepted = *(volatile uint32_t*)&s_np[qg];
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      if (sub == 0) s_np[qg] = 0u;  // (the contacts' counter from here on: the group's lanes have read it)
#endif
      if (n_accepted > (uint32_t)kRowCap && sub == 0) atomicOr(overflow, 1u);
      // the accepted partners whose bounding spheres come within reach during the tick (comp_pair_far: nine in ten of a pile do not)
      // join the block's pool
      const uint32_t na = min(n_accepted, (uint32_t)kRowCap);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // (the group reads the staging row its lanes wrote)
      if (na) {
        if (PARTS) {
          // the candidate was accepted on i's tight box against j's FAT box; a contact is a touching inside both bodies' TIGHT swept boxes of
          // this tick (k_narrow_pairs_parts: three candidates in four leave here)
          const float4 ca = B.tb_c[i], ra = B.tb_r[i];
          for (uint32_t a = (uint32_t)sub; a < na; a += kCoopLanes) {
            const uint32_t j = s_acc[qg][a];
            const float4 cb = B.tb_c[j], rb = B.tb_r[j];
            const float slack = 1e-3f + 1e-5f * (fabs_rs(ca.x) + fabs_rs(ca.y) + fabs_rs(ca.z) + fabs_rs(cb.x) + fabs_rs(cb.y) + fabs_rs(cb.z));
            if (!(fabs_rs(ca.x - cb.x) > ra.x + rb.x + slack || fabs_rs(ca.y - cb.y) > ra.y + rb.y + slack || fabs_rs(ca.z - cb.z) > ra.z + rb.z + slack))
              s_pool[atomicAdd(&s_pool_n, 1u)] = j | (qg << 26);
          }
        } else {
          V3 vA;
          const Comp A = load_comp_moving(B, i, &vA);
          for (uint32_t a = (uint32_t)sub; a < na; a += kCoopLanes) {
            const uint32_t j = s_acc[qg][a];
            V3 vB;
            const Comp Bc = load_comp_moving(B, j, &vB);
            if (!comp_pair_far(A, vA, Bc, vB)) s_pool[atomicAdd(&s_pool_n, 1u)] = j | (qg << 26);
          }
        }
      }
    }
  }
  __syncthreads();
  // ---- the pool through the pair test, a lane each: contacts go to their query's row
  const uint32_t pool_n = s_pool_n;
  if (PARTS) {
    for (uint32_t r0 = 0; r0 < pool_n; r0 += kPpRound) {  // (the same trips for every thread of the block)
      const uint32_t m = min(pool_n - r0, kPpRound);
      for (uint32_t x = threadIdx.x; x < 4u * m; x += kCoopBlock) {  // a lane per part pair, part-pair-major: consecutive lanes run the same pair of shapes
        const uint32_t slot = x / m, e = x - slot * m, w = s_pool[r0 + e];
        parts_item(B, s_qi[w >> 26], w & 0x03FFFFFFu, slot, &s_raw[4u * e + slot]);
      }
      __syncthreads();
      for (uint32_t e = threadIdx.x; e < m; e += kCoopBlock) {  // a lane per pair of bodies: the pruner
        const uint32_t w = s_pool[r0 + e], j = w & 0x03FFFFFFu, g2 = w >> 26, ia = s_qi[g2];
        const PairCentres P = pair_centres(B, ia, j);
        uint32_t keep; float min_t; V3 nrm;
        const int nc = parts_prune(P, &s_raw[4u * e], &keep, &min_t, &nrm);
        if (nc) { rows_p[(size_t)ia * kRowCap + atomicAdd(&s_np[g2], 1u)] = j | ((uint32_t)nc << 27); atomicAdd(&s_ncon[g2], (uint32_t)nc); }
      }
      __syncthreads();
    }
  } else {
    for (uint32_t e = threadIdx.x; e < pool_n; e += kCoopBlock) {
      const uint32_t w = s_pool[e], j = w & 0x03FFFFFFu, g2 = w >> 26, ia = s_qi[g2];
      V3 vA, vB;
      const Comp A = load_comp_moving(B, ia, &vA), Bc = load_comp_moving(B, j, &vB);
      LocalContact lc;
      if (comp_pair_local(A, vA, Bc, vB, &lc)) rows_p[(size_t)ia * kRowCap + atomicAdd(&s_np[g2], 1u)] = j;
    }
  }
  {  // accepted partners (World::step's candidate statistic): one atomic per block, spread over many words
    uint32_t v = (live && sub == 0) ? n_accepted : 0u;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    if (lane == 0 && v) atomicAdd(&s_sum, v);
  }
  __syncthreads();
  if (live && sub == 0) {
    if (PARTS) { p_cnt[i] = s_ncon[qg]; p_ent[i] = s_np[qg]; }
    else p_cnt[i] = s_np[qg];
  }
  if (threadIdx.x == 0 && s_sum) atomicAdd(&pair_stat[blockIdx.x & (kPairStatWords - 1u)], s_sum);
}

// ---- the partners-to-be of the WIDE bodies (WideSpec, k_bodies.h) ------------------------------------------------------------------
// The grid's pair search finds pair (i, j) - j the body of the smaller order id - from i's side: i's tight box against the fat boxes of
// the bodies in the cells around it, as far as the largest fat half extent of the scene reaches.  A wide j is kept out of that reach and
// out of the partners (its leaf record carries no order id); here it looks for its i's itself: a workgroup per listed body, the cells its
// FAT box can reach (a body whose tight box meets fat_j has its own fat box's centre within fat_j grown by rmax), the reference's
// acceptance test (bvh.rs:297, world.rs:266) on every record there with a larger order id.  An accepted pair goes where the pair search
// would have put it - behind the entries of i's row (the order inside a row never mattered) - through the pair test first where the rows
// hold contacts.  One launch of kWideCap workgroups behind the pair search, whose rows and counts it appends to.
struct PairWide {
  const float4* list; const uint32_t* count;
  Lbvh T; const SceneBounds* sb; const SceneBounds* box; float pad_abs, min_frac;
  uint32_t n, n_owned;
  uint32_t contacts;   // 1: the rows hold contacts (the pair test runs here), 0: accepted partners
  uint32_t* rows_p; uint32_t* p_cnt; uint32_t* overflow; uint32_t* too_wide; uint32_t* pair_stat;
  const uint32_t* guard;
};
__global__ __launch_bounds__(kBlock) void k_pair_wide(Bodies B, PairWide A) {
  __shared__ uint32_t s_acc;
  if (*A.guard) return;
  const uint32_t nw = min(*A.count, kWideCap);
  if (blockIdx.x >= nw) return;
  if (threadIdx.x == 0) s_acc = 0u;
  __syncthreads();
  const float4 fc = A.list[2 * blockIdx.x], fr = A.list[2 * blockIdx.x + 1];
  const uint32_t j = f2u(fc.w), oj = f2u(fr.w);
  Box fat; fat.c = xyz(fc); fat.r = xyz(fr);
  const uint32_t P = 2u * A.T.levels;
  const uint32_t nb[3] = {(P + 2u) / 3u, (P + 1u) / 3u, P / 3u};
  uint32_t ca[3], d[3];
  pair_query_region(fat.c, fat.r, pair_query_pad(fat.c, fat.r, A.pad_abs), A.sb, nb, ca, d, A.min_frac, A.box);
  const unsigned long long ncell = (unsigned long long)d[0] * d[1] * d[2];
  if (ncell > (1ull << 22)) { if (threadIdx.x == 0) *A.too_wide = 1u; return; }  // (four million cells: the host takes the tree walk)
  const int shift = kMortonBits - (int)P;
  V3 vB = mk3(0, 0, 0);
  Comp Bc; Bc.kind = KIND_SPHERE; Bc.p = mk3(0, 0, 0); Bc.d = mk3(0, 0, 0); Bc.r = 0.0f;
  if (A.contacts) Bc = load_comp_moving(B, j, &vB);
  uint32_t accepted = 0;
  for (uint32_t idx = threadIdx.x; idx < (uint32_t)ncell; idx += (uint32_t)kBlock) {
    const uint32_t cz = idx % d[2], t = idx / d[2], cy = t % d[1], cx = t / d[1];
    const uint32_t code = (expand10((ca[0] + cx) << (10u - nb[0])) << 2) | (expand10((ca[1] + cy) << (10u - nb[1])) << 1) | expand10((ca[2] + cz) << (10u - nb[2]));
    const uint32_t cell = code >> shift;
    for (uint32_t p = A.T.cell_lo[cell], p1 = A.T.cell_lo[cell + 1]; p < p1; ++p) {
      const uint32_t i = A.T.sidx[p];
      if (i == j || order_id(A.T.ext, i) <= oj) continue;  // world.rs:266: partners have the smaller order id (j is owned: listed by k_integrate)
      const float4 tc = A.T.ltb[2 * (size_t)p], tr = A.T.ltb[2 * (size_t)p + 1];
      Box q; q.c = xyz(tc); q.r = xyz(tr);
      if (!box_overlaps(q, fat)) continue;  // the reference's own acceptance test (bvh.rs:297): i's tight box, j's fat box
      ++accepted;
      if (A.contacts) {
        V3 vA;
        const Comp Ac = load_comp_moving(B, i, &vA);
        LocalContact lc;
        if (!comp_pair_local(Ac, vA, Bc, vB, &lc)) continue;
      }
      const uint32_t pos = atomicAdd(&A.p_cnt[i], 1u);
      if (pos < (uint32_t)kRowCap) A.rows_p[(size_t)i * kRowCap + pos] = j;
      else atomicOr(A.overflow, 1u);
    }
  }
  if (accepted) atomicAdd(&s_acc, accepted);
  __syncthreads();
  if (threadIdx.x == 0 && s_acc && A.pair_stat) atomicAdd(&A.pair_stat[blockIdx.x & (kPairStatWords - 1u)], s_acc);
}

// ContactConstraint::new for the rows of a world with bodies of up to two components: k_contacts_rows' work with a manifold of up to four
// contacts per row entry (consecutive single-contact records that share its normal and tangents: ContactConstraint::solve, solver.rs:219-248,
// handles the contacts of a constraint one after the other on the same velocities - k_setup_pairs) and up to four parked contacts per
// (body, face) slot.  A block per 256 bodies; the entries of its bodies as a list in LDS in canonical order, an entry per thread.
struct ContactsParts {
  const StepCounts* sc;
  uint32_t n, cap_c, cap_t, t_stride;
  const uint32_t *rows_p, *t_cnt, *p_cnt, *p_ent, *base, *tcn, *tpos;
  const NContact* t_out;  // t_stride per slot; .lb.w of the first = the slot's contact count
  float dt, baumgarte, slop;
  CRec* cons; uint2* ab; uint32_t* degb; RevEnt* rev; uint32_t rev_cap; uint32_t* rev_flag; uint32_t* flag;
  const uint32_t* ext;
};
constexpr uint32_t kCpEntCap = 1024;  // entries of a block staged per pass
__global__ __launch_bounds__(kBlock) void k_contacts_rows_parts(Bodies B, TerrainDev M, ContactsParts A) {
  __shared__ uint32_t s_tj[kCpEntCap], s_to[kCpEntCap];  // the bodies' rows as they are, and the partners' order ids
  __shared__ uint32_t s_j[kCpEntCap];            // the pass's entries in canonical order: partner | contacts << 27 ...
  __shared__ uint16_t s_b[kCpEntCap], s_off[kCpEntCap];  // ... the owner (the block's body) and the entry's first constraint behind the body's terrain constraints
  __shared__ uint32_t s_cb[kBlock];              // body -> id of its first partner constraint
  __shared__ NContact s_raw[4 * kPpRound];       // the raw contacts of a round's entries, four slots each
  __shared__ uint32_t s_wave[kBlock / 64];
  if (A.sc->fail) return;  // (the scan's closing thread found a flag up or a capacity exceeded: the host re-runs the phase)
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  // (the launch is TWICE the bodies' blocks: the second half does the terrain constraints of the same bodies, the first the entries - two
  // chains of dependent round trips side by side instead of one behind the other on a world too small to hide either)
  const uint32_t nbb = gridDim.x / 2u;
  const bool terrain_half = blockIdx.x >= nbb;
  const uint32_t i0 = (terrain_half ? blockIdx.x - nbb : blockIdx.x) * (uint32_t)kBlock, i = i0 + (uint32_t)t;
  uint32_t ne = 0, run = 0, base_i = 0;
  if (i < A.n) { ne = terrain_half ? 0u : A.p_ent[i]; run = A.tcn[i]; base_i = A.base[i]; }
  uint32_t inc = ne;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(inc, o); if (lane >= o) inc += v; }
  if (lane == 63) s_wave[wv] = inc;
  __syncthreads();
  uint32_t before = 0, total = 0;
  for (int k = 0; k < kBlock / 64; ++k) { const uint32_t v = s_wave[k]; if (k < wv) before += v; total += v; }
  const uint32_t excl = before + inc - ne;
  s_cb[t] = base_i + run;
  const uint32_t* rp = A.rows_p + (size_t)i * kRowCap;
  // ---- the terrain contacts' constraints (world.rs:243-251): the body's own thread
  if (run && terrain_half) {
    const uint32_t nt = A.t_cnt[i], tp = A.tpos[i];
    const V3 mx = mk3(M.x[0], M.x[1], M.x[2]);
    const BodyDyn Ad = load_dyn(B.srec, i), S = static_dyn();
    const BodyPack Pa = load_pack(B, i, false);
    uint32_t c = base_i;
    for (uint32_t a = 0; a < nt; ++a) {
      const NContact in0 = A.t_out[(size_t)A.t_stride * (tp + a)];
      const uint32_t nc = f2u(in0.lb.w);
      for (uint32_t k = 0; k < nc; ++k, ++c) {
        const NContact in = k == 0 ? in0 : A.t_out[(size_t)A.t_stride * (tp + a) + k];
        // Static{ center: terrain.center(), friction: 0.0 } world.rs:247
        const CRec r = make_constraint(i, kNone, Ad, xyz(Pa.ei), Pa.ei.w, Pa.dl.w, S, mx, 0.0f, 0.0f, xyz(in.n), xyz(in.la), xyz(in.lb), A.dt, A.baumgarte, A.slop);
        store_crec(&A.cons[c], r);
        A.ab[c] = make_uint2(i, kNone);
      }
    }
  }
  for (uint32_t w0 = 0; w0 < total; w0 += kCpEntCap) {
    __syncthreads();
    // the body's entries to their places: ascending order id (the canonical insertion order), each behind the contacts of the ones before it.
    // (The row and its partners' order ids once, into LDS: ranked from global memory a body of eight entries made 128 dependent look-ups.)
    const bool staged = excl >= w0 && excl + ne <= w0 + kCpEntCap;  // (a body across a window's edge ranks from global memory)
    if (staged) for (uint32_t a = 0; a < ne; ++a) { const uint32_t w = rp[a]; s_tj[excl - w0 + a] = w; s_to[excl - w0 + a] = order_id(A.ext, w & 0x07FFFFFFu); }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // (a lane reads what it wrote itself)
    for (uint32_t a = 0; a < ne; ++a) {
      const uint32_t w = staged ? s_tj[excl - w0 + a] : rp[a], oa = staged ? s_to[excl - w0 + a] : order_id(A.ext, w & 0x07FFFFFFu);
      uint32_t rank = 0, off = 0;
      for (uint32_t q = 0; q < ne; ++q) {
        const uint32_t x = staged ? s_tj[excl - w0 + q] : rp[q], ox = staged ? s_to[excl - w0 + q] : order_id(A.ext, x & 0x07FFFFFFu);
        if (ox < oa) { ++rank; off += x >> 27; }
      }
      const uint32_t pos = excl + rank - w0;
      if (pos < kCpEntCap) { s_j[pos] = w; s_b[pos] = (uint16_t)t; s_off[pos] = (uint16_t)off; }
    }
    __syncthreads();
    const uint32_t m = min(total - w0, kCpEntCap);
    for (uint32_t r0 = 0; r0 < m; r0 += kPpRound) {  // (the same trips for every thread of the block)
      const uint32_t mr = min(m - r0, kPpRound);
      __syncthreads();
      for (uint32_t x = (uint32_t)t; x < 4u * mr; x += (uint32_t)kBlock) {  // a lane per part pair
        const uint32_t slot = x / mr, e = x - slot * mr;
        parts_item(B, i0 + s_b[r0 + e], s_j[r0 + e] & 0x07FFFFFFu, slot, &s_raw[4u * e + slot]);
      }
      __syncthreads();
      for (uint32_t e = (uint32_t)t; e < mr; e += (uint32_t)kBlock) {  // a lane per entry: the pruner, then a record per contact kept
        const uint32_t w = s_j[r0 + e], j = w & 0x07FFFFFFu, nc = w >> 27, b = s_b[r0 + e], ia = i0 + b;
        const PairCentres P = pair_centres(B, ia, j);
        uint32_t keep; float min_t; V3 nrm;
        const uint32_t got = (uint32_t)parts_prune(P, &s_raw[4u * e], &keep, &min_t, &nrm);
        if (got != nc) { *A.flag = 1u; continue; }  // the pair search's evaluation of the same manifold and this one disagree
        const BodyPack Pa = load_pack(B, ia, false), Pb = load_pack(B, j, false);
        const BodyDyn Ad = load_dyn(B.srec, ia), Bd = load_dyn(B.srec, j);
        const uint32_t c0 = s_cb[b] + s_off[r0 + e];
        for (uint32_t k = 0; k < nc; ++k) {
          const uint32_t c = c0 + k;
          const LocalContact kc = parts_local(P, s_raw[4u * e + ((keep >> (2u * k)) & 3u)]);
          const CRec r = make_constraint(ia, j, Ad, xyz(Pa.ei), Pa.ei.w, Pa.dl.w, Bd, xyz(Pb.ei), Pb.ei.w, Pb.dl.w, nrm, kc.la, kc.lb, A.dt, A.baumgarte, A.slop);
          store_crec(&A.cons[c], r);
          A.ab[c] = make_uint2(ia, j);
          // body j's row of the constraints it takes part in as `b` (k_chain_rows); as `a` a body owns a contiguous id range
          const uint32_t pos = atomicAdd(&A.degb[j], 1u);
          if (pos < A.rev_cap) A.rev[(size_t)j * A.rev_cap + pos] = RevEnt{c, ia, order_id(A.ext, ia), 0u};
          else *A.rev_flag = 1u;
        }
      }
    }
  }
}

}  // namespace mgf
