// Resident body state, integration and the small per-tick helpers.  (Part of the kernel set described in kernels.h.)
#pragma once
#include <stddef.h>

#include <type_traits>

#include "dev_geom.h"
#include "host_bvh.h"

namespace mgf {

constexpr uint32_t kNone = 0xFFFFFFFFu;
constexpr int kBlock = 256;

__device__ __forceinline__ float4 ld4(const float4* p) { return *p; }
__device__ __forceinline__ V3 xyz(float4 v) { return mk3(v.x, v.y, v.z); }
__device__ __forceinline__ float4 mk4(V3 v, float w) { return make_float4(v.x, v.y, v.z, w); }
__device__ __forceinline__ float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }

// Resident rigid-body state (RigidBodyVec, physics.rs:141-155), SoA of 16-byte words so every
// streaming access is a coalesced dwordx4 per lane.
struct Bodies {
  float4* x;      // x.xyz, -
  float4* q;      // s, v.xyz
  float4* srec;   // 4 words/body, the record the solver gathers:
                  //   [0] v.xyz, w.x   [1] w.y, w.z, inv_mass, I00   [2] I01 I02 I10 I11   [3] I12 I20 I21 I22
                  //   (I = world inv_moment, column-major Icr)
  float4* sp0;    // force.xyz, restitution
  float4* sp1;    // torque.xyz, friction
  float4* ctor;   // constructor: kind bits, r, half_h, -
  float4* imb;    // 3 words/body: inv_moment_body columns
  float4* delta;  // collider.1 (= v*dt), friction
  float4* einfo;  // x + delta (RigidBodyInfo.x, physics.rs:282), restitution
  float4* col0;   // collider shape: p.xyz, r
  float4* col1;   //                 d.xyz, kind bits
  float4* bpk;    // 4 words/body, or null: col0, delta, einfo, col1 once more, side by side - what ContactConstraint::new needs of a body
                  // besides srec, in ONE 64-byte sector instead of four (k_setup_*; written by k_integrate / k_import_ghosts, valid from
                  // there until the bodies change: the host passes null otherwise)
  float4* tb_c;   // tight swept AABB centre / half extents
  float4* tb_r;
  float4* fb_c;   // fat AABB (persistent; world.rs:181,237)
  float4* fb_r;
  // Bodies of several components (BASELINE config 5; not in the reference): null unless the world has
  // one.  kMaxParts slots per body; pcount 0 = an ordinary body.  Local parts are fixed in the body frame relative to
  // the centre of mass; world parts are rebuilt by k_integrate like the single collider (physics.rs:243-251).
  uint32_t* pcount;
  float4* lp0;    // local part: p.xyz (sphere centre / capsule start), r
  float4* lp1;    //             d.xyz (capsule axis), kind bits
  float4* wp0;    // world part at the start of the tick's motion, same layout
  float4* wp1;
  // (r06) a body of MORE than kMaxParts components (up to kBigParts) keeps its parts in the world's pool arrays - local and world parts, the
  // layout of lp* / wp* - at an offset that sits in the body's first local part slot (lp0[kMaxParts * i].x, as bits): whatever moves a
  // body's row (re-sorting the store, clones) moves the offset along, the pool itself never moves.  Null unless the world has such a body.
  float4* xl0;
  float4* xl1;
  float4* xw0;
  float4* xw1;
};
constexpr int kMaxParts = 4;   // part slots per body in the part arrays (round 3: 2 -> 4)
constexpr int kBigParts = 32;  // the most components a body may have (r06; the reference's Compound, compound.rs:232-352, has no limit - it is static)
constexpr int kTileParts = 4;  // part slots of a ghost / migrant record of the tile protocol (its record sizes are part of the ABI; r04: 2 -> 4 =
                               // kMaxParts, every body the store can hold crosses tiles)

constexpr int kBoundSlots = 64, kBoundSlotInts = 32;  // partial scene bounds: lo[3], hi[3], rmax[3] per slot, one 128-byte line each
struct SceneBounds { int lo[3]; int hi[3]; uint32_t n_refits; uint32_t pad; int rmax[3]; uint32_t pad2; };  // ordered-int encoded floats; rmax = largest fat half extent

__device__ __forceinline__ int f_ord(float f) { int i = __builtin_bit_cast(int, f); return i >= 0 ? i : (i ^ 0x7FFFFFFF); }
__host__ __device__ __forceinline__ float ord_f(int i) { int j = i >= 0 ? i : (i ^ 0x7FFFFFFF); return __builtin_bit_cast(float, j); }

// Part k of body i (pc = its part count, > 0): the local or the world record pair, from the body's four slots or - a body of more than
// kMaxParts components - from the pool.
__device__ __forceinline__ size_t part_at(const Bodies& B, uint32_t i, uint32_t k, uint32_t pc) {
  return pc > (uint32_t)kMaxParts ? (size_t)__builtin_bit_cast(uint32_t, B.lp0[(size_t)kMaxParts * i].x) + k : (size_t)kMaxParts * i + k;
}
__device__ __forceinline__ void world_part(const Bodies& B, uint32_t i, uint32_t k, uint32_t pc, float4& a, float4& b) {
  const size_t at = part_at(B, i, k, pc);
  if (pc > (uint32_t)kMaxParts) { a = B.xw0[at]; b = B.xw1[at]; } else { a = B.wp0[at]; b = B.wp1[at]; }
}
__device__ __forceinline__ M3 load_imb(const float4* imb, uint32_t i) {
  float4 a = imb[3 * i], b = imb[3 * i + 1], c = imb[3 * i + 2];
  return m3_cols(xyz(a), xyz(b), xyz(c));
}

// ------------------------------------------------------------------------------------------
// complete_motion + integrate, one pass.
// ------------------------------------------------------------------------------------------
// Scene bounds of the fat boxes (what k_scene_bounds computes) while they are in a kernel's registers: every thread of the
// block brings its body's ordered-int centre (lo = hi) and half extents, or the identity; block reduce, then nine atomics into
// one of kBoundSlots partial records, each on its own cache line (same-line atomics serialise).  The clearing launch of the
// collide phase (k_zero_many) folds the partial records into *sb.
__device__ __forceinline__ void bounds_block_accumulate(const int blo[3], const int bhi[3], const int brm[3], int* sb_part) {
  __shared__ int s_red[9][kBlock / 64];
  for (int k = 0; k < 3; ++k) {
    int a = blo[k], b = bhi[k], c = brm[k];
    for (int off = 32; off > 0; off >>= 1) { a = min(a, __shfl_xor(a, off)); b = max(b, __shfl_xor(b, off)); c = max(c, __shfl_xor(c, off)); }
    if ((threadIdx.x & 63) == 0) { s_red[k][threadIdx.x >> 6] = a; s_red[3 + k][threadIdx.x >> 6] = b; s_red[6 + k][threadIdx.x >> 6] = c; }
  }
  __syncthreads();
  if (threadIdx.x < 9) {
    const int k = threadIdx.x;
    int v = s_red[k][0];
    for (int w = 1; w < kBlock / 64; ++w) v = k < 3 ? min(v, s_red[k][w]) : max(v, s_red[k][w]);
    int* slot = sb_part + (size_t)(blockIdx.x % kBoundSlots) * kBoundSlotInts + k;
    if (k < 3) atomicMin(slot, v); else atomicMax(slot, v);
  }
}

// `tail(i, tight box)` runs for every integrated body while its new bounds are still in registers (the terrain candidate
// rows of the one-synchronisation tick, k_broadphase.h); NoTail for everything else.
struct NoTail {
  static constexpr int kLdsWords = 1;
  __device__ __forceinline__ void stage(float4*) const {}
  static constexpr bool kNear = false;
  float4* near_list = nullptr; uint32_t* near_cnt = nullptr;
  __device__ __forceinline__ void operator()(uint32_t, const Box&, const float4*, uint32_t&, unsigned long long&) const {}
};
// The counting sort's first half (k_morton_count's work) while the body's fat box is in registers: its Morton cell over the box of
// the PREVIOUS tick's scene bounds (`grid`: the scene moves a fraction of a cell per tick; the quantisation clamps, and the pair search
// finds every body whatever the box is - only how evenly the cells fill depends on it) and its arrival rank inside the cell.
struct CellSort { const SceneBounds* grid; int shift; float min_frac; uint32_t* cell_of; uint32_t* rank; uint32_t* cell_cnt; };
// (r06) WIDE bodies: the few whose fat box is much larger than everybody else's - a body that left the scene and has been falling for a
// thousand ticks sweeps 4 m per tick.  SceneBounds::rmax, the largest fat half extent, is the reach of EVERY query of the cell grid, and
// the bounds are what the cells are laid over: one such body made a million-sphere world four times slower (EXPERIMENTS.md, round 5).
// With `limit` set (the host: 1.5 x the largest half extent of the bodies that were NOT wide in the last tick) k_integrate keeps the
// bodies above it out of the scene bounds and rmax and lists them (fat box, slot, order id; at most `cap`: more is a failed tick, run
// again without the list); the grid's pair search never accepts a listed body as a partner (its leaf record carries no order id:
// scatter_leaf) and k_pair_wide finds its partners-to-be from ITS side, by one launch over the few of them.  The accepted set is the
// reference's (the same predicate on the same boxes, bvh.rs:283-310), whatever the limit.
constexpr uint32_t kWideCap = 64;
struct WideSpec { float limit[3]; float4* list; uint32_t* count; const uint32_t* ext; };
__device__ __forceinline__ bool is_wide(const float* limit, float4 fr) { return fr.x > limit[0] || fr.y > limit[1] || fr.z > limit[2]; }
__device__ __forceinline__ uint32_t morton_cell_of(V3 c, const SceneBounds* sb, float min_frac, int shift);  // k_broadphase.h
template <class Tail>
__global__ __launch_bounds__(kBlock) void k_integrate(Bodies B, uint32_t n, float dt, float fat_margin, int do_complete,
                                                      int do_integrate, SceneBounds* sb, const uint32_t* guard, Tail tail, int* sb_part, CellSort cs, WideSpec wd) {
  if (guard && *guard) return;  // a speculative tick behind a failed one (see k_reset_step)
  __shared__ float4 s_tail[Tail::kLdsWords];
  __shared__ uint32_t s_near[2];  // bodies of this block that list a terrain face, where their records start in the tick's list
  tail.stage(s_tail);
  if (Tail::kNear && threadIdx.x == 0) s_near[0] = 0u;
  if (Tail::kLdsWords > 1) __syncthreads();
  float4 near0 = make_float4(0, 0, 0, 0), near1 = near0;
  uint32_t near_nt = 0;
  unsigned long long near_pk = 0ull;
  __shared__ float4 s_rec[kBlock / 64][8 * 65];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  bool live = i < n;
  bool refit = false;
  int blo[3] = {0x7FFFFFFF, 0x7FFFFFFF, 0x7FFFFFFF}, bhi[3] = {(int)0x80000000, (int)0x80000000, (int)0x80000000}, brm[3] = {0, 0, 0};
  if (live) {
    float4 xw = B.x[i];
    float4 dl = B.delta[i];
    V3 x = xyz(xw);
    if (do_complete) x = x + xyz(dl);  // physics.rs:262-269
    if (do_integrate) {
      float4 qw = B.q[i];
      float4 s0 = B.srec[4 * i], s1 = B.srec[4 * i + 1];
      float4 p0 = B.sp0[i], p1 = B.sp1[i], ct = B.ctor[i];
      V3 v = mk3(s0.x, s0.y, s0.z), w = mk3(s0.w, s1.x, s1.y);
      float inv_mass = s1.z;
      Quat q = mkq(qw.x, mk3(qw.y, qw.z, qw.w));
      // physics.rs:226-227
      q = normalize(q + mkq(0.0f, w * dt) * 0.5f * q);
      // physics.rs:231-232
      M3 R = m3_from_quat(q);
      M3 I = R * load_imb(B.imb, i) * transpose(R);
      // physics.rs:236, 240
      v = v + xyz(p0) * inv_mass * dt;
      w = w + I * xyz(p1) * dt;
      // physics.rs:244-250
      int kind = (int)f2u(ct.x);
      V3 d = v * dt;
      Comp col;
      Box tb;
      const uint32_t pc = B.pcount ? B.pcount[i] : 0u;
      if (pc) {  // a body of several parts: the collider slot carries the centre (a radius-0 sphere), the parts collide
        col.kind = KIND_SPHERE; col.p = x; col.d = mk3(0.0f, 0.0f, 0.0f); col.r = 0.0f;
        const bool big = pc > (uint32_t)kMaxParts;  // (its parts live in the pool: Bodies::xl0)
        const size_t at0 = part_at(B, i, 0u, pc);
        const float4 *L0 = big ? B.xl0 : B.lp0, *L1 = big ? B.xl1 : B.lp1;
        float4 *W0 = big ? B.xw0 : B.wp0, *W1 = big ? B.xw1 : B.wp1;
        for (uint32_t k = 0; k < pc; ++k) {
          float4 l0 = L0[at0 + k], l1 = L1[at0 + k];
          Comp part; part.kind = (int)f2u(l1.w); part.r = l0.w;
          part.p = x + rotate(q, xyz(l0));
          part.d = part.kind == KIND_SPHERE ? mk3(0.0f, 0.0f, 0.0f) : rotate(q, xyz(l1));
          W0[at0 + k] = mk4(part.p, part.r);
          W1[at0 + k] = mk4(part.d, l1.w);
          Box pb = swept_bounds(part, d);
          tb = k == 0 ? pb : box_combine(tb, pb);
        }
      } else {
        col = construct(kind, ct.y, ct.z, x, q);
        tb = swept_bounds(col, d);
      }
      B.q[i] = make_float4(q.s, q.v.x, q.v.y, q.v.z);
      // the two 64-byte records of the body leave through LDS (below): stored by their own lanes they would be 16-byte pieces
      // 64 bytes apart, 64 partial lines per instruction
      s_rec[wv][0 * 65 + lane] = make_float4(v.x, v.y, v.z, w.x);
      s_rec[wv][1 * 65 + lane] = make_float4(w.y, w.z, inv_mass, I.c[0].x);
      s_rec[wv][2 * 65 + lane] = make_float4(I.c[0].y, I.c[0].z, I.c[1].x, I.c[1].y);
      s_rec[wv][3 * 65 + lane] = make_float4(I.c[1].z, I.c[2].x, I.c[2].y, I.c[2].z);
      B.delta[i] = mk4(d, p1.w);
      B.einfo[i] = mk4(x + d, p0.w);
      B.col0[i] = mk4(col.p, col.r);
      B.col1[i] = mk4(col.d, u2f((uint32_t)col.kind));
      s_rec[wv][4 * 65 + lane] = mk4(col.p, col.r); s_rec[wv][5 * 65 + lane] = mk4(d, p1.w);
      s_rec[wv][6 * 65 + lane] = mk4(x + d, p0.w); s_rec[wv][7 * 65 + lane] = mk4(col.d, u2f((uint32_t)col.kind));
      B.tb_c[i] = mk4(tb.c, 0.0f);
      B.tb_r[i] = mk4(tb.r, 0.0f);
      Box fb; fb.c = xyz(B.fb_c[i]); fb.r = xyz(B.fb_r[i]);
      if (!box_contains(fb, tb)) {  // world.rs:235-238
        fb.c = tb.c;
        fb.r = tb.r + mk3(fat_margin, fat_margin, fat_margin);
        B.fb_c[i] = mk4(fb.c, 0.0f);
        B.fb_r[i] = mk4(fb.r, 0.0f);
        refit = true;
      }
      blo[0] = bhi[0] = f_ord(fb.c.x); blo[1] = bhi[1] = f_ord(fb.c.y); blo[2] = bhi[2] = f_ord(fb.c.z);
      brm[0] = f_ord(fb.r.x); brm[1] = f_ord(fb.r.y); brm[2] = f_ord(fb.r.z);
      if (wd.list && is_wide(wd.limit, mk4(fb.r, 0.0f))) {  // a wide body: listed, and not part of the scene's bounds
        const uint32_t at = atomicAdd(wd.count, 1u);
        if (at < kWideCap) { wd.list[2 * at] = mk4(fb.c, u2f(i)); wd.list[2 * at + 1] = mk4(fb.r, u2f(wd.ext ? wd.ext[i] : i)); }
        blo[0] = blo[1] = blo[2] = 0x7FFFFFFF; bhi[0] = bhi[1] = bhi[2] = (int)0x80000000; brm[0] = brm[1] = brm[2] = 0;
      }
      if (cs.grid) {
        const uint32_t cell = morton_cell_of(fb.c, cs.grid, cs.min_frac, cs.shift);
        cs.cell_of[i] = cell;
        cs.rank[i] = atomicAdd(&cs.cell_cnt[cell], 1u);
      }
      tail(i, tb, s_tail, near_nt, near_pk);
      if (Tail::kNear && near_nt) { near0 = mk4(col.p, col.r); near1 = mk4(d, u2f(i)); }
    } else if (do_complete) {
      B.einfo[i] = mk4(x + xyz(dl), B.einfo[i].w);
    }
    if (do_complete) B.x[i] = mk4(x, 0.0f);
  }
  if (do_integrate) {  // srec (and the packed copy of collider, motion, info): the wave's 64 records are one contiguous 4 KB each
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // (a wave's LDS accesses are served in order; it reads only what it wrote)
    const uint32_t i0 = blockIdx.x * kBlock + (uint32_t)wv * 64u;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int f = it * 64 + lane, rec = f >> 2, k = f & 3;
      if (i0 + (uint32_t)rec < n) {
        B.srec[4 * (size_t)i0 + f] = s_rec[wv][k * 65 + rec];
        if (B.bpk) B.bpk[4 * (size_t)i0 + f] = s_rec[wv][(4 + k) * 65 + rec];
      }
    }
  }
  if (!do_integrate || sb == nullptr) return;
  // the block's records for k_terrain_contacts (TerrainRowsTail): ranks through LDS, ONE atomic per block on the list's length (a wave
  // each - thousands of them on one word once the pile touches the walls - cost this kernel 9 us)
  uint32_t near_rank = 0;
  if (Tail::kNear && near_nt) near_rank = atomicAdd(&s_near[0], 1u);
  // refit count: one atomic per block
  int nref = __syncthreads_count(refit ? 1 : 0);
  if (threadIdx.x == 0 && nref) atomicAdd(&sb->n_refits, (uint32_t)nref);
  if (Tail::kNear && tail.near_list) {
    if (threadIdx.x == 0 && s_near[0]) s_near[1] = atomicAdd(tail.near_cnt, s_near[0]);
    __syncthreads();
    if (near_nt) {
      const size_t at = (size_t)s_near[1] + near_rank;
      tail.near_list[3 * at] = near0;
      tail.near_list[3 * at + 1] = near1;
      tail.near_list[3 * at + 2] = make_float4(u2f(near_nt), u2f((uint32_t)near_pk), u2f((uint32_t)(near_pk >> 32)), 0.0f);
    }
  }
  if (!sb_part) return;
  bounds_block_accumulate(blo, bhi, brm, sb_part);
}

// Scene bounds of the fat-box centres (Morton quantisation): grid-stride, block reduce in LDS,
// one atomic per block and axis.
__global__ __launch_bounds__(kBlock) void k_scene_bounds(const float4* fb_c, const float4* fb_r, uint32_t n, SceneBounds* sb) {
  int lo[3] = {0x7FFFFFFF, 0x7FFFFFFF, 0x7FFFFFFF}, hi[3] = {(int)0x80000000, (int)0x80000000, (int)0x80000000};
  int rm[3] = {0, 0, 0};  // half extents are >= 0: plain int order
  for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    float4 c = fb_c[i], r = fb_r[i];
    int o[3] = {f_ord(c.x), f_ord(c.y), f_ord(c.z)};
    int e[3] = {f_ord(r.x), f_ord(r.y), f_ord(r.z)};
    for (int k = 0; k < 3; ++k) { lo[k] = min(lo[k], o[k]); hi[k] = max(hi[k], o[k]); rm[k] = max(rm[k], e[k]); }
  }
  __shared__ int s_lo[3][kBlock / 64], s_hi[3][kBlock / 64], s_rm[3][kBlock / 64];
  for (int k = 0; k < 3; ++k) {
    int a = lo[k], b = hi[k], c = rm[k];
    for (int off = 32; off > 0; off >>= 1) { a = min(a, __shfl_xor(a, off)); b = max(b, __shfl_xor(b, off)); c = max(c, __shfl_xor(c, off)); }
    if ((threadIdx.x & 63) == 0) { s_lo[k][threadIdx.x >> 6] = a; s_hi[k][threadIdx.x >> 6] = b; s_rm[k][threadIdx.x >> 6] = c; }
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    int k = threadIdx.x, a = s_lo[k][0], b = s_hi[k][0], c = s_rm[k][0];
    for (int w = 1; w < kBlock / 64; ++w) { a = min(a, s_lo[k][w]); b = max(b, s_hi[k][w]); c = max(c, s_rm[k][w]); }
    atomicMin(&sb->lo[k], a);
    atomicMax(&sb->hi[k], b);
    atomicMax(&sb->rmax[k], c);
  }
}

// RigidBodyInfo.x after a state write (physics.rs:282).
__global__ __launch_bounds__(kBlock) void k_refresh_einfo(Bodies B, uint32_t n) {
  uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i < n) B.einfo[i] = mk4(xyz(B.x[i]) + xyz(B.delta[i]), B.einfo[i].w);
}
// `spec`: the tick is being enqueued before the previous one has been read back (mgf_world_step_many).  If that one
// turns out to have failed a capacity check (its StepCounts still sit in `sc`), this tick must not touch the state: the
// guard word makes k_integrate and the whole collide phase no-ops, and the host re-runs both ticks.
__device__ __forceinline__ void reset_step(SceneBounds* sb, uint32_t* err, uint32_t* guard, const uint32_t* prev_fail, int spec, int* sb_part, uint32_t* near_cnt) {
  if (near_cnt && threadIdx.x == 0) *near_cnt = 0u;  // (the length of the list k_integrate's tail is about to build)
  if (spec && (*prev_fail || err[2])) { if (threadIdx.x == 0) *guard = 1u; return; }  // (err[2]: the solvers' abort flag, see k_tick_clear)
  if (sb_part && threadIdx.x < kBoundSlots) {  // launched with 64 threads: one partial record each
    int* slot = sb_part + (size_t)threadIdx.x * kBoundSlotInts;
    for (int k = 0; k < 3; ++k) { slot[k] = 0x7FFFFFFF; slot[3 + k] = (int)0x80000000; slot[6 + k] = 0; }
  }
  if (threadIdx.x == 0) {
    *guard = 0u;
    for (int k = 0; k < 3; ++k) { sb->lo[k] = 0x7FFFFFFF; sb->hi[k] = (int)0x80000000; }
    sb->n_refits = 0; sb->pad = 0; sb->pad2 = 0;
    for (int k = 0; k < 3; ++k) sb->rmax[k] = 0;
    err[0] = 0; err[1] = 0; err[8] = 0;  // traversal stack overflow, candidate row overflow, fused narrowphase mismatch
  }
}
__global__ void k_reset_step(SceneBounds* sb, uint32_t* err, uint32_t* guard, const uint32_t* prev_fail, int spec, int* sb_part, uint32_t* near_cnt = nullptr) {
  reset_step(sb, err, guard, prev_fail, spec, sb_part, near_cnt);
}
// ... of several worlds on one stream in one launch (the tile set: a workgroup per world)
struct ResetOne { SceneBounds* sb; uint32_t* err; uint32_t* guard; const uint32_t* prev_fail; int* sb_part; uint32_t* near_cnt; };
constexpr int kWorldBatch = 8;
struct ResetBatch { ResetOne t[kWorldBatch]; };
__global__ void k_reset_step_batch(ResetBatch A) {
  const ResetOne& R = A.t[blockIdx.x];
  reset_step(R.sb, R.err, R.guard, R.prev_fail, 0, R.sb_part, R.near_cnt);
}

}  // namespace mgf
