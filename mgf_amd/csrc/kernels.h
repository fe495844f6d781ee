// Hand-written HIP kernels of the mgf per-tick hot path for gfx950 (wave64).
//
//   k_integrate        RigidBodyVec::complete_motion + integrate (physics.rs:222-269), swept AABB
//                      (bounds.rs:60-68), fat-AABB refit test (world.rs:234-238)
//   k_scene_bounds / k_morton_count / k_scatter_leaves
//                      bodies counting-sorted into Morton cells (cell = 2L-bit prefix of the 30-bit code of the fat-box
//                      centre); the same kernels build the static grid over a terrain mesh's face boxes
//   k_pair_grid        BVH::query (bvh.rs:283-310) of the world tree for every body at once by cell enumeration;
//                      k_lbvh_low / k_lbvh_top + k_pair_rows (implicit 4-ary tree over the cells, cooperative walk)
//                      when one body spans too many cells; the hit SET is the reference's because acceptance is its own
//                      predicate on the leaf boxes
//   k_terrain_grid     Mesh::contacts' BVH::query by cell enumeration over the face boxes, hits stored as DFS ranks;
//                      k_terrain_rows = the walk of the flattened reference tree in the reference's order
//   k_candidates<FILL> the exact two-pass (count, fill) tree walk used when a candidate row overflows
//   k_rows_to_csr      rows -> CSR, terrain faces in DFS order (partner contacts are ordered by k_count_contacts)
//   k_narrow_pairs<A,B> / k_narrow_terrain<A>
//                      one kernel per shape-pair type over the candidate lists
//   k_count_contacts / k_setup_pairs / k_setup_terrain
//                      Manifold::from + ContactConstraint::new (manifold.rs:120-148, solver.rs:101-191)
//   k_chain_rows       order-preserving dependency links of the tick's constraint list (compact arrays, ConsLinks);
//   k_adj_fill / k_chain  the same for a caller-supplied list in any order (mgf_world_set_constraints)
//   k_solve_flow5      ContactConstraint::solve (solver.rs:203-252) for a whole Solver::solve call: block-local persistent
//                      dataflow launch (a spatial block's velocities, arrival counters and ready queues in LDS)
//   k_solve_flow       the same graph with every hand-off through L2 (stand-by of k_solve_flow5, solver mode 1)
//   k_frontier0 / k_solve
//                      one launch per frontier of the unrolled (iterations x constraints) graph (solver mode 0, cross-check)
//   k_compound_* / k_bvh_raytrace / k_intersections_batch
//                      Compound (compound.rs:230-352), BVH::raytrace and Intersects (bvh.rs:345-369, collision.rs:169-373)
//
// All f32 arithmetic follows the reference's operation order; the TU is built with
// -ffp-contract=off.
#pragma once
#include <stddef.h>

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
  float4* tb_c;   // tight swept AABB centre / half extents
  float4* tb_r;
  float4* fb_c;   // fat AABB (persistent; world.rs:181,237)
  float4* fb_r;
};

struct SceneBounds { int lo[3]; int hi[3]; uint32_t n_refits; uint32_t pad; int rmax[3]; uint32_t pad2; };  // ordered-int encoded floats; rmax = largest fat half extent

__device__ __forceinline__ int f_ord(float f) { int i = __builtin_bit_cast(int, f); return i >= 0 ? i : (i ^ 0x7FFFFFFF); }
__host__ __device__ __forceinline__ float ord_f(int i) { int j = i >= 0 ? i : (i ^ 0x7FFFFFFF); return __builtin_bit_cast(float, j); }

__device__ __forceinline__ M3 load_imb(const float4* imb, uint32_t i) {
  float4 a = imb[3 * i], b = imb[3 * i + 1], c = imb[3 * i + 2];
  return m3_cols(xyz(a), xyz(b), xyz(c));
}

// ------------------------------------------------------------------------------------------
// complete_motion + integrate, one pass.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_integrate(Bodies B, uint32_t n, float dt, float fat_margin, int do_complete,
                                                      int do_integrate, SceneBounds* sb) {
  uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  bool live = i < n;
  bool refit = false;
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
      Comp col = construct(kind, ct.y, ct.z, x, q);
      V3 d = v * dt;
      B.q[i] = make_float4(q.s, q.v.x, q.v.y, q.v.z);
      B.srec[4 * i] = make_float4(v.x, v.y, v.z, w.x);
      B.srec[4 * i + 1] = make_float4(w.y, w.z, inv_mass, I.c[0].x);
      B.srec[4 * i + 2] = make_float4(I.c[0].y, I.c[0].z, I.c[1].x, I.c[1].y);
      B.srec[4 * i + 3] = make_float4(I.c[1].z, I.c[2].x, I.c[2].y, I.c[2].z);
      B.delta[i] = mk4(d, p1.w);
      B.einfo[i] = mk4(x + d, p0.w);
      B.col0[i] = mk4(col.p, col.r);
      B.col1[i] = mk4(col.d, u2f((uint32_t)col.kind));
      Box tb = swept_bounds(col, d);
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
    } else if (do_complete) {
      B.einfo[i] = mk4(x + xyz(dl), B.einfo[i].w);
    }
    if (do_complete) B.x[i] = mk4(x, 0.0f);
  }
  if (!do_integrate || sb == nullptr) return;
  // refit count: one atomic per block
  int nref = __syncthreads_count(refit ? 1 : 0);
  if (threadIdx.x == 0 && nref) atomicAdd(&sb->n_refits, (uint32_t)nref);
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
__global__ void k_reset_step(SceneBounds* sb, uint32_t* err) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    for (int k = 0; k < 3; ++k) { sb->lo[k] = 0x7FFFFFFF; sb->hi[k] = (int)0x80000000; }
    sb->n_refits = 0; sb->pad = 0; sb->pad2 = 0;
    for (int k = 0; k < 3; ++k) sb->rmax[k] = 0;
    err[0] = 0; err[1] = 0;  // traversal stack overflow, candidate row overflow
  }
}

// ------------------------------------------------------------------------------------------
// Linear BVH over the fat AABBs.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t expand10(uint32_t v) {
  v = (v * 0x00010001u) & 0xFF0000FFu;
  v = (v * 0x00000101u) & 0x0F00F00Fu;
  v = (v * 0x00000011u) & 0xC30C30C3u;
  v = (v * 0x00000005u) & 0x49249249u;
  return v;
}
// 10-bit coordinate of the Morton code: monotone in v (the grid broadphase relies on that)
__device__ __forceinline__ uint32_t morton_quant(float v, float lo, float hi) {
  float ext = hi - lo;
  float t = ext > 0.0f ? (v - lo) / ext : 0.0f;
  int qv = (int)(t * 1023.0f);
  return (uint32_t)(qv < 0 ? 0 : (qv > 1023 ? 1023 : qv));
}
// Counting sort of the bodies into Morton cells (a cell = one 2L-bit prefix of the 30-bit code): cell of every
// body + its arrival rank inside the cell.  After a scan of the per-cell counts k_scatter_leaves places body i
// at cell_lo[cell] + rank.  The order INSIDE a cell is arrival order (it varies from run to run); nothing
// downstream depends on it - candidate rows are sorted by body index before they are used.
__global__ __launch_bounds__(kBlock) void k_morton_count(const float4* fb_c, uint32_t n, const SceneBounds* sb, int shift, uint32_t* cell_of,
                                                         uint32_t* rank, uint32_t* cell_cnt) {
  uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  V3 c = xyz(fb_c[i]);
  uint32_t code = 0;
  for (int k = 0; k < 3; ++k) code |= expand10(morton_quant(at(c, k), ord_f(sb->lo[k]), ord_f(sb->hi[k]))) << (2 - k);
  uint32_t cell = code >> shift;
  cell_of[i] = cell;
  rank[i] = atomicAdd(&cell_cnt[cell], 1u);
}

// Zero several small arrays with one launch (instead of one fill kernel each).
struct ZeroList { uint32_t* p[8]; uint32_t words[8]; };
__global__ __launch_bounds__(kBlock) void k_zero_many(ZeroList z) {
  for (int a = 0; a < 8; ++a) {
    uint32_t* p = z.p[a];
    if (!p) continue;
    for (uint32_t e = blockIdx.x * kBlock + threadIdx.x; e < z.words[a]; e += gridDim.x * kBlock) p[e] = 0u;
  }
}

// Linear BVH as an implicit complete 4-ary tree over MORTON CELLS.  A leaf is the cell of one 2L-bit
// Morton prefix (an axis-aligned region of the scene) and owns the contiguous range of sorted bodies
// whose key has that prefix; internal nodes are shorter prefixes, so every node is a spatial region
// by construction and its box (union of the contained fat boxes) stays compact however the bodies
// move.  An internal node stores the boxes of its four children (128 bytes: one fetch decides four
// subtrees); last-level nodes also carry their children's body ranges in the .w words.  Level l
// holds 4^l nodes at heap offset (4^l - 1) / 3; node k's children are 4k+1 .. 4k+4.  Built by plain
// reductions (no atomics, no fences); traversed with a register-only bitmask trail; top levels in LDS.
struct QNode { float4 lo[4], hi[4]; };     // child c: min = lo[c].xyz, max = hi[c].xyz; last level: lo.w = first body, hi.w = end
struct LeafRec { float4 c, r; };           // fat box centre | body index, half extents (sorted order)
struct Lbvh {
  QNode* nodes;        // (4^levels - 1) / 3 internal nodes
  LeafRec* leaves;     // n records in Morton order
  uint32_t* sidx;      // body index of every leaf record
  float4* lcol;        // optional, 2 per leaf record: collider (p.xyz, r) and motion (delta.xyz) of the body (k_pair_grid<true>)
  uint32_t* cell_lo;   // 4^levels + 1 entries: cell c holds the leaf records [cell_lo[c], cell_lo[c + 1])
  uint32_t n;          // live bodies
  uint32_t levels;     // internal levels L >= 4; 4^L leaf cells
  uint32_t* err;
  unsigned long long* dbg;  // optional: [0] node fetches, [1] leaf records tested, [2] max fetches of one query
};
constexpr int kLdsQNodes = 341;  // levels 0..4 (1 + 4 + 16 + 64 + 256 nodes), 43 KB
constexpr int kMortonBits = 30;
__host__ __device__ __forceinline__ uint32_t qlevel_offset(uint32_t l) { return ((1u << (2 * l)) - 1u) / 3u; }

__device__ __forceinline__ void box_min_max(V3& lo, V3& hi, V3 l, V3 h) {
  lo = mk3(fminf(lo.x, l.x), fminf(lo.y, l.y), fminf(lo.z, l.z));
  hi = mk3(fmaxf(hi.x, h.x), fmaxf(hi.y, h.y), fmaxf(hi.z, h.z));
}

// Bodies -> leaf records in cell order (counting sort, second half).
__global__ __launch_bounds__(kBlock) void k_scatter_leaves(Lbvh T, const float4* fb_c, const float4* fb_r, const uint32_t* cell_of,
                                                           const uint32_t* rank, uint32_t* brank, const float4* col0, const float4* delta) {
  uint32_t body = blockIdx.x * kBlock + threadIdx.x;
  if (body >= T.n) return;
  uint32_t p = T.cell_lo[cell_of[body]] + rank[body];
  LeafRec lr; lr.c = mk4(xyz(fb_c[body]), u2f(body)); lr.r = mk4(xyz(fb_r[body]), 0.0f);
  T.leaves[p] = lr;
  if (T.lcol) { T.lcol[2 * p] = col0[body]; T.lcol[2 * p + 1] = delta[body]; }
  T.sidx[p] = body;
  brank[body] = p;  // position in cell order (the block-local solver groups bodies by it)
}

// One block per 256 consecutive cells: the 4 internal levels above them.
// Block b owns the subtree rooted at level L-4, index b; its union box goes to sub_lo/sub_hi[b].
__global__ __launch_bounds__(kBlock) void k_lbvh_low(Lbvh T, float4* sub_lo, float4* sub_hi) {
  __shared__ float s_lo[3][kBlock], s_hi[3][kBlock];
  __shared__ uint32_t s_rng[2][kBlock];
  const int t = threadIdx.x;
  uint32_t g = blockIdx.x * kBlock + t;
  V3 lo = mk3(kInf, kInf, kInf), hi = mk3(-kInf, -kInf, -kInf);
  uint32_t b0 = T.cell_lo[g], b1 = T.cell_lo[g + 1];
  for (uint32_t p = b0; p < b1; ++p) {
    LeafRec lr = T.leaves[p];
    box_min_max(lo, hi, xyz(lr.c) - xyz(lr.r), xyz(lr.c) + xyz(lr.r));
  }
  s_lo[0][t] = lo.x; s_lo[1][t] = lo.y; s_lo[2][t] = lo.z;
  s_hi[0][t] = hi.x; s_hi[1][t] = hi.y; s_hi[2][t] = hi.z;
  s_rng[0][t] = b0; s_rng[1][t] = b1;
  __syncthreads();
  // widths 64, 16, 4, 1 at levels L-1 .. L-4
  uint32_t lvl = T.levels;
  uint32_t first = blockIdx.x * kBlock;  // index of this block's first entry within the level below
  for (int w = kBlock / 4; w >= 1; w >>= 2) {
    lvl -= 1;
    first >>= 2;
    QNode nd;
    V3 ulo = mk3(kInf, kInf, kInf), uhi = mk3(-kInf, -kInf, -kInf);
    if (t < w) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        V3 l = mk3(s_lo[0][4 * t + c], s_lo[1][4 * t + c], s_lo[2][4 * t + c]);
        V3 h = mk3(s_hi[0][4 * t + c], s_hi[1][4 * t + c], s_hi[2][4 * t + c]);
        bool leaf_level = (w == kBlock / 4);
        nd.lo[c] = mk4(l, leaf_level ? u2f(s_rng[0][4 * t + c]) : 0.0f);
        nd.hi[c] = mk4(h, leaf_level ? u2f(s_rng[1][4 * t + c]) : 0.0f);
        box_min_max(ulo, uhi, l, h);
      }
      T.nodes[qlevel_offset(lvl) + first + t] = nd;
    }
    __syncthreads();
    if (t < w) {
      s_lo[0][t] = ulo.x; s_lo[1][t] = ulo.y; s_lo[2][t] = ulo.z;
      s_hi[0][t] = uhi.x; s_hi[1][t] = uhi.y; s_hi[2][t] = uhi.z;
    }
    __syncthreads();
  }
  if (t == 0) {
    sub_lo[blockIdx.x] = make_float4(s_lo[0][0], s_lo[1][0], s_lo[2][0], 0.0f);
    sub_hi[blockIdx.x] = make_float4(s_hi[0][0], s_hi[1][0], s_hi[2][0], 0.0f);
  }
}
// Single block: levels L-5 .. 0 above the per-block subtree roots (4^(L-4) of them), ping-ponging the
// per-node union boxes between two scratch arrays.
__global__ __launch_bounds__(1024) void k_lbvh_top(Lbvh T, float4* a_lo, float4* a_hi, float4* b_lo, float4* b_hi) {
  uint32_t m = 1u << (2 * (T.levels - 4));  // entries in a_lo/a_hi
  for (int lvl = (int)T.levels - 5; lvl >= 0; --lvl) {
    uint32_t w = m >> 2;
    for (uint32_t e = threadIdx.x; e < w; e += blockDim.x) {
      QNode nd;
      V3 ulo = mk3(kInf, kInf, kInf), uhi = mk3(-kInf, -kInf, -kInf);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float4 l = a_lo[4 * e + c], h = a_hi[4 * e + c];
        nd.lo[c] = l; nd.hi[c] = h;
        box_min_max(ulo, uhi, xyz(l), xyz(h));
      }
      T.nodes[qlevel_offset((uint32_t)lvl) + e] = nd;
      b_lo[e] = mk4(ulo, 0.0f); b_hi[e] = mk4(uhi, 0.0f);
    }
    __syncthreads();
    float4* t;
    t = a_lo; a_lo = b_lo; b_lo = t;
    t = a_hi; a_hi = b_hi; b_hi = t;
    m = w;
  }
}

// ------------------------------------------------------------------------------------------
// Candidate generation: for body i, terrain faces (mesh BVH, reference DFS order) and partner
// bodies j < i whose fat AABB overlaps i's tight swept AABB (world.rs:240-290).
// ------------------------------------------------------------------------------------------
struct TerrainDev {
  const DevNode* nodes;   // flattened reference-faithful mesh BVH (host_bvh.h)
  const float4* verts;    // mesh.verts
  const uint4* faces;     // mesh.faces (a, b, c, -)
  uint32_t root;
  uint32_t n_nodes;       // 0 = no terrain
  float x[3];             // mesh.x
  uint32_t* err;          // set to 1 if a traversal stack overflows
};

constexpr int kStack = 32;  // reference-built trees are AVL-balanced: depth <= 1.44 log2(faces)

// bvh.rs:283-310 with the reference's order: push lchild, push rchild, pop rchild first.
template <class F>
__device__ __forceinline__ void terrain_traverse(const TerrainDev& M, const Box& q, F&& emit) {
  if (M.n_nodes == 0) return;
  uint32_t stack[kStack];
  int sp = 0;
  stack[sp++] = M.root;
  while (sp > 0) {
    uint32_t top = stack[--sp];
    const float4* raw = reinterpret_cast<const float4*>(&M.nodes[top]);
    float4 n0 = raw[0], n1 = raw[1];
    Box nb; nb.c = xyz(n0); nb.r = xyz(n1);
    if (box_overlaps(q, nb)) {
      uint32_t w0 = f2u(n0.w), w1 = f2u(n1.w);
      if (w0 & 0x80000000u) emit(w0 & 0x7FFFFFFFu);
      else if (sp + 2 <= kStack) { stack[sp++] = w0; stack[sp++] = w1; }
      else if (M.err) *M.err = 1u;
    }
  }
}

// Depth-first traversal of the implicit 4-ary tree with a bitmask trail (4 pending-child bits per level)
// instead of a stack.  `top` = LDS copy of nodes [0, kLdsQNodes).
template <class F>
__device__ __forceinline__ void lbvh_traverse(const Lbvh& T, const QNode* top, uint32_t i, const Box& q, float pad_abs, F&& emit) {
  if (T.n < 2) return;  // a single body has no partner
  // Inner nodes hold min/max unions: test them against a query padded well past f32 rounding so the
  // exact (centre, half-extent) acceptance test at the leaves is never pre-empted.
  float pad = pad_abs + 1e-5f * (fabs_rs(q.c.x) + fabs_rs(q.c.y) + fabs_rs(q.c.z) + q.r.x + q.r.y + q.r.z);
  V3 qlo = q.c - q.r - mk3(pad, pad, pad), qhi = q.c + q.r + mk3(pad, pad, pad);
  const int last = (int)T.levels - 1;
  uint64_t trail = 0;
  uint32_t k = 0;
  int lvl = 0;
  bool fresh = true;
  uint32_t dbg_nodes = 0, dbg_leaves = 0;
  for (;;) {
    uint32_t m;
    if (fresh) {
      ++dbg_nodes;
      const QNode* nd = (k < (uint32_t)kLdsQNodes) ? &top[k] : &T.nodes[k];
      m = 0;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float4 lo = nd->lo[c], hi = nd->hi[c];
        bool ov = qlo.x <= hi.x && lo.x <= qhi.x && qlo.y <= hi.y && lo.y <= qhi.y && qlo.z <= hi.z && lo.z <= qhi.z;
        m |= ov ? (1u << c) : 0u;
      }
    } else {
      m = (uint32_t)(trail >> (4 * lvl)) & 15u;
    }
    if (m) {
      int c = __builtin_ctz(m);
      m &= m - 1;
      trail = (trail & ~(15ull << (4 * lvl))) | ((uint64_t)m << (4 * lvl));
      if (lvl == last) {
        // child c is a Morton cell: its body range rides in the node's .w words
        const QNode* nd = (k < (uint32_t)kLdsQNodes) ? &top[k] : &T.nodes[k];
        uint32_t p0 = f2u(nd->lo[c].w), p1 = f2u(nd->hi[c].w);
        dbg_leaves += p1 - p0;
        for (uint32_t pb = p0; pb < p1; pb += 4) {
          LeafRec lr[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) lr[e] = T.leaves[min(pb + e, p1 - 1)];  // independent loads in flight together
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            uint32_t j = f2u(lr[e].c.w);
            if (pb + e < p1 && j < i) {  // world.rs:266
              Box fb; fb.c = xyz(lr[e].c); fb.r = xyz(lr[e].r);
              if (box_overlaps(q, fb)) emit(j);  // the reference's own acceptance test (bvh.rs:297)
            }
          }
        }
        fresh = false;
        continue;
      }
      k = 4 * k + 1 + (uint32_t)c;
      ++lvl;
      fresh = true;
      continue;
    }
    if (lvl == 0) break;
    k = (k - 1) >> 2;
    --lvl;
    fresh = false;
  }
  if (T.dbg) {
    atomicAdd(&T.dbg[0], (unsigned long long)dbg_nodes);
    atomicAdd(&T.dbg[1], (unsigned long long)dbg_leaves);
    atomicMax(&T.dbg[2], (unsigned long long)dbg_nodes);
  }
}

// Sizes of the tick's variable-length lists, kept on the device so that the host can enqueue the whole
// tick without reading them back: buffers and grids are sized from host-side capacities (last tick's
// sizes plus slack), kernels take the real sizes from here.  If a capacity turns out too small the
// effective sizes become 0 (every later kernel of the tick is a no-op), `fail` says why, and the host
// grows the buffers and re-runs the collide phase.
struct StepCounts {
  uint32_t Mt, Mp, C, Ct;                      // effective: terrain / pair candidates, constraints, terrain constraints
  uint32_t fail;                               // kFail* bits
  uint32_t need_Mt, need_Mp, need_C, need_Ct;  // actual sizes (valid up to the first failing stage)
  uint32_t bins[6];                            // candidates per shape-pair type (scenes mixing spheres and capsules)
  uint32_t ct_sum;                             // terrain constraints, accumulated by k_count_contacts (zeroed by k_caps_candidates)
};
constexpr uint32_t kFailCandCap = 1u, kFailConsCap = 2u, kFailRowOverflow = 4u, kFailGridWide = 8u, kFailTerrainRow = 16u, kFailTerrainWide = 32u,
                   kFailRevRow = 64u;  // a body's row of `b` occurrences overflowed (k_setup_pairs / k_chain_rows)

__global__ void k_caps_candidates(const uint32_t* mt, const uint32_t* mp, uint32_t cap_t, uint32_t cap_p, const uint32_t* row_overflow,
                                  const uint32_t* grid_wide, const uint32_t* terrain_wide, StepCounts* sc) {
  StepCounts r;
  r.need_Mt = *mt; r.need_Mp = *mp; r.need_C = 0; r.need_Ct = 0;
  r.fail = 0;
  if (r.need_Mt > cap_t || r.need_Mp > cap_p) r.fail |= kFailCandCap;
  if (row_overflow && (*row_overflow & 1u)) r.fail |= kFailRowOverflow;
  if (row_overflow && (*row_overflow & 2u)) r.fail |= kFailTerrainRow;
  if (grid_wide && *grid_wide) r.fail |= kFailGridWide;
  if (terrain_wide && *terrain_wide) r.fail |= kFailTerrainWide;
  r.Mt = r.fail ? 0u : r.need_Mt; r.Mp = r.fail ? 0u : r.need_Mp; r.C = 0; r.Ct = 0;
  for (int k = 0; k < 6; ++k) r.bins[k] = 0;
  r.ct_sum = 0;
  *sc = r;
}
__global__ void k_caps_constraints(const uint32_t* c, const uint32_t* ct, uint32_t cap_c, StepCounts* sc) {
  if (sc->fail) return;
  sc->need_C = *c; sc->need_Ct = *ct;
  if (*c > cap_c) { sc->fail |= kFailConsCap; sc->Mt = 0; sc->Mp = 0; sc->C = 0; sc->Ct = 0; return; }
  sc->C = *c; sc->Ct = *ct;
}

// XCD-aware query mapping: workgroup b is observed to run on XCD b % 8, each with a private 4 MB L2.
// Give XCD x the x-th contiguous eighth of the Morton-ordered queries, so the part of the tree it
// walks (a spatial eighth of the scene) stays resident in its own L2.  Launch xcd_grid(n) blocks.
__host__ __device__ __forceinline__ uint32_t xcd_blocks_per(uint32_t n) { return ((n + kBlock - 1) / kBlock + 7) / 8; }
__device__ __forceinline__ uint32_t xcd_logical_block() { return (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3); }

// FILL = false: count hits per body.  FILL = true: write them (CSR), partners sorted ascending.
// Bodies [n_owned, n) are ghosts (copies of a neighbouring tile's bodies): they query the tree like
// any body, but their terrain contacts and ghost-ghost pairs belong to their owner tile.
template <bool FILL>
__global__ __launch_bounds__(kBlock) void k_candidates(Bodies B, uint32_t n, uint32_t n_owned, Lbvh T, TerrainDev M, float pad,
                                                       uint32_t* t_cnt, uint32_t* p_cnt, const uint32_t* t_off,
                                                       const uint32_t* p_off, uint32_t* t_cand, uint32_t* t_owner,
                                                       uint32_t* p_cand, uint32_t* p_owner, const StepCounts* sc) {
  __shared__ QNode s_top[kLdsQNodes];
  if (FILL && sc->fail) return;
  {
    uint32_t total = qlevel_offset(T.levels);
    uint32_t lim = T.n >= 2 ? min((uint32_t)kLdsQNodes, total) : 0u;
    const float4* src = reinterpret_cast<const float4*>(T.nodes);
    float4* dst = reinterpret_cast<float4*>(s_top);
    for (uint32_t e = threadIdx.x; e < lim * 8u; e += kBlock) dst[e] = src[e];
    __syncthreads();
  }
  uint32_t k = xcd_logical_block() * kBlock + threadIdx.x;
  if (k >= n) return;
  uint32_t i = T.n >= 1 ? T.sidx[k] : k;  // walk bodies in Morton order: neighbouring lanes share tree paths
  Box q; q.c = xyz(B.tb_c[i]); q.r = xyz(B.tb_r[i]);
  // terrain: Mesh::contacts queries bounds - mesh.x (mesh.rs:121)
  Box qm = q; qm.c = q.c + -mk3(M.x[0], M.x[1], M.x[2]);
  uint32_t nt = 0, np = 0;
  uint32_t tb = FILL ? t_off[i] : 0, pb = FILL ? p_off[i] : 0;
  if (i < n_owned) {
    terrain_traverse(M, qm, [&](uint32_t face) {
      if (FILL) { t_cand[tb + nt] = face; t_owner[tb + nt] = i; }
      ++nt;
    });
  }
  if (i != 0) {  // world.rs:256
    lbvh_traverse(T, s_top, i, q, pad, [&](uint32_t j) {
      if (j >= n_owned) return;  // ghost-ghost: the owners' business
      if (FILL) { p_cand[pb + np] = j; p_owner[pb + np] = i; }
      ++np;
    });
  }
  if (!FILL) { t_cnt[i] = nt; p_cnt[i] = np; return; }
  // canonical partner order: ascending j (insertion sort, segments are ~10 long)
  for (uint32_t a = 1; a < np; ++a) {
    uint32_t v = p_cand[pb + a];
    uint32_t b = a;
    while (b > 0 && p_cand[pb + b - 1] > v) { p_cand[pb + b] = p_cand[pb + b - 1]; --b; }
    p_cand[pb + b] = v;
  }
}

// Single-pass candidate generation into fixed-capacity global rows (the common case); bodies with more
// hits than a row holds raise `overflow` and the host re-runs the exact two-pass path (k_candidates).
constexpr int kRowCap = 48;   // partner row (a settled pile has bodies with > 32 fat-box neighbours)
constexpr int kRowCapT = 16;  // terrain row: initial capacity; the host doubles it (up to kRowCapTMax) when a body overflows
constexpr int kRowCapTMax = 128;

// Terrain faces per body, reference DFS order (mesh.rs:121, bvh.rs:283-310).  One lane per body.
__global__ __launch_bounds__(kBlock) void k_terrain_rows(Bodies B, uint32_t n_owned, TerrainDev M, uint32_t cap_row, uint32_t* rows_t,
                                                         uint32_t* t_cnt, uint32_t* overflow) {
  uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n_owned) return;
  Box q; q.c = xyz(B.tb_c[i]) + -mk3(M.x[0], M.x[1], M.x[2]); q.r = xyz(B.tb_r[i]);
  uint32_t* row = rows_t + (size_t)i * cap_row;
  uint32_t nt = 0;
  terrain_traverse(M, q, [&](uint32_t face) {
    if (nt < cap_row) row[nt] = face;
    ++nt;
  });
  t_cnt[i] = nt;
  if (nt > cap_row) atomicOr(overflow, 2u);  // bit 1: a terrain row, bit 0: a partner row
}

// Partner bodies per body: cooperative traversal, 8 lanes per query.  A 4-ary node is eight 16-byte
// words (lo[0..3], hi[0..3]); lane s of the group loads word s, so a node costs ONE cache-line lookup
// per query instead of eight per lane (the per-lane form is bound by L1 tag lookups once neighbouring
// queries stop walking in lock-step).  Lanes 0-3 test child s against the query (hi comes from lane
// s+4 by shuffle), a ballot yields the 4-bit child mask, and the traversal state (node, level, trail)
// is replicated in the group's lanes so its control flow stays uniform.  Leaf cells: lane pairs load
// one 32-byte record each (4 records per step).
constexpr int kCoopLanes = 8;
constexpr int kCoopBlock = 512;                 // 64 queries per block
constexpr int kCoopLdsNodes = 85;               // levels 0..3 staged in LDS (10.9 KB)
__device__ __forceinline__ uint32_t xcd_logical_block_coop() { return (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3); }

__global__ __launch_bounds__(kCoopBlock) void k_pair_rows(Bodies B, uint32_t n, uint32_t n_owned, Lbvh T, float pad_abs, uint32_t* rows_p,
                                                          uint32_t* p_cnt, uint32_t* overflow) {
  __shared__ float4 s_top[kCoopLdsNodes * 8];
  {
    uint32_t total = qlevel_offset(T.levels);
    uint32_t lim = T.n >= 2 ? min((uint32_t)kCoopLdsNodes, total) : 0u;
    const float4* src = reinterpret_cast<const float4*>(T.nodes);
    for (uint32_t e = threadIdx.x; e < lim * 8u; e += kCoopBlock) s_top[e] = src[e];
    __syncthreads();
  }
  const int lane = threadIdx.x & 63;
  const int sub = lane & 7;
  const int gbase = lane & ~7;
  uint32_t kq = xcd_logical_block_coop() * (kCoopBlock / kCoopLanes) + (threadIdx.x >> 3);
  if (kq >= n) return;  // whole group leaves together
  uint32_t i = T.sidx[kq];  // Morton order: neighbouring groups walk neighbouring subtrees
  uint32_t np = 0;
  if (i != 0 && T.n >= 2) {  // world.rs:256
    Box q; q.c = xyz(B.tb_c[i]); q.r = xyz(B.tb_r[i]);
    float pad = pad_abs + 1e-5f * (fabs_rs(q.c.x) + fabs_rs(q.c.y) + fabs_rs(q.c.z) + q.r.x + q.r.y + q.r.z);
    V3 qlo = q.c - q.r - mk3(pad, pad, pad), qhi = q.c + q.r + mk3(pad, pad, pad);
    uint32_t* row = rows_p + (size_t)i * kRowCap;
    const float4* gnodes = reinterpret_cast<const float4*>(T.nodes);
    const float4* gleaves = reinterpret_cast<const float4*>(T.leaves);
    const int last = (int)T.levels - 1;
    uint64_t trail = 0;
    uint32_t k = 0;
    int lvl = 0;
    bool fresh = true;
    float4 v = make_float4(0, 0, 0, 0);  // this lane's word of the current node
    for (;;) {
      uint32_t m;
      if (fresh) {
        v = (k < (uint32_t)kCoopLdsNodes) ? s_top[k * 8 + sub] : gnodes[(size_t)k * 8 + sub];
        // lanes 0-3: lo[sub]; their hi[sub] sits in lane sub + 4
        float hx = __shfl(v.x, gbase + (sub & 3) + 4), hy = __shfl(v.y, gbase + (sub & 3) + 4), hz = __shfl(v.z, gbase + (sub & 3) + 4);
        bool ov = sub < 4 && qlo.x <= hx && v.x <= qhi.x && qlo.y <= hy && v.y <= qhi.y && qlo.z <= hz && v.z <= qhi.z;
        unsigned long long bal = __ballot(ov);
        m = (uint32_t)(bal >> gbase) & 15u;
      } else {
        m = (uint32_t)(trail >> (4 * lvl)) & 15u;
      }
      if (m) {
        int c = __builtin_ctz(m);
        m &= m - 1;
        trail = (trail & ~(15ull << (4 * lvl))) | ((uint64_t)m << (4 * lvl));
        if (lvl == last) {
          // child c is a Morton cell; its body range rides in the .w words of lo[c] / hi[c]
          uint32_t p0 = f2u(__shfl(v.w, gbase + c)), p1 = f2u(__shfl(v.w, gbase + c + 4));
          for (uint32_t pb = p0; pb < p1; pb += 4) {
            uint32_t rec = min(pb + (uint32_t)(sub >> 1), p1 - 1);
            float4 w = gleaves[(size_t)rec * 2 + (sub & 1)];  // even lane: centre | body, odd lane: half extents
            float rx = __shfl(w.x, lane | 1), ry = __shfl(w.y, lane | 1), rz = __shfl(w.z, lane | 1);
            uint32_t j = f2u(w.w);
            bool hit = false;
            if (!(sub & 1) && pb + (uint32_t)(sub >> 1) < p1 && j < i && j < n_owned) {  // world.rs:266; ghost-ghost skipped
              Box fb; fb.c = xyz(w); fb.r = mk3(rx, ry, rz);
              hit = box_overlaps(q, fb);  // the reference's own acceptance test (bvh.rs:297)
            }
            unsigned long long hb = __ballot(hit);
            uint32_t gm = (uint32_t)(hb >> gbase) & 255u;
            if (hit) {
              uint32_t slot = np + __popc(gm & ((1u << sub) - 1u));
              if (slot < (uint32_t)kRowCap) row[slot] = j;
            }
            np += __popc(gm);
          }
          fresh = false;
          continue;
        }
        k = 4 * k + 1 + (uint32_t)c;
        ++lvl;
        fresh = true;
        continue;
      }
      if (lvl == 0) break;
      k = (k - 1) >> 2;
      --lvl;
      fresh = false;
    }
  }
  if (sub == 0) {
    p_cnt[i] = np;
    if (np > (uint32_t)kRowCap) atomicOr(overflow, 1u);
  }
}

__device__ __forceinline__ Comp load_comp(const Bodies& B, uint32_t i) {
  float4 c0 = B.col0[i], c1 = B.col1[i];
  Comp k; k.p = xyz(c0); k.r = c0.w; k.d = xyz(c1); k.kind = (int)f2u(c1.w);
  return k;
}

// Partner bodies per body without a tree walk.  The leaf level of the Morton-cell tree IS a uniform grid: cell
// (cx, cy, cz) is the 2L-bit Morton prefix of its interleaved coordinates, and cell_lo gives its bodies.
// A body j can only be accepted by query i (tight_i overlaps fat_j) if its fat-box centre lies within
// tight_i grown by the largest fat half extent of the scene (SceneBounds::rmax), so the query enumerates the
// cells of that region directly: ~50 independent 8-byte look-ups and as many independent leaf records, two
// dependent memory round trips instead of the ~30 of the top-down walk.  8 lanes share a query, one cell per
// lane per round.  Scenes whose largest body spans many cells raise `too_wide` and the host switches to the
// tree walk (k_pair_rows) - the accepted set is the same either way (the reference's predicate on the leaf
// records).
// SPHERES (a world of spheres only): an accepted partner goes straight through the sphere-sphere narrowphase test
// (the same function k_narrow_pairs runs) and only contacts are written to the row - a dense pile accepts ~11 partners
// per body by their fat boxes and keeps ~2, so everything downstream of the rows handles a sixth of the entries.  The
// accepted partners are still counted (World::step's candidate statistic): per wave into one of 64 words of pair_stat.
constexpr uint32_t kGridMaxCells = 512;
template <bool SPHERES>
__global__ __launch_bounds__(kCoopBlock) void k_pair_grid(Bodies B, uint32_t n, uint32_t n_owned, Lbvh T, const SceneBounds* sb, float pad_abs,
                                                          uint32_t* rows_p, uint32_t* p_cnt, uint32_t* overflow, uint32_t* too_wide,
                                                          uint32_t* pair_stat) {
  __shared__ uint32_t s_acc[SPHERES ? kCoopBlock / kCoopLanes : 1][SPHERES ? kRowCap : 1];  // accepted partners of a query (leaf positions)
  const int lane = threadIdx.x & 63;
  const int sub = lane & 7;
  const int gbase = lane & ~7;
  uint32_t kq = xcd_logical_block_coop() * (kCoopBlock / kCoopLanes) + (threadIdx.x >> 3);
  const bool live = kq < n;  // whole groups are live or not
  uint32_t i = live ? T.sidx[kq] : 0u;
  uint32_t np = 0, n_accepted = 0;
  if (live && i != 0 && T.n >= 2) {  // world.rs:256
    Box q; q.c = xyz(B.tb_c[i]); q.r = xyz(B.tb_r[i]);
    Comp A; V3 vA = mk3(0, 0, 0);
    if (SPHERES) { A = load_comp(B, i); A.kind = KIND_SPHERE; vA = xyz(B.delta[i]); }
    float pad = pad_abs + 1e-5f * (fabs_rs(q.c.x) + fabs_rs(q.c.y) + fabs_rs(q.c.z) + q.r.x + q.r.y + q.r.z);
    const uint32_t P = 2u * T.levels;
    const uint32_t nb[3] = {(P + 2u) / 3u, (P + 1u) / 3u, P / 3u};  // prefix bits per axis (x is the most significant)
    uint32_t ca[3], d[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      float lo = ord_f(sb->lo[k]), hi = ord_f(sb->hi[k]), rm = ord_f(sb->rmax[k]);
      float a = at(q.c, k) - at(q.r, k) - rm - pad, b = at(q.c, k) + at(q.r, k) + rm + pad;
      uint32_t c0 = morton_quant(a, lo, hi) >> (10u - nb[k]), c1 = morton_quant(b, lo, hi) >> (10u - nb[k]);
      ca[k] = c0; d[k] = c1 - c0 + 1u;
    }
    const uint32_t ncell = d[0] * d[1] * d[2];
    if (ncell > kGridMaxCells) {
      if (sub == 0) *too_wide = 1u;
    } else {
      uint32_t* row = rows_p + (size_t)i * kRowCap;
      const int shift = kMortonBits - (int)P;
      for (uint32_t cb = 0; cb < ncell; cb += kCoopLanes) {
        uint32_t idx = cb + (uint32_t)sub;
        uint32_t p0 = 0, p1 = 0;
        if (idx < ncell) {
          uint32_t cz = idx % d[2], t = idx / d[2];
          uint32_t cy = t % d[1], cx = t / d[1];
          uint32_t code = (expand10((ca[0] + cx) << (10u - nb[0])) << 2) | (expand10((ca[1] + cy) << (10u - nb[1])) << 1) |
                          expand10((ca[2] + cz) << (10u - nb[2]));
          uint32_t cell = code >> shift;
          p0 = T.cell_lo[cell]; p1 = T.cell_lo[cell + 1];
        }
        // every lane walks its own cell's bodies; the group stays together for the ballots
        for (;;) {
          bool more = p0 < p1;
          unsigned long long mb = __ballot(more);
          if (((uint32_t)(mb >> gbase) & 255u) == 0u) break;
          bool hit = false;
          uint32_t j = 0;
          if (more) {
            LeafRec lr = T.leaves[p0];
            j = f2u(lr.c.w);
            if (j < i && j < n_owned) {  // world.rs:266; ghost-ghost skipped
              Box fb; fb.c = xyz(lr.c); fb.r = xyz(lr.r);
              hit = box_overlaps(q, fb);  // the reference's own acceptance test (bvh.rs:297)
            }
            if (SPHERES) j = p0;  // the row holds leaf positions until the second phase below
            ++p0;
          }
          unsigned long long hb = __ballot(hit);
          uint32_t gm = (uint32_t)(hb >> gbase) & 255u;
          if (hit) {
            uint32_t slot = np + __popc(gm & ((1u << sub) - 1u));
            if (slot < (uint32_t)kRowCap) {
              if (SPHERES) s_acc[threadIdx.x >> 3][slot] = j;
              else row[slot] = j;
            }
          }
          np += __popc(gm);
        }
      }
      if (SPHERES) {
        // second phase: the accepted partners (staged in LDS), sixteen at a time - two per lane, so a typical query
        // needs one round trip for its partners' records - through the sphere-sphere test; contacts go to the row
        n_accepted = np;
        if (np > (uint32_t)kRowCap) atomicOr(overflow, 1u);
        const uint32_t na = min(np, (uint32_t)kRowCap);
        const uint32_t* acc = s_acc[threadIdx.x >> 3];
        uint32_t nc = 0;
        for (uint32_t a0 = 0; a0 < na; a0 += 2 * kCoopLanes) {
          bool hit[2] = {false, false};
          uint32_t jj[2] = {0, 0};
          float4 c0[2], d0[2];
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const uint32_t a = a0 + (uint32_t)u * kCoopLanes + (uint32_t)sub;
            const uint32_t pj = a < na ? acc[a] : 0u;
            c0[u] = T.lcol[2 * pj]; d0[u] = T.lcol[2 * pj + 1];
            jj[u] = T.sidx[pj];
          }
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const uint32_t a = a0 + (uint32_t)u * kCoopLanes + (uint32_t)sub;
            if (a < na) {
              // cheap and conservative first: the centres never come closer than |d| - |v| during the tick
              const V3 d = xyz(c0[u]) - A.p, v = xyz(d0[u]) - vA;
              const float lim = A.r + c0[u].w + __builtin_sqrtf(dot(v, v));
              if (dot(d, d) <= lim * lim * 1.001f) {
                Comp Bc; Bc.kind = KIND_SPHERE; Bc.p = xyz(c0[u]); Bc.r = c0[u].w; Bc.d = mk3(0, 0, 0);
                LocalContact lc;
                hit[u] = comp_pair_local(A, vA, Bc, xyz(d0[u]), &lc);
              }
            }
          }
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const uint32_t gm = (uint32_t)(__ballot(hit[u]) >> gbase) & 255u;
            if (hit[u]) row[nc + __popc(gm & ((1u << sub) - 1u))] = jj[u];
            nc += __popc(gm);
          }
        }
        np = nc;
      }
    }
  }
  if (live && sub == 0) {
    p_cnt[i] = np;
    if (!SPHERES && np > (uint32_t)kRowCap) atomicOr(overflow, 1u);
  }
  if (SPHERES) {  // accepted partners of the wave's 8 queries -> one atomic
    uint32_t v = (live && sub == 0) ? n_accepted : 0u;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    if (lane == 0 && v) atomicAdd(&pair_stat[(blockIdx.x * (kCoopBlock / 64) + (threadIdx.x >> 6)) & 63u], v);
  }
}

// Terrain faces per body without walking the reference tree.  A static mesh gets the same Morton-cell grid as the
// bodies (cells over the face boxes, built once per set_terrain); a query enumerates the cells its box can reach,
// applies Mesh::contacts' own acceptance test (query overlaps the face's leaf bounds, bvh.rs:297) and - because the
// reference only reaches a leaf through its ancestors - re-checks the ancestors' boxes for hits that are within
// rounding distance of not overlapping (an ancestor box is the union of its children up to f32 rounding, so a clear
// overlap with the leaf implies an overlap with every ancestor).  Hits are stored as DFS RANKS: BVH::query reports
// leaves in one fixed order whatever it prunes (HostBvh::dfs_ranks), so sorting a body's row by rank restores the
// reference's callback order.  Meshes whose faces span many cells raise `too_wide`; the host then uses the tree walk.
struct FaceGrid {
  Lbvh T;                      // cells over the face boxes: leaves[].c.w = face id
  const SceneBounds* sb;
  const uint32_t* rank_of_face;
  const uint32_t* leaf_of_face;  // node id of the face's leaf in the reference tree
  const uint32_t* parent;        // per node of the reference tree
};
__global__ __launch_bounds__(kCoopBlock) void k_terrain_grid(Bodies B, uint32_t n_owned, const uint32_t* order, TerrainDev M, FaceGrid G,
                                                             float pad_abs, uint32_t cap_row, uint32_t* rows_t, uint32_t* t_cnt,
                                                             uint32_t* overflow, uint32_t* too_wide) {
  const int lane = threadIdx.x & 63;
  const int sub = lane & 7;
  const int gbase = lane & ~7;
  uint32_t kq = xcd_logical_block_coop() * (kCoopBlock / kCoopLanes) + (threadIdx.x >> 3);
  if (kq >= n_owned) return;  // whole group leaves together
  uint32_t i = order ? order[kq] : kq;
  if (i >= n_owned) {  // cell order runs over owned + ghost bodies: ghosts have no terrain row
    return;
  }
  Box q; q.c = xyz(B.tb_c[i]) + -mk3(M.x[0], M.x[1], M.x[2]); q.r = xyz(B.tb_r[i]);
  float mag = fabs_rs(q.c.x) + fabs_rs(q.c.y) + fabs_rs(q.c.z) + q.r.x + q.r.y + q.r.z;
  float pad = pad_abs + 1e-5f * mag;
  const uint32_t P = 2u * G.T.levels;
  const uint32_t nb[3] = {(P + 2u) / 3u, (P + 1u) / 3u, P / 3u};
  uint32_t ca[3], d[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    float lo = ord_f(G.sb->lo[k]), hi = ord_f(G.sb->hi[k]), rm = ord_f(G.sb->rmax[k]);
    float a = at(q.c, k) - at(q.r, k) - rm - pad, b = at(q.c, k) + at(q.r, k) + rm + pad;
    uint32_t c0 = morton_quant(a, lo, hi) >> (10u - nb[k]), c1 = morton_quant(b, lo, hi) >> (10u - nb[k]);
    ca[k] = c0; d[k] = c1 - c0 + 1u;
  }
  const uint32_t ncell = d[0] * d[1] * d[2];
  uint32_t nt = 0;
  if (ncell > kGridMaxCells) {
    if (sub == 0) *too_wide = 1u;
  } else {
    uint32_t* row = rows_t + (size_t)i * cap_row;
    const int shift = kMortonBits - (int)P;
    for (uint32_t cb = 0; cb < ncell; cb += kCoopLanes) {
      uint32_t idx = cb + (uint32_t)sub;
      uint32_t p0 = 0, p1 = 0;
      if (idx < ncell) {
        uint32_t cz = idx % d[2], t = idx / d[2];
        uint32_t cy = t % d[1], cx = t / d[1];
        uint32_t code = (expand10((ca[0] + cx) << (10u - nb[0])) << 2) | (expand10((ca[1] + cy) << (10u - nb[1])) << 1) |
                        expand10((ca[2] + cz) << (10u - nb[2]));
        uint32_t cell = code >> shift;
        p0 = G.T.cell_lo[cell]; p1 = G.T.cell_lo[cell + 1];
      }
      for (;;) {
        bool more = p0 < p1;
        unsigned long long mb = __ballot(more);
        if (((uint32_t)(mb >> gbase) & 255u) == 0u) break;
        bool hit = false;
        uint32_t rank = 0;
        if (more) {
          LeafRec lr = G.T.leaves[p0];
          uint32_t face = f2u(lr.c.w);
          Box fb; fb.c = xyz(lr.c); fb.r = xyz(lr.r);
          if (box_overlaps(q, fb)) {  // the reference's acceptance test at the leaf (bvh.rs:297)
            hit = true;
            // by how much?  a clear overlap needs no ancestor check
            float gap = fmin_rs(fmin_rs(q.r.x + fb.r.x - fabs_rs(q.c.x - fb.c.x), q.r.y + fb.r.y - fabs_rs(q.c.y - fb.c.y)),
                                q.r.z + fb.r.z - fabs_rs(q.c.z - fb.c.z));
            float tol = 1e-4f * (mag + fabs_rs(fb.c.x) + fabs_rs(fb.c.y) + fabs_rs(fb.c.z) + fb.r.x + fb.r.y + fb.r.z);
            if (!(gap > tol)) {
              uint32_t node = G.leaf_of_face[face];
              while (node != M.root) {
                node = G.parent[node];
                const float4* raw = reinterpret_cast<const float4*>(&M.nodes[node]);
                Box nbx; nbx.c = xyz(raw[0]); nbx.r = xyz(raw[1]);
                if (!box_overlaps(q, nbx)) { hit = false; break; }
              }
            }
            rank = G.rank_of_face[face];
          }
          ++p0;
        }
        unsigned long long hb = __ballot(hit);
        uint32_t gm = (uint32_t)(hb >> gbase) & 255u;
        if (hit) {
          uint32_t slot = nt + __popc(gm & ((1u << sub) - 1u));
          if (slot < cap_row) row[slot] = rank;
        }
        nt += __popc(gm);
      }
    }
  }
  if (sub == 0) {
    t_cnt[i] = nt;
    if (nt > cap_row) atomicOr(overflow, 2u);
  }
}

// rows -> CSR (terrain and partner candidate lists with their owners)
// (canonical insertion order).
__global__ __launch_bounds__(kBlock) void k_rows_to_csr(const StepCounts* sc, uint32_t n, uint32_t cap_row_t, const uint32_t* face_of_rank,
                                                        const uint32_t* rows_t, const uint32_t* rows_p,
                                                        const uint32_t* t_off, const uint32_t* p_off, uint32_t* t_cand, uint32_t* t_owner,
                                                        uint32_t* p_cand, uint32_t* p_owner) {
  uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n || sc->fail) return;
  uint32_t tb = t_off[i], nt = t_off[i + 1] - tb, pb = p_off[i], np = p_off[i + 1] - pb;
  if (nt > cap_row_t || np > (uint32_t)kRowCap) return;  // overflowed body: the host re-runs with wider rows or the two-pass path
  const uint32_t* rt = rows_t + (size_t)i * cap_row_t;
  const uint4* rp = reinterpret_cast<const uint4*>(rows_p + (size_t)i * kRowCap);  // rows are 16-byte aligned (kRowCap % 4 == 0)
  if (face_of_rank) {  // the row holds DFS ranks in discovery order: sort, then name the faces
    for (uint32_t a = 0; a < nt; ++a) {
      uint32_t x = rt[a];
      uint32_t b = a;
      while (b > 0 && t_cand[tb + b - 1] > x) { t_cand[tb + b] = t_cand[tb + b - 1]; --b; }
      t_cand[tb + b] = x;
    }
    for (uint32_t a = 0; a < nt; ++a) { t_cand[tb + a] = face_of_rank[t_cand[tb + a]]; t_owner[tb + a] = i; }
  } else {
    for (uint32_t a = 0; a < nt; ++a) { t_cand[tb + a] = rt[a]; t_owner[tb + a] = i; }
  }
  // partners stay in discovery order: only the few that turn into contacts need the canonical (ascending) order, and
  // k_count_contacts numbers those by partner id
  for (uint32_t a = 0; a < np; a += 4) {
    uint4 v = rp[a >> 2];
    uint32_t e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) if (a + k < np) { p_cand[pb + a + k] = e[k]; p_owner[pb + a + k] = i; }
  }
}

// ------------------------------------------------------------------------------------------
// Narrowphase, one kernel per shape-pair type.  Output per candidate: contact count and the
// LocalContact reduced to what Manifold/ContactConstraint::new consume (local_a, local_b, n).
// ------------------------------------------------------------------------------------------
struct NContact { float4 la, lb, n; };  // la.xyz + t, lb.xyz, n.xyz


// work = nullptr: dense over [0, m); else the m candidate ids of this pair type.
template <int KA, int KB>
__global__ __launch_bounds__(kBlock) void k_narrow_pairs(Bodies B, const uint32_t* work, const uint32_t* m_ptr, const uint32_t* p_owner,
                                                         const uint32_t* p_cand, uint32_t* p_nc, NContact* p_out) {
  uint32_t t = blockIdx.x * kBlock + threadIdx.x;
  if (t >= *m_ptr) return;
  uint32_t p = work ? work[t] : t;
  uint32_t i = p_owner[p], j = p_cand[p];
  Comp A = load_comp(B, i), Bc = load_comp(B, j);
  A.kind = KA; Bc.kind = KB;  // compile-time dispatch: the list holds only this pair type
  V3 vA = xyz(B.delta[i]), vB = xyz(B.delta[j]);
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
  Comp A = load_comp(B, i);
  A.kind = KA;
  V3 vA = xyz(B.delta[i]);
  uint4 fi = M.faces[f];
  V3 mx = mk3(M.x[0], M.x[1], M.x[2]);
  Triangle tri = mkt(xyz(M.verts[fi.x]) + mx, xyz(M.verts[fi.y]) + mx, xyz(M.verts[fi.z]) + mx);  // mesh.rs:122-126
  LocalContact lc[2];
  int nc = comp_tri_local(A, vA, tri, mx, lc);
  t_nc[p] = (uint32_t)nc;
  for (int k = 0; k < nc; ++k) {
    NContact o; o.la = mk4(lc[k].la, lc[k].g.t); o.lb = mk4(lc[k].lb, 0.0f); o.n = mk4(lc[k].g.n, 0.0f);  // Manifold::from(lc) manifold.rs:120-128
    t_out[2 * p + k] = o;
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
                                                           uint32_t* p_pre, uint32_t* cnt) {
  constexpr int kHitCap = 12;  // a sphere touches at most 12 equal ones
  __shared__ uint32_t s_j[kHitCap][kBlock], s_p[kHitCap][kBlock];
  const int tid = threadIdx.x;
  __shared__ uint32_t s_ct;
  uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (tid == 0) s_ct = 0;
  __syncthreads();
  bool active = i < n;
  if (active && sc->fail) { cnt[i] = 0; active = false; }
  if (active) {
  uint32_t run = 0;
  for (uint32_t p = t_off[i]; p < t_off[i + 1]; ++p) { t_pre[p] = run; run += t_nc[p]; }
  if (run) atomicAdd(&s_ct, run);  // only the total is needed: one global atomic per block
  // partner contacts are numbered in ascending partner order (the canonical insertion order); the candidate list itself
  // is in discovery order, and only a few of its ~10 entries are contacts (at most one per partner): collect them, then
  // rank them among themselves
  const uint32_t lo = p_off[i], hi = p_off[i + 1];
  uint32_t h = 0;
  for (uint32_t base = lo; base < hi; base += 4) {  // four counts per round trip
    uint32_t nc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) nc[k] = base + k < hi ? p_nc[base + k] : 0u;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (nc[k]) {
        if (h < (uint32_t)kHitCap) { s_j[h][tid] = p_cand[base + k]; s_p[h][tid] = base + k; }
        ++h;
      }
    }
  }
  if (h <= (uint32_t)kHitCap) {
    for (uint32_t a = 0; a < h; ++a) {
      const uint32_t j = s_j[a][tid];
      uint32_t before = 0;
      for (uint32_t q = 0; q < h; ++q) before += s_j[q][tid] < j ? 1u : 0u;
      p_pre[s_p[a][tid]] = run + before;
    }
  } else {  // a crowded body: the same by rescanning its list
    for (uint32_t p = lo; p < hi; ++p) {
      if (p_nc[p] == 0) continue;
      const uint32_t j = p_cand[p];
      uint32_t before = 0;
      for (uint32_t q = lo; q < hi; ++q) before += (p_cand[q] < j) ? p_nc[q] : 0u;
      p_pre[p] = run + before;
    }
  }
  cnt[i] = run + h;
  }
  __syncthreads();
  if (tid == 0 && s_ct) atomicAdd(&sc->ct_sum, s_ct);
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
__device__ __forceinline__ CRec make_constraint(uint32_t ia, uint32_t ib, const BodyDyn& A, V3 xa, float rest_a, float fric_a,
                                                const BodyDyn& Bd, V3 xb, float rest_b, float fric_b, V3 normal, V3 ra, V3 rb,
                                                float dt, float baumgarte, float slop) {
  CRec c;
  c.a = ia; c.b = ib;
  float restitution = fmax_rs(rest_a, rest_b);
  c.friction = __builtin_sqrtf(fric_a * fric_b);
  V3 t0, t1;
  compute_basis(normal, &t0, &t1);  // manifold.rs:125,144
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

__global__ __launch_bounds__(kBlock) void k_setup_pairs(Bodies B, const StepCounts* sc, const uint32_t* p_owner, const uint32_t* p_cand,
                                                        const uint32_t* p_nc, const uint32_t* p_pre, const NContact* p_in,
                                                        const uint32_t* base, float dt, float baumgarte, float slop,
                                                        CRec* cons, uint2* ab, uint32_t* degb, uint32_t* rev, uint32_t rev_cap,
                                                        uint32_t* rev_flag) {
  uint32_t p = blockIdx.x * kBlock + threadIdx.x;
  if (p >= sc->Mp || p_nc[p] == 0) return;
  uint32_t i = p_owner[p], j = p_cand[p];
  uint32_t c = base[i] + p_pre[p];
  NContact k = p_in[p];
  BodyDyn A = load_dyn(B.srec, i), Bd = load_dyn(B.srec, j);
  float4 ea = B.einfo[i], eb = B.einfo[j];
  CRec r = make_constraint(i, j, A, xyz(ea), ea.w, B.delta[i].w, Bd, xyz(eb), eb.w, B.delta[j].w, xyz(k.n), xyz(k.la), xyz(k.lb),
                           dt, baumgarte, slop);
  store_crec(&cons[c], r);
  ab[c] = make_uint2(i, j);
  // body j's row of the constraints it takes part in as `b` (k_chain_rows); as `a` a body owns a contiguous id range
  uint32_t pos = atomicAdd(&degb[j], 1u);
  if (pos < rev_cap) rev[(size_t)j * rev_cap + pos] = c;
  else *rev_flag = 1u;
}

__global__ __launch_bounds__(kBlock) void k_setup_terrain(Bodies B, TerrainDev M, const StepCounts* sc, const uint32_t* t_owner,
                                                          const uint32_t* t_nc, const uint32_t* t_pre, const NContact* t_in,
                                                          const uint32_t* base, float dt, float baumgarte, float slop, CRec* cons,
                                                          uint2* ab) {
  uint32_t p = blockIdx.x * kBlock + threadIdx.x;
  if (p >= sc->Mt) return;
  uint32_t nc = t_nc[p];
  if (nc == 0) return;
  uint32_t i = t_owner[p];
  BodyDyn A = load_dyn(B.srec, i), S = static_dyn();
  float4 ea = B.einfo[i];
  V3 center = mk3(M.x[0], M.x[1], M.x[2]);  // Static{ center: terrain.center(), friction: 0.0 } world.rs:247
  for (uint32_t k = 0; k < nc; ++k) {
    NContact in = t_in[2 * p + k];
    CRec r = make_constraint(i, kNone, A, xyz(ea), ea.w, B.delta[i].w, S, center, 0.0f, 0.0f, xyz(in.n), xyz(in.la), xyz(in.lb), dt,
                             baumgarte, slop);
    store_crec(&cons[base[i] + t_pre[p] + k], r);
    ab[base[i] + t_pre[p] + k] = make_uint2(i, kNone);
  }
}

// ------------------------------------------------------------------------------------------
// Dependency DAG of the insertion-ordered constraint list.  A constraint may run once the
// previous constraint touching each of its bodies has run; running all ready constraints
// together ("a level") is exactly the sequential Gauss-Seidel result (solver.rs:72-78).
// Per body: the list of constraints touching it, sorted by insertion index; consecutive entries
// are linked (succ_a / succ_b by the body's role in the earlier one).
// ------------------------------------------------------------------------------------------
// entry = (constraint id << 1) | role, role 0: the body is `a`, role 1: the body is `b`.
__global__ __launch_bounds__(kBlock) void k_adj_fill(const uint2* ab, const uint32_t* C_ptr, const uint32_t* adj_off, uint32_t* adj_fill,
                                                     uint32_t* adj_list) {
  uint32_t c = blockIdx.x * kBlock + threadIdx.x;
  if (c >= *C_ptr) return;
  uint2 e = ab[c];
  adj_list[adj_off[e.x] + atomicAdd(&adj_fill[e.x], 1u)] = (c << 1);
  if (e.y != kNone) adj_list[adj_off[e.y] + atomicAdd(&adj_fill[e.y], 1u)] = (c << 1) | 1u;
}
// (a, b) and per-body degrees of a caller-supplied constraint list (mgf_world_set_constraints); the tick's own
// list gets them from the setup kernels.
__global__ __launch_bounds__(kBlock) void k_links_from_records(const CRec* cons, uint32_t C, uint2* ab, uint32_t* deg) {
  uint32_t c = blockIdx.x * kBlock + threadIdx.x;
  if (c >= C) return;
  uint32_t a = cons[c].a, b = cons[c].b;
  ab[c] = make_uint2(a, b);
  atomicAdd(&deg[a], 1u);
  if (b != kNone) atomicAdd(&deg[b], 1u);
}

// Successor word: bits 0..29 constraint id, bit 30 = successor has two dynamic bodies (its
// per-round in-degree is 2, else 1), bit 31 = the link wraps to the next solver iteration.
constexpr uint32_t kSuccId = 0x3FFFFFFFu, kSuccTwo = 0x40000000u, kSuccWrap = 0x80000000u;

__global__ __launch_bounds__(kBlock) void k_chain(uint32_t n, ConsLinks K, const uint32_t* adj_off, uint32_t* adj_list) {
  uint32_t x = blockIdx.x * kBlock + threadIdx.x;
  if (x >= n) return;
  uint32_t lo = adj_off[x], hi = adj_off[x + 1];
  if (lo == hi) return;
  for (uint32_t a = lo + 1; a < hi; ++a) {  // ascending constraint id = insertion order
    uint32_t v = adj_list[a];
    uint32_t b = a;
    while (b > lo && adj_list[b - 1] > v) { adj_list[b] = adj_list[b - 1]; --b; }
    adj_list[b] = v;
  }
  uint32_t* succ = reinterpret_cast<uint32_t*>(K.succ);
  for (uint32_t a = lo; a < hi; ++a) {
    bool last = (a + 1 == hi);
    uint32_t u = adj_list[a], w = adj_list[last ? lo : a + 1];
    uint32_t wid = w >> 1;
    uint32_t word = wid | (K.ab[wid].y != kNone ? kSuccTwo : 0u) | (last ? kSuccWrap : 0u);
    succ[2 * (u >> 1) + (u & 1u)] = word;
    K.pred[2 * (u >> 1) + (u & 1u)] = a > lo ? 1 : 0;  // predecessor on this body inside one iteration
  }
}
__device__ __forceinline__ uint32_t links_indeg0(const ConsLinks& K, uint32_t c) {
  return (uint32_t)K.pred[2 * c] + (K.ab[c].y != kNone ? (uint32_t)K.pred[2 * c + 1] : 0u);
}

// The same links for the tick's own constraint list, without the global adjacency build.  In canonical order a body x
// is `a` exactly in the contiguous ids [base[x], base[x+1]) (its terrain contacts, then its partners j < x) and `b` only
// in constraints of bodies i > x, whose ids are all larger: its chain is the own range followed by its row of `b`
// occurrences (written by k_setup_pairs in arrival order, sorted here).  A row that overflowed raises kFailRevRow and
// empties the tick (C = 0): the host widens the rows and re-runs the collide phase.
__global__ __launch_bounds__(kBlock) void k_chain_rows(uint32_t n, ConsLinks K, const uint32_t* base, const uint32_t* degb, uint32_t* rev,
                                                       uint32_t rev_cap, const uint32_t* rev_flag, StepCounts* sc) {
  uint32_t x = blockIdx.x * kBlock + threadIdx.x;
  if (*rev_flag) {
    if (x == 0) { sc->C = 0; sc->Ct = 0; sc->fail |= kFailRevRow; }
    return;
  }
  if (x >= n) return;
  const uint32_t lo = base[x], na = base[x + 1] - lo, nb = degb[x];
  const uint32_t total = na + nb;
  if (total == 0) return;
  uint32_t* row = rev + (size_t)x * rev_cap;
  for (uint32_t a = 1; a < nb; ++a) {  // ascending constraint id = insertion order
    uint32_t v = row[a], b = a;
    while (b > 0 && row[b - 1] > v) { row[b] = row[b - 1]; --b; }
    row[b] = v;
  }
  uint32_t* succ = reinterpret_cast<uint32_t*>(K.succ);
  const uint32_t first = na ? lo : row[0];
  for (uint32_t k = 0; k < total; ++k) {
    const bool last = k + 1 == total;
    const uint32_t u = k < na ? lo + k : row[k - na], role = k < na ? 0u : 1u;
    const uint32_t wid = last ? first : (k + 1 < na ? lo + k + 1 : row[k + 1 - na]);
    succ[2 * u + role] = wid | (K.ab[wid].y != kNone ? kSuccTwo : 0u) | (last ? kSuccWrap : 0u);
    K.pred[2 * u + role] = k > 0 ? 1 : 0;  // predecessor on this body inside one iteration
  }
}

// The solver walks the dependency graph of the WHOLE Solver::solve call (iters x constraints,
// solver.rs:72-78) as one frontier process: a constraint's round k may run once the previous
// constraint on each of its bodies has run (its round k, or round k-1 across the wrap).  Every
// launch solves the current frontier and appends the constraints it released.  This is exactly the
// sequential Gauss-Seidel result; rounds of different constraints overlap, so the number of
// launches is the depth of the unrolled graph (about half of iters x per-iteration depth).
struct Frontier {
  uint32_t* order;     // frontier lists, appended launch after launch (capacity iters * C)
  uint32_t* lvl_off;   // lvl_off[r] = start of launch r's list
  uint32_t* cnt;       // 3 rotating list-size counters: launch r reads cnt[r%3], appends under cnt[(r+1)%3],
                       // clears cnt[(r+2)%3] (nobody touches it during launch r) - no fences, no last-block logic
};

// Start of a Solver::solve call: reset round / in-degree of every record; launch 0's list =
// constraints without predecessors in iteration 0.  Block-aggregated append.
__global__ __launch_bounds__(kBlock) void k_frontier0(uint32_t C, CRec* cons, ConsLinks K, Frontier F) {
  __shared__ uint32_t s_n, s_base;
  for (uint32_t c0 = blockIdx.x * kBlock; c0 < C; c0 += gridDim.x * kBlock) {
    uint32_t c = c0 + threadIdx.x;
    bool ready = false;
    if (c < C) {
      uint32_t d0 = links_indeg0(K, c);
      ready = d0 == 0;
      cons[c].round = 0;
      cons[c].indeg = ready ? (cons[c].b != kNone ? 2u : 1u) : d0;  // ready ones are armed for their later rounds
    }
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    uint32_t slot = 0;
    if (ready) slot = atomicAdd(&s_n, 1u);
    __syncthreads();
    if (threadIdx.x == 0 && s_n) s_base = atomicAdd(&F.cnt[0], s_n);
    __syncthreads();
    if (ready) F.order[s_base + slot] = c;
    __syncthreads();
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) { F.lvl_off[0] = 0; F.cnt[1] = 0; F.cnt[2] = 0; }
}

// ContactConstraint::solve solver.rs:203-252 (single contact), incl. the reference's quirks:
// both friction rows use the dv from before the first row (:217-232); the friction impulse is
// applied unclamped (:226-231).
__device__ __forceinline__ void solve_core(V3 n, V3 t0, V3 t1, V3 ra, V3 rb, float bias, float nmass, float tmass0, float tmass1,
                                           float& nimp, BodyDyn& A, BodyDyn& Bd) {
  V3 va = A.v, oa = A.w, vb = Bd.v, ob = Bd.w;
  V3 dv = vb + cross(ob, rb) - va - cross(oa, ra);
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    V3 t = k == 0 ? t0 : t1;
    float tm = k == 0 ? tmass0 : tmass1;
    float lambda = -dot(dv, t) * tm;
    V3 impulse = t * lambda;
    va = va - impulse * A.im;
    oa = oa - A.I * cross(ra, impulse);
    vb = vb + impulse * Bd.im;
    ob = ob + Bd.I * cross(rb, impulse);
  }
  V3 dv2 = vb + cross(ob, rb) - va - cross(oa, ra);
  float vn = dot(dv2, n);
  float lambda = nmass * (-vn + bias);
  float prev = nimp;
  nimp = fmax_rs(prev + lambda, 0.0f);
  lambda = nimp - prev;
  V3 impulse = n * lambda;
  va = va - impulse * A.im;
  oa = oa - A.I * cross(ra, impulse);
  vb = vb + impulse * Bd.im;
  ob = ob + Bd.I * cross(rb, impulse);
  A.v = va; A.w = oa; Bd.v = vb; Bd.w = ob;
}
__device__ __forceinline__ void solve_one(CRec& c, BodyDyn& A, BodyDyn& Bd) {
  solve_core(ld3(c.n), ld3(c.t0), ld3(c.t1), ld3(c.ra), ld3(c.rb), c.bias, c.nmass, c.tmass0, c.tmass1, c.nimp, A, Bd);
}
__device__ __forceinline__ void store_vel(float4* srec, uint32_t i, const BodyDyn& d) {  // ConstrainedSet::set physics.rs:306-314
  srec[4 * i] = make_float4(d.v.x, d.v.y, d.v.z, d.w.x);
  float2* p = reinterpret_cast<float2*>(&srec[4 * i + 1]);
  *p = make_float2(d.w.y, d.w.z);
}

// One launch of the frontier process.
__global__ __launch_bounds__(kBlock) void k_solve(float4* srec, CRec* cons, ConsLinks K, Frontier F, uint32_t launch, uint32_t iters) {
  __shared__ uint32_t s_q[2 * kBlock];
  __shared__ uint32_t s_n, s_base;
  const uint32_t lo = F.lvl_off[launch];
  const uint32_t hi = lo + F.cnt[launch % 3];
  uint32_t* next_cnt = F.cnt + (launch + 1) % 3;
  if (blockIdx.x == 0 && threadIdx.x == 0) { F.lvl_off[launch + 1] = hi; F.cnt[(launch + 2) % 3] = 0; }
  for (uint32_t p0 = lo + blockIdx.x * kBlock; p0 < hi; p0 += gridDim.x * kBlock) {
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    uint32_t p = p0 + threadIdx.x;
    if (p < hi) {
      uint32_t cid = F.order[p];
      CRec c = load_crec(&cons[cid]);
      BodyDyn A = load_dyn(srec, c.a);
      BodyDyn Bd = (c.b == kNone) ? static_dyn() : load_dyn(srec, c.b);
      solve_one(c, A, Bd);
      store_vel(srec, c.a, A);
      if (c.b != kNone) store_vel(srec, c.b, Bd);
      uint32_t k = c.round;
      *reinterpret_cast<float2*>(&cons[cid].nimp) = make_float2(c.nimp, u2f(k + 1));  // nimp, round
      const uint2 sw = K.succ[cid];
#pragma unroll
      for (int side = 0; side < 2; ++side) {
        if (side == 1 && c.b == kNone) break;
        uint32_t w = side == 0 ? sw.x : sw.y;
        uint32_t ks = k + (w >> 31);  // the successor's round this release belongs to
        if (ks >= iters) continue;
        uint32_t sid = w & kSuccId;
        if (atomicSub(&cons[sid].indeg, 1u) == 1u) {
          cons[sid].indeg = (w & kSuccTwo) ? 2u : 1u;  // re-arm for its next round (nobody decrements before it runs)
          s_q[atomicAdd(&s_n, 1u)] = sid;
        }
      }
    }
    __syncthreads();
    uint32_t m = s_n;
    if (threadIdx.x == 0 && m) s_base = atomicAdd(next_cnt, m);
    __syncthreads();
    for (uint32_t e = threadIdx.x; e < m; e += kBlock) F.order[hi + s_base + e] = s_q[e];
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------
// Dataflow solver: the same dependency graph as k_solve, walked by ONE persistent launch.
//
// Every constraint c has a fixed owner lane (c mod L, L = lanes of the resident grid); a lane runs its
// nodes in (iteration, constraint) order - a topological order of the unrolled graph, so the globally
// smallest pending node is always runnable and the process cannot deadlock while every lane is resident.
// Readiness is an arrival counter: a finished node adds 1 (2 if the successor has a single dynamic
// body) to each successor's counter, and node (c, k) may run once arr[c] >= 2 (k + 1).  No queues, no
// kernel boundaries: a hand-off costs one write-through store + one device-scope atomic on the
// producer and one polled load on the consumer (MI355X guide, Guideline 16 recipe R1):
//   * body velocities are exchanged with sc1 (write-through / L1-bypassing) 16-byte buffer accesses,
//   * the producer drains its stores (s_waitcnt vmcnt(0)) before the relaxed agent-scope atomic,
//   * the consumer polls with relaxed agent-scope loads, then issues its sc1 loads.
// Constraint records are private to their owner lane (plain accesses).  Spins are bounded: a lane that
// waits too long raises `abort` and every lane leaves (the host reports MGF_ERR_HIP).
// ------------------------------------------------------------------------------------------
typedef float v4f_t __attribute__((ext_vector_type(4)));
typedef float v2f_t __attribute__((ext_vector_type(2)));
constexpr int kSc1 = 16;  // aux bits of the raw buffer builtins on gfx950: sc1

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(p, 0, 0x7FFFFFFF, 0x00020000);
}
__device__ __forceinline__ BodyDyn load_dyn_sc1(__amdgpu_buffer_rsrc_t r, uint32_t i) {
  v4f_t s0 = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(i * 64u), 0, kSc1);
  v4f_t s1 = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(i * 64u + 16u), 0, kSc1);
  v4f_t s2 = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(i * 64u + 32u), 0, kSc1);
  v4f_t s3 = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(i * 64u + 48u), 0, kSc1);
  BodyDyn d;
  d.v = mk3(s0.x, s0.y, s0.z); d.w = mk3(s0.w, s1.x, s1.y); d.im = s1.z;
  d.I = m3_cols(mk3(s1.w, s2.x, s2.y), mk3(s2.z, s2.w, s3.x), mk3(s3.y, s3.z, s3.w));
  return d;
}
__device__ __forceinline__ void store_vel_sc1(__amdgpu_buffer_rsrc_t r, uint32_t i, const BodyDyn& d) {
  v4f_t a = {d.v.x, d.v.y, d.v.z, d.w.x};
  v2f_t b = {d.w.y, d.w.z};
  __builtin_amdgcn_raw_buffer_store_b128(a, r, (int)(i * 64u), 0, kSc1);
  __builtin_amdgcn_raw_buffer_store_b64(b, r, (int)(i * 64u + 16u), 0, kSc1);
}

// the same by byte offset (an offset beyond the buffer's range reads zeros / drops the store: used for "no global body")
__device__ __forceinline__ BodyDyn load_dyn_off_sc1(__amdgpu_buffer_rsrc_t r, uint32_t off) {
  v4f_t s0 = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, kSc1);
  v4f_t s1 = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(off + 16u), 0, kSc1);
  v4f_t s2 = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(off + 32u), 0, kSc1);
  v4f_t s3 = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(off + 48u), 0, kSc1);
  BodyDyn d;
  d.v = mk3(s0.x, s0.y, s0.z); d.w = mk3(s0.w, s1.x, s1.y); d.im = s1.z;
  d.I = m3_cols(mk3(s1.w, s2.x, s2.y), mk3(s2.z, s2.w, s3.x), mk3(s3.y, s3.z, s3.w));
  return d;
}
__device__ __forceinline__ void store_vel_off_sc1(__amdgpu_buffer_rsrc_t r, uint32_t off, const BodyDyn& d) {
  v4f_t a = {d.v.x, d.v.y, d.v.z, d.w.x};
  v2f_t b = {d.w.y, d.w.z};
  __builtin_amdgcn_raw_buffer_store_b128(a, r, (int)off, 0, kSc1);
  __builtin_amdgcn_raw_buffer_store_b64(b, r, (int)(off + 16u), 0, kSc1);
}
__device__ __forceinline__ V3 sel3(bool c, V3 a, V3 b) { return mk3(c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z); }
__device__ __forceinline__ BodyDyn select_dyn(bool c, const BodyDyn& a, const BodyDyn& b) {
  BodyDyn d;
  d.v = sel3(c, a.v, b.v); d.w = sel3(c, a.w, b.w); d.im = c ? a.im : b.im;
  d.I = m3_cols(sel3(c, a.I.c[0], b.I.c[0]), sel3(c, a.I.c[1], b.I.c[1]), sel3(c, a.I.c[2], b.I.c[2]));
  return d;
}

// arr[c] = 2 - (weighted predecessors inside iteration 0): node (c, 0) is ready at arr >= 2.
__global__ __launch_bounds__(kBlock) void k_flow_init(const uint32_t* C_ptr, ConsLinks K, uint32_t* arr, uint32_t* abort_flag) {
  uint32_t c = blockIdx.x * kBlock + threadIdx.x;
  if (c == 0) *abort_flag = 0;
  if (c >= *C_ptr) return;
  uint32_t d0 = links_indeg0(K, c);
  arr[c] = 2u - d0 * (K.ab[c].y != kNone ? 1u : 2u);
}

template <bool TRACE>
__global__ __launch_bounds__(kBlock) void k_solve_flow(float4* srec, CRec* cons, ConsLinks K, uint32_t* arr, const uint32_t* C_ptr, uint32_t iters,
                                                       uint32_t* abort_flag, uint32_t spin_limit, int sleep_mode, uint64_t* trace,
                                                       const uint32_t* run_if) {
  if (run_if && *run_if == 0u) return;  // stand-by launch behind the block-local solver: runs only if that one declined
  const uint32_t C = *C_ptr;
  const uint32_t L = gridDim.x * kBlock;
  const uint32_t gl = blockIdx.x * kBlock + threadIdx.x;
  __amdgpu_buffer_rsrc_t rs = make_rsrc(srec);
  uint32_t c = gl, round = 0;
  bool done = (c >= C) || iters == 0;
  bool have_rec = false;
  CRec rec;
  uint2 sw = make_uint2(0u, 0u);
  uint32_t spins = 0;
  for (;;) {
    if (!__any(!done)) break;
    bool progressed = false;
    if (!done) {
      // the record is private to this lane: fetch it while the node is still waiting for its predecessors
      if (!have_rec) { rec = load_crec(&cons[c]); sw = K.succ[c]; have_rec = true; }
      uint32_t a = __hip_atomic_load(&arr[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (a >= 2u * (round + 1u)) {
        asm volatile("" ::: "memory");  // nothing below may be hoisted above the poll
        if (TRACE) trace[2 * ((size_t)round * C + c)] = wall_clock64();
        BodyDyn A = load_dyn_sc1(rs, rec.a);
        BodyDyn Bd = (rec.b == kNone) ? static_dyn() : load_dyn_sc1(rs, rec.b);
        solve_one(rec, A, Bd);
        store_vel_sc1(rs, rec.a, A);
        if (rec.b != kNone) store_vel_sc1(rs, rec.b, Bd);
        cons[c].nimp = rec.nimp;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // velocities are out before the successors hear of it
#pragma unroll
        for (int side = 0; side < 2; ++side) {
          if (side == 1 && rec.b == kNone) break;
          uint32_t w = side == 0 ? sw.x : sw.y;
          if (round + (w >> 31) >= iters) continue;
          __hip_atomic_fetch_add(&arr[w & kSuccId], (w & kSuccTwo) ? 1u : 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (TRACE) trace[2 * ((size_t)round * C + c) + 1] = wall_clock64();
        progressed = true;
        have_rec = false;
        c += L;
        if (c >= C) { c = gl; ++round; if (round >= iters) done = true; }
      }
    }
    if (__any(progressed)) { spins = 0; continue; }
    if (sleep_mode == 1) __builtin_amdgcn_s_sleep(1);
    else if (sleep_mode == 2) __builtin_amdgcn_s_sleep(4);
    else if (sleep_mode == 3) __builtin_amdgcn_s_sleep(16);
    if ((++spins & 255u) == 0u) {
      bool give_up = spins > spin_limit;
      if (give_up) __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (give_up || __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
    }
  }
}

// ------------------------------------------------------------------------------------------
// Dataflow solver with KS out-of-order slots per lane (solver mode 4).  Same protocol as k_solve_flow
// (arrival counters, write-through velocity hand-offs), but a lane's node sequence is dealt round-robin
// onto KS slots, each slot walks its own sub-sequence in (iteration, constraint) order, and every trip
// polls the head of every slot and runs the first ready one.  With KS * L >= C each slot holds one
// constraint, so a ready node never waits behind an unready earlier node of the same lane (the
// head-of-line blocking that dominates k_solve_flow's critical path: median hand-off 1.5 us, mean 5.4 us).
// Still deadlock-free: the globally smallest pending node is the head of its slot.
// ------------------------------------------------------------------------------------------
template <int KS, bool TRACE>
__global__ __launch_bounds__(kBlock) void k_solve_flowk(float4* srec, CRec* cons, ConsLinks K, uint32_t* arr, const uint32_t* C_ptr, uint32_t iters,
                                                        uint32_t* abort_flag, uint32_t spin_limit, int sleep_mode, uint64_t* trace) {
  const uint32_t C = *C_ptr;
  const uint32_t L = gridDim.x * kBlock;
  const uint32_t gl = blockIdx.x * kBlock + threadIdx.x;
  __amdgpu_buffer_rsrc_t rs = make_rsrc(srec);
  uint32_t sc[KS], sr[KS], sa[KS], sb[KS];  // head node of each slot: constraint, iteration, its two bodies
#pragma unroll
  for (int j = 0; j < KS; ++j) {
    sc[j] = gl + (uint32_t)j * L; sr[j] = iters; sa[j] = 0; sb[j] = kNone;
    if (sc[j] < C && iters > 0) { sr[j] = 0; uint2 ab = K.ab[sc[j]]; sa[j] = ab.x; sb[j] = ab.y; }
  }
  uint32_t spins = 0;
  for (;;) {
    bool live = false;
#pragma unroll
    for (int j = 0; j < KS; ++j) live |= sr[j] < iters;
    if (!__any(live)) break;
    uint32_t av[KS];
#pragma unroll
    for (int j = 0; j < KS; ++j) av[j] = (sr[j] < iters) ? __hip_atomic_load(&arr[sc[j]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    int pick = -1;
#pragma unroll
    for (int j = KS - 1; j >= 0; --j)
      if (sr[j] < iters && av[j] >= 2u * (sr[j] + 1u)) pick = j;
    if (pick >= 0) {
      asm volatile("" ::: "memory");  // nothing below may be hoisted above the poll
      uint32_t c = sc[0], round = sr[0], ia = sa[0], ib = sb[0];
#pragma unroll
      for (int j = 1; j < KS; ++j)
        if (pick == j) { c = sc[j]; round = sr[j]; ia = sa[j]; ib = sb[j]; }
      if (TRACE) trace[2 * ((size_t)round * C + c)] = wall_clock64();
      CRec rec = load_crec(&cons[c]);  // private to this lane; in flight together with the body records
      const uint2 sw = K.succ[c];
      BodyDyn A = load_dyn_sc1(rs, ia);
      BodyDyn Bd = (ib == kNone) ? static_dyn() : load_dyn_sc1(rs, ib);
      solve_one(rec, A, Bd);
      store_vel_sc1(rs, ia, A);
      if (ib != kNone) store_vel_sc1(rs, ib, Bd);
      cons[c].nimp = rec.nimp;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // velocities are out before the successors hear of it
#pragma unroll
      for (int side = 0; side < 2; ++side) {
        if (side == 1 && ib == kNone) break;
        uint32_t w = side == 0 ? sw.x : sw.y;
        if (round + (w >> 31) >= iters) continue;
        __hip_atomic_fetch_add(&arr[w & kSuccId], (w & kSuccTwo) ? 1u : 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (TRACE) trace[2 * ((size_t)round * C + c) + 1] = wall_clock64();
      // next node of this slot
      uint32_t cn = c + (uint32_t)KS * L;
      if (cn >= C) { cn = gl + (uint32_t)pick * L; ++round; }
      if (cn != c && round < iters) { uint2 ab = K.ab[cn]; ia = ab.x; ib = ab.y; }
#pragma unroll
      for (int j = 0; j < KS; ++j)
        if (pick == j) { sc[j] = cn; sr[j] = round; sa[j] = ia; sb[j] = ib; }
    }
    if (__any(pick >= 0)) { spins = 0; continue; }
    if (sleep_mode == 1) __builtin_amdgcn_s_sleep(1);
    else if (sleep_mode == 2) __builtin_amdgcn_s_sleep(4);
    else if (sleep_mode == 3) __builtin_amdgcn_s_sleep(16);
    if ((++spins & 255u) == 0u) {
      bool give_up = spins > spin_limit;
      if (give_up) __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (give_up || __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
    }
  }
}

// ------------------------------------------------------------------------------------------
// Block-local dataflow solver (solver mode 5).  Same dependency graph and the same arrival-counter protocol as
// k_solve_flow, but the work is cut into spatial blocks: bodies in cell (Morton) order, `nb` consecutive bodies per
// block, ONE 512-thread workgroup per block, one block per CU.  A constraint belongs to the block of its body a.
//   * A body touched only by its own block's constraints is PRIVATE: its 64-byte solver record lives in the
//     workgroup's LDS for the whole Solver::solve call.  Bodies touched from two blocks stay in global memory and
//     are exchanged with write-through (sc1) accesses as in k_solve_flow.
//   * A constraint whose predecessors all belong to its own block has its arrival counter in LDS; the others use
//     the global counter array.
//   * Constraint records stay in global memory; ready nodes go through two LDS queues (one for constraints that
//     live entirely in LDS, one for those that touch global memory) and ANY lane of the serving waves may run
//     any ready node: a wave takes up to 64 nodes per trip, instead of the few its own lanes would hold if
//     constraints were pinned to lanes (measured: pinned lanes ran ~10 of 64 lanes per trip, issue-bound).
// A hand-off inside a block is an LDS write + an LDS atomic; only hand-offs across block faces pay the L2 price.
// The waves serving the all-LDS queue only touch global memory for the record fetch; a few waves serve the
// other queue and poll the global counters.  Every ready node is eventually taken, so the scheme is
// deadlock-free as long as all blocks are resident.
// ------------------------------------------------------------------------------------------
constexpr int kF5Threads = 512;
// Two LDS layouts (template parameter WIDE of k_solve_flow5), chosen by the host from last tick's largest block:
//   WIDE = false: up to 3072 constraints per block, every slot's constants in LDS (25 B per slot);
//   WIDE = true : up to 5120 (a settled 64^3 pile reaches ~4600), only the class-0 slots' constants in LDS (16 B each,
//                 at most 3328), classes 1 + 2 (at most 3072) read theirs from the block's global table.
constexpr uint32_t kF5MaxCons = 5120;                     // rows per block in the global slot tables
constexpr uint32_t kF5NarrowCons = 3072;
constexpr uint32_t kF5MaxFast = 3328, kF5MaxSlow = 3072;
constexpr uint32_t kF5MaxBodies = 1100;                   // bodies per block (LDS: 64 B each; both layouts must fit 160 KB)
constexpr uint32_t kF5LdsWide = 16u * kF5MaxFast + 5u * kF5MaxCons + 2u * (kF5MaxFast + kF5MaxSlow) + 64u;
constexpr uint32_t kF5LdsNarrow = 25u * kF5NarrowCons + 2u * 2u * 4096u + 64u;
constexpr uint32_t kRefGlobal = 0x80000000u;              // body ref: bit 31 = global id (sc1 path), else LDS index; kNone = static
constexpr uint32_t kSuccLocal = 0x20000000u;              // successor word: low bits are a block-local slot
constexpr uint32_t kRefHasLocal = 0x40000000u;            // a-ref of a class-1 slot: one of its two predecessors is in-block
constexpr uint32_t kF5RemoteDone = 0x100u;                // class-1 LDS counter: the arrivals from other blocks are in (set by the poller)
// One 32-byte row per slot (written once per tick by k_flow5_table with two 16-byte stores, read coalesced).
struct F5Row {
  uint32_t c;       // constraint id
  uint32_t aref;    // body refs (LDS index, or id | kRefGlobal, or kNone); a-ref bit kRefHasLocal
  uint32_t bref;
  uint32_t cnt0;    // bits 0-7: in-block arrival counter of iteration 0; bits 8..: the same for arrivals from other blocks
  uint32_t succ0, succ1;  // successor words, block-local slots or arr5 rows
  uint32_t pad0, pad1;
};
static_assert(sizeof(F5Row) == 32, "F5Row is two 16-byte words");
struct Flow5 {
  const uint32_t* sidx;    // cell-ordered body ids
  const uint32_t* brank;   // body -> position in cell order
  uint8_t* shared;         // body touched by constraints of two blocks
  uint32_t* gcnt;          // per constraint: weight of its predecessors in OTHER blocks (bits 0-1: per iteration, 2 in total with the
                           // in-block ones; bits 2-3: those that arrive inside iteration 0).  Non-zero = class 1.
  uint32_t* arr5;          // class 1: arrivals from other blocks, scaled to 2 per iteration (in-block arrivals count in LDS).
                           // Indexed like the slot tables (block * kF5MaxCons + slot), so a block's counters are contiguous and
                           // its pollers read them coalesced; written by k_flow5_table, re-armed by the solve kernel on exit.
  uint32_t* lslot;         // constraint -> (class << 12) | index inside its block's class
  uint32_t* wg_cnt;        // per block and class k (0 all-LDS, 1 global counter, 2 LDS counter + shared body): f5_cnt(F, g, k),
                           // one 128-byte line per counter (same-line atomics serialise)
  // per block, kF5MaxCons rows, final slot order (class 0, then 1, then 2): what k_solve_flow5 copies into LDS
  F5Row* table;
  uint32_t* fail;          // set when a block does not fit (the host falls back to k_solve_flow)
  uint32_t* max_block;     // largest block of this tick (the host picks next tick's LDS layout from it)
  uint32_t nb, nblocks, n;
  uint32_t cap_fast, cap_slow, cap_all;  // limits of the chosen layout
  uint32_t slow_x2;        // waves serving the slow queue = slow share of the slots x slow_x2 / 2 (tuning knob, 3)
  uint32_t poller;         // 1: the last slow wave only polls the global counters (all of them), the others only serve
};
__device__ __forceinline__ uint32_t f5_ref(const Flow5& F, uint32_t g, uint32_t body) {
  if (body == kNone) return kNone;
  return (F.brank[body] / F.nb != g || F.shared[body]) ? (body | kRefGlobal) : (F.brank[body] - g * F.nb);
}
constexpr uint32_t kF5CntStride = 32;  // words
__device__ __forceinline__ uint32_t* f5_cnt(const Flow5& F, uint32_t g, uint32_t k) { return F.wg_cnt + (size_t)(4u * g + k) * kF5CntStride; }
// final slot of a constraint inside its block: classes are laid out 0 | 1 | 2
__device__ __forceinline__ uint32_t f5_slot(const Flow5& F, uint32_t g, uint32_t packed) {
  uint32_t k = packed >> 12, idx = packed & 0xFFFu;
  uint32_t base = k == 0 ? 0u : (k == 1 ? *f5_cnt(F, g, 0) : *f5_cnt(F, g, 0) + *f5_cnt(F, g, 1));
  return base + idx;
}
__global__ __launch_bounds__(kBlock) void k_flow5_mark(Flow5 F, ConsLinks K, const uint32_t* C_ptr) {
  uint32_t c = blockIdx.x * kBlock + threadIdx.x;
  if (c >= *C_ptr) return;
  uint2 e = K.ab[c];
  uint32_t ga = F.brank[e.x] / F.nb;
  if (e.y != kNone && F.brank[e.y] / F.nb != ga) F.shared[e.y] = 1;
  uint2 sw = K.succ[c];
  uint32_t w[2] = {sw.x, sw.y};
#pragma unroll
  for (int side = 0; side < 2; ++side) {
    if (side == 1 && e.y == kNone) break;
    uint32_t sid = w[side] & kSuccId;
    if (F.brank[K.ab[sid].x] / F.nb != ga) {
      uint32_t add = (w[side] & kSuccTwo) ? 1u : 2u;
      atomicAdd(&F.gcnt[sid], add | ((w[side] & kSuccWrap) ? 0u : add << 2));
    }
  }
}

// Per constraint: class (0: arrival counter and both bodies in LDS; 1: global arrival counter - a predecessor lives in
// another block; 2: LDS counter, but a body shared with another block) and an index inside that class of its block.
// The order inside a class is arrival order; any order is valid (every ready node may run).
__global__ __launch_bounds__(kBlock) void k_flow5_assign(Flow5 F, ConsLinks K, const uint32_t* C_ptr) {
  uint32_t c = blockIdx.x * kBlock + threadIdx.x;
  if (c >= *C_ptr) return;
  uint2 e = K.ab[c];
  uint32_t g = F.brank[e.x] / F.nb;
  uint32_t k;
  if (F.gcnt[c]) k = 1;
  else k = ((f5_ref(F, g, e.x) & kRefGlobal) || (e.y != kNone && (f5_ref(F, g, e.y) & kRefGlobal))) ? 2u : 0u;
  uint32_t idx = atomicAdd(f5_cnt(F, g, k), 1u);
  if (idx >= kF5MaxCons) { *F.fail = 1u; idx = 0; }
  F.lslot[c] = (k << 12) | idx;
}
// Per constraint: its row of the block's slot table, in final slot order.
__global__ __launch_bounds__(kBlock) void k_flow5_table(Flow5 F, ConsLinks K, const uint32_t* C_ptr) {
  uint32_t c = blockIdx.x * kBlock + threadIdx.x;
  if (c >= *C_ptr) return;
  uint2 e = K.ab[c];
  uint32_t g = F.brank[e.x] / F.nb;
  {
    uint32_t n0 = *f5_cnt(F, g, 0), n12 = *f5_cnt(F, g, 1) + *f5_cnt(F, g, 2);
    if (F.lslot[c] == 0u) atomicMax(F.max_block, n0 + n12);  // once per block: its class-0 slot 0 (or nobody, for a block without one)
    if (n0 > F.cap_fast || n12 > F.cap_slow || n0 + n12 > F.cap_all) { *F.fail = 1u; return; }
  }
  uint32_t slot = f5_slot(F, g, F.lslot[c]);
  size_t row = (size_t)g * kF5MaxCons + slot;
  const uint32_t rw = F.gcnt[c] & 3u, rnw = (F.gcnt[c] >> 2) & 3u;
  F5Row R;
  R.c = c;
  R.aref = f5_ref(F, g, e.x) | (rw == 1u ? kRefHasLocal : 0u);
  R.bref = f5_ref(F, g, e.y);
  // the LDS counter counts in-block arrivals only: iteration 0 starts with the credit of the in-block wrap edges;
  // bits 8..: the same for the arrivals from other blocks (arr5, scaled to 2 per iteration)
  const uint32_t remote0 = rw ? 2u - rnw * (2u / rw) : 0u;
  R.cnt0 = (2u - (links_indeg0(K, c) * (e.y != kNone ? 1u : 2u) - rnw)) | (remote0 << 8);
  F.arr5[row] = remote0;
  uint2 sw = K.succ[c];
  uint32_t w[2] = {sw.x, sw.y};
#pragma unroll
  for (int side = 0; side < 2; ++side) {
    if (side == 1 && e.y == kNone) { w[1] = 0u; break; }
    uint32_t sid = w[side] & kSuccId;
    bool local = F.brank[K.ab[sid].x] / F.nb == g;
    if (local) w[side] = (w[side] & (kSuccTwo | kSuccWrap)) | kSuccLocal | f5_slot(F, g, F.lslot[sid]);
    else {  // in another block: the word names its row of arr5
      const uint32_t gs = F.brank[K.ab[sid].x] / F.nb;
      uint32_t flags = w[side] & (kSuccTwo | kSuccWrap);
      if ((F.gcnt[sid] & 3u) == 1u) flags &= ~kSuccTwo;  // its only arrival from outside counts 2 (2 per iteration, uniformly)
      w[side] = flags | (gs * kF5MaxCons + f5_slot(F, gs, F.lslot[sid]));
    }
  }
  R.succ0 = w[0]; R.succ1 = w[1]; R.pad0 = R.pad1 = 0u;
  uint4* dst = reinterpret_cast<uint4*>(&F.table[row]);
  dst[0] = make_uint4(R.c, R.aref, R.bref, R.cnt0);
  dst[1] = make_uint4(R.succ0, R.succ1, 0u, 0u);
}

__device__ __forceinline__ BodyDyn f5_load_body(const float4* s_body, __amdgpu_buffer_rsrc_t rs, uint32_t ref) {
  if (ref == kNone) return static_dyn();
  if (ref & kRefGlobal) return load_dyn_sc1(rs, ref & ~kRefGlobal);
  return load_dyn(s_body, ref);
}
__device__ __forceinline__ void f5_store_vel(float4* s_body, __amdgpu_buffer_rsrc_t rs, uint32_t ref, const BodyDyn& d) {
  if (ref == kNone) return;
  if (ref & kRefGlobal) { store_vel_sc1(rs, ref & ~kRefGlobal, d); return; }
  store_vel(s_body, ref, d);
}

// LDS per block: slot constants (constraint id, body refs, successor words) - of every slot (narrow layout) or of the
// class-0 slots only (wide layout; classes 1 and 2 touch global memory anyway and read theirs from the block's global
// table) - and for EVERY slot its arrival counter and iteration counter.
// Ready queues in LDS (one for the all-LDS class, one for the rest): any lane of the serving waves may run any ready
// node, so a wave takes up to 64 of them per trip instead of the few its own lanes would hold.  A slot is queued at most
// once at a time, so a ring as long as its class never overflows.
struct F5Queue { uint16_t* ring; uint32_t* head; uint32_t* tail; uint32_t cap; };
__device__ __forceinline__ void f5_push(const F5Queue& q, uint32_t slot) {
  uint32_t pos = __hip_atomic_fetch_add(q.tail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  q.ring[pos % q.cap] = (uint16_t)(slot | 0x8000u);
}

template <bool WIDE, bool TRACE>
__global__ __launch_bounds__(kF5Threads) void k_solve_flow5(float4* srec, CRec* cons, ConsLinks K, Flow5 F, uint32_t* arr, uint32_t iters,
                                                            uint32_t* abort_flag, uint32_t spin_limit, uint64_t* trace, uint32_t C_trace) {
  if (*F.fail) return;  // a block did not fit: the stand-by k_solve_flow launch behind this one does the work
  constexpr uint32_t kMeta = WIDE ? kF5MaxFast : kF5NarrowCons;   // slots with constants in LDS
  constexpr uint32_t kAll = WIDE ? kF5MaxCons : kF5NarrowCons;    // slots with counters in LDS
  constexpr uint32_t kRingF = WIDE ? kF5MaxFast : 4096u, kRingS = WIDE ? kF5MaxSlow : 4096u;
  extern __shared__ float4 s_dyn[];
  float4* s_body = s_dyn;  // 4 x nb
  uint2* s_succ = reinterpret_cast<uint2*>(s_dyn + 4 * (size_t)F.nb);      // [kMeta]
  uint32_t* s_c = reinterpret_cast<uint32_t*>(s_succ + kMeta);             // [kMeta]
  uint32_t* s_a = s_c + kMeta;                                             // [kMeta] WIDE: aref | bref << 16; else aref
  uint32_t* s_b = s_a + kMeta;                                             // [kMeta] narrow layout only
  uint32_t* s_cnt = WIDE ? s_b : s_b + kMeta;                              // [kAll] arrivals since the slot last ran: ready at 2
  uint32_t* s_ctl = s_cnt + kAll;  // [0,1] fast head/tail, [2,3] slow head/tail, [4] nodes left
  F5Queue qf, qs;
  qf.head = s_ctl; qf.tail = s_ctl + 1; qs.head = s_ctl + 2; qs.tail = s_ctl + 3;
  uint32_t* s_left = s_ctl + 4;
  qf.ring = reinterpret_cast<uint16_t*>(s_ctl + 8); qf.cap = kRingF;
  qs.ring = qf.ring + kRingF; qs.cap = kRingS;
  uint8_t* s_round = reinterpret_cast<uint8_t*>(qs.ring + kRingS);         // [kAll] iterations done; bit 7: queued by its poller
  const uint32_t g = blockIdx.x, t = threadIdx.x;
  const uint32_t p_lo = g * F.nb, p_hi = min(F.n, p_lo + F.nb);
  __amdgpu_buffer_rsrc_t rs = make_rsrc(srec);
  uint32_t* const arr5 = F.arr5;
  for (uint32_t p = p_lo + t; p < p_hi; p += kF5Threads) {
    uint32_t x = F.sidx[p];
#pragma unroll
    for (int k = 0; k < 4; ++k) s_body[4 * (p - p_lo) + k] = srec[4 * (size_t)x + k];
  }
  const uint32_t N0 = *f5_cnt(F, g, 0), N01 = N0 + *f5_cnt(F, g, 1), N = N01 + *f5_cnt(F, g, 2);
  const uint32_t n_meta = WIDE ? N0 : N;
  for (uint32_t e = t; e < (kRingF + kRingS) / 2u; e += kF5Threads) reinterpret_cast<uint32_t*>(qf.ring)[e] = 0u;  // both rings
  if (t < 8) s_ctl[t] = t == 4 ? N * iters : 0u;
  __syncthreads();
  // the block's slot table (built once per tick by k_flow5_table): a coalesced copy of what the LDS side needs
  const size_t row0 = (size_t)g * kF5MaxCons;
  for (uint32_t idx = t; idx < N; idx += kF5Threads) {
    const uint4* src = reinterpret_cast<const uint4*>(&F.table[row0 + idx]);
    const uint4 r0 = src[0];  // c, aref, bref, cnt0
    uint32_t c0 = r0.w & 0xFFu;
    s_cnt[idx] = c0;
    s_round[idx] = 0;
    if (idx < n_meta) {
      const uint4 r1 = src[1];  // successor words
      s_c[idx] = r0.x;
      uint32_t ar = r0.y, br = r0.z;
      if (WIDE) s_a[idx] = (ar & 0xFFFFu) | ((br == kNone ? 0xFFFFu : br) << 16);  // class 0: LDS indices or static
      else { s_a[idx] = ar; s_b[idx] = br; }
      s_succ[idx] = make_uint2(r1.x, r1.y);
    }
    // iteration 0's frontier (slots with a global counter are found by their pollers)
    if (!(idx >= N0 && idx < N01) && c0 >= 2u && iters > 0) f5_push(idx < N0 ? qf : qs, idx);
  }
  __syncthreads();
  // waves [0, nfast) serve the fast queue, the rest the slow queue; the slow waves also poll the global counters
  // (a dedicated polling wave was tried: slower, it keeps the CU's memory queue busy)
  const uint32_t wave = t >> 6, lane = t & 63u, nwaves = kF5Threads / 64u;
  uint32_t nslow = N > N0 ? (F.slow_x2 * nwaves * (N - N0) + 2u * N - 1u) / (2u * N) : 0u;
  if (N > N0 && nslow < 1u) nslow = 1u;
  if (nslow > nwaves - 1u && N0 > 0u) nslow = nwaves - 1u;
  if (nslow > nwaves) nslow = nwaves;
  const bool slow_wave = wave >= nwaves - nslow;
  const F5Queue& q = slow_wave ? qs : qf;
  // who polls the global counters: every slow wave a share (between its serving trips), or one wave that does nothing else
  const bool poller_wave = F.poller != 0u && nslow >= 2u && wave == nwaves - 1u;
  const bool polls = (F.poller != 0u && nslow >= 2u) ? poller_wave : slow_wave;
  const uint32_t poll_lanes = poller_wave ? 64u : nslow * 64u, poll_id = poller_wave ? lane : (wave - (nwaves - nslow)) * 64u + lane;
  uint32_t spins = 0;
  for (;;) {
    if (__hip_atomic_load(s_left, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0u) break;
    if (polls) {  // arrivals from other blocks that reached their iteration's threshold: say so once per round
      // The block's counters are contiguous: a sweep is a few coalesced loads per wave, issued back to back and tested
      // afterwards (one memory round trip per batch).
      constexpr int kPB = 4;
      for (uint32_t base = N0 + poll_id; base < N01; base += kPB * poll_lanes) {
        uint32_t r[kPB], av[kPB];
#pragma unroll
        for (int k = 0; k < kPB; ++k) {
          uint32_t idx = base + (uint32_t)k * poll_lanes;
          r[k] = idx < N01 ? s_round[idx] : 0xFFu;
        }
#pragma unroll
        for (int k = 0; k < kPB; ++k) {
          uint32_t idx = base + (uint32_t)k * poll_lanes;
          av[k] = __hip_atomic_load(&arr5[row0 + (idx < N01 ? idx : N0)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int k = 0; k < kPB; ++k) {
          uint32_t idx = base + (uint32_t)k * poll_lanes;
          if (r[k] < iters && av[k] >= 2u * (r[k] + 1u)) {
            s_round[idx] = (uint8_t)(r[k] | 0x80u);
            uint32_t old = __hip_atomic_fetch_add(&s_cnt[idx], kF5RemoteDone, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (old == 2u) f5_push(qs, idx);  // the in-block ones too
          }
        }
      }
      if (poller_wave) {  // never serves; leaves with the others
        if ((++spins & 1023u) == 0u && __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
        continue;
      }
    }
    // take up to 64 ready nodes
    uint32_t h = 0, take = 0;
    if (lane == 0) {
      h = __hip_atomic_load(q.head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      uint32_t tl = __hip_atomic_load(q.tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      take = min(tl - h, 64u);
      if (take) {
        uint32_t expect = h;
        if (!__hip_atomic_compare_exchange_strong(q.head, &expect, h + take, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) take = 0;
      }
    }
    h = __shfl(h, 0); take = __shfl(take, 0);
    if (take) {
      spins = 0;
      if (lane < take) {
        uint16_t* cell = &q.ring[(h + lane) % q.cap];
        uint32_t e;
        // (an atomic load, not a volatile one: volatile accesses keep the generic address space and become flat loads)
        do { e = __hip_atomic_load(cell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); } while (!(e & 0x8000u));  // the pusher is between its two writes
        *cell = 0;
        const uint32_t slot = e & 0x7FFFu;
        const uint32_t round = s_round[slot] & 0x7Fu;
        uint64_t t_seen = 0;
        if (TRACE) t_seen = wall_clock64();
        uint32_t c, aref, bref;
        uint2 sw;
        // LDS reads first, unconditionally (clamped), the global table only for the wide layout's slow classes: an
        // if/else over the two sources is merged into flat loads through a selected pointer
        {
          const uint32_t ms = WIDE ? min(slot, kMeta - 1u) : slot;
          c = s_c[ms]; sw = s_succ[ms];
          if (WIDE) { uint32_t ab = s_a[ms]; aref = ab & 0xFFFFu; bref = (ab >> 16) == 0xFFFFu ? kNone : (ab >> 16); }
          else { aref = s_a[ms]; bref = s_b[ms]; }
        }
        if (WIDE) asm volatile("" : "+v"(c), "+v"(sw.x), "+v"(sw.y));  // keeps the LDS reads above the branch (else: sunk and merged into flat loads)
        if (WIDE && slot >= n_meta) {
          const uint4* src = reinterpret_cast<const uint4*>(&F.table[row0 + slot]);
          const uint4 r0 = src[0], r1 = src[1];
          c = r0.x; aref = r0.y; bref = r0.z; sw = make_uint2(r1.x, r1.y);
        }
        const bool has_local = !WIDE || slot >= n_meta ? (aref & kRefHasLocal) != 0u : false;
        if (!WIDE || slot >= n_meta) aref &= ~kRefHasLocal;
        CRec rec = load_crec_solve(&cons[c]);  // only the lane running the constraint touches its record
        BodyDyn A, Bd;
        // byte offsets of the bodies in the global array, or out of the buffer's range for LDS / static refs: such a
        // load returns zeros (= the static body) and such a store is dropped, without touching memory
        uint32_t ga = 0x80000000u, gb = 0x80000000u;
        if (slow_wave) {
          // both bodies' write-through loads go out back to back behind the record's, branch-free: one memory round
          // trip per node instead of three (a branch per source made the compiler wait inside each arm)
          if (aref & kRefGlobal) ga = (aref & ~kRefGlobal) * 64u;
          if (bref != kNone && (bref & kRefGlobal)) gb = (bref & ~kRefGlobal) * 64u;
          BodyDyn Ag = load_dyn_off_sc1(rs, ga), Bg = load_dyn_off_sc1(rs, gb);
          const bool la = !(aref & kRefGlobal), lb = bref != kNone && !(bref & kRefGlobal);
          BodyDyn Al = load_dyn(s_body, la ? aref : 0u), Bl = load_dyn(s_body, lb ? bref : 0u);
          A = select_dyn(la, Al, Ag);
          Bd = select_dyn(lb, Bl, Bg);
        } else {  // the all-LDS class
          A = load_dyn(s_body, aref);
          Bd = bref == kNone ? static_dyn() : load_dyn(s_body, bref);
        }
        solve_one(rec, A, Bd);
        if (slow_wave) {
          store_vel_off_sc1(rs, ga, A);
          store_vel_off_sc1(rs, gb, Bd);
          if (!(aref & kRefGlobal)) store_vel(s_body, aref, A);
          if (bref != kNone && !(bref & kRefGlobal)) store_vel(s_body, bref, Bd);
        } else {
          store_vel(s_body, aref, A);
          if (bref != kNone) store_vel(s_body, bref, Bd);
        }
        cons[c].nimp = rec.nimp;
        const bool gcounter = slot >= N0 && slot < N01;
        // re-arm (no arrival of the next iteration can come before this node's own releases); a class-1 slot counts
        // its in-block arrivals only: one, or none when both predecessors are outside
        s_cnt[slot] = gcounter ? (has_local ? 1u : 2u) : 0u;
        s_round[slot] = (uint8_t)(round + 1u);
        // velocities are out (LDS, write-through stores) before any successor hears of it; the all-LDS class has nothing
        // in flight to memory that a successor could read (the impulse is this constraint's own)
        if (slow_wave) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (TRACE) {  // (taken from the queue, released) + the slot's class in the low bits of the first stamp
          const uint32_t cls = slot < N0 ? 0u : (slot < N01 ? 1u : 2u);
          trace[2 * ((size_t)round * C_trace + c)] = (t_seen & ~3ull) | cls;
          trace[2 * ((size_t)round * C_trace + c) + 1] = wall_clock64();
        }
#pragma unroll
        for (int side = 0; side < 2; ++side) {
          if (side == 1 && bref == kNone) break;
          uint32_t w = side == 0 ? sw.x : sw.y;
          if (round + (w >> 31) >= iters) continue;
          uint32_t add = (w & kSuccTwo) ? 1u : 2u;
          if (w & kSuccLocal) {
            uint32_t sl = w & 0xFFFFu;
            uint32_t old = __hip_atomic_fetch_add(&s_cnt[sl], add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const bool cls1 = sl >= N0 && sl < N01;  // ready when the poller has seen the outside arrivals as well
            if (cls1 ? old + add == (2u | kF5RemoteDone) : old + add >= 2u) f5_push(sl < N0 ? qf : qs, sl);
          } else {
            __hip_atomic_fetch_add(&F.arr5[w & kSuccId], add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
      }
      if (lane == 0) __hip_atomic_fetch_sub(s_left, take, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      continue;
    }
    __builtin_amdgcn_s_sleep(2);
    if ((++spins & 255u) == 0u) {
      bool give_up = spins > spin_limit;
      if (give_up) __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (give_up || __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
    }
  }
  __syncthreads();
  // every node has run, so every arrival is in: re-arm the block's outside-arrival counters for the next Solver::solve
  // call on this constraint list (a tiled tick makes several)
  for (uint32_t idx = N0 + t; idx < N01; idx += kF5Threads) arr5[row0 + idx] = F.table[row0 + idx].cnt0 >> 8;
  // private bodies go back to the RigidBodyVec
  for (uint32_t p = p_lo + t; p < p_hi; p += kF5Threads) {
    uint32_t x = F.sidx[p];
    if (!F.shared[x]) {
      srec[4 * (size_t)x] = s_body[4 * (p - p_lo)];
      float4 s1 = s_body[4 * (p - p_lo) + 1];
      *reinterpret_cast<float2*>(&srec[4 * (size_t)x + 1]) = make_float2(s1.x, s1.y);
    }
  }
}

// ------------------------------------------------------------------------------------------
// Spatial tiling (one process per GPU): boundary selection, ghost export / import.
// Ghost record, 36 floats: x3 q4 v3 w3 delta3 | tag p3 d3 r | inv_mass I9 restitution friction.
// ------------------------------------------------------------------------------------------
constexpr int kGhostFloats = 36;

// flags[i] bit0: owned body i's fat box reaches below x_left; bit1: above x_right.
__global__ __launch_bounds__(kBlock) void k_boundary_flags(Bodies B, uint32_t n_owned, float x_left, float x_right, uint32_t* fl,
                                                           uint32_t* fr) {
  uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i > n_owned) return;
  uint32_t l = 0, r = 0;
  if (i < n_owned) {
    float c = B.fb_c[i].x, h = B.fb_r[i].x;
    l = (c - h < x_left) ? 1u : 0u;
    r = (c + h > x_right) ? 1u : 0u;
  }
  fl[i] = l; fr[i] = r;  // slot n_owned = 0 so the exclusive scan yields the total there
}
__global__ __launch_bounds__(kBlock) void k_boundary_scatter(uint32_t n_owned, const uint32_t* fl, const uint32_t* sl, const uint32_t* fr,
                                                             const uint32_t* sr, uint32_t* ids_l, uint32_t* ids_r) {
  uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n_owned) return;
  if (fl[i]) ids_l[sl[i]] = i;
  if (fr[i]) ids_r[sr[i]] = i;
}
__global__ __launch_bounds__(kBlock) void k_export_bodies(Bodies B, const uint32_t* ids, uint32_t m, float* out) {
  uint32_t t = blockIdx.x * kBlock + threadIdx.x;
  if (t >= m) return;
  uint32_t i = ids[t];
  float* o = out + (size_t)t * kGhostFloats;
  float4 x = B.x[i], q = B.q[i], s0 = B.srec[4 * i], s1 = B.srec[4 * i + 1], s2 = B.srec[4 * i + 2], s3 = B.srec[4 * i + 3];
  float4 d = B.delta[i], e = B.einfo[i], c0 = B.col0[i], c1 = B.col1[i];
  o[0] = x.x; o[1] = x.y; o[2] = x.z;
  o[3] = q.x; o[4] = q.y; o[5] = q.z; o[6] = q.w;
  o[7] = s0.x; o[8] = s0.y; o[9] = s0.z;
  o[10] = s0.w; o[11] = s1.x; o[12] = s1.y;
  o[13] = d.x; o[14] = d.y; o[15] = d.z;
  o[16] = c1.w; o[17] = c0.x; o[18] = c0.y; o[19] = c0.z; o[20] = c1.x; o[21] = c1.y; o[22] = c1.z; o[23] = c0.w;
  o[24] = s1.z;
  o[25] = s1.w; o[26] = s2.x; o[27] = s2.y; o[28] = s2.z; o[29] = s2.w; o[30] = s3.x; o[31] = s3.y; o[32] = s3.z; o[33] = s3.w;
  o[34] = e.w; o[35] = d.w;
}
__global__ __launch_bounds__(kBlock) void k_import_ghosts(Bodies B, uint32_t n_owned, uint32_t m, const float* in, float fat_margin) {
  uint32_t t = blockIdx.x * kBlock + threadIdx.x;
  if (t >= m) return;
  uint32_t i = n_owned + t;
  const float* o = in + (size_t)t * kGhostFloats;
  V3 x = mk3(o[0], o[1], o[2]), d = mk3(o[13], o[14], o[15]);
  B.x[i] = mk4(x, 0.0f);
  B.q[i] = make_float4(o[3], o[4], o[5], o[6]);
  B.srec[4 * i] = make_float4(o[7], o[8], o[9], o[10]);
  B.srec[4 * i + 1] = make_float4(o[11], o[12], o[24], o[25]);
  B.srec[4 * i + 2] = make_float4(o[26], o[27], o[28], o[29]);
  B.srec[4 * i + 3] = make_float4(o[30], o[31], o[32], o[33]);
  B.delta[i] = mk4(d, o[35]);
  B.einfo[i] = mk4(x + d, o[34]);  // RigidBodyInfo.x = x + delta (physics.rs:282)
  Comp k; k.kind = (int)f2u(o[16]); k.p = mk3(o[17], o[18], o[19]); k.d = mk3(o[20], o[21], o[22]); k.r = o[23];
  B.col0[i] = mk4(k.p, k.r);
  B.col1[i] = mk4(k.d, o[16]);
  Box tb = swept_bounds(k, d);
  B.tb_c[i] = mk4(tb.c, 0.0f); B.tb_r[i] = mk4(tb.r, 0.0f);
  B.fb_c[i] = mk4(tb.c, 0.0f); B.fb_r[i] = mk4(tb.r + mk3(fat_margin, fat_margin, fat_margin), 0.0f);
  B.sp0[i] = make_float4(0, 0, 0, o[34]); B.sp1[i] = make_float4(0, 0, 0, o[35]);
  B.ctor[i] = make_float4(o[16], k.r, 0.0f, 0.0f);
  B.imb[3 * i] = make_float4(0, 0, 0, 0); B.imb[3 * i + 1] = make_float4(0, 0, 0, 0); B.imb[3 * i + 2] = make_float4(0, 0, 0, 0);
}
// velocity record: 8 floats (v3, w3, 0, 0)
__global__ __launch_bounds__(kBlock) void k_export_vel(const float4* srec, const uint32_t* ids, uint32_t m, float4* out) {
  uint32_t t = blockIdx.x * kBlock + threadIdx.x;
  if (t >= m) return;
  uint32_t i = ids[t];
  float4 s0 = srec[4 * i], s1 = srec[4 * i + 1];
  out[2 * t] = s0;
  out[2 * t + 1] = make_float4(s1.x, s1.y, 0.0f, 0.0f);
}
__global__ __launch_bounds__(kBlock) void k_import_ghost_vel(float4* srec, uint32_t n_owned, uint32_t m, const float4* in) {
  uint32_t t = blockIdx.x * kBlock + threadIdx.x;
  if (t >= m) return;
  uint32_t i = n_owned + t;
  srec[4 * i] = in[2 * t];
  float2* p = reinterpret_cast<float2*>(&srec[4 * i + 1]);
  *p = make_float2(in[2 * t + 1].x, in[2 * t + 1].y);
}

// ---- migration of owned bodies between tiles -----------------------------------------------------
// A migrant record is the body's row of every Bodies array, verbatim (kMigrantWords float4 = 80 floats): the
// receiving tile continues bit-identically, persistent fat box and constructor tag (ctor.w) included.
constexpr int kMigrantWords = 20;
__device__ __forceinline__ float4* body_word(const Bodies& B, uint32_t e, uint32_t i) {
  switch (e) {
    case 0: return B.x + i;
    case 1: return B.q + i;
    case 2: case 3: case 4: case 5: return B.srec + 4 * (size_t)i + (e - 2);
    case 6: return B.sp0 + i;
    case 7: return B.sp1 + i;
    case 8: return B.ctor + i;
    case 9: case 10: case 11: return B.imb + 3 * (size_t)i + (e - 9);
    case 12: return B.delta + i;
    case 13: return B.einfo + i;
    case 14: return B.col0 + i;
    case 15: return B.col1 + i;
    case 16: return B.tb_c + i;
    case 17: return B.tb_r + i;
    case 18: return B.fb_c + i;
    default: return B.fb_r + i;
  }
}
// cnt[0] / cnt[1] += owned bodies whose centre lies below x_lo / at or above x_hi (the slab is [x_lo, x_hi))
__global__ __launch_bounds__(kBlock) void k_migrant_count(Bodies B, uint32_t n_owned, float x_lo, float x_hi, uint32_t* cnt) {
  uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n_owned) return;
  float cx = B.x[i].x;
  if (cx < x_lo) atomicAdd(cnt, 1u);
  else if (cx >= x_hi) atomicAdd(cnt + 1, 1u);
}
__global__ __launch_bounds__(kBlock) void k_migrant_flags(Bodies B, uint32_t n_owned, float x_lo, float x_hi, uint32_t* fl, uint32_t* fr) {
  uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i > n_owned) return;
  uint32_t l = 0, r = 0;
  if (i < n_owned) {
    float cx = B.x[i].x;
    l = (cx < x_lo) ? 1u : 0u;
    r = (!l && cx >= x_hi) ? 1u : 0u;
  }
  fl[i] = l; fr[i] = r;
}
__global__ __launch_bounds__(kBlock) void k_export_migrants(Bodies B, const uint32_t* ids, uint32_t m, float4* out) {
  uint32_t t = blockIdx.x * kBlock + threadIdx.x;
  if (t >= m * kMigrantWords) return;
  uint32_t b = t / kMigrantWords, e = t % kMigrantWords;
  out[t] = *body_word(B, e, ids[b]);
}
__global__ __launch_bounds__(kBlock) void k_import_migrants(Bodies B, uint32_t base, uint32_t m, const float4* in) {
  uint32_t t = blockIdx.x * kBlock + threadIdx.x;
  if (t >= m * kMigrantWords) return;
  uint32_t b = t / kMigrantWords, e = t % kMigrantWords;
  *body_word(B, e, base + b) = in[t];
}
// keep[i] = 1 for i < n, keep[n] = 0 (scan total); then the listed bodies are cleared
__global__ __launch_bounds__(kBlock) void k_keep_fill(uint32_t* keep, uint32_t n) {
  uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i <= n) keep[i] = i < n ? 1u : 0u;
}
__global__ __launch_bounds__(kBlock) void k_keep_clear(uint32_t* keep, const uint32_t* ids, uint32_t m, uint32_t n, uint32_t* err) {
  uint32_t t = blockIdx.x * kBlock + threadIdx.x;
  if (t >= m) return;
  uint32_t i = ids[t];
  if (i >= n || atomicExch(&keep[i], 0u) == 0u) atomicOr(err, 1u);  // out of range or listed twice
}
// stable compaction through a scratch copy: tmp[pos[i]] = row i for kept bodies, then rows [0, n_new) = tmp
__global__ __launch_bounds__(kBlock) void k_compact_gather(Bodies B, uint32_t n, const uint32_t* keep, const uint32_t* pos, float4* tmp) {
  uint32_t t = blockIdx.x * kBlock + threadIdx.x;
  if (t >= n * kMigrantWords) return;
  uint32_t i = t / kMigrantWords, e = t % kMigrantWords;
  if (keep[i]) tmp[(size_t)pos[i] * kMigrantWords + e] = *body_word(B, e, i);
}
__global__ __launch_bounds__(kBlock) void k_kind_mask(const float4* col1, uint32_t base, uint32_t m, uint32_t* mask) {
  uint32_t t = blockIdx.x * kBlock + threadIdx.x;
  if (t < m) atomicOr(mask, f2u(col1[base + t].w) == (uint32_t)KIND_SPHERE ? 1u : 2u);
}
__global__ __launch_bounds__(kBlock) void k_tags_set(float4* ctor, const uint32_t* tags, uint32_t n) {
  uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i < n) ctor[i].w = u2f(tags[i]);
}
__global__ __launch_bounds__(kBlock) void k_tags_get(const float4* ctor, uint32_t* tags, uint32_t n) {
  uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i < n) tags[i] = f2u(ctor[i].w);
}

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
      if (ka == MGF_TRIANGLE) return tri_mcapsule(Tr(a), Cp(b), vb, out);
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

}  // namespace mgf
