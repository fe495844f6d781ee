// Broadphase: Morton cells, the implicit 4-ary tree over them, grid and tree queries, terrain rows, candidate lists.  (Part of the kernel set described in kernels.h.)
#pragma once
#include "k_bodies.h"

namespace mgf {

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
__device__ __forceinline__ uint32_t face_cell(uint32_t cx, uint32_t cy, uint32_t cz, uint3 bits) {  // row-major cell of the face grid
  return (((cx << bits.y) | cy) << bits.z) | cz;
}
// The box the Morton coordinates are quantised over: the scene bounds with every axis widened (upwards) to at least
// `min_frac` of the longest one.  Every axis gets the same number of cells, so a scene much shorter along one axis - an x-slab
// tile of a wide pile - would otherwise get cells as thin along it, and a query's region many cells across (measured: 96
// instead of 50 us for 131 072 bodies in a 19 x 130 x 66 tile).  0 = the bounds as they are (a terrain mesh: flat on purpose).
constexpr int kMortonBits = 30;
constexpr float kBodyGridMinFrac = 0.5f;
// ... and (bodies only, min_frac > 0) to at least one largest-fat-half-extent per cell, if that takes no more than half as
// much again: a pile that has settled to two thirds of its height would otherwise get cells two thirds as high, every query's region
// would reach three cells up and down instead of two, and the staged box of k_pair_brick (two cells around its brick) would not hold
// it (measured: the settled pile fell back to k_pair_grid at 150 us).  `P` = prefix bits of the cells (2 x levels).
__device__ __forceinline__ void grid_box(const SceneBounds* sb, float min_frac, uint32_t P, float* lo, float* hi) {
  float ext = 0.0f;
#pragma unroll
  for (int k = 0; k < 3; ++k) { lo[k] = ord_f(sb->lo[k]); hi[k] = ord_f(sb->hi[k]); ext = fmaxf(ext, hi[k] - lo[k]); }
  if (!(min_frac > 0.0f)) return;
  const uint32_t nb[3] = {(P + 2u) / 3u, (P + 1u) / 3u, P / 3u};
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    float e = hi[k] - lo[k];
    // (r06) ... but not beyond cells of 0.4 of a query's reach along the axis (a query is about 3.4 largest fat half extents across: its own
    // tight box and the partners' fat ones): wider cells only hold more bodies - config 3's 337 x 60 x 87 capsule field widened to half its
    // length had 5.2 x 2.6 x 2.6 cells with twenty capsules each in the pile (k_pair_grid_n 117 us; 88 with cells of 1.4).  An x-slab
    // tile's thin axis and config 5's flat field end up where the plain half-of-the-longest rule put them.
    const float cell_cap = 1.36f * ord_f(sb->rmax[k]) * (float)(1u << nb[k]);
    if (e < min_frac * ext) e = fmaxf(e, fminf(min_frac * ext, cell_cap));
    const float want = ord_f(sb->rmax[k]) * (float)(1u << nb[k]) * 1.02f;
    if (e < want && want <= 1.5f * e) e = want;
    hi[k] = lo[k] + e;
  }
}
__device__ __forceinline__ uint32_t morton_cell_of(V3 c, const SceneBounds* sb, float min_frac, int shift) {
  float glo[3], ghi[3];
  grid_box(sb, min_frac, (uint32_t)(kMortonBits - shift), glo, ghi);
  uint32_t code = 0;
  for (int k = 0; k < 3; ++k) code |= expand10(morton_quant(at(c, k), glo[k], ghi[k])) << (2 - k);
  return code >> shift;
}
// Counting sort of the bodies into Morton cells (a cell = one 2L-bit prefix of the 30-bit code): cell of every
// body + its arrival rank inside the cell.  After a scan of the per-cell counts k_scatter_leaves places body i
// at cell_lo[cell] + rank.  The order INSIDE a cell is arrival order (it varies from run to run); nothing
// downstream depends on it - candidate rows are sorted by body index before they are used.
// `axis_bits` (the static mesh's face grid): when not all zero the cells are ROW-MAJOR with that many bits per axis instead of
// Morton prefixes - a flat mesh gives its thin axis no bits at all (see build_face_grid).
// `sb_part` (the fused tick: this tick's k_integrate gathered the scene bounds into partial records and nothing has folded them
// yet): every block folds them for itself - 64 x 9 words that sit in L2 - and block 0 also writes the result to *sb_out, where the
// kernels behind the scan and the tick's read-back find it.  Null: *sb is final.
__global__ __launch_bounds__(kBlock) void k_morton_count(const float4* fb_c, uint32_t n, const SceneBounds* sb, int shift, uint32_t* cell_of,
                                                         uint32_t* rank, uint32_t* cell_cnt, float min_frac, uint3 axis_bits,
                                                         const int* sb_part = nullptr, SceneBounds* sb_out = nullptr) {
  __shared__ SceneBounds s_sb;
  if (sb_part) {
    if (threadIdx.x < 9) {
      const int k = threadIdx.x;
      int v = sb_part[k];
      for (int a = 1; a < kBoundSlots; ++a) { const int u = sb_part[(size_t)a * kBoundSlotInts + k]; v = k < 3 ? min(v, u) : max(v, u); }
      if (k < 3) s_sb.lo[k] = v; else if (k < 6) s_sb.hi[k - 3] = v; else s_sb.rmax[k - 6] = v;
      if (blockIdx.x == 0) { if (k < 3) sb_out->lo[k] = v; else if (k < 6) sb_out->hi[k - 3] = v; else sb_out->rmax[k - 6] = v; }
    }
    __syncthreads();
    sb = &s_sb;
  }
  uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  V3 c = xyz(fb_c[i]);
  float glo[3], ghi[3];
  grid_box(sb, min_frac, (uint32_t)(kMortonBits - shift), glo, ghi);
  uint32_t cell;
  if (axis_bits.x + axis_bits.y + axis_bits.z) {
    cell = face_cell(morton_quant(c.x, glo[0], ghi[0]) >> (10u - axis_bits.x), morton_quant(c.y, glo[1], ghi[1]) >> (10u - axis_bits.y),
                     morton_quant(c.z, glo[2], ghi[2]) >> (10u - axis_bits.z), axis_bits);
  } else {
    uint32_t code = 0;
    for (int k = 0; k < 3; ++k) code |= expand10(morton_quant(at(c, k), glo[k], ghi[k])) << (2 - k);
    cell = code >> shift;
  }
  cell_of[i] = cell;
  rank[i] = atomicAdd(&cell_cnt[cell], 1u);
}

// Zero several small arrays with one launch (instead of one fill kernel each).
constexpr int kZeroSlots = 22;
struct ZeroList { uint32_t* p[kZeroSlots]; uint32_t words[kZeroSlots]; SceneBounds* sb; const int* sb_part; };
// one array of the list, by the whole grid: 16 bytes per lane where the array allows (the per-body arrays are a megabyte each)
__device__ __forceinline__ void zero_words(uint32_t* p, uint32_t words) {
  if (!p || !words) return;
  const uint32_t head = min(words, (uint32_t)((16u - ((uintptr_t)p & 15u)) & 15u) / 4u);  // words in front of the first 16-byte boundary
  const uint32_t quads = (words - head) / 4u;
  uint4* q = reinterpret_cast<uint4*>(p + head);
  for (uint32_t e = blockIdx.x * kBlock + threadIdx.x; e < quads; e += gridDim.x * kBlock) q[e] = make_uint4(0u, 0u, 0u, 0u);
  const uint32_t t = blockIdx.x * kBlock + threadIdx.x;
  if (t < head) p[t] = 0u;
  const uint32_t tail0 = head + 4u * quads;
  if (t < words - tail0) p[tail0 + t] = 0u;
}
__global__ __launch_bounds__(kBlock) void k_zero_many(ZeroList z) {
  if (z.sb_part && blockIdx.x == 0 && threadIdx.x < 9) {  // fold k_integrate's partial scene bounds (see there)
    const int k = threadIdx.x;
    int v = z.sb_part[k];
    for (int a = 1; a < kBoundSlots; ++a) { int u = z.sb_part[(size_t)a * kBoundSlotInts + k]; v = k < 3 ? min(v, u) : max(v, u); }
    if (k < 3) z.sb->lo[k] = v; else if (k < 6) z.sb->hi[k - 3] = v; else z.sb->rmax[k - 6] = v;
  }
  for (int a = 0; a < kZeroSlots; ++a) zero_words(z.p[a], z.words[a]);
}

// The fused tick's first launch: k_reset_step and k_zero_many in one, AHEAD of k_integrate (round 3: a launch less per tick).
// Nothing it clears is read by k_integrate; what k_integrate's tail writes (the owned bodies' terrain counts, the row-overflow flag)
// it writes afterwards.  The partial scene bounds are folded by k_morton_count instead.
// The tick's read-back without a copy engine and without an event (round 3): the block of counts, bounds and flags is written straight
// into the world's pinned host memory, then - behind a system-scope fence - the slot's sequence word; the host polls that word.
// (hipMemcpyAsync + hipEventRecord put a blit kernel and a barrier packet between two ticks: 11.6 us of an idle GPU per tick.)
__device__ __forceinline__ void publish(const uint32_t* rb, uint32_t* pin, uint32_t words, uint32_t seq_word, uint32_t seq, const uint32_t* sb_words, uint32_t* grid_words) {
  // (the tick's scene bounds: the box the NEXT tick's k_integrate quantises its Morton cells over - CellSort)
  if (grid_words && threadIdx.x < sizeof(SceneBounds) / 4) grid_words[threadIdx.x] = sb_words[threadIdx.x];
  for (uint32_t i = threadIdx.x; i < words; i += kBlock) pin[i] = rb[i];
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(pin + seq_word, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ __launch_bounds__(kBlock) void k_publish(const uint32_t* rb, uint32_t* pin, uint32_t words, uint32_t seq_word, uint32_t seq, const uint32_t* sb_words = nullptr,
                                                    uint32_t* grid_words = nullptr) {
  publish(rb, pin, words, seq_word, seq, sb_words, grid_words);
}
// ... of several worlds on one stream in one launch (the tile set: a workgroup per world)
struct PublishOne { const uint32_t* rb; uint32_t* pin; uint32_t words, seq_word, seq, pad; const uint32_t* sb_words; uint32_t* grid_words; };
struct PublishBatch { PublishOne t[kWorldBatch]; };
__global__ __launch_bounds__(kBlock) void k_publish_batch(PublishBatch A) {
  const PublishOne& P = A.t[blockIdx.x];
  publish(P.rb, P.pin, P.words, P.seq_word, P.seq, P.sb_words, P.grid_words);
}
__global__ __launch_bounds__(kBlock) void k_tick_clear(ZeroList z, SceneBounds* sb, uint32_t* err, uint32_t* guard, const uint32_t* prev_fail, int spec,
                                                       int* sb_part) {
  // (see k_reset_step: a speculative tick behind a failed one raises the guard and resets nothing - but the counters are cleared all
  // the same: the kernels of the cell sort run unguarded, on the unchanged bodies, and must start from zero)
  // (err[2]: the solvers' abort flag - a tick whose persistent launch gave up is solved again by the host, solver_abort_fallback; this
  // launch clears the flag below, its first wave has read it here)
  const bool skip = spec && (*prev_fail || err[2]);
  if (skip && blockIdx.x == 0 && threadIdx.x == 0) *guard = 1u;
  if (blockIdx.x == 0 && !skip) {
    if (sb_part && threadIdx.x < kBoundSlots) {
      int* slot = sb_part + (size_t)threadIdx.x * kBoundSlotInts;
      for (int k = 0; k < 3; ++k) { slot[k] = 0x7FFFFFFF; slot[3 + k] = (int)0x80000000; slot[6 + k] = 0; }
    }
    if (threadIdx.x == 0) {
      *guard = 0u;
      for (int k = 0; k < 3; ++k) { sb->lo[k] = 0x7FFFFFFF; sb->hi[k] = (int)0x80000000; }
      sb->n_refits = 0; sb->pad = 0; sb->pad2 = 0;
      for (int k = 0; k < 3; ++k) sb->rmax[k] = 0;
      err[0] = 0; err[1] = 0; err[8] = 0;
    }
  }
  for (int a = 0; a < kZeroSlots; ++a) zero_words(z.p[a], z.words[a]);
}

struct StepCounts {
  uint32_t Mt, Mp, C, Ct;                      // effective: terrain / pair candidates, constraints, terrain constraints
  uint32_t fail;                               // kFail* bits
  uint32_t need_Mt, need_Mp, need_C, need_Ct;  // actual sizes (valid up to the first failing stage)
  uint32_t bins[6];                            // candidates per shape-pair type (scenes mixing spheres and capsules)
  uint32_t ct_sum;                             // terrain constraints, accumulated by k_count_contacts (zeroed by the candidate scan's epilogue, caps_candidates)
};
constexpr uint32_t kFailCandCap = 1u, kFailConsCap = 2u, kFailRowOverflow = 4u, kFailGridWide = 8u, kFailTerrainRow = 16u, kFailTerrainWide = 32u,
                   kFailRevRow = 64u,  // a body's row of `b` occurrences overflowed (k_setup_pairs / k_chain_rows)
                   kFailSkipped = 128u,  // a speculative tick behind a failed one: nothing was done (k_reset_step)
                   kFailWide = 512u,     // more wide bodies than their list holds (k_integrate, WideSpec): the tick is run again without the list
                   kFailFlow6 = 256u;    // the block-local solver's tables did not fit (k_flow6_*): the solve did nothing; the tick is re-run with the global dataflow solver

// The list sizes of the tick, checked against the capacities the lists were allocated with, at the end of the scans that
// produce them (the thread of k_scan that writes the last prefix runs these: no launch of their own).
struct ScanEpilogue {
  int kind;                       // 0 none, 1 candidate lists (after the scan of the terrain / partner rows), 2 constraint list, 3 both at once (k_contacts_spheres' tick: no candidate lists)
  uint32_t cap_a, cap_b;          // kind 1: cap_t, cap_p; kind 2: cap_c; kind 3: cap_t, cap_c
  const uint32_t *row_overflow, *grid_wide, *terrain_wide, *guard;  // kinds 1, 3 (each may be null but guard)
  StepCounts* sc;
  const uint32_t *sum_t, *sum_ct; // kind 3: terrain candidates and terrain contacts (k_terrain_contacts' counters)
  const uint32_t* wide_n;         // kinds 1, 3: the wide bodies listed (null: no list this tick)
  uint32_t sum_t_parts, sum_t_stride;  // kind 3: the terrain candidates are the sum of this many partial counts, that many words apart (0: the one word)
};
__device__ __forceinline__ void caps_candidates(const ScanEpilogue& E, uint32_t mt, uint32_t mp) {
  StepCounts r;
  r.need_Mt = mt; r.need_Mp = mp; r.need_C = 0; r.need_Ct = 0;
  r.fail = 0;
  if (r.need_Mt > E.cap_a || r.need_Mp > E.cap_b) r.fail |= kFailCandCap;
  if (E.row_overflow && (*E.row_overflow & 1u)) r.fail |= kFailRowOverflow;
  if (E.row_overflow && (*E.row_overflow & 2u)) r.fail |= kFailTerrainRow;
  if (E.grid_wide && *E.grid_wide) r.fail |= kFailGridWide;
  if (E.terrain_wide && *E.terrain_wide) r.fail |= kFailTerrainWide;
  if (E.wide_n && *E.wide_n > kWideCap) r.fail |= kFailWide;
  if (*E.guard) r.fail |= kFailSkipped;
  r.Mt = r.fail ? 0u : r.need_Mt; r.Mp = r.fail ? 0u : r.need_Mp; r.C = 0; r.Ct = 0;
  for (int k = 0; k < 6; ++k) r.bins[k] = 0;
  r.ct_sum = 0;
  *E.sc = r;
}
// kind 3: the scanned counts were the bodies' constraint counts themselves (contacts only in the partner rows, terrain contacts counted by
// k_terrain_contacts): the tick's counts in one go
__device__ __forceinline__ void caps_contacts(const ScanEpilogue& E, uint32_t c) {
  StepCounts r;
  uint32_t mt = *E.sum_t;
  for (uint32_t k = 1; k < E.sum_t_parts; ++k) mt += E.sum_t[(size_t)k * E.sum_t_stride];
  const uint32_t ct = *E.sum_ct;
  r.need_Mt = mt; r.need_Mp = c - ct; r.need_C = c; r.need_Ct = ct;
  r.fail = 0;
  if (E.row_overflow && (*E.row_overflow & 4u)) { r.fail |= kFailCandCap; r.need_Mt = max(mt, 2u * E.cap_a); }  // (k_terrain_near: a region of the slots ran full)
  if (E.row_overflow && (*E.row_overflow & 1u)) r.fail |= kFailRowOverflow;
  if (E.row_overflow && (*E.row_overflow & 2u)) r.fail |= kFailTerrainRow;
  if (E.grid_wide && *E.grid_wide) r.fail |= kFailGridWide;
  if (E.terrain_wide && *E.terrain_wide) r.fail |= kFailTerrainWide;
  if (E.wide_n && *E.wide_n > kWideCap) r.fail |= kFailWide;
  if (*E.guard) r.fail |= kFailSkipped;
  if (!r.fail && mt > E.cap_a) r.fail |= kFailCandCap;
  if (!r.fail && c > E.cap_b) r.fail |= kFailConsCap;
  r.Mt = r.fail ? 0u : mt; r.Mp = r.fail ? 0u : r.need_Mp; r.C = r.fail ? 0u : c; r.Ct = r.fail ? 0u : ct;
  for (int k = 0; k < 6; ++k) r.bins[k] = 0;
  r.ct_sum = ct;
  *E.sc = r;
}
__device__ __forceinline__ void caps_constraints(const ScanEpilogue& E, uint32_t c) {
  StepCounts* sc = E.sc;
  if (sc->fail) return;
  sc->need_C = c; sc->need_Ct = sc->ct_sum;
  if (c > E.cap_a) { sc->fail |= kFailConsCap; sc->Mt = 0; sc->Mp = 0; sc->C = 0; sc->Ct = 0; return; }
  sc->C = c; sc->Ct = sc->ct_sum;
}

// Device-wide exclusive prefix sum in ONE launch (the tick's three scans: cells, candidate rows, constraints per body).
// Tiles of kScanTile items are handed out through a ticket (so a tile's predecessors are always running: no assumption about
// dispatch order); a block publishes its tile's total at once and then looks back - one wave, 64 predecessors per step -
// until it meets a tile whose inclusive prefix is known (decoupled look-back).  The status words and the ticket are
// zeroed by the tick's clearing launch (k_zero_many).  W = 2 scans two arrays of equal length together (both totals share a
// status word: 31 bits each).
#ifndef MGF_SCAN_ROUNDS
#define MGF_SCAN_ROUNDS 4
#endif
constexpr int kScanBlock = 256, kScanRounds = MGF_SCAN_ROUNDS, kScanTile = kScanBlock * 4 * kScanRounds;  // 4096 items per tile
constexpr unsigned long long kScanAgg = 1ull << 62, kScanInc = 2ull << 62, kScanFlag = 3ull << 62;
struct ScanJob { const uint32_t* in[2]; uint32_t* out[2]; uint32_t n; unsigned long long* status; uint32_t* ticket; ScanEpilogue epi;
                 const uint32_t* add;   // add (W = 1, or null): a second array of the same length, summed into the first item by item
                 const int* sb_part; SceneBounds* sb_out; };  // optional side job of the last workgroup's idle fourth wave: fold k_integrate's partial scene bounds (k_morton_count did)
template <int W>
__global__ __launch_bounds__(kScanBlock) void k_scan(ScanJob J) {
  __shared__ uint32_t s_tile;
  __shared__ uint32_t s_wave[W][kScanBlock / 64];
  __shared__ uint32_t s_prev[W];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  // (a ticket, always: a tile's predecessors are then running whatever the order workgroups start in and however few fit on the device at once
  // - a part with fewer CUs, a CU mask, a device shared with another process; r05 took the block's own index for launches of up to 1024
  // workgroups, which relies on all of them being resident or dispatched in order: ADVICE r5)
  if (t == 0) s_tile = atomicAdd(J.ticket, 1u);
  if (J.sb_part && blockIdx.x == gridDim.x - 1 && wv == kScanBlock / 64 - 1) {  // (one wave of the launch, beside its first loads: a partial record per lane)
    static_assert(kBoundSlots == 64, "a lane per partial record");
    int v[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) v[k] = J.sb_part[(size_t)lane * kBoundSlotInts + k];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) { const int u = __shfl_xor(v[k], o); v[k] = k < 3 ? min(v[k], u) : max(v[k], u); }
    }
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < 3; ++k) { J.sb_out->lo[k] = v[k]; J.sb_out->hi[k] = v[3 + k]; J.sb_out->rmax[k] = v[6 + k]; }
    }
  }
  __syncthreads();
  const uint32_t tile = s_tile;
  const uint32_t i0 = tile * (uint32_t)kScanTile;
  // the tile in registers: kScanRounds rounds of 4 consecutive items per thread (one 16-byte load per lane, coalesced)
  uint4 v[W][kScanRounds];
  uint32_t tot[W];
#pragma unroll
  for (int a = 0; a < W; ++a) {
    tot[a] = 0;
#pragma unroll
    for (int r = 0; r < kScanRounds; ++r) {
      const uint32_t i = i0 + (uint32_t)r * (kScanBlock * 4) + (uint32_t)t * 4u;
      uint4 x = make_uint4(0, 0, 0, 0);
      if (i + 3u < J.n) x = *reinterpret_cast<const uint4*>(J.in[a] + i);
      else { if (i < J.n) x.x = J.in[a][i]; if (i + 1u < J.n) x.y = J.in[a][i + 1]; if (i + 2u < J.n) x.z = J.in[a][i + 2]; }
      if (W == 1 && J.add) {
        uint4 y = make_uint4(0, 0, 0, 0);
        if (i + 3u < J.n) y = *reinterpret_cast<const uint4*>(J.add + i);
        else { if (i < J.n) y.x = J.add[i]; if (i + 1u < J.n) y.y = J.add[i + 1]; if (i + 2u < J.n) y.z = J.add[i + 2]; }
        x.x += y.x; x.y += y.y; x.z += y.z; x.w += y.w;
      }
      v[a][r] = x;
      tot[a] += x.x + x.y + x.z + x.w;
    }
  }
  // the tile's total
#pragma unroll
  for (int a = 0; a < W; ++a) {
    uint32_t u = tot[a];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) u += __shfl_xor(u, o);
    if (lane == 0) s_wave[a][wv] = u;
  }
  __syncthreads();
  uint32_t agg[W];
#pragma unroll
  for (int a = 0; a < W; ++a) { agg[a] = 0; for (int k = 0; k < kScanBlock / 64; ++k) agg[a] += s_wave[a][k]; }
  auto pack = [](const uint32_t* x) -> unsigned long long { return W == 1 ? (unsigned long long)x[0] : ((unsigned long long)x[0] | ((unsigned long long)x[W - 1] << 31)); };
  if (wv == 0) {
    uint32_t prev[W];
#pragma unroll
    for (int a = 0; a < W; ++a) prev[a] = 0;
    if (tile == 0) {
      if (lane == 0) __hip_atomic_store(&J.status[0], kScanInc | pack(agg), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      if (lane == 0) __hip_atomic_store(&J.status[tile], kScanAgg | pack(agg), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      int back = (int)tile - 1;  // nearest predecessor not yet summed
      for (;;) {
        const int idx = back - lane;
        unsigned long long st = kScanInc;  // (lanes before tile 0 read as "inclusive, 0")
        if (idx >= 0) {
          do { st = __hip_atomic_load(&J.status[idx], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT); } while ((st & kScanFlag) == 0ull);
        }
        const unsigned long long inc = __ballot((st & kScanFlag) == kScanInc);
        const int first = inc ? __builtin_ctzll(inc) : 64;  // lanes 0..first contribute (first = the nearest inclusive prefix)
        uint32_t c[W];
        c[0] = lane <= first ? (uint32_t)(W == 1 ? (st & 0xFFFFFFFFull) : (st & 0x7FFFFFFFull)) : 0u;
        if (W == 2) c[W - 1] = lane <= first ? (uint32_t)((st >> 31) & 0x7FFFFFFFull) : 0u;
#pragma unroll
        for (int a = 0; a < W; ++a) {
          uint32_t u = c[a];
#pragma unroll
          for (int o = 32; o >= 1; o >>= 1) u += __shfl_xor(u, o);
          prev[a] += u;
        }
        if (inc) break;
        back -= 64;
      }
      uint32_t incl[W];
#pragma unroll
      for (int a = 0; a < W; ++a) incl[a] = prev[a] + agg[a];
      if (lane == 0) __hip_atomic_store(&J.status[tile], kScanInc | pack(incl), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (lane == 0) {
#pragma unroll
      for (int a = 0; a < W; ++a) s_prev[a] = prev[a];
    }
  }
  __syncthreads();
  // exclusive prefixes inside the tile: round by round, wave scan + the waves before + the rounds before + the tiles before
  uint32_t last[W];
  bool is_last = false;
#pragma unroll
  for (int a = 0; a < W; ++a) last[a] = 0;
#pragma unroll
  for (int a = 0; a < W; ++a) {
    uint32_t carry = s_prev[a];
#pragma unroll
    for (int r = 0; r < kScanRounds; ++r) {
      const uint4 x = v[a][r];
      const uint32_t mine = x.x + x.y + x.z + x.w;
      uint32_t inc = mine;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { const uint32_t u = __shfl_up(inc, o); if (lane >= o) inc += u; }
      __syncthreads();  // (s_wave of the previous use has been read)
      if (lane == 63) s_wave[a][wv] = inc;
      __syncthreads();
      uint32_t before = carry, round_total = 0;
      for (int k = 0; k < kScanBlock / 64; ++k) { const uint32_t u = s_wave[a][k]; if (k < wv) before += u; round_total += u; }
      const uint32_t e0 = before + inc - mine;
      const uint32_t i = i0 + (uint32_t)r * (kScanBlock * 4) + (uint32_t)t * 4u;
      const uint4 o4 = make_uint4(e0, e0 + x.x, e0 + x.x + x.y, e0 + x.x + x.y + x.z);
      if (i + 3u < J.n) *reinterpret_cast<uint4*>(J.out[a] + i) = o4;
      else { if (i < J.n) J.out[a][i] = o4.x; if (i + 1u < J.n) J.out[a][i + 1] = o4.y; if (i + 2u < J.n) J.out[a][i + 2] = o4.z; }
      if (J.n - 1u - i < 4u) { const uint32_t k = J.n - 1u - i; last[a] = k == 0 ? o4.x : k == 1 ? o4.y : k == 2 ? o4.z : o4.w; is_last = true; }
      carry += round_total;
    }
  }
  // the thread that wrote the last prefix (= the sum of everything before the closing element) checks the list sizes
  if (is_last && J.epi.kind == 1) caps_candidates(J.epi, last[0], last[W - 1]);
  if (is_last && J.epi.kind == 2) caps_constraints(J.epi, last[0]);
  if (is_last && J.epi.kind == 3) caps_contacts(J.epi, last[0]);
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
// `order id`: what decides `j < i` (world.rs:266) and every other order the Gauss-Seidel sequence depends on.  The body store may be
// kept in an internal (cell) order - slot s holds the caller's body ext[s], see host_perm.inc - and then the caller's index is the
// order id; with the store in the caller's own order (ext = null) it is the slot itself.
struct LeafRec { float4 c, r; };           // fat box centre | body slot, half extents | order id (sorted order)
__device__ __forceinline__ uint32_t order_id(const uint32_t* ext, uint32_t slot) { return ext ? ext[slot] : slot; }
struct Lbvh {
  QNode* nodes;        // (4^levels - 1) / 3 internal nodes
  LeafRec* leaves;     // n records in Morton order
  uint32_t* sidx;      // body index of every leaf record
  float4* lcol;        // optional, 2 per leaf record: collider (p.xyz, r) and motion (delta.xyz) of the body (k_pair_grid<true>)
  float4* ltb;         // optional, 2 per leaf record: the body's tight box = its query, and in the .w words the cells that query has to
                       // look at: (centre, ca packed 10 bits per axis), (half extents, d packed likewise) (k_pair_brick)
  uint32_t* cell_lo;   // 4^levels + 1 entries: cell c holds the leaf records [cell_lo[c], cell_lo[c + 1])
  const uint32_t* ext; // slot -> order id (the caller's body index), or null: the store is in the caller's order
  uint32_t n;          // live bodies
  uint32_t levels;     // internal levels L >= 4; 4^L leaf cells
  uint32_t* err;
  unsigned long long* dbg;  // optional: [0] node fetches, [1] leaf records tested, [2] max fetches of one query
};
constexpr int kLdsQNodes = 341;  // levels 0..4 (1 + 4 + 16 + 64 + 256 nodes), 43 KB
__host__ __device__ __forceinline__ uint32_t qlevel_offset(uint32_t l) { return ((1u << (2 * l)) - 1u) / 3u; }

__device__ __forceinline__ void box_min_max(V3& lo, V3& hi, V3 l, V3 h) {
  lo = mk3(fminf(lo.x, l.x), fminf(lo.y, l.y), fminf(lo.z, l.z));
  hi = mk3(fmaxf(hi.x, h.x), fmaxf(hi.y, h.y), fmaxf(hi.z, h.z));
}

// The cells a query (a body's tight box) has to look at in the grid broadphase: a body j can only be accepted by query i
// (tight_i overlaps fat_j) if its fat-box centre lies within tight_i grown by the largest fat half extent of the scene.
// [ca, ca + d) per axis, in cells of nb[k] prefix bits.
__device__ __forceinline__ float pair_query_pad(const V3& c, const V3& r, float pad_abs) {
  return pad_abs + 1e-5f * (fabs_rs(c.x) + fabs_rs(c.y) + fabs_rs(c.z) + r.x + r.y + r.z);
}
// (`box`: the bounds the cells were quantised over - this tick's, or the previous tick's when k_integrate did the counting sort's first
// half; the half width rmax is always this tick's)
__device__ __forceinline__ void pair_query_region(const V3& qc, const V3& qr, float pad, const SceneBounds* sb, const uint32_t* nb, uint32_t* ca, uint32_t* d,
                                                  float min_frac, const SceneBounds* box = nullptr) {
  float glo[3], ghi[3];
  grid_box(box ? box : sb, min_frac, nb[0] + nb[1] + nb[2], glo, ghi);
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    float lo = glo[k], hi = ghi[k], rm = ord_f(sb->rmax[k]);
    float a = at(qc, k) - at(qr, k) - rm - pad, b = at(qc, k) + at(qr, k) + rm + pad;
    uint32_t c0 = morton_quant(a, lo, hi) >> (10u - nb[k]), c1 = morton_quant(b, lo, hi) >> (10u - nb[k]);
    ca[k] = c0; d[k] = c1 - c0 + 1u;
  }
}

// Bodies -> leaf records in cell order (counting sort, second half).
__device__ __forceinline__ void scatter_leaf(uint32_t body, const Lbvh& T, const float4* fb_c, const float4* fb_r, const uint32_t* cell_of,
                                             const uint32_t* rank, uint32_t* brank, const float4* col0, const float4* delta, const float4* tb_c,
                                             const float4* tb_r, const SceneBounds* sb, float pad_abs, float min_frac, const SceneBounds* box = nullptr,
                                             float3 wide_limit = make_float3(0.0f, 0.0f, 0.0f), uint32_t n_owned = 0u) {
  if (body >= T.n) return;
  uint32_t p = T.cell_lo[cell_of[body]] + rank[body];
  LeafRec lr; lr.c = mk4(xyz(fb_c[body]), u2f(body)); lr.r = mk4(xyz(fb_r[body]), u2f(order_id(T.ext, body)));
  // (a wide body - WideSpec, k_bodies.h - is never a partner of the grid's pair search: no order id is below 0xFFFFFFFF.  k_pair_wide pairs it.)
  if (wide_limit.x > 0.0f && body < n_owned) { const float lim[3] = {wide_limit.x, wide_limit.y, wide_limit.z}; if (is_wide(lim, fb_r[body])) lr.r.w = u2f(0xFFFFFFFFu); }
  T.leaves[p] = lr;
  if (T.lcol) { T.lcol[2 * p] = col0[body]; T.lcol[2 * p + 1] = delta[body]; }
  if (T.ltb) {
    const float4 c = tb_c[body], r = tb_r[body];
    const uint32_t P = 2u * T.levels;
    const uint32_t nb[3] = {(P + 2u) / 3u, (P + 1u) / 3u, P / 3u};
    uint32_t ca[3], d[3];
    pair_query_region(xyz(c), xyz(r), pair_query_pad(xyz(c), xyz(r), pad_abs), sb, nb, ca, d, min_frac, box);
    T.ltb[2 * p] = mk4(xyz(c), u2f(ca[0] | (ca[1] << 10) | (ca[2] << 20)));  // (ca < 1024, d <= 1024 - ca)
    T.ltb[2 * p + 1] = mk4(xyz(r), u2f(min(d[0], 1023u) | (min(d[1], 1023u) << 10) | (min(d[2], 1023u) << 20)));  // (1023 cells across is "too wide" like 1024)
  }
  T.sidx[p] = body;
  brank[body] = p;  // position in cell order (the block-local solver groups bodies by it)
}
__global__ __launch_bounds__(kBlock) void k_scatter_leaves(Lbvh T, const float4* fb_c, const float4* fb_r, const uint32_t* cell_of,
                                                           const uint32_t* rank, uint32_t* brank, const float4* col0, const float4* delta, const float4* tb_c,
                                                           const float4* tb_r, const SceneBounds* sb, float pad_abs, float min_frac, const SceneBounds* box = nullptr,
                                                           float3 wide_limit = make_float3(0.0f, 0.0f, 0.0f), uint32_t n_owned = 0u) {
  scatter_leaf(blockIdx.x * kBlock + threadIdx.x, T, fb_c, fb_r, cell_of, rank, brank, col0, delta, tb_c, tb_r, sb, pad_abs, min_frac, box, wide_limit, n_owned);
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
// `nodes`: the tree's nodes as float4 pairs, in global memory or (small meshes, k_integrate's tail) staged in LDS
template <class F>
__device__ __forceinline__ void terrain_traverse_at(const TerrainDev& M, const float4* nodes, const Box& q, F&& emit) {
  if (M.n_nodes == 0) return;
  uint32_t stack[kStack];
  int sp = 0;
  stack[sp++] = M.root;
  while (sp > 0) {
    uint32_t top = stack[--sp];
    const float4* raw = nodes + 2 * (size_t)top;
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

template <class F>
__device__ __forceinline__ void terrain_traverse(const TerrainDev& M, const Box& q, F&& emit) {
  terrain_traverse_at(M, reinterpret_cast<const float4*>(M.nodes), q, emit);
}

// Depth-first traversal of the implicit 4-ary tree with a bitmask trail (4 pending-child bits per level)
// instead of a stack.  `top` = LDS copy of nodes [0, kLdsQNodes).
template <class F>
__device__ __forceinline__ void lbvh_traverse(const Lbvh& T, const QNode* top, uint32_t oi /* the query's order id */, const Box& q, float pad_abs, F&& emit) {
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
            if (pb + e < p1 && f2u(lr[e].r.w) < oi) {  // world.rs:266
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
  const uint32_t oi = order_id(T.ext, i);
  if (oi != 0) {  // world.rs:256
    lbvh_traverse(T, s_top, oi, q, pad, [&](uint32_t j) {
      if (j >= n_owned) return;  // ghost-ghost: the owners' business
      if (FILL) { p_cand[pb + np] = j; p_owner[pb + np] = i; }
      ++np;
    });
  }
  if (!FILL) { t_cnt[i] = nt; p_cnt[i] = np; return; }
  // canonical partner order: ascending j (insertion sort, segments are ~10 long)
  for (uint32_t a = 1; a < np; ++a) {
    uint32_t v = p_cand[pb + a];
    const uint32_t ov = order_id(T.ext, v);
    uint32_t b = a;
    while (b > 0 && order_id(T.ext, p_cand[pb + b - 1]) > ov) { p_cand[pb + b] = p_cand[pb + b - 1]; --b; }
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

// the same rows written from k_integrate's tail (no second pass over the bodies)
struct TerrainRowsTail {
  static constexpr int kLdsWords = 2 * 64;  // a mesh of up to 64 tree nodes (the demo's 10-face box: 19) is walked in LDS
  TerrainDev M; uint32_t cap_row; uint32_t* rows_t; uint32_t* t_cnt; uint32_t* overflow;
  // optional: the bodies that list a face, compacted, each with what the sphere-triangle test needs of it - k_terrain_contacts works on
  // these records alone.  3 words per body: (collider p, r), (motion, body), (faces listed, the first eight of them as bytes - or
  // 0xFFFFFFFF: read the row -, -)
  float4* near_list; uint32_t* near_cnt;
  __device__ __forceinline__ void stage(float4* s) const {
    if (2u * M.n_nodes > (uint32_t)kLdsWords) return;
    const float4* src = reinterpret_cast<const float4*>(M.nodes);
    for (uint32_t e = threadIdx.x; e < 2u * M.n_nodes; e += blockDim.x) s[e] = src[e];
  }
  static constexpr bool kNear = true;
  // -> near_nt = the faces listed (0: none), near_pk = the first eight of them as bytes (~0: read the row); k_integrate appends the record
  __device__ __forceinline__ void operator()(uint32_t i, const Box& tb, const float4* s, uint32_t& near_nt, unsigned long long& near_pk) const {
    Box q; q.c = tb.c + -mk3(M.x[0], M.x[1], M.x[2]); q.r = tb.r;
    uint32_t* row = rows_t + (size_t)i * cap_row;
    uint32_t nt = 0, big = 0;
    unsigned long long pk = 0ull;
    const uint32_t cap = cap_row;
    auto emit = [&](uint32_t face) {
      if (nt < cap) row[nt] = face;
      big |= face;
      pk |= nt < 8u ? (unsigned long long)face << (8u * nt) : 0ull;
      ++nt;
    };
    if (2u * M.n_nodes <= (uint32_t)kLdsWords) terrain_traverse_at(M, s, q, emit);  // (two calls: a selected pointer would make the loads flat)
    else terrain_traverse(M, q, emit);
    t_cnt[i] = nt;
    if (nt > cap) atomicOr(overflow, 2u);
    if (big > 255u || nt > 8u) pk = ~0ull;
    near_nt = near_list ? nt : 0u; near_pk = pk;
  }
};

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
  const uint32_t oi = order_id(T.ext, i);
  if (oi != 0 && T.n >= 2) {  // world.rs:256
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
            const uint32_t oj = f2u(__shfl(w.w, lane | 1));  // the record's order id rides with the half extents
            uint32_t j = f2u(w.w);
            bool hit = false;
            if (!(sub & 1) && pb + (uint32_t)(sub >> 1) < p1 && oj < oi && j < n_owned) {  // world.rs:266; ghost-ghost skipped
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
// ... and the body's motion with it, from the packed copy when the host vouches for it: one 64-byte sector instead of three
// look-ups in three arrays (the narrowphase kernels are bound by the rate of such look-ups, not by their arithmetic)
__device__ __forceinline__ Comp load_comp_moving(const Bodies& B, uint32_t i, V3* v) {
  float4 c0, c1, dl;
  if (B.bpk) { c0 = B.bpk[4 * (size_t)i]; dl = B.bpk[4 * (size_t)i + 1]; c1 = B.bpk[4 * (size_t)i + 3]; }
  else { c0 = B.col0[i]; c1 = B.col1[i]; dl = B.delta[i]; }
  Comp k; k.p = xyz(c0); k.r = c0.w; k.d = xyz(c1); k.kind = (int)f2u(c1.w);
  *v = xyz(dl);
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
// accepted partners are still counted (World::step's candidate statistic): per block into one of kPairStatWords words.
constexpr uint32_t kGridMaxCells = 512;
constexpr uint32_t kPairStatWords = 1024;  // partial sums of the accepted partners (same-word atomics from many CUs serialise: ~0.3 us each)

// Where a query's cells and leaf records come from: global memory (the sorted arrays themselves) ...
struct PairSrcGlobal {
  Lbvh T;
  __device__ __forceinline__ void range(uint32_t cell, const uint32_t*, uint32_t& p0, uint32_t& p1) const { p0 = T.cell_lo[cell]; p1 = T.cell_lo[cell + 1]; }
  __device__ __forceinline__ void leaf(uint32_t p, float4& c, float4& r) const { LeafRec lr = T.leaves[p]; c = lr.c; r = lr.r; }
  __device__ __forceinline__ void col(uint32_t p, float4& c0, float4& d0, uint32_t& j) const { c0 = T.lcol[2 * p]; d0 = T.lcol[2 * p + 1]; j = T.sidx[p]; }
};
// One query (8 lanes) over the cells [ca, ca + d): the reference's acceptance test on every leaf record of those cells, and -
// SPHERES - the sphere-sphere test on the accepted ones (staged in `acc`).  np = entries written to the row; n_accepted = fat-box
// partners.  The lanes of a group stay together (ballots).
template <bool SPHERES, class Src, class AccT>
__device__ __forceinline__ void pair_query_cells(const Src& S, const Box& q, const Comp& A, V3 vA, uint32_t oi /* the query's order id */, uint32_t n_owned, const uint32_t* ca,
                                                 const uint32_t* d, const uint32_t* nb, int shift, int sub, int gbase, uint32_t* row, AccT* acc,
                                                 uint32_t* overflow, uint32_t& np, uint32_t& n_accepted) {
  const uint32_t ncell = d[0] * d[1] * d[2];
  // idx -> (cx, cy, cz) by two multiplications: ncell <= kGridMaxCells = 512, so idx * d < 2^18 and the 20-bit reciprocals are exact
  const uint32_t m2 = ((1u << 20) + d[2] - 1u) / d[2], m1 = ((1u << 20) + d[1] - 1u) / d[1];
  np = 0;
  for (uint32_t cb = 0; cb < ncell; cb += kCoopLanes) {
    uint32_t idx = cb + (uint32_t)sub;
    uint32_t p0 = 0, p1 = 0;
    if (idx < ncell) {
      const uint32_t t = (idx * m2) >> 20, cz = idx - t * d[2];
      const uint32_t cx = (t * m1) >> 20, cy = t - cx * d[1];
      const uint32_t cc3[3] = {ca[0] + cx, ca[1] + cy, ca[2] + cz};
      uint32_t code = (expand10(cc3[0] << (10u - nb[0])) << 2) | (expand10(cc3[1] << (10u - nb[1])) << 1) | expand10(cc3[2] << (10u - nb[2]));
      S.range(code >> shift, cc3, p0, p1);
    }
    // every lane walks its own cell's bodies; the group stays together for the ballots
    for (;;) {
      bool more = p0 < p1;
      unsigned long long mb = __ballot(more);
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
        if (SPHERES) j = p0;  // the staging row holds record positions until the second phase below
        ++p0;
      }
      unsigned long long hb = __ballot(hit);
      uint32_t gm = (uint32_t)(hb >> gbase) & 255u;
      if (hit) {
        uint32_t slot = np + __popc(gm & ((1u << sub) - 1u));
        if (slot < (uint32_t)kRowCap) {
          if (SPHERES) acc[slot] = (AccT)j;
          else row[slot] = j;
        }
      }
      np += __popc(gm);
    }
  }
  n_accepted = np;
  if (SPHERES) {
    // second phase: the accepted partners, sixteen at a time - two per lane, so a typical query needs one round trip for its
    // partners' records - through the sphere-sphere test; contacts go to the row
    if (np > (uint32_t)kRowCap) atomicOr(overflow, 1u);
    const uint32_t na = min(np, (uint32_t)kRowCap);
    uint32_t nc = 0;
    for (uint32_t a0 = 0; a0 < na; a0 += 2 * kCoopLanes) {
      bool hit[2] = {false, false};
      uint32_t jj[2] = {0, 0};
      float4 c0[2], d0[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const uint32_t a = a0 + (uint32_t)u * kCoopLanes + (uint32_t)sub;
        const uint32_t pj = a < na ? (uint32_t)acc[a] : (uint32_t)acc[0];
        S.col(pj, c0[u], d0[u], jj[u]);
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const uint32_t a = a0 + (uint32_t)u * kCoopLanes + (uint32_t)sub;
        if (a < na) {
          // cheap and conservative first: the centres never come closer than |d| - |v| during the tick
          const V3 dd = xyz(c0[u]) - A.p, v = xyz(d0[u]) - vA;
          const float lim = A.r + c0[u].w + __builtin_sqrtf(dot(v, v));
          if (dot(dd, dd) <= lim * lim * 1.001f) {
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

// k_pair_grid with the cells in LDS.  k_pair_grid is bound by instruction issue, not by memory: 8 lanes share a query and
// spend ~1900 wave instructions per 8 queries on cell arithmetic (divisions, Morton interleaves), ballots and lanes that
// wait for each other.  Here a block takes a BRICK of 4 x 4 x 4 cells (64 consecutive Morton cells: its queries are one
// contiguous range of the sorted bodies) and copies the 8 x 8 x 8 cells around it into LDS once - one thread per cell,
// records re-sorted x-major so that a column of cells along z is one contiguous range - and 8 lanes share a query: lane s
// takes the z-columns s, s + 8, ... of the query's cells (a column is typically 3-4 cells, ~4 records, read four at a time);
// hits are appended through an LDS counter.  A query whose cells reach outside the box (a body much larger than a cell), or a brick whose box holds
// more records than the LDS copy has room for, is answered by one lane from global memory with the same code: the accepted
// set never depends on which way it was found (the order inside a row never mattered: contacts are numbered by partner id).
__device__ __forceinline__ uint32_t compact10(uint32_t v) {  // inverse of expand10
  v &= 0x09249249u;
  v = (v ^ (v >> 2)) & 0x030C30C3u;
  v = (v ^ (v >> 4)) & 0x0300F00Fu;
  v = (v ^ (v >> 8)) & 0xFF0000FFu;
  v = (v ^ (v >> 16)) & 0x000003FFu;
  return v;
}
constexpr int kBrickLanes = 8;                                  // lanes per query
constexpr int kBrickQueries = kCoopBlock / kBrickLanes;         // queries per pass of a block
// Records in the LDS copy are PADDED by one in sixteen (r05): the eight lanes of a query read the heads of eight neighbouring z-columns -
// ~4 records, 64 bytes, apart - so lanes k and k + 4 met in the same banks; with the pad the second four are a record further on.
// k_pair_brick 72.9 -> 65.4 us on the falling pile (one in eight: 64.4 with fewer records staged; one in 32: 67.0).  The physical length
// stays 672 (three blocks per CU: a record more per block and the CU holds two - 83 us), so a brick stages 632 records.
constexpr uint32_t kBrickCap = 632;                             // leaf records staged per brick (64 B each; 32 B when the sphere test is not fused)
__host__ __device__ constexpr uint32_t brick_phys(uint32_t p) { return p + (p >> 4); }
__host__ __device__ constexpr uint32_t brick_phys_cap(uint32_t cap) { return brick_phys(cap) + 1u; }
struct BrickSrcLds {   // the staged box: records in x-major cell order, start[] = first record of every cell (+ end)
  const float4 *rc, *rr, *cc, *cd;
  const uint16_t* start;
  int hb[3];
  static constexpr bool kColumns = true;
  __device__ __forceinline__ void column(uint32_t cx, uint32_t cy, uint32_t cz, uint32_t dz, uint32_t& p0, uint32_t& p1) const {
    const uint32_t c = ((uint32_t)((int)cx - hb[0]) << 6) | ((uint32_t)((int)cy - hb[1]) << 3) | (uint32_t)((int)cz - hb[2]);
    p0 = start[c]; p1 = start[c + dz];
  }
  __device__ __forceinline__ void leaf(uint32_t p, float4& c, float4& r) const { c = rc[brick_phys(p)]; r = rr[brick_phys(p)]; }
  __device__ __forceinline__ void col(uint32_t p, float4& c0, float4& d0, uint32_t& j) const { c0 = cc[brick_phys(p)]; d0 = cd[brick_phys(p)]; j = f2u(rc[brick_phys(p)].w); }
};
struct BrickSrcGlobal {  // the sorted arrays themselves, one cell at a time
  Lbvh T;
  uint32_t nb[3];
  int shift;
  static constexpr bool kColumns = false;
  __device__ __forceinline__ void column(uint32_t cx, uint32_t cy, uint32_t cz, uint32_t, uint32_t& p0, uint32_t& p1) const {
    const uint32_t code = (expand10(cx << (10u - nb[0])) << 2) | (expand10(cy << (10u - nb[1])) << 1) | expand10(cz << (10u - nb[2]));
    const uint32_t cell = code >> shift;
    p0 = T.cell_lo[cell]; p1 = T.cell_lo[cell + 1];
  }
  __device__ __forceinline__ void leaf(uint32_t p, float4& c, float4& r) const { LeafRec lr = T.leaves[p]; c = lr.c; r = lr.r; }
  __device__ __forceinline__ void col(uint32_t p, float4& c0, float4& d0, uint32_t& j) const { c0 = T.lcol[2 * p]; d0 = T.lcol[2 * p + 1]; j = T.sidx[p]; }
};
// Lane s of LQ over the cells [ca, ca + d) of one query.  cnt[0..2] (LDS, zero on entry): fat-box partners, partners that pass
// the conservative distance test, contacts.  acc: staging row of record positions (kRowCap entries).
template <bool SPHERES, class Src, class AccT>
__device__ __forceinline__ void brick_query(const Src& S, const Box& q, const Comp& A, V3 vA, uint32_t oi /* the query's order id */, uint32_t n_owned, const uint32_t* ca,
                                            const uint32_t* d, uint32_t s, uint32_t LQ, uint32_t* row, AccT* acc, uint32_t* cnt) {
  // lane s takes the z-columns s, s + LQ, ... of the d[0] x d[1] columns (division by the small d[1] through a multiplier)
  const uint32_t ncol = d[0] * d[1];
  const uint32_t magic = d[1] <= 8u ? ((65536u + d[1] - 1u) / d[1]) : 0u;  // (constant divisors: the compiler folds the eight cases)
  for (uint32_t m = s; m < ncol; m += LQ) {
    const uint32_t cx = magic ? (m * magic) >> 16 : m / d[1], cy = m - cx * d[1];
    {
      for (uint32_t cz = 0; cz < (Src::kColumns ? 1u : d[2]); ++cz) {
        uint32_t p0, p1;
        S.column(ca[0] + cx, ca[1] + cy, ca[2] + cz, d[2], p0, p1);
        for (uint32_t p = p0; p < p1; p += 4) {  // four records in flight
          float4 lc[4], lr[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) S.leaf(min(p + (uint32_t)u, p1 - 1u), lc[u], lr[u]);
          uint32_t hits = 0;
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const uint32_t oj = f2u(lr[u].w);
            Box fb; fb.c = xyz(lc[u]); fb.r = xyz(lr[u]);
            // world.rs:266 (ghost-ghost skipped) and the reference's own acceptance test (bvh.rs:297).  (The ghost test on the order
            // id: a world with ghosts is in the caller's order, where it is the slot; a re-sorted store has none.  The record's slot
            // word is then only read for a hit: four registers less across the box test.)
            if (p + (uint32_t)u < p1 && oj < oi && oj < n_owned && box_overlaps(q, fb)) hits |= 1u << u;
          }
          if (hits) {
            uint32_t slot = atomicAdd(&cnt[0], (uint32_t)__popc(hits));
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              if ((hits >> u) & 1u) {
                if (slot < (uint32_t)kRowCap) {
                  if (SPHERES) acc[slot] = (AccT)(p + (uint32_t)u);
                  else row[slot] = f2u(lc[u].w);
                }
                ++slot;
              }
            }
          }
        }
      }
    }
  }
  if (!SPHERES) return;
  // the accepted partners through the sphere-sphere test: first the cheap conservative reject (the centres never come closer
  // than |d| - |v| during the tick), survivors compacted in place, then the narrowphase's own function on those
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  const uint32_t na = min(*(volatile uint32_t*)&cnt[0], (uint32_t)kRowCap);
  for (uint32_t a = s; a < na; a += LQ) {
    const uint32_t pj = (uint32_t)acc[a];
    float4 c0, d0; uint32_t jj;
    S.col(pj, c0, d0, jj);
    const V3 dd = xyz(c0) - A.p, v = xyz(d0) - vA;
    const float lim = A.r + c0.w + __builtin_sqrtf(dot(v, v));
    if (dot(dd, dd) <= lim * lim * 1.001f) acc[atomicAdd(&cnt[1], 1u)] = (AccT)pj;  // (entries below a + LQ have been read: the lanes of a query move together)
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  const uint32_t ns = *(volatile uint32_t*)&cnt[1];
  for (uint32_t a = s; a < ns; a += LQ) {
    const uint32_t pj = (uint32_t)acc[a];
    float4 c0, d0; uint32_t jj;
    S.col(pj, c0, d0, jj);
    Comp Bc; Bc.kind = KIND_SPHERE; Bc.p = xyz(c0); Bc.r = c0.w; Bc.d = mk3(0, 0, 0);
    LocalContact lc;
    if (comp_pair_local(A, vA, Bc, xyz(d0), &lc)) row[atomicAdd(&cnt[2], 1u)] = jj;
  }
}

// (k_pair_grid itself: behind brick_query, whose walk it may use)
template <bool SPHERES>
__global__ __launch_bounds__(kCoopBlock) void k_pair_grid(Bodies B, uint32_t n, uint32_t n_owned, Lbvh T, const SceneBounds* sb, float pad_abs,
                                                          uint32_t* rows_p, uint32_t* p_cnt, uint32_t* overflow, uint32_t* too_wide,
                                                          uint32_t* pair_stat, float min_frac) {
  __shared__ uint32_t s_acc[SPHERES ? kCoopBlock / kCoopLanes : 1][SPHERES ? kRowCap : 1];  // accepted partners of a query (leaf positions)
#if defined(MGF_PG_WALK) && MGF_PG_WALK == 1
  __shared__ uint32_t s_cnt3[kCoopBlock / kCoopLanes][4];
#endif
  const int lane = threadIdx.x & 63;
  const int sub = lane & 7;
  const int gbase = lane & ~7;
  uint32_t kq = xcd_logical_block_coop() * (kCoopBlock / kCoopLanes) + (threadIdx.x >> 3);
  const bool live = kq < n;  // whole groups are live or not
  uint32_t i = live ? T.sidx[kq] : 0u;
  uint32_t np = 0, n_accepted = 0;
  const uint32_t oi = live ? order_id(T.ext, i) : 0u;
  if (live && oi != 0 && T.n >= 2) {  // world.rs:256
    Box q;
    Comp A; V3 vA = mk3(0, 0, 0);
    uint32_t ca[3], d[3];
    const uint32_t P = 2u * T.levels;
    const uint32_t nb[3] = {(P + 2u) / 3u, (P + 1u) / 3u, P / 3u};  // prefix bits per axis (x is the most significant)
    if (T.ltb) {  // the query in cell order, its cells worked out by k_scatter_leaves: no look-up through the body index, no divisions
      const float4 qc = T.ltb[2 * kq], qr = T.ltb[2 * kq + 1];
      q.c = xyz(qc); q.r = xyz(qr);
      const uint32_t ra = f2u(qc.w), rd = f2u(qr.w);
      ca[0] = ra & 1023u; ca[1] = (ra >> 10) & 1023u; ca[2] = ra >> 20;
      d[0] = rd & 1023u; d[1] = (rd >> 10) & 1023u; d[2] = rd >> 20;
      if (SPHERES) { const float4 c0 = T.lcol[2 * kq]; A.p = xyz(c0); A.r = c0.w; A.d = mk3(0, 0, 0); A.kind = KIND_SPHERE; vA = xyz(T.lcol[2 * kq + 1]); }
    } else {
      q.c = xyz(B.tb_c[i]); q.r = xyz(B.tb_r[i]);
      if (SPHERES) { A = load_comp(B, i); A.kind = KIND_SPHERE; vA = xyz(B.delta[i]); }
      float pad = pad_abs + 1e-5f * (fabs_rs(q.c.x) + fabs_rs(q.c.y) + fabs_rs(q.c.z) + q.r.x + q.r.y + q.r.z);
      pair_query_region(q.c, q.r, pad, sb, nb, ca, d, min_frac);
    }
    if (d[0] * d[1] * d[2] > kGridMaxCells) {
      if (sub == 0) *too_wide = 1u;
    } else {
#if defined(MGF_PG_WALK) && MGF_PG_WALK == 1
      // (r06) the walk of k_pair_brick's slow path: every lane its own z-columns of cells, four leaf records in flight, hits through an LDS counter
      BrickSrcGlobal S; S.T = T; S.nb[0] = nb[0]; S.nb[1] = nb[1]; S.nb[2] = nb[2]; S.shift = kMortonBits - (int)P;
      uint32_t* cnt3 = s_cnt3[threadIdx.x >> 3];
      if (sub < 3) cnt3[sub] = 0u;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      uint32_t* row = rows_p + (size_t)i * kRowCap;
      if (SPHERES) brick_query<SPHERES>(S, q, A, vA, oi, n_owned, ca, d, (uint32_t)sub, (uint32_t)kCoopLanes, row, s_acc[SPHERES ? threadIdx.x >> 3 : 0], cnt3);
      else brick_query<SPHERES>(S, q, A, vA, oi, n_owned, ca, d, (uint32_t)sub, (uint32_t)kCoopLanes, row, row, cnt3);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      n_accepted = *(volatile uint32_t*)&cnt3[0];
      np = SPHERES ? *(volatile uint32_t*)&cnt3[2] : n_accepted;
      if (SPHERES && n_accepted > (uint32_t)kRowCap && sub == 0) atomicOr(overflow, 1u);
#else
      PairSrcGlobal S; S.T = T;
      pair_query_cells<SPHERES>(S, q, A, vA, oi, n_owned, ca, d, nb, kMortonBits - (int)P, sub, gbase, rows_p + (size_t)i * kRowCap,
                                s_acc[SPHERES ? threadIdx.x >> 3 : 0], overflow, np, n_accepted);
#endif
    }
  }
  if (live && sub == 0) {
    p_cnt[i] = np;
    if (!SPHERES && np > (uint32_t)kRowCap) atomicOr(overflow, 1u);
  }
  if (SPHERES) {  // accepted partners: one atomic per block, spread over many words
    __shared__ uint32_t s_sum;
    if (threadIdx.x == 0) s_sum = 0;
    __syncthreads();
    uint32_t v = (live && sub == 0) ? n_accepted : 0u;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    if (lane == 0 && v) atomicAdd(&s_sum, v);
    __syncthreads();
    if (threadIdx.x == 0 && s_sum) atomicAdd(&pair_stat[blockIdx.x & (kPairStatWords - 1u)], s_sum);
  }
}

template <bool SPHERES>
__global__ __launch_bounds__(kCoopBlock) __attribute__((amdgpu_waves_per_eu(6, 8))) void k_pair_brick(uint32_t n, uint32_t n_owned, Lbvh T,
                                                           uint32_t* rows_p, uint32_t* p_cnt, uint32_t* overflow, uint32_t* too_wide,
                                                           uint32_t* pair_stat, uint32_t* slow_queries, uint32_t cap) {
  extern __shared__ float4 s_dyn[];                      // records: rc | rr | (cc | cd), cap each
  __shared__ uint16_t s_start[516];
  __shared__ uint32_t s_wsum[kCoopBlock / 64];
  __shared__ uint16_t s_acc[SPHERES ? kBrickQueries : 1][SPHERES ? kRowCap : 1];
  __shared__ uint32_t s_cnt[kBrickQueries][3];
  __shared__ uint32_t s_qo[kBrickQueries];  // the order id of each group's query (fetched ahead with the query, parked here: a register less)
  __shared__ uint32_t s_sum, s_slow, s_q[2];
  const uint32_t P = 2u * T.levels;
  const uint32_t nbricks = (1u << P) >> 6;
  const uint32_t brick = xcd_logical_block_coop();
  if (brick >= nbricks) return;
  const uint32_t nb[3] = {(P + 2u) / 3u, (P + 1u) / 3u, P / 3u};  // prefix bits per axis (x is the most significant)
  const int shift = kMortonBits - (int)P;
  BrickSrcLds L;
  const uint32_t capp = brick_phys_cap(cap);  // (the arrays' physical length)
  L.rc = s_dyn; L.rr = s_dyn + capp; L.cc = s_dyn + 2 * capp; L.cd = s_dyn + 3 * capp; L.start = s_start;
  {
    const uint32_t code = (brick * 64u) << shift;
    L.hb[0] = (int)(compact10(code >> 2) >> (10u - nb[0])) - 2;
    L.hb[1] = (int)(compact10(code >> 1) >> (10u - nb[1])) - 2;
    L.hb[2] = (int)(compact10(code) >> (10u - nb[2])) - 2;
  }
  // round trip 1: this thread's cell of the box (x-major) and its range of leaf records
  const uint32_t t = threadIdx.x;
  const int lane = t & 63;
  uint32_t g0 = 0, cnt = 0;
  {
    const int c[3] = {L.hb[0] + (int)(t >> 6), L.hb[1] + (int)((t >> 3) & 7u), L.hb[2] + (int)(t & 7u)};
    if (c[0] >= 0 && c[1] >= 0 && c[2] >= 0 && c[0] < (1 << nb[0]) && c[1] < (1 << nb[1]) && c[2] < (1 << nb[2])) {
      const uint32_t cd = (expand10((uint32_t)c[0] << (10u - nb[0])) << 2) | (expand10((uint32_t)c[1] << (10u - nb[1])) << 1) |
                          expand10((uint32_t)c[2] << (10u - nb[2]));
      g0 = T.cell_lo[cd >> shift];
      cnt = T.cell_lo[(size_t)(cd >> shift) + 1] - g0;
    }
  }
  if (t == 0) { s_sum = 0; s_slow = 0; }
  uint32_t inc = cnt;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { uint32_t u = __shfl_up(inc, o); if (lane >= o) inc += u; }
  if (lane == 63) s_wsum[t >> 6] = inc;
  if (t == ((2u << 6) | (2u << 3) | 2u)) s_q[0] = g0;         // the brick's own cells are (2..5)^3 of the box: the Morton-first ...
  if (t == ((5u << 6) | (5u << 3) | 5u)) s_q[1] = g0 + cnt;   // ... and the Morton-last one bound its queries
  __syncthreads();
  const uint32_t q0 = s_q[0], q1 = s_q[1];
  if (q0 == q1) return;  // nobody lives here
  uint32_t base = 0, total = 0;
#pragma unroll
  for (int w = 0; w < kCoopBlock / 64; ++w) { const uint32_t u = s_wsum[w]; if (w < (int)(t >> 6)) base += u; total += u; }
  const uint32_t start = base + inc - cnt;
  const bool staged = total <= cap;
  s_start[t] = (uint16_t)min(start, 0xFFFFu);
  if (t == 0) { s_start[512] = (uint16_t)min(total, 0xFFFFu); s_start[513] = s_start[512]; }
  // round trip 2: the cell's records into the box copy, and the first pass's queries (cell-ordered copies: no look-up through
  // the body index)
  uint32_t kq = q0 + t / (uint32_t)kBrickLanes;
  float4 qc, qr, qa, qv;
  qc = qr = qa = qv = make_float4(0, 0, 0, 0);
  uint32_t qi = 0;  // the query's slot (its order id - its own leaf record carries it - waits in s_qo)
  uint2 qreg = make_uint2(0, 0);
  if (kq < q1) { qc = T.ltb[2 * kq]; qr = T.ltb[2 * kq + 1]; if (SPHERES) { qa = T.lcol[2 * kq]; qv = T.lcol[2 * kq + 1]; } qi = T.sidx[kq]; qreg = make_uint2(f2u(qc.w), f2u(qr.w)); }
  if ((t & (uint32_t)(kBrickLanes - 1)) == 0u) s_qo[t / (uint32_t)kBrickLanes] = kq < q1 ? f2u(T.leaves[kq].r.w) : 0u;
  if (staged) {
    float4* rc = s_dyn; float4* rr = s_dyn + capp; float4* cc = s_dyn + 2 * capp; float4* cd = s_dyn + 3 * capp;
    for (uint32_t r = 0; r < cnt; ++r) {
      LeafRec lr = T.leaves[g0 + r];
      const uint32_t pp = brick_phys(start + r);
      rc[pp] = lr.c; rr[pp] = lr.r;
      if (SPHERES) { cc[pp] = T.lcol[2 * (g0 + r)]; cd[pp] = T.lcol[2 * (g0 + r) + 1]; }
    }
  }
  __syncthreads();
  const uint32_t sub = t & (uint32_t)(kBrickLanes - 1), qg = t / (uint32_t)kBrickLanes;
  uint32_t acc_total = 0, slow = 0;
  for (;;) {
    const bool live = kq < q1;  // whole groups are live or not
    const uint32_t i = qi, oi = s_qo[qg];
    uint32_t np = 0;
    if (sub == 0) { s_cnt[qg][0] = 0; s_cnt[qg][1] = 0; s_cnt[qg][2] = 0; }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (live && oi != 0 && T.n >= 2) {  // world.rs:256
      Box q; q.c = xyz(qc); q.r = xyz(qr);
      Comp A; A.p = xyz(qa); A.r = qa.w; A.d = mk3(0, 0, 0); A.kind = KIND_SPHERE;
      const V3 vA = xyz(qv);
      const uint32_t ca[3] = {qreg.x & 1023u, (qreg.x >> 10) & 1023u, qreg.x >> 20}, d[3] = {qreg.y & 1023u, (qreg.y >> 10) & 1023u, qreg.y >> 20};
      uint32_t* row = rows_p + (size_t)i * kRowCap;
      if (d[0] * d[1] * d[2] > kGridMaxCells) {
        if (sub == 0) *too_wide = 1u;
      } else if (staged && (int)ca[0] >= L.hb[0] && (int)ca[1] >= L.hb[1] && (int)ca[2] >= L.hb[2] && (int)(ca[0] + d[0]) <= L.hb[0] + 8 &&
                 (int)(ca[1] + d[1]) <= L.hb[1] + 8 && (int)(ca[2] + d[2]) <= L.hb[2] + 8) {
        brick_query<SPHERES>(L, q, A, vA, oi, n_owned, ca, d, sub, (uint32_t)kBrickLanes, row, s_acc[SPHERES ? qg : 0], s_cnt[qg]);
      } else if (sub == 0) {
        BrickSrcGlobal S; S.T = T; S.nb[0] = nb[0]; S.nb[1] = nb[1]; S.nb[2] = nb[2]; S.shift = shift;
        brick_query<SPHERES>(S, q, A, vA, oi, n_owned, ca, d, 0u, 1u, row, row, s_cnt[qg]);  // (the row itself stages the accepted partners)
        ++slow;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      const uint32_t accepted = *(volatile uint32_t*)&s_cnt[qg][0];
      np = SPHERES ? *(volatile uint32_t*)&s_cnt[qg][2] : accepted;
      if (sub == 0) {
        acc_total += accepted;
        if (accepted > (uint32_t)kRowCap) atomicOr(overflow, 1u);
      }
    }
    if (live && sub == 0) p_cnt[i] = np;
    kq += kBrickQueries;
    if (kq - qg >= q1) break;  // (the block's decision: every group sees the same pass base)
    qi = 0;
    if (kq < q1) { qc = T.ltb[2 * kq]; qr = T.ltb[2 * kq + 1]; if (SPHERES) { qa = T.lcol[2 * kq]; qv = T.lcol[2 * kq + 1]; } qi = T.sidx[kq]; qreg = make_uint2(f2u(qc.w), f2u(qr.w)); }
    if (sub == 0) s_qo[qg] = kq < q1 ? f2u(T.leaves[kq].r.w) : 0u;  // (this pass's value was read at its top: the lanes of a group move together)
  }
  {  // accepted partners: one atomic per block, spread over many words
    uint32_t v = acc_total, u = slow;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { v += __shfl_xor(v, o); u += __shfl_xor(u, o); }
    if (lane == 0 && v) atomicAdd(&s_sum, v);
    if (lane == 0 && u) atomicAdd(&s_slow, u);
    __syncthreads();
    if (SPHERES && t == 0 && s_sum) atomicAdd(&pair_stat[blockIdx.x & (kPairStatWords - 1u)], s_sum);
    if (t == 0 && s_slow) atomicAdd(slow_queries, s_slow);
  }
}
constexpr size_t brick_lds_bytes(bool spheres, uint32_t cap) { return (size_t)brick_phys_cap(cap) * 16u * (spheres ? 4u : 2u); }

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
  uint3 bits;                  // cells per axis = 2^bits, row-major (k_morton_count's axis_bits)
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
  const uint32_t nb[3] = {G.bits.x, G.bits.y, G.bits.z};
  uint32_t ca[3], d[3];
  bool away = false;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    float lo = ord_f(G.sb->lo[k]), hi = ord_f(G.sb->hi[k]), rm = ord_f(G.sb->rmax[k]);
    float a = at(q.c, k) - at(q.r, k) - rm - pad, b = at(q.c, k) + at(q.r, k) + rm + pad;
    uint32_t c0 = morton_quant(a, lo, hi) >> (10u - nb[k]), c1 = morton_quant(b, lo, hi) >> (10u - nb[k]);
    ca[k] = c0; d[k] = c1 - c0 + 1u;
    // a query that ends before the first face box or starts behind the last one on any axis meets no face (the quantisation
    // clamps, so without this test a body far above a flat mesh would still read the cells under it): Mesh::contacts returns
    // at the root of its tree in that case (bvh.rs:283-297)
    away = away || b < lo || a > hi;
  }
  const uint32_t ncell = d[0] * d[1] * d[2];
  uint32_t nt = 0;
  if (away) {
  } else if (ncell > kGridMaxCells) {
    if (sub == 0) *too_wide = 1u;
  } else {
    uint32_t* row = rows_t + (size_t)i * cap_row;
    for (uint32_t cb = 0; cb < ncell; cb += kCoopLanes) {
      uint32_t idx = cb + (uint32_t)sub;
      uint32_t p0 = 0, p1 = 0;
      if (idx < ncell) {
        uint32_t cz = idx % d[2], t = idx / d[2];
        uint32_t cy = t % d[1], cx = t / d[1];
        uint32_t cell = face_cell(ca[0] + cx, ca[1] + cy, ca[2] + cz, G.bits);
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
                // the same argument one level up: a CLEAR overlap with an ancestor implies an overlap with everything above it,
                // and ancestors grow fast (a resting body's thin overlap with a face box is a deep one two or three unions up;
                // without this every such hit climbed all ~17 levels: k_terrain_grid 80 -> see DESIGN.md section 4)
                float ga = fmin_rs(fmin_rs(q.r.x + nbx.r.x - fabs_rs(q.c.x - nbx.c.x), q.r.y + nbx.r.y - fabs_rs(q.c.y - nbx.c.y)),
                                   q.r.z + nbx.r.z - fabs_rs(q.c.z - nbx.c.z));
                float ta = 1e-4f * (mag + fabs_rs(nbx.c.x) + fabs_rs(nbx.c.y) + fabs_rs(nbx.c.z) + nbx.r.x + nbx.r.y + nbx.r.z);
                if (ga > ta) break;
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
  if (sc->fail) return;
  const bool live = i < n;
  uint32_t tb = 0, nt = 0, pb = 0, np = 0;
  if (live) { tb = t_off[i]; nt = t_off[i + 1] - tb; pb = p_off[i]; np = p_off[i + 1] - pb; }
  const bool ok = live && nt <= cap_row_t && np <= (uint32_t)kRowCap;  // an overflowed body: the host re-runs with wider rows or the two-pass path
  if (face_of_rank) {
    // Terrain rows hold DFS ranks in discovery order: every entry goes straight to its place - the number of smaller ranks in the
    // row (ranks are distinct) - and is named.  The wave takes its 64 bodies one after the other with a lane per ENTRY: the row is
    // read, and the list written, by consecutive lanes (a thread walking its own row touched a 64-byte sector per entry and array:
    // 105 us on 131 072 capsules over a heightfield).
    // (round 3: sixteen lanes per body, four bodies at a time - a wave whose 64 bodies all lie on the floor, which is what a store
    // in cell order gives, used to walk them one after the other with a third of its lanes: 38 -> 73 us on config 3)
    const int lane = threadIdx.x & 63, sl = lane & 15;
    for (int sft = 0; sft < 16; ++sft) {
      const int src = (lane >> 4) * 16 + sft;                 // this group's body of the step: lane src of the wave holds it
      const uint32_t bi = __shfl(i, src), bnt = __shfl(ok ? nt : 0u, src), btb = __shfl(tb, src);
      // (groups whose body has no row idle through the shuffles below with bnt = 0: the loops do not run)
      const uint32_t* brow = rows_t + (size_t)bi * cap_row_t;
      for (uint32_t e0 = 0; e0 < bnt; e0 += 16u) {
        const uint32_t e = e0 + (uint32_t)sl;
        const uint32_t x = e < bnt ? brow[e] : 0xFFFFFFFFu;
        uint32_t before = 0;
        for (uint32_t c0 = 0; c0 < bnt; c0 += 16u) {  // (rows longer than a group: the other chunks are read again)
          const uint32_t y = c0 == e0 ? x : (c0 + (uint32_t)sl < bnt ? brow[c0 + (uint32_t)sl] : 0xFFFFFFFFu);
          const uint32_t m = min(bnt - c0, 16u);
          // lane k of the group's row of sixteen, to all of them: a DPP row_share (no LDS round trip, unlike a shuffle by a lane-dependent index)
#define MGF_ROW_SHARE(K) { const uint32_t yk = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)y, 0x150 + (K), 0xF, 0xF, false); before += ((K) < m && yk < x) ? 1u : 0u; }
          MGF_ROW_SHARE(0) MGF_ROW_SHARE(1) MGF_ROW_SHARE(2) MGF_ROW_SHARE(3) MGF_ROW_SHARE(4) MGF_ROW_SHARE(5) MGF_ROW_SHARE(6) MGF_ROW_SHARE(7)
          MGF_ROW_SHARE(8) MGF_ROW_SHARE(9) MGF_ROW_SHARE(10) MGF_ROW_SHARE(11) MGF_ROW_SHARE(12) MGF_ROW_SHARE(13) MGF_ROW_SHARE(14) MGF_ROW_SHARE(15)
#undef MGF_ROW_SHARE
        }
        if (e < bnt) { t_cand[btb + before] = face_of_rank[x]; t_owner[btb + before] = bi; }
      }
    }
  } else if (ok) {
    const uint32_t* rt = rows_t + (size_t)i * cap_row_t;
    for (uint32_t a = 0; a < nt; ++a) { t_cand[tb + a] = rt[a]; t_owner[tb + a] = i; }
  }
  if (!ok) return;
  const uint4* rp = reinterpret_cast<const uint4*>(rows_p + (size_t)i * kRowCap);  // rows are 16-byte aligned (kRowCap % 4 == 0)
  // partners stay in discovery order: only the few that turn into contacts need the canonical (ascending) order, and
  // k_count_contacts numbers those by partner id
  for (uint32_t a = 0; a < np; a += 4) {
    uint4 v = rp[a >> 2];
    uint32_t e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) if (a + k < np) { p_cand[pb + a + k] = e[k]; p_owner[pb + a + k] = i; }
  }
}

}  // namespace mgf
