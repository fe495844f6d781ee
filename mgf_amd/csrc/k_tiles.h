// Spatial tiling: boundary selection, ghost and migrant records.  (Part of the kernel set described in kernels.h.)
#pragma once
#include "k_solver_flow6.h"

namespace mgf {

// ------------------------------------------------------------------------------------------
// Spatial tiling (one process per GPU): boundary selection, ghost export / import.
// Ghost record, 72 floats: x3 q4 v3 w3 delta3 | tag p3 d3 r | inv_mass I9 restitution friction | n_parts, 3 pad |
// kTileParts x (p3 r d3 kind) world parts of a body of several components (zeros for an ordinary body).
// ------------------------------------------------------------------------------------------
constexpr int kGhostFloats = 40 + 8 * kTileParts;
// (r06) between the tiles of mgf_tiles_step a world WITHOUT bodies of several components sends the first kGhostFloatsPlain floats only - the part
// slots of its records would be zeros, 32 of 72 floats; the receiver learns the width from the kinds the sender announces with its counts.  The
// C-ABI's own export / import calls (mgf_world_export_ghosts, ..) keep the one width their callers size their buffers for.
constexpr int kGhostFloatsPlain = 40;

// ---- what a tile sends: selection ------------------------------------------------------------------------------------------------
// Every launch of the tile protocol's own kernels takes up to kTileBatch tiles (blockIdx.y = the tile): a rank that holds several tiles
// - the whole scene on one GPU, two tiles per GPU - pays one launch where it paid one per tile (r05: these kernels move a few KB each and
// cost ~4.5 us apiece: 158 of them per tick of 8 tiles, now 14).
constexpr int kTileBatch = 8;
// The four ascending lists of a tile: owned bodies whose fat box reaches below x_left (ghosts of the left neighbour) / above x_right,
// and bodies whose centre left the slab [x_lo, x_hi) (migrants: left-goers first) - per-block counts, their offsets, the scatter.
struct SelTile {
  const float4 *fb_c, *fb_r, *x;
  uint32_t n;
  float x_left, x_right, x_lo, x_hi;
  uint32_t* blk;   // [4 * blocks]: per block (nl, nr, ml, mr) - counts (k_tile_select_count), then exclusive offsets (k_tile_select_offsets)
  uint32_t* tot;   // [4] the totals
  uint32_t* tot_host;  // null, or where the host reads them (pinned memory as the device sees it: no copy command behind the kernels)
  uint32_t *ids_l, *ids_r, *ids_m;  // (null: not wanted - the count kernels run without them)
};
struct SelBatch { SelTile t[kTileBatch]; };
__device__ __forceinline__ uint32_t sel_flags(const SelTile& S, uint32_t i) {
  const float c = S.fb_c[i].x, h = S.fb_r[i].x, cx = S.x[i].x;
  uint32_t f = 0;
  if (c - h < S.x_left) f |= 1u;
  if (c + h > S.x_right) f |= 2u;
  if (cx < S.x_lo) f |= 4u; else if (cx >= S.x_hi) f |= 8u;
  return f;
}
__global__ __launch_bounds__(kBlock) void k_tile_select_count(SelBatch A) {
  const SelTile& S = A.t[blockIdx.y];
  if (blockIdx.x * kBlock >= S.n) return;
  __shared__ uint32_t s_c[4];
  if (threadIdx.x < 4) s_c[threadIdx.x] = 0u;
  __syncthreads();
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  const uint32_t f = i < S.n ? sel_flags(S, i) : 0u;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint32_t c = (uint32_t)__popcll(__ballot((f >> k) & 1u));
    if ((threadIdx.x & 63u) == 0u && c) atomicAdd(&s_c[k], c);
  }
  __syncthreads();
  if (threadIdx.x < 4) S.blk[4 * (size_t)blockIdx.x + threadIdx.x] = s_c[threadIdx.x];
}
// one workgroup per tile: the per-block counts become exclusive offsets, the totals go to `tot`
constexpr uint32_t kSelScanThreads = 1024;
__global__ __launch_bounds__(kSelScanThreads) void k_tile_select_offsets(SelBatch A) {
  const SelTile& S = A.t[blockIdx.x];
  __shared__ uint32_t s_w[kSelScanThreads / 64][4];
  __shared__ uint32_t s_run[4];
  const uint32_t t = threadIdx.x, lane = t & 63u, wv = t >> 6;
  const uint32_t nb = (S.n + kBlock - 1) / kBlock;
  if (t < 4) s_run[t] = 0u;
  __syncthreads();
  uint4* blk = reinterpret_cast<uint4*>(S.blk);
  for (uint32_t b0 = 0; b0 < nb; b0 += kSelScanThreads) {
    const uint32_t b = b0 + t;
    const uint4 c = b < nb ? blk[b] : make_uint4(0, 0, 0, 0);
    uint32_t v[4] = {c.x, c.y, c.z, c.w}, inc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      inc[k] = v[k];
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { const uint32_t u = __shfl_up(inc[k], o); if ((int)lane >= o) inc[k] += u; }
      if (lane == 63u) s_w[wv][k] = inc[k];
    }
    __syncthreads();
    uint32_t before[4], total[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      before[k] = s_run[k]; total[k] = 0u;
      for (uint32_t w = 0; w < kSelScanThreads / 64; ++w) { const uint32_t u = s_w[w][k]; if (w < wv) before[k] += u; total[k] += u; }
    }
    if (b < nb) blk[b] = make_uint4(before[0] + inc[0] - v[0], before[1] + inc[1] - v[1], before[2] + inc[2] - v[2], before[3] + inc[3] - v[3]);
    __syncthreads();
    if (t < 4) s_run[t] += total[t];
    __syncthreads();
  }
  if (t < 4) { S.tot[t] = s_run[t]; if (S.tot_host) S.tot_host[t] = s_run[t]; }
}
__global__ __launch_bounds__(kBlock) void k_tile_select_scatter(SelBatch A) {
  const SelTile& S = A.t[blockIdx.y];
  if (blockIdx.x * kBlock >= S.n) return;
  __shared__ uint32_t s_w[kBlock / 64][4];
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x, lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
  const uint32_t f = i < S.n ? sel_flags(S, i) : 0u;
  uint32_t rank[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const unsigned long long m = __ballot((f >> k) & 1u);
    rank[k] = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0u) s_w[wv][k] = (uint32_t)__popcll(m);
  }
  __syncthreads();
  if (!f) return;
  const uint4 off = reinterpret_cast<const uint4*>(S.blk)[blockIdx.x];
  const uint32_t o4[4] = {off.x, off.y, off.z, off.w};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (!((f >> k) & 1u)) continue;
    uint32_t r = o4[k] + rank[k];
    for (uint32_t w = 0; w < wv; ++w) r += s_w[w][k];
    if (k == 0) { if (S.ids_l) S.ids_l[r] = i; }
    else if (k == 1) { if (S.ids_r) S.ids_r[r] = i; }
    else if (S.ids_m) S.ids_m[(k == 3 ? S.tot[2] : 0u) + r] = i;
  }
}
// (written word by word: 16-byte stores - 288 bytes apart between lanes - were TWICE as slow as the 72 scalar ones, 95 against 49 us for
// the 64 000 records of eight tiles; the import's 16-byte loads are the faster ones, 31 against 75 us)
__device__ __forceinline__ void export_body(const Bodies& B, uint32_t i, float* o, uint32_t rec = (uint32_t)kGhostFloats) {
  constexpr bool vec = false;
  float4 x = B.x[i], q = B.q[i], s0 = B.srec[4 * i], s1 = B.srec[4 * i + 1], s2 = B.srec[4 * i + 2], s3 = B.srec[4 * i + 3];
  float4 d = B.delta[i], e = B.einfo[i], c0 = B.col0[i], c1 = B.col1[i];
  const uint32_t pc = B.pcount ? B.pcount[i] : 0u;
  float4 r[10];
  r[0] = make_float4(x.x, x.y, x.z, q.x);
  r[1] = make_float4(q.y, q.z, q.w, s0.x);
  r[2] = make_float4(s0.y, s0.z, s0.w, s1.x);
  r[3] = make_float4(s1.y, d.x, d.y, d.z);
  r[4] = make_float4(c1.w, c0.x, c0.y, c0.z);
  r[5] = make_float4(c1.x, c1.y, c1.z, c0.w);
  r[6] = make_float4(s1.z, s1.w, s2.x, s2.y);
  r[7] = make_float4(s2.z, s2.w, s3.x, s3.y);
  r[8] = make_float4(s3.z, s3.w, e.w, d.w);
  r[9] = make_float4(u2f(pc), 0.0f, 0.0f, 0.0f);
  auto put = [&](uint32_t k, const float4& v) {
    if (vec) reinterpret_cast<float4*>(o)[k] = v;
    else { o[4 * k] = v.x; o[4 * k + 1] = v.y; o[4 * k + 2] = v.z; o[4 * k + 3] = v.w; }
  };
#pragma unroll
  for (uint32_t k = 0; k < 10; ++k) put(k, r[k]);
  if (rec <= (uint32_t)kGhostFloatsPlain) return;  // (a world of single-component bodies: no part slots in its records)
  for (uint32_t k = 0; k < (uint32_t)kTileParts; ++k) {
    float4 a = make_float4(0, 0, 0, 0), b = a;
    if (k < pc) { a = B.wp0[kMaxParts * i + k]; b = B.wp1[kMaxParts * i + k]; }
    put(10 + 2 * k, a); put(11 + 2 * k, b);
  }
}
// both faces' records of every tile of the batch: the left face's first (ids_r == null: one list)
struct ExpTile { Bodies B; const uint32_t *ids_l, *ids_r; uint32_t nl, nr; float* out; uint32_t rec, pad; };  // rec: floats per record (kGhostFloats or kGhostFloatsPlain)
struct ExpBatch { ExpTile t[kTileBatch]; };
__global__ __launch_bounds__(kBlock) void k_export_bodies(ExpBatch A) {
  const ExpTile& E = A.t[blockIdx.y];
  const uint32_t t = blockIdx.x * kBlock + threadIdx.x;
  if (t >= E.nl + E.nr) return;
  export_body(E.B, t < E.nl ? E.ids_l[t] : E.ids_r[t - E.nl], E.out + (size_t)t * E.rec, E.rec);
}
__device__ __forceinline__ void import_ghost(Bodies B, uint32_t i, const float* src, float fat_margin, int blo[3], int bhi[3], int brm[3], uint32_t rec = (uint32_t)kGhostFloats) {
  // the record's first 40 floats, as ten 16-byte words when the buffer is 16-byte aligned (the tile set's own buffers are)
  const bool vec = (reinterpret_cast<uintptr_t>(src) & 15u) == 0u;
  auto get = [&](uint32_t k) -> float4 {
    if (vec) return reinterpret_cast<const float4*>(src)[k];
    return make_float4(src[4 * k], src[4 * k + 1], src[4 * k + 2], src[4 * k + 3]);
  };
  float o[40];
#pragma unroll
  for (uint32_t k = 0; k < 10; ++k) { const float4 v = get(k); o[4 * k] = v.x; o[4 * k + 1] = v.y; o[4 * k + 2] = v.z; o[4 * k + 3] = v.w; }
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
  if (B.bpk) { B.bpk[4 * i] = mk4(k.p, k.r); B.bpk[4 * i + 1] = mk4(d, o[35]); B.bpk[4 * i + 2] = mk4(x + d, o[34]); B.bpk[4 * i + 3] = mk4(k.d, o[16]); }
  Box tb = swept_bounds(k, d);
  const bool parts_in = rec > (uint32_t)kGhostFloatsPlain;  // (the record carries part slots)
  const uint32_t pc = parts_in ? min(f2u(o[36]), (uint32_t)kTileParts) : 0u;
  if (B.pcount) {
    B.pcount[i] = pc;
    for (uint32_t pk = 0; pk < (uint32_t)kMaxParts; ++pk) {  // a ghost is never integrated here: its world parts are all that matters
      float4 a = make_float4(0, 0, 0, 0), b = a;
      if (parts_in && pk < (uint32_t)kTileParts) { a = get(10 + 2 * pk); b = get(11 + 2 * pk); }
      B.wp0[kMaxParts * i + pk] = a; B.wp1[kMaxParts * i + pk] = b;
      B.lp0[kMaxParts * i + pk] = a; B.lp1[kMaxParts * i + pk] = b;
      if (pk < pc) {
        Comp part; part.kind = (int)f2u(b.w); part.p = xyz(a); part.r = a.w; part.d = xyz(b);
        Box pb = swept_bounds(part, d);
        tb = pk == 0 ? pb : box_combine(tb, pb);
      }
    }
  }
  B.tb_c[i] = mk4(tb.c, 0.0f); B.tb_r[i] = mk4(tb.r, 0.0f);
  const V3 fr = tb.r + mk3(fat_margin, fat_margin, fat_margin);
  B.fb_c[i] = mk4(tb.c, 0.0f); B.fb_r[i] = mk4(fr, 0.0f);
  blo[0] = bhi[0] = f_ord(tb.c.x); blo[1] = bhi[1] = f_ord(tb.c.y); blo[2] = bhi[2] = f_ord(tb.c.z);
  brm[0] = f_ord(fr.x); brm[1] = f_ord(fr.y); brm[2] = f_ord(fr.z);
  B.sp0[i] = make_float4(0, 0, 0, o[34]); B.sp1[i] = make_float4(0, 0, 0, o[35]);
  B.ctor[i] = make_float4(pc ? u2f(2u) : o[16], k.r, 0.0f, 0.0f);
  B.imb[3 * i] = make_float4(0, 0, 0, 0); B.imb[3 * i + 1] = make_float4(0, 0, 0, 0); B.imb[3 * i + 2] = make_float4(0, 0, 0, 0);
}
// (in2 / m1: the rows from m1 on come from a second buffer - the two neighbours' send buffers read in place, no copy in between)
struct ImpTile { Bodies B; uint32_t n_owned, m, m1; float fat_margin; const float *in, *in2; int* sb_part; uint32_t rec1, rec2; };  // rec1 / rec2: floats per record of `in` / `in2`
struct ImpBatch { ImpTile t[kTileBatch]; };
__global__ __launch_bounds__(kBlock) void k_import_ghosts(ImpBatch A) {
  const ImpTile& I = A.t[blockIdx.y];
  if (blockIdx.x * kBlock >= I.m) return;
  const uint32_t t = blockIdx.x * kBlock + threadIdx.x;
  int blo[3] = {0x7FFFFFFF, 0x7FFFFFFF, 0x7FFFFFFF}, bhi[3] = {(int)0x80000000, (int)0x80000000, (int)0x80000000}, brm[3] = {0, 0, 0};
  if (t < I.m) import_ghost(I.B, I.n_owned + t, t < I.m1 ? I.in + (size_t)t * I.rec1 : I.in2 + (size_t)(t - I.m1) * I.rec2, I.fat_margin, blo, bhi, brm, t < I.m1 ? I.rec1 : I.rec2);
  if (I.sb_part) bounds_block_accumulate(blo, bhi, brm, I.sb_part);  // the scene bounds gathered by this tick's k_integrate take the ghosts in
}
// velocity record: 8 floats (v3, w3, 0, 0); both slab faces of a tile in one list of records, the left face's first.  (r06) rec = 6: the
// two zeros stay at home - between the tiles of mgf_tiles_step; the C-ABI's calls keep 8.
constexpr uint32_t kVelFloats = 8, kVelFloatsPacked = 6;
struct VelTile { float4* srec; const uint32_t *ids_l, *ids_r; uint32_t nl, nr; float4* out; uint32_t rec, pad; };   // export: srec -> out
struct VelBatch { VelTile t[kTileBatch]; };
__global__ __launch_bounds__(kBlock) void k_export_vel(VelBatch A) {
  const VelTile& V = A.t[blockIdx.y];
  const uint32_t t = blockIdx.x * kBlock + threadIdx.x;
  if (t >= V.nl + V.nr) return;
  const uint32_t i = t < V.nl ? V.ids_l[t] : V.ids_r[t - V.nl];
  const float4 s0 = V.srec[4 * (size_t)i], s1 = V.srec[4 * (size_t)i + 1];
  if (V.rec == kVelFloatsPacked) {
    float2* o = reinterpret_cast<float2*>(reinterpret_cast<float*>(V.out) + (size_t)kVelFloatsPacked * t);  // (24-byte records: 8-byte aligned)
    o[0] = make_float2(s0.x, s0.y); o[1] = make_float2(s0.z, s0.w); o[2] = make_float2(s1.x, s1.y);
    return;
  }
  V.out[2 * (size_t)t] = s0;
  V.out[2 * (size_t)t + 1] = make_float4(s1.x, s1.y, 0.0f, 0.0f);
}
struct GVelTile { float4* srec; uint32_t n_owned, m, m1; const float4 *in, *in2; uint32_t rec, pad; };  // import: the rows from m1 on come from in2
struct GVelBatch { GVelTile t[kTileBatch]; };
__global__ __launch_bounds__(kBlock) void k_import_ghost_vel(GVelBatch A) {
  const GVelTile& G = A.t[blockIdx.y];
  const uint32_t t = blockIdx.x * kBlock + threadIdx.x;
  if (t >= G.m) return;
  const size_t i = (size_t)G.n_owned + t;
  if (G.rec == kVelFloatsPacked) {
    const float2* r = reinterpret_cast<const float2*>(t < G.m1 ? reinterpret_cast<const float*>(G.in) + (size_t)kVelFloatsPacked * t
                                                               : reinterpret_cast<const float*>(G.in2) + (size_t)kVelFloatsPacked * (t - G.m1));
    const float2 a = r[0], b = r[1], c = r[2];
    G.srec[4 * i] = make_float4(a.x, a.y, b.x, b.y);
    *reinterpret_cast<float2*>(&G.srec[4 * i + 1]) = c;
    return;
  }
  const float4* r = t < G.m1 ? G.in + 2 * (size_t)t : G.in2 + 2 * (size_t)(t - G.m1);
  G.srec[4 * i] = r[0];
  *reinterpret_cast<float2*>(&G.srec[4 * i + 1]) = make_float2(r[1].x, r[1].y);
}

// the solver flags of every tile of the batch (16 words each: abort, fail bits, block sizes) straight into the tiles' pinned read-back blocks:
// one launch where eight device-to-host copies - each a blit kernel with ~18 us of queue handling around it - stood
struct FlagTile { const uint32_t* src; uint32_t* dst; uint32_t words; uint32_t pad; };  // words <= 16
struct FlagBatch { FlagTile t[kTileBatch]; };
__global__ __launch_bounds__(16 * kTileBatch) void k_fetch_flags(FlagBatch A) {
  const FlagTile& F = A.t[threadIdx.x >> 4];
  if ((threadIdx.x & 15u) < F.words) F.dst[threadIdx.x & 15u] = F.src[threadIdx.x & 15u];
}
// What a tile's tick changes of its owned bodies that cannot be worked out again - position, orientation, velocities, motion, the
// persistent fat box (7 words per body) - and back: a tick lost to a solver launch that gave up is repeated from here (mgf_tiles_step).
constexpr int kTickSnapWords = 7;
struct SnapTile { Bodies B; uint32_t n, pad; float4* snap; };
struct SnapBatch { SnapTile t[kTileBatch]; };
__global__ __launch_bounds__(kBlock) void k_tick_snapshot(SnapBatch A, int restore) {
  const Bodies& B = A.t[blockIdx.y].B;
  const uint32_t n = A.t[blockIdx.y].n;
  float4* snap = A.t[blockIdx.y].snap;
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  // word-major (snap[e * n + i]): consecutive lanes, consecutive addresses on both sides (body-major, 112 bytes apart: 16.7 us per 131 072 bodies; so: see EXPERIMENTS.md)
  float4* s = snap + i;
  const size_t n_ = n;
  if (!restore) {
    s[0] = B.x[i]; s[n_] = B.q[i]; s[2 * n_] = B.srec[4 * (size_t)i]; s[3 * n_] = B.srec[4 * (size_t)i + 1]; s[4 * n_] = B.delta[i]; s[5 * n_] = B.fb_c[i]; s[6 * n_] = B.fb_r[i];
  } else {
    B.x[i] = s[0]; B.q[i] = s[n_]; B.srec[4 * (size_t)i] = s[2 * n_]; B.srec[4 * (size_t)i + 1] = s[3 * n_]; B.delta[i] = s[4 * n_]; B.fb_c[i] = s[5 * n_]; B.fb_r[i] = s[6 * n_];
  }
}
// ---- migration of owned bodies between tiles -----------------------------------------------------
// A migrant record is the body's row of every Bodies array, verbatim (kMigrantWords float4 = 116 floats): the
// receiving tile continues bit-identically, persistent fat box and constructor tag (ctor.w) included.
constexpr int kMigrantWords = 21 + 4 * kTileParts;  // the tile protocol's record: 20 words of the ordinary arrays + part count + kTileParts slots of each of the four part arrays
constexpr int kBodyWords = 21 + 4 * kMaxParts;      // a body's whole row (internal moves: re-sorting the store, compaction): every part slot
__device__ __forceinline__ float4* body_word(const Bodies& B, uint32_t e, uint32_t i) {  // e < 20: the ordinary arrays
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
__device__ __forceinline__ float4* part_word(const Bodies& B, uint32_t arr, uint32_t slot, uint32_t i) {
  float4* base = arr == 0 ? B.lp0 : (arr == 1 ? B.lp1 : (arr == 2 ? B.wp0 : B.wp1));
  return base + (size_t)kMaxParts * i + slot;
}
// Word e of a record with `slots` part slots per array (kTileParts: a tile record; kMaxParts: a whole row): 0..19 ordinary, 20 the
// part count, then `slots` words of lp0, lp1, wp0, wp1.  Words 20.. exist only in worlds that hold bodies of several parts
// (elsewhere they read as zeros and writes are dropped); part slots a record does not carry are cleared on the way in.
__device__ __forceinline__ float4 migrant_get(const Bodies& B, uint32_t e, uint32_t i, uint32_t slots) {
  if (e < 20u) return *body_word(B, e, i);
  if (!B.pcount) return make_float4(0, 0, 0, 0);
  if (e == 20u) return make_float4(u2f(B.pcount[i]), 0, 0, 0);
  const uint32_t k = e - 21u;
  return *part_word(B, k / slots, k % slots, i);
}
__device__ __forceinline__ void migrant_put(const Bodies& B, uint32_t e, uint32_t i, float4 v, uint32_t slots) {
  if (e < 20u) { *body_word(B, e, i) = v; return; }
  if (!B.pcount) return;
  if (e == 20u) { B.pcount[i] = f2u(v.x); return; }
  const uint32_t k = e - 21u, arr = k / slots, slot = k % slots;
  *part_word(B, arr, slot, i) = v;
  if (slots < (uint32_t)kMaxParts && slot + 1u == slots)
    for (uint32_t z = slots; z < (uint32_t)kMaxParts; ++z) *part_word(B, arr, z, i) = make_float4(0, 0, 0, 0);
}
// `slots` part slots per array in the records: kTileParts (words = kMigrantWords: the tile protocol) or kMaxParts (words = kBodyWords)
__global__ __launch_bounds__(kBlock) void k_export_migrants(Bodies B, const uint32_t* ids, uint32_t m, float4* out, uint32_t slots) {
  const uint32_t words = 21u + 4u * slots;
  uint32_t t = blockIdx.x * kBlock + threadIdx.x;
  if (t >= m * words) return;
  uint32_t b = t / words, e = t % words;
  out[t] = migrant_get(B, e, ids[b], slots);
}
// (in2 / m1: the records from m1 on come from a second buffer - the two neighbours' send buffers read in place)
__global__ __launch_bounds__(kBlock) void k_import_migrants(Bodies B, uint32_t base, uint32_t m, const float4* in, uint32_t slots, const float4* in2 = nullptr,
                                                            uint32_t m1 = 0xFFFFFFFFu) {
  const uint32_t words = 21u + 4u * slots;
  uint32_t t = blockIdx.x * kBlock + threadIdx.x;
  if (t >= m * words) return;
  uint32_t b = t / words, e = t % words;
  migrant_put(B, e, base + b, b < m1 ? in[t] : in2[t - m1 * words], slots);
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
// pos = exclusive scan of the 0 / 1 flags `keep` (m entries; kNone where the flag is 0, the total in the last entry) in three small launches: per-block counts, their offsets (one workgroup), the
// positions.  (rocPRIM's scan did this in two - behind ~175 us of host time per call, device queries, during which the stream ran dry.)
__global__ __launch_bounds__(kBlock) void k_flag_count(const uint32_t* keep, uint32_t m, uint32_t* blk) {
  __shared__ uint32_t s_c;
  if (threadIdx.x == 0) s_c = 0u;
  __syncthreads();
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  const uint32_t c = (uint32_t)__popcll(__ballot(i < m && keep[i] != 0u));
  if ((threadIdx.x & 63u) == 0u && c) atomicAdd(&s_c, c);
  __syncthreads();
  if (threadIdx.x == 0) blk[blockIdx.x] = s_c;
}
__global__ __launch_bounds__(kSelScanThreads) void k_flag_offsets(uint32_t* blk, uint32_t nb) {
  __shared__ uint32_t s_w[kSelScanThreads / 64];
  __shared__ uint32_t s_run;
  const uint32_t t = threadIdx.x, lane = t & 63u, wv = t >> 6;
  if (t == 0) s_run = 0u;
  __syncthreads();
  for (uint32_t b0 = 0; b0 < nb; b0 += kSelScanThreads) {
    const uint32_t b = b0 + t, v = b < nb ? blk[b] : 0u;
    uint32_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t u = __shfl_up(inc, o); if ((int)lane >= o) inc += u; }
    if (lane == 63u) s_w[wv] = inc;
    __syncthreads();
    uint32_t before = s_run, total = 0u;
    for (uint32_t w = 0; w < kSelScanThreads / 64; ++w) { const uint32_t u = s_w[w]; if (w < wv) before += u; total += u; }
    if (b < nb) blk[b] = before + inc - v;
    __syncthreads();
    if (t == 0) s_run += total;
    __syncthreads();
  }
}
__global__ __launch_bounds__(kBlock) void k_flag_positions(const uint32_t* keep, uint32_t m, const uint32_t* blk, uint32_t* pos) {
  __shared__ uint32_t s_w[kBlock / 64];
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x, lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
  const unsigned long long mk = __ballot(i < m && keep[i] != 0u);
  if (lane == 0u) s_w[wv] = (uint32_t)__popcll(mk);
  __syncthreads();
  if (i >= m) return;
  uint32_t r = blk[blockIdx.x] + (uint32_t)__popcll(mk & ((1ull << lane) - 1ull));
  for (uint32_t w = 0; w < wv; ++w) r += s_w[w];
  pos[i] = (keep[i] != 0u || i + 1u == m) ? r : kNone;  // (the last entry - the list's own zero - carries the total)
}
// The same positions for a SHORT list of removed bodies (a tile's hand-over: a handful per tick) in one launch: pos[i] = i - (ids below i),
// kNone for a listed body; err |= 1 for an id out of range or listed twice.  (The list in any order.)
constexpr uint32_t kRemoveShort = 64;
__global__ __launch_bounds__(kBlock) void k_remove_positions_short(const uint32_t* ids, uint32_t m, uint32_t n, uint32_t* pos, uint32_t* err) {
  __shared__ uint32_t s_ids[kRemoveShort];
  if (threadIdx.x < m) s_ids[threadIdx.x] = ids[threadIdx.x];
  __syncthreads();
  if (blockIdx.x == 0 && threadIdx.x < m) {
    const uint32_t id = s_ids[threadIdx.x];
    bool bad = id >= n;
    for (uint32_t j = 0; j < threadIdx.x; ++j) bad = bad || s_ids[j] == id;
    if (bad) atomicOr(err, 1u);
  }
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i > n) return;  // (pos[n] = the new length, as the scan's total)
  uint32_t below = 0;
  bool listed = false;
  for (uint32_t k = 0; k < m; ++k) { const uint32_t id = s_ids[k]; below += id < i ? 1u : 0u; listed = listed || id == i; }
  pos[i] = listed ? kNone : i - below;
}
// stable compaction through a scratch copy: tmp[pos[i]] = row i for kept bodies, then rows [0, n_new) = tmp.  Word-major like the
// re-sort's copy (blockIdx.y = the word, tmp[e * n + k]): consecutive lanes read one array at consecutive bodies and write consecutive
// words (r05; body-major - 37 arrays per wave load - took 29 + 21 us per removal on a 131 072-body tile).
__global__ __launch_bounds__(kBlock) void k_compact_gather(Bodies B, uint32_t n, const uint32_t* pos /* kNone: removed */, float4* tmp) {
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x, e = blockIdx.y;
  if (i >= n) return;
  const uint32_t p = pos[i];
  if (p != kNone && p != i) tmp[(size_t)e * n + p] = migrant_get(B, e, i, kMaxParts);  // (a body in front of the first removed one stays where it is)
}
// ... rows [0, n_new) = tmp, except the rows that did not move: row k kept its place iff every body up to it was kept (pos[k] == k)
__global__ __launch_bounds__(kBlock) void k_compact_put(Bodies B, uint32_t n, uint32_t n_new, const uint32_t* pos, const float4* tmp) {
  const uint32_t k = blockIdx.x * kBlock + threadIdx.x, e = blockIdx.y;
  if (k >= n_new) return;
  if (pos[k] == k) return;
  migrant_put(B, e, k, tmp[(size_t)e * n + k], kMaxParts);
}
// ---- the body store in an internal order (host_perm.inc) ----------------------------------------------------------
// Slot k of the new order takes the row of old slot order[k] - the body's row of every Bodies array, verbatim, like a
// migrant's.  Word-major through the scratch copy (tmp[e * n + k]): consecutive lanes read one array at nearly consecutive
// bodies (the store is re-sorted every few ticks: the old order is close to the new one) and write consecutive words.
__global__ __launch_bounds__(kBlock) void k_permute_gather(Bodies B, uint32_t n, const uint32_t* order, float4* tmp) {
  const uint32_t k = blockIdx.x * kBlock + threadIdx.x, e = blockIdx.y;
  if (k >= n) return;
  tmp[(size_t)e * n + k] = migrant_get(B, e, order[k], kMaxParts);
}
__global__ __launch_bounds__(kBlock) void k_permute_put(Bodies B, uint32_t n, const float4* tmp) {
  const uint32_t k = blockIdx.x * kBlock + threadIdx.x, e = blockIdx.y;
  if (k >= n) return;
  migrant_put(B, e, k, tmp[(size_t)e * n + k], kMaxParts);
}
// ---- the order of a re-sort: compact blocks (host_perm.inc, partition_order) ------------------------------------------
// Sort keys of one level of the three-level split (x slabs, y rows inside a slab, z inside a row): the unit the body fell into at
// the level above - found from its position in that level's sorted order: units are runs of positions holding whole blocks, dealt
// out evenly (the first B % f units one block more) - in the high bits, the 20-bit coordinate along this level's axis below.
// vals[p] = the body (slot) at position p; null = p itself.
constexpr uint32_t kPartCoordBits = 20;
struct PartPlan { uint32_t nb, B, fx, fy; uint32_t cb[3]; };  // bodies per block, blocks, x slabs, y rows per slab (at most); coarse bits per axis
// A cut by COUNT falls in the middle of a layer of bodies (a lattice, a settled pile): ordered by the exact coordinate the layer's
// bodies go left or right by their jitter - dust on both sides of the cut, every neighbour pair inside the layer a pair across a block
// face (11.2 % of the fresh lattice's pairs against 9 % for aligned boxes).  So the level's coordinate is quantised COARSELY (cells of
// about a body) and the next axis breaks the ties: the layer that holds the cut is itself cut along a line.
__host__ __device__ __forceinline__ void part_deal(uint32_t total, uint32_t f, uint32_t blk, uint32_t* unit, uint32_t* first, uint32_t* count) {
  // `total` blocks over f units, the first total % f one more: which unit holds block `blk`, where that unit starts, how many it has
  const uint32_t q = total / f, r = total % f, big = r * (q + 1u);
  if (blk < big) { *unit = blk / (q + 1u); *first = *unit * (q + 1u); *count = q + 1u; }
  else { const uint32_t k = q ? (blk - big) / q : 0u; *unit = r + k; *first = big + k * q; *count = q; }
}
__global__ __launch_bounds__(kBlock) void k_part_keys(uint32_t n, const uint32_t* vals, const float4* fb_c, const SceneBounds* sb, int axis,
                                                      PartPlan P, uint32_t* keys, uint32_t* vals_out) {
  const uint32_t p = blockIdx.x * kBlock + threadIdx.x;
  if (p >= n) return;
  const uint32_t b = vals ? vals[p] : p;
  uint32_t unit = 0;
  if (axis > 0) {
    const uint32_t blk = p / P.nb;
    uint32_t slab, first, cnt;
    part_deal(P.B, P.fx, blk, &slab, &first, &cnt);
    unit = slab;
    if (axis == 2) {  // the row inside the slab
      uint32_t row, rf, rc;
      part_deal(cnt, min(P.fy, max(cnt, 1u)), blk - first, &row, &rf, &rc);
      unit = slab * P.fy + row;
    }
  }
  const float4 c = fb_c[b];
  auto quant = [&](int ax) -> uint32_t {
    const float v = ax == 0 ? c.x : (ax == 1 ? c.y : c.z);
    const float lo_f = ord_f(sb->lo[ax]), hi_f = ord_f(sb->hi[ax]);
    const float ext = hi_f - lo_f;
    float t = ext > 0.0f ? (v - lo_f) / ext : 0.0f;
    t = t < 0.0f ? 0.0f : (t > 1.0f ? 1.0f : t);
    return (uint32_t)(t * (float)((1u << kPartCoordBits) - 1u));
  };
  const uint32_t cb = P.cb[axis];                              // the level's own coordinate keeps its top cb bits ...
  const uint32_t q = ((quant(axis) >> (kPartCoordBits - cb)) << (kPartCoordBits - cb)) | (quant((axis + 1) % 3) >> cb);  // ... the next axis fills the rest
  keys[p] = (unit << kPartCoordBits) | q;
  if (vals_out) vals_out[p] = b;
}
// ... and inside a block (a run of nb positions of the last level's order) the bodies in Morton order, 7 bits per axis over the
// scene: the kernels of the tick walk the store beside the cell-sorted lists, which are in that order
__global__ __launch_bounds__(kBlock) void k_part_keys_block(uint32_t n, const uint32_t* vals, const float4* fb_c, const SceneBounds* sb, uint32_t nb,
                                                            uint32_t* keys) {
  const uint32_t p = blockIdx.x * kBlock + threadIdx.x;
  if (p >= n) return;
  const float4 c = fb_c[vals[p]];
  uint32_t code = 0;
#pragma unroll
  for (int k = 0; k < 3; ++k) code |= expand10(morton_quant(k == 0 ? c.x : (k == 1 ? c.y : c.z), ord_f(sb->lo[k]), ord_f(sb->hi[k])) >> 3) << (2 - k);
  keys[p] = ((p / nb) << 21) | (code & 0x1FFFFFu);
}

__global__ __launch_bounds__(kBlock) void k_iota(uint32_t* a, uint32_t n) {
  const uint32_t k = blockIdx.x * kBlock + threadIdx.x;
  if (k < n) a[k] = k;
}
// the caller's index of every slot after the move (old_ext = null: the store was in the caller's order), and its inverse
__global__ __launch_bounds__(kBlock) void k_permute_ids(uint32_t n, const uint32_t* order, const uint32_t* old_ext, uint32_t* new_ext, uint32_t* slot_of) {
  const uint32_t k = blockIdx.x * kBlock + threadIdx.x;
  if (k >= n) return;
  const uint32_t o = order[k], e = old_ext ? old_ext[o] : o;
  new_ext[k] = e;
  slot_of[e] = k;
}

__global__ __launch_bounds__(kBlock) void k_kind_mask(const float4* col1, const uint32_t* pcount, uint32_t base, uint32_t m, uint32_t* mask) {
  uint32_t t = blockIdx.x * kBlock + threadIdx.x;
  if (t >= m) return;
  const uint32_t pc = pcount ? pcount[base + t] : 0u;
  atomicOr(mask, pc ? (pc > 2u ? 12u : 4u) : (f2u(col1[base + t].w) == (uint32_t)KIND_SPHERE ? 1u : 2u));
}
// the most parts any of m incoming records carries (a ghost record keeps the count in float 36, a migrant record in word 20): the
// receiving world needs part arrays - and the *_parts kernels for that many slots - BEFORE the records are unpacked
__global__ __launch_bounds__(kBlock) void k_record_max_parts(const float* rec, uint32_t m, uint32_t stride_floats, uint32_t at, uint32_t* out) {
  uint32_t t = blockIdx.x * kBlock + threadIdx.x;
  if (t < m) { const uint32_t pc = f2u(rec[(size_t)t * stride_floats + at]); if (pc) atomicMax(out, pc); }
}
__global__ __launch_bounds__(kBlock) void k_tags_set(float4* ctor, const uint32_t* tags, uint32_t n) {
  uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i < n) ctor[i].w = u2f(tags[i]);
}
__global__ __launch_bounds__(kBlock) void k_tags_get(const float4* ctor, uint32_t* tags, uint32_t n) {
  uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i < n) tags[i] = f2u(ctor[i].w);
}

}  // namespace mgf
