// Persistent dataflow solvers: global (modes 1, 4) and block-local (mode 5).  (Part of the kernel set described in kernels.h.)
#pragma once
#include "k_links.h"

namespace mgf {

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
                                                       const uint32_t* run_if, uint32_t* standby_bar, const uint32_t* canon) {
  // `canon` (the body store is in an internal order, host_perm.inc: constraint ids follow the slots): canon[r] = id of the r-th
  // constraint in insertion order.  A lane must walk ITS nodes in a topological order of the graph (see above) - the insertion
  // order is one, the ids then are not.
  if (run_if && *run_if == 0u) return;  // stand-by launch behind the block-local solver: runs only if that one declined
  const uint32_t C = *C_ptr;
  const uint32_t L = gridDim.x * kBlock;
  const uint32_t gl = blockIdx.x * kBlock + threadIdx.x;
  if (run_if) {
    // The stand-by initialises the arrival counters itself (what k_flow_init does for the other modes), so the common
    // tick does not pay a launch for a path it does not take.  Every block of this launch is resident (the grid is
    // sized from the occupancy), so a counting barrier between the initialisation and the first poll is safe.
    for (uint32_t i = gl; i < C; i += L)
      __hip_atomic_store(&arr[i], 2u - links_indeg0(K, i) * (K.ab[i].y != kNone ? 1u : 2u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(standby_bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      uint32_t sp = 0;
      while (__hip_atomic_load(standby_bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) {
        __builtin_amdgcn_s_sleep(2);
        if (++sp > spin_limit) { __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
      }
    }
    __syncthreads();
  }
  __amdgpu_buffer_rsrc_t rs = make_rsrc(srec);
  uint32_t r = gl, c = 0, round = 0;
  bool done = (r >= C) || iters == 0;
  bool have_rec = false;
  CRec rec;
  uint2 sw = make_uint2(0u, 0u);
  uint32_t spins = 0;
  for (;;) {
    if (!__any(!done)) break;
    bool progressed = false;
    if (!done) {
      // the record is private to this lane: fetch it while the node is still waiting for its predecessors
      if (!have_rec) { c = canon ? canon[r] : r; rec = load_crec(&cons[c]); sw = K.succ[c]; have_rec = true; }
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
        r += L;
        if (r >= C) { r = gl; ++round; if (round >= iters) done = true; }
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
  if (run_if) {  // the last block to leave re-arms the stand-by's barrier
    __syncthreads();
    if (threadIdx.x == 0 && __hip_atomic_fetch_add(standby_bar + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == gridDim.x) {
      __hip_atomic_store(standby_bar, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(standby_bar + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
//   WIDE = false: every slot's constants in LDS (25 B per slot); as many slots as fit beside the block's bodies in 160 KB
//                 (f5_narrow_cap: 3264 for blocks of 1024 bodies);
//   WIDE = true : up to 5120 (a settled 64^3 pile reaches ~4600), only the class-0 slots' constants in LDS (16 B each,
//                 at most 3328), classes 1 + 2 (at most 3072) read theirs from the block's global table.
constexpr uint32_t kF5MaxCons = 5120;                     // rows per block in the global slot tables
constexpr uint32_t kF5LdsBytes = 160u * 1024u;            // LDS of a CU: one workgroup per CU takes all of it
constexpr uint32_t kF5MaxFast = 3328, kF5MaxSlow = 3072;
constexpr uint32_t kF5MaxBodies = 1100;                   // bodies per block (LDS: 64 B each; both layouts must fit 160 KB)
constexpr uint32_t kF5LdsWide = 8u * kF5MaxFast + 9u * kF5MaxCons + 2u * (kF5MaxFast + kF5MaxSlow) + 64u;
// narrow layout: slots that fit beside nb bodies and the two rings (25 B each: succ 8, c 4, aref 4, bref 4, counter 4, round 1);
// the rings keep a power-of-two length (their positions wrap with a mask), which also bounds the slots
constexpr uint32_t kF5NarrowRing = 4096u;
__host__ __device__ constexpr uint32_t f5_narrow_cap(uint32_t nb) {
  const uint32_t fit = ((kF5LdsBytes - 64u - 2u * 2u * kF5NarrowRing - 64u * nb) / 25u) & ~63u;
  return fit < kF5NarrowRing ? fit : kF5NarrowRing;
}
__host__ __device__ constexpr uint32_t f5_lds_narrow(uint32_t cap) { return 25u * cap + 2u * 2u * kF5NarrowRing + 64u; }
static_assert(f5_narrow_cap(kF5MaxBodies) >= 3072u, "narrow layout smaller than it was");
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

// a class-0 slot's successor word in 16 bits (its successors are in-block): slot | two << 13 | wrap << 14 | valid << 15
__device__ __forceinline__ uint32_t f5_pack_succ(uint32_t w) {
  return (w & 0x1FFFu) | ((w & kSuccTwo) ? 0x2000u : 0u) | ((w & kSuccWrap) ? 0x4000u : 0u) | ((w & kSuccLocal) ? 0x8000u : 0u);
}
__device__ __forceinline__ uint32_t f5_unpack_succ(uint32_t h) {
  return (h & 0x1FFFu) | ((h & 0x2000u) ? kSuccTwo : 0u) | ((h & 0x4000u) ? kSuccWrap : 0u) | ((h & 0x8000u) ? kSuccLocal : 0u);
}
template <bool WIDE, bool TRACE>
__global__ __launch_bounds__(kF5Threads) void k_solve_flow5(float4* srec, CRec* cons, ConsLinks K, Flow5 F, uint32_t* arr, uint32_t iters,
                                                            uint32_t* abort_flag, uint32_t spin_limit, uint64_t* trace, uint32_t C_trace) {
  if (*F.fail) return;  // a block did not fit: the stand-by k_solve_flow launch behind this one does the work
  const uint32_t kNarrow = f5_narrow_cap(F.nb);         // (the wide layout's sizes are compile-time constants)
  const uint32_t kMeta = WIDE ? kF5MaxFast : kNarrow;   // slots with constants in LDS
  const uint32_t kAll = WIDE ? kF5MaxCons : kNarrow;    // slots with counters in LDS
  constexpr uint32_t kRingF = WIDE ? kF5MaxFast : kF5NarrowRing, kRingS = WIDE ? kF5MaxSlow : kF5NarrowRing;
  extern __shared__ float4 s_dyn[];
  float4* s_body = s_dyn;  // 4 x nb
  // narrow: succ (8 B), c, aref, bref for every slot.  wide: the constraint id of EVERY slot (so a slow node's record
  // fetch does not wait for its table row: the row and the record travel together), and for the class-0 slots the body
  // refs (two 16-bit LDS indices) and both successor words packed into 32 bits (their successors are always in-block:
  // slot 13 bits | "two predecessors" | wrap | valid, per half).
  uint32_t* s_w0 = reinterpret_cast<uint32_t*>(s_dyn + 4 * (size_t)F.nb);
  uint2* s_succ = reinterpret_cast<uint2*>(s_w0);                          // narrow: [kMeta]
  uint32_t* s_succ32 = s_w0;                                               // wide:   [kMeta]
  uint32_t* s_c = WIDE ? s_w0 + kMeta : s_w0 + 2 * kMeta;                  // [kAll] (wide) / [kMeta] (narrow)
  uint32_t* s_a = s_c + (WIDE ? kAll : kMeta);                             // [kMeta] WIDE: aref | bref << 16; else aref
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
    if (WIDE) s_c[idx] = r0.x;
    if (idx < n_meta) {
      const uint4 r1 = src[1];  // successor words
      uint32_t ar = r0.y, br = r0.z;
      if (WIDE) {
        s_a[idx] = (ar & 0xFFFFu) | ((br == kNone ? 0xFFFFu : br) << 16);  // class 0: LDS indices or static
        s_succ32[idx] = f5_pack_succ(r1.x) | (f5_pack_succ(r1.y) << 16);
      } else {
        s_c[idx] = r0.x; s_a[idx] = ar; s_b[idx] = br;
        s_succ[idx] = make_uint2(r1.x, r1.y);
      }
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
        if (WIDE) {
          const uint32_t ms = min(slot, kMeta - 1u);
          c = s_c[slot];
          const uint32_t ab = s_a[ms], s32 = s_succ32[ms];
          aref = ab & 0xFFFFu; bref = (ab >> 16) == 0xFFFFu ? kNone : (ab >> 16);
          sw = make_uint2(f5_unpack_succ(s32 & 0xFFFFu), f5_unpack_succ(s32 >> 16));
          asm volatile("" : "+v"(c), "+v"(sw.x), "+v"(sw.y), "+v"(aref), "+v"(bref));  // keeps the LDS reads above the branch (else: sunk and merged into flat loads)
          if (slot >= n_meta) {  // a slow slot: its refs and successor words come from the block's table row (the record fetch below does not wait for it)
            const uint4* src = reinterpret_cast<const uint4*>(&F.table[row0 + slot]);
            const uint4 r0 = src[0], r1 = src[1];
            aref = r0.y; bref = r0.z; sw = make_uint2(r1.x, r1.y);
          }
        } else {
          c = s_c[slot]; sw = s_succ[slot];
          aref = s_a[slot]; bref = s_b[slot];
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

}  // namespace mgf
