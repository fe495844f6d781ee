// Block-local dataflow solver with message channels (solver mode 6).  (Part of the kernel set described in kernels.h.)
#pragma once
#include "k_solver_flow.h"

namespace mgf {

// ------------------------------------------------------------------------------------------
// Same dependency graph, same arrival-counter protocol and the same spatial blocks as k_solve_flow5 (bodies in cell order,
// `nb` per block, one 512-thread workgroup per block, one block per CU, a constraint belongs to the block of its body a) -
// but NO body record is exchanged through global memory while the solve runs:
//   * every body a block's constraints touch has a slot in that block's LDS: its own `nb` bodies and the FOREIGN bodies its
//     constraints meet as `b` (velocity only, 32 B; the constant inverse mass / inertia come from the RigidBodyVec by plain
//     cached loads that travel beside the constraint record's);
//   * every constraint's arrival counter is in LDS, and every node runs LDS-to-LDS (one class, one ready queue);
//   * a dependency edge that crosses a block face is a 48-byte MESSAGE: the producer writes the body's new velocity, the
//     successor's slot and a tag with three write-through 16-byte stores into a FIFO channel in global memory (one channel
//     per ordered pair of neighbouring blocks, position taken from an LDS counter) and goes on - no store acknowledgement, no
//     separate flag (MI355X guide, price list "handoff-1to1" vs "handoff-flag"); the consumer block's polling wave reads
//     the next few positions of each of its incoming channels, and when all three granules of a message carry this launch's
//     tag it copies the velocity into the block's LDS slot of that body and counts the arrival like a local one.
// A body's velocity therefore travels along its chain of constraints: LDS -> (message) -> LDS.  The constraint that ends a
// body's chain in the last iteration writes the result to the RigidBodyVec if the body is foreign to its block; every block
// writes back its own bodies except those ("skipwb").
// Deadlock-free like k_solve_flow5: every block is resident, every ready node is eventually taken, every message is
// eventually seen.  Bit-identical to the sequential order: any topological order of the graph is.
// ------------------------------------------------------------------------------------------
#ifndef MGF_F6_THREADS
#define MGF_F6_THREADS 768  // 12 waves: 11 serving + 1 polling (r04: with quad trips - 16 nodes per wave trip - the settled pile gains 4 % over 8 waves, the other scenes are within 1 %; 6 waves lose; a run-time size costs 2 %)
#endif
constexpr int kF6Threads = MGF_F6_THREADS;
constexpr uint32_t kF6Chan = 64;                       // neighbour blocks per block, each direction (hash slots)
constexpr uint32_t kF6SlotBits = 13, kF6BodyBits = 11;
constexpr uint32_t kF6MaxSlots = (1u << kF6SlotBits) - 1u;  // per block
constexpr uint32_t kF6NoBody = (1u << kF6BodyBits) - 1u;    // b-ref of a constraint against a Static body
constexpr uint32_t kF6Remote = 0x80000000u, kF6Wrap = 0x40000000u;
// a slot's state word in LDS: arrivals still missing (bits 0-1: 0..2), iterations done (bits 2-8: <= kF6MaxIters), and - constant -
// the row's body references (bits 10-31)
constexpr uint32_t kF6StArrMask = 3u, kF6StIterShift = 2u, kF6StIterMask = 0x7Fu, kF6StRefShift = 10u;
static_assert(2u * kF6BodyBits + kF6StRefShift <= 32u, "the body references fit above the counters");
constexpr uint32_t kF6MsgWords = 3;                    // uint4 granules per message
constexpr uint32_t kF6MaxIters = 64;                   // (ring positions are reduced with a 32-bit reciprocal; 7 bits in the state word)
constexpr uint32_t kF6WlLen = 128;                     // work items (channel, position) a polling wave lists per sweep
constexpr uint32_t kF6MaxPollers = 4;
constexpr uint32_t kF6CntStride = 32;                  // words between per-block counters (same-line atomics serialise)
constexpr uint32_t kF6TraceWords = 32;                 // TRACE: 64-bit words of statistics per block behind the node stamps (polling: 0-6; clocks: 8 entry,
                                                       // 9 bodies in LDS, 10 tables in LDS, 11 serving loop left, 12 written back; 24-29 messages by latency, 30 seen by quiet sweeps)

// One 32-byte row per slot, written once per tick - the constraint's own part and the links inside a body's own range by
// k_flow6_blocks, the links that need the body's whole chain by k_flow6_links - and copied into LDS by the solve kernel.
struct F6Row {
  uint32_t c;        // constraint id
  uint32_t ref;      // a's LDS index (bits 0-10) | b's (bits 11-21, kF6NoBody = Static)
  uint32_t succ0, succ1;  // local: slot | kF6Wrap;  remote: kF6Remote | kF6Wrap | out channel << 24 | LDS index of the body in the
                          // successor's block << 13 | the successor's slot there
  uint32_t pred_a;   // 1: the constraint has a predecessor on body a inside an iteration (k_flow6_blocks)
  uint32_t pred_b;   // ... on body b (1 by default, cleared by k_flow6_links where the chain starts); their sum = arrivals missing in iteration 0
  uint32_t pad[2];
};
static_assert(sizeof(F6Row) == 32, "F6Row is two 16-byte words");

struct Flow6 {
  const uint32_t* C_ptr;     // constraints of the list (0: the tick's collide phase failed a capacity check and is re-run: nothing to do)
  const uint32_t* sidx;      // cell-ordered body ids
  const uint32_t* brank;     // body -> position in cell order
  const uint32_t* base;      // body -> id of its first own constraint (canonical order: a body's `a` constraints are contiguous)
  uint4* binfo;              // body -> (position in cell order, slot of its first own constraint inside its block, id of that constraint, -):
                             // what a constraint's row needs of a body, one 16-byte look-up instead of three
  uint32_t* bref;            // constraint -> LDS index of its body b in the constraint's block
  uint8_t* skipwb;           // body -> 1: the chain's last constraint runs in another block, which writes the result
  uint32_t* fcnt;            // per block (stride kF6CntStride): foreign bodies
  uint32_t* fbody;           // [nblocks * fcap] their ids
  uint32_t* nslots;          // per block (stride kF6CntStride): constraints
  F6Row* table;              // [nblocks * rows]
  uint32_t* in_key;          // [nblocks * kF6Chan] incoming channels of a block: source block + 1 (0 = empty hash slot)
  uint32_t* in_cnt;          // ... and the edges they carry per iteration
  uint32_t* out_key;         // outgoing: destination block + 1
  uint32_t* out_val;         // ... and the hash slot of this block in the destination's in_key
  uint32_t* chan_prefix;     // [nblocks * kF6Chan] exclusive prefix of in_cnt inside the block: a channel's first message (per iteration)
  unsigned long long* tails; // [nblocks * kF6Chan] per incoming channel: (launch tag << 32) | positions handed out so far - a HINT that lets
                             // the consumer read only what was sent (a message counts when its own granules carry the tag)
  uint4* mbox;               // the channels: every block owns mbox_cap / nblocks messages; a channel starts chan_prefix * iters into
                             // its consumer's region; kF6MsgWords granules per message
  uint32_t mbox_cap;         // messages the buffer holds
  uint32_t* fail;            // a limit was exceeded (1 foreign slots, 2 constraint slots, 4 channels, 8 channel buffer, 16 index width):
                             // the stand-by k_solve_flow launch does the work; fail[1] = edges across block faces per iteration
  uint32_t* tick_fail;       // the tick's StepCounts::fail: gets kFailFlow6 when `fail` is up after the preparation (the host re-runs the tick)
  uint32_t* max_slots;       // largest block / most foreign bodies of this tick (the host sizes the next tick's LDS split)
  uint32_t* max_foreign;
  uint32_t ident;            // 1: sidx and brank are the identity (blocks cut from the slots of a re-sorted store): their look-ups are skipped
  uint32_t nb, nblocks, n;
  uint32_t rows;             // table rows per block
  uint32_t fcap, slot_cap;   // LDS split of this launch: foreign body slots, constraint slots
  uint32_t poll_waves, poll_k;  // waves that poll the incoming channels; 1: every sweep through the worklist, >= 2: quiet sweeps read straight
  uint32_t quad_max;            // QD: a wave takes a four-lanes-per-node trip while the ready queue holds at most this many nodes
  uint32_t poll_prio;           // s_setprio of the polling waves (0..3): their few instructions issue ahead of the serving waves'
  uint32_t poll_spec_wl;        // worklist sweeps: positions behind the producers' hint read on spec (0: the head only)
  uint32_t poll_spec;           // positions behind a channel's head a quiet sweep reads on spec (1..8; the wave's idle lanes take them)
};
constexpr uint32_t kF6RecWords = 5;  // RL: float4 words of a constraint's solver half in LDS (80 bytes: CRec words 2..20 and the accumulated impulse)
// successor words 8, id 4, state 4, ring 2 (+ impulse 4; or + the record's solver half 80, which holds the impulse)
__host__ __device__ constexpr uint32_t f6_slot_bytes(bool nimp_lds, bool rec_lds = false) { return rec_lds ? 18u + 16u * kF6RecWords : (nimp_lds ? 22u : 18u); }
__host__ __device__ constexpr uint32_t f6_lds_bytes(uint32_t nb, uint32_t fcap, uint32_t slot_cap, int const_lds = 0, bool nimp_lds = false, bool rec_lds = false) {
  return 32u * (nb + fcap) + (const_lds == 2 ? 40u * (nb + fcap) : const_lds == 1 ? 40u * nb : 0u) + f6_slot_bytes(nimp_lds, rec_lds) * slot_cap + 4u * (16u + 8u * kF6Chan) + 4u * (2u * kF6WlLen + 16u) + 32u;
}

// ---- preparation, once per constraint list ------------------------------------------------------------------------------
// Per block, one workgroup: (1) the slot of every own body's first constraint (prefix sum of the own-constraint counts in cell
// order); (2) the block's FOREIGN bodies - bodies of other blocks that its constraints meet as `b` - numbered through an LDS
// hash set (no global counters: 256 contended words cost 160 us here), and for every constraint of the block the LDS index
// of its body b.
constexpr uint32_t kF6Hash = 4096;
constexpr uint32_t kF6PrepThreads = 1024;
__global__ __launch_bounds__(kF6PrepThreads) void k_flow6_blocks(Flow6 F, ConsLinks K) {
  constexpr uint32_t kBlock = kF6PrepThreads;  // (this kernel's own block size)
  __shared__ uint32_t s_wave[kBlock / 64];
  __shared__ uint32_t s_key[kF6Hash], s_val[kF6Hash];
  __shared__ uint32_t s_cnt, s_run;
  const uint32_t g = blockIdx.x, t = threadIdx.x, lane = t & 63u, wv = t >> 6;
  if (*F.C_ptr == 0u) return;  // (an empty or failed list: the solve kernel does nothing either)
  const uint32_t p_lo = g * F.nb, p_hi = min(F.n, p_lo + F.nb);
  for (uint32_t e = t; e < kF6Hash; e += kBlock) { s_key[e] = 0u; s_val[e] = 0u; }
  if (t == 0) { s_cnt = 0u; s_run = 0u; }
  __syncthreads();
  const bool single = p_hi - p_lo <= kBlock;  // one body per thread: what the first pass looked up serves the second
  uint32_t keep_b[4] = {kNone, kNone, kNone, kNone}, keep_pb[4] = {0u, 0u, 0u, 0u};
  // (1) slots: a running prefix over the block's bodies in cell order, kBlock bodies per round (wave scan + the waves before)
  // (2) on the way, the foreign bodies among each body's partners go into the hash set
  for (uint32_t p0 = p_lo; p0 < p_hi; p0 += kBlock) {
    const uint32_t p = p0 + t;
    uint32_t x = 0, b0 = 0, cnt = 0;
    if (p < p_hi) { x = F.ident ? p : F.sidx[p]; b0 = F.base[x]; cnt = F.base[x + 1] - b0; }
    uint32_t inc = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t u = __shfl_up(inc, o); if ((int)lane >= o) inc += u; }
    if (lane == 63u) s_wave[wv] = inc;
    __syncthreads();
    uint32_t before = s_run, total = 0;
    for (uint32_t k = 0; k < kBlock / 64; ++k) { const uint32_t u = s_wave[k]; if (k < wv) before += u; total += u; }
    if (p < p_hi) F.binfo[x] = make_uint4(p, before + inc - cnt, b0, cnt);
    for (uint32_t k0 = 0; k0 < cnt; k0 += 4u) {  // four partners' look-ups in flight; the first four are kept for the rows below
      uint32_t b[4], pb[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = k0 + (uint32_t)j < cnt ? K.ab[b0 + k0 + (uint32_t)j].y : kNone;
#pragma unroll
      for (int j = 0; j < 4; ++j) pb[j] = b[j] != kNone ? (F.ident ? b[j] : F.brank[b[j]]) : 0u;
      if (k0 == 0u) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { keep_b[j] = b[j]; keep_pb[j] = pb[j]; }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (b[j] == kNone || pb[j] / F.nb == g) continue;
        uint32_t i = (b[j] * 2654435761u) >> 20;
        for (uint32_t probe = 0; probe < kF6Hash; ++probe, i = (i + 1u) & (kF6Hash - 1u)) {
          const uint32_t cur = atomicCAS(&s_key[i], 0u, b[j] + 1u);
          if (cur == 0u || cur == b[j] + 1u) break;
        }
      }
    }
    __syncthreads();
    if (t == 0) s_run += total;
    __syncthreads();
  }
  if (t == 0) {
    const uint32_t total = s_run;
    F.nslots[(size_t)g * kF6CntStride] = total;
    atomicMax(F.max_slots, total);
    if (total > F.slot_cap || total > kF6MaxSlots) atomicOr(F.fail, 2u);
  }
  for (uint32_t e = t; e < kF6Hash; e += kBlock) {
    if (s_key[e]) {
      const uint32_t k = atomicAdd(&s_cnt, 1u);
      s_val[e] = k;
      if (k < F.fcap) F.fbody[(size_t)g * F.fcap + k] = s_key[e] - 1u;
    }
  }
  __syncthreads();
  if (t == 0) {
    F.fcnt[(size_t)g * kF6CntStride] = s_cnt;
    atomicMax(F.max_foreign, s_cnt);
    if (s_cnt > F.fcap || s_cnt >= kF6Hash / 2u) atomicOr(F.fail, 1u);
  }
  // every own constraint's row: id, LDS indices of its bodies, the link to the next constraint of body a's own range (the
  // last of the range is linked by k_flow6_links, which knows where the body's chain goes on)
  for (uint32_t p = p_lo + t; p < p_hi; p += kBlock) {
    const uint32_t x = F.ident ? p : F.sidx[p];
    const uint4 ix = F.binfo[x];
    const uint32_t aref = p - g * F.nb;
    if (aref >= kF6NoBody) atomicOr(F.fail, 16u);
    for (uint32_t k0 = 0; k0 < ix.w; k0 += 4u) {  // four constraints' look-ups in flight
      uint32_t b[4], pb[4];
      if (single && k0 == 0u) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { b[j] = keep_b[j]; pb[j] = keep_pb[j]; }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = k0 + (uint32_t)j < ix.w ? K.ab[ix.z + k0 + (uint32_t)j].y : kNone;
#pragma unroll
        for (int j = 0; j < 4; ++j) pb[j] = b[j] != kNone ? (F.ident ? b[j] : F.brank[b[j]]) : 0u;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t k = k0 + (uint32_t)j;
        if (k >= ix.w) break;
        const uint32_t c = ix.z + k, slot = ix.y + k;
        uint32_t bref = kF6NoBody;
        if (b[j] != kNone) {
          if (pb[j] / F.nb == g) bref = pb[j] - g * F.nb;
          else {
            uint32_t i = (b[j] * 2654435761u) >> 20;
            while (s_key[i] != b[j] + 1u) i = (i + 1u) & (kF6Hash - 1u);
            bref = F.nb + s_val[i];
          }
          F.bref[c] = bref;
          if (bref >= kF6NoBody) atomicOr(F.fail, 16u);
        }
        if (slot >= F.slot_cap || slot >= kF6MaxSlots) continue;  // (fail bit 2 is up)
        uint4* dst = reinterpret_cast<uint4*>(&F.table[(size_t)g * F.rows + slot]);
        dst[0] = make_uint4(c, aref | (min(bref, kF6NoBody) << kF6BodyBits), k + 1u < ix.w ? slot + 1u : 0u, 0u);
        dst[1] = make_uint4(k > 0u ? 1u : 0u, b[j] != kNone ? 1u : 0u, 0u, 0u);  // (pred_b: all but the first constraint of a chain without own ones - k_flow6_links clears that one)
      }
    }
  }
}
// hash slot of `key1` (= key + 1, never 0) in a table of kF6Chan words; inserts it if absent
__device__ __forceinline__ uint32_t f6_chan_slot(uint32_t* keys, uint32_t key1, uint32_t* fail) {
  const uint32_t h = (key1 * 2654435761u) >> 26;
  for (uint32_t probe = 0; probe < kF6Chan; ++probe) {
    const uint32_t i = (h + probe) & (kF6Chan - 1u);
    uint32_t cur = __hip_atomic_load(&keys[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (cur == 0u) cur = atomicCAS(&keys[i], 0u, key1);
    if (cur == 0u || cur == key1) return i;
  }
  atomicOr(fail, 4u);
  return 0u;
}
// Per body, in cell order: the links of its chain that leave its own range of constraints - last own constraint -> first
// constraint it takes part in as `b` (its row of such constraints, written by k_setup_pairs in arrival order, sorted here)
// -> ... -> back to the first (the wrap to the next iteration) - written straight into the rows of the block tables, with
// the channel of every link that crosses a block face.  This is k_chain_rows and the old per-constraint table kernel in
// one: no (succ, pred) arrays in between (k_chain_rows still builds them for the other solver modes, and - launched behind
// this kernel with a guard - for the stand-by when a block did not fit).
struct F6Ent { uint32_t home, slot, role, c, bref; };
__device__ __forceinline__ void f6_links_body(const Flow6& F, const ConsLinks& K, uint32_t n, const uint32_t* degb, RevEnt* rev, uint32_t rev_cap,
                                              const uint32_t* rev_flag, StepCounts* sc, uint32_t n_owned, uint32_t* n_ghost_cons, const uint32_t* ext) {
  const uint32_t t = blockIdx.x * kBlock + threadIdx.x;
  // constraints whose obj_a is a ghost (ids are ascending in obj_a): the copies of seam constraints (tiles count them once)
  if (t == 0) *n_ghost_cons = (sc->fail || *rev_flag) ? 0u : F.base[n] - F.base[n_owned];
  if (*rev_flag) {
    if (t == 0) { sc->C = 0; sc->Ct = 0; sc->fail |= kFailRevRow; }
    return;
  }
  if (t >= n || sc->fail) return;
  const uint32_t x = F.ident ? t : F.sidx[t];
  const uint4 ix = F.binfo[x];  // (position = t, first slot, first constraint, constraints of its own)
  const uint32_t g = t / F.nb, na = ix.w, nbr = degb[x];
  if (na + nbr == 0u) return;
  RevEnt* row = rev + (size_t)x * rev_cap;
  // the row in insertion order - ascending (order id of the constraint's body a, constraint id): up to eight entries (a settled pile's bodies
  // have four on average) sorted in registers, longer rows where they lie.  The entries carry their keys and body a's slot (r05): no
  // look-up through the constraint id, neither for the sort nor for the links below
  const bool small = nbr <= 8u;
  uint32_t s8[8] = {kNone, kNone, kNone, kNone, kNone, kNone, kNone, kNone}, a8[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
  if (small) {
    unsigned long long k8[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      RevEnt e{kNone, 0u, 0xFFFFFFFFu, 0u};
      if ((uint32_t)j < nbr) e = row[j];
      k8[j] = rev_key(e); a8[j] = e.a;  // (kNone's key - all ones - sorts last)
    }
    auto cx = [&](int p, int q) {
      const bool sw = k8[p] > k8[q];
      const unsigned long long lo = sw ? k8[q] : k8[p], hi = sw ? k8[p] : k8[q];
      const uint32_t al = sw ? a8[q] : a8[p], ah = sw ? a8[p] : a8[q];
      k8[p] = lo; k8[q] = hi; a8[p] = al; a8[q] = ah;
    };
    // Batcher's odd-even merge sort of eight (nineteen exchanges)
    cx(0, 1); cx(2, 3); cx(4, 5); cx(6, 7);
    cx(0, 2); cx(1, 3); cx(4, 6); cx(5, 7);
    cx(1, 2); cx(5, 6);
    cx(0, 4); cx(1, 5); cx(2, 6); cx(3, 7);
    cx(2, 4); cx(3, 5);
    cx(1, 2); cx(3, 4); cx(5, 6);
#pragma unroll
    for (int j = 0; j < 8; ++j) s8[j] = (uint32_t)k8[j];  // (the id is the key's low half; ~0 = kNone)
  } else {
    rev_sort_in_place(row, nbr);
  }
  // the chain: [last own constraint,] b_0 .. b_{nbr-1}, and back to its FIRST constraint (first own, else b_0)
  F6Ent first;
  first.home = g; first.slot = ix.y; first.role = 0u; first.c = ix.z; first.bref = 0u;  // (na > 0; else set from the first `b` entry below)
  F6Ent u = first;
  if (na) { u.slot = ix.y + na - 1u; u.c = ix.z + na - 1u; }
  bool have_u = na != 0u;
  // the `b` entries four at a time: their look-ups (constraint -> body a -> block, slot; LDS index of x over there) are
  // independent of each other and go out together; the links are then written in order
  // the `b` entries' look-ups (constraint -> body a's block and slot there; LDS index of x over there) go out together - all eight of a
  // short row before the first link is written (r05) - the links are then written in order
  auto look4 = [&](const uint32_t c[4], const uint32_t ca[4], uint4 ii[4], uint32_t br[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      ii[j] = make_uint4(0, 0, 0, 0); br[j] = 0u;
      if (c[j] != kNone) { ii[j] = F.binfo[ca[j]]; br[j] = F.bref[c[j]]; }  // the constraint lives in its body a's block
    }
  };
  auto link4 = [&](uint32_t k0, const uint32_t c[4], const uint4 ii[4], const uint32_t br[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t kb = k0 + (uint32_t)j;
      if (kb > nbr) break;
      const bool last = kb == nbr;
      F6Ent w;
      if (last) w = first;
      else { w.c = c[j]; w.role = 1u; w.home = ii[j].x / F.nb; w.slot = ii[j].y + (c[j] - ii[j].z); w.bref = br[j]; }
      if (!have_u) {  // (no own constraints: the chain starts at the first `b` entry, which has no predecessor on this body)
        first = w; u = w; have_u = true;
        if (w.slot < F.rows) reinterpret_cast<uint32_t*>(&F.table[(size_t)w.home * F.rows + w.slot])[5] = 0u;  // pred_b (1 by default)
        continue;
      }
      const uint32_t wrap = last ? kF6Wrap : 0u;
      uint32_t word;
      if (u.home == w.home) {
        word = wrap | w.slot;
      } else {  // the link crosses a block face: a message on the channel u.home -> w.home
        const uint32_t dbody = w.role == 0u ? t - w.home * F.nb : w.bref;  // as a: own there; as b: what k_flow6_blocks gave it
        uint32_t* ik = F.in_key + (size_t)w.home * kF6Chan;
        uint32_t* ok = F.out_key + (size_t)u.home * kF6Chan;
        // both hash tables at their home slots first (one round trip); the probing / inserting path only on a miss
        const uint32_t hi = ((u.home + 1u) * 2654435761u) >> 26, ho = ((w.home + 1u) * 2654435761u) >> 26;
        const uint32_t ci = __hip_atomic_load(&ik[hi], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t co = __hip_atomic_load(&ok[ho], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t k_in = ci == u.home + 1u ? hi : f6_chan_slot(ik, u.home + 1u, F.fail);
        const uint32_t k_out = co == w.home + 1u ? ho : f6_chan_slot(ok, w.home + 1u, F.fail);
        atomicAdd(&F.in_cnt[(size_t)w.home * kF6Chan + k_in], 1u);
        F.out_val[(size_t)u.home * kF6Chan + k_out] = k_in;
        if (w.slot >= kF6MaxSlots || dbody >= kF6NoBody) atomicOr(F.fail, 16u);
        word = kF6Remote | wrap | (k_out << 24) | (dbody << kF6SlotBits) | w.slot;
      }
      if (u.slot < F.rows) reinterpret_cast<uint32_t*>(&F.table[(size_t)u.home * F.rows + u.slot])[2 + u.role] = word;
      // the chain ends in a constraint of another block: that block writes the body's result, its own does not
      if (last && u.role == 1u && u.home != g) F.skipwb[x] = 1;
      u = w;
    }
  };
  if (small) {
    const uint32_t lo4[4] = {s8[0], s8[1], s8[2], s8[3]}, hi4[4] = {s8[4], s8[5], s8[6], s8[7]}, none[4] = {kNone, kNone, kNone, kNone};
    const uint32_t la4[4] = {a8[0], a8[1], a8[2], a8[3]}, ha4[4] = {a8[4], a8[5], a8[6], a8[7]};
    uint4 il[4], ih[4];
    uint32_t bl[4], bh[4];
    look4(lo4, la4, il, bl);
    look4(hi4, ha4, ih, bh);
    link4(0u, lo4, il, bl);
    if (nbr >= 4u) link4(4u, hi4, ih, bh);
    if (nbr == 8u) link4(8u, none, il, bl);
  } else {
    for (uint32_t k0 = 0; k0 <= nbr; k0 += 4u) {
      uint32_t c[4], ca[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        c[j] = kNone; ca[j] = 0u;
        if (k0 + (uint32_t)j < nbr) { const RevEnt e = row[k0 + (uint32_t)j]; c[j] = e.c; ca[j] = e.a; }
      }
      uint4 ii[4];
      uint32_t br[4];
      look4(c, ca, ii, br);
      link4(k0, c, ii, br);
    }
  }
}
// The channel layout of one block (what k_flow6_chan does with a wave, here one thread): exclusive prefix of the block's incoming
// edge counts; returns their sum.  The counts were built by device-scope atomics: read past L1 / a stale L2 line (sc1).
__device__ __forceinline__ uint32_t f6_chan_one(const Flow6& F, uint32_t hs) {
  __amdgpu_buffer_rsrc_t rc = make_rsrc(F.in_cnt);
  v4f_t c[kF6Chan / 4];
#pragma unroll
  for (uint32_t k = 0; k < kF6Chan / 4; ++k) c[k] = __builtin_amdgcn_raw_buffer_load_b128(rc, (int)((hs * kF6Chan + 4u * k) * 4u), 0, kSc1);
  uint32_t run = 0;
  uint4* dst = reinterpret_cast<uint4*>(F.chan_prefix + (size_t)hs * kF6Chan);
#pragma unroll
  for (uint32_t k = 0; k < kF6Chan / 4; ++k) {
    uint4 o;
    o.x = run; run += f2u(c[k].x); o.y = run; run += f2u(c[k].y); o.z = run; run += f2u(c[k].z); o.w = run; run += f2u(c[k].w);
    dst[k] = o;
  }
  return run;
}
// The tick's launch: the links, and - by the block that finishes last (a ticket; the edge counts are complete when every block's
// atomics have been acknowledged) - the channel layout, which used to be a launch of its own (k_flow6_chan: still there for a
// caller's list and for a changed iteration count).
__global__ __launch_bounds__(kBlock) void k_flow6_links(Flow6 F, ConsLinks K, uint32_t n, const uint32_t* degb, RevEnt* rev, uint32_t rev_cap,
                                                        const uint32_t* rev_flag, StepCounts* sc, uint32_t n_owned, uint32_t* n_ghost_cons,
                                                        const uint32_t* ext, uint32_t* ticket, uint32_t iters, const float4* srec, float4* vsnap) {
  {  // the velocities as Solver::solve is about to find them (k_solver_snapshot's work on the way: the solve follows this launch in the fused tick)
    const uint32_t t = blockIdx.x * kBlock + threadIdx.x;
    // (not in a tick that is being skipped - a speculative one behind a tick that failed or gave up: the copy in place is that tick's)
    if (vsnap && t < n && !sc->fail) { vsnap[2 * (size_t)t] = srec[4 * (size_t)t]; vsnap[2 * (size_t)t + 1] = srec[4 * (size_t)t + 1]; }
  }
  f6_links_body(F, K, n, degb, rev, rev_cap, rev_flag, sc, n_owned, n_ghost_cons, ext);
  if (!ticket) return;
  __shared__ uint32_t s_last;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's atomics on the edge counts are done ...
  __syncthreads();                                   // ... and the block's
  if (threadIdx.x == 0) s_last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u ? 1u : 0u;
  __syncthreads();
  if (!s_last) return;
  // (the totals through the block, not through 256 atomics on one word: those alone took 6 us)
  __shared__ uint32_t s_sum[kBlock / 64], s_max[kBlock / 64];
  uint32_t sum = 0, mx = 0;
  for (uint32_t hs = threadIdx.x; hs < F.nblocks; hs += kBlock) { const uint32_t run = f6_chan_one(F, hs); sum += run; mx = max(mx, run); }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) { sum += __shfl_xor(sum, o); mx = max(mx, (uint32_t)__shfl_xor(mx, o)); }
  if ((threadIdx.x & 63u) == 0u) { s_sum[threadIdx.x >> 6] = sum; s_max[threadIdx.x >> 6] = mx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    sum = 0; mx = 0;
    for (uint32_t k = 0; k < kBlock / 64; ++k) { sum += s_sum[k]; mx = max(mx, s_max[k]); }
    atomicAdd(&F.fail[1], sum);  // edges that cross a block face, per iteration (the host sizes the channel buffer from it)
    atomicMax(&F.fail[2], mx);   // ... the most any block receives
    if ((uint64_t)mx * iters > F.mbox_cap / F.nblocks) atomicOr(F.fail, 8u);
    *ticket = 0u;  // (re-armed for the next tick)
    if (__hip_atomic_load(F.fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) atomicOr(F.tick_fail, kFailFlow6);
  }
}
// One wave per block (its kF6Chan hash slots = the wave's lanes): where each incoming channel's messages start inside the
// block's region of the channel buffer (exclusive prefix of the per-iteration edge counts), and whether the region suffices.
__global__ __launch_bounds__(kBlock) void k_flow6_chan(Flow6 F, uint32_t iters) {
  static_assert(kF6Chan == 64, "one lane per hash slot");
  const uint32_t hs = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
  if (hs >= F.nblocks) return;
  const uint32_t cnt = F.in_cnt[(size_t)hs * kF6Chan + lane];
  uint32_t incl = cnt;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const uint32_t v = __shfl_up(incl, d); if ((int)lane >= d) incl += v; }
  F.chan_prefix[(size_t)hs * kF6Chan + lane] = incl - cnt;
  if (lane == 63u) {
    atomicAdd(&F.fail[1], incl);  // edges that cross a block face, per iteration (the host sizes the channel buffer from it)
    atomicMax(&F.fail[2], incl);  // ... the most any block receives
    if ((uint64_t)incl * iters > F.mbox_cap / F.nblocks) atomicOr(F.fail, 8u);
    // the preparation ends here: whatever went wrong in it is now visible - tell the tick (its read-back carries the word)
    if (__hip_atomic_load(F.fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) atomicOr(F.tick_fail, kFailFlow6);
  }
}

// ---- the same tables for ANY insertion-ordered list (a caller's: mgf_world_set_constraints, mgf_solver_*) ----------------------
// The tick's own list has structure the kernels above lean on (a body's own constraints are a contiguous id range, the rest of its
// chain is its row of `b` occurrences).  A caller's list has none: its dependency links come from the generic adjacency build
// (k_adj_fill + k_chain: successor words and predecessor flags per constraint), and the block tables are derived from those links:
//   k_flow6g_assign  a constraint belongs to the block of its body a; its slot there is its arrival rank; the block's list of ids;
//   k_flow6g_blocks  per block: the foreign bodies (LDS hash set, as above), every row's constraint id and body references;
//   k_flow6g_links   per constraint: its two successor words translated - (block, slot) of the successor, and for a successor in
//                    another block the channel and the LDS index the carried body has over there - and its predecessor flags.
// k_flow6_chan and the solve kernel are the same.  Bodies are grouped by `brank` (cell order, or the slots of a re-sorted store).
__global__ __launch_bounds__(kBlock) void k_flow6g_assign(Flow6 F, ConsLinks K, uint32_t* cslot, uint32_t* clist) {
  const uint32_t c = blockIdx.x * kBlock + threadIdx.x;
  if (c >= *F.C_ptr) return;
  const uint32_t g = F.brank[K.ab[c].x] / F.nb;
  const uint32_t slot = atomicAdd(&F.nslots[(size_t)g * kF6CntStride], 1u);
  cslot[c] = slot;
  if (slot < F.rows && slot < kF6MaxSlots) clist[(size_t)g * F.rows + slot] = c;
  else atomicOr(F.fail, 2u);
}
__global__ __launch_bounds__(kF6PrepThreads) void k_flow6g_blocks(Flow6 F, ConsLinks K, const uint32_t* clist) {
  constexpr uint32_t kBlock = kF6PrepThreads;
  __shared__ uint32_t s_key[kF6Hash], s_val[kF6Hash];
  __shared__ uint32_t s_cnt;
  const uint32_t g = blockIdx.x, t = threadIdx.x;
  if (*F.C_ptr == 0u) return;
  for (uint32_t e = t; e < kF6Hash; e += kBlock) { s_key[e] = 0u; s_val[e] = 0u; }
  if (t == 0) s_cnt = 0u;
  __syncthreads();
  const uint32_t total = F.nslots[(size_t)g * kF6CntStride], N = min(min(total, F.rows), kF6MaxSlots);
  if (t == 0) { atomicMax(F.max_slots, total); if (total > F.slot_cap || total > kF6MaxSlots) atomicOr(F.fail, 2u); }
  const uint32_t* list = clist + (size_t)g * F.rows;
  for (uint32_t sl = t; sl < N; sl += kBlock) {  // the foreign bodies among the block's partners
    const uint32_t b = K.ab[list[sl]].y;
    if (b == kNone || F.brank[b] / F.nb == g) continue;
    uint32_t i = (b * 2654435761u) >> 20;
    for (uint32_t probe = 0; probe < kF6Hash; ++probe, i = (i + 1u) & (kF6Hash - 1u)) {
      const uint32_t cur = atomicCAS(&s_key[i], 0u, b + 1u);
      if (cur == 0u || cur == b + 1u) break;
    }
  }
  __syncthreads();
  for (uint32_t e = t; e < kF6Hash; e += kBlock) {
    if (s_key[e]) {
      const uint32_t k = atomicAdd(&s_cnt, 1u);
      s_val[e] = k;
      if (k < F.fcap) F.fbody[(size_t)g * F.fcap + k] = s_key[e] - 1u;
    }
  }
  __syncthreads();
  if (t == 0) {
    F.fcnt[(size_t)g * kF6CntStride] = s_cnt;
    atomicMax(F.max_foreign, s_cnt);
    if (s_cnt > F.fcap || s_cnt >= kF6Hash / 2u) atomicOr(F.fail, 1u);
  }
  for (uint32_t sl = t; sl < N; sl += kBlock) {
    const uint32_t c = list[sl];
    const uint2 e = K.ab[c];
    const uint32_t aref = F.brank[e.x] - g * F.nb;
    uint32_t bref = kF6NoBody;
    if (e.y != kNone) {
      const uint32_t pb = F.brank[e.y];
      if (pb / F.nb == g) bref = pb - g * F.nb;
      else {
        uint32_t i = (e.y * 2654435761u) >> 20;
        while (s_key[i] != e.y + 1u) i = (i + 1u) & (kF6Hash - 1u);
        bref = F.nb + s_val[i];
      }
      F.bref[c] = bref;
    }
    if (aref >= kF6NoBody || (e.y != kNone && bref >= kF6NoBody)) atomicOr(F.fail, 16u);
    uint4* dst = reinterpret_cast<uint4*>(&F.table[(size_t)g * F.rows + sl]);
    dst[0] = make_uint4(c, aref | (min(bref, kF6NoBody) << kF6BodyBits), 0u, 0u);  // (the successor words: k_flow6g_links)
  }
}
__global__ __launch_bounds__(kBlock) void k_flow6g_links(Flow6 F, ConsLinks K, const uint32_t* cslot) {
  const uint32_t c = blockIdx.x * kBlock + threadIdx.x;
  if (c >= *F.C_ptr) return;
  const uint2 e = K.ab[c];
  const uint32_t g = F.brank[e.x] / F.nb, sl = cslot[c];
  if (sl >= F.rows || sl >= kF6MaxSlots) return;  // (fail bit 2 is up)
  const uint2 sw = K.succ[c];
  uint32_t out[2] = {0u, 0u};
#pragma unroll
  for (int side = 0; side < 2; ++side) {
    if (side == 1 && e.y == kNone) break;
    const uint32_t w = side == 0 ? sw.x : sw.y, body = side == 0 ? e.x : e.y;
    const uint32_t wid = w & kSuccId, wrap = (w & kSuccWrap) ? kF6Wrap : 0u;
    const uint2 we = K.ab[wid];
    const uint32_t hw = F.brank[we.x] / F.nb, ws = cslot[wid];
    if (hw == g) { out[side] = wrap | ws; }
    else {  // a message on the channel g -> hw, addressed to the body's LDS slot over there
      const uint32_t dbody = we.x == body ? F.brank[body] - hw * F.nb : F.bref[wid];
      uint32_t* ik = F.in_key + (size_t)hw * kF6Chan;
      uint32_t* ok = F.out_key + (size_t)g * kF6Chan;
      const uint32_t k_in = f6_chan_slot(ik, g + 1u, F.fail), k_out = f6_chan_slot(ok, hw + 1u, F.fail);
      atomicAdd(&F.in_cnt[(size_t)hw * kF6Chan + k_in], 1u);
      F.out_val[(size_t)g * kF6Chan + k_out] = k_in;
      if (ws >= kF6MaxSlots || dbody >= kF6NoBody) atomicOr(F.fail, 16u);
      out[side] = kF6Remote | wrap | (k_out << 24) | (dbody << kF6SlotBits) | ws;
    }
    // the body's chain ends here, in a block that is not its own: this block writes its result, its own does not
    if (side == 1 && wrap && F.brank[body] / F.nb != g) F.skipwb[body] = 1;
  }
  uint32_t* row = reinterpret_cast<uint32_t*>(&F.table[(size_t)g * F.rows + sl]);
  row[2] = out[0]; row[3] = out[1];
  row[4] = K.pred[2 * c]; row[5] = e.y != kNone ? K.pred[2 * c + 1] : 0u; row[6] = 0u; row[7] = 0u;
}

// ---- the solve ------------------------------------------------------------------------------------------------------------
struct F6Ring { uint16_t* ring; uint32_t* head; uint32_t* tail; uint32_t cap, magic; };
__device__ __forceinline__ uint32_t f6_wrap(const F6Ring& q, uint32_t pos) {  // pos % cap for pos < cap * kF6MaxIters
  return pos - __umulhi(pos, q.magic) * q.cap;
}
__device__ __forceinline__ void f6_push(const F6Ring& q, uint32_t slot) {
  const uint32_t pos = __hip_atomic_fetch_add(q.tail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  q.ring[f6_wrap(q, pos)] = (uint16_t)(slot | 0x8000u);
}
// one arrival at `slot`: the last one queues it
__device__ __forceinline__ void f6_arrive(const F6Ring& q, uint32_t* s_state, uint32_t slot) {
  const uint32_t old = __hip_atomic_fetch_sub(&s_state[slot], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  if ((old & kF6StArrMask) == 1u) f6_push(q, slot);
}

// CL: the constant half of the own bodies' solver records (inverse mass, world inverse inertia: 40 B) is kept in LDS as well -
// chosen by the host when the block's constraints leave room for it (the first ~100 ticks of the bench pile); otherwise the
// lanes read it from the RigidBodyVec beside the constraint record.
// NL: ContactState::normal_impulse of every slot's constraint lives in LDS for the launch as well (4 bytes per slot, when there is room).
// RL: the solver half of EVERY constraint record of the block (normal, tangents, arms, bias, effective masses: 76 bytes) and its
// accumulated impulse live in LDS for the launch as well, read once in the prologue by coalesced loads - chosen by the host when
// the block's slots leave room for 80 more bytes each (worlds of few constraints per block: BASELINE configs 3 and 5, tiles).  A
// node then touches no global memory at all (with CL; messages apart): its service is LDS reads + arithmetic.  Implies NL's effect.
#ifndef MGF_F6_IDLE_SLEEP
#define MGF_F6_IDLE_SLEEP 2  // s_sleep argument of a serving wave that found the ready queue empty (0, 1, 2: within 1.5 % of each other, r04)
#endif
#ifdef MGF_F6_PROFILE  // (the profile build measures the trips themselves: no per-node clock reads, whose latency would be most of a trip)
#define F6_NODE_CLOCK() 0ull
#else
#define F6_NODE_CLOCK() wall_clock64()
#endif
template <bool TRACE, int CL, bool NL, bool RL, bool QD>
__global__ __launch_bounds__(kF6Threads) void k_solve_flow6(float4* srec, CRec* cons, Flow6 F, uint32_t iters, uint32_t epoch, uint32_t* abort_flag,
                                                            uint32_t spin_limit, uint64_t* trace, uint32_t C_trace) {
  if (*F.fail || *F.C_ptr == 0u) return;  // a limit was exceeded: the stand-by k_solve_flow launch behind this one does the work
  // tests (option flow_spin_limit = 1): every other workgroup behaves as one that never became resident - the others wait for its messages,
  // give up at their first look at the limit, and the world is left half-solved for solver_abort_fallback to put right
  if (spin_limit == 1u && (blockIdx.x & 1u)) { if (threadIdx.x == 0) __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
  extern __shared__ float4 s_dyn[];
  uint64_t* tstat = TRACE ? trace + 2 * (size_t)iters * C_trace + kF6TraceWords * (size_t)blockIdx.x : nullptr;
  uint64_t trace_c0 = 0;
  if (TRACE && threadIdx.x == 0) { tstat[8] = wall_clock64(); trace_c0 = clock64(); }
  const uint32_t nbod = F.nb + F.fcap, cap = F.slot_cap;
  constexpr bool NLS = NL && !RL;                                      // the impulses in an array of their own
  float4* s_body = s_dyn;                                              // [2 * nbod]: {v, w.x}, {w.y, w.z, body id, -}
  float4* s_rec = s_dyn + 2 * (size_t)nbod;                            // RL: [kF6RecWords * cap] CRec words 4..19, then {n.x, n.y, tmass1, nimp}
  float2* s_const = reinterpret_cast<float2*>(s_rec + (RL ? kF6RecWords * (size_t)cap : 0u));  // CL: [5 * nb] inverse mass and inertia of the own bodies
  uint2* s_succ = reinterpret_cast<uint2*>(s_const + (CL == 2 ? 5 * (size_t)nbod : CL == 1 ? 5 * (size_t)F.nb : 0));  // [cap]
  uint32_t* s_c = reinterpret_cast<uint32_t*>(s_succ + cap);           // [cap]
  uint32_t* s_state = s_c + cap;                                       // [cap] arrivals missing | iterations done | body references (kF6St*)
  float* s_nimp = reinterpret_cast<float*>(s_state + cap);            // [cap] ContactState::normal_impulse of the slot's constraint: read and written
                                                                       // once per solve - in LDS (NL), not in the record (a 4-byte store per solve
                                                                       // costs the launch 8 %: it queues in front of the next records' loads)
  uint32_t* s_ctl = reinterpret_cast<uint32_t*>(s_nimp + (NLS ? cap : 0u));  // [16]: 0 head, 1 tail, 2 nodes left, 3 incoming channels
  uint32_t* s_out_base = s_ctl + 16;                                   // [kF6Chan] first message of the outgoing channel
  uint32_t* s_out_tail = s_out_base + kF6Chan;                         // [kF6Chan] messages sent
  uint32_t* s_out_tidx = s_out_tail + kF6Chan;                         // [kF6Chan] the channel's word of F.tails
  uint32_t* s_in_base = s_out_tidx + kF6Chan;                          // [kF6Chan] incoming channels, compacted
  uint32_t* s_in_lim = s_in_base + kF6Chan;                            // [kF6Chan] messages the channel can carry in this launch
  uint32_t* s_in_head = s_in_lim + kF6Chan;                            // [kF6Chan] first position not yet consumed
  uint32_t* s_in_mask = s_in_head + kF6Chan;                           // [kF6Chan] consumed positions of the window behind the head
  uint32_t* s_in_slot = s_in_mask + kF6Chan;                           // [kF6Chan] the channel's hash slot (its word of F.tails)
  F6Ring q;
  q.head = s_ctl; q.tail = s_ctl + 1; q.cap = cap; q.magic = 0xFFFFFFFFu / cap + 1u;
  uint32_t* s_wl_cnt = s_in_slot + kF6Chan;                            // [16] per polling wave: items listed in this sweep
  uint16_t* s_wl = reinterpret_cast<uint16_t*>(s_wl_cnt + 16);          // [kF6MaxPollers * kF6WlLen] (channel << 8) | position behind its head
  q.ring = s_wl + kF6MaxPollers * kF6WlLen;                            // [cap]
  uint32_t* s_left = s_ctl + 2;
  const uint32_t g = blockIdx.x, t = threadIdx.x;
  const uint32_t p_lo = g * F.nb, p_hi = min(F.n, p_lo + F.nb), n_own = p_hi - p_lo;
  const uint32_t n_for = min(F.fcnt[(size_t)g * kF6CntStride], F.fcap);
  const uint32_t N = F.nslots[(size_t)g * kF6CntStride];
  __amdgpu_buffer_rsrc_t rmb = make_rsrc(F.mbox);
  // every body this block touches: its own, then the foreign ones
  for (uint32_t i = t; i < n_own + n_for; i += kF6Threads) {
    const uint32_t x = i < n_own ? (F.ident ? p_lo + i : F.sidx[p_lo + i]) : F.fbody[(size_t)g * F.fcap + (i - n_own)];
    const uint32_t idx = i < n_own ? i : F.nb + (i - n_own);
    const float4 r0 = srec[4 * (size_t)x], r1 = srec[4 * (size_t)x + 1];
    s_body[2 * idx] = r0;
    s_body[2 * idx + 1] = make_float4(r1.x, r1.y, u2f(x), 0.0f);
    if (CL == 2 || (CL == 1 && i < n_own)) {
      const float4 r2 = srec[4 * (size_t)x + 2], r3 = srec[4 * (size_t)x + 3];
      s_const[5 * idx] = make_float2(r1.z, r1.w); s_const[5 * idx + 1] = make_float2(r2.x, r2.y); s_const[5 * idx + 2] = make_float2(r2.z, r2.w);
      s_const[5 * idx + 3] = make_float2(r3.x, r3.y); s_const[5 * idx + 4] = make_float2(r3.z, r3.w);
    }
  }
  for (uint32_t e = t; e < (cap + 1u) / 2u; e += kF6Threads) reinterpret_cast<uint32_t*>(q.ring)[e] = 0u;
  if (t < 16) s_ctl[t] = t == 2 ? N * iters : 0u;
  const uint32_t region = F.mbox_cap / F.nblocks;  // messages of the channel buffer each block owns
  if (t < 64) {  // wave 0: channel tables (the incoming ones compacted: the polling wave gives every channel a few lanes)
    uint32_t okey = 0, ikey = 0;
    if (t < kF6Chan) { okey = F.out_key[(size_t)g * kF6Chan + t]; ikey = F.in_key[(size_t)g * kF6Chan + t]; }
    if (t < kF6Chan) {
      s_out_tail[t] = 0u;
      const uint32_t tix = okey ? (okey - 1u) * kF6Chan + F.out_val[(size_t)g * kF6Chan + t] : 0u;
      s_out_base[t] = okey ? (okey - 1u) * region + F.chan_prefix[tix] * iters : 0u;
      s_out_tidx[t] = tix;
      s_in_head[t] = 0u; s_in_mask[t] = 0u; s_in_lim[t] = 0u; s_in_base[t] = 0u;
    }
    const unsigned long long m = __ballot(ikey != 0u);
    if (ikey) {
      const uint32_t r = (uint32_t)__popcll(m & ((1ull << t) - 1ull));
      s_in_base[r] = g * region + F.chan_prefix[(size_t)g * kF6Chan + t] * iters;
      s_in_lim[r] = F.in_cnt[(size_t)g * kF6Chan + t] * iters;
      s_in_slot[r] = t;
    }
    if (t == 0) s_ctl[3] = (uint32_t)__popcll(m);
  }
  __syncthreads();
  if (TRACE && threadIdx.x == 0) tstat[9] = wall_clock64();
  // the block's slot table (built once per tick by k_flow6_blocks and k_flow6_links)
  const F6Row* rows = F.table + (size_t)g * F.rows;
  for (uint32_t idx = t; idx < N; idx += kF6Threads) {
    const uint4* src = reinterpret_cast<const uint4*>(&rows[idx]);
    const uint4 r0 = src[0];
    const uint4 r1 = src[1];
    const uint32_t st0 = r1.x + r1.y;
    s_c[idx] = r0.x; s_succ[idx] = make_uint2(r0.z, r0.w); s_state[idx] = st0 | (r0.y << kF6StRefShift);
    if (RL) {
      const float4* g = reinterpret_cast<const float4*>(&cons[r0.x]);
      const float4 g0 = g[0], g1 = g[1], g2 = g[2], g3 = g[3], g4 = g[4], g5 = g[5];
      float4* d = s_rec + kF6RecWords * (size_t)idx;
      d[0] = g1; d[1] = g2; d[2] = g3; d[3] = g4; d[4] = make_float4(g0.z, g0.w, g5.x, g5.z);
    }
    if (NLS) s_nimp[idx] = cons[r0.x].nimp;  // (0 in a tick's first Solver::solve; what the last one left in a later one)
    if (st0 == 0u && iters > 0) f6_push(q, idx);  // iteration 0's frontier
  }
  __syncthreads();
  if (TRACE && threadIdx.x == 0) tstat[10] = wall_clock64();
  const uint32_t wave = t >> 6, lane = t & 63u, nwaves = kF6Threads / 64u;
  const uint32_t n_in = s_ctl[3];
  // the last F.poll_waves waves poll the incoming channels (channel r belongs to poller r % P), the others serve the queue
  const uint32_t P = n_in == 0u ? 0u : min(min(F.poll_waves, kF6MaxPollers), n_in);
  const bool poller = wave + P >= nwaves;
  uint32_t spins = 0;
  if (poller) {
    // Polling wave pw serves the incoming channels r = pw, pw + P, ...; lane i OWNS channel i of them: its head (first position
    // not yet consumed), the consumed bits of the 32 positions behind it and the producers' HINT (positions handed out so far,
    // tagged with the launch) live in that lane's registers.  Every sweep the owners list the positions worth reading - what
    // the last hint says was sent and is not consumed, and the head itself on spec (a lone message then costs one memory
    // round trip) - in an LDS worklist; the wave's 64 lanes take one item each, read the message, and the owners read the
    // hint anew: four load instructions per sweep whatever the traffic, nothing read that was not sent (apart from one slot
    // per channel), and a burst on one channel is drained as fast as a trickle on all of them.
    // A message counts when its three granules carry this launch's tag (granules land whole; their order is not defined).
    const uint32_t pw = wave - (nwaves - P);
    const uint32_t n_loc = (n_in - pw + P - 1u) / P;
    const bool owner = lane < n_loc;
    const uint32_t ch = lane * P + pw;
    const uint32_t lim = owner ? s_in_lim[ch] : 0u, in_base = owner ? s_in_base[ch] : 0u;
    const unsigned long long* tail_ptr = F.tails + (size_t)g * kF6Chan + (owner ? s_in_slot[ch] : 0u);
    uint32_t* wl_cnt = s_wl_cnt + pw;
    uint16_t* wl = s_wl + pw * kF6WlLen;
    if (F.poll_prio == 1u) __builtin_amdgcn_s_setprio(1); else if (F.poll_prio == 2u) __builtin_amdgcn_s_setprio(2); else if (F.poll_prio >= 3u) __builtin_amdgcn_s_setprio(3);
    uint32_t head = 0, mask = 0, known = 0;
    // the quiet sweep's lanes: position my_j behind the head of owner my_o's channel
    const uint32_t kq = max(1u, min(min(F.poll_spec, 8u), 64u / max(n_loc, 1u)));
    const uint32_t my_o = lane % max(n_loc, 1u), my_j = lane / max(n_loc, 1u);
    const bool my_act = my_j < kq;
    const uint32_t lim_o = (uint32_t)__shfl((int)lim, (int)my_o), base_o = (uint32_t)__shfl((int)in_base, (int)my_o);
    uint32_t st_sweeps = 0, st_hits = 0, st_lat_sum = 0, st_lat_max = 0, st_full = 0, st_wait = 0;  // TRACE: polling statistics
    uint32_t st_h0 = 0, st_h1 = 0, st_h2 = 0, st_h3 = 0, st_h4 = 0, st_h5 = 0, st_quiet = 0;  // messages by latency (< 1, 2, 3, 4, 6 us, more); seen by a quiet sweep
#define F6_LAT_HIST(lat) { if ((lat) < 100u) ++st_h0; else if ((lat) < 200u) ++st_h1; else if ((lat) < 300u) ++st_h2; else if ((lat) < 400u) ++st_h3; else if ((lat) < 600u) ++st_h4; else ++st_h5; }
    if (lane == 0) __hip_atomic_store(wl_cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    for (;;) {
      if (__hip_atomic_load(s_left, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0u) break;
      uint64_t tq0 = 0;
      if (TRACE) { ++st_sweeps; tq0 = wall_clock64(); }
      // what every owner wants to read in this sweep
      uint32_t todo = 0;
      if (owner) {
        const uint32_t pend = known > head ? min(known - head, 32u) : 0u;
        todo = ~mask & (pend >= 32u ? 0xFFFFFFFFu : (1u << pend) - 1u);
        if (F.poll_spec_wl == 0u) { if (head < lim) todo |= 1u & ~mask; }  // the head, on spec
        else if (head + pend < lim && pend < 32u) {  // the positions behind what the producers' hint covers - where the next messages land
          const uint32_t ns = min(F.poll_spec_wl, min(32u - pend, lim - head - pend));
          todo |= (((1u << ns) - 1u) << pend) & ~mask;
        }
      }
      unsigned long long hits = 0;
      if (F.poll_k >= 2u && __ballot((todo >> kq) != 0u) == 0ull) {
        // the quiet sweep (no channel is known to hold more than the `kq` positions behind its head - the state a message on a critical
        // path finds): lane l reads position head + l / n_loc of channel l % n_loc straight, no worklist, no shared words.  Positions
        // behind the head are read ON SPEC (nothing says they were sent): a message that is not the first of its channel is seen by
        // the sweep it arrives in instead of the one after the producers' hint was read - the load instructions are the same three.
        const uint32_t h_o = (uint32_t)__shfl((int)head, (int)my_o), m_o = (uint32_t)__shfl((int)mask, (int)my_o);
        const uint32_t pos = h_o + my_j;
        const bool have = my_act && pos < lim_o && ((m_o >> my_j) & 1u) == 0u;
        const uint32_t byte = have ? (base_o + pos) * (16u * kF6MsgWords) : 0x80000000u;
        const v4f_t g0 = __builtin_amdgcn_raw_buffer_load_b128(rmb, (int)byte, 0, kSc1);
        const v4f_t g1 = __builtin_amdgcn_raw_buffer_load_b128(rmb, (int)(byte + 16u), 0, kSc1);
        const v4f_t g2 = __builtin_amdgcn_raw_buffer_load_b128(rmb, (int)(byte + 32u), 0, kSc1);
        if (owner) {
          const unsigned long long tv = __hip_atomic_load(tail_ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          known = (uint32_t)(tv >> 32) == epoch ? min((uint32_t)tv, lim) : 0u;
        }
        if (TRACE) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); st_wait += (uint32_t)(wall_clock64() - tq0); }
        const bool hit = have && f2u(g0.w) == epoch && f2u(g1.w) == epoch && f2u(g2.w) == epoch;
        if (hit) {
          const uint32_t addr = f2u(g2.x);
          const uint32_t slot = addr & ((1u << kF6SlotBits) - 1u), bi = addr >> kF6SlotBits;
          s_body[2 * bi] = make_float4(g0.x, g0.y, g0.z, g1.x);
          *reinterpret_cast<float2*>(&s_body[2 * bi + 1]) = make_float2(g1.y, g1.z);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the velocity is in LDS before the arrival counts
          f6_arrive(q, s_state, slot);
          if (TRACE) { const uint32_t lat = (uint32_t)wall_clock64() - f2u(g2.y); ++st_hits; st_lat_sum += lat; st_lat_max = max(st_lat_max, lat); F6_LAT_HIST(lat); ++st_quiet; }
        }
        hits = __ballot(hit);
        if (owner && hits) {  // lane j * n_loc + o read position head + j of owner o's channel
          for (uint32_t j = 0; j < kq; ++j) mask |= (uint32_t)((hits >> (j * n_loc + lane)) & 1ull) << j;
          const uint32_t k = mask == 0xFFFFFFFFu ? 32u : (uint32_t)__builtin_ctz(~mask);
          head += k; mask = k >= 32u ? 0u : mask >> k;
        }
      } else {
      // the owners list their positions
      if (owner) {
        const uint32_t need = (uint32_t)__popc(todo);
        if (need) {
          uint32_t at = __hip_atomic_fetch_add(wl_cnt, need, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          __hip_atomic_store(&s_in_head[ch], head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          while (todo) {
            const uint32_t bit = (uint32_t)__builtin_ctz(todo);
            todo &= todo - 1u;
            if (at < kF6WlLen) wl[at] = (uint16_t)((lane << 8) | bit);
            ++at;
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const uint32_t total = min(__hip_atomic_load(wl_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP), kF6WlLen);
      if (TRACE && total >= 64u) ++st_full;
      for (uint32_t b = 0; b < total || b == 0u; b += 64u) {
        const uint32_t j = b + lane;
        const bool have = j < total;
        const uint32_t item = have ? wl[j] : 0u;
        const uint32_t oi = item >> 8, bit = item & 0xFFu, chn = oi * P + pw;
        const uint32_t pos = have ? __hip_atomic_load(&s_in_head[chn], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) + bit : 0u;
        const uint32_t byte = have ? (s_in_base[chn] + pos) * (16u * kF6MsgWords) : 0x80000000u;  // (out of range: zeros, no traffic)
        const v4f_t g0 = __builtin_amdgcn_raw_buffer_load_b128(rmb, (int)byte, 0, kSc1);
        const v4f_t g1 = __builtin_amdgcn_raw_buffer_load_b128(rmb, (int)(byte + 16u), 0, kSc1);
        const v4f_t g2 = __builtin_amdgcn_raw_buffer_load_b128(rmb, (int)(byte + 32u), 0, kSc1);
        if (b == 0u && owner) {
          const unsigned long long tv = __hip_atomic_load(tail_ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          known = (uint32_t)(tv >> 32) == epoch ? min((uint32_t)tv, lim) : 0u;
        }
        if (TRACE && b == 0u) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); st_wait += (uint32_t)(wall_clock64() - tq0); }
        const bool hit = have && f2u(g0.w) == epoch && f2u(g1.w) == epoch && f2u(g2.w) == epoch;
        if (hit) {
          const uint32_t addr = f2u(g2.x);
          const uint32_t slot = addr & ((1u << kF6SlotBits) - 1u), bi = addr >> kF6SlotBits;
          s_body[2 * bi] = make_float4(g0.x, g0.y, g0.z, g1.x);
          *reinterpret_cast<float2*>(&s_body[2 * bi + 1]) = make_float2(g1.y, g1.z);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the velocity is in LDS before the arrival counts
          f6_arrive(q, s_state, slot);
          __hip_atomic_fetch_or(&s_in_mask[chn], 1u << bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          if (TRACE) { const uint32_t lat = (uint32_t)wall_clock64() - f2u(g2.y); ++st_hits; st_lat_sum += lat; st_lat_max = max(st_lat_max, lat); F6_LAT_HIST(lat); }
        }
        hits |= __ballot(hit);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (lane == 0) __hip_atomic_store(wl_cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (owner && hits) {  // what was consumed of this lane's channel (these words are this wave's alone: program order holds)
        mask |= __hip_atomic_exchange(&s_in_mask[ch], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const uint32_t k = mask == 0xFFFFFFFFu ? 32u : (uint32_t)__builtin_ctz(~mask);
        head += k; mask = k >= 32u ? 0u : mask >> k;
      }
      }
      if (hits) { spins = 0; continue; }
      __builtin_amdgcn_s_sleep(1);
      if ((++spins & 1023u) == 0u) {
        bool give_up = spins > spin_limit;
        if (give_up) __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (give_up || __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
      }
    }
    if (TRACE) {  // per block: sweeps of the first polling lane, messages, latency sum / max (clock ticks), full batches
      uint64_t* st = trace + 2 * (size_t)iters * C_trace + kF6TraceWords * (size_t)g;
      if (lane == 0 && pw == 0) st[0] = st_sweeps;
      atomicAdd(reinterpret_cast<unsigned long long*>(&st[1]), (unsigned long long)st_hits);
      atomicAdd(reinterpret_cast<unsigned long long*>(&st[2]), (unsigned long long)st_lat_sum);
      atomicMax(reinterpret_cast<unsigned long long*>(&st[3]), (unsigned long long)st_lat_max);
      if (lane == 0) atomicAdd(reinterpret_cast<unsigned long long*>(&st[4]), (unsigned long long)st_full);
      if (lane == 0 && pw == 0) { st[5] = n_in; st[6] = st_wait; }
      atomicAdd(reinterpret_cast<unsigned long long*>(&st[24]), (unsigned long long)st_h0); atomicAdd(reinterpret_cast<unsigned long long*>(&st[25]), (unsigned long long)st_h1);
      atomicAdd(reinterpret_cast<unsigned long long*>(&st[26]), (unsigned long long)st_h2); atomicAdd(reinterpret_cast<unsigned long long*>(&st[27]), (unsigned long long)st_h3);
      atomicAdd(reinterpret_cast<unsigned long long*>(&st[28]), (unsigned long long)st_h4); atomicAdd(reinterpret_cast<unsigned long long*>(&st[29]), (unsigned long long)st_h5);
      atomicAdd(reinterpret_cast<unsigned long long*>(&st[30]), (unsigned long long)st_quiet);
    }
  } else {
    // ---- the serving waves ---------------------------------------------------------------------------------------------------
    // Written for few LDS round trips (r04: a trip's time is its dependent LDS accesses - ~150 clocks each - plus one lane's chain of
    // arithmetic, tools/r04_trip_profile.py): the queue's three words are broadcast reads, both releases of a node are in flight together.
    constexpr uint32_t kSlotMask = (1u << kF6SlotBits) - 1u;
    uint32_t* s_dummy = s_ctl + 8;  // [2] words nobody reads: the target of a release that has no local successor on a side
#ifdef MGF_F6_PROFILE  // (variant build: where a trip's time goes, shader clocks summed over wave 0's trips -> the block's trace words 13..21; QD = 0)
    uint64_t pf_t0 = 0, pf_t1 = 0, pf_t2 = 0, pf_t2a = 0, pf_t3 = 0, pf_acc6 = 0, pf_acc[5] = {0, 0, 0, 0, 0}, pf_trips = 0, pf_nodes = 0, pf_idle_t = 0;
#define PF_STAMP(x) do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); x = clock64(); } while (0)
#define PF_STAMP_V(x) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); x = clock64(); } while (0)
#else
#define PF_STAMP(x) do { } while (0)
#define PF_STAMP_V(x) do { } while (0)
#endif
    // One trip with ONE LANE PER NODE: up to 64 ready nodes taken from the queue (head h0, tail tl as just read), run, released.
    // Returns false when another wave took them first.
    auto scalar_trip = [&](uint32_t h0, uint32_t tl) -> bool {
      uint32_t take = min(tl - h0, 64u);
      if (take) {
        uint32_t got = 0u;
        if (lane == 0) {
          uint32_t expect = h0;
          if (__hip_atomic_compare_exchange_strong(q.head, &expect, h0 + take, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) got = take;
        }
        take = __builtin_amdgcn_readfirstlane(got);
      }
      PF_STAMP(pf_t1);
      if (!take) return false;
      const bool act = lane < take;
      if (act) {
        uint16_t* cell = &q.ring[f6_wrap(q, h0 + lane)];
        uint32_t e;
        do { e = __hip_atomic_load(cell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); } while (!(e & 0x8000u));  // the pusher is between its two writes
        *cell = 0;
        const uint32_t my_slot = e & 0x7FFFu;
        const uint32_t my_stw = __hip_atomic_load(&s_state[my_slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        {
          const uint32_t slot = my_slot, stw = my_stw;
          const uint32_t round = (stw >> kF6StIterShift) & kF6StIterMask;
          uint64_t t_seen = 0;
          if (TRACE) t_seen = F6_NODE_CLOCK();
          const uint32_t c = s_c[slot], ref = stw >> kF6StRefShift;
          const uint2 sw = s_succ[slot];
          const uint32_t ai = ref & kF6NoBody, bi_raw = (ref >> kF6BodyBits) & kF6NoBody;
          const bool has_b = bi_raw != kF6NoBody;
          const uint32_t bi = has_b ? bi_raw : ai;
          CRec rec;
          float4 q0, q1, q2, q3, q4;
          if (RL) { const float4* d = s_rec + kF6RecWords * (size_t)slot; q0 = d[0]; q1 = d[1]; q2 = d[2]; q3 = d[3]; q4 = d[4]; }
          else rec = load_crec_solve(&cons[c]);  // only the lane running the constraint touches its record
          if (NLS) rec.nimp = s_nimp[slot];
          const float4 a0 = s_body[2 * ai], a1 = s_body[2 * ai + 1], b0 = s_body[2 * bi], b1 = s_body[2 * bi + 1];
          const uint32_t ga = f2u(a1.z), gb = f2u(b1.z);
          // the constant half of ConstrainedSet::get (inverse mass, world inverse inertia): from LDS for own bodies (CL >= 1) and
          // foreign ones (CL == 2), else plain loads beside the record's
          float4 ca1, ca2, ca3, cb1, cb2, cb3;
          if (CL) {
            const float2 k0 = s_const[5 * ai], k1 = s_const[5 * ai + 1], k2 = s_const[5 * ai + 2], k3 = s_const[5 * ai + 3], k4 = s_const[5 * ai + 4];
            ca1 = make_float4(0, 0, k0.x, k0.y); ca2 = make_float4(k1.x, k1.y, k2.x, k2.y); ca3 = make_float4(k3.x, k3.y, k4.x, k4.y);
            if (CL == 2 || bi < F.nb) {
              const float2 j0 = s_const[5 * bi], j1 = s_const[5 * bi + 1], j2 = s_const[5 * bi + 2], j3 = s_const[5 * bi + 3], j4 = s_const[5 * bi + 4];
              cb1 = make_float4(0, 0, j0.x, j0.y); cb2 = make_float4(j1.x, j1.y, j2.x, j2.y); cb3 = make_float4(j3.x, j3.y, j4.x, j4.y);
            } else {
              cb1 = srec[4 * (size_t)gb + 1]; cb2 = srec[4 * (size_t)gb + 2]; cb3 = srec[4 * (size_t)gb + 3];
            }
          } else {
            ca1 = srec[4 * (size_t)ga + 1]; ca2 = srec[4 * (size_t)ga + 2]; ca3 = srec[4 * (size_t)ga + 3];
            cb1 = srec[4 * (size_t)gb + 1]; cb2 = srec[4 * (size_t)gb + 2]; cb3 = srec[4 * (size_t)gb + 3];
          }
          BodyDyn A, Bd;
          A.v = mk3(a0.x, a0.y, a0.z); A.w = mk3(a0.w, a1.x, a1.y); A.im = ca1.z;
          A.I = m3_cols(mk3(ca1.w, ca2.x, ca2.y), mk3(ca2.z, ca2.w, ca3.x), mk3(ca3.y, ca3.z, ca3.w));
          Bd.v = mk3(b0.x, b0.y, b0.z); Bd.w = mk3(b0.w, b1.x, b1.y); Bd.im = cb1.z;
          Bd.I = m3_cols(mk3(cb1.w, cb2.x, cb2.y), mk3(cb2.z, cb2.w, cb3.x), mk3(cb3.y, cb3.z, cb3.w));
          if (!has_b) Bd = static_dyn();
          PF_STAMP(pf_t2a);
          PF_STAMP_V(pf_t2);
          if (RL) {
            float nimp = q4.w;
            solve_core(mk3(q4.x, q4.y, q0.x), mk3(q0.y, q0.z, q0.w), mk3(q1.x, q1.y, q1.z), mk3(q1.w, q2.x, q2.y), mk3(q2.z, q2.w, q3.x), q3.y, q3.z, q3.w, q4.z,
                       nimp, A, Bd);
            s_rec[kF6RecWords * (size_t)slot + 4].w = nimp;
          } else solve_one(rec, A, Bd);
          s_body[2 * ai] = make_float4(A.v.x, A.v.y, A.v.z, A.w.x);
          *reinterpret_cast<float2*>(&s_body[2 * ai + 1]) = make_float2(A.w.y, A.w.z);
          if (has_b) {
            s_body[2 * bi] = make_float4(Bd.v.x, Bd.v.y, Bd.v.z, Bd.w.x);
            *reinterpret_cast<float2*>(&s_body[2 * bi + 1]) = make_float2(Bd.w.y, Bd.w.z);
          }
          if (NLS) s_nimp[slot] = rec.nimp; else if (!RL) cons[c].nimp = rec.nimp;
          // re-arm: one arrival per dynamic body and iteration from now on (no arrival of the next iteration can come before
          // this node's own releases), and one more iteration done
          __hip_atomic_fetch_add(&s_state[slot], (1u << kF6StIterShift) + (has_b ? 2u : 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          if (round + 1u == iters) {  // the end of a foreign body's chain: its home block does not write it back
            if (has_b && (sw.y & kF6Wrap) && bi >= F.nb) store_vel(srec, gb, Bd);  // (a is always the block's own)
          }
#ifdef MGF_F6_STORE_WAIT  // (r04: not needed - a wave's DS instructions execute in issue order, so the release's atomic cannot overtake the stores; 1-2.5 % faster without)
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // velocities are in LDS before any local successor hears of it
#else
          asm volatile("" ::: "memory");
#endif
          PF_STAMP(pf_t3);
#ifndef MGF_F6_PROFILE
          if (TRACE) {
            trace[2 * ((size_t)round * C_trace + c)] = t_seen & ~3ull;
            trace[2 * ((size_t)round * C_trace + c) + 1] = F6_NODE_CLOCK();
          }
#endif
          // The release: first the messages of the sides whose successor lives in another block, then the arrivals at local successors -
          // two LDS atomics in flight together (a side without a local successor decrements a dummy word: no branch between them) -
          // and whatever became ready is queued.
          const uint32_t w0 = sw.x, w1 = sw.y;
          const bool live0 = round + ((w0 & kF6Wrap) ? 1u : 0u) < iters, live1 = has_b && round + ((w1 & kF6Wrap) ? 1u : 0u) < iters;
          const bool loc0 = live0 && !(w0 & kF6Remote), loc1 = live1 && !(w1 & kF6Remote);
          const uint32_t ws0 = w0 & kSlotMask, ws1 = w1 & kSlotMask;
#pragma unroll
          for (int side = 0; side < 2; ++side) {
            const uint32_t w = side == 0 ? w0 : w1;
            if (!((side == 0 ? live0 : live1) && (w & kF6Remote))) continue;
            // a message: the body's velocity, where it goes, the launch's tag in every granule - and on we go
            const uint32_t chn = (w >> 24) & (kF6Chan - 1u);  // (6 bits)
            const uint32_t pos = __hip_atomic_fetch_add(&s_out_tail[chn], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const uint32_t byte = (s_out_base[chn] + pos) * (16u * kF6MsgWords);
            const BodyDyn& X = side == 0 ? A : Bd;
            const float tg = u2f(epoch);
            v4f_t m0 = {X.v.x, X.v.y, X.v.z, tg}, m1 = {X.w.x, X.w.y, X.w.z, tg}, m2 = {u2f(w & 0x00FFFFFFu), TRACE ? u2f((uint32_t)F6_NODE_CLOCK()) : 0.0f, 0.0f, tg};
            __builtin_amdgcn_raw_buffer_store_b128(m0, rmb, (int)byte, 0, kSc1);
            __builtin_amdgcn_raw_buffer_store_b128(m1, rmb, (int)(byte + 16u), 0, kSc1);
            __builtin_amdgcn_raw_buffer_store_b128(m2, rmb, (int)(byte + 32u), 0, kSc1);
            __hip_atomic_fetch_max(F.tails + s_out_tidx[chn], ((unsigned long long)epoch << 32) | (pos + 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          {  // (the messages first, then the local releases: the slow edge leaves first)
            const uint32_t was0 = __hip_atomic_fetch_sub(loc0 ? &s_state[ws0] : &s_dummy[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const uint32_t was1 = __hip_atomic_fetch_sub(loc1 ? &s_state[ws1] : &s_dummy[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const bool rdy0 = loc0 && (was0 & kF6StArrMask) == 1u, rdy1 = loc1 && (was1 & kF6StArrMask) == 1u;
            if (rdy0) f6_push(q, ws0);
            if (rdy1) f6_push(q, ws1);
          }
        }
      }
      if (lane == 0) __hip_atomic_fetch_sub(s_left, take, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#ifdef MGF_F6_PROFILE
      {
        uint64_t pf_t4; PF_STAMP(pf_t4);
        const uint64_t a2 = __shfl(pf_t2, 0), a3 = __shfl(pf_t3, 0), a2a = __shfl(pf_t2a, 0);
        pf_acc6 += a2a - pf_t1;
        pf_acc[0] += pf_t1 - pf_t0; pf_acc[1] += a2 - pf_t1; pf_acc[2] += a3 - a2; pf_acc[3] += pf_t4 - a3; ++pf_trips; pf_nodes += (uint64_t)take;
      }
#endif
      return true;
    };
    // QD: FOUR LANES PER NODE.  Lane c of a quad (lane 3 mirrors lane 2) holds component c of every vector of the solve and runs
    // ContactConstraint::solve (solver.rs:203-252) in the reference's operation order with the other components fetched through DPP
    // quad permutes: a cross product is 2 multiplies and a subtraction per lane (its rotated operands are DPP sources or were loaded
    // rotated), a matrix-vector product 3 multiplies and 2 adds, a dot product a multiply, a move and two adds.  ~105 float
    // instructions per lane where the one-lane form needs ~330 on one dependent chain - and that chain is what a trip's time is made
    // of - at 16 nodes per wave trip.  Bit-identical: same operations, same order, no contraction.  A wave takes a quad trip while the
    // queue holds at most F.quad_max nodes (the latency-bound stretches of a launch: few nodes ready, every one on somebody's critical
    // path) and a one-lane-per-node trip of up to 64 when it holds more (the throughput-bound stretches).
    const uint32_t k4 = lane & 3u, quad = lane >> 2;
    const uint32_t cc = k4 == 3u ? 2u : k4, c1 = cc == 2u ? 0u : cc + 1u, c2 = cc == 0u ? 2u : cc - 1u;  // this lane's component, the next, the one after
    // quad permutes (dpp_ctrl = sel0 | sel1 << 2 | sel2 << 4 | sel3 << 6): the value of the lane holding the NEXT component, the one after, component j
#define QP(x, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, (float)(x)), (ctrl), 0xF, 0xF, true))
#define Q_NEXT(x) QP(x, 0x09)   /* [1, 2, 0, 0] */
#define Q_PREV(x) QP(x, 0x52)   /* [2, 0, 1, 1] */
#define Q_X(x) QP(x, 0x00)
#define Q_Y(x) QP(x, 0x55)
#define Q_Z(x) QP(x, 0xAA)
    auto quad_trip = [&](uint32_t h0, uint32_t tl) -> bool {
      uint32_t take = min(tl - h0, 16u);  // up to 16 ready nodes: one per quad
      if (take) {
        uint32_t got = 0u;
        if (lane == 0) {
          uint32_t expect = h0;
          if (__hip_atomic_compare_exchange_strong(q.head, &expect, h0 + take, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) got = take;
        }
        take = __builtin_amdgcn_readfirstlane(got);
      }
      PF_STAMP(pf_t1);
      if (!take) return false;
      const bool act = quad < take;
      if (act) {
        uint16_t* cell = &q.ring[f6_wrap(q, h0 + quad)];
        uint32_t e;
        do { e = __hip_atomic_load(cell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); } while (!(e & 0x8000u));  // the pusher is between its two writes
        if (k4 == 0u) *cell = 0;  // (the quad's four lanes read the cell together, above)
        const uint32_t slot = e & 0x7FFFu;
        const uint32_t stw = __hip_atomic_load(&s_state[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const uint32_t round = (stw >> kF6StIterShift) & kF6StIterMask;
        uint64_t t_seen = 0;
        if (TRACE) t_seen = F6_NODE_CLOCK();
        const uint32_t c = s_c[slot], ref = stw >> kF6StRefShift;
        const uint2 sw = s_succ[slot];
        const uint32_t ai = ref & kF6NoBody, bi_raw = (ref >> kF6BodyBits) & kF6NoBody;
        const bool has_b = bi_raw != kF6NoBody;
        const uint32_t bi = has_b ? bi_raw : ai;
        // ---- operands: this lane's component of every vector
        float* fa = reinterpret_cast<float*>(s_body + 2 * (size_t)ai);
        float* fb = reinterpret_cast<float*>(s_body + 2 * (size_t)bi);
        float va = fa[cc], oa = fa[3u + cc], vb = fb[cc], ob = fb[3u + cc];
        const uint32_t ga = f2u(fa[6]), gb = f2u(fb[6]);
        // constants of a body as a flat row: [0] inverse mass, [1 + 3 j + i] = column j, row i of the world inverse inertia
        float im_a, ia0, ia1, ia2, im_b, ib0, ib1, ib2;
        {
          const float* ka;
          if (CL) ka = reinterpret_cast<const float*>(s_const + 5 * (size_t)ai);
          else ka = reinterpret_cast<const float*>(srec + 4 * (size_t)ga) + 6;
          im_a = ka[0]; ia0 = ka[1u + cc]; ia1 = ka[4u + cc]; ia2 = ka[7u + cc];
          if (CL == 2 || (CL == 1 && bi < F.nb)) {
            const float* kb = reinterpret_cast<const float*>(s_const + 5 * (size_t)bi);
            im_b = kb[0]; ib0 = kb[1u + cc]; ib1 = kb[4u + cc]; ib2 = kb[7u + cc];
          } else {
            const float* kb = reinterpret_cast<const float*>(srec + 4 * (size_t)gb) + 6;
            im_b = kb[0]; ib0 = kb[1u + cc]; ib1 = kb[4u + cc]; ib2 = kb[7u + cc];
          }
        }
        float n, t0, t1, ra1, ra2, rb1, rb2, bias, nmass, tm0, tm1, nimp;  // (the arms only ever enter cross products: their rotated components)
        if (RL) {  // the staged half: words 0..15 = CRec words 4..19, then {n.x, n.y, tmass1, nimp}
          const float* R = reinterpret_cast<const float*>(s_rec + kF6RecWords * (size_t)slot);
          n = R[cc == 2u ? 0u : 16u + cc]; t0 = R[1u + cc]; t1 = R[4u + cc];
          ra1 = R[7u + c1]; ra2 = R[7u + c2]; rb1 = R[10u + c1]; rb2 = R[10u + c2];
          bias = R[13]; nmass = R[14]; tm0 = R[15]; tm1 = R[18]; nimp = R[19];
        } else {  // CRec words: 2 n, 5 t0, 8 t1, 11 ra, 14 rb, 17 bias, 18 nmass, 19 20 tmass, 22 nimp - one 128-byte line
          const float* G = reinterpret_cast<const float*>(&cons[c]);
          n = G[2u + cc]; t0 = G[5u + cc]; t1 = G[8u + cc];
          ra1 = G[11u + c1]; ra2 = G[11u + c2]; rb1 = G[14u + c1]; rb2 = G[14u + c2];
          bias = G[17]; nmass = G[18]; tm0 = G[19]; tm1 = G[20];
          nimp = NLS ? s_nimp[slot] : G[22];
        }
        if (!has_b) { vb = 0.0f; ob = 0.0f; im_b = 0.0f; ib0 = 0.0f; ib1 = 0.0f; ib2 = 0.0f; }  // static_dyn(): physics.rs:289-302
        PF_STAMP(pf_t2a);
        PF_STAMP_V(pf_t2);
        // ---- the solve, component-parallel (solve_core in k_links.h is the one-lane statement of the same operations)
        // cross(w, r)[c] = w[c+1] r[c+2] - w[c+2] r[c+1];  dot = (p.x + p.y) + p.z;  (M v)[c] = (M[c][0] v.x + M[c][1] v.y) + M[c][2] v.z
        auto cross_wr = [&](float w, float r1_, float r2_) -> float { return Q_NEXT(w) * r2_ - Q_PREV(w) * r1_; };  // dynamic w, rotated constant r
        auto cross_rw = [&](float r1_, float r2_, float w) -> float { return r1_ * Q_PREV(w) - r2_ * Q_NEXT(w); };  // constant r, dynamic w
        auto dot3 = [&](float p) -> float { return (Q_X(p) + Q_Y(p)) + Q_Z(p); };
        auto mat3 = [&](float m0, float m1, float m2, float v) -> float { return (m0 * Q_X(v) + m1 * Q_Y(v)) + m2 * Q_Z(v); };
        const float dv = ((vb + cross_wr(ob, rb1, rb2)) - va) - cross_wr(oa, ra1, ra2);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const float t = k == 0 ? t0 : t1, tm = k == 0 ? tm0 : tm1;
          const float lambda = -dot3(dv * t) * tm;
          const float imp = t * lambda;
          va = va - imp * im_a;
          oa = oa - mat3(ia0, ia1, ia2, cross_rw(ra1, ra2, imp));
          vb = vb + imp * im_b;
          ob = ob + mat3(ib0, ib1, ib2, cross_rw(rb1, rb2, imp));
        }
        {
          const float dv2 = ((vb + cross_wr(ob, rb1, rb2)) - va) - cross_wr(oa, ra1, ra2);
          const float vn = dot3(dv2 * n);
          float lambda = nmass * (-vn + bias);
          const float prev = nimp;
          nimp = fmax_rs(prev + lambda, 0.0f);
          lambda = nimp - prev;
          const float imp = n * lambda;
          va = va - imp * im_a;
          oa = oa - mat3(ia0, ia1, ia2, cross_rw(ra1, ra2, imp));
          vb = vb + imp * im_b;
          ob = ob + mat3(ib0, ib1, ib2, cross_rw(rb1, rb2, imp));
        }
        // ---- ConstrainedSet::set: every lane its component (lane 3 holds lane 2's: it writes nothing)
        if (k4 < 3u) {
          fa[cc] = va; fa[3u + cc] = oa;
          if (has_b) { fb[cc] = vb; fb[3u + cc] = ob; }
        }
        if (k4 == 0u) {
          if (RL) reinterpret_cast<float*>(s_rec + kF6RecWords * (size_t)slot)[19] = nimp;
          else if (NLS) s_nimp[slot] = nimp;
          else cons[c].nimp = nimp;
          // re-arm: one arrival per dynamic body and iteration from now on, and one more iteration done
          __hip_atomic_fetch_add(&s_state[slot], (1u << kF6StIterShift) + (has_b ? 2u : 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
#ifdef MGF_F6_STORE_WAIT  // (r04: not needed - a wave's DS instructions execute in issue order, so the release's atomic cannot overtake the stores; 1-2.5 % faster without)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // velocities are in LDS before any local successor hears of it
#else
        asm volatile("" ::: "memory");
#endif
        // (a body's whole velocity - what a message or the final write-back of a foreign body carries - is read back from its LDS slot by
        // the one lane that needs it: no gathering across the quad on the path of the nodes that need neither)
        if (round + 1u == iters && k4 == 1u) {  // the end of a foreign body's chain: its home block does not write it back
          if (has_b && (sw.y & kF6Wrap) && bi >= F.nb) {
            const float4 p0 = s_body[2 * (size_t)bi], p1 = s_body[2 * (size_t)bi + 1];
            srec[4 * (size_t)gb] = p0;
            *reinterpret_cast<float2*>(&srec[4 * (size_t)gb + 1]) = make_float2(p1.x, p1.y);
          }
        }
        PF_STAMP(pf_t3);
#ifndef MGF_F6_PROFILE
        if (TRACE && k4 == 0u) {
          trace[2 * ((size_t)round * C_trace + c)] = t_seen & ~3ull;
          trace[2 * ((size_t)round * C_trace + c) + 1] = F6_NODE_CLOCK();
        }
#endif
        // ---- the release: lane 0 of the quad releases body a's successor, lane 1 body b's - both in the same instructions
        if (k4 < 2u) {
          const uint32_t w = k4 == 0u ? sw.x : sw.y;
          const bool live = (k4 == 0u || has_b) && round + ((w & kF6Wrap) ? 1u : 0u) < iters;
          const bool loc = live && !(w & kF6Remote);
          const uint32_t ws = w & kSlotMask;
          if (live && (w & kF6Remote)) {  // a message: the body's velocity, where it goes, the launch's tag in every granule - and on we go
            const uint32_t chn = (w >> 24) & (kF6Chan - 1u);  // (6 bits)
            const uint32_t pos = __hip_atomic_fetch_add(&s_out_tail[chn], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const uint32_t byte = (s_out_base[chn] + pos) * (16u * kF6MsgWords);
            const float tg = u2f(epoch);
            const uint32_t xi = k4 == 0u ? ai : bi;
            const float4 p0 = s_body[2 * (size_t)xi], p1 = s_body[2 * (size_t)xi + 1];
            const v4f_t m0 = {p0.x, p0.y, p0.z, tg}, m1 = {p0.w, p1.x, p1.y, tg};
            v4f_t m2 = {u2f(w & 0x00FFFFFFu), TRACE ? u2f((uint32_t)F6_NODE_CLOCK()) : 0.0f, 0.0f, tg};
            __builtin_amdgcn_raw_buffer_store_b128(m0, rmb, (int)byte, 0, kSc1);
            __builtin_amdgcn_raw_buffer_store_b128(m1, rmb, (int)(byte + 16u), 0, kSc1);
            __builtin_amdgcn_raw_buffer_store_b128(m2, rmb, (int)(byte + 32u), 0, kSc1);
            __hip_atomic_fetch_max(F.tails + s_out_tidx[chn], ((unsigned long long)epoch << 32) | (pos + 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          // (the local release comes AFTER the message: the slow edge leaves first - config 2 -1.6 %, settled -0.7 %, r04)
          const uint32_t was = __hip_atomic_fetch_sub(loc ? &s_state[ws] : &s_dummy[k4], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          if (loc && (was & kF6StArrMask) == 1u) f6_push(q, ws);
        }
      }
      if (lane == 0) __hip_atomic_fetch_sub(s_left, take, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#ifdef MGF_F6_PROFILE
      {
        uint64_t pf_t4; PF_STAMP(pf_t4);
        const uint64_t a2 = __shfl(pf_t2, 0), a3 = __shfl(pf_t3, 0), a2a = __shfl(pf_t2a, 0);
        pf_acc6 += a2a - pf_t1;
        pf_acc[0] += pf_t1 - pf_t0; pf_acc[1] += a2 - pf_t1; pf_acc[2] += a3 - a2; pf_acc[3] += pf_t4 - a3; ++pf_trips; pf_nodes += (uint64_t)take;
      }
#endif
      return true;
    };
    for (;;) {
      PF_STAMP(pf_t0);
      const uint32_t left = __builtin_amdgcn_readfirstlane(__hip_atomic_load(s_left, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
      const uint32_t h0 = __builtin_amdgcn_readfirstlane(__hip_atomic_load(q.head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
      const uint32_t tl = __builtin_amdgcn_readfirstlane(__hip_atomic_load(q.tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
      if (left == 0u) break;
      if (tl != h0) {
        const bool did = (QD && tl - h0 <= F.quad_max) ? quad_trip(h0, tl) : scalar_trip(h0, tl);
        if (did) { spins = 0; continue; }
      }
#ifdef MGF_F6_PROFILE
      { uint64_t pf_now; PF_STAMP(pf_now); pf_idle_t += pf_now - pf_t0; pf_acc[4] += 1; }
#endif
      if (MGF_F6_IDLE_SLEEP > 0) __builtin_amdgcn_s_sleep(MGF_F6_IDLE_SLEEP);
      if ((++spins & 255u) == 0u) {
        bool give_up = spins > spin_limit;
        if (give_up) __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (give_up || __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
      }
    }
#ifdef MGF_F6_PROFILE
    if (TRACE && wave == 0 && lane == 0) {
      tstat[13] = pf_acc[0]; tstat[14] = pf_acc[1]; tstat[15] = pf_acc[2]; tstat[16] = pf_acc[3]; tstat[17] = pf_trips; tstat[18] = pf_nodes; tstat[19] = pf_idle_t; tstat[20] = pf_acc[4]; tstat[21] = pf_acc6;
    }
#endif
#undef QP
#undef Q_NEXT
#undef Q_PREV
#undef Q_X
#undef Q_Y
#undef Q_Z
  }
#undef PF_STAMP
#undef PF_STAMP_V
  __syncthreads();
  if (TRACE && threadIdx.x == 0) { tstat[11] = wall_clock64(); tstat[23] = clock64() - trace_c0; }  // (23: shader clocks between stamps 8 and 11)
  // the accumulated normal impulses go back to the records (ContactState lives on: mgf_world_read_constraints, a later solve)
  if (NLS) for (uint32_t idx = t; idx < N; idx += kF6Threads) cons[s_c[idx]].nimp = s_nimp[idx];
  if (RL) for (uint32_t idx = t; idx < N; idx += kF6Threads) cons[s_c[idx]].nimp = s_rec[kF6RecWords * (size_t)idx + 4].w;
  // own bodies go back to the RigidBodyVec (a body whose chain ends in another block was written there)
  for (uint32_t i = t; i < n_own; i += kF6Threads) {
    const uint32_t x = F.ident ? p_lo + i : F.sidx[p_lo + i];
    if (!F.skipwb[x]) {
      srec[4 * (size_t)x] = s_body[2 * i];
      const float4 s1 = s_body[2 * i + 1];
      *reinterpret_cast<float2*>(&srec[4 * (size_t)x + 1]) = make_float2(s1.x, s1.y);
    }
  }
  if (TRACE) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); if (threadIdx.x == 0) tstat[12] = wall_clock64(); }
}

}  // namespace mgf
