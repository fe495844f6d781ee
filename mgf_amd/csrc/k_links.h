// Order-preserving dependency links of a constraint list and the launch-per-frontier solver (mode 0).  (Part of the kernel set described in kernels.h.)
#pragma once
#include "k_front_rows.h"

namespace mgf {

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
// `ext` (the body store is kept in an internal order, host_perm.inc): a body's own range still comes first - the bodies it
// meets as `b` are constraints of bodies with a LARGER order id, all inserted later - but ids no longer ascend with the
// insertion order across bodies: the row is sorted by (order id of the constraint's body a, id).
// (r05: a row entry carries what the sort and the links need of its constraint - its id, body a's slot and body a's order id - so
// that nothing has to be looked up through the id: the key used to be two dependent gathers, K.ab[c].x and ext[..], per comparison)
__device__ __forceinline__ unsigned long long rev_key(const RevEnt& e) { return ((unsigned long long)e.oid << 32) | e.c; }
__device__ __forceinline__ void rev_sort_in_place(RevEnt* row, uint32_t nb) {  // (rows longer than the register path: insertion sort where they lie)
  for (uint32_t a = 1; a < nb; ++a) {
    const RevEnt v = row[a];
    const unsigned long long kv = rev_key(v);
    uint32_t b = a;
    while (b > 0 && rev_key(row[b - 1]) > kv) { row[b] = row[b - 1]; --b; }
    row[b] = v;
  }
}
__global__ __launch_bounds__(kBlock) void k_chain_rows(uint32_t n, ConsLinks K, const uint32_t* base, const uint32_t* degb, RevEnt* rev,
                                                       uint32_t rev_cap, const uint32_t* rev_flag, StepCounts* sc, const uint32_t* tcn, uint32_t n_owned,
                                                       uint32_t* n_ghost_cons, const uint32_t* only_if, const uint32_t* ext) {
  // (behind k_flow6_links, which has done the tick's bookkeeping: the links are only needed if the block-local solver declined)
  if (only_if && *only_if == 0u) return;
  uint32_t x = blockIdx.x * kBlock + threadIdx.x;
  // constraints whose obj_a is a ghost (ids are ascending in obj_a): the copies of seam constraints (tiles count them once)
  if (x == 0 && n_ghost_cons) *n_ghost_cons = (sc->fail || *rev_flag) ? 0u : base[n] - base[n_owned];
  if (*rev_flag) {
    if (x == 0) { sc->C = 0; sc->Ct = 0; sc->fail |= kFailRevRow; }
    return;
  }
  // (a list-capacity miss: base[] counts constraints that were never written; the tick is re-run.  kFailFlow6 alone is not
  // such a miss: the list is whole, only the block-local solver's tables did not fit - and these links are what its stand-by walks)
  if (x >= n || (sc->fail & ~kFailFlow6)) return;
  const uint32_t lo = base[x], na = base[x + 1] - lo, nb = degb[x];
  const uint32_t total = na + nb;
  if (total == 0) return;
  const uint32_t nt = tcn[x];
  RevEnt* row = rev + (size_t)x * rev_cap;
  rev_sort_in_place(row, nb);  // ascending (order id of body a, constraint id) = insertion order
  uint32_t* succ = reinterpret_cast<uint32_t*>(K.succ);
  const uint32_t first = na ? lo : row[0].c;
  for (uint32_t k = 0; k < total; ++k) {
    const bool last = k + 1 == total;
    const uint32_t u = k < na ? lo + k : row[k - na].c, role = k < na ? 0u : 1u;
    const uint32_t wid = last ? first : (k + 1 < na ? lo + k + 1 : row[k + 1 - na].c);
    // the successor has two dynamic bodies unless it is one of this body's own terrain constraints - the first tcn[x] of its
    // range (no look-up of the successor's (a, b): that was a dependent gather per link)
    const bool two = !(wid >= lo && wid - lo < nt);
    succ[2 * u + role] = wid | (two ? kSuccTwo : 0u) | (last ? kSuccWrap : 0u);
    K.pred[2 * u + role] = k > 0 ? 1 : 0;  // predecessor on this body inside one iteration
  }
}

// The insertion order of the tick's list when the body store is in an internal order (host_perm.inc): ids follow the slots, the
// insertion order follows the caller's body indices.  cnt_e[e] = constraints body e inserts; after its scan (base_e),
// canon[base_e[e] + k] = the k-th constraint of body e.  Needed by the executors that walk ids in order (k_solve_flow).
__global__ __launch_bounds__(kBlock) void k_canon_counts(uint32_t n, const uint32_t* base, const uint32_t* ext, uint32_t* cnt_e) {
  const uint32_t s = blockIdx.x * kBlock + threadIdx.x;
  if (s < n) cnt_e[ext[s]] = base[s + 1] - base[s];
  else if (s == n) cnt_e[n] = 0u;
}
__global__ __launch_bounds__(kBlock) void k_canon_fill(uint32_t n, const StepCounts* sc, const uint32_t* base, const uint32_t* ext, const uint32_t* base_e,
                                                       uint32_t* canon) {
  const uint32_t s = blockIdx.x * kBlock + threadIdx.x;
  if (s >= n || (sc->fail & ~kFailFlow6)) return;  // (a capacity miss: base[] counts constraints that were never written)
  const uint32_t lo = base[s], m = base[s + 1] - lo, at = base_e[ext[s]];
  for (uint32_t k = 0; k < m; ++k) canon[at + k] = lo + k;
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

// Solver::solve cannot fail (solver.rs:72-78), a persistent launch can (it needs all its workgroups resident at once: a device shared
// with another process, a CU mask): the velocities and accumulated impulses as they were before the launch, and back - the host then
// runs the list with the launch-per-frontier executor, which needs no residency (solver_abort_fallback).
__global__ __launch_bounds__(kBlock) void k_solver_snapshot(const float4* srec, uint32_t n, float4* vsnap, const CRec* cons, uint32_t C, float* nsnap, const uint32_t* guard) {
  if (*guard) return;  // (a speculative tick behind one that failed or gave up: the copy in place is that tick's)
  const uint32_t t = blockIdx.x * kBlock + threadIdx.x;
  if (t < n) { vsnap[2 * (size_t)t] = srec[4 * (size_t)t]; vsnap[2 * (size_t)t + 1] = srec[4 * (size_t)t + 1]; }
  if (nsnap && t < C) nsnap[t] = cons[t].nimp;
}
__global__ __launch_bounds__(kBlock) void k_solver_restore(float4* srec, uint32_t n, const float4* vsnap, CRec* cons, uint32_t C, const float* nsnap, uint32_t* abort_flag) {
  const uint32_t t = blockIdx.x * kBlock + threadIdx.x;
  if (t < n) { srec[4 * (size_t)t] = vsnap[2 * (size_t)t]; srec[4 * (size_t)t + 1] = vsnap[2 * (size_t)t + 1]; }  // (v, w; inv_mass and I00 are constants)
  if (t < C) cons[t].nimp = nsnap ? nsnap[t] : 0.0f;  // (no snapshot: the list is fresh from ContactConstraint::new)
  if (t == 0) *abort_flag = 0u;
}

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

}  // namespace mgf
