// C-ABI of the MI355X-native mgf hot path (include/mgf_hip.h): host orchestration of the
// hand-written kernels in kernels.h.  No CPU compute fallback exists: every compute entry point
// needs a HIP device and returns MGF_ERR_HIP without one.
#include <stdarg.h>

#include <algorithm>
#include <cmath>
#include <memory>

#include "common.h"
#include "kernels.h"
#include "scene_io.h"

using namespace mgf;

// ---------------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
namespace mgf {
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace mgf
extern "C" const char* mgf_last_error(void) { return g_err; }
extern "C" const char* mgf_version(void) { return "mgf-hip 0.1 (gfx950)"; }
extern "C" mgf_params mgf_default_params(void) {
  mgf_params p;
  p.baumgarte = 0.2f;               // solver.rs:278
  p.penetration_slop = 0.05f;       // solver.rs:277
  p.persistent_threshold_sq = 0.5f; // manifold.rs:38
  p.collision_epsilon = kCollisionEps;
  p.fat_margin = 0.25f;             // world.rs:181
  return p;
}

static inline unsigned nblk(size_t n) { return (unsigned)((n + kBlock - 1) / kBlock); }
static mgf_status fail(mgf_status s, const char* msg) { set_error("%s", msg); return s; }
#define LAUNCH_CHECK() MGF_HIP_TRY(hipGetLastError())

// ---------------------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------------------
extern "C" mgf_status mgf_ctx_create(int device, mgf_ctx** out) {
  if (!out) return fail(MGF_ERR_INVALID, "out is NULL");
  *out = nullptr;
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count == 0) {
    set_error("no HIP device available (%s); mgf-hip has no CPU fallback", e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    return MGF_ERR_HIP;
  }
  if (device < 0 || device >= count) return fail(MGF_ERR_INVALID, "device index out of range");
  MGF_HIP_TRY(hipSetDevice(device));
  std::unique_ptr<mgf_ctx> c(new mgf_ctx());
  c->device = device;
  MGF_HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  c->pinned_bytes = 1 << 16;
  MGF_HIP_TRY(hipHostMalloc(&c->pinned, c->pinned_bytes, hipHostMallocDefault));
  hipDeviceProp_t prop;
  MGF_HIP_TRY(hipGetDeviceProperties(&prop, device));
  c->num_cus = prop.multiProcessorCount;
  *out = c.release();
  return MGF_OK;
}
extern "C" mgf_status mgf_ctx_set_stream(mgf_ctx* ctx, void* stream) {
  if (!ctx) return fail(MGF_ERR_INVALID, "ctx is NULL");
  MGF_HIP_TRY(hipSetDevice(ctx->device));
  MGF_HIP_TRY(hipStreamSynchronize(ctx->stream));
  if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
  ctx->stream = static_cast<hipStream_t>(stream);
  ctx->own_stream = false;
  return MGF_OK;
}
extern "C" void mgf_ctx_destroy(mgf_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
  ctx->stream = nullptr;
  if (ctx->prim_tmp) (void)hipFree(ctx->prim_tmp);
  if (ctx->pinned) (void)hipHostFree(ctx->pinned);
  delete ctx;
}

static mgf_status ctx_bind(mgf_ctx* ctx) {
  if (!ctx) return fail(MGF_ERR_HIP, "no device context: this entry point computes on the GPU (mgf-hip has no CPU fallback)");
  MGF_HIP_TRY(hipSetDevice(ctx->device));
  (void)hipGetLastError();  // drop any stale sticky error from unrelated earlier calls
  return MGF_OK;
}
template <class T>
static mgf_status h2d(mgf_ctx* ctx, T* dst, const T* src, size_t n) {
  if (n == 0) return MGF_OK;
  MGF_HIP_TRY(hipMemcpyAsync(dst, src, n * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
  MGF_HIP_TRY(hipStreamSynchronize(ctx->stream));
  return MGF_OK;
}
template <class T>
static mgf_status d2h(mgf_ctx* ctx, T* dst, const T* src, size_t n) {
  if (n == 0) return MGF_OK;
  MGF_HIP_TRY(hipMemcpyAsync(dst, src, n * sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
  MGF_HIP_TRY(hipStreamSynchronize(ctx->stream));
  return MGF_OK;
}

// ---------------------------------------------------------------------------------------------
// host tree + device mirror (shared by mgf_bvh and mgf_mesh)
// ---------------------------------------------------------------------------------------------
struct TreeMirror {
  HostBvh tree;
  DBuf<DevNode> d_nodes;
  uint64_t uploaded_version = ~0ull;
  mgf_status sync(mgf_ctx* ctx) {
    if (uploaded_version == tree.version()) return MGF_OK;
    std::vector<DevNode> flat;
    tree.flatten(&flat);
    MGF_TRY(d_nodes.ensure(std::max<size_t>(flat.size(), 1), ctx->stream));
    MGF_TRY(h2d(ctx, d_nodes.p, flat.data(), flat.size()));
    uploaded_version = tree.version();
    return MGF_OK;
  }
  TerrainDev dev(const float4* verts, const uint4* faces, V3 x, uint32_t* err) const {
    TerrainDev t;
    t.nodes = d_nodes.p; t.verts = verts; t.faces = faces;
    t.root = (uint32_t)tree.root();
    t.n_nodes = tree.empty() ? 0u : (uint32_t)tree.slots();
    t.x[0] = x.x; t.x[1] = x.y; t.x[2] = x.z;
    t.err = err;
    return t;
  }
};

struct mgf_bvh {
  mgf_ctx* ctx;
  TreeMirror m;
};
struct mgf_mesh {
  mgf_ctx* ctx;
  V3 x = mk3(0, 0, 0);
  std::vector<V3> verts;
  std::vector<uint32_t> faces;  // 3 per face
  TreeMirror m;                 // BVH<AABB, usize> over faces (mesh.rs:36)
  DBuf<float4> d_verts;
  DBuf<uint4> d_faces;
  uint64_t geom_version = 0, uploaded_geom = ~0ull;
  mgf_status sync() {
    MGF_TRY(m.sync(ctx));
    if (uploaded_geom != geom_version) {
      std::vector<float4> hv(verts.size());
      for (size_t i = 0; i < verts.size(); ++i) hv[i] = make_float4(verts[i].x, verts[i].y, verts[i].z, 0.0f);
      std::vector<uint4> hf(faces.size() / 3);
      for (size_t i = 0; i < hf.size(); ++i) hf[i] = make_uint4(faces[3 * i], faces[3 * i + 1], faces[3 * i + 2], 0);
      MGF_TRY(d_verts.ensure(std::max<size_t>(hv.size(), 1), ctx->stream));
      MGF_TRY(d_faces.ensure(std::max<size_t>(hf.size(), 1), ctx->stream));
      MGF_TRY(h2d(ctx, d_verts.p, hv.data(), hv.size()));
      MGF_TRY(h2d(ctx, d_faces.p, hf.data(), hf.size()));
      uploaded_geom = geom_version;
    }
    return MGF_OK;
  }
  TerrainDev dev(uint32_t* err) const { return m.dev(d_verts.p, d_faces.p, x, err); }
  // Morton-cell grid over the face boxes + DFS ranks of the faces (k_terrain_grid); built by build_face_grid
  struct Grid {
    bool ready = false;
    uint32_t levels = 0, n_faces = 0;
    DBuf<float4> fb_c, fb_r;
    DBuf<uint32_t> cell_of, cell_rank, cell_cnt, cell_lo, sidx, brank, rank_of_face, face_of_rank, leaf_of_face, parent;
    DBuf<LeafRec> leaves;
  DBuf<float4> lcol;             // cell-ordered (collider, motion) copies for the fused sphere test of k_pair_grid
  DBuf<uint32_t> pair_stat;      // 64 partial sums of the partners accepted by the fused broadphase
  bool tick_fused = false;
  int64_t opt_no_fused_narrowphase = 0;
    DBuf<SceneBounds> sb;
  } grid;
  FaceGrid face_grid() const {
    FaceGrid G;
    G.T.nodes = nullptr; G.T.leaves = grid.leaves.p; G.T.sidx = grid.sidx.p; G.T.cell_lo = grid.cell_lo.p; G.T.lcol = nullptr;
    G.T.n = grid.n_faces; G.T.levels = grid.levels; G.T.err = nullptr; G.T.dbg = nullptr;
    G.sb = grid.sb.p; G.rank_of_face = grid.rank_of_face.p; G.leaf_of_face = grid.leaf_of_face.p; G.parent = grid.parent.p;
    return G;
  }
};

static inline Box to_box(const mgf_aabb& a) { Box b; b.c = mk3(a.c.x, a.c.y, a.c.z); b.r = mk3(a.r.x, a.r.y, a.r.z); return b; }
static inline mgf_aabb from_box(const Box& b) { mgf_aabb a; a.c = {b.c.x, b.c.y, b.c.z}; a.r = {b.r.x, b.r.y, b.r.z}; return a; }

// ---- mgf_bvh ---------------------------------------------------------------------------------
extern "C" mgf_status mgf_bvh_new(mgf_ctx* ctx, mgf_bvh** out) {  // ctx may be NULL: host-only tree (insert/remove/inspect; queries need a device)
  if (!out) return fail(MGF_ERR_INVALID, "NULL argument");
  *out = new mgf_bvh{ctx, {}};
  return MGF_OK;
}
extern "C" mgf_status mgf_bvh_with_capacity(mgf_ctx* ctx, uint64_t cap, mgf_bvh** out) {
  MGF_TRY(mgf_bvh_new(ctx, out));
  (*out)->m.tree.reserve(cap);
  return MGF_OK;
}
extern "C" void mgf_bvh_free(mgf_bvh* b) {
  if (!b) return;
  if (b->ctx) (void)hipSetDevice(b->ctx->device);
  delete b;
}
extern "C" int32_t mgf_bvh_empty(const mgf_bvh* b) { return (!b || b->m.tree.empty()) ? 1 : 0; }
extern "C" mgf_status mgf_bvh_clear(mgf_bvh* b) {
  if (!b) return fail(MGF_ERR_INVALID, "bvh is NULL");
  b->m.tree.clear();
  return MGF_OK;
}
extern "C" mgf_status mgf_bvh_insert(mgf_bvh* b, const mgf_aabb* key, uint64_t val, uint64_t* id) {
  if (!b || !key) return fail(MGF_ERR_INVALID, "NULL argument");
  if (val > 0x7FFFFFFFull) return fail(MGF_ERR_INVALID, "leaf values are limited to 31 bits on the device");
  uint64_t r = b->m.tree.insert(to_box(*key), val);
  if (id) *id = r;
  return MGF_OK;
}
extern "C" mgf_status mgf_bvh_remove(mgf_bvh* b, uint64_t id) {
  if (!b) return fail(MGF_ERR_INVALID, "bvh is NULL");
  if (!b->m.tree.used(id)) return fail(MGF_ERR_NOT_OCCUPIED, "index is not occupied");
  b->m.tree.remove(id);
  return MGF_OK;
}
extern "C" mgf_status mgf_bvh_root(const mgf_bvh* b, uint64_t* id) {
  if (!b || !id) return fail(MGF_ERR_INVALID, "NULL argument");
  if (b->m.tree.empty()) return fail(MGF_ERR_EMPTY, "BVH is empty, there is no root node");
  *id = b->m.tree.root();
  return MGF_OK;
}
extern "C" mgf_status mgf_bvh_get_leaf(const mgf_bvh* b, uint64_t id, uint64_t* val) {
  if (!b || !val) return fail(MGF_ERR_INVALID, "NULL argument");
  if (!b->m.tree.used(id)) return fail(MGF_ERR_NOT_OCCUPIED, "index is not occupied");
  if (!b->m.tree.node(id).leaf) return fail(MGF_ERR_NOT_LEAF, "node is not a leaf");
  *val = b->m.tree.node(id).value;
  return MGF_OK;
}
extern "C" mgf_status mgf_bvh_bounds(const mgf_bvh* b, uint64_t id, mgf_aabb* out) {
  if (!b || !out) return fail(MGF_ERR_INVALID, "NULL argument");
  if (!b->m.tree.used(id)) return fail(MGF_ERR_NOT_OCCUPIED, "index is not occupied");
  *out = from_box(b->m.tree.node(id).box);
  return MGF_OK;
}
// Debug/introspection used by the structural parity tests: per slot
// {used, height, parent, is_leaf, value|child1, child2}; returns slot count.
extern "C" MGF_API int64_t mgf_bvh_dump(const mgf_bvh* b, int64_t* out6, mgf_aabb* boxes, int64_t cap) {
  if (!b) return -1;
  int64_t n = (int64_t)b->m.tree.slots();
  for (int64_t i = 0; i < n && i < cap; ++i) {
    int64_t* o = out6 + 6 * i;
    if (!b->m.tree.used((uint64_t)i)) { o[0] = o[1] = o[2] = o[3] = o[4] = o[5] = 0; continue; }
    const HostBvh::Node& nd = b->m.tree.node((uint64_t)i);
    o[0] = 1; o[1] = nd.height; o[2] = (int64_t)nd.parent; o[3] = nd.leaf ? 1 : 0;
    o[4] = nd.leaf ? (int64_t)nd.value : (int64_t)nd.kid[0];
    o[5] = nd.leaf ? 0 : (int64_t)nd.kid[1];
    if (boxes) boxes[i] = from_box(nd.box);
  }
  return n;
}

static mgf_status tree_query_many(mgf_ctx* ctx, TreeMirror& m, const mgf_aabb* args, int64_t n, std::vector<uint32_t>* off,
                                  std::vector<uint32_t>* vals) {
  MGF_TRY(ctx_bind(ctx));
  off->assign((size_t)n + 1, 0);
  vals->clear();
  if (n == 0 || m.tree.empty()) return MGF_OK;
  MGF_TRY(m.sync(ctx));
  DBuf<float> d_boxes;
  DBuf<uint32_t> d_cnt, d_off, d_vals, d_err;
  MGF_TRY(d_boxes.ensure((size_t)n * 6, ctx->stream));
  MGF_TRY(d_cnt.ensure((size_t)n + 1, ctx->stream));
  MGF_TRY(d_off.ensure((size_t)n + 1, ctx->stream));
  MGF_TRY(d_err.ensure(1, ctx->stream));
  MGF_HIP_TRY(hipMemsetAsync(d_err.p, 0, 4, ctx->stream));
  MGF_TRY(h2d(ctx, d_boxes.p, reinterpret_cast<const float*>(args), (size_t)n * 6));
  TerrainDev T = m.dev(nullptr, nullptr, mk3(0, 0, 0), d_err.p);
  k_bvh_query<false><<<nblk(n), kBlock, 0, ctx->stream>>>(T, d_boxes.p, n, d_cnt.p, nullptr, nullptr);
  LAUNCH_CHECK();
  MGF_TRY(prim_exclusive_scan_u32(ctx, d_cnt.p, d_off.p, (size_t)n + 1));
  MGF_TRY(d2h(ctx, off->data(), d_off.p, (size_t)n + 1));
  uint32_t total = (*off)[n];
  vals->resize(total);
  if (total) {
    MGF_TRY(d_vals.ensure(total, ctx->stream));
    k_bvh_query<true><<<nblk(n), kBlock, 0, ctx->stream>>>(T, d_boxes.p, n, nullptr, d_off.p, d_vals.p);
    LAUNCH_CHECK();
    MGF_TRY(d2h(ctx, vals->data(), d_vals.p, total));
  }
  uint32_t err = 0;
  MGF_TRY(d2h(ctx, &err, d_err.p, 1));
  if (err) return fail(MGF_ERR_CAPACITY, "BVH traversal stack overflow (tree deeper than 64)");
  return MGF_OK;
}
extern "C" mgf_status mgf_bvh_query(mgf_bvh* b, const mgf_aabb* arg, mgf_bvh_hit_fn cb, void* user) {
  if (!b || !arg || !cb) return fail(MGF_ERR_INVALID, "NULL argument");
  std::vector<uint32_t> off, vals;
  MGF_TRY(tree_query_many(b->ctx, b->m, arg, 1, &off, &vals));
  for (uint32_t v : vals) { uint64_t v64 = v; cb(&v64, user); }
  return MGF_OK;
}
extern "C" mgf_status mgf_bvh_query_many(mgf_bvh* b, const mgf_aabb* args, int64_t n, uint64_t* out_offsets, uint64_t* out_vals,
                                         int64_t cap, int64_t* total) {
  if (!b || (!args && n) || !out_offsets) return fail(MGF_ERR_INVALID, "NULL argument");
  std::vector<uint32_t> off, vals;
  MGF_TRY(tree_query_many(b->ctx, b->m, args, n, &off, &vals));
  for (int64_t i = 0; i <= n; ++i) out_offsets[i] = off[(size_t)i];
  if (total) *total = (int64_t)vals.size();
  if ((int64_t)vals.size() > cap) return fail(MGF_ERR_CAPACITY, "out_vals too small");
  for (size_t i = 0; i < vals.size(); ++i) out_vals[i] = vals[i];
  return MGF_OK;
}

// BVH::raytrace bvh.rs:345-369 for a batch of particles: per particle the leaf values and the intersections
// with the leaf bounds, in the reference's visiting order (count pass, scan, fill pass).
static_assert(sizeof(ParticleIn) == sizeof(mgf_particle), "particle layout");
static_assert(sizeof(InterOut) == sizeof(mgf_intersection), "intersection layout");
static mgf_status tree_raytrace_many(mgf_ctx* ctx, TreeMirror& m, const mgf_particle* parts, int64_t n, std::vector<uint32_t>* off,
                                     std::vector<uint32_t>* vals, std::vector<mgf_intersection>* inters) {
  MGF_TRY(ctx_bind(ctx));
  off->assign((size_t)n + 1, 0);
  vals->clear(); inters->clear();
  if (n == 0 || m.tree.empty()) return MGF_OK;
  MGF_TRY(m.sync(ctx));
  DBuf<ParticleIn> d_parts;
  DBuf<uint32_t> d_cnt, d_off, d_vals, d_err;
  DBuf<InterOut> d_int;
  MGF_TRY(d_parts.ensure((size_t)n, ctx->stream));
  MGF_TRY(d_cnt.ensure((size_t)n + 1, ctx->stream));
  MGF_TRY(d_off.ensure((size_t)n + 1, ctx->stream));
  MGF_TRY(d_err.ensure(1, ctx->stream));
  MGF_HIP_TRY(hipMemsetAsync(d_err.p, 0, 4, ctx->stream));
  MGF_TRY(h2d(ctx, d_parts.p, reinterpret_cast<const ParticleIn*>(parts), (size_t)n));
  TerrainDev T = m.dev(nullptr, nullptr, mk3(0, 0, 0), d_err.p);
  k_bvh_raytrace<false><<<nblk(n), kBlock, 0, ctx->stream>>>(T, d_parts.p, n, d_cnt.p, nullptr, nullptr, nullptr);
  LAUNCH_CHECK();
  MGF_TRY(prim_exclusive_scan_u32(ctx, d_cnt.p, d_off.p, (size_t)n + 1));
  MGF_TRY(d2h(ctx, off->data(), d_off.p, (size_t)n + 1));
  uint32_t total = (*off)[n];
  vals->resize(total); inters->resize(total);
  if (total) {
    MGF_TRY(d_vals.ensure(total, ctx->stream));
    MGF_TRY(d_int.ensure(total, ctx->stream));
    k_bvh_raytrace<true><<<nblk(n), kBlock, 0, ctx->stream>>>(T, d_parts.p, n, nullptr, d_off.p, d_vals.p, d_int.p);
    LAUNCH_CHECK();
    MGF_TRY(d2h(ctx, vals->data(), d_vals.p, total));
    MGF_TRY(d2h(ctx, reinterpret_cast<InterOut*>(inters->data()), d_int.p, total));
  }
  uint32_t err = 0;
  MGF_TRY(d2h(ctx, &err, d_err.p, 1));
  if (err) return fail(MGF_ERR_CAPACITY, "BVH traversal stack overflow (tree deeper than 64)");
  return MGF_OK;
}
extern "C" mgf_status mgf_bvh_raytrace(mgf_bvh* b, const mgf_particle* arg, mgf_bvh_ray_fn cb, void* user) {
  if (!b || !arg || !cb) return fail(MGF_ERR_INVALID, "NULL argument");
  std::vector<uint32_t> off, vals;
  std::vector<mgf_intersection> inters;
  MGF_TRY(tree_raytrace_many(b->ctx, b->m, arg, 1, &off, &vals, &inters));
  for (size_t k = 0; k < vals.size(); ++k) { uint64_t v64 = vals[k]; cb(&v64, &inters[k], user); }
  return MGF_OK;
}
extern "C" mgf_status mgf_bvh_raytrace_many(mgf_bvh* b, const mgf_particle* args, int64_t n, uint64_t* out_offsets, uint64_t* out_vals,
                                            mgf_intersection* out_inter, int64_t cap, int64_t* total) {
  if (!b || (!args && n) || !out_offsets) return fail(MGF_ERR_INVALID, "NULL argument");
  std::vector<uint32_t> off, vals;
  std::vector<mgf_intersection> inters;
  MGF_TRY(tree_raytrace_many(b->ctx, b->m, args, n, &off, &vals, &inters));
  for (int64_t i = 0; i <= n; ++i) out_offsets[i] = off[(size_t)i];
  if (total) *total = (int64_t)vals.size();
  if ((int64_t)vals.size() > cap) return fail(MGF_ERR_CAPACITY, "output buffers too small");
  for (size_t i = 0; i < vals.size(); ++i) { if (out_vals) out_vals[i] = vals[i]; if (out_inter) out_inter[i] = inters[i]; }
  return MGF_OK;
}
// Intersects<Shape> / Intersects<AABB> for a batch of particles (collision.rs:169-373); hit[i] = 1 / 0.
extern "C" mgf_status mgf_intersections_batch(mgf_ctx* ctx, int64_t n, const mgf_particle* parts, const mgf_shape* shapes, const mgf_aabb* boxes,
                                              mgf_intersection* out, int32_t* hit) {
  if (n < 0 || (n && (!parts || (!shapes && !boxes) || (shapes && boxes) || !out || !hit))) return fail(MGF_ERR_INVALID, "bad argument");
  MGF_TRY(ctx_bind(ctx));
  if (n == 0) return MGF_OK;
  if (shapes)
    for (int64_t i = 0; i < n; ++i)
      if (shapes[i].kind != MGF_SPHERE && shapes[i].kind != MGF_CAPSULE && shapes[i].kind != MGF_TRIANGLE && shapes[i].kind != MGF_PLANE)
        return fail(MGF_ERR_INVALID, "intersection: shape must be sphere, capsule, triangle or plane");
  DBuf<ParticleIn> dp; DBuf<ShapeIn> ds; DBuf<float> db; DBuf<InterOut> dout; DBuf<int32_t> dh;
  MGF_TRY(dp.ensure((size_t)n, ctx->stream)); MGF_TRY(dout.ensure((size_t)n, ctx->stream)); MGF_TRY(dh.ensure((size_t)n, ctx->stream));
  MGF_TRY(h2d(ctx, dp.p, reinterpret_cast<const ParticleIn*>(parts), (size_t)n));
  if (shapes) { MGF_TRY(ds.ensure((size_t)n, ctx->stream)); MGF_TRY(h2d(ctx, ds.p, reinterpret_cast<const ShapeIn*>(shapes), (size_t)n)); }
  else { MGF_TRY(db.ensure(6 * (size_t)n, ctx->stream)); MGF_TRY(h2d(ctx, db.p, reinterpret_cast<const float*>(boxes), 6 * (size_t)n)); }
  k_intersections_batch<<<nblk(n), kBlock, 0, ctx->stream>>>(n, dp.p, shapes ? ds.p : nullptr, shapes ? nullptr : db.p, dout.p, dh.p);
  LAUNCH_CHECK();
  MGF_TRY(d2h(ctx, reinterpret_cast<InterOut*>(out), dout.p, (size_t)n));
  MGF_TRY(d2h(ctx, hit, dh.p, (size_t)n));
  return MGF_OK;
}

static inline Comp comp_of(const mgf_component& c) {
  Comp k; k.kind = c.tag; k.p = mk3(c.p.x, c.p.y, c.p.z); k.d = mk3(c.d.x, c.d.y, c.d.z); k.r = c.r;
  return k;
}
// ContactPruner + Manifold::from for n groups of LocalContacts (manifold.rs:42-148)
static_assert(sizeof(ManifoldOut) == sizeof(mgf_manifold), "manifold layout");
extern "C" mgf_status mgf_manifolds_from_contacts(mgf_ctx* ctx, const mgf_params* params, int64_t n, const uint64_t* offsets,
                                                  const mgf_local_contact* contacts, mgf_manifold* out) {
  if (n < 0 || (n && (!offsets || !out))) return fail(MGF_ERR_INVALID, "bad argument");
  MGF_TRY(ctx_bind(ctx));
  if (n == 0) return MGF_OK;
  for (int64_t i = 0; i < n; ++i) if (offsets[i + 1] < offsets[i]) return fail(MGF_ERR_INVALID, "offsets must be non-decreasing");
  const uint64_t total = offsets[n];
  if (total && !contacts) return fail(MGF_ERR_INVALID, "NULL contacts");
  mgf_params P = params ? *params : mgf_default_params();
  DBuf<unsigned long long> d_off; DBuf<LocalOut> d_lc; DBuf<ManifoldOut> d_out; DBuf<uint32_t> d_ovf;
  MGF_TRY(d_off.ensure((size_t)n + 1, ctx->stream)); MGF_TRY(d_lc.ensure(std::max<size_t>(total, 1), ctx->stream));
  MGF_TRY(d_out.ensure((size_t)n, ctx->stream)); MGF_TRY(d_ovf.ensure(1, ctx->stream));
  MGF_HIP_TRY(hipMemsetAsync(d_ovf.p, 0, 4, ctx->stream));
  MGF_TRY(h2d(ctx, d_off.p, reinterpret_cast<const unsigned long long*>(offsets), (size_t)n + 1));
  MGF_TRY(h2d(ctx, d_lc.p, reinterpret_cast<const LocalOut*>(contacts), (size_t)total));
  k_manifolds<<<nblk(n), kBlock, 0, ctx->stream>>>(n, d_off.p, d_lc.p, P.persistent_threshold_sq, P.collision_epsilon, d_out.p, d_ovf.p);
  LAUNCH_CHECK();
  MGF_TRY(d2h(ctx, reinterpret_cast<ManifoldOut*>(out), d_out.p, (size_t)n));
  uint32_t ovf = 0;
  MGF_TRY(d2h(ctx, &ovf, d_ovf.p, 1));
  if (ovf) return fail(MGF_ERR_CAPACITY, "a manifold keeps more than MGF_MANIFOLD_CAP contacts (n_contacts reports how many)");
  return MGF_OK;
}

// ---- scene I/O (serde_json shape of BVH<AABB, usize> and Mesh; scene_io.h) --------------------------------
static mgf_status emit_json(const std::string& js, char* buf, int64_t cap, int64_t* len) {
  if (len) *len = (int64_t)js.size();
  if (!buf || cap < (int64_t)js.size() + 1) return fail(MGF_ERR_CAPACITY, "JSON buffer too small (len reports the size needed, plus one for the terminator)");
  memcpy(buf, js.data(), js.size());
  buf[js.size()] = 0;
  return MGF_OK;
}
extern "C" mgf_status mgf_bvh_to_json(const mgf_bvh* b, char* buf, int64_t cap, int64_t* len) {
  if (!b) return fail(MGF_ERR_INVALID, "NULL argument");
  std::string js;
  sio::write_bvh(js, b->m.tree);
  return emit_json(js, buf, cap, len);
}
extern "C" mgf_status mgf_bvh_from_json(mgf_ctx* ctx, const char* json, int64_t len, mgf_bvh** out) {
  if (!json || len < 0 || !out) return fail(MGF_ERR_INVALID, "bad argument");
  sio::Parser P{json, json + len, {}};
  sio::Val v;
  if (!P.value(&v)) { set_error("JSON: %s at byte %lld", P.err.c_str(), (long long)(P.p - json)); return MGF_ERR_INVALID; }
  std::unique_ptr<mgf_bvh> b(new mgf_bvh{ctx, {}});
  std::string err;
  if (!sio::read_bvh(v, &b->m.tree, &err)) { set_error("%s", err.c_str()); return MGF_ERR_INVALID; }
  *out = b.release();
  return MGF_OK;
}
extern "C" mgf_status mgf_mesh_to_json(const mgf_mesh* m, char* buf, int64_t cap, int64_t* len) {
  if (!m) return fail(MGF_ERR_INVALID, "NULL argument");
  std::string js = "{\"x\":";
  sio::put_v3(js, m->x);
  js += ",\"verts\":[";
  for (size_t i = 0; i < m->verts.size(); ++i) { if (i) js += ','; sio::put_v3(js, m->verts[i]); }
  js += "],\"faces\":[";
  for (size_t i = 0; i + 2 < m->faces.size(); i += 3) {
    if (i) js += ',';
    js += '['; sio::put_u64(js, m->faces[i]); js += ','; sio::put_u64(js, m->faces[i + 1]); js += ','; sio::put_u64(js, m->faces[i + 2]); js += ']';
  }
  js += "],\"bvh\":";
  sio::write_bvh(js, m->m.tree);
  js += "}";
  return emit_json(js, buf, cap, len);
}
extern "C" mgf_status mgf_mesh_from_json(mgf_ctx* ctx, const char* json, int64_t len, mgf_mesh** out) {
  if (!json || len < 0 || !out) return fail(MGF_ERR_INVALID, "bad argument");
  sio::Parser P{json, json + len, {}};
  sio::Val v;
  if (!P.value(&v)) { set_error("JSON: %s at byte %lld", P.err.c_str(), (long long)(P.p - json)); return MGF_ERR_INVALID; }
  std::unique_ptr<mgf_mesh> m(new mgf_mesh());
  m->ctx = ctx;
  const sio::Val* verts = v.get("verts");
  const sio::Val* faces = v.get("faces");
  const sio::Val* bvh = v.get("bvh");
  if (v.kind != sio::Val::Obj || !sio::as_v3(v.get("x"), &m->x) || !verts || verts->kind != sio::Val::Arr || !faces || faces->kind != sio::Val::Arr || !bvh)
    return fail(MGF_ERR_INVALID, "Mesh: expected {x, verts, faces, bvh}");
  for (const sio::Val& e : verts->arr) { V3 p; if (!sio::as_v3(&e, &p)) return fail(MGF_ERR_INVALID, "Mesh: bad vertex"); m->verts.push_back(p); }
  for (const sio::Val& e : faces->arr) {
    uint64_t id[3];
    if (e.kind != sio::Val::Arr || e.arr.size() != 3 || !sio::as_u64(&e.arr[0], &id[0]) || !sio::as_u64(&e.arr[1], &id[1]) || !sio::as_u64(&e.arr[2], &id[2]))
      return fail(MGF_ERR_INVALID, "Mesh: bad face");
    for (int k = 0; k < 3; ++k) { if (id[k] >= m->verts.size()) return fail(MGF_ERR_INVALID, "Mesh: face index out of range"); m->faces.push_back((uint32_t)id[k]); }
  }
  std::string err;
  if (!sio::read_bvh(*bvh, &m->m.tree, &err)) { set_error("%s", err.c_str()); return MGF_ERR_INVALID; }
  // every leaf of the face tree must name a face
  for (uint64_t i = 0; i < m->m.tree.slots(); ++i)
    if (m->m.tree.used(i) && m->m.tree.node(i).leaf && m->m.tree.node(i).value >= m->faces.size() / 3)
      return fail(MGF_ERR_INVALID, "Mesh: BVH leaf names a face that does not exist");
  ++m->geom_version;
  *out = m.release();
  return MGF_OK;
}

// ---- mgf_compound (compound.rs:230-352) -------------------------------------------------------
static_assert(sizeof(CompIn) == sizeof(mgf_component), "component layout");
struct mgf_compound {
  mgf_ctx* ctx;
  std::vector<mgf_component> comps;
  TreeMirror m;            // BVH<AABB, Component>: leaf value = index into comps (the reference stores the Component itself)
  DBuf<CompIn> d_comps;
  bool comps_uploaded = false;
  V3 disp = mk3(0, 0, 0);
  Quat rot = mkq(1.0f, mk3(0, 0, 0));
  mgf_status sync() {
    MGF_TRY(m.sync(ctx));
    if (!comps_uploaded) {
      MGF_TRY(d_comps.ensure(std::max<size_t>(comps.size(), 1), ctx->stream));
      MGF_TRY(h2d(ctx, d_comps.p, reinterpret_cast<const CompIn*>(comps.data()), comps.size()));
      comps_uploaded = true;
    }
    return MGF_OK;
  }
  CompoundDev dev(uint32_t* err) const {
    CompoundDev D;
    D.tree = m.dev(nullptr, nullptr, mk3(0, 0, 0), err);
    D.comps = d_comps.p;
    D.disp[0] = disp.x; D.disp[1] = disp.y; D.disp[2] = disp.z;
    D.rot[0] = rot.s; D.rot[1] = rot.v.x; D.rot[2] = rot.v.y; D.rot[3] = rot.v.z;
    return D;
  }
};
// Compound::new compound.rs:244-257: components are inserted into the internal BVH in order
extern "C" mgf_status mgf_compound_new(mgf_ctx* ctx, const mgf_component* comps, int64_t n, mgf_compound** out) {
  if (!out || n < 0 || (n && !comps)) return fail(MGF_ERR_INVALID, "bad argument");
  std::unique_ptr<mgf_compound> c(new mgf_compound());
  c->ctx = ctx;
  for (int64_t i = 0; i < n; ++i) {
    if (comps[i].tag != MGF_SPHERE && comps[i].tag != MGF_CAPSULE) return fail(MGF_ERR_INVALID, "component tag must be sphere or capsule");
    if (!(comps[i].r >= 0.0f)) return fail(MGF_ERR_INVALID, "radius must be >= 0 (geom.rs:300,328)");
    c->comps.push_back(comps[i]);
    c->m.tree.insert(comp_bounds(comp_of(comps[i])), (uint64_t)i);
  }
  *out = c.release();
  return MGF_OK;
}
extern "C" void mgf_compound_free(mgf_compound* c) {
  if (!c) return;
  if (c->ctx) { (void)hipSetDevice(c->ctx->device); (void)hipStreamSynchronize(c->ctx->stream); }
  delete c;
}
extern "C" mgf_status mgf_compound_set_pose(mgf_compound* c, mgf_vec3 disp, mgf_quat rot) {  // pub fields disp, rot (:234-236)
  if (!c) return fail(MGF_ERR_INVALID, "compound is NULL");
  c->disp = mk3(disp.x, disp.y, disp.z);
  c->rot = mkq(rot.s, mk3(rot.x, rot.y, rot.z));
  return MGF_OK;
}
// contacts of n moving components against the compound (Contacts<RHS> for Compound :334-352, RHS = Moving<Sphere> /
// Moving<Capsule>); per rhs the contacts in the order the reference's callback sees them (CSR offsets).
extern "C" mgf_status mgf_compound_contacts_many(mgf_compound* c, const mgf_moving_component* rhs, int64_t n, uint64_t* out_offsets,
                                                 mgf_contact* out, int64_t cap, int64_t* total) {
  if (!c || n < 0 || (n && !rhs) || !out_offsets) return fail(MGF_ERR_INVALID, "bad argument");
  mgf_ctx* ctx = c->ctx;
  MGF_TRY(ctx_bind(ctx));
  for (int64_t i = 0; i <= n; ++i) out_offsets[i] = 0;
  if (total) *total = 0;
  if (n == 0 || c->comps.empty()) return MGF_OK;
  for (int64_t i = 0; i < n; ++i)
    if (rhs[i].shape.tag != MGF_SPHERE && rhs[i].shape.tag != MGF_CAPSULE) return fail(MGF_ERR_INVALID, "rhs must be a moving sphere or capsule");
  MGF_TRY(c->sync());
  DBuf<MovingIn> d_rhs; DBuf<uint32_t> d_cnt, d_off, d_err; DBuf<ContactOut> d_out;
  MGF_TRY(d_rhs.ensure((size_t)n, ctx->stream)); MGF_TRY(d_cnt.ensure((size_t)n + 1, ctx->stream)); MGF_TRY(d_off.ensure((size_t)n + 1, ctx->stream));
  MGF_TRY(d_err.ensure(1, ctx->stream));
  MGF_HIP_TRY(hipMemsetAsync(d_err.p, 0, 4, ctx->stream));
  MGF_TRY(h2d(ctx, d_rhs.p, reinterpret_cast<const MovingIn*>(rhs), (size_t)n));
  CompoundDev D = c->dev(d_err.p);
  k_compound_contacts<false><<<nblk(n), kBlock, 0, ctx->stream>>>(D, d_rhs.p, n, d_cnt.p, nullptr, nullptr);
  LAUNCH_CHECK();
  MGF_TRY(prim_exclusive_scan_u32(ctx, d_cnt.p, d_off.p, (size_t)n + 1));
  std::vector<uint32_t> off((size_t)n + 1);
  MGF_TRY(d2h(ctx, off.data(), d_off.p, (size_t)n + 1));
  for (int64_t i = 0; i <= n; ++i) out_offsets[i] = off[(size_t)i];
  uint32_t tot = off[(size_t)n];
  if (total) *total = tot;
  if ((int64_t)tot > cap) return fail(MGF_ERR_CAPACITY, "contact buffer too small");
  if (tot) {
    if (!out) return fail(MGF_ERR_INVALID, "NULL contact buffer");
    MGF_TRY(d_out.ensure(tot, ctx->stream));
    k_compound_contacts<true><<<nblk(n), kBlock, 0, ctx->stream>>>(D, d_rhs.p, n, nullptr, d_off.p, d_out.p);
    LAUNCH_CHECK();
    MGF_TRY(d2h(ctx, reinterpret_cast<ContactOut*>(out), d_out.p, tot));
  }
  uint32_t err = 0;
  MGF_TRY(d2h(ctx, &err, d_err.p, 1));
  if (err) return fail(MGF_ERR_CAPACITY, "BVH traversal stack overflow");
  return MGF_OK;
}
// Intersects<Compound> for n particles (:309-332)
extern "C" mgf_status mgf_compound_intersections(mgf_compound* c, const mgf_particle* parts, int64_t n, mgf_intersection* out, int32_t* hit) {
  if (!c || n < 0 || (n && (!parts || !out || !hit))) return fail(MGF_ERR_INVALID, "bad argument");
  mgf_ctx* ctx = c->ctx;
  MGF_TRY(ctx_bind(ctx));
  if (n == 0) return MGF_OK;
  if (c->comps.empty()) { for (int64_t i = 0; i < n; ++i) hit[i] = 0; return MGF_OK; }
  MGF_TRY(c->sync());
  DBuf<ParticleIn> dp; DBuf<InterOut> dout; DBuf<int32_t> dh; DBuf<uint32_t> d_err;
  MGF_TRY(dp.ensure((size_t)n, ctx->stream)); MGF_TRY(dout.ensure((size_t)n, ctx->stream)); MGF_TRY(dh.ensure((size_t)n, ctx->stream));
  MGF_TRY(d_err.ensure(1, ctx->stream));
  MGF_HIP_TRY(hipMemsetAsync(d_err.p, 0, 4, ctx->stream));
  MGF_TRY(h2d(ctx, dp.p, reinterpret_cast<const ParticleIn*>(parts), (size_t)n));
  k_compound_intersections<<<nblk(n), kBlock, 0, ctx->stream>>>(c->dev(d_err.p), dp.p, n, dout.p, dh.p);
  LAUNCH_CHECK();
  MGF_TRY(d2h(ctx, reinterpret_cast<InterOut*>(out), dout.p, (size_t)n));
  MGF_TRY(d2h(ctx, hit, dh.p, (size_t)n));
  uint32_t err = 0;
  MGF_TRY(d2h(ctx, &err, d_err.p, 1));
  if (err) return fail(MGF_ERR_CAPACITY, "BVH traversal stack overflow");
  return MGF_OK;
}
// BoundedBy<AABB> for Compound :274-278 (host arithmetic, same functions as the device code)
extern "C" mgf_status mgf_compound_bounds(const mgf_compound* c, mgf_aabb* out) {
  if (!c || !out) return fail(MGF_ERR_INVALID, "NULL argument");
  if (c->m.tree.empty()) return fail(MGF_ERR_EMPTY, "BVH is empty, there is no root node");
  Box b = c->m.tree.node(c->m.tree.root()).box;
  b = box_rotate(b, c->rot);
  b.c = b.c + c->disp;
  *out = from_box(b);
  return MGF_OK;
}

// ---- mgf_mesh --------------------------------------------------------------------------------
extern "C" mgf_status mgf_mesh_new(mgf_ctx* ctx, mgf_mesh** out) {  // ctx may be NULL: host-only mesh
  if (!out) return fail(MGF_ERR_INVALID, "NULL argument");
  mgf_mesh* m = new mgf_mesh();
  m->ctx = ctx;
  *out = m;
  return MGF_OK;
}
extern "C" void mgf_mesh_free(mgf_mesh* m) {
  if (!m) return;
  if (m->ctx) (void)hipSetDevice(m->ctx->device);
  delete m;
}
extern "C" mgf_status mgf_mesh_push_vert(mgf_mesh* m, mgf_vec3 p, uint64_t* id) {
  if (!m) return fail(MGF_ERR_INVALID, "mesh is NULL");
  if (id) *id = m->verts.size();
  m->verts.push_back(mk3(p.x, p.y, p.z));
  ++m->geom_version;
  return MGF_OK;
}
extern "C" mgf_status mgf_mesh_push_face(mgf_mesh* m, uint64_t a, uint64_t b, uint64_t c, uint64_t* id) {
  if (!m) return fail(MGF_ERR_INVALID, "mesh is NULL");
  size_t nv = m->verts.size();
  if (a >= nv || b >= nv || c >= nv) return fail(MGF_ERR_INVALID, "vertex index out of bounds");
  uint64_t index = m->faces.size() / 3;
  Triangle tri = mkt(m->verts[a], m->verts[b], m->verts[c]);
  m->faces.push_back((uint32_t)a); m->faces.push_back((uint32_t)b); m->faces.push_back((uint32_t)c);
  m->m.tree.insert(tri_bounds(tri), index);  // mesh.rs:71
  ++m->geom_version;
  if (id) *id = index;
  return MGF_OK;
}
extern "C" mgf_status mgf_mesh_set_pos(mgf_mesh* m, mgf_vec3 p) {
  if (!m) return fail(MGF_ERR_INVALID, "mesh is NULL");
  V3 disp = mk3(p.x, p.y, p.z) - m->x;  // geom.rs:459-462 with center() = x (mesh.rs:89)
  m->x = m->x + disp;                   // AddAssign mesh.rs:76-80
  return MGF_OK;
}
extern "C" mgf_status mgf_mesh_build(mgf_mesh* m, const mgf_vec3* verts, int64_t nverts, const uint32_t* faces, int64_t nfaces) {
  if (!m || (!verts && nverts) || (!faces && nfaces)) return fail(MGF_ERR_INVALID, "NULL argument");
  for (int64_t i = 0; i < nverts; ++i) MGF_TRY(mgf_mesh_push_vert(m, verts[i], nullptr));
  for (int64_t i = 0; i < nfaces; ++i) MGF_TRY(mgf_mesh_push_face(m, faces[3 * i], faces[3 * i + 1], faces[3 * i + 2], nullptr));
  return MGF_OK;
}
extern "C" MGF_API mgf_bvh* mgf_mesh_bvh_view(mgf_mesh* m) {  // debug: not owned by the caller
  static thread_local mgf_bvh view;
  view.ctx = m->ctx;
  view.m.tree = m->m.tree;
  return &view;
}

// ---------------------------------------------------------------------------------------------
// single-shot narrowphase
// ---------------------------------------------------------------------------------------------
static_assert(sizeof(ShapeIn) == sizeof(mgf_shape), "shape layout");
static_assert(sizeof(ContactOut) == sizeof(mgf_contact), "contact layout");
static_assert(sizeof(LocalOut) == sizeof(mgf_local_contact), "local contact layout");
static_assert(sizeof(MovingIn) == sizeof(mgf_moving_component), "moving component layout");
static_assert(sizeof(CRec) == 128, "constraint record layout");

extern "C" mgf_status mgf_contacts_batch(mgf_ctx* ctx, int64_t n, const mgf_shape* a, const mgf_vec3* vel_a, const mgf_shape* b,
                                         const mgf_vec3* vel_b, const uint8_t* has_vel, mgf_contact* out, int32_t* counts) {
  MGF_TRY(ctx_bind(ctx));
  if (n < 0 || (n && (!a || !b || !has_vel || !out || !counts))) return fail(MGF_ERR_INVALID, "NULL argument");
  if (n == 0) return MGF_OK;
  for (int64_t i = 0; i < n; ++i)
    if (a[i].kind == MGF_RECTANGLE || b[i].kind == MGF_RECTANGLE || a[i].kind < 0 || a[i].kind > 4 || b[i].kind < 0 || b[i].kind > 4)
      return fail(MGF_ERR_INVALID, "shape kind not on the hot path");
  DBuf<ShapeIn> da, db;
  DBuf<float> dva, dvb;
  DBuf<uint8_t> dh;
  DBuf<ContactOut> dout;
  DBuf<int32_t> dc;
  size_t N = (size_t)n;
  MGF_TRY(da.ensure(N, ctx->stream)); MGF_TRY(db.ensure(N, ctx->stream));
  MGF_TRY(dva.ensure(3 * N, ctx->stream)); MGF_TRY(dvb.ensure(3 * N, ctx->stream));
  MGF_TRY(dh.ensure(N, ctx->stream)); MGF_TRY(dout.ensure(2 * N, ctx->stream)); MGF_TRY(dc.ensure(N, ctx->stream));
  std::vector<float> zeros(3 * N, 0.0f);
  MGF_TRY(h2d(ctx, da.p, reinterpret_cast<const ShapeIn*>(a), N));
  MGF_TRY(h2d(ctx, db.p, reinterpret_cast<const ShapeIn*>(b), N));
  MGF_TRY(h2d(ctx, dva.p, vel_a ? reinterpret_cast<const float*>(vel_a) : zeros.data(), 3 * N));
  MGF_TRY(h2d(ctx, dvb.p, vel_b ? reinterpret_cast<const float*>(vel_b) : zeros.data(), 3 * N));
  MGF_TRY(h2d(ctx, dh.p, has_vel, N));
  k_contacts_batch<<<(unsigned)((N + 63) / 64), 64, 0, ctx->stream>>>(n, da.p, dva.p, db.p, dvb.p, dh.p, dout.p, dc.p);
  LAUNCH_CHECK();
  MGF_TRY(d2h(ctx, reinterpret_cast<ContactOut*>(out), dout.p, 2 * N));
  MGF_TRY(d2h(ctx, counts, dc.p, N));
  for (int64_t i = 0; i < n; ++i)
    if (counts[i] < 0) return fail(MGF_ERR_INVALID, "unsupported shape pair");
  return MGF_OK;
}
extern "C" mgf_status mgf_contacts(mgf_ctx* ctx, const mgf_shape* a, const mgf_vec3* vel_a, const mgf_shape* b, const mgf_vec3* vel_b,
                                   mgf_contact* out, int32_t cap, int32_t* count) {
  if (!a || !b || !count) return fail(MGF_ERR_INVALID, "NULL argument");
  uint8_t hv = (vel_a ? 1 : 0) | (vel_b ? 2 : 0);
  mgf_contact tmp[2];
  int32_t n = 0;
  mgf_vec3 z = {0, 0, 0};
  MGF_TRY(mgf_contacts_batch(ctx, 1, a, vel_a ? vel_a : &z, b, vel_b ? vel_b : &z, &hv, tmp, &n));
  *count = n;
  for (int k = 0; k < n && k < cap; ++k) out[k] = tmp[k];
  if (n > cap) return fail(MGF_ERR_CAPACITY, "contact buffer too small");
  return MGF_OK;
}
extern "C" mgf_status mgf_local_contacts_pair(mgf_ctx* ctx, const mgf_moving_component* a, const mgf_moving_component* b,
                                              mgf_local_contact* out, int32_t cap, int32_t* count) {
  MGF_TRY(ctx_bind(ctx));
  if (!a || !b || !count) return fail(MGF_ERR_INVALID, "NULL argument");
  DBuf<LocalOut> dout;
  DBuf<int32_t> dc;
  MGF_TRY(dout.ensure(1, ctx->stream)); MGF_TRY(dc.ensure(1, ctx->stream));
  MovingIn ma, mb;
  memcpy(&ma, a, sizeof(ma)); memcpy(&mb, b, sizeof(mb));
  k_local_pair<<<1, 1, 0, ctx->stream>>>(ma, mb, dout.p, dc.p);
  LAUNCH_CHECK();
  int32_t n = 0;
  MGF_TRY(d2h(ctx, &n, dc.p, 1));
  *count = n;
  if (n > cap) return fail(MGF_ERR_CAPACITY, "contact buffer too small");
  if (n) MGF_TRY(d2h(ctx, reinterpret_cast<LocalOut*>(out), dout.p, 1));
  return MGF_OK;
}
extern "C" mgf_status mgf_local_contacts_mesh(mgf_ctx* ctx, const mgf_moving_component* body, const mgf_mesh* mesh_c,
                                              mgf_local_contact* out, int32_t cap, int32_t* count) {
  MGF_TRY(ctx_bind(ctx));
  if (!body || !mesh_c || !count) return fail(MGF_ERR_INVALID, "NULL argument");
  mgf_mesh* mesh = const_cast<mgf_mesh*>(mesh_c);
  *count = 0;
  if (mesh->m.tree.empty()) return MGF_OK;
  MGF_TRY(mesh->sync());
  const int32_t dcap = 64;
  DBuf<LocalOut> dout;
  DBuf<int32_t> dc;
  DBuf<uint32_t> derr;
  MGF_TRY(dout.ensure(dcap, ctx->stream)); MGF_TRY(dc.ensure(1, ctx->stream)); MGF_TRY(derr.ensure(1, ctx->stream));
  MGF_HIP_TRY(hipMemsetAsync(derr.p, 0, 4, ctx->stream));
  MovingIn ma;
  memcpy(&ma, body, sizeof(ma));
  k_local_mesh<<<1, 1, 0, ctx->stream>>>(ma, mesh->dev(derr.p), dout.p, dcap, dc.p);
  LAUNCH_CHECK();
  int32_t n = 0;
  MGF_TRY(d2h(ctx, &n, dc.p, 1));
  *count = n;
  if (n > cap || n > dcap) return fail(MGF_ERR_CAPACITY, "contact buffer too small");
  if (n) MGF_TRY(d2h(ctx, reinterpret_cast<LocalOut*>(out), dout.p, (size_t)n));
  return MGF_OK;
}
extern "C" mgf_status mgf_ray_capsule(mgf_ctx* ctx, const mgf_vec3* p, const mgf_vec3* d, const mgf_shape* cap, mgf_vec3* ip, float* t,
                                      int32_t* hit) {
  MGF_TRY(ctx_bind(ctx));
  if (!p || !d || !cap || !hit || cap->kind != MGF_CAPSULE) return fail(MGF_ERR_INVALID, "bad argument");
  DBuf<float> dout;
  DBuf<int32_t> dh;
  MGF_TRY(dout.ensure(4, ctx->stream)); MGF_TRY(dh.ensure(1, ctx->stream));
  Capsule c = mkcap(mk3(cap->v[0], cap->v[1], cap->v[2]), mk3(cap->v[3], cap->v[4], cap->v[5]), cap->v[6]);
  k_ray_capsule<<<1, 1, 0, ctx->stream>>>(mk3(p->x, p->y, p->z), mk3(d->x, d->y, d->z), c, dout.p, dh.p);
  LAUNCH_CHECK();
  float o[4];
  MGF_TRY(d2h(ctx, hit, dh.p, 1));
  if (*hit) {
    MGF_TRY(d2h(ctx, o, dout.p, 4));
    if (ip) { ip->x = o[0]; ip->y = o[1]; ip->z = o[2]; }
    if (t) *t = o[3];
  }
  return MGF_OK;
}

// Inertia::tensor physics.rs:30-93 (setup-time, host)
static M3 tensor_of(const Comp& k, float m) {
  V3 disp;
  M3 i;
  if (k.kind == KIND_SPHERE) {
    float s = 0.4f * m * k.r * k.r;
    i = m3_diag(s, s, s);
    disp = k.p;
  } else {
    float h = mag(k.d), r = k.r;
    float mh = m * 2.0f * r / (4.0f * r + 3.0f * h);
    float mc = m * h / (4.0f / 3.0f * r + h);
    float ic_x = 1.0f / 12.0f * mc * (3.0f * r * r + h * h);
    float ic_y = 0.5f * mc * r * r;
    float is_x = mh * (3.0f * r + 2.0f * h) / 4.0f * h;
    float is_y = 4.0f / 5.0f * mh * r * r;
    float i_x = ic_x + is_x, i_y = ic_y + is_y, i_z = ic_x + is_x;
    M3 rot = m3_from_quat(quat_from_arc(mk3(0.0f, 1.0f, 0.0f) * h, k.d));
    i = rot * m3_diag(i_x, i_y, i_z) * transpose(rot);
    disp = comp_center(k);
  }
  M3 outer = m3_cols(disp * disp.x, disp * disp.y, disp * disp.z);
  return i + m * (m3_diag(1.0f, 1.0f, 1.0f) * dot(disp, disp) - outer);
}
extern "C" mgf_status mgf_inertia_tensor(const mgf_component* c, float mass, float out9[9]) {
  if (!c || !out9 || (c->tag != MGF_SPHERE && c->tag != MGF_CAPSULE)) return fail(MGF_ERR_INVALID, "bad component");
  M3 t = tensor_of(comp_of(*c), mass);
  for (int k = 0; k < 3; ++k) { out9[3 * k] = t.c[k].x; out9[3 * k + 1] = t.c[k].y; out9[3 * k + 2] = t.c[k].z; }
  return MGF_OK;
}

// ---------------------------------------------------------------------------------------------
// World
// ---------------------------------------------------------------------------------------------
struct mgf_world {
  mgf_ctx* ctx = nullptr;
  mgf_params params;
  uint32_t n = 0;        // local bodies = owned + ghosts of the current tick
  uint32_t n_owned = 0;  // bodies of this world's RigidBodyVec (added through add_bodies)
  bool has_sphere = false, has_capsule = false;
  bool has_compound = false;      // a body of several parts exists: the *_parts narrowphase kernels serve every pair
  DBuf<uint32_t> pcount;          // parts per body (0: an ordinary body); the four part arrays hold kMaxParts slots per body
  DBuf<float4> lp0, lp1, wp0, wp1;
  DBuf<uint32_t> bflag_l, bflag_r, bscan_l, bscan_r;  // boundary selection scratch
  DBuf<uint32_t> mig_cnt;  // migration: [0] left-goers, [1] right-goers (also remove_bodies' error word)
  DBuf<float4> mig_tmp;    // remove_bodies: compacted copy of every body array
  // RigidBodyVec
  DBuf<float4> x, q, srec, sp0, sp1, ctor, imb, delta, einfo, col0, col1, tb_c, tb_r, fb_c, fb_r;
  // terrain (copy of the caller's Mesh)
  std::unique_ptr<mgf_mesh> terrain;
  // broadphase
  DBuf<uint32_t> cell_of, cell_rank, sidx;
  DBuf<QNode> lnodes;
  DBuf<LeafRec> leaves;
  DBuf<float4> lcol;             // cell-ordered (collider, motion) copies for the fused sphere test of k_pair_grid
  DBuf<uint32_t> pair_stat;      // 64 partial sums of the partners accepted by the fused broadphase
  bool tick_fused = false;
  int64_t opt_no_fused_narrowphase = 0;
  DBuf<float4> sub_lo, sub_hi, sub2_lo, sub2_hi;
  DBuf<uint32_t> cell_lo, cell_cnt;
  DBuf<uint32_t> t_cnt, p_cnt, t_off, p_off, t_cand, t_owner, p_cand, p_owner, rows, rows_t;
  // 5 (default) = block-local dataflow launch (k_solve_flow5: a spatial block's velocities, counters and ready queues in
  //     LDS), with k_solve_flow as its stand-by when a block does not fit;
  // 1 = one persistent dataflow launch per Solver::solve, everything through L2 (k_solve_flow);
  // 0 = one launch per frontier of the dependency graph (k_solve, the independent cross-check);
  // 4 = dataflow launch with out-of-order slots per lane (k_solve_flowk)
  int64_t opt_solver_mode = 5;
  DBuf<uint32_t> flow_arr, flow_arr5;
  DBuf<uint64_t> flow_trace;
  // block-local solver (mode 5)
  DBuf<uint32_t> brank, f5_shared, f5_gcnt, f5_lslot, f5_wg_cnt;
  DBuf<F5Row> f5_table;
  bool flow5_ok = false, flow5_prepped = false;  // the constraint list came from collide (own-block ranges valid) / prep done
  uint32_t f5_nb = 0, f5_nblocks = 0;
  int flow_grid = 0;
  // device-resident list sizes + speculative capacities (see StepCounts)
  DBuf<StepCounts> sc;
  uint32_t cap_t = 0, cap_p = 0, cap_c = 0;
  uint64_t n_cap_retries = 0, n_flow5_fallbacks = 0;
  bool tick_two_pass = false;
  uint32_t row_cap_t = kRowCapT;    // terrain faces per body in the row path (grows on overflow, sticky)
  bool terrain_grid_off = false;    // sticky: the mesh's faces span too many cells for the face grid (k_terrain_grid)
  bool grid_too_wide = false;       // sticky: the largest body spans too many Morton cells for the grid broadphase
  int64_t opt_terrain_tree = 0;     // 1 = always walk the mesh BVH (k_terrain_rows) instead of the face grid
  int64_t opt_broadphase_tree = 0;  // 1 = always walk the tree (k_pair_rows) instead of enumerating grid cells
  int64_t opt_flow_blocks_per_cu = 0, opt_flow_sleep = 2;
  int flowk_grid = 0;
  int64_t opt_debug_bvh = 0, opt_flow_trace = 0;
  int64_t opt_flow5_slow_x2 = 3;
  int64_t opt_flow5_poller = -1;   // 1: one slow wave only polls the outside-arrival counters; -1: with the narrow layout only (measured)
  int64_t opt_flow5_block = 0;     // minimum bodies per block of the block-local solver (tests)
  bool flow5_attr_set = false;
  bool flow5_wide = false;          // LDS layout of k_solve_flow5 for this tick (chosen from the largest block of the last one)
  uint32_t flow5_last_max = 0;
  int64_t opt_stream_ordered = 0;  // 1: the tiling calls (begin_tick, export_*, import_*) do not synchronise the ctx stream
  bool solve_pending = false;      // a dataflow launch was enqueued by mgf_world_solve_enqueue and not yet checked
  DBuf<unsigned long long> dbg;
  int64_t opt_two_pass = 0;  // 1 = always use the exact two-pass candidate path (tests the overflow fallback)
  uint64_t n_row_overflows = 0;
  // narrowphase
  DBuf<uint32_t> t_nc, p_nc, t_pre, p_pre, cnt, base, work_lists, work_counts;
  DBuf<NContact> t_out, p_out;
  // solver
  DBuf<CRec> cons_nat;
  DBuf<uint2> c_ab, c_succ;  // compact dependency links (ConsLinks)
  DBuf<uint8_t> c_pred;
  DBuf<uint32_t> deg, adj_off, adj_fill, adj_list, order, lvl_off;
  DBuf<uint32_t> degb, rev;      // per body: count and row (rev_cap ids) of the constraints it takes part in as `b`
  uint32_t rev_cap = 16;          // grows on overflow (kFailRevRow), sticky
  DBuf<uint32_t> scalars;  // [0..2] rotating level counters, [3] err
  DBuf<SceneBounds> sb;
  uint32_t Mt = 0, Mp = 0, C = 0, Ct = 0, depth = 0, last_launches = 0, lvl_cap = 0;
  size_t order_cap_iters = 0;
  bool constraints_ready = false;
  float last_dt = 0.0f;
  int64_t opt_time_solver_kernels = 0;
  hipEvent_t ev[8] = {};
  std::vector<hipEvent_t> kev;  // per-launch events (option)
  size_t kev_used = 0;
  mgf_step_stats stats;

  Bodies bodies() {
    Bodies B;
    B.x = x.p; B.q = q.p; B.srec = srec.p; B.sp0 = sp0.p; B.sp1 = sp1.p; B.ctor = ctor.p; B.imb = imb.p; B.delta = delta.p;
    B.einfo = einfo.p; B.col0 = col0.p; B.col1 = col1.p; B.tb_c = tb_c.p; B.tb_r = tb_r.p; B.fb_c = fb_c.p; B.fb_r = fb_r.p;
    B.pcount = has_compound ? pcount.p : nullptr; B.lp0 = lp0.p; B.lp1 = lp1.p; B.wp0 = wp0.p; B.wp1 = wp1.p;
    return B;
  }
  uint32_t* d_cnt() { return scalars.p; }
  uint32_t* d_err() { return scalars.p + 3; }
  Flow5 flow5() {
    Flow5 F;
    F.sidx = sidx.p; F.brank = brank.p; F.shared = reinterpret_cast<uint8_t*>(f5_shared.p);
    F.gcnt = f5_gcnt.p; F.arr5 = flow_arr5.p; F.lslot = f5_lslot.p; F.wg_cnt = f5_wg_cnt.p;
    F.table = f5_table.p;
    F.fail = d_err() + 4; F.max_block = d_err() + 6;
    F.cap_fast = flow5_wide ? kF5MaxFast : kF5NarrowCons; F.cap_slow = flow5_wide ? kF5MaxSlow : kF5NarrowCons;
    F.cap_all = flow5_wide ? kF5MaxCons : kF5NarrowCons;
    F.slow_x2 = (uint32_t)opt_flow5_slow_x2;
    F.poller = opt_flow5_poller < 0 ? (flow5_wide ? 0u : 1u) : (uint32_t)opt_flow5_poller;
    F.nb = f5_nb; F.nblocks = f5_nblocks; F.n = n;
    return F;
  }
  ConsLinks links() { ConsLinks K; K.ab = c_ab.p; K.succ = c_succ.p; K.pred = c_pred.p; return K; }
  Frontier frontier() { Frontier F; F.order = order.p; F.lvl_off = lvl_off.p; F.cnt = d_cnt(); return F; }
};

extern "C" mgf_status mgf_world_new(mgf_ctx* ctx, const mgf_params* params, mgf_world** out) {
  MGF_TRY(ctx_bind(ctx));
  if (!out) return fail(MGF_ERR_INVALID, "out is NULL");
  std::unique_ptr<mgf_world> w(new mgf_world());
  w->ctx = ctx;
  w->params = params ? *params : mgf_default_params();
  memset(&w->stats, 0, sizeof(w->stats));
  MGF_TRY(w->scalars.ensure(16, ctx->stream));
  MGF_TRY(w->sb.ensure(1, ctx->stream));
  MGF_TRY(w->sc.ensure(1, ctx->stream));
  MGF_HIP_TRY(hipMemsetAsync(w->sc.p, 0, sizeof(StepCounts), ctx->stream));
  MGF_HIP_TRY(hipMemsetAsync(w->scalars.p, 0, 64, ctx->stream));
  for (auto& e : w->ev) MGF_HIP_TRY(hipEventCreate(&e));
  *out = w.release();
  return MGF_OK;
}
extern "C" void mgf_world_free(mgf_world* w) {
  if (!w) return;
  (void)hipSetDevice(w->ctx->device);
  (void)hipStreamSynchronize(w->ctx->stream);
  for (auto& e : w->ev) if (e) (void)hipEventDestroy(e);
  for (auto& e : w->kev) (void)hipEventDestroy(e);
  delete w;
}
extern "C" int64_t mgf_world_len(const mgf_world* w) { return w ? (int64_t)w->n_owned : 0; }

extern "C" mgf_status mgf_world_set_option(mgf_world* w, const char* key, int64_t value) {
  if (!w || !key) return fail(MGF_ERR_INVALID, "NULL argument");
  if (!strcmp(key, "time_solver_kernels")) { w->opt_time_solver_kernels = value; return MGF_OK; }
  if (!strcmp(key, "two_pass_candidates")) { w->opt_two_pass = value; return MGF_OK; }
  if (!strcmp(key, "broadphase_tree")) { w->opt_broadphase_tree = value; return MGF_OK; }
  if (!strcmp(key, "terrain_tree")) { w->opt_terrain_tree = value; return MGF_OK; }
  if (!strcmp(key, "flow_trace")) { w->opt_flow_trace = value; return MGF_OK; }
  if (!strcmp(key, "debug_bvh")) { w->opt_debug_bvh = value; return MGF_OK; }
  if (!strcmp(key, "solver_mode")) { w->opt_solver_mode = value; return MGF_OK; }
  if (!strcmp(key, "flow_blocks_per_cu")) { w->opt_flow_blocks_per_cu = value; w->flow_grid = 0; w->flowk_grid = 0; return MGF_OK; }
  if (!strcmp(key, "flow_sleep")) { w->opt_flow_sleep = value; return MGF_OK; }
  if (!strcmp(key, "flow5_poller")) { w->opt_flow5_poller = value < 0 ? -1 : (value ? 1 : 0); return MGF_OK; }
  if (!strcmp(key, "flow5_slow_x2")) { if (value < 1 || value > 16) return fail(MGF_ERR_INVALID, "flow5_slow_x2 out of range"); w->opt_flow5_slow_x2 = value; return MGF_OK; }
  if (!strcmp(key, "flow5_block")) { w->opt_flow5_block = value; w->flow5_prepped = false; return MGF_OK; }
  if (!strcmp(key, "no_fused_narrowphase")) { w->opt_no_fused_narrowphase = value; return MGF_OK; }
  if (!strcmp(key, "stream_ordered")) { w->opt_stream_ordered = value; return MGF_OK; }
  if (!strcmp(key, "body_kinds")) {  // OR-in: kinds (bit0 sphere, bit1 capsule) that ghosts of this world may have
    if (value & 1) w->has_sphere = true;
    if (value & 2) w->has_capsule = true;
    return MGF_OK;
  }
  if (!strcmp(key, "list_capacity")) {  // tests: force the speculative list capacities (the next tick must re-run its collide phase)
    if (value < 1 || value > 0x7FFFFFF0ll) return fail(MGF_ERR_INVALID, "list_capacity out of range");
    w->cap_t = w->cap_p = w->cap_c = (uint32_t)value;
    return MGF_OK;
  }
  return fail(MGF_ERR_INVALID, "unknown option");
}

extern "C" mgf_status mgf_world_counter(const mgf_world* w, const char* name, int64_t* out) {
  if (!w || !name || !out) return fail(MGF_ERR_INVALID, "NULL argument");
  if (!strcmp(name, "row_overflows")) { *out = (int64_t)w->n_row_overflows; return MGF_OK; }
  if (!strcmp(name, "capacity_retries")) { *out = (int64_t)w->n_cap_retries; return MGF_OK; }
  if (!strcmp(name, "flow5_fallbacks")) { *out = (int64_t)w->n_flow5_fallbacks; return MGF_OK; }
  if (!strcmp(name, "grid_too_wide")) { *out = w->grid_too_wide ? 1 : 0; return MGF_OK; }
  if (!strcmp(name, "terrain_grid")) { *out = (w->terrain && w->terrain->grid.ready && !w->terrain_grid_off && !w->opt_terrain_tree) ? 1 : 0; return MGF_OK; }
  if (!strcmp(name, "terrain_row_capacity")) { *out = (int64_t)w->row_cap_t; return MGF_OK; }
  if (!strcmp(name, "rev_row_capacity")) { *out = (int64_t)w->rev_cap; return MGF_OK; }
  if (!strcmp(name, "body_kinds")) { *out = (w->has_sphere ? 1 : 0) | (w->has_capsule ? 2 : 0); return MGF_OK; }
  if (!strcmp(name, "flow5_blocks")) { *out = (int64_t)w->f5_nblocks; return MGF_OK; }
  if (!strncmp(name, "flow5_class", 11) && (name[11] == '0' || name[11] == '1' || name[11] == '2') && !name[12]) {
    // constraints of the last prepared tick in class 0 / 1 / 2 (all-LDS / global counter / LDS counter + shared body)
    *out = 0;
    if (!w->flow5_prepped || w->f5_nblocks == 0) return MGF_OK;
    std::vector<uint32_t> h(4 * (size_t)w->f5_nblocks * kF5CntStride);
    mgf_world* mw = const_cast<mgf_world*>(w);
    MGF_TRY(ctx_bind(mw->ctx));
    MGF_TRY(d2h(mw->ctx, h.data(), mw->f5_wg_cnt.p, h.size()));
    int k = name[11] - '0';
    for (uint32_t g = 0; g < w->f5_nblocks; ++g) *out += (int64_t)h[(size_t)(4 * g + k) * kF5CntStride];
    return MGF_OK;
  }
  if (!strcmp(name, "flow5_max_block") || !strcmp(name, "flow5_max_fast")) {  // largest block (all classes / class 0) of the last prepared tick
    *out = 0;
    if (!w->flow5_prepped || w->f5_nblocks == 0) return MGF_OK;
    std::vector<uint32_t> h(4 * (size_t)w->f5_nblocks * kF5CntStride);
    mgf_world* mw = const_cast<mgf_world*>(w);
    MGF_TRY(ctx_bind(mw->ctx));
    MGF_TRY(d2h(mw->ctx, h.data(), mw->f5_wg_cnt.p, h.size()));
    const bool fast = name[10] == 'f';
    for (uint32_t g = 0; g < w->f5_nblocks; ++g) {
      int64_t v = h[(size_t)(4 * g) * kF5CntStride];
      if (!fast) v += (int64_t)h[(size_t)(4 * g + 1) * kF5CntStride] + h[(size_t)(4 * g + 2) * kF5CntStride];
      *out = std::max(*out, v);
    }
    return MGF_OK;
  }
  return fail(MGF_ERR_INVALID, "unknown counter");
}

// The static mesh's face grid: face boxes = the leaf bounds of the reference tree, counting-sorted into Morton cells with
// the same kernels as the bodies; DFS ranks and parent links from the host tree.
static mgf_status build_face_grid(mgf_mesh* t) {
  mgf_ctx* ctx = t->ctx;
  hipStream_t s = ctx->stream;
  mgf_mesh::Grid& G = t->grid;
  G.ready = false;
  const size_t nf = t->faces.size() / 3;
  if (nf < 64 || t->m.tree.empty()) return MGF_OK;  // tiny meshes: the tree walk is cheap
  std::vector<uint32_t> rank_of, face_of, leaf_of, parent;
  t->m.tree.dfs_ranks(nf, &rank_of, &face_of, &leaf_of, &parent);
  if (face_of.size() != nf) return MGF_OK;  // a face without a leaf (or two): not a Mesh::push_face tree, keep the walk
  std::vector<float4> bc(nf), br(nf);
  for (size_t f = 0; f < nf; ++f) {
    const Box& b = t->m.tree.node(leaf_of[f]).box;
    bc[f] = make_float4(b.c.x, b.c.y, b.c.z, 0.0f); br[f] = make_float4(b.r.x, b.r.y, b.r.z, 0.0f);
  }
  uint32_t levels = 4;
  while (((uint64_t)1 << (2 * levels)) < nf && levels < (uint32_t)kMortonBits / 2) ++levels;
  const uint32_t cells = 1u << (2 * levels);
  G.levels = levels; G.n_faces = (uint32_t)nf;
  MGF_TRY(G.fb_c.ensure(nf, s)); MGF_TRY(G.fb_r.ensure(nf, s)); MGF_TRY(G.cell_of.ensure(nf, s)); MGF_TRY(G.cell_rank.ensure(nf, s));
  MGF_TRY(G.cell_cnt.ensure((size_t)cells + 1, s)); MGF_TRY(G.cell_lo.ensure((size_t)cells + 1, s)); MGF_TRY(G.sidx.ensure(nf, s));
  MGF_TRY(G.brank.ensure(nf, s)); MGF_TRY(G.leaves.ensure(nf, s)); MGF_TRY(G.sb.ensure(1, s));
  MGF_TRY(G.rank_of_face.ensure(nf, s)); MGF_TRY(G.face_of_rank.ensure(nf, s)); MGF_TRY(G.leaf_of_face.ensure(nf, s));
  MGF_TRY(G.parent.ensure(std::max<size_t>(parent.size(), 1), s));
  MGF_TRY(h2d(ctx, G.fb_c.p, bc.data(), nf)); MGF_TRY(h2d(ctx, G.fb_r.p, br.data(), nf));
  MGF_TRY(h2d(ctx, G.rank_of_face.p, rank_of.data(), nf)); MGF_TRY(h2d(ctx, G.face_of_rank.p, face_of.data(), nf));
  MGF_TRY(h2d(ctx, G.leaf_of_face.p, leaf_of.data(), nf)); MGF_TRY(h2d(ctx, G.parent.p, parent.data(), parent.size()));
  MGF_HIP_TRY(hipMemsetAsync(G.cell_cnt.p, 0, ((size_t)cells + 1) * 4, s));
  SceneBounds sb0;
  for (int k = 0; k < 3; ++k) { sb0.lo[k] = 0x7FFFFFFF; sb0.hi[k] = (int)0x80000000; sb0.rmax[k] = 0; }
  sb0.n_refits = 0; sb0.pad = 0; sb0.pad2 = 0;
  MGF_TRY(h2d(ctx, G.sb.p, &sb0, 1));
  k_scene_bounds<<<std::min<unsigned>(nblk(nf), 256u), kBlock, 0, s>>>(G.fb_c.p, G.fb_r.p, (uint32_t)nf, G.sb.p);
  LAUNCH_CHECK();
  k_morton_count<<<nblk(nf), kBlock, 0, s>>>(G.fb_c.p, (uint32_t)nf, G.sb.p, kMortonBits - 2 * (int)levels, G.cell_of.p, G.cell_rank.p, G.cell_cnt.p);
  LAUNCH_CHECK();
  MGF_TRY(prim_exclusive_scan_u32(ctx, G.cell_cnt.p, G.cell_lo.p, (size_t)cells + 1));
  FaceGrid FG = t->face_grid();
  k_scatter_leaves<<<nblk(nf), kBlock, 0, s>>>(FG.T, G.fb_c.p, G.fb_r.p, G.cell_of.p, G.cell_rank.p, G.brank.p, nullptr, nullptr);
  LAUNCH_CHECK();
  MGF_HIP_TRY(hipStreamSynchronize(s));
  G.ready = true;
  return MGF_OK;
}

extern "C" mgf_status mgf_world_set_terrain(mgf_world* w, const mgf_mesh* mesh) {
  if (!w) return fail(MGF_ERR_INVALID, "world is NULL");
  MGF_TRY(ctx_bind(w->ctx));
  if (!mesh) { w->terrain.reset(); return MGF_OK; }
  std::unique_ptr<mgf_mesh> t(new mgf_mesh());
  t->ctx = w->ctx;
  t->x = mesh->x;
  t->verts = mesh->verts;
  t->faces = mesh->faces;
  t->m.tree = mesh->m.tree;
  t->geom_version = 1;
  MGF_TRY(t->sync());
  MGF_TRY(build_face_grid(t.get()));
  w->terrain = std::move(t);
  w->terrain_grid_off = false;
  return MGF_OK;
}

template <class T>
static mgf_status append(mgf_ctx* ctx, DBuf<T>& buf, size_t old_n, const std::vector<T>& add) {
  MGF_TRY(buf.ensure(old_n + add.size(), ctx->stream, true, old_n));
  return h2d(ctx, buf.p + old_n, add.data(), add.size());
}

// RigidBodyVec::add_body physics.rs:200-218 + World::add_body world.rs:178-184 (initial fat AABB).
extern "C" mgf_status mgf_world_add_bodies(mgf_world* w, const mgf_component* comps, int64_t n, const float* mass, const float* restitution,
                                           const float* friction, const mgf_vec3* world_force, uint64_t* first_id) {
  if (!w || (n && (!comps || !mass || !restitution || !friction || !world_force))) return fail(MGF_ERR_INVALID, "NULL argument");
  MGF_TRY(ctx_bind(w->ctx));
  w->n = w->n_owned;  // ghosts of a previous tick are dropped
  if (first_id) *first_id = w->n_owned;
  if (n <= 0) return MGF_OK;
  if ((uint64_t)w->n_owned + (uint64_t)n > 0x7FFFFFF0ull) return fail(MGF_ERR_INVALID, "too many bodies");
  size_t N = (size_t)n;
  std::vector<float4> hx(N), hq(N), hs(4 * N), h0(N), h1(N), hc(N), hi(3 * N), hd(N), he(N), c0(N), c1(N), tc(N), tr(N), fc(N), fr(N);
  bool hs_ = w->has_sphere, hc_ = w->has_capsule;
  for (size_t i = 0; i < N; ++i) {
    const mgf_component& mc = comps[i];
    if (mc.tag != MGF_SPHERE && mc.tag != MGF_CAPSULE) return fail(MGF_ERR_INVALID, "component tag must be sphere or capsule");
    if (!(mc.r > 0.0f)) return fail(MGF_ERR_INVALID, "radius must be > 0 (geom.rs:300,328)");
    Comp k = comp_of(mc);
    if (k.kind == KIND_SPHERE) k.d = mk3(0, 0, 0);
    // Component::deconstruct compound.rs:42-52
    V3 px; Quat pq; float half_h = 0.0f;
    if (k.kind == KIND_SPHERE) { px = k.p; pq = mkq(1.0f, mk3(0, 0, 0)); hs_ = true; }
    else {
      float h = mag(k.d);
      pq = quat_from_arc(mk3(0.0f, 1.0f, 0.0f) * h, k.d);
      px = k.p + k.d * 0.5f;
      half_h = h * 0.5f;
      hc_ = true;
    }
    Comp local = k; local.p = k.p + -px;  // collider - x.to_vec()
    M3 inv;
    if (!invert(tensor_of(local, mass[i]), &inv)) return fail(MGF_ERR_SINGULAR, "inertia tensor is not invertible (physics.rs:212)");
    float inv_mass = 1.0f / mass[i];
    V3 force = mk3(world_force[i].x, world_force[i].y, world_force[i].z) * mass[i];
    hx[i] = make_float4(px.x, px.y, px.z, 0.0f);
    hq[i] = make_float4(pq.s, pq.v.x, pq.v.y, pq.v.z);
    hs[4 * i] = make_float4(0, 0, 0, 0);
    hs[4 * i + 1] = make_float4(0, 0, inv_mass, inv.c[0].x);
    hs[4 * i + 2] = make_float4(inv.c[0].y, inv.c[0].z, inv.c[1].x, inv.c[1].y);
    hs[4 * i + 3] = make_float4(inv.c[1].z, inv.c[2].x, inv.c[2].y, inv.c[2].z);
    h0[i] = make_float4(force.x, force.y, force.z, restitution[i]);
    h1[i] = make_float4(0, 0, 0, friction[i]);
    uint32_t kind_bits = (uint32_t)k.kind;
    float kf; memcpy(&kf, &kind_bits, 4);
    hc[i] = make_float4(kf, k.r, half_h, 0.0f);
    for (int c = 0; c < 3; ++c) hi[3 * i + c] = make_float4(inv.c[c].x, inv.c[c].y, inv.c[c].z, 0.0f);
    hd[i] = make_float4(0, 0, 0, friction[i]);
    he[i] = make_float4(px.x, px.y, px.z, restitution[i]);
    c0[i] = make_float4(k.p.x, k.p.y, k.p.z, k.r);
    c1[i] = make_float4(k.d.x, k.d.y, k.d.z, kf);
    Box tb = swept_bounds(k, mk3(0, 0, 0));
    tc[i] = make_float4(tb.c.x, tb.c.y, tb.c.z, 0); tr[i] = make_float4(tb.r.x, tb.r.y, tb.r.z, 0);
    V3 fm = mk3(w->params.fat_margin, w->params.fat_margin, w->params.fat_margin);
    V3 frr = tb.r + fm;
    fc[i] = tc[i]; fr[i] = make_float4(frr.x, frr.y, frr.z, 0);
  }
  mgf_ctx* ctx = w->ctx;
  size_t o = w->n_owned;
  MGF_TRY(append(ctx, w->x, o, hx)); MGF_TRY(append(ctx, w->q, o, hq)); MGF_TRY(append(ctx, w->srec, 4 * o, hs));
  MGF_TRY(append(ctx, w->sp0, o, h0)); MGF_TRY(append(ctx, w->sp1, o, h1)); MGF_TRY(append(ctx, w->ctor, o, hc));
  MGF_TRY(append(ctx, w->imb, 3 * o, hi)); MGF_TRY(append(ctx, w->delta, o, hd)); MGF_TRY(append(ctx, w->einfo, o, he));
  MGF_TRY(append(ctx, w->col0, o, c0)); MGF_TRY(append(ctx, w->col1, o, c1)); MGF_TRY(append(ctx, w->tb_c, o, tc));
  MGF_TRY(append(ctx, w->tb_r, o, tr)); MGF_TRY(append(ctx, w->fb_c, o, fc)); MGF_TRY(append(ctx, w->fb_r, o, fr));
  if (w->has_compound) {  // ordinary bodies in a world that has bodies of several parts: empty part slots
    std::vector<uint32_t> pz(N, 0u);
    std::vector<float4> fz(kMaxParts * N, make_float4(0, 0, 0, 0));
    MGF_TRY(append(ctx, w->pcount, o, pz));
    MGF_TRY(append(ctx, w->lp0, kMaxParts * o, fz)); MGF_TRY(append(ctx, w->lp1, kMaxParts * o, fz));
    MGF_TRY(append(ctx, w->wp0, kMaxParts * o, fz)); MGF_TRY(append(ctx, w->wp1, kMaxParts * o, fz));
  }
  w->n_owned += (uint32_t)n;
  w->n = w->n_owned;
  w->has_sphere = hs_; w->has_capsule = hc_;
  w->constraints_ready = false;
  return MGF_OK;
}

// Bodies of several components (BASELINE config 5).  NOT in the reference (physics.rs:200 takes one Component); the
// definition is the oracle's RigidBodyVec::add_compound_body: mass = sum, x = centre of mass, q = identity, tensor = sum
// of the components' tensors about the centre of mass (physics.rs:30-93), parts fixed in the body frame.
extern "C" mgf_status mgf_world_add_compound_bodies(mgf_world* w, const mgf_component* comps, const float* comp_mass, const int64_t* offsets,
                                                    int64_t n, const float* restitution, const float* friction, const mgf_vec3* world_force,
                                                    uint64_t* first_id) {
  if (!w || (n && (!comps || !comp_mass || !offsets || !restitution || !friction || !world_force))) return fail(MGF_ERR_INVALID, "NULL argument");
  MGF_TRY(ctx_bind(w->ctx));
  w->n = w->n_owned;
  if (first_id) *first_id = w->n_owned;
  if (n <= 0) return MGF_OK;
  if ((uint64_t)w->n_owned + (uint64_t)n > 0x7FFFFFF0ull) return fail(MGF_ERR_INVALID, "too many bodies");
  const size_t N = (size_t)n, o = w->n_owned;
  std::vector<float4> hx(N), hq(N), hs(4 * N), h0(N), h1(N), hc(N), hi(3 * N), hd(N), he(N), c0(N), c1(N), tc(N), tr(N), fc(N), fr(N);
  std::vector<uint32_t> hp(N);
  std::vector<float4> l0(kMaxParts * N, make_float4(0, 0, 0, 0)), l1(l0), w0(l0), w1(l0);
  bool hs_ = w->has_sphere, hc_ = w->has_capsule;
  for (size_t b = 0; b < N; ++b) {
    const int64_t k0 = offsets[b], k1 = offsets[b + 1];
    if (k1 <= k0 || k1 - k0 > kMaxParts) return fail(MGF_ERR_INVALID, "a body needs 1..2 components");
    Comp part[kMaxParts];
    float total = 0.0f;
    V3 acc = mk3(0, 0, 0);
    for (int64_t k = k0; k < k1; ++k) {
      const mgf_component& mc = comps[k];
      if (mc.tag != MGF_SPHERE && mc.tag != MGF_CAPSULE) return fail(MGF_ERR_INVALID, "component tag must be sphere or capsule");
      if (!(mc.r > 0.0f)) return fail(MGF_ERR_INVALID, "radius must be > 0 (geom.rs:300,328)");
      Comp c = comp_of(mc);
      if (c.kind == KIND_SPHERE) { c.d = mk3(0, 0, 0); hs_ = true; } else hc_ = true;
      part[k - k0] = c;
      total += comp_mass[k];
      acc = acc + comp_center(c) * comp_mass[k];
    }
    const V3 com = acc / total;
    M3 t = m3_cols(mk3(0, 0, 0), mk3(0, 0, 0), mk3(0, 0, 0));
    for (int64_t k = k0; k < k1; ++k) { Comp l = part[k - k0]; l.p = l.p + -com; t = t + tensor_of(l, comp_mass[k]); }
    M3 inv;
    if (!invert(t, &inv)) return fail(MGF_ERR_SINGULAR, "inertia tensor is not invertible");
    const float inv_mass = 1.0f / total;
    const V3 force = mk3(world_force[b].x, world_force[b].y, world_force[b].z) * total;
    hx[b] = make_float4(com.x, com.y, com.z, 0.0f);
    hq[b] = make_float4(1.0f, 0.0f, 0.0f, 0.0f);
    hs[4 * b] = make_float4(0, 0, 0, 0);
    hs[4 * b + 1] = make_float4(0, 0, inv_mass, inv.c[0].x);
    hs[4 * b + 2] = make_float4(inv.c[0].y, inv.c[0].z, inv.c[1].x, inv.c[1].y);
    hs[4 * b + 3] = make_float4(inv.c[1].z, inv.c[2].x, inv.c[2].y, inv.c[2].z);
    h0[b] = make_float4(force.x, force.y, force.z, restitution[b]);
    h1[b] = make_float4(0, 0, 0, friction[b]);
    uint32_t kind_bits = 2u;  // constructor kind: a body of several parts
    float kf; memcpy(&kf, &kind_bits, 4);
    hc[b] = make_float4(kf, 0.0f, 0.0f, 0.0f);
    for (int c = 0; c < 3; ++c) hi[3 * b + c] = make_float4(inv.c[c].x, inv.c[c].y, inv.c[c].z, 0.0f);
    hd[b] = make_float4(0, 0, 0, friction[b]);
    he[b] = make_float4(com.x, com.y, com.z, restitution[b]);
    uint32_t sph = (uint32_t)KIND_SPHERE; float sf; memcpy(&sf, &sph, 4);
    c0[b] = make_float4(com.x, com.y, com.z, 0.0f);  // the carrier: a radius-0 sphere at the centre of mass
    c1[b] = make_float4(0, 0, 0, sf);
    Box tb;
    for (int64_t k = k0; k < k1; ++k) {
      const Comp& c = part[k - k0];
      uint32_t kb = (uint32_t)c.kind; float kbf; memcpy(&kbf, &kb, 4);
      const V3 lp = c.p + -com;
      l0[kMaxParts * b + (k - k0)] = make_float4(lp.x, lp.y, lp.z, c.r);
      l1[kMaxParts * b + (k - k0)] = make_float4(c.d.x, c.d.y, c.d.z, kbf);
      w0[kMaxParts * b + (k - k0)] = make_float4(c.p.x, c.p.y, c.p.z, c.r);
      w1[kMaxParts * b + (k - k0)] = make_float4(c.d.x, c.d.y, c.d.z, kbf);
      Box pb = swept_bounds(c, mk3(0, 0, 0));
      tb = k == k0 ? pb : box_combine(tb, pb);
    }
    hp[b] = (uint32_t)(k1 - k0);
    tc[b] = make_float4(tb.c.x, tb.c.y, tb.c.z, 0); tr[b] = make_float4(tb.r.x, tb.r.y, tb.r.z, 0);
    const V3 fm = mk3(w->params.fat_margin, w->params.fat_margin, w->params.fat_margin), frr = tb.r + fm;
    fc[b] = tc[b]; fr[b] = make_float4(frr.x, frr.y, frr.z, 0);
  }
  mgf_ctx* ctx = w->ctx;
  if (!w->has_compound && o > 0) {  // the bodies added so far are ordinary: empty part slots for them
    std::vector<uint32_t> pz(o, 0u);
    std::vector<float4> fz(kMaxParts * o, make_float4(0, 0, 0, 0));
    MGF_TRY(append(ctx, w->pcount, 0, pz));
    MGF_TRY(append(ctx, w->lp0, 0, fz)); MGF_TRY(append(ctx, w->lp1, 0, fz)); MGF_TRY(append(ctx, w->wp0, 0, fz)); MGF_TRY(append(ctx, w->wp1, 0, fz));
  }
  MGF_TRY(append(ctx, w->x, o, hx)); MGF_TRY(append(ctx, w->q, o, hq)); MGF_TRY(append(ctx, w->srec, 4 * o, hs));
  MGF_TRY(append(ctx, w->sp0, o, h0)); MGF_TRY(append(ctx, w->sp1, o, h1)); MGF_TRY(append(ctx, w->ctor, o, hc));
  MGF_TRY(append(ctx, w->imb, 3 * o, hi)); MGF_TRY(append(ctx, w->delta, o, hd)); MGF_TRY(append(ctx, w->einfo, o, he));
  MGF_TRY(append(ctx, w->col0, o, c0)); MGF_TRY(append(ctx, w->col1, o, c1)); MGF_TRY(append(ctx, w->tb_c, o, tc));
  MGF_TRY(append(ctx, w->tb_r, o, tr)); MGF_TRY(append(ctx, w->fb_c, o, fc)); MGF_TRY(append(ctx, w->fb_r, o, fr));
  MGF_TRY(append(ctx, w->pcount, o, hp));
  MGF_TRY(append(ctx, w->lp0, kMaxParts * o, l0)); MGF_TRY(append(ctx, w->lp1, kMaxParts * o, l1));
  MGF_TRY(append(ctx, w->wp0, kMaxParts * o, w0)); MGF_TRY(append(ctx, w->wp1, kMaxParts * o, w1));
  w->has_compound = true;
  w->n_owned += (uint32_t)n;
  w->n = w->n_owned;
  w->has_sphere = hs_; w->has_capsule = hc_;
  w->constraints_ready = false;
  return MGF_OK;
}

// ---- state access ----------------------------------------------------------------------------
extern "C" mgf_status mgf_world_read_state(mgf_world* w, mgf_vec3* x, mgf_quat* q, mgf_vec3* v, mgf_vec3* omega, mgf_vec3* delta, int64_t cap) {
  if (!w) return fail(MGF_ERR_INVALID, "world is NULL");
  MGF_TRY(ctx_bind(w->ctx));
  size_t n = w->n_owned;
  if ((int64_t)n > cap) return fail(MGF_ERR_CAPACITY, "state buffers too small");
  std::vector<float4> t(4 * n);
  if (x) { MGF_TRY(d2h(w->ctx, t.data(), w->x.p, n)); for (size_t i = 0; i < n; ++i) x[i] = {t[i].x, t[i].y, t[i].z}; }
  if (q) { MGF_TRY(d2h(w->ctx, t.data(), w->q.p, n)); for (size_t i = 0; i < n; ++i) q[i] = {t[i].x, t[i].y, t[i].z, t[i].w}; }
  if (delta) { MGF_TRY(d2h(w->ctx, t.data(), w->delta.p, n)); for (size_t i = 0; i < n; ++i) delta[i] = {t[i].x, t[i].y, t[i].z}; }
  if (v || omega) {
    MGF_TRY(d2h(w->ctx, t.data(), w->srec.p, 4 * n));
    for (size_t i = 0; i < n; ++i) {
      if (v) v[i] = {t[4 * i].x, t[4 * i].y, t[4 * i].z};
      if (omega) omega[i] = {t[4 * i].w, t[4 * i + 1].x, t[4 * i + 1].y};
    }
  }
  return MGF_OK;
}
extern "C" mgf_status mgf_world_write_state(mgf_world* w, const mgf_vec3* x, const mgf_quat* q, const mgf_vec3* v, const mgf_vec3* omega,
                                            const mgf_vec3* delta, int64_t n_in) {
  if (!w) return fail(MGF_ERR_INVALID, "world is NULL");
  MGF_TRY(ctx_bind(w->ctx));
  size_t n = w->n_owned;
  if ((size_t)n_in != n) return fail(MGF_ERR_INVALID, "n must equal the number of bodies");
  std::vector<float4> t(4 * n);
  if (x) { for (size_t i = 0; i < n; ++i) t[i] = make_float4(x[i].x, x[i].y, x[i].z, 0); MGF_TRY(h2d(w->ctx, w->x.p, t.data(), n)); }
  if (q) { for (size_t i = 0; i < n; ++i) t[i] = make_float4(q[i].s, q[i].x, q[i].y, q[i].z); MGF_TRY(h2d(w->ctx, w->q.p, t.data(), n)); }
  if (delta) {
    MGF_TRY(d2h(w->ctx, t.data(), w->delta.p, n));
    for (size_t i = 0; i < n; ++i) t[i] = make_float4(delta[i].x, delta[i].y, delta[i].z, t[i].w);
    MGF_TRY(h2d(w->ctx, w->delta.p, t.data(), n));
  }
  if (v || omega) {
    MGF_TRY(d2h(w->ctx, t.data(), w->srec.p, 4 * n));
    for (size_t i = 0; i < n; ++i) {
      if (v) { t[4 * i].x = v[i].x; t[4 * i].y = v[i].y; t[4 * i].z = v[i].z; }
      if (omega) { t[4 * i].w = omega[i].x; t[4 * i + 1].x = omega[i].y; t[4 * i + 1].y = omega[i].z; }
    }
    MGF_TRY(h2d(w->ctx, w->srec.p, t.data(), 4 * n));
  }
  if (n) { k_refresh_einfo<<<nblk(n), kBlock, 0, w->ctx->stream>>>(w->bodies(), (uint32_t)n); LAUNCH_CHECK(); }
  MGF_HIP_TRY(hipStreamSynchronize(w->ctx->stream));
  return MGF_OK;
}
extern "C" mgf_status mgf_world_read_colliders(mgf_world* w, mgf_moving_component* out, int64_t cap) {
  if (!w || !out) return fail(MGF_ERR_INVALID, "NULL argument");
  MGF_TRY(ctx_bind(w->ctx));
  size_t n = w->n_owned;
  if ((int64_t)n > cap) return fail(MGF_ERR_CAPACITY, "buffer too small");
  std::vector<float4> a(n), b(n), d(n);
  MGF_TRY(d2h(w->ctx, a.data(), w->col0.p, n)); MGF_TRY(d2h(w->ctx, b.data(), w->col1.p, n)); MGF_TRY(d2h(w->ctx, d.data(), w->delta.p, n));
  for (size_t i = 0; i < n; ++i) {
    uint32_t kind; memcpy(&kind, &b[i].w, 4);
    out[i].shape.tag = (int32_t)kind;
    out[i].shape.p = {a[i].x, a[i].y, a[i].z};
    out[i].shape.d = {b[i].x, b[i].y, b[i].z};
    out[i].shape.r = a[i].w;
    out[i].delta = {d[i].x, d[i].y, d[i].z};
  }
  return MGF_OK;
}
// ConstrainedSet::get physics.rs:273-304
extern "C" mgf_status mgf_world_get(mgf_world* w, const mgf_body_ref* r, mgf_velocity* vel, mgf_rigid_body_info* info) {
  if (!w || !r) return fail(MGF_ERR_INVALID, "NULL argument");
  MGF_TRY(ctx_bind(w->ctx));
  if (r->tag == 1) {
    if (vel) { vel->linear = {0, 0, 0}; vel->angular = {0, 0, 0}; }
    if (info) { info->x = r->center; info->restitution = 0.0f; info->friction = r->friction; info->inv_mass = 0.0f; for (float& f : info->inv_moment) f = 0.0f; }
    return MGF_OK;
  }
  if (r->index >= w->n_owned) return fail(MGF_ERR_INVALID, "index out of bounds");
  float4 s[4], e, d;
  MGF_TRY(d2h(w->ctx, s, w->srec.p + 4 * (size_t)r->index, 4));
  MGF_TRY(d2h(w->ctx, &e, w->einfo.p + r->index, 1));
  MGF_TRY(d2h(w->ctx, &d, w->delta.p + r->index, 1));
  if (vel) { vel->linear = {s[0].x, s[0].y, s[0].z}; vel->angular = {s[0].w, s[1].x, s[1].y}; }
  if (info) {
    info->x = {e.x, e.y, e.z}; info->restitution = e.w; info->friction = d.w; info->inv_mass = s[1].z;
    float im[9] = {s[1].w, s[2].x, s[2].y, s[2].z, s[2].w, s[3].x, s[3].y, s[3].z, s[3].w};
    memcpy(info->inv_moment, im, sizeof(im));
  }
  return MGF_OK;
}
// ConstrainedSet::set physics.rs:306-314
extern "C" mgf_status mgf_world_set(mgf_world* w, const mgf_body_ref* r, const mgf_velocity* vel) {
  if (!w || !r || !vel) return fail(MGF_ERR_INVALID, "NULL argument");
  MGF_TRY(ctx_bind(w->ctx));
  if (r->tag == 1) return MGF_OK;
  if (r->index >= w->n_owned) return fail(MGF_ERR_INVALID, "index out of bounds");
  float4 s[2];
  MGF_TRY(d2h(w->ctx, s, w->srec.p + 4 * (size_t)r->index, 2));
  s[0] = make_float4(vel->linear.x, vel->linear.y, vel->linear.z, vel->angular.x);
  s[1].x = vel->angular.y; s[1].y = vel->angular.z;
  return h2d(w->ctx, w->srec.p + 4 * (size_t)r->index, s, 2);
}
extern "C" mgf_status mgf_world_device_ptr(mgf_world* w, const char* name, void** ptr, int64_t* bytes) {
  if (!w || !name || !ptr) return fail(MGF_ERR_INVALID, "NULL argument");
  struct { const char* n; void* p; size_t per; } tab[] = {
      {"x", w->x.p, 16}, {"q", w->q.p, 16}, {"solver_rec", w->srec.p, 64}, {"delta", w->delta.p, 16}};
  for (auto& t : tab)
    if (!strcmp(name, t.n)) { *ptr = t.p; if (bytes) *bytes = (int64_t)(t.per * w->n_owned); return MGF_OK; }
  return fail(MGF_ERR_INVALID, "unknown array name");
}

// The tiling entry points are synchronous by default (the caller may touch its buffers as soon as the call
// returns); with option stream_ordered = 1 they only enqueue on the ctx stream - for a caller that issues its
// own work (copies, RCCL) on that same stream (mgf_ctx_set_stream).
static mgf_status sync_unless_ordered(mgf_world* w) {
  if (!w->opt_stream_ordered) MGF_HIP_TRY(hipStreamSynchronize(w->ctx->stream));
  return MGF_OK;
}

// ---- the tick ----------------------------------------------------------------------------------
static mgf_status world_integrate(mgf_world* w, float dt, bool complete, bool integrate, bool with_bounds) {
  mgf_ctx* ctx = w->ctx;
  if (w->n_owned == 0) return MGF_OK;
  k_integrate<<<nblk(w->n_owned), kBlock, 0, ctx->stream>>>(w->bodies(), w->n_owned, dt, w->params.fat_margin, complete ? 1 : 0, integrate ? 1 : 0,
                                                      with_bounds ? w->sb.p : nullptr);
  LAUNCH_CHECK();
  return MGF_OK;
}
extern "C" mgf_status mgf_world_complete_motion(mgf_world* w) {
  if (!w) return fail(MGF_ERR_INVALID, "world is NULL");
  MGF_TRY(ctx_bind(w->ctx));
  MGF_TRY(world_integrate(w, 0.0f, true, false, false));
  MGF_HIP_TRY(hipStreamSynchronize(w->ctx->stream));
  return MGF_OK;
}
extern "C" mgf_status mgf_world_integrate(mgf_world* w, float dt) {
  if (!w) return fail(MGF_ERR_INVALID, "world is NULL");
  MGF_TRY(ctx_bind(w->ctx));
  MGF_TRY(world_integrate(w, dt, false, true, false));
  MGF_HIP_TRY(hipStreamSynchronize(w->ctx->stream));
  return MGF_OK;
}

static mgf_status launch_pairs(mgf_world* w, int ka, int kb, const uint32_t* work, const uint32_t* m_ptr, uint32_t cap) {
  if (cap == 0) return MGF_OK;
  hipStream_t s = w->ctx->stream;
  Bodies B = w->bodies();
  unsigned g = nblk(cap);
  if (ka == 0 && kb == 0) k_narrow_pairs<0, 0><<<g, kBlock, 0, s>>>(B, work, m_ptr, w->p_owner.p, w->p_cand.p, w->p_nc.p, w->p_out.p);
  else if (ka == 0 && kb == 1) k_narrow_pairs<0, 1><<<g, kBlock, 0, s>>>(B, work, m_ptr, w->p_owner.p, w->p_cand.p, w->p_nc.p, w->p_out.p);
  else if (ka == 1 && kb == 0) k_narrow_pairs<1, 0><<<g, kBlock, 0, s>>>(B, work, m_ptr, w->p_owner.p, w->p_cand.p, w->p_nc.p, w->p_out.p);
  else k_narrow_pairs<1, 1><<<g, kBlock, 0, s>>>(B, work, m_ptr, w->p_owner.p, w->p_cand.p, w->p_nc.p, w->p_out.p);
  LAUNCH_CHECK();
  return MGF_OK;
}
static mgf_status launch_terrain(mgf_world* w, int ka, const TerrainDev& M, const uint32_t* work, const uint32_t* m_ptr, uint32_t cap) {
  if (cap == 0) return MGF_OK;
  hipStream_t s = w->ctx->stream;
  Bodies B = w->bodies();
  unsigned g = nblk(cap);
  if (ka == 0) k_narrow_terrain<0><<<g, kBlock, 0, s>>>(B, M, work, m_ptr, w->t_owner.p, w->t_cand.p, w->t_nc.p, w->t_out.p);
  else k_narrow_terrain<1><<<g, kBlock, 0, s>>>(B, M, work, m_ptr, w->t_owner.p, w->t_cand.p, w->t_nc.p, w->t_out.p);
  LAUNCH_CHECK();
  return MGF_OK;
}

// Dependency links of the insertion-ordered list cons_nat[0..C) (ConsLinks).
static mgf_status links_ensure(mgf_world* w, uint32_t cap_c, bool zero = false) {
  hipStream_t s = w->ctx->stream;
  uint32_t n = w->n;
  MGF_TRY(w->c_ab.ensure(std::max(cap_c, 1u), s)); MGF_TRY(w->c_succ.ensure(std::max(cap_c, 1u), s)); MGF_TRY(w->c_pred.ensure(2 * (size_t)std::max(cap_c, 1u), s));
  MGF_TRY(w->deg.ensure(n + 1, s)); MGF_TRY(w->adj_off.ensure(n + 1, s)); MGF_TRY(w->adj_fill.ensure(n + 1, s));
  MGF_TRY(w->adj_list.ensure(2 * (size_t)std::max(cap_c, 1u), s));
  if (zero) {
    MGF_HIP_TRY(hipMemsetAsync(w->deg.p, 0, (n + 1) * 4, s));
    MGF_HIP_TRY(hipMemsetAsync(w->adj_fill.p, 0, (n + 1) * 4, s));
  }
  return MGF_OK;
}
// c_ab and deg are filled (by the setup kernels, or k_links_from_records): sorted per-body adjacency, successor
// words, predecessor flags.  C is read on the device (sc->C); grids and buffers are sized by `cap_c`.
static mgf_status build_dag(mgf_world* w, uint32_t cap_c) {
  mgf_ctx* ctx = w->ctx;
  hipStream_t s = ctx->stream;
  uint32_t n = w->n;
  w->depth = 0;
  if (cap_c == 0) return MGF_OK;
  if (cap_c >= kSuccId) return fail(MGF_ERR_CAPACITY, "too many constraints");
  MGF_TRY(prim_exclusive_scan_u32(ctx, w->deg.p, w->adj_off.p, (size_t)n + 1));
  k_adj_fill<<<nblk(cap_c), kBlock, 0, s>>>(w->c_ab.p, &w->sc.p->C, w->adj_off.p, w->adj_fill.p, w->adj_list.p);
  LAUNCH_CHECK();
  k_chain<<<nblk(n), kBlock, 0, s>>>(n, w->links(), w->adj_off.p, w->adj_list.p);
  LAUNCH_CHECK();
  return MGF_OK;
}

// First half of the tick: drop last tick's ghosts, complete_motion + integrate the owned bodies
// (world.rs:230-231).  After this call a tiled driver may import ghost bodies.
static mgf_status world_begin(mgf_world* w, float dt) {
  if (!w) return fail(MGF_ERR_INVALID, "world is NULL");
  MGF_TRY(ctx_bind(w->ctx));
  hipStream_t s = w->ctx->stream;
  memset(&w->stats, 0, sizeof(w->stats));
  w->n = w->n_owned;
  w->stats.n_bodies = w->n_owned;
  w->last_dt = dt;
  w->constraints_ready = false;
  w->tick_two_pass = false;
  w->C = w->Ct = w->Mt = w->Mp = 0;
  MGF_HIP_TRY(hipEventRecord(w->ev[0], s));
  k_reset_step<<<1, 64, 0, s>>>(w->sb.p, w->d_err());
  LAUNCH_CHECK();
  MGF_TRY(world_integrate(w, dt, true, true, true));
  return MGF_OK;
}
extern "C" mgf_status mgf_world_begin_tick(mgf_world* w, float dt) {
  MGF_TRY(world_begin(w, dt));
  return sync_unless_ordered(w);
}

// Second half: broadphase, narrowphase, ContactConstraint::new over owned + ghost bodies.  Enqueue only:
// nothing is read back, list sizes stay on the device (StepCounts), buffers and grids use the
// capacities cap_t / cap_p / cap_c.
static mgf_status collide_enqueue(mgf_world* w, float dt) {
  mgf_ctx* ctx = w->ctx;
  hipStream_t s = ctx->stream;
  const uint32_t n = w->n;
  StepCounts* sc = w->sc.p;
  if (n == 0) { MGF_HIP_TRY(hipMemsetAsync(sc, 0, sizeof(StepCounts), s)); return MGF_OK; }
  Bodies B = w->bodies();
  // first tick: room for a few candidates per body; later ticks: last tick's sizes plus slack (collide_finish)
  if (w->cap_p == 0) { w->cap_p = std::max(4 * n, 1024u); w->cap_t = std::max(2 * n, 1024u); w->cap_c = std::max(4 * n, 1024u); }
  const uint32_t cap_t = w->cap_t, cap_p = w->cap_p, cap_c = w->cap_c;
  // 2. linear BVH over the fat AABBs
  uint32_t levels = 4;  // 4^levels Morton cells, about one body per cell
  while (((uint64_t)1 << (2 * levels)) < n && levels < (uint32_t)kMortonBits / 2) ++levels;
  const uint32_t cells = 1u << (2 * levels), nblocks = cells / kBlock;
  MGF_TRY(w->cell_of.ensure(n, s)); MGF_TRY(w->cell_rank.ensure(n, s)); MGF_TRY(w->sidx.ensure(n, s)); MGF_TRY(w->brank.ensure(n, s));
  w->flow5_ok = true; w->flow5_prepped = false;
  MGF_TRY(w->lnodes.ensure(qlevel_offset(levels), s)); MGF_TRY(w->leaves.ensure(n, s));
  MGF_TRY(w->cell_lo.ensure((size_t)cells + 1, s)); MGF_TRY(w->cell_cnt.ensure((size_t)cells + 1, s));
  MGF_TRY(w->sub_lo.ensure(nblocks, s)); MGF_TRY(w->sub_hi.ensure(nblocks, s));
  MGF_TRY(w->sub2_lo.ensure(nblocks / 4 + 1, s)); MGF_TRY(w->sub2_hi.ensure(nblocks / 4 + 1, s));
  MGF_TRY(w->t_cnt.ensure(n + 1, s)); MGF_TRY(w->p_cnt.ensure(n + 1, s)); MGF_TRY(w->t_off.ensure(n + 1, s)); MGF_TRY(w->p_off.ensure(n + 1, s));
  MGF_TRY(w->cons_nat.ensure(cap_c, s));
  MGF_TRY(links_ensure(w, cap_c));
  MGF_TRY(w->degb.ensure(n + 1, s)); MGF_TRY(w->rev.ensure((size_t)n * w->rev_cap, s)); MGF_TRY(w->pair_stat.ensure(64, s));
  {  // one launch clears every per-tick counter array
    ZeroList z;
    memset(&z, 0, sizeof(z));
    z.p[0] = w->cell_cnt.p; z.words[0] = cells + 1;
    z.p[1] = w->t_cnt.p; z.words[1] = n + 1;  // ghosts have no terrain row
    z.p[2] = w->degb.p; z.words[2] = n + 1;
    z.p[3] = w->d_err() + 7; z.words[3] = 1;  // row-of-b-occurrences overflow flag
    z.p[4] = w->d_err() + 1; z.words[4] = 1;  // row-overflow flag (re-armed for a re-run inside the tick)
    z.p[5] = w->d_err() + 3; z.words[5] = 1;  // grid-too-wide flag
    z.p[6] = w->d_err() + 5; z.words[6] = 1;  // terrain-grid-too-wide flag
    z.p[7] = w->pair_stat.p; z.words[7] = 64;
    k_zero_many<<<256, kBlock, 0, s>>>(z);
    LAUNCH_CHECK();
  }
  k_scene_bounds<<<std::min<unsigned>(nblk(n), 256u), kBlock, 0, s>>>(w->fb_c.p, w->fb_r.p, n, w->sb.p);
  LAUNCH_CHECK();
  k_morton_count<<<nblk(n), kBlock, 0, s>>>(w->fb_c.p, n, w->sb.p, kMortonBits - 2 * (int)levels, w->cell_of.p, w->cell_rank.p, w->cell_cnt.p);
  LAUNCH_CHECK();
  MGF_TRY(prim_exclusive_scan_u32(ctx, w->cell_cnt.p, w->cell_lo.p, (size_t)cells + 1));
  Lbvh T;
  T.nodes = w->lnodes.p; T.leaves = w->leaves.p; T.sidx = w->sidx.p; T.cell_lo = w->cell_lo.p;
  T.n = n; T.levels = levels; T.err = w->d_err();
  T.dbg = nullptr;
  const bool two_pass = w->opt_two_pass != 0 || w->tick_two_pass;
  const bool use_grid = !two_pass && !w->opt_broadphase_tree && !w->grid_too_wide;
  // a world of spheres: the grid broadphase runs the sphere-sphere test on the partners it accepts and lists contacts only
  const bool fused = use_grid && !w->has_capsule && !w->has_compound && !w->opt_no_fused_narrowphase;
  w->tick_fused = fused;
  T.lcol = nullptr;
  if (fused) { MGF_TRY(w->lcol.ensure(2 * (size_t)n, s)); T.lcol = w->lcol.p; }
  if (w->opt_debug_bvh) {
    MGF_TRY(w->dbg.ensure(4, s));
    MGF_HIP_TRY(hipMemsetAsync(w->dbg.p, 0, 32, s));
    T.dbg = w->dbg.p;
  }
  k_scatter_leaves<<<nblk(n), kBlock, 0, s>>>(T, w->fb_c.p, w->fb_r.p, w->cell_of.p, w->cell_rank.p, w->brank.p, w->col0.p, w->delta.p);
  LAUNCH_CHECK();
  if (!use_grid) {  // inner nodes are only needed by the tree walks
    k_lbvh_low<<<nblocks, kBlock, 0, s>>>(T, w->sub_lo.p, w->sub_hi.p);
    LAUNCH_CHECK();
    if (levels > 4) { k_lbvh_top<<<1, 1024, 0, s>>>(T, w->sub_lo.p, w->sub_hi.p, w->sub2_lo.p, w->sub2_hi.p); LAUNCH_CHECK(); }
  }
  MGF_HIP_TRY(hipEventRecord(w->ev[1], s));
  // 3. candidates
  TerrainDev M;
  if (w->terrain && !w->terrain->m.tree.empty()) M = w->terrain->dev(w->d_err());
  else { memset(&M, 0, sizeof(M)); }
  MGF_TRY(w->t_cand.ensure(cap_t, s)); MGF_TRY(w->t_owner.ensure(cap_t, s));
  MGF_TRY(w->p_cand.ensure(cap_p, s)); MGF_TRY(w->p_owner.ensure(cap_p, s));
  MGF_TRY(w->t_nc.ensure(cap_t, s)); MGF_TRY(w->p_nc.ensure(cap_p, s));
  MGF_TRY(w->t_pre.ensure(cap_t, s)); MGF_TRY(w->p_pre.ensure(cap_p, s));
  const uint32_t t_stride = w->has_compound ? (uint32_t)kTerrainContacts : 2u, p_stride = w->has_compound ? (uint32_t)kPairContacts : 1u;
  MGF_TRY(w->t_out.ensure((size_t)t_stride * cap_t, s)); MGF_TRY(w->p_out.ensure((size_t)p_stride * cap_p, s));
  const bool terrain_grid = !two_pass && M.n_nodes && w->terrain->grid.ready && !w->terrain_grid_off && !w->opt_terrain_tree;
  if (!two_pass) {
    // fast path: one pass, hits written to fixed-capacity rows
    MGF_TRY(w->rows.ensure((size_t)n * kRowCap, s));
    MGF_TRY(w->rows_t.ensure((size_t)n * w->row_cap_t, s));
    if (M.n_nodes && w->n_owned) {
      if (terrain_grid) {
        const uint32_t per_block = kCoopBlock / kCoopLanes;
        const uint32_t tg = 8 * (((w->n_owned + per_block - 1) / per_block + 7) / 8);
        k_terrain_grid<<<tg, kCoopBlock, 0, s>>>(B, w->n_owned, nullptr, M, w->terrain->face_grid(), 1e-3f, w->row_cap_t, w->rows_t.p, w->t_cnt.p,
                                                 w->d_err() + 1, w->d_err() + 5);
      } else {
        k_terrain_rows<<<nblk(w->n_owned), kBlock, 0, s>>>(B, w->n_owned, M, w->row_cap_t, w->rows_t.p, w->t_cnt.p, w->d_err() + 1);
      }
      LAUNCH_CHECK();
    }
    {
      const uint32_t per_block = kCoopBlock / kCoopLanes;
      const uint32_t grid = 8 * (((n + per_block - 1) / per_block + 7) / 8);
      if (fused) k_pair_grid<true><<<grid, kCoopBlock, 0, s>>>(B, n, w->n_owned, T, w->sb.p, 1e-3f, w->rows.p, w->p_cnt.p, w->d_err() + 1, w->d_err() + 3, w->pair_stat.p);
      else if (use_grid) k_pair_grid<false><<<grid, kCoopBlock, 0, s>>>(B, n, w->n_owned, T, w->sb.p, 1e-3f, w->rows.p, w->p_cnt.p, w->d_err() + 1, w->d_err() + 3, nullptr);
      else k_pair_rows<<<grid, kCoopBlock, 0, s>>>(B, n, w->n_owned, T, 1e-3f, w->rows.p, w->p_cnt.p, w->d_err() + 1);
      LAUNCH_CHECK();
    }
  } else {
    k_candidates<false><<<8 * xcd_blocks_per(n), kBlock, 0, s>>>(B, n, w->n_owned, T, M, 1e-3f, w->t_cnt.p, w->p_cnt.p, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
    LAUNCH_CHECK();
  }
  MGF_TRY(prim_exclusive_scan_u32(ctx, w->t_cnt.p, w->t_off.p, (size_t)n + 1));
  MGF_TRY(prim_exclusive_scan_u32(ctx, w->p_cnt.p, w->p_off.p, (size_t)n + 1));
  k_caps_candidates<<<1, 1, 0, s>>>(w->t_off.p + n, w->p_off.p + n, cap_t, cap_p, two_pass ? nullptr : w->d_err() + 1, use_grid ? w->d_err() + 3 : nullptr,
                                    terrain_grid ? w->d_err() + 5 : nullptr, sc);
  LAUNCH_CHECK();
  if (two_pass) {
    k_candidates<true><<<8 * xcd_blocks_per(n), kBlock, 0, s>>>(B, n, w->n_owned, T, M, 1e-3f, nullptr, nullptr, w->t_off.p, w->p_off.p, w->t_cand.p, w->t_owner.p,
                                                  w->p_cand.p, w->p_owner.p, sc);
  } else {
    k_rows_to_csr<<<nblk(n), kBlock, 0, s>>>(sc, n, w->row_cap_t, terrain_grid ? w->terrain->grid.face_of_rank.p : nullptr, w->rows_t.p, w->rows.p, w->t_off.p, w->p_off.p, w->t_cand.p, w->t_owner.p, w->p_cand.p, w->p_owner.p);
  }
  LAUNCH_CHECK();
  MGF_HIP_TRY(hipEventRecord(w->ev[2], s));
  // 4. narrowphase, one kernel per shape-pair type
  bool mixed = w->has_sphere && w->has_capsule;
  if (w->has_compound) {  // bodies of several parts: one kernel over every pair of parts (ordinary bodies are bodies of one part)
    if (cap_p) { k_narrow_pairs_parts<<<nblk(cap_p), kBlock, 0, s>>>(B, &sc->Mp, w->p_owner.p, w->p_cand.p, w->p_nc.p, w->p_out.p); LAUNCH_CHECK(); }
    if (M.n_nodes && cap_t) { k_narrow_terrain_parts<<<nblk(cap_t), kBlock, 0, s>>>(B, M, &sc->Mt, w->t_owner.p, w->t_cand.p, w->t_nc.p, w->t_out.p); LAUNCH_CHECK(); }
  } else if (!mixed) {
    int k = w->has_capsule ? 1 : 0;
    MGF_TRY(launch_pairs(w, k, k, nullptr, &sc->Mp, cap_p));
    if (M.n_nodes) MGF_TRY(launch_terrain(w, k, M, nullptr, &sc->Mt, cap_t));
  } else {
    MGF_TRY(w->work_lists.ensure(4 * (size_t)cap_p + 2 * (size_t)cap_t, s));
    uint32_t* lists_p = w->work_lists.p;
    uint32_t* lists_t = w->work_lists.p + 4 * (size_t)cap_p;
    k_bin_pairs<<<nblk(cap_p), kBlock, 0, s>>>(B, &sc->Mp, cap_p, w->p_owner.p, w->p_cand.p, lists_p, sc->bins);
    LAUNCH_CHECK();
    for (int ty = 0; ty < 4; ++ty) MGF_TRY(launch_pairs(w, ty >> 1, ty & 1, lists_p + (size_t)ty * cap_p, &sc->bins[ty], cap_p));
    if (M.n_nodes) {
      k_bin_terrain<<<nblk(cap_t), kBlock, 0, s>>>(B, &sc->Mt, cap_t, w->t_owner.p, lists_t, sc->bins + 4);
      LAUNCH_CHECK();
      for (int ty = 0; ty < 2; ++ty) MGF_TRY(launch_terrain(w, ty, M, lists_t + (size_t)ty * cap_t, &sc->bins[4 + ty], cap_t));
    }
  }
  MGF_HIP_TRY(hipEventRecord(w->ev[3], s));
  // 5. constraint numbering in insertion order + ContactConstraint::new
  MGF_TRY(w->cnt.ensure(n + 1, s)); MGF_TRY(w->base.ensure(n + 1, s));
  k_count_contacts<<<nblk(n), kBlock, 0, s>>>(sc, n, w->t_off.p, w->p_off.p, w->t_nc.p, w->p_nc.p, w->p_cand.p, w->t_pre.p, w->p_pre.p, w->cnt.p);
  LAUNCH_CHECK();
  MGF_TRY(prim_exclusive_scan_u32(ctx, w->cnt.p, w->base.p, (size_t)n + 1));
  k_caps_constraints<<<1, 1, 0, s>>>(w->base.p + n, &sc->ct_sum, cap_c, sc);
  LAUNCH_CHECK();
  if (M.n_nodes) {
    k_setup_terrain<<<nblk(cap_t), kBlock, 0, s>>>(B, M, sc, w->t_owner.p, w->t_nc.p, w->t_pre.p, w->t_out.p, w->base.p, dt, w->params.baumgarte,
                                                   w->params.penetration_slop, w->cons_nat.p, w->c_ab.p, t_stride);
    LAUNCH_CHECK();
  }
  k_setup_pairs<<<nblk(cap_p), kBlock, 0, s>>>(B, sc, w->p_owner.p, w->p_cand.p, w->p_nc.p, w->p_pre.p, w->p_out.p, w->base.p, dt,
                                               w->params.baumgarte, w->params.penetration_slop, w->cons_nat.p, w->c_ab.p, w->degb.p, w->rev.p,
                                               w->rev_cap, w->d_err() + 7, p_stride);
  LAUNCH_CHECK();
  if (cap_c >= kSuccId) return fail(MGF_ERR_CAPACITY, "too many constraints");
  w->depth = 0;
  k_chain_rows<<<nblk(n), kBlock, 0, s>>>(n, w->links(), w->base.p, w->degb.p, w->rev.p, w->rev_cap, w->d_err() + 7, sc);
  LAUNCH_CHECK();
  MGF_HIP_TRY(hipEventRecord(w->ev[4], s));
  return MGF_OK;
}

// Read the tick's sizes and flags back (the stream must have been synchronised by the caller's copy):
// MGF_OK + *retry=false when the collide phase is complete; *retry=true after growing a capacity or
// switching to the exact two-pass candidate path (the caller re-enqueues the phase).
static mgf_status collide_finish(mgf_world* w, bool* retry) {
  mgf_ctx* ctx = w->ctx;
  hipStream_t s = ctx->stream;
  *retry = false;
  if (w->n == 0) { MGF_HIP_TRY(hipStreamSynchronize(s)); w->constraints_ready = true; return MGF_OK; }
  uint32_t* pin = static_cast<uint32_t*>(ctx->pinned);
  MGF_HIP_TRY(hipMemcpyAsync(pin, w->sc.p, sizeof(StepCounts), hipMemcpyDeviceToHost, s));
  MGF_HIP_TRY(hipMemcpyAsync(pin + 32, w->sb.p, sizeof(SceneBounds), hipMemcpyDeviceToHost, s));
  MGF_HIP_TRY(hipMemcpyAsync(pin + 64, w->d_err(), 12, hipMemcpyDeviceToHost, s));
  if (w->tick_fused) MGF_HIP_TRY(hipMemcpyAsync(pin + 96, w->pair_stat.p, 256, hipMemcpyDeviceToHost, s));
  MGF_HIP_TRY(hipStreamSynchronize(s));
  StepCounts h = *reinterpret_cast<StepCounts*>(pin);
  w->stats.n_refits = reinterpret_cast<SceneBounds*>(pin + 32)->n_refits;
  if (pin[64]) return fail(MGF_ERR_CAPACITY, "BVH traversal stack overflow");
  auto grown = [](uint32_t need) { return (uint32_t)std::min<uint64_t>((uint64_t)need + need / 2 + 1024, 0x7FFFFFF0ull); };
  if (h.fail & kFailRevRow) { w->rev_cap *= 2; *retry = true; }  // wider rows from now on
  if (h.fail & kFailTerrainWide) { w->terrain_grid_off = true; *retry = true; }
  else if (h.fail & kFailGridWide) { w->grid_too_wide = true; *retry = true; }
  else if (h.fail & (kFailRowOverflow | kFailTerrainRow)) {
    w->n_row_overflows++;
    *retry = true;
    if ((h.fail & kFailTerrainRow) && !(h.fail & kFailRowOverflow) && w->row_cap_t < (uint32_t)kRowCapTMax) w->row_cap_t *= 2;  // wider terrain rows from now on
    else w->tick_two_pass = true;  // exact count / fill path for this tick
  }
  if (h.fail & kFailCandCap) {
    if (h.need_Mt > w->cap_t) w->cap_t = grown(h.need_Mt);
    if (h.need_Mp > w->cap_p) w->cap_p = grown(h.need_Mp);
    *retry = true;
  }
  if (h.fail & kFailConsCap) {
    if (h.need_C >= 0x7FFFFFF0u) return fail(MGF_ERR_CAPACITY, "too many constraints");
    w->cap_c = grown(h.need_C);
    *retry = true;
  }
  if (*retry) { w->n_cap_retries++; return MGF_OK; }
  // keep headroom for the next tick (contact counts drift slowly): grow ahead of need, without a re-run
  if ((uint64_t)h.need_Mt * 5 > (uint64_t)w->cap_t * 4) w->cap_t = grown(h.need_Mt);
  if ((uint64_t)h.need_Mp * 5 > (uint64_t)w->cap_p * 4) w->cap_p = grown(h.need_Mp);
  if ((uint64_t)h.need_C * 5 > (uint64_t)w->cap_c * 4) w->cap_c = grown(h.need_C);
  w->Mt = h.Mt; w->Mp = h.Mp; w->C = h.C; w->Ct = h.Ct;
  w->stats.n_terrain_candidates = h.Mt; w->stats.n_pair_candidates = h.Mp;
  if (w->tick_fused) {  // the candidate lists hold contacts only; the accepted partners were counted on the way
    uint64_t acc = 0;
    for (int k = 0; k < 64; ++k) acc += pin[96 + k];
    w->stats.n_pair_candidates = acc;
  }
  w->stats.n_constraints = h.C; w->stats.n_terrain_constraints = h.Ct;
  w->constraints_ready = true;
  if (w->opt_debug_bvh) {
    unsigned long long hd[3];
    MGF_TRY(d2h(ctx, hd, w->dbg.p, 3));
    fprintf(stderr, "[mgf debug_bvh] n=%u node fetches/query=%.1f leaf records/query=%.1f max fetches=%llu\n", w->n, (double)hd[0] / w->n,
            (double)hd[1] / w->n, hd[2]);
  }
  float ms;
  MGF_HIP_TRY(hipEventElapsedTime(&ms, w->ev[0], w->ev[1])); w->stats.ms_integrate = ms;
  MGF_HIP_TRY(hipEventElapsedTime(&ms, w->ev[1], w->ev[2])); w->stats.ms_broadphase = ms;
  MGF_HIP_TRY(hipEventElapsedTime(&ms, w->ev[2], w->ev[3])); w->stats.ms_narrowphase = ms;
  MGF_HIP_TRY(hipEventElapsedTime(&ms, w->ev[3], w->ev[4])); w->stats.ms_setup = ms;
  return MGF_OK;
}

extern "C" mgf_status mgf_world_collide(mgf_world* w, float dt, mgf_step_stats* stats) {
  if (!w) return fail(MGF_ERR_INVALID, "world is NULL");
  MGF_TRY(ctx_bind(w->ctx));
  w->constraints_ready = false;
  w->C = w->Ct = w->Mt = w->Mp = 0;
  for (int attempt = 0; attempt < 8; ++attempt) {
    bool retry = false;
    MGF_TRY(collide_enqueue(w, dt));
    MGF_TRY(collide_finish(w, &retry));
    if (!retry) { if (stats) *stats = w->stats; return MGF_OK; }
  }
  return fail(MGF_ERR_HIP, "internal error: collide phase did not settle its buffer capacities");
}

extern "C" mgf_status mgf_world_build_constraints(mgf_world* w, float dt, mgf_step_stats* stats) {
  MGF_TRY(world_begin(w, dt));
  return mgf_world_collide(w, dt, stats);
}

extern "C" mgf_status mgf_world_select_boundary(mgf_world* w, float x_left, float x_right, uint32_t* ids_left, uint32_t* ids_right,
                                                int64_t cap, int64_t* n_left, int64_t* n_right) {
  if (!w || !n_left || !n_right) return fail(MGF_ERR_INVALID, "NULL argument");
  MGF_TRY(ctx_bind(w->ctx));
  mgf_ctx* ctx = w->ctx;
  hipStream_t s = ctx->stream;
  uint32_t n = w->n_owned;
  *n_left = *n_right = 0;
  if (n == 0) return MGF_OK;
  MGF_TRY(w->bflag_l.ensure(n + 1, s)); MGF_TRY(w->bflag_r.ensure(n + 1, s)); MGF_TRY(w->bscan_l.ensure(n + 1, s)); MGF_TRY(w->bscan_r.ensure(n + 1, s));
  k_boundary_flags<<<nblk(n + 1), kBlock, 0, s>>>(w->bodies(), n, x_left, x_right, w->bflag_l.p, w->bflag_r.p);
  LAUNCH_CHECK();
  MGF_TRY(prim_exclusive_scan_u32(ctx, w->bflag_l.p, w->bscan_l.p, (size_t)n + 1));
  MGF_TRY(prim_exclusive_scan_u32(ctx, w->bflag_r.p, w->bscan_r.p, (size_t)n + 1));
  uint32_t* pin = static_cast<uint32_t*>(ctx->pinned);
  MGF_HIP_TRY(hipMemcpyAsync(pin, w->bscan_l.p + n, 4, hipMemcpyDeviceToHost, s));
  MGF_HIP_TRY(hipMemcpyAsync(pin + 1, w->bscan_r.p + n, 4, hipMemcpyDeviceToHost, s));
  MGF_HIP_TRY(hipStreamSynchronize(s));
  *n_left = pin[0]; *n_right = pin[1];
  if ((int64_t)pin[0] > cap || (int64_t)pin[1] > cap) return fail(MGF_ERR_CAPACITY, "boundary id buffers too small");
  if (!ids_left || !ids_right) return fail(MGF_ERR_INVALID, "NULL id buffer");
  k_boundary_scatter<<<nblk(n), kBlock, 0, s>>>(n, w->bflag_l.p, w->bscan_l.p, w->bflag_r.p, w->bscan_r.p, ids_left, ids_right);
  LAUNCH_CHECK();
  return sync_unless_ordered(w);
}
extern "C" mgf_status mgf_world_export_bodies(mgf_world* w, const uint32_t* ids, int64_t n, float* dst) {
  if (w && w->has_compound) return fail(MGF_ERR_INVALID, "bodies of several parts are supported in single-process worlds only (no ghost / migrant records)");
  if (!w || (n && (!ids || !dst))) return fail(MGF_ERR_INVALID, "NULL argument");
  MGF_TRY(ctx_bind(w->ctx));
  if (n > 0) { k_export_bodies<<<nblk(n), kBlock, 0, w->ctx->stream>>>(w->bodies(), ids, (uint32_t)n, dst); LAUNCH_CHECK(); }
  return sync_unless_ordered(w);
}
template <class T>
static mgf_status grow_keep(mgf_world* w, DBuf<T>& b, size_t per, size_t need) {
  return b.ensure(per * need, w->ctx->stream, true, per * (size_t)w->n_owned);
}
extern "C" mgf_status mgf_world_import_ghosts(mgf_world* w, const float* src, int64_t n_ghost) {
  if (w && w->has_compound) return fail(MGF_ERR_INVALID, "bodies of several parts are supported in single-process worlds only (no ghost / migrant records)");
  if (!w || (n_ghost && !src) || n_ghost < 0) return fail(MGF_ERR_INVALID, "bad argument");
  MGF_TRY(ctx_bind(w->ctx));
  size_t need = (size_t)w->n_owned + (size_t)n_ghost;
  if (need > 0x7FFFFFF0ull) return fail(MGF_ERR_INVALID, "too many bodies");
  MGF_TRY(grow_keep(w, w->x, 1, need)); MGF_TRY(grow_keep(w, w->q, 1, need)); MGF_TRY(grow_keep(w, w->srec, 4, need));
  MGF_TRY(grow_keep(w, w->sp0, 1, need)); MGF_TRY(grow_keep(w, w->sp1, 1, need)); MGF_TRY(grow_keep(w, w->ctor, 1, need));
  MGF_TRY(grow_keep(w, w->imb, 3, need)); MGF_TRY(grow_keep(w, w->delta, 1, need)); MGF_TRY(grow_keep(w, w->einfo, 1, need));
  MGF_TRY(grow_keep(w, w->col0, 1, need)); MGF_TRY(grow_keep(w, w->col1, 1, need)); MGF_TRY(grow_keep(w, w->tb_c, 1, need));
  MGF_TRY(grow_keep(w, w->tb_r, 1, need)); MGF_TRY(grow_keep(w, w->fb_c, 1, need)); MGF_TRY(grow_keep(w, w->fb_r, 1, need));
  if (n_ghost > 0) {
    k_import_ghosts<<<nblk(n_ghost), kBlock, 0, w->ctx->stream>>>(w->bodies(), w->n_owned, (uint32_t)n_ghost, src, w->params.fat_margin);
    LAUNCH_CHECK();
  }
  w->n = (uint32_t)need;
  w->constraints_ready = false;
  return sync_unless_ordered(w);
}
// ---- migration of owned bodies between tiles ---------------------------------------
// mgf_world_select_tile = select_boundary + the bodies whose centre left the slab [x_lo, x_hi).  The common tick has
// no migrant: its cost over select_boundary is one counting kernel, and the counts ride on the same read-back.
extern "C" mgf_status mgf_world_select_tile(mgf_world* w, float x_left, float x_right, float x_lo, float x_hi, uint32_t* ids_left,
                                            uint32_t* ids_right, uint32_t* ids_migrants, int64_t cap, int64_t* counts) {
  if (!w || !counts) return fail(MGF_ERR_INVALID, "NULL argument");
  MGF_TRY(ctx_bind(w->ctx));
  mgf_ctx* ctx = w->ctx;
  hipStream_t s = ctx->stream;
  uint32_t n = w->n_owned;
  counts[0] = counts[1] = counts[2] = counts[3] = 0;
  if (n == 0) return MGF_OK;
  MGF_TRY(w->bflag_l.ensure(n + 1, s)); MGF_TRY(w->bflag_r.ensure(n + 1, s)); MGF_TRY(w->bscan_l.ensure(n + 1, s)); MGF_TRY(w->bscan_r.ensure(n + 1, s));
  MGF_TRY(w->mig_cnt.ensure(2, s));
  MGF_HIP_TRY(hipMemsetAsync(w->mig_cnt.p, 0, 8, s));
  k_boundary_flags<<<nblk(n + 1), kBlock, 0, s>>>(w->bodies(), n, x_left, x_right, w->bflag_l.p, w->bflag_r.p);
  LAUNCH_CHECK();
  k_migrant_count<<<nblk(n), kBlock, 0, s>>>(w->bodies(), n, x_lo, x_hi, w->mig_cnt.p);
  LAUNCH_CHECK();
  MGF_TRY(prim_exclusive_scan_u32(ctx, w->bflag_l.p, w->bscan_l.p, (size_t)n + 1));
  MGF_TRY(prim_exclusive_scan_u32(ctx, w->bflag_r.p, w->bscan_r.p, (size_t)n + 1));
  uint32_t* pin = static_cast<uint32_t*>(ctx->pinned);
  MGF_HIP_TRY(hipMemcpyAsync(pin, w->bscan_l.p + n, 4, hipMemcpyDeviceToHost, s));
  MGF_HIP_TRY(hipMemcpyAsync(pin + 1, w->bscan_r.p + n, 4, hipMemcpyDeviceToHost, s));
  MGF_HIP_TRY(hipMemcpyAsync(pin + 2, w->mig_cnt.p, 8, hipMemcpyDeviceToHost, s));
  MGF_HIP_TRY(hipStreamSynchronize(s));
  const uint32_t nl = pin[0], nr = pin[1], ml = pin[2], mr = pin[3];
  counts[0] = nl; counts[1] = nr; counts[2] = ml; counts[3] = mr;
  if ((int64_t)nl > cap || (int64_t)nr > cap || (int64_t)ml + (int64_t)mr > cap) return fail(MGF_ERR_CAPACITY, "id buffers too small");
  if (!ids_left || !ids_right) return fail(MGF_ERR_INVALID, "NULL id buffer");
  k_boundary_scatter<<<nblk(n), kBlock, 0, s>>>(n, w->bflag_l.p, w->bscan_l.p, w->bflag_r.p, w->bscan_r.p, ids_left, ids_right);
  LAUNCH_CHECK();
  if (ml + mr) {  // rare: build the two ascending lists, left-goers first
    if (!ids_migrants) return fail(MGF_ERR_INVALID, "NULL migrant id buffer");
    k_migrant_flags<<<nblk(n + 1), kBlock, 0, s>>>(w->bodies(), n, x_lo, x_hi, w->bflag_l.p, w->bflag_r.p);
    LAUNCH_CHECK();
    MGF_TRY(prim_exclusive_scan_u32(ctx, w->bflag_l.p, w->bscan_l.p, (size_t)n + 1));
    MGF_TRY(prim_exclusive_scan_u32(ctx, w->bflag_r.p, w->bscan_r.p, (size_t)n + 1));
    k_boundary_scatter<<<nblk(n), kBlock, 0, s>>>(n, w->bflag_l.p, w->bscan_l.p, w->bflag_r.p, w->bscan_r.p, ids_migrants, ids_migrants + ml);
    LAUNCH_CHECK();
  }
  return sync_unless_ordered(w);
}
extern "C" mgf_status mgf_world_export_migrants(mgf_world* w, const uint32_t* ids, int64_t n, float* dst) {
  if (w && w->has_compound) return fail(MGF_ERR_INVALID, "bodies of several parts are supported in single-process worlds only (no ghost / migrant records)");
  if (!w || n < 0 || (n && (!ids || !dst))) return fail(MGF_ERR_INVALID, "bad argument");
  MGF_TRY(ctx_bind(w->ctx));
  if (n > 0) {
    k_export_migrants<<<nblk((size_t)n * kMigrantWords), kBlock, 0, w->ctx->stream>>>(w->bodies(), ids, (uint32_t)n, reinterpret_cast<float4*>(dst));
    LAUNCH_CHECK();
  }
  return sync_unless_ordered(w);
}
// Removes the listed owned bodies (distinct ids, any order); the others keep their relative order, so ids above a
// removed one shift down.  Ghosts of the current tick are dropped.
extern "C" mgf_status mgf_world_remove_bodies(mgf_world* w, const uint32_t* ids, int64_t n_ids) {
  if (w && w->has_compound) return fail(MGF_ERR_INVALID, "bodies of several parts are supported in single-process worlds only (no ghost / migrant records)");
  if (!w || n_ids < 0 || (n_ids && !ids)) return fail(MGF_ERR_INVALID, "bad argument");
  MGF_TRY(ctx_bind(w->ctx));
  mgf_ctx* ctx = w->ctx;
  hipStream_t s = ctx->stream;
  const uint32_t n = w->n_owned;
  w->n = n;
  w->constraints_ready = false;
  if (n_ids == 0) return MGF_OK;
  if ((uint64_t)n_ids > n) return fail(MGF_ERR_INVALID, "more ids than bodies");
  MGF_TRY(w->bflag_l.ensure(n + 1, s)); MGF_TRY(w->bscan_l.ensure(n + 1, s));
  MGF_TRY(w->mig_cnt.ensure(2, s));
  MGF_TRY(w->mig_tmp.ensure((size_t)n * kMigrantWords, s));
  MGF_HIP_TRY(hipMemsetAsync(w->mig_cnt.p, 0, 8, s));
  k_keep_fill<<<nblk(n + 1), kBlock, 0, s>>>(w->bflag_l.p, n);
  LAUNCH_CHECK();
  k_keep_clear<<<nblk(n_ids), kBlock, 0, s>>>(w->bflag_l.p, ids, (uint32_t)n_ids, n, w->mig_cnt.p);
  LAUNCH_CHECK();
  MGF_TRY(prim_exclusive_scan_u32(ctx, w->bflag_l.p, w->bscan_l.p, (size_t)n + 1));
  k_compact_gather<<<nblk((size_t)n * kMigrantWords), kBlock, 0, s>>>(w->bodies(), n, w->bflag_l.p, w->bscan_l.p, w->mig_tmp.p);
  LAUNCH_CHECK();
  uint32_t* pin = static_cast<uint32_t*>(ctx->pinned);
  MGF_HIP_TRY(hipMemcpyAsync(pin, w->mig_cnt.p, 4, hipMemcpyDeviceToHost, s));
  MGF_HIP_TRY(hipStreamSynchronize(s));
  if (pin[0]) return fail(MGF_ERR_INVALID, "remove_bodies: an id is out of range or listed twice");
  const uint32_t n_new = n - (uint32_t)n_ids;
  if (n_new) {
    k_import_migrants<<<nblk((size_t)n_new * kMigrantWords), kBlock, 0, s>>>(w->bodies(), 0, n_new, w->mig_tmp.p);
    LAUNCH_CHECK();
  }
  w->n_owned = w->n = n_new;
  w->stats.n_bodies = n_new;
  return sync_unless_ordered(w);
}
// Appends bodies exported by another world's mgf_world_export_migrants as owned bodies (ghosts are dropped).
extern "C" mgf_status mgf_world_import_migrants(mgf_world* w, const float* src, int64_t n_in) {
  if (w && w->has_compound) return fail(MGF_ERR_INVALID, "bodies of several parts are supported in single-process worlds only (no ghost / migrant records)");
  if (!w || n_in < 0 || (n_in && !src)) return fail(MGF_ERR_INVALID, "bad argument");
  MGF_TRY(ctx_bind(w->ctx));
  w->n = w->n_owned;
  w->constraints_ready = false;
  if (n_in == 0) return MGF_OK;
  size_t need = (size_t)w->n_owned + (size_t)n_in;
  if (need > 0x7FFFFFF0ull) return fail(MGF_ERR_INVALID, "too many bodies");
  MGF_TRY(grow_keep(w, w->x, 1, need)); MGF_TRY(grow_keep(w, w->q, 1, need)); MGF_TRY(grow_keep(w, w->srec, 4, need));
  MGF_TRY(grow_keep(w, w->sp0, 1, need)); MGF_TRY(grow_keep(w, w->sp1, 1, need)); MGF_TRY(grow_keep(w, w->ctor, 1, need));
  MGF_TRY(grow_keep(w, w->imb, 3, need)); MGF_TRY(grow_keep(w, w->delta, 1, need)); MGF_TRY(grow_keep(w, w->einfo, 1, need));
  MGF_TRY(grow_keep(w, w->col0, 1, need)); MGF_TRY(grow_keep(w, w->col1, 1, need)); MGF_TRY(grow_keep(w, w->tb_c, 1, need));
  MGF_TRY(grow_keep(w, w->tb_r, 1, need)); MGF_TRY(grow_keep(w, w->fb_c, 1, need)); MGF_TRY(grow_keep(w, w->fb_r, 1, need));
  k_import_migrants<<<nblk((size_t)n_in * kMigrantWords), kBlock, 0, w->ctx->stream>>>(w->bodies(), w->n_owned, (uint32_t)n_in,
                                                                                      reinterpret_cast<const float4*>(src));
  LAUNCH_CHECK();
  w->n_owned = w->n = (uint32_t)need;
  w->stats.n_bodies = w->n_owned;
  // the arrivals may be of a kind this tile has not seen yet (the narrowphase dispatch is chosen on the host)
  MGF_TRY(w->mig_cnt.ensure(2, w->ctx->stream));
  MGF_HIP_TRY(hipMemsetAsync(w->mig_cnt.p, 0, 8, w->ctx->stream));
  k_kind_mask<<<nblk(n_in), kBlock, 0, w->ctx->stream>>>(w->col1.p, w->n_owned - (uint32_t)n_in, (uint32_t)n_in, w->mig_cnt.p);
  LAUNCH_CHECK();
  uint32_t mask = 0;
  MGF_TRY(d2h(w->ctx, &mask, w->mig_cnt.p, 1));
  if (mask & 1) w->has_sphere = true;
  if (mask & 2) w->has_capsule = true;
  return MGF_OK;
}
// A caller-defined 32-bit tag per body (kept in the constructor record, travels with a migrant): the tiles driver
// stores the body's global id in it.
extern "C" mgf_status mgf_world_set_tags(mgf_world* w, const uint32_t* tags, int64_t n) {
  if (!w || (n && !tags)) return fail(MGF_ERR_INVALID, "NULL argument");
  if ((uint64_t)n != (uint64_t)w->n_owned) return fail(MGF_ERR_INVALID, "one tag per owned body");
  MGF_TRY(ctx_bind(w->ctx));
  if (n == 0) return MGF_OK;
  MGF_TRY(w->bflag_l.ensure((size_t)n + 1, w->ctx->stream));
  MGF_TRY(h2d(w->ctx, w->bflag_l.p, tags, (size_t)n));
  k_tags_set<<<nblk(n), kBlock, 0, w->ctx->stream>>>(w->ctor.p, w->bflag_l.p, (uint32_t)n);
  LAUNCH_CHECK();
  MGF_HIP_TRY(hipStreamSynchronize(w->ctx->stream));
  return MGF_OK;
}
extern "C" mgf_status mgf_world_read_tags(mgf_world* w, uint32_t* tags, int64_t cap) {
  if (!w || (cap && !tags)) return fail(MGF_ERR_INVALID, "NULL argument");
  if ((uint64_t)cap < (uint64_t)w->n_owned) return fail(MGF_ERR_CAPACITY, "tag buffer too small");
  MGF_TRY(ctx_bind(w->ctx));
  const uint32_t n = w->n_owned;
  if (n == 0) return MGF_OK;
  MGF_TRY(w->bflag_l.ensure((size_t)n + 1, w->ctx->stream));
  k_tags_get<<<nblk(n), kBlock, 0, w->ctx->stream>>>(w->ctor.p, w->bflag_l.p, n);
  LAUNCH_CHECK();
  return d2h(w->ctx, tags, w->bflag_l.p, (size_t)n);
}
extern "C" mgf_status mgf_world_export_velocities(mgf_world* w, const uint32_t* ids, int64_t n, float* dst) {
  if (!w || (n && (!ids || !dst))) return fail(MGF_ERR_INVALID, "NULL argument");
  MGF_TRY(ctx_bind(w->ctx));
  if (n > 0) { k_export_vel<<<nblk(n), kBlock, 0, w->ctx->stream>>>(w->srec.p, ids, (uint32_t)n, reinterpret_cast<float4*>(dst)); LAUNCH_CHECK(); }
  return sync_unless_ordered(w);
}
extern "C" mgf_status mgf_world_import_ghost_velocities(mgf_world* w, const float* src, int64_t n_ghost) {
  if (!w || (n_ghost && !src)) return fail(MGF_ERR_INVALID, "NULL argument");
  MGF_TRY(ctx_bind(w->ctx));
  if ((uint64_t)n_ghost != (uint64_t)(w->n - w->n_owned)) return fail(MGF_ERR_INVALID, "ghost count mismatch");
  if (n_ghost > 0) {
    k_import_ghost_vel<<<nblk(n_ghost), kBlock, 0, w->ctx->stream>>>(w->srec.p, w->n_owned, (uint32_t)n_ghost, reinterpret_cast<const float4*>(src));
    LAUNCH_CHECK();
  }
  return sync_unless_ordered(w);
}
extern "C" int64_t mgf_world_ghost_len(const mgf_world* w) { return w ? (int64_t)(w->n - w->n_owned) : 0; }

// development aid: per-node (ready seen, released) timestamps of a dataflow launch -> /tmp/mgf_flow_trace.bin
static mgf_status dump_flow_trace(mgf_world* w, uint32_t C, int32_t iters) {
  std::vector<uint64_t> h(2 * (size_t)iters * C);
  MGF_TRY(d2h(w->ctx, h.data(), w->flow_trace.p, h.size()));
  // block-local solver: the body -> cell-order rank table and the block size tell which block ran a constraint
  const bool f5 = w->opt_solver_mode == 5 && w->flow5_prepped && w->f5_nb > 0;
  std::vector<uint32_t> rank(f5 ? (size_t)w->n + (w->n & 1u) : 0);
  if (f5) MGF_TRY(d2h(w->ctx, rank.data(), w->brank.p, (size_t)w->n));
  if (FILE* f = fopen("/tmp/mgf_flow_trace.bin", "wb")) {
    uint64_t hdr[4] = {C, (uint64_t)iters, f5 ? (uint64_t)rank.size() : 0u, f5 ? (uint64_t)w->f5_nb : 0u};
    fwrite(hdr, 8, 4, f); fwrite(h.data(), 8, h.size(), f);
    if (f5) fwrite(rank.data(), 4, rank.size(), f);
    fclose(f);
  }
  return MGF_OK;
}

// Persistent dataflow launch (modes 1 and 4), enqueue only.  Every lane must be resident, so the grid is sized
// from the occupancy query with one block per CU of margin (the API over-reports by one for some
// kernels on ROCm 7.2).  C is read on the device; `cap_c` bounds the grid.
static mgf_status solve_flow_enqueue(mgf_world* w, int32_t iters, uint32_t cap_c) {
  mgf_ctx* ctx = w->ctx;
  hipStream_t s = ctx->stream;
  const bool kslots = w->opt_solver_mode == 4;
  bool use5 = w->opt_solver_mode == 5 && w->flow5_ok && w->n > 0 && iters <= 100;  // iteration counters are 7-bit in LDS
  if (use5) {
    // blocks of nb bodies in cell order, one workgroup each, all resident: at most one per CU
    const uint32_t need = (w->n + (uint32_t)ctx->num_cus - 1u) / (uint32_t)ctx->num_cus;
    // one block per CU when there are enough bodies (262 144 -> 1024 per block); smaller worlds keep every CU busy with
    // blocks down to 256 bodies (config 5, 65 536 bodies: 0.52 -> 0.45 ms solve)
    uint32_t nb = std::max(need, std::min(w->n, 256u));
    if (w->opt_flow5_block > 0) nb = std::max(need, (uint32_t)w->opt_flow5_block);  // tests: small blocks on small scenes
    if (nb > kF5MaxBodies) use5 = false;
    else { w->f5_nb = nb; w->f5_nblocks = (w->n + nb - 1u) / nb; }
  }
  if (use5 && !w->flow5_prepped) {
    const uint32_t n = w->n;
    MGF_TRY(w->f5_shared.ensure(n / 4 + 1, s)); MGF_TRY(w->f5_gcnt.ensure((size_t)cap_c + 1, s)); MGF_TRY(w->f5_lslot.ensure(std::max(cap_c, 1u), s));
    const size_t rows = (size_t)w->f5_nblocks * kF5MaxCons;
    MGF_TRY(w->f5_wg_cnt.ensure(4 * (size_t)w->f5_nblocks * kF5CntStride, s));
    MGF_TRY(w->f5_table.ensure(rows, s)); MGF_TRY(w->flow_arr5.ensure(rows, s));
    if (!w->flow5_attr_set) {
      const int lds_n = (int)(64 * (size_t)kF5MaxBodies + kF5LdsNarrow), lds_w = (int)(64 * (size_t)kF5MaxBodies + kF5LdsWide);
      MGF_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_solve_flow5<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_n));
      MGF_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_solve_flow5<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_w));
      MGF_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_solve_flow5<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_n));
      MGF_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_solve_flow5<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_w));
      w->flow5_attr_set = true;
    }
    // narrow layout (all slot constants in LDS) while the blocks are small, wide layout once they approach its limit
    w->flow5_wide = w->flow5_last_max > kF5NarrowCons - kF5NarrowCons / 16;
    ZeroList z;
    memset(&z, 0, sizeof(z));
    z.p[0] = w->f5_shared.p; z.words[0] = n / 4 + 1;
    z.p[1] = w->f5_gcnt.p; z.words[1] = cap_c + 1;
    z.p[2] = w->d_err() + 4; z.words[2] = 1;
    z.p[3] = w->f5_wg_cnt.p; z.words[3] = 4 * w->f5_nblocks * kF5CntStride;
    z.p[4] = w->d_err() + 6; z.words[4] = 1;  // largest block of the tick
    k_zero_many<<<64, kBlock, 0, s>>>(z);
    LAUNCH_CHECK();
    Flow5 F = w->flow5();
    const unsigned gc = std::max(1u, nblk(cap_c));
    k_flow5_mark<<<gc, kBlock, 0, s>>>(F, w->links(), &w->sc.p->C);
    LAUNCH_CHECK();
    k_flow5_assign<<<gc, kBlock, 0, s>>>(F, w->links(), &w->sc.p->C);
    LAUNCH_CHECK();
    k_flow5_table<<<gc, kBlock, 0, s>>>(F, w->links(), &w->sc.p->C);
    LAUNCH_CHECK();
    w->flow5_prepped = true;
  }
  int& grid = kslots ? w->flowk_grid : w->flow_grid;
  if (grid == 0) {
    int per_cu = 0;
    if (kslots) MGF_HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_solve_flowk<4, false>, kBlock, 0));
    else MGF_HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_solve_flow<false>, kBlock, 0));
    int want = w->opt_flow_blocks_per_cu > 0 ? (int)w->opt_flow_blocks_per_cu : 4;
    if (w->opt_flow_blocks_per_cu < 0) per_cu = (int)-w->opt_flow_blocks_per_cu;  // experiment: exact value, no occupancy margin
    else per_cu = std::max(1, std::min(per_cu - 1, want));
    if (getenv("MGF_FLOW_DEBUG")) fprintf(stderr, "[mgf] dataflow grid: %d blocks/CU x %d CUs\n", per_cu, ctx->num_cus);
    grid = per_cu * ctx->num_cus;
  }
  const uint32_t* C_ptr = &w->sc.p->C;
  uint32_t* abort_flag = w->d_err() + 2;
  unsigned g = std::min<unsigned>((unsigned)grid, std::max(1u, nblk(cap_c)));
  MGF_TRY(w->flow_arr.ensure(std::max(cap_c, 1u), s));
  k_flow_init<<<std::max(1u, nblk(cap_c)), kBlock, 0, s>>>(C_ptr, w->links(), w->flow_arr.p, abort_flag);
  LAUNCH_CHECK();
  const bool timed = w->opt_time_solver_kernels != 0;
  if (!w->solve_pending) w->kev_used = 0;  // a tiled tick enqueues several launches before it reads the events
  if (timed) {
    while (w->kev.size() < w->kev_used + 2) { hipEvent_t e; MGF_HIP_TRY(hipEventCreate(&e)); w->kev.push_back(e); }
    MGF_HIP_TRY(hipEventRecord(w->kev[w->kev_used], s));
  }
  uint64_t* trace = nullptr;
  if (w->opt_flow_trace) {  // development aid (needs the host-side C: only after a synchronous collide)
    MGF_TRY(w->flow_trace.ensure(2 * (size_t)iters * std::max(w->C, 1u), s));
    trace = w->flow_trace.p;
  }
  const uint32_t spin_limit = 4u << 20;
  const int sleep = (int)w->opt_flow_sleep;
  if (use5) {
    Flow5 F = w->flow5();
    const size_t lds = 64 * (size_t)F.nb + (w->flow5_wide ? kF5LdsWide : kF5LdsNarrow);
    auto kern = w->flow5_wide ? (trace ? k_solve_flow5<true, true> : k_solve_flow5<true, false>)
                              : (trace ? k_solve_flow5<false, true> : k_solve_flow5<false, false>);
    kern<<<F.nblocks, kF5Threads, lds, s>>>(w->srec.p, w->cons_nat.p, w->links(), F, w->flow_arr.p, (uint32_t)iters, abort_flag, spin_limit, trace, w->C);
    LAUNCH_CHECK();
    // stand-by: the global dataflow kernel runs only if a block did not fit its workgroup (flag raised by k_flow5_prep)
    k_solve_flow<false><<<g, kBlock, 0, s>>>(w->srec.p, w->cons_nat.p, w->links(), w->flow_arr.p, C_ptr, (uint32_t)iters, abort_flag, spin_limit, sleep, nullptr, F.fail);
  } else if (kslots) {
    if (trace) k_solve_flowk<4, true><<<g, kBlock, 0, s>>>(w->srec.p, w->cons_nat.p, w->links(), w->flow_arr.p, C_ptr, (uint32_t)iters, abort_flag, spin_limit, sleep, trace);
    else k_solve_flowk<4, false><<<g, kBlock, 0, s>>>(w->srec.p, w->cons_nat.p, w->links(), w->flow_arr.p, C_ptr, (uint32_t)iters, abort_flag, spin_limit, sleep, nullptr);
  } else {
    if (trace) k_solve_flow<true><<<g, kBlock, 0, s>>>(w->srec.p, w->cons_nat.p, w->links(), w->flow_arr.p, C_ptr, (uint32_t)iters, abort_flag, spin_limit, sleep, trace, nullptr);
    else k_solve_flow<false><<<g, kBlock, 0, s>>>(w->srec.p, w->cons_nat.p, w->links(), w->flow_arr.p, C_ptr, (uint32_t)iters, abort_flag, spin_limit, sleep, nullptr, nullptr);
  }
  LAUNCH_CHECK();
  if (timed) { MGF_HIP_TRY(hipEventRecord(w->kev[w->kev_used + 1], s)); w->kev_used += 2; }
  return MGF_OK;
}

// One launch per frontier of the unrolled dependency graph (mode 0): host loop with read-backs.
static mgf_status solve_frontier(mgf_world* w, int32_t iters) {
  mgf_ctx* ctx = w->ctx;
  hipStream_t s = ctx->stream;
  const uint32_t C = w->C;
  const bool timed = w->opt_time_solver_kernels != 0;
  size_t kev_used = 0;
  auto tick = [&](void) -> mgf_status {  // event before/after a solver kernel (option)
    if (!timed) return MGF_OK;
    if (kev_used == w->kev.size()) { hipEvent_t e; MGF_HIP_TRY(hipEventCreate(&e)); w->kev.push_back(e); }
    MGF_HIP_TRY(hipEventRecord(w->kev[kev_used++], s));
    return MGF_OK;
  };
  // frontier lists hold every (constraint, round) once: iters * C entries
  MGF_TRY(w->order.ensure((size_t)iters * C, s));
  if (w->lvl_cap == 0) { w->lvl_cap = 1u << 16; MGF_TRY(w->lvl_off.ensure(w->lvl_cap + 4, s)); }
  Frontier F = w->frontier();
  MGF_HIP_TRY(hipMemsetAsync(w->scalars.p, 0, 12, s));
  k_frontier0<<<std::min<unsigned>(nblk(C), 1024u), kBlock, 0, s>>>(C, w->cons_nat.p, w->links(), F);
  LAUNCH_CHECK();
  uint32_t r = 0;
  // grid: frontiers hold roughly C / (per-iteration depth) constraints; grid-stride covers the rest
  unsigned g0 = std::min<unsigned>(std::max<unsigned>(nblk(C) / 4, 1u), 2048u);
  uint32_t batch = std::max<uint32_t>(w->last_launches + 2, 8u);
  uint32_t* pin = static_cast<uint32_t*>(ctx->pinned);
  for (;;) {
    if (r + batch + 2 > w->lvl_cap) return fail(MGF_ERR_CAPACITY, "constraint dependency graph deeper than 65536 launches");
    for (uint32_t k = 0; k < batch; ++k) {
      MGF_TRY(tick());
      k_solve<<<g0, kBlock, 0, s>>>(w->srec.p, w->cons_nat.p, w->links(), F, r + k, (uint32_t)iters);
      LAUNCH_CHECK();
      MGF_TRY(tick());
      w->stats.solver_kernel_launches++;
    }
    r += batch;
    // launch r's list was filled by the last launch under counter r % 3; empty means everything ran
    MGF_HIP_TRY(hipMemcpyAsync(pin, w->d_cnt() + (r % 3), 4, hipMemcpyDeviceToHost, s));
    MGF_HIP_TRY(hipMemcpyAsync(pin + 1, w->lvl_off.p + r, 4, hipMemcpyDeviceToHost, s));
    MGF_HIP_TRY(hipStreamSynchronize(s));
    if (pin[0] == 0) {
      if ((uint64_t)pin[1] != (uint64_t)iters * C) return fail(MGF_ERR_HIP, "internal error: solver schedule does not cover iters x constraints");
      break;
    }
    batch = 4;
  }
  // number of non-empty launches (for stats and for sizing the next tick's batch)
  std::vector<uint32_t> h(r + 1);
  MGF_TRY(d2h(ctx, h.data(), w->lvl_off.p, r + 1));
  uint32_t used = 0;
  while (used < r && h[used] < (uint64_t)iters * C) ++used;
  w->depth = used;
  w->last_launches = used;
  if (timed) {
    float total = 0.0f, ms;
    for (size_t k = 0; k + 1 < kev_used; k += 2) { MGF_HIP_TRY(hipEventElapsedTime(&ms, w->kev[k], w->kev[k + 1])); total += ms; }
    w->stats.ms_solver_kernels = total;
  }
  return MGF_OK;
}

// After a dataflow launch has been synchronised: abort flag, timings.
static mgf_status solve_flow_finish(mgf_world* w) {
  uint32_t* pin = static_cast<uint32_t*>(w->ctx->pinned);
  MGF_HIP_TRY(hipMemcpyAsync(pin + 80, w->d_err() + 2, 20, hipMemcpyDeviceToHost, w->ctx->stream));  // abort, grid-wide, flow5 fail, terrain-wide, flow5 max block
  MGF_HIP_TRY(hipStreamSynchronize(w->ctx->stream));
  if (pin[80]) return fail(MGF_ERR_HIP, "dataflow solver gave up waiting (grid not fully resident?)");
  if (w->opt_solver_mode == 5 && w->flow5_prepped) {
    if (pin[82]) w->n_flow5_fallbacks++;
    w->flow5_last_max = pin[84];
  }
  w->stats.solver_kernel_launches = 1;
  w->depth = 1;
  if (w->opt_time_solver_kernels) {
    float ms, total = 0.0f;
    for (size_t k = 0; k + 1 < w->kev_used; k += 2) { MGF_HIP_TRY(hipEventElapsedTime(&ms, w->kev[k], w->kev[k + 1])); total += ms; }
    w->stats.ms_solver_kernels = total;
  }
  return MGF_OK;
}

// Solver::solve solver.rs:72-78 (exact sequential order, see kernels.h).
extern "C" mgf_status mgf_world_solve(mgf_world* w, int32_t iters, mgf_step_stats* stats) {
  if (!w) return fail(MGF_ERR_INVALID, "world is NULL");
  if (iters < 0) return fail(MGF_ERR_INVALID, "iters must be >= 0");
  MGF_TRY(ctx_bind(w->ctx));
  if (!w->constraints_ready) return fail(MGF_ERR_INVALID, "no constraint list: call mgf_world_build_constraints or mgf_world_set_constraints first");
  hipStream_t s = w->ctx->stream;
  w->stats.iters = (uint32_t)iters;
  w->stats.solver_kernel_launches = 0;
  w->stats.ms_solver_kernels = 0.0f;
  w->depth = 0;
  MGF_HIP_TRY(hipEventRecord(w->ev[5], s));
  if (w->C > 0 && iters > 0) {
    if (w->opt_solver_mode == 0) {
      MGF_TRY(solve_frontier(w, iters));
    } else {
      MGF_TRY(solve_flow_enqueue(w, iters, w->C));
      if (w->opt_flow_trace) MGF_TRY(dump_flow_trace(w, w->C, iters));
      MGF_HIP_TRY(hipEventRecord(w->ev[6], s));
      MGF_TRY(solve_flow_finish(w));
    }
  }
  if (w->opt_solver_mode == 0 || !(w->C > 0 && iters > 0)) MGF_HIP_TRY(hipEventRecord(w->ev[6], s));
  MGF_HIP_TRY(hipStreamSynchronize(s));
  float ms;
  MGF_HIP_TRY(hipEventElapsedTime(&ms, w->ev[5], w->ev[6]));
  w->stats.ms_solve = ms;
  w->stats.n_levels = w->depth;
  if (stats) *stats = w->stats;
  return MGF_OK;
}

// Solver::solve without the read-back: the launch is enqueued on the ctx stream and its outcome is checked by
// the next mgf_world_finish (a tiled driver interleaves single iterations with ghost velocity exchanges
// on the same stream).  Solver mode 0 has a host loop and runs synchronously here.
extern "C" mgf_status mgf_world_solve_enqueue(mgf_world* w, int32_t iters) {
  if (!w) return fail(MGF_ERR_INVALID, "world is NULL");
  if (iters < 0) return fail(MGF_ERR_INVALID, "iters must be >= 0");
  if (w->opt_solver_mode == 0 || w->opt_flow_trace) return mgf_world_solve(w, iters, nullptr);
  MGF_TRY(ctx_bind(w->ctx));
  if (!w->constraints_ready) return fail(MGF_ERR_INVALID, "no constraint list: call mgf_world_build_constraints or mgf_world_set_constraints first");
  if (!w->solve_pending) {
    MGF_HIP_TRY(hipEventRecord(w->ev[5], w->ctx->stream));
    w->stats.iters = 0;
    w->stats.solver_kernel_launches = 0;
    w->stats.ms_solver_kernels = 0.0f;
  }
  if (w->C > 0 && iters > 0) {
    MGF_TRY(solve_flow_enqueue(w, iters, w->C));
    w->stats.solver_kernel_launches++;
  }
  w->stats.iters += (uint32_t)iters;
  w->solve_pending = true;
  return MGF_OK;
}
// Synchronise the ctx stream and report what the enqueued work did (solver abort flag, timings).
extern "C" mgf_status mgf_world_finish(mgf_world* w, mgf_step_stats* stats) {
  if (!w) return fail(MGF_ERR_INVALID, "world is NULL");
  MGF_TRY(ctx_bind(w->ctx));
  hipStream_t s = w->ctx->stream;
  if (w->solve_pending) {
    w->solve_pending = false;
    MGF_HIP_TRY(hipEventRecord(w->ev[6], s));
    uint32_t launches = w->stats.solver_kernel_launches;
    if (launches) MGF_TRY(solve_flow_finish(w)); else MGF_HIP_TRY(hipStreamSynchronize(s));
    w->stats.solver_kernel_launches = launches;
    float ms;
    MGF_HIP_TRY(hipEventElapsedTime(&ms, w->ev[5], w->ev[6]));
    w->stats.ms_solve = ms;
    w->stats.n_levels = launches;
    MGF_HIP_TRY(hipEventElapsedTime(&ms, w->ev[0], w->ev[6]));
    w->stats.ms_total = ms;
  } else {
    MGF_HIP_TRY(hipStreamSynchronize(s));
  }
  if (stats) *stats = w->stats;
  return MGF_OK;
}

// World::step world.rs:227-294.  With a dataflow solver the whole tick is enqueued without a read-back
// and synchronised once at the end; a capacity miss re-runs the collide phase (the solver was a no-op).
extern "C" mgf_status mgf_world_step(mgf_world* w, float dt, int32_t iters, mgf_step_stats* stats) {
  if (!w) return fail(MGF_ERR_INVALID, "world is NULL");
  if (iters < 0) return fail(MGF_ERR_INVALID, "iters must be >= 0");
  if (w->opt_solver_mode == 0 || w->opt_flow_trace || w->opt_debug_bvh) {
    MGF_TRY(mgf_world_build_constraints(w, dt, nullptr));
    MGF_TRY(mgf_world_solve(w, iters, nullptr));
  } else {
    MGF_TRY(world_begin(w, dt));
    hipStream_t s = w->ctx->stream;
    bool done = false;
    for (int attempt = 0; attempt < 8 && !done; ++attempt) {
      bool retry = false;
      MGF_TRY(collide_enqueue(w, dt));
      MGF_HIP_TRY(hipEventRecord(w->ev[5], s));
      if (w->n > 0 && iters > 0) MGF_TRY(solve_flow_enqueue(w, iters, w->cap_c));
      MGF_HIP_TRY(hipEventRecord(w->ev[6], s));
      MGF_TRY(collide_finish(w, &retry));
      done = !retry;
    }
    if (!done) return fail(MGF_ERR_HIP, "internal error: collide phase did not settle its buffer capacities");
    w->stats.iters = (uint32_t)iters;
    w->stats.solver_kernel_launches = 0;
    w->stats.ms_solver_kernels = 0.0f;
    w->depth = 0;
    if (w->C > 0 && iters > 0) MGF_TRY(solve_flow_finish(w));
    float ms;
    MGF_HIP_TRY(hipEventElapsedTime(&ms, w->ev[5], w->ev[6]));
    w->stats.ms_solve = ms;
    w->stats.n_levels = w->depth;
  }
  float ms;
  MGF_HIP_TRY(hipEventElapsedTime(&ms, w->ev[0], w->ev[6]));
  w->stats.ms_total = ms;
  if (stats) *stats = w->stats;
  return MGF_OK;
}

static void crec_to_public(const CRec& c, mgf_constraint* o) {
  o->a = (int32_t)c.a;
  o->b = c.b == kNone ? -1 : (int32_t)c.b;
  o->n_contacts = 1;
  o->normal = {c.n[0], c.n[1], c.n[2]}; o->t0 = {c.t0[0], c.t0[1], c.t0[2]}; o->t1 = {c.t1[0], c.t1[1], c.t1[2]};
  o->ra = {c.ra[0], c.ra[1], c.ra[2]}; o->rb = {c.rb[0], c.rb[1], c.rb[2]};
  o->bias = c.bias; o->normal_mass = c.nmass; o->tangent_mass0 = c.tmass0; o->tangent_mass1 = c.tmass1;
  o->normal_impulse = c.nimp; o->friction = c.friction;
}
extern "C" mgf_status mgf_world_read_constraints(mgf_world* w, mgf_constraint* out, int64_t cap, int64_t* count) {
  if (!w) return fail(MGF_ERR_INVALID, "world is NULL");
  MGF_TRY(ctx_bind(w->ctx));
  if (count) *count = w->C;
  if (!out) return MGF_OK;
  if ((int64_t)w->C > cap) return fail(MGF_ERR_CAPACITY, "constraint buffer too small");
  if (w->C == 0) return MGF_OK;
  std::vector<CRec> h(w->C);
  MGF_TRY(d2h(w->ctx, h.data(), w->cons_nat.p, w->C));
  for (uint32_t i = 0; i < w->C; ++i) crec_to_public(h[i], &out[i]);
  return MGF_OK;
}
extern "C" mgf_status mgf_world_set_constraints(mgf_world* w, const mgf_constraint* cons, int64_t n) {
  if (!w || (n && !cons)) return fail(MGF_ERR_INVALID, "NULL argument");
  MGF_TRY(ctx_bind(w->ctx));
  if (n < 0 || n >= 0x7FFFFFF0ll) return fail(MGF_ERR_INVALID, "bad constraint count");
  std::vector<CRec> h((size_t)n);
  for (int64_t i = 0; i < n; ++i) {
    const mgf_constraint& c = cons[i];
    if (c.n_contacts != 1) return fail(MGF_ERR_INVALID, "only single-contact manifolds are supported on this path");
    if (c.a < 0 || (uint32_t)c.a >= w->n) return fail(MGF_ERR_STATIC_REF, "obj_a must be a Dynamic body in range");
    if (c.b >= 0 && (uint32_t)c.b >= w->n) return fail(MGF_ERR_INVALID, "obj_b out of range");
    if (c.a == c.b) return fail(MGF_ERR_INVALID, "a constraint needs two different bodies");
    CRec r;
    memset(&r, 0, sizeof(r));
    r.a = (uint32_t)c.a; r.b = c.b < 0 ? kNone : (uint32_t)c.b;
    const float* src[5] = {&c.normal.x, &c.t0.x, &c.t1.x, &c.ra.x, &c.rb.x};
    float* dst[5] = {r.n, r.t0, r.t1, r.ra, r.rb};
    for (int k = 0; k < 5; ++k) memcpy(dst[k], src[k], 12);
    r.bias = c.bias; r.nmass = c.normal_mass; r.tmass0 = c.tangent_mass0; r.tmass1 = c.tangent_mass1; r.nimp = c.normal_impulse;
    r.friction = c.friction;
    h[(size_t)i] = r;
  }
  w->C = (uint32_t)n; w->Ct = 0;
  MGF_TRY(w->cons_nat.ensure(std::max<size_t>((size_t)n, 1), w->ctx->stream));
  MGF_TRY(h2d(w->ctx, w->cons_nat.p, h.data(), (size_t)n));
  memset(&w->stats, 0, sizeof(w->stats));
  w->stats.n_bodies = w->n; w->stats.n_constraints = w->C;
  StepCounts hc;
  memset(&hc, 0, sizeof(hc));
  hc.C = hc.need_C = w->C;
  MGF_TRY(h2d(w->ctx, w->sc.p, &hc, 1));
  w->flow5_ok = false; w->flow5_prepped = false;  // a caller's list has no per-body blocks
  MGF_TRY(links_ensure(w, w->C, true));
  if (w->C) {
    k_links_from_records<<<nblk(w->C), kBlock, 0, w->ctx->stream>>>(w->cons_nat.p, w->C, w->c_ab.p, w->deg.p);
    LAUNCH_CHECK();
  }
  MGF_TRY(build_dag(w, w->C));
  w->constraints_ready = true;
  MGF_HIP_TRY(hipStreamSynchronize(w->ctx->stream));
  return MGF_OK;
}
