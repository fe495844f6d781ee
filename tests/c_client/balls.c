/* A C99 client of include/mgf_hip.h, linked against libmgf_hip.so: the layer a Rust `extern "C"` block binds (by-value structs,
 * callbacks, opaque handles, status codes) - INTEGRATION.md's shim cannot be compiled in this image (no rustc), this can.
 * It assembles what mgf_demo/balls.rs:67-96 assembles through the reference's API - the terrain of mgf_demo/world.rs:118-150 vertex by
 * vertex and face by face (Mesh::push_vert / push_face), `num`^3 balls (World::add_body), a BVH<AABB, usize> over their fat boxes
 * queried with a closure (BVH::insert / query) - steps the world (World::step) and writes the states after 1, 2, 10, 60 and 300 ticks
 * as raw f32 for tests/test_c_client.py to compare with tests/golden/world_snapshots.npz.
 *   usage: balls <out.bin> <num> <iters> <order: 0 canonical | 1 world.rs>      */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mgf_hip.h"

#define CHECK(expr)                                                                                   \
  do {                                                                                                \
    mgf_status st_ = (expr);                                                                          \
    if (st_ != MGF_OK) { fprintf(stderr, "%s -> %d: %s\n", #expr, (int)st_, mgf_last_error()); return 1; } \
  } while (0)

typedef struct hits { uint64_t count, sum; } hits;
static void on_hit(const uint64_t* val, void* user) {  /* the FnMut(&V) of BVH::query */
  hits* h = (hits*)user;
  h->count += 1;
  h->sum += *val;
}

int main(int argc, char** argv) {
  if (argc < 5) { fprintf(stderr, "usage: balls <out.bin> <num> <iters> <order>\n"); return 2; }
  const int num = atoi(argv[2]), iters = atoi(argv[3]), order = atoi(argv[4]);
  mgf_ctx* ctx = NULL;
  CHECK(mgf_ctx_create(0, &ctx));

  /* terrain: world.rs:118-150 - 8 vertices, 10 faces, open-top box of half width 10 and height 10 at (0, -10, 0) */
  mgf_mesh* mesh = NULL;
  CHECK(mgf_mesh_new(ctx, &mesh));
  const float h = 10.0f, H = 10.0f;
  const mgf_vec3 verts[8] = {{-h, 0, -h}, {-h, 0, h}, {h, 0, h}, {h, 0, -h}, {-h, H, -h}, {-h, H, h}, {h, H, h}, {h, H, -h}};
  const uint64_t faces[10][3] = {{0, 1, 3}, {1, 2, 3}, {0, 5, 1}, {0, 4, 5}, {0, 3, 7}, {0, 7, 4}, {2, 6, 3}, {3, 6, 7}, {1, 5, 2}, {2, 5, 6}};
  for (int i = 0; i < 8; ++i) { uint64_t id; CHECK(mgf_mesh_push_vert(mesh, verts[i], &id)); if (id != (uint64_t)i) return 3; }
  for (int i = 0; i < 10; ++i) { uint64_t id; CHECK(mgf_mesh_push_face(mesh, faces[i][0], faces[i][1], faces[i][2], &id)); if (id != (uint64_t)i) return 3; }
  const mgf_vec3 pos = {0.0f, -10.0f, 0.0f};
  CHECK(mgf_mesh_set_pos(mesh, pos));

  /* balls.rs:74-92 */
  const int64_t n = (int64_t)num * num * num;
  mgf_component* comps = (mgf_component*)calloc((size_t)n, sizeof(mgf_component));
  float* mass = (float*)malloc(sizeof(float) * (size_t)n);
  float* rest = (float*)malloc(sizeof(float) * (size_t)n);
  float* fric = (float*)malloc(sizeof(float) * (size_t)n);
  mgf_vec3* force = (mgf_vec3*)malloc(sizeof(mgf_vec3) * (size_t)n);
  const float rad = 0.5f, shift = 2.5f * rad;
  const float centerx = shift * (float)num / 2.0f, centery = shift * (float)num / 2.0f;
  int64_t b = 0;
  for (int i = 0; i < num; ++i)
    for (int j = 0; j < num; ++j)
      for (int k = 0; k < num; ++k, ++b) {
        comps[b].tag = MGF_SPHERE;
        comps[b].p.x = (float)i * 2.5f * rad - centerx;
        comps[b].p.y = 10.0f + (float)j * 2.5f * rad + centery * 2.0f;
        comps[b].p.z = (float)k * 2.5f * rad - centerx;
        comps[b].r = rad;
        mass[b] = 1.0f; rest[b] = 0.3f; fric[b] = 0.6f;
        force[b].x = 0.0f; force[b].y = -9.8f; force[b].z = 0.0f;
      }

  mgf_world* world = NULL;
  CHECK(mgf_world_new(ctx, NULL, &world));
  CHECK(mgf_world_set_terrain(world, mesh));
  uint64_t first = 99;
  CHECK(mgf_world_add_bodies(world, comps, n, mass, rest, fric, force, &first));
  if (first != 0 || mgf_world_len(world) != n) return 4;
  if (order) CHECK(mgf_world_set_option(world, "constraint_order", 1));

  /* a BVH<AABB, usize> over the balls' fat boxes (World::add_body world.rs:178-184), queried with a callback */
  mgf_bvh* bvh = NULL;
  CHECK(mgf_bvh_with_capacity(ctx, (uint64_t)n, &bvh));
  if (!mgf_bvh_empty(bvh)) return 5;
  for (int64_t i = 0; i < n; ++i) {
    mgf_aabb box;
    box.c = comps[i].p;
    box.r.x = box.r.y = box.r.z = rad + 0.25f;
    uint64_t id;
    CHECK(mgf_bvh_insert(bvh, &box, (uint64_t)i, &id));
  }
  mgf_aabb q;
  q.c.x = 0.0f; q.c.y = comps[0].p.y + 2.0f; q.c.z = 0.0f;
  q.r.x = 2.0f; q.r.y = 1.0f; q.r.z = 3.0f;
  hits hq = {0, 0};
  CHECK(mgf_bvh_query(bvh, &q, on_hit, &hq));
  uint64_t root;
  CHECK(mgf_bvh_root(bvh, &root));
  mgf_aabb rb;
  CHECK(mgf_bvh_bounds(bvh, root, &rb));

  FILE* out = fopen(argv[1], "wb");
  if (!out) return 6;
  const int targets[5] = {1, 2, 10, 60, 300};
  mgf_vec3* x = (mgf_vec3*)malloc(sizeof(mgf_vec3) * (size_t)n);
  mgf_quat* qq = (mgf_quat*)malloc(sizeof(mgf_quat) * (size_t)n);
  mgf_vec3* v = (mgf_vec3*)malloc(sizeof(mgf_vec3) * (size_t)n);
  mgf_vec3* om = (mgf_vec3*)malloc(sizeof(mgf_vec3) * (size_t)n);
  int tick = 0;
  for (int t = 0; t < 5; ++t) {
    mgf_step_stats st;
    memset(&st, 0, sizeof(st));
    while (tick < targets[t]) { CHECK(mgf_world_step(world, 1.0f / 60.0f, iters, &st)); ++tick; }
    CHECK(mgf_world_read_state(world, x, qq, v, om, NULL, n));
    const uint64_t head[2] = {(uint64_t)tick, st.n_constraints};
    fwrite(head, sizeof(uint64_t), 2, out);
    fwrite(x, sizeof(mgf_vec3), (size_t)n, out);
    fwrite(qq, sizeof(mgf_quat), (size_t)n, out);
    fwrite(v, sizeof(mgf_vec3), (size_t)n, out);
    fwrite(om, sizeof(mgf_vec3), (size_t)n, out);
  }
  fclose(out);
  /* ConstrainedSet::get through the by-value structs (physics.rs:272-288) */
  mgf_body_ref ref;
  memset(&ref, 0, sizeof(ref));
  ref.tag = 0; ref.index = 7;
  mgf_velocity vel;
  mgf_rigid_body_info info;
  CHECK(mgf_world_get(world, &ref, &vel, &info));
  printf("bvh_hits %llu bvh_hit_sum %llu root_r %.9g %.9g %.9g body7 %.9g %.9g %.9g inv_mass %.9g\n", (unsigned long long)hq.count,
         (unsigned long long)hq.sum, rb.r.x, rb.r.y, rb.r.z, vel.linear.x, vel.linear.y, vel.linear.z, info.inv_mass);
  mgf_bvh_free(bvh);
  mgf_world_free(world);
  mgf_mesh_free(mesh);
  mgf_ctx_destroy(ctx);
  free(comps); free(mass); free(rest); free(fric); free(force); free(x); free(qq); free(v); free(om);
  return 0;
}
