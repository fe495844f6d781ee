// Hand-written HIP kernels of the mgf per-tick hot path for gfx950 (wave64).
//
//   k_integrate<Tail>  RigidBodyVec::complete_motion + integrate (physics.rs:222-269), swept AABB
//                      (bounds.rs:60-68), fat-AABB refit test (world.rs:234-238).  In mgf_world_step (collide follows at
//                      once on the same bodies) its tail also lists each body's terrain faces (what k_terrain_rows does)
//                      and the blocks gather the scene bounds (what k_scene_bounds does; folded by k_zero_many); in the fused tick
//                      (r05) it also works out the body's Morton cell and rank over the previous tick's bounds (CellSort: no
//                      k_morton_count launch) and appends a 48-byte record of every body that lists a terrain face
//   k_scene_bounds / k_morton_count / k_scatter_leaves
//                      bodies counting-sorted into Morton cells (cell = 2L-bit prefix of the 30-bit code of the fat-box
//                      centre); the same kernels build the static grid over a terrain mesh's face boxes
//   k_scan<W>          the tick's prefix sums (cells, candidate rows, constraints per body): single pass, ticketed tiles, wave-wide
//                      decoupled look-back; its last thread checks the list capacities (caps_candidates / caps_constraints)
//   k_pair_brick       BVH::query (bvh.rs:283-310) of the world tree for every body at once: a brick of 4x4x4 Morton cells per
//                      block, the 8x8x8 cells around it staged in LDS, 8 lanes per query; queries that do not fit, and
//   k_pair_grid        the same by cell enumeration from global memory (the fallback, and what tiles with thin cells run);
//                      k_lbvh_low / k_lbvh_top + k_pair_rows (implicit 4-ary tree over the cells, cooperative walk)
//                      when one body spans too many cells; the hit SET is the reference's because acceptance is its own
//                      predicate on the leaf boxes
//   k_terrain_grid     Mesh::contacts' BVH::query by cell enumeration over the face boxes (row-major cells, bits per axis chosen
//                      from the mesh by build_face_grid: a heightfield's vertical axis gets none), hits stored as DFS ranks;
//                      k_terrain_rows = the walk of the flattened reference tree in the reference's order
//   k_candidates<FILL> the exact two-pass (count, fill) tree walk used when a candidate row overflows
//   k_rows_to_csr      rows -> CSR, terrain faces in DFS order (partner contacts are ordered by k_count_contacts)
//   k_narrow_pairs<A,B> / k_narrow_terrain<A>
//                      one kernel per shape-pair type over the candidate lists; pairs whose bounding spheres never come within
//                      reach leave first (comp_pair_far) and a block's survivors are packed into its first lanes
//   k_count_contacts / k_setup_pairs<SPHERES> (its first blocks: the terrain candidates, setup_terrain_one)
//                      Manifold::from + ContactConstraint::new (manifold.rs:120-148, solver.rs:101-191); in a world of
//                      spheres only the broadphase lists contacts (k_pair_grid<true> runs the sphere test) and
//                      k_setup_pairs<true> evaluates each one itself, so k_narrow_pairs is not launched
//   k_terrain_contacts / k_contacts_spheres (r05)
//                      a world of spheres over a small mesh, no candidate lists: the sphere-triangle tests of the bodies near the
//                      mesh (records from k_integrate's tail; a face by four lanes, tri_msphere_x4; as the last blocks of
//                      k_scatter_leaves_tc's launch), k_scan over p_cnt + tcn with the tick's counts in its epilogue
//                      (caps_contacts), then a block per 256 bodies: its partner contacts as an LDS list in canonical order,
//                      ContactConstraint::new for each
//   k_near_list / k_terrain_near<L> / k_terrain_tests / k_pair_grid_n<PARTS> / k_contacts_rows<false> / k_contacts_rows_parts (r06, k_front_rows.h)
//                      the same list-free front end for worlds that are NOT spheres only - capsules, mixed kinds, bodies of two
//                      components: the bodies near the mesh, their faces from the face grid with a cheap conservative reject
//                      (comp_tri_far) and a slot each, the body-triangle tests a lane per slot (on the context's second stream,
//                      beside the pair search); the pair search pools the accepted partners of a block's queries that may touch
//                      and runs the pair test - or the two-part manifold, raw contacts staged in LDS - a lane each, the rows
//                      hold contacts only; records from the rows.  k_contacts_rows<true> is r05's k_contacts_spheres.
//   k_pair_wide (r06)  the few bodies whose fat box is far larger than the rest's (WideSpec, k_bodies.h: kept out of the scene
//                      bounds and rmax by k_integrate, never partners of the grid's pair search) find their partners-to-be
//                      from their own side: a workgroup per listed body
//   k_narrow_pairs_big / k_narrow_terrain_big (r06)
//                      worlds with a body of 5..32 components (its parts in the world's pool: Bodies::xl0): a wave per candidate pair of
//                      bodies, its lanes the part pairs in the oracle's order behind the bounding-sphere reject, the contacts packed in
//                      lane order into LDS and the pruner run by lane 0; a wave per (body, face), a lane per part behind comp_tri_far
//   k_tri_reject_batch the test entry mgf_tri_reject_batch: comp_tri_far's verdict beside the reference's contact count, per problem
//   k_solver_snapshot / k_solver_restore (r05)
//                      the velocities / impulses a persistent solver launch finds, and back, if it gives up (solver_abort_fallback)
//   k_chain_rows       order-preserving dependency links of the tick's constraint list (compact arrays, ConsLinks);
//   k_adj_fill / k_chain  the same for a caller-supplied list in any order (mgf_world_set_constraints)
//   k_flow6_blocks / k_flow6_links / k_flow6_chan
//                      block tables of the default solver: slots, foreign bodies, rows; the links along every body's chain
//                      written straight into the rows with their message channels; channel layout
//   k_solve_flow6      ContactConstraint::solve (solver.rs:203-252) for a whole Solver::solve call: one workgroup per spatial
//                      block, every body it touches, the arrival counters, the ready queue and the accumulated impulses in LDS,
//                      edges across block faces as 48-byte messages polled by one wave (solver mode 6, the default)
//   k_solve_flow5      ContactConstraint::solve (solver.rs:203-252) for a whole Solver::solve call: block-local persistent
//                      dataflow launch (a spatial block's velocities, arrival counters and ready queues in LDS)
//   k_solve_flow       the same graph with every hand-off through L2 (stand-by of k_solve_flow5, solver mode 1)
//   k_frontier0 / k_solve
//                      one launch per frontier of the unrolled (iterations x constraints) graph (solver mode 0, cross-check)
//   k_tile_select_* / k_export_bodies / k_import_ghosts / k_export_vel / k_import_ghost_vel / k_fetch_flags (k_tiles.h; batched in r05)
//                      the tile protocol: the boundary bodies and migrants of up to eight tiles per launch (per-block counts,
//                      offsets by one workgroup per tile, ordered scatter), 72-float ghost records out and in, the velocity refreshes
//                      between solver launches, every tile's solver flags into its pinned block; k_remove_positions_short /
//                      k_flag_* + k_compact_gather / _put: the stable compaction behind a hand-over; k_tick_snapshot: what a
//                      tile's tick changes, kept for the retry of a tick in which a persistent launch gave up
//   k_compound_* / k_bvh_raytrace / k_intersections_batch
//                      Compound (compound.rs:230-352), BVH::raytrace and Intersects (bvh.rs:345-369, collision.rs:169-373)
//
// All f32 arithmetic follows the reference's operation order; the TU is built with
// -ffp-contract=off.
#pragma once
// The kernels live in the k_*.h parts, each including the one before it:
//   k_bodies.h -> k_broadphase.h -> k_contacts.h -> k_front_rows.h -> k_links.h -> k_solver_flow.h -> k_tiles.h -> k_api.h
#include "k_api.h"
