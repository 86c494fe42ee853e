/* rkh_diag.h -- profiling and diagnostic entry points of librkh.so.
 *
 * NOT part of the drop-in boundary (rkh.h): nothing here replaces a reference interface.  bench.py uses them to time
 * kernels with HIP events on the launch stream, the tests/diag_*.py scripts to read per-phase cycle counts.
 */
#ifndef RKH_DIAG_H
#define RKH_DIAG_H

#include "rkh.h"

#ifdef __cplusplus
extern "C" {
#endif

/* One-shot: the next rkh_nn_query1_async records the two hipEvent_t (passed as void*) immediately before and after
 * its sweep kernel on the context stream (bench.py times the kernel itself, not the launch sequence). */
rkh_status rkh_nn_set_events(rkh_nn* nn, void* ev_start, void* ev_stop);
/* Name of the sweep kernel the last query launched (for profile bookkeeping). */
const char* rkh_nn_kernel_name(void);

/* Shader-clock cycles of `iters` back-to-back f-evals + proximity tests.  One-wave-per-edge kernel: one wave per state,
 * cycles[B][8] = {sincos, forward sweep, jacobian columns, force sweep, mass matrix, cholesky, proximity, total};
 * two-lanes-per-edge kernels (RKH_LANES_PER_EDGE = 1 | 2 in the environment): one record per wave of 32 states,
 * {frames + sincos, jacobian columns, mass matrix, force sweep, (assembly +) cholesky, proximity: joint frames, cull
 * (+ queueing), closed forms}. */
rkh_status rkh_diag_feval_cycles(rkh_scene* scene, const double* x, const double* u, uint32_t B, int iters,
                                 uint64_t* cycles);

/* GJK distance of n world-anchored shape pairs (a[i], b[i]), any kinds GJK knows (sphere, capped cylinder, box, mesh):
 * the check of the support-map query against the closed forms on primitive pairs. */
rkh_status rkh_diag_gjk_distance(rkh_ctx* ctx, const rkh_shape* a, const rkh_shape* b, uint32_t n,
                                 const double* mesh_vertices, uint32_t n_mesh_vertices, double* dist);

/* The planner-regime 1-NN sweep (half-precision mirror of the vertex rows + exact resolution of the one or two rows the
 * estimate leaves, reak_amd/csrc/nn_mirror.hip) on a caller's point cloud: n points [n][D], B queries [B][D], every
 * |coordinate| <= coord_bound (1e-3 .. 32), D <= 12.  idx / dist: the nearest point of each query, first minimum wins
 * (min_dist_linear_search, topological_search.hpp:95-118), bit-identical to the fp64 sweeps. */
rkh_status rkh_diag_nn_mirror_query(rkh_ctx* ctx, const double* pts, uint64_t n, int D, const double* q, uint32_t B,
                                    double coord_bound, uint32_t* idx, double* dist);

/* With RKH_PROFILE_NN=1 in the environment at rkh_planner_create, every round brackets its NN sweep kernel with
 * HIP events on the planner stream: total kernel time, algorithmic bytes (n*D*8 per sweep) and launch count. */
rkh_status rkh_planner_nn_profile(rkh_planner* p, double* total_ms, uint64_t* total_bytes, uint64_t* launches);
/* Same profile: (vertex, query) pairs the profiled sweeps evaluated = sum over rounds and problems of n * B. */
rkh_status rkh_planner_nn_pairs(rkh_planner* p, uint64_t* pairs);
/* Same switch: HIP events around the steer launches (both kernel mappings) of every round: total time, rounds. */
rkh_status rkh_planner_steer_profile(rkh_planner* p, double* total_ms, uint64_t* launches);
/* Always on: RK4 steps the steer kernels of this planner integrated so far, over all edges (a step counts when it
 * starts from a live edge, the step that ends the edge included; steps skipped because the edge had ended do not).
 * The executed work of the steer launches -- an edge is launched for n_steps but stops at its first state that is not
 * free (MEAQR_topology.hpp:550-559).  (The first-generation lane kernel, RKH_LANE_VARIANT=1, does not count.) */
rkh_status rkh_planner_steer_steps(rkh_planner* p, uint64_t* executed_steps);

/* The proximity test of the two-lanes steer kernels on B states (2 n_dof doubles each), counting what reaches each of
 * its stages ("collide"): counts[0] = states tested, [1] = (robot shape, obstacle) pairs that pass the
 * static reach and the bounding cull, [2] = closed forms evaluated, [3] = golden-section searches (capped cylinder /
 * box), [4] = states found in collision, [5] = proxy pairs of the scene (the tests per state before any culling),
 * [6] = those within the shapes' static reach.  Scenes of the two-lanes mapping only (serial chains of <= 7 joints). */
rkh_status rkh_diag_proximity_counts(rkh_scene* scene, const double* x, uint32_t B, uint64_t counts[8]);

#ifdef __cplusplus
}
#endif
#endif
