/* rkh.h -- C-ABI of the MI355X-native planning hot path (librkh.so).
 *
 * ReaK itself has no FFI for this path: the hot path sits behind C++ template concepts
 * (SURVEY.md 8(b)).  These entry points are what a ReaK-side adaptor binds (INTEGRATION.md shows
 * the C++ classes modelling ReaK's NNFinder / Steerable C_free / planner concepts on top of them).
 * Each function names the reference interface it replaces (paths relative to
 * /root/reference/src/ReaK/).
 *
 * Conventions: every function returns an rkh_status (0 = ok, negative = error); no exception
 * crosses the boundary; the caller owns every host buffer; one host thread per rkh_ctx; calls are
 * synchronous (they return after the device work finished) unless named *_async.
 * Pointers named d_* are device (HBM) pointers, all others are host pointers.
 * Profiling and diagnostic entry points (kernel timing for bench.py, per-phase cycle counts) are NOT part of this
 * boundary: they live in rkh_diag.h.  The C++ classes that model ReaK's concepts over these functions are in
 * rkh_adaptors.hpp.
 */
#ifndef RKH_H
#define RKH_H

#include <stddef.h>
#include <stdint.h>

#include "rkh_types.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef enum rkh_status {
  RKH_OK = 0,
  RKH_ERR_BAD_ARG = -1,     /* std::range_error in the reference (kte_nl_system.hpp:181-188) */
  RKH_ERR_OOM = -2,
  RKH_ERR_SINGULAR = -3,    /* singularity_error (mat_cholesky.hpp:80-82, manip_dynamics_model.cpp:206-214) */
  RKH_ERR_DEVICE = -4,      /* HIP runtime error; rkh_last_error() has the text */
  RKH_ERR_UNSUPPORTED = -5, /* KTE program / shape not covered by the HIP kernels */
  RKH_ERR_CAPACITY = -6
} rkh_status;

typedef struct rkh_ctx rkh_ctx;         /* one device + stream */
typedef struct rkh_nn rkh_nn;           /* device-resident vertex set for NN queries */
typedef struct rkh_scene rkh_scene;     /* KTE chain + proxy environment on the device */
typedef struct rkh_planner rkh_planner; /* batched RRT driver over a scene */
typedef struct rkh_rrtstar rkh_rrtstar; /* batched RRT* driver over a scene */
typedef struct rkh_prm rkh_prm;         /* batched PRM driver over a scene */
typedef struct rkh_birrt rkh_birrt;     /* batched bidirectional-RRT driver over a scene */

const char* rkh_last_error(void);
const char* rkh_version(void);
/* ABI number of this header: bumped whenever a public POD changes size or meaning (2: rkh_rrtstar_stats gained
 * pruned / skipped, rkh_qs_space gained speed_limits; 3: this handshake).  A binding checks once after loading:
 *   rkh_abi_check(RKH_ABI_VERSION, sizeof(rkh_dyn_space), sizeof(rkh_qs_space), ...) == RKH_OK
 * (RKH_ABI_CHECK() below spells the call out). */
#define RKH_ABI_VERSION 3u
uint32_t rkh_abi_version(void);
rkh_status rkh_abi_check(uint32_t abi_version, size_t sizeof_dyn_space, size_t sizeof_qs_space, size_t sizeof_rrt_params,
                         size_t sizeof_prm_params, size_t sizeof_planner_stats, size_t sizeof_rrtstar_stats,
                         size_t sizeof_prm_stats, size_t sizeof_birrt_stats, size_t sizeof_shape, size_t sizeof_kte_op);
#define RKH_ABI_CHECK()                                                                                                  \
  rkh_abi_check(RKH_ABI_VERSION, sizeof(rkh_dyn_space), sizeof(rkh_qs_space), sizeof(rkh_rrt_params), sizeof(rkh_prm_params), \
                sizeof(rkh_planner_stats), sizeof(rkh_rrtstar_stats), sizeof(rkh_prm_stats), sizeof(rkh_birrt_stats),   \
                sizeof(rkh_shape), sizeof(rkh_kte_op))

/* ---- context ------------------------------------------------------------------------------ */
rkh_status rkh_ctx_create(int device, rkh_ctx** out);
rkh_status rkh_ctx_destroy(rkh_ctx* ctx);
rkh_status rkh_ctx_synchronize(rkh_ctx* ctx);
/* hipStream_t of the context as void* (bench.py brackets launches with HIP events on it). */
void* rkh_ctx_stream(rkh_ctx* ctx);

/* ---- nearest neighbours -------------------------------------------------------------------
 * Replaces linear_neighbor_search<Graph> (ctrl/path_planning/topological_search.hpp:529-690):
 *   rkh_nn_query1  = min_dist_linear_search 1-NN            (topological_search.hpp:95-118)
 *   rkh_nn_queryk  = min_dist_linear_search k-NN + radius   (topological_search.hpp:244-274)
 * and any_knn_synchro::added_vertex (ctrl/path_planning/any_knn_synchro.hpp:69-76) = rkh_nn_append.
 * Distance is euclidean_distance_metric (ctrl/topologies/vect_distance_metrics.hpp:113-150),
 * evaluated in the reference's fp64 operation order; results are bit-identical to the CPU search,
 * including "first minimum wins" on ties.  Vertex ids are insertion indices. */
rkh_status rkh_nn_create(rkh_ctx* ctx, int dims, uint64_t capacity, rkh_nn** out);
rkh_status rkh_nn_destroy(rkh_nn* nn);
rkh_status rkh_nn_clear(rkh_nn* nn);
uint64_t rkh_nn_size(const rkh_nn* nn);
/* any_knn_synchro::removed_vertex (ctrl/path_planning/any_knn_synchro.hpp:77-84; called from
 * planning_visitor_base::vertex_to_be_removed, planning_visitors.hpp:213-216): vertex `index` is never returned by a
 * query again.  The store stays append-only -- indices of the other vertices do not change and rkh_nn_size still counts
 * the row -- and the row itself is overwritten with +inf coordinates: every sweep then computes an infinite distance for
 * it, which no comparison of the reference accepts (d < radius, d < best), at no extra bytes per sweep (a separate n/8
 * mask would add a read per row).  Removing a vertex twice is allowed.  rkh_nn_live_size counts the vertices left. */
rkh_status rkh_nn_remove(rkh_nn* nn, uint64_t index);
uint64_t rkh_nn_live_size(const rkh_nn* nn);
/* pts: n points, row-major [n][dims] */
rkh_status rkh_nn_append(rkh_nn* nn, const double* pts, uint64_t n);
/* q: [B][dims]; idx[B] (0xFFFFFFFF if the set is empty), dist[B] */
rkh_status rkh_nn_query1(rkh_nn* nn, const double* q, uint32_t B, uint32_t* idx, double* dist);
/* k-NN with radius: idx/dist are [B][k], nearest first, padded with 0xFFFFFFFF / +inf; count[B].
 * Among exactly equal distances the reference's order is std::heap-defined; this returns
 * ascending (distance, index). */
rkh_status rkh_nn_queryk(rkh_nn* nn, const double* q, uint32_t B, uint32_t k, double radius, uint32_t* idx,
                         double* dist, uint32_t* count);
/* Device-pointer variants for callers that keep queries / results in HBM (bench, planner).
 * d_q: [B][dims] row-major in HBM; d_idx[B]; d_dist[B]. Asynchronous on the context stream. */
rkh_status rkh_nn_query1_async(rkh_nn* nn, const double* d_q, uint32_t B, uint32_t* d_idx, double* d_dist);
rkh_status rkh_nn_queryk_async(rkh_nn* nn, const double* d_q, uint32_t B, uint32_t k, double radius,
                               uint32_t* d_idx, double* d_dist, uint32_t* d_count);
/* Fill the set with n uniform points of the unit hypercube directly on the device (synthetic
 * trees for the sweep microbenchmark); deterministic in seed. */
/* Promise that every |coordinate| of the stored vertices and of all later queries is <= bound (for a hyperbox_topology:
 * the largest |corner coordinate|, ctrl/topologies/hyperbox_topology.hpp:97-103).  Sweeps of 5 or more queries then run a
 * single-precision pre-filter (the fp32 matrix instructions for 5..32 queries over at least 8192 vertices and for more
 * than 32 queries, as a split-bf16 estimate from 7 dimensions on) in front of the exact fp64 test; results stay
 * bit-identical.  bound = 0 (default) switches the pre-filters off, and so does a bound outside [1e-6, 1e6].
 * The promise is checked where the library holds the data on the host: rkh_nn_append and rkh_nn_query1 return
 * RKH_ERR_BAD_ARG for a coordinate beyond the bound, and so does this call if rows already stored exceed it.  Queries
 * handed over in HBM (rkh_nn_query1_async) are the caller's responsibility: a query outside the bound may return a
 * wrong neighbour. */
rkh_status rkh_nn_set_coord_bound(rkh_nn* nn, double bound);
rkh_status rkh_nn_fill_uniform(rkh_nn* nn, uint64_t n, uint64_t seed);

/* ---- scene: KTE chain + proximity environment ----------------------------------------------
 * rkh_scene_create flattens what the reference holds as kte_map_chain + mass_matrix_calc
 * (ctrl/mbd_kte/kte_map_chain.hpp, mass_matrix_calculator.cpp) and one proxy_query_pair_3D
 * (robot model = shapes with anchor >= 0, environment model = anchor -1;
 * geometry/proximity/proxy_query_model.cpp:215-402).  The HIP kernels cover serial chains of
 * {driving_actuator_gen, inertia_gen, revolute_joint_3D, rigid_link_3D, inertia_3D} groups
 * (the pattern of examples/robot_airship/old/CRS_A465_models.cpp:751-785); anything else returns
 * RKH_ERR_UNSUPPORTED. */
rkh_status rkh_scene_create(rkh_ctx* ctx, const rkh_kte_op* prog, int n_ops, const rkh_chain_base* base,
                            const rkh_shape* shapes, int n_shapes, rkh_scene** out);
/* The same with convex vertex sets among the shapes (RKH_SHAPE_MESH): mesh_vertices = the pool [n_mesh_vertices][3] the
 * mesh shapes index into.  Pairs with a mesh are evaluated by GJK over support maps (sphere = point + radius, capped
 * cylinder = segment + radius, box, vertex set); the reference's closed forms stay in place for all other pairs.
 * BASELINE config C4 ("200 mesh obstacles, batched GJK").  Meshes pair with spheres, capped cylinders, boxes and
 * meshes; they have no finder against planes and cylinders. */
rkh_status rkh_scene_create_with_meshes(rkh_ctx* ctx, const rkh_kte_op* prog, int n_ops, const rkh_chain_base* base,
                                        const rkh_shape* shapes, int n_shapes, const double* mesh_vertices,
                                        uint32_t n_mesh_vertices, rkh_scene** out);
rkh_status rkh_scene_destroy(rkh_scene* scene);
int rkh_scene_num_dof(const rkh_scene* scene);
int rkh_scene_num_pairs(const rkh_scene* scene);

/* kte_nl_system::get_state_derivative (ctrl/ctrl_sys/kte_nl_system.hpp:239-290) for B (x,u) pairs:
 * x [B][2n] interleaved (q,qd), u [B][n]; pd [B][2n]; optional M [B][n][n], f [B][n].
 * Returns RKH_ERR_SINGULAR if any mass matrix pivot < 1e-8 (singularity_error). */
rkh_status rkh_state_derivative(rkh_scene* scene, const double* x, const double* u, uint32_t B, double* pd,
                                double* M, double* f);
/* manip_dk_proxy_env_impl::is_free's distance (ctrl/topologies/manip_free_workspace.hpp:79-99):
 * minimum proxy-pair distance for B states (apply_to_model + findMinimumDistance). */
rkh_status rkh_min_distance(rkh_scene* scene, const double* x, uint32_t B, double* dist);
/* steer_position_toward of the steerable dynamic free space (rkh_dyn_space) for B (a,b) pairs:
 * x_out [B][2n] last collision-free state, steps_free[B] accepted RK4 steps, record (optional)
 * [B][steps_per_edge+1][2n] the steer record. RK4 = runge_kutta4_integrate_impl
 * (ctrl/sys_integrators/runge_kutta4_integrator_sys.hpp:53-97). */
rkh_status rkh_propagate(rkh_scene* scene, const rkh_dyn_space* space, const double* a, const double* b,
                         uint32_t B, double fraction, double* x_out, uint32_t* steps_free, double* record);
/* Quasi-static edge walk interp_topo_move_position_toward_pred
 * (ctrl/interpolation/interpolated_topologies.hpp:137-163) over joint positions, for B (a,b) pairs:
 * lower/upper [n] joint box, out [B][n] last free point, n_checked[B] is_free calls made. */
rkh_status rkh_edge_check(rkh_scene* scene, const double* lower, const double* upper, double min_interval,
                          const double* a, const double* b, uint32_t B, double fraction, double* out,
                          uint32_t* n_checked);

/* ---- planner: rrt_planner::solve_planning_query (LINEAR_SEARCH_KNN, UNIDIRECTIONAL) ----------
 * (ctrl/path_planning/rrt_path_planner.tpp:66-145 -> generate_rrt, ctrl/graph_alg/rr_tree.hpp:179-199)
 * over the steerable dynamic space.  Expansion is speculative in batches but commits vertices in
 * exactly the order of the sequential algorithm, so vertex ids, parents and sample consumption
 * equal the CPU planner's on the same seed. */
typedef struct rkh_planner_stats {
  uint64_t num_vertices;   /* num_vertices(g), root included */
  uint64_t iterations;     /* samples consumed = generate_rrt loop iterations */
  uint64_t edges_checked;  /* steer_towards_position + goal probes committed */
  uint64_t edges_speculated; /* propagate kernel edges launched (incl. discarded speculation) */
  uint64_t rounds;         /* speculative batches */
  uint64_t num_solutions;
  double best_cost;
  uint32_t done;
} rkh_planner_stats;

rkh_status rkh_planner_create(rkh_scene* scene, const rkh_dyn_space* space, const rkh_rrt_params* prm,
                              rkh_planner** out);
/* A batch of n_problems independent planning problems (seeds / start-goal queries) on one scene.  They advance in
 * lock-step rounds and share every kernel launch, which is what fills the chip (the reference's evaluation mode is
 * Monte-Carlo over independent runs, ctrl/path_planning/planner_exec_engines.hpp:139-206).  Each problem is
 * still, bit for bit, the sequential planner on its own seed.  Every stats pointer below is an array of
 * rkh_planner_num_problems() entries. */
rkh_status rkh_planner_create_batch(rkh_scene* scene, const rkh_dyn_space* space, const rkh_rrt_params* prms,
                                    uint32_t n_problems, rkh_planner** out);
uint32_t rkh_planner_num_problems(const rkh_planner* p);
/* The same planner over the quasi-static free space manip_quasi_static_env (points = joint positions, steering =
 * move_position_toward with the min_interval collision walk, goal probe = interp_topo_get_distance_pred). */
rkh_status rkh_planner_create_qs_batch(rkh_scene* scene, const rkh_qs_space* space, const rkh_rrt_params* prms,
                                       uint32_t n_problems, rkh_planner** out);
rkh_status rkh_planner_destroy(rkh_planner* p);
/* Enqueue `rounds` speculative batches on the planner's stream (no host sync). */
rkh_status rkh_planner_enqueue(rkh_planner* p, uint32_t rounds);
/* Wait for the enqueued batches and refresh stats; *done is set once keep_going() turned false. */
rkh_status rkh_planner_sync(rkh_planner* p, rkh_planner_stats* stats);
/* Run to completion (keep_going() false): enqueue + sync until done. */
rkh_status rkh_planner_solve(rkh_planner* p, rkh_planner_stats* stats);
/* Copy the motion graph of one problem out: pos [num_vertices][2n], parent[num_vertices] (root 0xFFFFFFFF),
 * nn_seq[iterations], accept[iterations], goal_dist[num_vertices-1].  Any pointer may be NULL. */
rkh_status rkh_planner_get_tree(rkh_planner* p, uint32_t problem, double* pos, uint32_t* parent, uint32_t* nn_seq,
                                uint8_t* accept, double* goal_dist);
void* rkh_planner_stream(rkh_planner* p);

/* ---- RRT*: rrtstar_planner::solve_planning_query (LINEAR_SEARCH_KNN, UNIDIRECTIONAL, undirected motion graph) ----
 * (ctrl/path_planning/rrtstar_path_planner.tpp:298- -> generate_rrt_star, ctrl/graph_alg/rrt_star.hpp:169-190,530-570;
 * rrg_node_generator node_generators.hpp:137-172; star_neighborhood neighborhood_functors.hpp:95-102;
 * lazy_node_connector lazy_connector.hpp:79-123,230-275,332-372) over the quasi-static free space.
 * Iterations are sequential (rewiring); the two k-NN sweeps and all candidate edges of an iteration run batched on
 * the device, for all problems of the batch at once.  Vertex 0 = start, vertex 1 = goal. */
typedef struct rkh_rrtstar_stats {
  uint64_t num_vertices, samples, loop_iterations, num_solutions, rewires, edges_checked;
  double best_cost;
  uint64_t pruned, skipped; /* branch-and-bound: vertices removed / points dropped before a vertex was created */
} rkh_rrtstar_stats;
rkh_status rkh_rrtstar_create_qs_batch(rkh_scene* scene, const rkh_qs_space* space, const rkh_rrt_params* prms,
                                       uint32_t n_problems, rkh_rrtstar** out);
/* The same planner over the steerable dynamic free space (rkh_dyn_space: vertices are states (q, qd), every candidate
 * edge -- expand_to_nearest, can_be_connected in each direction -- is an RK4 propagation with collision checks, as in
 * rkh_planner_*).  For a space with an asymmetric metric the reference builds a directed motion graph
 * (motion_graph_structures.hpp:73-74) and runs rrg_node_generator's directed overload (node_generators.hpp:176-206) and
 * lazy_node_connector's (lazy_connector.hpp:418-460: connect_best_predecessor over the predecessor neighbourhood,
 * connect_successors :277-325 over the successor neighbourhood; both from min_dist_linear_search
 * topological_search.hpp:296-345).  This space's metric is the symmetric Euclidean state distance, under which the two
 * neighbourhoods coincide (one k-NN sweep serves both); what is directional is every can_be_connected call, and those
 * are evaluated per direction. */
rkh_status rkh_rrtstar_create_batch(rkh_scene* scene, const rkh_dyn_space* space, const rkh_rrt_params* prms,
                                    uint32_t n_problems, rkh_rrtstar** out);
/* USE_BRANCH_AND_BOUND_PRUNING_FLAG (rrtstar_path_planner.tpp:270-283; unidirectional planners, before the first solve):
 * generate_bnb_rrt_star (rrt_star.hpp:690-730) with branch_and_bound_connector (branch_and_bound_connector.hpp:105-330).
 * Once the goal has a predecessor, a new point whose straight-line bound |start p| + |p goal| exceeds the goal's cost is
 * dropped, and after every connection the vertices whose distance_accum + |v goal| exceeds it are removed from the graph
 * (vertex_to_be_removed -> the KNN synchro's removed_vertex: the vertex keeps its index, its row on the device becomes
 * +inf).  Kept as in the reference: with nothing but uniform sampling almost every later point is dropped, so a run is
 * normally ended by max_loop_iterations; children of a removed vertex keep their cost and stay candidate parents.
 * Defined here: ids are append-only (the reference's pooled container would re-use the hole), the pruning loop stops on
 * an empty queue. */
rkh_status rkh_rrtstar_set_branch_and_bound(rkh_rrtstar* p, int enabled);
rkh_status rkh_rrtstar_get_removed(rkh_rrtstar* p, uint32_t problem, uint8_t* removed);
rkh_status rkh_rrtstar_destroy(rkh_rrtstar* p);
/* stats: array of n_problems entries.  max_loop_iterations < 0: run until keep_going() is false. */
rkh_status rkh_rrtstar_solve(rkh_rrtstar* p, int64_t max_loop_iterations, rkh_rrtstar_stats* stats);
/* pos [num_vertices][n_dof], pred[num_vertices] (0xFFFFFFFF = unconnected), dist[num_vertices] (distance_accum),
 * near_seq[loop_iterations] (x_near returned by the node generator).  Any pointer may be NULL. */
rkh_status rkh_rrtstar_get_graph(rkh_rrtstar* p, uint32_t problem, double* pos, uint32_t* pred, double* dist,
                                 uint32_t* near_seq);
/* ---- bidirectional RRT*: generate_rrt_star_bidir (ctrl/graph_alg/rrt_star.hpp:197-236,612-659) with rrg_bidir_generator
 * (node_generators.hpp:215-277) and the bidirectional lazy_node_connector (lazy_connector.hpp:465-518), over the
 * quasi-static free space (the reference requires a reversible space, rrtstar_path_planner.tpp:70).  Every vertex
 * carries a predecessor / distance_accum towards the start (forward tree, root = vertex 0) and a successor /
 * fwd_distance_accum towards the goal (backward tree, root = vertex 1).  As in the reference, no solution is ever
 * registered: the goal has a successor (itself), so connect_successors never gives it a predecessor, which is what
 * vertex_added tests (planning_visitors.hpp:108-116); the graph grows to max_vertices.  Vertices that end up with both
 * links are counted (`joins`, best_join_cost = distance_accum + fwd_distance_accum).  A loop iteration can add two
 * vertices, so the graph may hold max_vertices + 3.  The handle type is rkh_rrtstar; rkh_rrtstar_destroy frees it. */
typedef struct rkh_birrtstar_stats {
  uint64_t num_vertices, samples, loop_iterations, rewires, fwd_rewires, joins, edges_checked;
  double best_join_cost;
} rkh_birrtstar_stats;
rkh_status rkh_birrtstar_create_qs_batch(rkh_scene* scene, const rkh_qs_space* space, const rkh_rrt_params* prms,
                                         uint32_t n_problems, rkh_rrtstar** out);
rkh_status rkh_birrtstar_solve(rkh_rrtstar* p, int64_t max_loop_iterations, rkh_birrtstar_stats* stats);
/* pos [num_vertices][n_dof]; pred / succ (0xFFFFFFFF = none); dist / fwd_dist; near_pred / near_succ [loop_iterations]:
 * the generator's x_pred / x_succ.  Any pointer may be NULL. */
rkh_status rkh_birrtstar_get_graph(rkh_rrtstar* p, uint32_t problem, double* pos, uint32_t* pred, double* dist, uint32_t* succ,
                                   double* fwd_dist, uint32_t* near_pred, uint32_t* near_succ);
/* ---- PRM: prm_planner::solve_planning_query (LINEAR_SEARCH_KNN, ADJ_LIST_MOTION_GRAPH, undirected graph) ----
 * (ctrl/path_planning/prm_path_planner.tpp:131-365 -> generate_prm, ctrl/graph_alg/probabilistic_roadmap.hpp:211-249,
 * 309-404; prm_node_connector prm_connector.hpp:68-182; prm_conn_visitor probabilistic_roadmap.hpp:75-196;
 * prm_density_calculator density_calculators.hpp:45-73; random_walk planning_visitors.hpp:403-432) over the
 * quasi-static free space.  One loop iteration (construct: rejection sampling + k-NN + can_be_connected to every
 * neighbour; expand: the 11 random-walk attempts from the lowest-density vertex + k-NN + connections) is one batched
 * device step for all problems.  Vertex 0 = start, vertex 1 = goal.  As in the reference (which instantiates
 * density_plan_visitor here), no solution is registered: the roadmap grows to max_vertices; merged_at_vertex is the
 * vertex count at which start and goal first shared a connected component (-1: never). */
typedef struct rkh_prm_stats {
  uint64_t num_vertices, num_edges, samples, rejected, loop_iterations, num_components, publish_calls;
  int64_t merged_at_vertex;
  uint64_t edges_checked, device_steps;
} rkh_prm_stats;
rkh_status rkh_prm_create_qs_batch(rkh_scene* scene, const rkh_qs_space* space, const rkh_prm_params* prms,
                                   uint32_t n_problems, rkh_prm** out);
/* The same planner over the steerable dynamic free space: vertices are states (q, qd); the rejection sampling tests
 * is_free(state), every random-walk attempt is steer_position_toward(v, target_dist / dist, p_rnd) -- an RK4 propagation
 * over that fraction of the edge time -- and every connection a full propagation judged by can_be_connected.  A walk
 * whose fraction asks for more RK4 steps than an edge's budget (64) is refused with RKH_ERR_UNSUPPORTED. */
rkh_status rkh_prm_create_batch(rkh_scene* scene, const rkh_dyn_space* space, const rkh_prm_params* prms, uint32_t n_problems,
                                rkh_prm** out);
rkh_status rkh_prm_destroy(rkh_prm* p);
/* stats: array of n_problems entries.  max_loop_iterations < 0: run until keep_going() is false. */
rkh_status rkh_prm_solve(rkh_prm* p, int64_t max_loop_iterations, rkh_prm_stats* stats);
/* pos [num_vertices][n_dof]; edge_u/edge_v/edge_w [num_edges] in insertion order; density, cc_root [num_vertices];
 * kind [loop_iterations] (0 construct, 1 expand, 2 expand failed -> Q.pop()); expanded [loop_iterations] (Q.top() of
 * expansion iterations, else 0xFFFFFFFF).  Any pointer may be NULL. */
rkh_status rkh_prm_get_graph(rkh_prm* p, uint32_t problem, double* pos, uint32_t* edge_u, uint32_t* edge_v,
                             double* edge_w, double* density, uint32_t* cc_root, uint8_t* kind, uint32_t* expanded);
/* ---- Bidirectional RRT: rrt_planner with BIDIRECTIONAL_PLANNING (the reference's default flag,
 * ctrl/path_planning/rrt_path_planner.hpp:114) -> generate_bidirectional_rrt (ctrl/graph_alg/rr_tree.hpp:256-317),
 * joining_vertex_found (planning_visitors.hpp:223-231), two-graph solution registration
 * (solution_path_factories.hpp:359-408), over the quasi-static free space.  Tree 1 is rooted at the start, tree 2 at
 * the goal; one expansion (nearest neighbour + edge walk) is one batched device step for all problems. */
typedef struct rkh_birrt_stats {
  uint64_t num_vertices_1, num_vertices_2, loop_iterations, samples, num_solutions, joins, edges_checked;
  double best_cost;
} rkh_birrt_stats;
rkh_status rkh_birrt_create_qs_batch(rkh_scene* scene, const rkh_qs_space* space, const rkh_rrt_params* prms,
                                     uint32_t n_problems, rkh_birrt** out);
rkh_status rkh_birrt_destroy(rkh_birrt* p);
/* stats: array of n_problems entries.  max_loop_iterations < 0: run until keep_going() is false. */
rkh_status rkh_birrt_solve(rkh_birrt* p, int64_t max_loop_iterations, rkh_birrt_stats* stats);
/* pos1 [num_vertices_1][n_dof], parent1 (root: 0xFFFFFFFF), the same for tree 2; nn_seq / accept [2 * loop_iterations]:
 * nearest vertex and reached_new of every expansion (tree 1, tree 2, tree 1, ...).  Any pointer may be NULL. */
rkh_status rkh_birrt_get_trees(rkh_birrt* p, uint32_t problem, double* pos1, uint32_t* parent1, double* pos2,
                               uint32_t* parent2, uint32_t* nn_seq, uint8_t* accept);
/* ---- Solution paths (what register_*_solution_path_impl of ctrl/path_planning/solution_path_factories.hpp walks) ----
 * Vertex indices into the arrays of rkh_planner_get_tree / rkh_rrtstar_get_graph / rkh_birrt_get_trees, start first;
 * n_path = 0 if no solution is registered; the path pointers may be NULL to query the lengths; cost = the registered
 * solution cost (RRT: path + goal-probe distance of the last vertex; RRT*: distance_accum of the goal; bidirectional:
 * both tree paths + joining distance). */
rkh_status rkh_planner_get_solution(rkh_planner* p, uint32_t problem, uint32_t* path, uint32_t capacity,
                                    uint32_t* n_path, double* cost);
rkh_status rkh_rrtstar_get_solution(rkh_rrtstar* p, uint32_t problem, uint32_t* path, uint32_t capacity, uint32_t* n_path,
                                    double* cost);
rkh_status rkh_birrt_get_solution(rkh_birrt* p, uint32_t problem, uint32_t* path1, uint32_t* n_path1, uint32_t* path2,
                                  uint32_t* n_path2, uint32_t capacity, double* cost);

#ifdef __cplusplus
}
#endif
#endif
