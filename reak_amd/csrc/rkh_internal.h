// rkh_internal.h -- shared declarations of librkh.so (host side + device structs).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/rkh.h"
#include "../../include/rkh_diag.h"

namespace rkh {

void set_error(const std::string& msg);

#define RKH_HIP(expr)                                                                          \
  do {                                                                                         \
    hipError_t _e = (expr);                                                                    \
    if (_e != hipSuccess) {                                                                    \
      ::rkh::set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                     \
      return (_e == hipErrorOutOfMemory) ? RKH_ERR_OOM : RKH_ERR_DEVICE;                       \
    }                                                                                          \
  } while (0)

constexpr int kMaxDof = 12;         // joints of a scene (dynamics kernels: 1, 2, 3, 6; quasi-static kernels also 12)
constexpr int kMaxEnvShapes = 256;  // environment shapes resident in LDS
constexpr int kMaxSteps = 64;       // RK4 steps per edge

// ---- NN sweep (nn_sweep.hip) -----------------------------------------------------------------
// Vertex positions live row-major [n][D] in HBM (one contiguous 8*D-byte row per vertex): the sweep
// streams them linearly through LDS tiles and the propagate kernel gathers a parent with one row read.
struct NnStore {
  double* d_pos = nullptr;  // [capacity][D]
  uint64_t capacity = 0;
  int D = 0;
};

struct NnArgs {  // one 1-NN problem: tree rows, queries, outputs (device pointers)
  const double* pos = nullptr;        // [n][DP] vertex rows
  uint64_t n = 0;                      // vertex count (used if d_n == nullptr)
  const uint32_t* d_n = nullptr;       // vertex count read on the device (planner rounds enqueued without host sync)
  const double* q = nullptr;           // [B][D] queries
  const uint32_t* d_qoff = nullptr;    // optional row offset of the query block, read on the device
  uint32_t B = 0;
  const uint32_t* d_B = nullptr;       // query count read on the device
  double* part_dist = nullptr;         // [blocks][Bpad] per-block partial minima
  uint32_t* part_idx = nullptr;
  uint32_t* idx = nullptr;             // [B] results
  double* dist = nullptr;
  uint32_t* seed = nullptr;            // [B] optional: matrix-core sweeps start from a sampled minimum (all 0xFF between sweeps)
  // planner-regime sweep over the half-precision mirror of the tree (nn_mirror.hip)
  const void* mirror = nullptr;          // [ceil(capacity / 32)] slabs of 1 KB, nn_mirror.h
  const uint32_t* dx_max_bits = nullptr; // max over the rows of |x - x_h| (bits of a float)
  uint4* qfrag = nullptr;                // [B][2] B operands of the round's queries
  double* qinfo = nullptr;               // [B][3] |q_h|^2, |q - q_h|, |q|
  float* thr = nullptr;                  // [B] smallest estimate + band
  uint32_t* cand_cnt = nullptr;          // [B] rows at or below thr (zero between sweeps)
  uint32_t* cand_rows = nullptr;         // [B][nn1_mirror_cand_cap()]
  float* est = nullptr;                  // [nn1_mirror_max_slices()][est_stride] smallest estimate per (row slice, query)
  uint32_t est_stride = 0;
};
// nn_mirror.hip
uint32_t nn1_mirror_queries();
uint32_t nn1_mirror_cand_cap();
uint32_t nn1_mirror_max_slices();
size_t nn1_mirror_bytes(uint64_t capacity_rows);
size_t nn1_mirror_query_bytes();
void nn1_mirror_carve(void* base, uint32_t b_max, NnArgs* a);
bool nn1_mirror_applies(int D, double coord_bound);
rkh_status launch_mirror_fill(hipStream_t s, void* d_mirror, uint64_t capacity_rows);
rkh_status launch_mirror_build(hipStream_t s, void* d_mirror, const double* d_pos, uint64_t n, int D, int DP,
                               uint32_t* d_dx_max_bits);
rkh_status launch_nn1_mirror(hipStream_t s, int D, const NnArgs* d_table, uint32_t n_problems, uint64_t n_upper,
                             uint32_t B_upper, double x_norm_bound, const uint32_t* d_yblock_base, hipEvent_t ev0,
                             hipEvent_t ev1);
int nn_padded_dims(int D);
rkh_status launch_nn1(hipStream_t s, int D, const NnArgs& single, const NnArgs* d_table, uint32_t n_problems,
                      uint64_t n_upper, uint32_t B, uint32_t part_capacity_blocks, hipEvent_t ev0 = nullptr,
                      hipEvent_t ev1 = nullptr, double coord_bound = 0.0, const uint32_t* d_yblock_base = nullptr,
                      bool table_has_seed = false);
// d_yblock_base (table launches, matrix-core kernel): [n_problems + 1] exclusive prefix of ceil(B_p / nn1_mfma_queries())
// over the problems; the grid's blocks then take the working (row slice, query block) pairs in dispatch order.
uint32_t nn1_mfma_queries();
// coord_bound > 0: every |coordinate| of rows and queries is <= coord_bound; sweeps with >= 32 queries then run the
// single-precision pre-filter variant (identical results).
uint32_t nn1_partial_blocks(uint64_t n_upper, uint32_t B, uint32_t n_problems = 1);
// k-NN with radius (knn_sweep.hip).  ws: device workspace from knn_workspace_bytes(); *d_overflow is set (non-zero)
// if a query met more than the candidate capacity (pathological ties).
struct KnnWorkspace {
  double* sub = nullptr;       // [M][Bpad] per-subrange minima (phase A)
  double* tau = nullptr;       // [B] inclusive bound on the k-th smallest distance
  uint32_t* cnt = nullptr;     // [B] candidates collected
  double* cand_d = nullptr;    // [B][cmax]
  uint32_t* cand_i = nullptr;  // [B][cmax]
  uint32_t* overflow = nullptr;
  uint32_t cmax = 0, m_sub = 0, gx = 0;
  uint32_t ksel = 8;           // subset minima a block contributes per query (1..8)
};
rkh_status knn_plan(uint64_t n, uint32_t B, uint32_t k, KnnWorkspace* ws, size_t* bytes);
void knn_carve(void* base, uint32_t B, KnnWorkspace* ws);
rkh_status launch_nnk(hipStream_t s, const NnStore& st, uint64_t n, const double* d_q, uint32_t B, uint32_t k,
                      double radius, uint32_t* d_idx, double* d_dist, uint32_t* d_count, const KnnWorkspace& ws);
// One k-NN job (one tree, B queries).  A launch serves either one job (by value) or a device table of jobs
// (blockIdx.z = job; RRT* / PRM batches: one query per problem, all problems in the same four launches).
struct KnnArgs {
  const double* pos = nullptr;  // [n][DP] vertex rows
  uint64_t n = 0;
  const double* q = nullptr;    // [B][D]
  int D = 0;
  uint32_t B = 0, k = 0;
  double radius = 0.0;
  KnnWorkspace ws;
  uint32_t m_pow2 = 0;          // next_pow2(ws.m_sub)
  uint32_t* out_idx = nullptr;  // [B][k]
  double* out_dist = nullptr;
  uint32_t* out_cnt = nullptr;  // [B]
};
uint32_t next_pow2(uint32_t v);
// Launch a table of jobs (d_table on the device, h_table its host copy for grid sizing).  All jobs share D.
rkh_status launch_nnk_table(hipStream_t s, int D, const KnnArgs* d_table, const KnnArgs* h_table, uint32_t n_jobs);
rkh_status launch_fill_uniform(hipStream_t s, const NnStore& st, uint64_t n, uint64_t seed);

// ---- scene (propagate.hip) ---------------------------------------------------------------------
struct JointDev {  // one {actuator, rotor inertia, revolute joint, rigid link, link inertia} group
  double axis[3];     // revolute_joint_3D::mAxis as given
  double axis_n[3];   // axis_angle's normalised copy (rotations_3D.hpp:1961-1974)
  double joint_inertia;
  double off_pos[3];
  double off_quat[4];
  double off_R[9];   // rotmat(off_quat), row-major (same formula as the device would use)
  double mass;
  double inertia[6]; // a11,a12,a13,a22,a23,a33
};
struct ShapeDev {
  int32_t kind;
  int32_t link;      // robot shapes: joint index whose end frame anchors the shape; env: -1
  double pos[3];
  double quat[4];
  double dims[3];
  double brad;       // getBoundingRadius()
};
struct SceneDev {
  int32_t n_dof;
  int32_t n_robot;   // robot shapes (anchored)
  int32_t n_env;     // environment shapes
  int32_t beam_on;   // 1: the chain carries a flexible_beam_3D (see beam_j1 / beam_j2)
  double base_pos[3];
  double base_quat[4];
  double base_acc[3];
  JointDev joints[kMaxDof];
  ShapeDev robot[kMaxDof * 2];
  ShapeDev env[kMaxEnvShapes];
  // cull table of the environment: global centre of the bounding sphere (pose.transformToGlobal(0) = the pose's
  // position) and its radius; one bit per shape and kind (sphere, box, capped cylinder) in chunks of 64 shapes
  double env_cull[kMaxEnvShapes][4];
  unsigned long long env_kind_mask[3][kMaxEnvShapes / 64];
  // Branching (quasi-static kernels only): joint j with branch_start[j] != 0 does not continue the previous link but
  // starts from the chain base through a fixed mount (rigid_link_3D from frame 0; identity if the joint sits on the base)
  int32_t branch_start[kMaxDof];
  int32_t n_branches;  // joints with branch_start (0 for a plain serial chain)
  int32_t beam_j1;     // the beam's anchor 1 is the link end frame of this joint
  int32_t beam_j2;     // anchor 2: link end frame of this joint, or -1 = the world anchor (beam_pos, beam_quat)
  int32_t planar_dynamics;  // planar chain given with its actuators and inertias: the dynamics entry points accept it
  int32_t planar;      // 1: planar chain (revolute_joint_2D / rigid_link_2D, 2D shapes): poses carry (x, y) and (cos, sin)
  int32_t branch_first[kMaxDof];  // first joint of the branch joint j belongs to
  double mount_pos[kMaxDof][3];
  double mount_quat[kMaxDof][4];
  // flexible_beam_3D (flexible_beam.cpp:155-193): rest length, stiffness, torsion stiffness, world anchor pose
  double beam_rest, beam_k, beam_kt;
  double beam_pos[3], beam_quat[4];
  // Static reach of the robot shapes (serial chains).  The environment shapes are stored in ascending order of
  // "closeness" = |centre - chain base| - bounding radius, and robot shape r can only ever touch the first
  // robot_n_reach[r] of them: its centre stays within (sum of the link offsets below its joint) + |local position| of the
  // chain base, plus its bounding radius.  Pairs beyond that have a positive bounding-sphere gap in every configuration,
  // i.e. they are the pairs the cull of proxy_query_pair_3D::findMinimumDistance (proxy_query_model.cpp:384-389) drops.
  int32_t robot_n_reach[kMaxDof * 2];
  // bit o of env_finder_mask[k][chunk]: the reference has a finder for (robot shape of kind k, environment shape o)
  // (createProxFinderList, proxy_query_model.cpp:215-374: no finder for box-box, cylinder-cylinder, cylinder-box and
  // capped cylinder-cylinder)
  unsigned long long env_finder_mask[9][kMaxEnvShapes / 64];
  int32_t has_ext_shapes;  // 1: the scene holds a plane or a cylinder (not handled by the first-generation lane kernel)
  int32_t has_meshes;      // 1: convex vertex sets among the shapes (GJK pairs; wave-per-edge and quasi-static kernels)
  const double* mesh_verts;  // device pointer: the vertex pool [n][3] the mesh shapes index into (dims[0], dims[1])
};

}  // namespace rkh

struct rkh_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
};

struct rkh_nn {
  rkh_ctx* ctx = nullptr;
  rkh::NnStore st;
  uint64_t n = 0;
  // scratch for host-pointer queries
  double* d_q = nullptr;
  uint32_t* d_idx = nullptr;
  double* d_dist = nullptr;
  uint32_t* d_count = nullptr;
  uint64_t q_cap = 0, res_cap = 0;
  double* d_part_dist = nullptr;
  uint32_t* d_part_idx = nullptr;
  uint64_t part_cap = 0;
  uint32_t* d_seed = nullptr;  // NnArgs::seed
  uint32_t seed_cap = 0;
  double max_abs_coord = 0.0;  // over the rows appended from the host (checked against coord_bound)
  std::vector<uint8_t> removed;  // tombstones (host copy; the device row of a removed vertex holds +inf)
  uint64_t n_removed = 0;
  uint32_t part_blocks = 0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;  // one-shot: bracket the next sweep kernel
  void* d_knn_ws = nullptr;  // k-NN workspace
  double coord_bound = 0.0;  // rkh_nn_set_coord_bound: |coordinate| bound enabling the single-precision pre-filters
  size_t knn_ws_bytes = 0;
};

struct rkh_scene {
  rkh_ctx* ctx = nullptr;
  rkh::SceneDev host;
  rkh::SceneDev* d_scene = nullptr;
  void* d_pairs = nullptr;  // PairDev[n_pairs], sorted by routine
  double* d_mesh_verts = nullptr;  // vertex pool of the mesh shapes
  int* d_err = nullptr;
  int n_pairs = 0;
  int n_pairs_verdict = -1;  // the first entries of d_pairs: pairs whose shapes can touch at all (verdict kernels scan these)
};

namespace rkh {

struct PairDev {
  uint8_t routine;      // PairRoutine (proximity_device.h)
  uint8_t s1_is_robot;  // 1: (shape1, shape2) = (robot, env); 0: (env, robot)
  uint16_t robot;
  uint16_t env;
  uint16_t pad;
};

struct DynDev {  // rkh_dyn_space on the device (passed by value)
  double dt, kp, kd, u_max, goal_tol;
  double lower[2 * kMaxDof], upper[2 * kMaxDof];
  double full_time;         // steps_per_edge * dt (the travel time of fraction 1)
  int n_steps;              // RK4 steps for this fraction
  int8_t inner[kMaxSteps];  // runge_kutta4_integrate_impl loop iterations of step k (normally 1), whole step budget
};

enum EdgeMode : int {
  EDGE_PLAIN = 0,
  EDGE_STEER_ACCEPT = 1,
  EDGE_GOAL_PROBE = 2,
  EDGE_CONNECT = 3,
  EDGE_WALK_ACCEPT = 4,  // random_walk: traveled > steer_tol * best_case[e] (best_case carries the target distance)
  EDGE_STEER_BOTH = 6,   // quasi-static kernel: EDGE_STEER_ACCEPT in bit 0 of accept, bit 1 = the walk ran to its end
  EDGE_POINT = 5,        // accept = is_free(target point), no walk (quasi-static kernel, one-wave-per-edge dynamics kernel)
};

struct QsDev {  // manip_quasi_static_env on the device (passed by value)
  double min_interval, fraction;
  double lower[kMaxDof], upper[kMaxDof];
  double speed[kMaxDof];  // joint = point * speed (rate-limited joint space; 1.0 otherwise)
};
inline void qs_set_speed(QsDev& qs, const double* speed_limits, int n) {
  for (int i = 0; i < kMaxDof; ++i) qs.speed[i] = (speed_limits && i < n && speed_limits[i] != 0.0) ? speed_limits[i] : 1.0;
}

struct EdgeIO {  // inputs / outputs of one propagate launch (all device pointers)
  const double* src = nullptr;         // source rows
  const uint32_t* src_idx = nullptr;   // row of edge e (null: *d_src_first + e, or e)
  const uint32_t* d_src_first = nullptr;
  uint32_t src_stride = 0;
  const double* tgt = nullptr;         // target rows
  const uint32_t* d_tgt_off = nullptr; // row offset read on the device
  const uint32_t* tgt_idx = nullptr;   // optional: target row of edge e (quasi-static edge kernel)
  uint32_t tgt_stride = 0;             // 0: one target for all edges
  const double* frac = nullptr;        // optional per-edge travel fraction (quasi-static kernel; null: QsDev::fraction)
  uint32_t B = 0;
  const uint32_t* d_B = nullptr;
  double* x_out = nullptr;
  uint32_t* steps_free = nullptr;
  double* record = nullptr;
  int record_stride = 0;
  int mode = EDGE_PLAIN;
  const double* best_case = nullptr;
  double steer_tol = 0.1;
  uint8_t* accept = nullptr;
  double* goal_dist = nullptr;         // indexed by source row - 1
  int* err_flag = nullptr;
  // second and later phases of a split launch (two-lanes kernel): slot s of the launch is edge edge_ids[s], which
  // resumes from row edge_ids[s] of `resume` (the x_out of the phase before) with KernelGate::step0 steps behind it
  const uint32_t* edge_ids = nullptr;
  const double* resume = nullptr;
};

// A steer kernel with a gate runs only if lo <= *count < hi (read on the device); count == nullptr: always.
struct KernelGate {
  const uint32_t* count = nullptr;
  uint32_t lo = 0, hi = 0xFFFFFFFFu;
  // two-lanes kernel, table launches: exclusive prefix of the working waves per segment (segment 2p = candidates of
  // problem p, 2p+1 = its goal probes; wave_base[n_segments] = total).  The blocks of the grid, in dispatch order, then
  // take the working waves one after the other, so the round-robin of blocks over the 8 XCDs spreads the work evenly
  // whatever the per-problem counts are (a (wave, problem) grid leaves holes that land unevenly on the XCDs).
  const uint32_t* wave_base = nullptr;
  uint32_t n_segments = 0;
  // the steps [step0, min(step1, n_steps)) of every edge (two-lanes kernel; the whole edge by default)
  uint32_t step0 = 0, step1 = 0xFFFFFFFFu;
  // optional diagnostics: the kernel adds the edge-steps it integrated (steps that began with a live edge, the one that
  // ended it included) -- the executed work of a launch, as opposed to n_steps per launched edge
  unsigned long long* steps_exec = nullptr;
};
rkh_status launch_propagate(hipStream_t s, int n_dof, int n_env, const SceneDev* d_scene, const void* d_pairs,
                            int n_pairs, const DynDev& dyn, const EdgeIO& io, uint32_t grid_edges,
                            const EdgeIO* io_b = nullptr, uint32_t grid_b = 0, int lanes_per_edge = 64,
                            const EdgeIO* tab_a = nullptr, const EdgeIO* tab_b = nullptr, uint32_t n_problems = 1,
                            double* d_lane_ws = nullptr, KernelGate gate = KernelGate());
// the two-lanes-per-edge kernel handles one serial chain, with at most a tip-to-world beam
inline bool scene_fits_lane_kernel(const SceneDev& S, int variant = 2) {
  if (variant == 1 && S.has_ext_shapes) return false;  // propagate_lane.hip knows spheres, boxes and capped cylinders
  if (S.has_meshes) return false;                       // GJK pairs run in the wave-per-edge / quasi-static kernels
  return S.n_branches == 0 && (!S.beam_on || (S.beam_j1 == S.n_dof - 1 && S.beam_j2 < 0));
}
// one lane per edge (propagate_lane.hip); d_ws: propagate_lanes_workspace_bytes() of device memory
size_t propagate_lanes_workspace_bytes(int n_dof, uint32_t edges_a, uint32_t edges_b, uint32_t n_problems);
rkh_status launch_propagate_lanes(hipStream_t s, int n_dof, const SceneDev* d_scene, const DynDev& dyn, const EdgeIO& io,
                                  uint32_t grid_edges, const EdgeIO* io_b, uint32_t grid_b, const EdgeIO* tab_a,
                                  const EdgeIO* tab_b, uint32_t n_problems, double* d_ws, KernelGate gate = KernelGate());
rkh_status launch_state_derivative(hipStream_t s, int n_dof, const SceneDev* d_scene, const double* d_x,
                                   const double* d_u, uint32_t B, double* d_pd, double* d_M, double* d_f, int* d_err);
rkh_status launch_min_distance(hipStream_t s, int n_dof, int n_env, const SceneDev* d_scene, const void* d_pairs,
                               int n_pairs, const double* d_x, uint32_t B, double* d_dist);
rkh_status launch_edge_check(hipStream_t s, int n_dof, int n_env, const SceneDev* d_scene, const void* d_pairs,
                             int n_pairs, const QsDev& qs, const EdgeIO& io, uint32_t grid_edges,
                             const EdgeIO* io_b = nullptr, uint32_t grid_b = 0, const EdgeIO* tab_a = nullptr,
                             const EdgeIO* tab_b = nullptr, uint32_t n_problems = 1);
rkh_status launch_feval_cycles_duo(hipStream_t s, int n_dof, int n_env, const SceneDev* d_scene, const double* d_x,
                                   const double* d_u, uint32_t B, int iters, unsigned long long* d_out, double* d_sink);
rkh_status launch_feval_cycles(hipStream_t s, int n_dof, int n_env, const SceneDev* d_scene, const void* d_pairs,
                               int n_pairs, const double* d_x, const double* d_u, uint32_t B, int iters,
                               unsigned long long* d_out, double* d_sink);
uint32_t lane_kernel_waves_per_cu(int n_dof);
uint32_t lane_kernel_edges_per_wave();
// second-generation two-lanes-per-edge kernel (propagate_pair.hip): registers + DPP instead of LDS, two waves per SIMD.
// Same edges per wave, same scenes (scene_fits_lane_kernel), same results.
size_t propagate_pairs_workspace_bytes(int n_dof, uint32_t edges_a, uint32_t edges_b, uint32_t n_problems);
// planar chains (propagate_planar.hip): one lane per edge; scenes register themselves at upload
void register_planar_scene(const SceneDev* d_scene);
void forget_planar_scene(const SceneDev* d_scene);
bool is_planar_scene(const SceneDev* d_scene);
// scenes with vertex-set shapes (PR_GJK pairs); the others may run kernels compiled without the support-map query
void register_mesh_scene(const SceneDev* d_scene);
void forget_mesh_scene(const SceneDev* d_scene);
bool is_mesh_scene(const SceneDev* d_scene);
rkh_status launch_propagate_planar(hipStream_t s, int n_dof, const SceneDev* d_scene, const void* d_pairs, int n_pairs,
                                   const DynDev& dyn, const EdgeIO& io, uint32_t grid_edges, const EdgeIO* io_b,
                                   uint32_t grid_b, const EdgeIO* tab_a, const EdgeIO* tab_b, uint32_t n_problems,
                                   KernelGate gate);
rkh_status launch_state_derivative_planar(hipStream_t s, int n_dof, const SceneDev* d_scene, const double* d_x,
                                          const double* d_u, uint32_t B, double* d_pd, double* d_M, double* d_f, int* d_err);
rkh_status launch_propagate_pairs(hipStream_t s, int n_dof, const SceneDev* d_scene, const DynDev& dyn, const EdgeIO& io,
                                  uint32_t grid_edges, const EdgeIO* io_b, uint32_t grid_b, const EdgeIO* tab_a,
                                  const EdgeIO* tab_b, uint32_t n_problems, double* d_ws, KernelGate gate = KernelGate());
// the same mapping, one launch per RK4 step over the live edges of all problems (see propagate_pair_step_kernel)
size_t propagate_pair_step_workspace_bytes(int n_dof, uint32_t blocks);
rkh_status launch_propagate_pair_steps(hipStream_t s, int n_dof, const SceneDev* d_scene, const DynDev& dyn,
                                       const EdgeIO* tab_a, const EdgeIO* tab_b, uint32_t n_problems,
                                       const uint32_t* d_edge_base, uint2* d_list0, uint2* d_list1, uint32_t* d_cnt,
                                       double* d_ws, uint32_t blocks, KernelGate gate, unsigned long long* d_steps_exec,
                                       uint32_t pool_blocks = 0, uint32_t* d_pool_cursor = nullptr);
uint32_t pair_kernel_waves_per_cu(int n_dof);
uint32_t pair_kernel_edges_per_wave();
rkh_status launch_pair_counts(hipStream_t s, int n_dof, const SceneDev* d_scene, const double* d_x, uint32_t B,
                              unsigned long long* d_out);
rkh_status launch_pair_cycles(hipStream_t s, int n_dof, const SceneDev* d_scene, const double* d_x, const double* d_u,
                              uint32_t B, int iters, unsigned long long* d_out, double* d_sink);
rkh_status launch_lane_cycles(hipStream_t s, int n_dof, const SceneDev* d_scene, const double* d_x, const double* d_u,
                              uint32_t B, int iters, unsigned long long* d_out, double* d_sink);
rkh_status build_dyn_dev(const rkh_dyn_space& sp, double fraction, DynDev* out);
}  // namespace rkh
