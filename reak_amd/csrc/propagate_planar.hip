// propagate_planar.hip -- the steer loop over PLANAR chains (revolute_joint_2D / rigid_link_2D / inertia_2D): RK4 forward
// dynamics + proximity test per step, one LANE per edge.
//
// Reference: revolute_joint_2D::{doMotion, doForce, applyReactionForce} (ctrl/mbd_kte/revolute_joint.cpp:30-115),
// rigid_link_2D::{doMotion, doForce} (rigid_link.cpp:87-110), inertia_2D::doForce (inertia.cpp:73-81),
// inertia_gen / driving_actuator_gen (inertia.cpp:47-54, driving_actuator.cpp:31-39), the 2D rows of
// mass_matrix_calc::get_TMT_TdMT (mass_matrix_calculator.cpp:100-145,262-285) with jacobian_gen_2D::get_jac_relative_to
// (core/kinetostatics/motion_jacobians.hpp:138-146), linsolve_Cholesky (core/lin_alg/mat_cholesky.hpp:546-554), the
// steer loop / RK4 / accept rules of propagate.hip, proxy_query_pair_2D::findMinimumDistance replayed in finder order
// (proxy_query_model.cpp:163-189).  Oracle twin: KteChain's 2D branches in oracle/reak_kte.hpp.
//
// Mapping: a planar chain has at most 7 joints and a state of 14 doubles: one edge fits one lane's registers, the
// f-eval is a few hundred fp64 operations with no data-dependent control flow, so 64 edges share a wave without any LDS
// or cross-lane traffic; the proximity test walks the finder list serially per lane (the order matters:
// proxy_query_model.cpp:176-180 culls against the running minimum).  Every product and sum is formed in the order of the
// restated reference (-ffp-contract=off).
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cmath>
#include <unordered_set>

#include "proximity_planar_device.h"
#include "rkh_internal.h"

namespace rkh {

RKH_DI d2 cross_sv(double S, d2 V) { return d2{-V.y * S, V.x * S}; }  // vect_alg.hpp:1171-1176
RKH_DI double cross_vv(d2 a, d2 b) { return a.x * b.y - a.y * b.x; }  // vect_alg.hpp:1142-1144

template <int N>
struct PlanarFrames {
  d2 Epos[N], Erot[N], Eacc[N];  // revolute_joint_2D end frames
  d2 Lpos[N], Lrot[N], Lacc[N];  // rigid_link_2D end frames
  double w[N], alpha[N];         // angular velocity / acceleration of joint j's end frame (the link's is the same)
  double cq[N], sq[N];           // cos / sin of the joint angles
};

// kte_map_chain::doMotion over {revolute_joint_2D, rigid_link_2D} (velocities only where a force term reads them)
template <int N>
RKH_DI void planar_motion(const SceneDev* __restrict__ sc, const double* __restrict__ x, PlanarFrames<N>& F) {
  d2 pos = mk2(sc->base_pos[0], sc->base_pos[1]);
  d2 R = mk2(sc->base_quat[0], sc->base_quat[1]);
  d2 acc = mk2(sc->base_acc[0], sc->base_acc[1]);
  double w = 0.0, alpha = 0.0;
#pragma unroll
  for (int j = 0; j < N; ++j) {
    double sn, cs;
    sincos(x[2 * j], &sn, &cs);
    F.cq[j] = cs;
    F.sq[j] = sn;
    // revolute_joint_2D::doMotion: position / acceleration of the base, rotation * rot(q), rates + (qd, qdd = 0)
    const d2 ER = rmul(R, mk2(cs, sn));
    w = w + x[2 * j + 1];
    alpha = alpha + 0.0;
    F.Epos[j] = pos;
    F.Erot[j] = ER;
    F.Eacc[j] = acc;
    F.w[j] = w;
    F.alpha[j] = alpha;
    // rigid_link_2D::doMotion
    const d2 off = mk2(sc->joints[j].off_pos[0], sc->joints[j].off_pos[1]);
    const d2 offR = mk2(sc->joints[j].off_quat[0], sc->joints[j].off_quat[1]);
    pos = pos + rrot(ER, off);
    acc = acc + rrot(ER, (-w * w) * off + cross_sv(alpha, off));
    R = rmul(ER, offR);
    F.Lpos[j] = pos;
    F.Lrot[j] = R;
    F.Lacc[j] = acc;
  }
}

// x' = f(x, u) of kte_nl_system::get_state_derivative for the planar chain; returns false on a singular mass matrix
template <int N>
RKH_DI bool planar_state_derivative(const SceneDev* __restrict__ sc, const double* __restrict__ x,
                                    const double* __restrict__ u, double* __restrict__ dp, double* M_out = nullptr,
                                    double* f_out = nullptr) {
  PlanarFrames<N> F;
  planar_motion<N>(sc, x, F);
  // kte_map_chain::doForce, reverse op order: inertia_2D, rigid_link_2D, revolute_joint_2D, inertia_gen, actuator per joint
  double f[N];
  d2 Lforce = mk2(0.0, 0.0);   // force / torque accumulated on the current link end frame (= next joint's base)
  double Ltorque = 0.0;
#pragma unroll
  for (int j = N - 1; j >= 0; --j) {
    const JointDev& J = sc->joints[j];
    // inertia_2D::doForce on the link's end frame
    Lforce = Lforce - J.mass * rrotT(F.Lacc[j], F.Lrot[j]);
    Ltorque = Ltorque - J.inertia[0] * F.alpha[j];
    // rigid_link_2D::doForce
    const d2 off = mk2(J.off_pos[0], J.off_pos[1]);
    const d2 offR = mk2(J.off_quat[0], J.off_quat[1]);
    const d2 tmp_force = rrot(offR, Lforce);
    const d2 Eforce = mk2(0.0, 0.0) + tmp_force;
    const double Etorque = 0.0 + (Ltorque + cross_vv(off, tmp_force));
    // revolute_joint_2D::doForce: base.Force += R(q) * end.Force; f += end.Torque (nothing reaches the base's torque)
    Lforce = mk2(0.0, 0.0) + rrot(mk2(F.cq[j], F.sq[j]), Eforce);
    f[j] = 0.0 + Etorque;
    // inertia_gen::doForce: f -= J q_ddot (q_ddot = 0); driving_actuator_gen::doForce: f += u, base.Torque -= u
    f[j] = f[j] - J.joint_inertia * 0.0;
    f[j] = f[j] + u[j];
    Ltorque = 0.0 - u[j];
  }
  // mass matrix: Tcm columns.  Rows: N generalized inertias, then (v_x, v_y, omega) of every inertia_2D
  double tv[N][N][2];  // tv[j][i] = qd_vel of joint i's jacobian relative to link j's end frame (i <= j)
#pragma unroll
  for (int j = 0; j < N; ++j) {
#pragma unroll
    for (int i = 0; i <= j; ++i) {
      // get_jac_relative_to: f2 = (~E_i) * L_j, v = (1 % f2.Position + 0) * f2.Rotation
      const d2 inv_pos = rrotT(-F.Epos[i], F.Erot[i]);
      const d2 inv_rot = mk2(F.Erot[i].x, -F.Erot[i].y);
      const d2 f2_pos = inv_pos + rrot(inv_rot, F.Lpos[j]);
      const d2 f2_rot = rmul(inv_rot, F.Lrot[j]);
      const d2 v = rrotT(cross_sv(1.0, f2_pos) + mk2(0.0, 0.0), f2_rot);
      tv[j][i][0] = v.x;
      tv[j][i][1] = v.y;
    }
  }
  // Mfull = Tcm^T (Mcm Tcm), summed over the rows in order (structural zeros skipped: they add exact zeros), then the
  // symmetric conversion 0.5 (M_ji + M_ij)
  double M[N][N];
#pragma unroll
  for (int i = 0; i < N; ++i) {
#pragma unroll
    for (int jj = 0; jj < N; ++jj) {
      double s = 0.0;
      if (i == jj) s = s + 1.0 * (sc->joints[i].joint_inertia * 1.0);
      const int j0 = i > jj ? i : jj;
#pragma unroll
      for (int j = 0; j < N; ++j) {
        if (j < j0) continue;
        const JointDev& J = sc->joints[j];
        s = s + tv[j][i][0] * (J.mass * tv[j][jj][0]);
        s = s + tv[j][i][1] * (J.mass * tv[j][jj][1]);
        s = s + 1.0 * (J.inertia[0] * 1.0);
      }
      M[i][jj] = s;
    }
  }
  double A[N][N];
#pragma unroll
  for (int i = 0; i < N; ++i) {
#pragma unroll
    for (int j = 0; j < i; ++j) {
      const double v = 0.5 * (M[j][i] + M[i][j]);
      A[i][j] = v;
      A[j][i] = v;
    }
    A[i][i] = M[i][i];
  }
  if (M_out)
    for (int i = 0; i < N; ++i)
      for (int j = 0; j < N; ++j) M_out[i * N + j] = A[i][j];
  if (f_out)
    for (int i = 0; i < N; ++i) f_out[i] = f[i];
  // decompose_Cholesky + backsub_Cholesky (mat_cholesky.hpp)
  double L[N][N];
  bool ok = true;
#pragma unroll
  for (int i = 0; i < N; ++i) {
#pragma unroll
    for (int j = 0; j < i; ++j) {
      double v = A[i][j];
#pragma unroll
      for (int k = 0; k < j; ++k) v = v - L[i][k] * L[j][k];
      L[i][j] = v / L[j][j];
    }
    double d = A[i][i];
#pragma unroll
    for (int k = 0; k < i; ++k) d = d - L[i][k] * L[i][k];
    if (d < 1e-8) ok = false;
    L[i][i] = sqrt(d);
  }
#pragma unroll
  for (int i = 0; i < N; ++i) {
#pragma unroll
    for (int k = 0; k < i; ++k) f[i] = f[i] - L[i][k] * f[k];
    f[i] = f[i] / L[i][i];
  }
#pragma unroll
  for (int i = N - 1; i >= 0; --i) {
#pragma unroll
    for (int k = N - 1; k > i; --k) f[i] = f[i] - L[k][i] * f[k];
    f[i] = f[i] / L[i][i];
  }
#pragma unroll
  for (int i = 0; i < N; ++i) {
    dp[2 * i] = x[2 * i + 1];
    dp[2 * i + 1] = f[i];
  }
  return ok;
}

// is_free(x): hyperbox bounds, then the planar proximity query in finder order
template <int N>
RKH_DI bool planar_is_free(const SceneDev* __restrict__ sc, const PairDev* __restrict__ pairs, int n_pairs,
                           const DynDev& dyn, const double* __restrict__ x) {
#pragma unroll
  for (int d = 0; d < 2 * N; ++d) {
    const double lo = dyn.lower[d], hi = dyn.upper[d];
    if (lo < hi) {
      if ((x[d] < lo) || (x[d] > hi)) return false;
    } else if ((x[d] > lo) || (x[d] < hi)) {
      return false;
    }
  }
  d2 Epos[N], Erot[N];
  {
    d2 pos = mk2(sc->base_pos[0], sc->base_pos[1]);
    d2 R = mk2(sc->base_quat[0], sc->base_quat[1]);
#pragma unroll
    for (int j = 0; j < N; ++j) {
      double sn, cs;
      sincos(x[2 * j], &sn, &cs);
      const d2 ER = rmul(R, mk2(cs, sn));
      Epos[j] = pos;
      Erot[j] = ER;
      pos = pos + rrot(ER, mk2(sc->joints[j].off_pos[0], sc->joints[j].off_pos[1]));
      R = rmul(ER, mk2(sc->joints[j].off_quat[0], sc->joints[j].off_quat[1]));
    }
  }
  double min_dist = INFINITY;
  for (int p = 0; p < n_pairs; ++p) {
    const PairDev pr = pairs[p];
    const ShapeDev& rs = sc->robot[pr.robot];
    const ShapeDev& es = sc->env[pr.env];
    ShapeP A, Bv;
    A.kind = rs.kind;
    {
      d2 pp = Epos[0], pq = Erot[0];
#pragma unroll
      for (int j = 1; j < N; ++j)
        if (rs.link == j) {
          pp = Epos[j];
          pq = Erot[j];
        }
      A.pos = pp + rrot(pq, mk2(rs.pos[0], rs.pos[1]));
      A.rot = rmul(pq, mk2(rs.quat[0], rs.quat[1]));
    }
    A.d0 = rs.dims[0]; A.d1 = rs.dims[1];
    Bv.kind = es.kind;
    Bv.pos = mk2(es.pos[0], es.pos[1]);
    Bv.rot = mk2(es.quat[0], es.quat[1]);
    Bv.d0 = es.dims[0]; Bv.d1 = es.dims[1];
    const ShapeP& s1 = pr.s1_is_robot ? A : Bv;
    const ShapeP& s2 = pr.s1_is_robot ? Bv : A;
    const double r1 = pr.s1_is_robot ? rs.brad : es.brad;
    const double r2 = pr.s1_is_robot ? es.brad : rs.brad;
    const double c = norm_2(to_parent(s2, mk2(0.0, 0.0)) - to_parent(s1, mk2(0.0, 0.0))) - r1 - r2;
    if (p == 0) {
      min_dist = pair_distance_planar(pr.routine, s1, s2);  // the first finder is always computed
    } else if (!(c > min_dist)) {
      const double d = pair_distance_planar(pr.routine, s1, s2);
      if (d < min_dist) min_dist = d;
    }
  }
  return !(min_dist < 0.0);
}

template <int N>
RKH_DI double planar_norm(const double* a, const double* b) {  // euclidean metric of the state difference, left to right
  double s = 0.0;
#pragma unroll
  for (int d = 0; d < 2 * N; ++d) {
    const double df = a[d] - b[d];
    s = s + df * df;
  }
  return sqrt(s);
}

// One lane per edge.  Grid: x = 64-edge blocks of group a, then of group b; y = problem (table launches).
template <int N>
__global__ __launch_bounds__(64) void planar_propagate_kernel(const SceneDev* __restrict__ sc, const PairDev* __restrict__ pairs,
                                                              int n_pairs, DynDev dyn, EdgeIO io_a, EdgeIO io_b,
                                                              const EdgeIO* __restrict__ tab_a,
                                                              const EdgeIO* __restrict__ tab_b, uint32_t blocks_a,
                                                              KernelGate gate) {
  if (gate.count) {
    const uint32_t c = *gate.count;
    if (c < gate.lo || c >= gate.hi) return;
  }
  constexpr int D = 2 * N;
  const bool group_b = blockIdx.x >= blocks_a;
  const EdgeIO io = tab_a ? (group_b ? tab_b[blockIdx.y] : tab_a[blockIdx.y]) : (group_b ? io_b : io_a);
  const uint32_t B = io.d_B ? *io.d_B : io.B;
  const uint32_t e = (group_b ? blockIdx.x - blocks_a : blockIdx.x) * 64u + threadIdx.x;
  if (e >= B) return;
  const uint32_t si = io.src_idx ? io.src_idx[e] : ((io.d_src_first ? *io.d_src_first : 0u) + e);
  const uint64_t trow = io.tgt_idx ? uint64_t(io.tgt_idx[e]) : (io.d_tgt_off ? uint64_t(*io.d_tgt_off) : 0ull) + e;
  double a[D], b[D], x[D];
#pragma unroll
  for (int d = 0; d < D; ++d) {
    a[d] = io.src[uint64_t(si) * io.src_stride + d];
    b[d] = io.tgt[trow * io.tgt_stride + d];
    x[d] = a[d];
  }
  double* __restrict__ record = io.record;
  const int record_stride = io.record_stride;
  if (record)
    for (int d = 0; d < D; ++d) record[(uint64_t(e) * record_stride + 0) * D + d] = x[d];
  int n_steps = dyn.n_steps;
  if (io.frac) {  // the edge's own travel fraction, cut with the steer loop's comparison
    const double T_goal = io.frac[e] * dyn.full_time;
    double current_time = 0.0;
    n_steps = 0;
    while (current_time < T_goal && n_steps < kMaxSteps) {
      current_time += dyn.dt;
      ++n_steps;
    }
  }
  uint32_t n_free = 0;
  bool singular = false;
  if (io.mode == EDGE_POINT) {
    n_steps = 0;
#pragma unroll
    for (int d = 0; d < D; ++d) x[d] = b[d];
    io.accept[e] = planar_is_free<N>(sc, pairs, n_pairs, dyn, x) ? 1 : 0;
  }
  for (int k = 0; k < n_steps; ++k) {
    if (!(planar_norm<N>(x, b) > dyn.goal_tol)) break;
    double u[N];
#pragma unroll
    for (int j = 0; j < N; ++j) {
      double v = dyn.kp * (b[2 * j] - x[2 * j]) + dyn.kd * (b[2 * j + 1] - x[2 * j + 1]);
      if (v > dyn.u_max) v = dyn.u_max;
      else if (v < -dyn.u_max) v = -dyn.u_max;
      u[j] = v;
    }
    // runge_kutta4_integrate_impl (runge_kutta4_integrator_sys.hpp:53-97), stages rolled like in propagate.hip
    const double h = dyn.dt;
    double xe[D], w[D], k1[D], k2[D], k3[D], dp[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
      xe[d] = x[d];
      w[d] = x[d];
      k1[d] = k2[d] = k3[d] = 0.0;
    }
    bool sing_now = false;
    const int n_evals = 4 * dyn.inner[k];
#pragma unroll 1
    for (int ev = 0; ev < n_evals; ++ev) {
      if (!planar_state_derivative<N>(sc, xe, u, dp)) sing_now = true;
      const int stage = ev & 3;
#pragma unroll
      for (int d = 0; d < D; ++d) {
        if (stage == 0) {
          w[d] = xe[d];
          k1[d] = h * dp[d];
          xe[d] = xe[d] + 0.5 * k1[d];
        } else if (stage == 1) {
          k2[d] = h * dp[d];
          xe[d] = w[d] + 0.5 * k2[d];
        } else if (stage == 2) {
          k3[d] = h * dp[d];
          xe[d] = w[d] + k3[d];
        } else {
          xe[d] = xe[d] + ((((1.0 / 6.0) * k1[d] + (2.0 / 6.0) * k2[d]) + (h / 6.0) * dp[d]) - (2.0 / 3.0) * k3[d]);
        }
      }
    }
    if (sing_now) {
      singular = true;
      break;
    }
    if (!planar_is_free<N>(sc, pairs, n_pairs, dyn, xe)) break;
#pragma unroll
    for (int d = 0; d < D; ++d) x[d] = xe[d];
    ++n_free;
    if (record)
      for (int d = 0; d < D; ++d) record[(uint64_t(e) * record_stride + n_free) * D + d] = x[d];
  }
  if (singular) atomicExch(io.err_flag, int(RKH_ERR_SINGULAR));
#pragma unroll
  for (int d = 0; d < D; ++d) io.x_out[uint64_t(e) * D + d] = x[d];
  io.steps_free[e] = n_free;
  if (io.mode != EDGE_PLAIN && io.mode != EDGE_POINT) {
    const double n_ar = planar_norm<N>(a, x);
    const double n_ab = planar_norm<N>(a, b);
    const double n_rb = planar_norm<N>(x, b);
    if (io.mode == EDGE_STEER_ACCEPT) {
      const double best_case = io.best_case ? io.best_case[e] : n_ab;
      io.accept[e] = ((!isinf(n_ar)) && (n_ar < 2.0 * best_case) && (n_ar > io.steer_tol * best_case)) ? 1 : 0;
    } else if (io.mode == EDGE_CONNECT) {
      io.accept[e] = ((!isinf(n_ar)) && (n_rb < io.steer_tol * n_ar)) ? 1 : 0;
    } else if (io.mode == EDGE_WALK_ACCEPT) {
      io.accept[e] = ((!isinf(n_ar)) && (n_ar > io.steer_tol * io.best_case[e])) ? 1 : 0;
    } else if (io.mode == EDGE_GOAL_PROBE) {
      io.goal_dist[si - 1] = (n_ab * 0.05 > n_rb) ? n_ab : INFINITY;
    }
  }
}

template <int N>
__global__ __launch_bounds__(64) void planar_state_derivative_kernel(const SceneDev* __restrict__ sc,
                                                                     const double* __restrict__ x,
                                                                     const double* __restrict__ u, uint32_t B,
                                                                     double* __restrict__ pd, double* __restrict__ M,
                                                                     double* __restrict__ f, int* __restrict__ err_flag) {
  const uint32_t e = blockIdx.x * 64u + threadIdx.x;
  if (e >= B) return;
  constexpr int D = 2 * N;
  double xs[D], us[N], dp[D], Mo[N * N], fo[N];
  for (int d = 0; d < D; ++d) xs[d] = x[uint64_t(e) * D + d];
  for (int j = 0; j < N; ++j) us[j] = u[uint64_t(e) * N + j];
  const bool ok = planar_state_derivative<N>(sc, xs, us, dp, Mo, fo);
  for (int d = 0; d < D; ++d) pd[uint64_t(e) * D + d] = dp[d];
  if (M)
    for (int i = 0; i < N * N; ++i) M[uint64_t(e) * N * N + i] = Mo[i];
  if (f)
    for (int j = 0; j < N; ++j) f[uint64_t(e) * N + j] = fo[j];
  if (!ok) atomicExch(err_flag, int(RKH_ERR_SINGULAR));
}

// ---- host side ---------------------------------------------------------------------------------------------------
static std::unordered_set<const void*>& planar_scenes() {
  static std::unordered_set<const void*> s;
  return s;
}
void register_planar_scene(const SceneDev* d_scene) { planar_scenes().insert(d_scene); }
void forget_planar_scene(const SceneDev* d_scene) { planar_scenes().erase(d_scene); }
bool is_planar_scene(const SceneDev* d_scene) { return planar_scenes().count(d_scene) != 0; }

#define RKH_DISPATCH_N_PLANAR(N_, CALL)                                                                  \
  switch (N_) {                                                                                          \
    case 1: { constexpr int N = 1; CALL; } break;                                                        \
    case 2: { constexpr int N = 2; CALL; } break;                                                        \
    case 3: { constexpr int N = 3; CALL; } break;                                                        \
    case 4: { constexpr int N = 4; CALL; } break;                                                        \
    case 6: { constexpr int N = 6; CALL; } break;                                                        \
    case 7: { constexpr int N = 7; CALL; } break;                                                        \
    default:                                                                                             \
      set_error("planar dynamics: chains with this number of joints are not instantiated (1,2,3,4,6,7)"); \
      return RKH_ERR_UNSUPPORTED;                                                                        \
  }

rkh_status launch_propagate_planar(hipStream_t s, int n_dof, const SceneDev* d_scene, const void* d_pairs, int n_pairs,
                                   const DynDev& dyn, const EdgeIO& io, uint32_t grid_edges, const EdgeIO* io_b,
                                   uint32_t grid_b, const EdgeIO* tab_a, const EdgeIO* tab_b, uint32_t n_problems,
                                   KernelGate gate) {
  const uint32_t eb = (io_b || tab_b) ? grid_b : 0u;
  if (grid_edges + eb == 0 || n_problems == 0) return RKH_OK;
  const uint32_t blocks_a = (grid_edges + 63) / 64, blocks_b = (eb + 63) / 64;
  const EdgeIO second = io_b ? *io_b : EdgeIO();
  gate.wave_base = nullptr;  // the compact wave numbering of the 3D mappings does not apply: plain (block, problem) grid
  gate.n_segments = 0;
  RKH_DISPATCH_N_PLANAR(n_dof, hipLaunchKernelGGL((planar_propagate_kernel<N>), dim3(blocks_a + blocks_b, n_problems), dim3(64),
                                                  0, s, d_scene, static_cast<const PairDev*>(d_pairs), n_pairs, dyn, io, second,
                                                  tab_a, tab_b, blocks_a, gate));
  RKH_HIP(hipGetLastError());
  return RKH_OK;
}

rkh_status launch_state_derivative_planar(hipStream_t s, int n_dof, const SceneDev* d_scene, const double* d_x,
                                          const double* d_u, uint32_t B, double* d_pd, double* d_M, double* d_f, int* d_err) {
  if (B == 0) return RKH_OK;
  RKH_DISPATCH_N_PLANAR(n_dof, hipLaunchKernelGGL((planar_state_derivative_kernel<N>), dim3((B + 63) / 64), dim3(64), 0, s,
                                                  d_scene, d_x, d_u, B, d_pd, d_M, d_f, d_err));
  RKH_HIP(hipGetLastError());
  return RKH_OK;
}

}  // namespace rkh
