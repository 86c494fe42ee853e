// propagate.hip -- one wavefront per candidate edge: RK4 forward-dynamics propagation of a KTE
// serial chain with a proximity (collision) test after every step (gfx950, wave64).
//
// What it replaces (paths relative to /root/reference/src/ReaK/):
//   steer loop        examples/misc/MEAQR_topology.hpp:503-565 (steer_with_constant_control pattern)
//   RK4               ctrl/sys_integrators/runge_kutta4_integrator_sys.hpp:53-97
//   x' = f(x,u)       ctrl/ctrl_sys/kte_nl_system.hpp:180-290 (apply_states_and_inputs, get_state_derivative)
//   KTE passes        ctrl/mbd_kte/kte_map_chain.hpp:71-89; revolute_joint.cpp:121-213; rigid_link.cpp:152-186;
//                     inertia.cpp:47-54,111-122; driving_actuator.cpp:31-39
//   mass matrix       ctrl/mbd_kte/mass_matrix_calculator.cpp:80-295 + core/kinetostatics/motion_jacobians.hpp:238-251
//   Cholesky solve    core/lin_alg/mat_cholesky.hpp:63-84,160-178,546-554
//   is_free           ctrl/topologies/manip_free_workspace.hpp:79-99 + geometry/proximity (proximity_device.h)
//
// Mapping onto the wave (block = 1 wave = 1 edge):
//   * state x, RK4 temporaries, bounds: lane d < 2N owns component d  (registers, 1 double each)
//   * sin/cos of the N joint angles (half- and full-angle): lanes 0..2N-1 in parallel
//   * base->tip kinematic sweep and tip->base force sweep: serial by nature; every lane runs them on
//     wave-uniform values (no cross-lane traffic), joints fully unrolled (template N)
//   * Jacobian columns Tcm(body b, coord c<=b): one lane per (b,c) pair; M(i,j): one lane per entry
//   * proximity pairs: one lane per (robot shape, obstacle) pair, ballot for "any distance < 0";
//     obstacle table staged in LDS, chain parameters read as wave-uniform (scalar) loads.
// Compiled with -ffp-contract=off: products and sums round exactly as in the CPU reference; only
// sin/cos (OCML vs glibc) differ by ulps (stated tolerance: 1e-10 relative on propagated states).
#include <hip/hip_runtime.h>

#include <cmath>

#include "device_math.h"
#include "proximity_device.h"
#include "rkh_internal.h"

namespace rkh {

template <int N>
struct WaveWs {  // per-wave LDS workspace
  double Epos[N][3], Equat[N][4];  // joint end frames (jacobian parents)
  double Lpos[N][3], Lquat[N][4];  // link end frames (inertia frames)
  double Tcm[N][N][6];             // [body][coord] jacobian column (v, w)
  double M[N][N];
  double Rpos[2 * N][3], Rquat[2 * N][4];  // robot shapes, global pose
};

template <int N>
struct ChainRegs {  // wave-uniform per-joint values carried from the forward to the backward sweep
  double c1[N], s1[N];  // cos/sin of the full joint angle
  d3 Fi[N], Ti[N];      // inertia_3D d'Alembert force / torque (to be subtracted)
};

// revolute_joint_3D / rigid_link_3D kinematics of joint j (position + orientation only)
template <int N>
RKH_DI void fk_pose_chain(const SceneDev* __restrict__ sc, const double (&c2)[N], const double (&s2)[N], WaveWs<N>& ws,
                          int lane) {
  d3 pos = mk3(sc->base_pos[0], sc->base_pos[1], sc->base_pos[2]);
  d4 Q = d4{sc->base_quat[0], sc->base_quat[1], sc->base_quat[2], sc->base_quat[3]};
#pragma unroll
  for (int j = 0; j < N; ++j) {
    const JointDev& jd = sc->joints[j];
    const d4 tq = d4{c2[j], jd.axis_n[0] * s2[j], jd.axis_n[1] * s2[j], jd.axis_n[2] * s2[j]};
    const d4 EQ = qmul(Q, tq);
    if (lane == 0) {
      ws.Epos[j][0] = pos.x; ws.Epos[j][1] = pos.y; ws.Epos[j][2] = pos.z;
      ws.Equat[j][0] = EQ.w; ws.Equat[j][1] = EQ.x; ws.Equat[j][2] = EQ.y; ws.Equat[j][3] = EQ.z;
    }
    const m33 R = rotmat(EQ);
    pos = pos + mul(R, mk3(jd.off_pos[0], jd.off_pos[1], jd.off_pos[2]));
    Q = qmul(EQ, d4{jd.off_quat[0], jd.off_quat[1], jd.off_quat[2], jd.off_quat[3]});
    if (lane == 0) {
      ws.Lpos[j][0] = pos.x; ws.Lpos[j][1] = pos.y; ws.Lpos[j][2] = pos.z;
      ws.Lquat[j][0] = Q.w; ws.Lquat[j][1] = Q.x; ws.Lquat[j][2] = Q.y; ws.Lquat[j][3] = Q.z;
    }
  }
}

// axis_angle::getRotMat (rotations_3D.hpp:2160-2180) from cos/sin of the angle and the unit axis
RKH_DI m33 axis_angle_rotmat(double ca, double sa, d3 ax) {
  const double omc = 1.0 - ca;
  const double t11 = ca + omc * ax.x * ax.x, t22 = ca + omc * ax.y * ax.y, t33 = ca + omc * ax.z * ax.z;
  const double t12 = omc * ax.x * ax.y, t13 = omc * ax.x * ax.z, t23 = omc * ax.y * ax.z;
  const double t01 = sa * ax.x, t02 = sa * ax.y, t03 = sa * ax.z;
  return m33{t11, t12 - t03, t13 + t02, t12 + t03, t22, t23 - t01, t13 - t02, t23 + t01, t33};
}

// x' = f(x,u).  xv: lane d < 2N holds x[d]; uv: lane j < N holds u[j].
// Returns dp for lane d (< 2N); sets *singular if a Cholesky pivot is < 1e-8.
// If M_out/f_out (global, optional) are given, lane-parallel copies of M and the bias force are written.
template <int N>
__device__ double state_derivative(const SceneDev* __restrict__ sc, WaveWs<N>& ws, double xv, double uv, int lane,
                                   bool* singular, double* M_out, double* f_out) {
  // ---- sin/cos, lane-parallel: lane 2j -> half angle, lane 2j+1 -> full angle
  const double q_here = __shfl(xv, lane & ~1, 64);
  double sn, cs;
  sincos((lane & 1) ? q_here : 0.5 * q_here, &sn, &cs);
  double c2[N], s2[N], qd[N], u[N];
  ChainRegs<N> cr;
#pragma unroll
  for (int j = 0; j < N; ++j) {
    c2[j] = __shfl(cs, 2 * j, 64);
    s2[j] = __shfl(sn, 2 * j, 64);
    cr.c1[j] = __shfl(cs, 2 * j + 1, 64);
    cr.s1[j] = __shfl(sn, 2 * j + 1, 64);
    qd[j] = __shfl(xv, 2 * j + 1, 64);
    u[j] = __shfl(uv, j, 64);
  }

  // ---- base -> tip sweep (kte_map_chain::doMotion), wave-uniform
  {
    d3 pos = mk3(sc->base_pos[0], sc->base_pos[1], sc->base_pos[2]);
    d4 Q = d4{sc->base_quat[0], sc->base_quat[1], sc->base_quat[2], sc->base_quat[3]};
    d3 w = mk3(0, 0, 0), alpha = mk3(0, 0, 0);
    d3 acc = mk3(sc->base_acc[0], sc->base_acc[1], sc->base_acc[2]);
#pragma unroll
    for (int j = 0; j < N; ++j) {
      const JointDev& jd = sc->joints[j];
      const d3 axis = mk3(jd.axis[0], jd.axis[1], jd.axis[2]);
      // revolute_joint_3D::doMotion (revolute_joint.cpp:121-148)
      const d4 tq = d4{c2[j], jd.axis_n[0] * s2[j], jd.axis_n[1] * s2[j], jd.axis_n[2] * s2[j]};
      const m33 R2 = rotmat(tq);
      const d4 EQ = qmul(Q, tq);
      const d3 wb = mulT(w, R2);
      const d3 qa = qd[j] * axis;
      const d3 Ew = wb + qa;
      const d3 Ealpha = mulT(alpha, R2) + cross(wb, qa);
      if (lane == 0) {
        ws.Epos[j][0] = pos.x; ws.Epos[j][1] = pos.y; ws.Epos[j][2] = pos.z;
        ws.Equat[j][0] = EQ.w; ws.Equat[j][1] = EQ.x; ws.Equat[j][2] = EQ.y; ws.Equat[j][3] = EQ.z;
      }
      // rigid_link_3D::doMotion = frame * pose (frame_3D.hpp:240-255)
      const d3 op = mk3(jd.off_pos[0], jd.off_pos[1], jd.off_pos[2]);
      const m33 R = rotmat(EQ);
      pos = pos + mul(R, op);
      acc = acc + mul(R, cross(Ew, cross(Ew, op)) + cross(Ealpha, op));
      const m33 Ro = m33{jd.off_R[0], jd.off_R[1], jd.off_R[2], jd.off_R[3], jd.off_R[4],
                         jd.off_R[5], jd.off_R[6], jd.off_R[7], jd.off_R[8]};
      Q = qmul(EQ, d4{jd.off_quat[0], jd.off_quat[1], jd.off_quat[2], jd.off_quat[3]});
      alpha = mulT(Ealpha, Ro);
      w = mulT(Ew, Ro);
      if (lane == 0) {
        ws.Lpos[j][0] = pos.x; ws.Lpos[j][1] = pos.y; ws.Lpos[j][2] = pos.z;
        ws.Lquat[j][0] = Q.w; ws.Lquat[j][1] = Q.x; ws.Lquat[j][2] = Q.y; ws.Lquat[j][3] = Q.z;
      }
      // inertia_3D::doForce terms (inertia.cpp:111-122), applied in the backward sweep
      cr.Fi[j] = jd.mass * qrot(qinv(Q), acc);
      cr.Ti[j] = sym_mul(jd.inertia, alpha) + cross(w, sym_mul(jd.inertia, w));
    }
  }
  __syncthreads();

  // ---- jacobian columns, one lane per (body b, coord c <= b): get_jac_relative_to
  //      (motion_jacobians.hpp:238-251) with f2 = (~F_c) * F_b (frame_3D.hpp:184-189,222-238,368-382)
  {
    int b = 0, c = lane;  // unrank lane -> (b, c), c <= b
    while (c > b) {
      c -= b + 1;
      ++b;
    }
    if (b < N) {
      const d3 cp = mk3(ws.Epos[c][0], ws.Epos[c][1], ws.Epos[c][2]);
      const d4 cq = d4{ws.Equat[c][0], ws.Equat[c][1], ws.Equat[c][2], ws.Equat[c][3]};
      const d3 bp = mk3(ws.Lpos[b][0], ws.Lpos[b][1], ws.Lpos[b][2]);
      const d4 bq = d4{ws.Lquat[b][0], ws.Lquat[b][1], ws.Lquat[b][2], ws.Lquat[b][3]};
      const m33 R = rotmat(cq);
      const d4 iq = qinv(cq);
      const d3 ipos = mulT(-cp, R);
      const m33 Ri = rotmat(iq);
      const d3 f2pos = ipos + mul(Ri, bp);
      const d4 f2q = qmul(iq, bq);
      const m33 Rf = rotmat(f2q);
      const d3 axis = mk3(sc->joints[c].axis[0], sc->joints[c].axis[1], sc->joints[c].axis[2]);
      const d3 wt = mulT(axis, Rf);
      const d3 vt = mulT(cross(axis, f2pos), Rf);
      ws.Tcm[b][c][0] = vt.x; ws.Tcm[b][c][1] = vt.y; ws.Tcm[b][c][2] = vt.z;
      ws.Tcm[b][c][3] = wt.x; ws.Tcm[b][c][4] = wt.y; ws.Tcm[b][c][5] = wt.z;
    }
  }

  // ---- tip -> base sweep (kte_map_chain::doForce in reverse op order), wave-uniform
  double f[N];
  {
    d3 LF = mk3(0, 0, 0), LT = mk3(0, 0, 0);
#pragma unroll
    for (int j = N - 1; j >= 0; --j) {
      const JointDev& jd = sc->joints[j];
      const d3 axis = mk3(jd.axis[0], jd.axis[1], jd.axis[2]);
      // inertia_3D::doForce on the link end frame
      LF = LF - cr.Fi[j];
      LT = LT - cr.Ti[j];
      // rigid_link_3D::doForce (rigid_link.cpp:170-178)
      const m33 Ro = m33{jd.off_R[0], jd.off_R[1], jd.off_R[2], jd.off_R[3], jd.off_R[4],
                         jd.off_R[5], jd.off_R[6], jd.off_R[7], jd.off_R[8]};
      const d3 op = mk3(jd.off_pos[0], jd.off_pos[1], jd.off_pos[2]);
      const d3 tmp_force = mul(Ro, LF);
      const d3 EF = tmp_force;
      const d3 ET = mul(Ro, LT) + cross(op, tmp_force);
      // revolute_joint_3D::doForce (revolute_joint.cpp:170-181)
      const m33 Ra = axis_angle_rotmat(cr.c1[j], cr.s1[j], mk3(jd.axis_n[0], jd.axis_n[1], jd.axis_n[2]));
      const double ta = dot(ET, axis);
      LF = mul(Ra, EF);
      LT = mul(Ra, ET - ta * axis);
      // inertia_gen::doForce: f -= q_ddot * mass with q_ddot = 0 ; driving_actuator_gen::doForce
      f[j] = ta + u[j];
      LT = LT - u[j] * axis;
    }
  }
  __syncthreads();

  // ---- M = Tcm^T (Mcm Tcm), one lane per entry (i,j), summation order of
  //      mat_alg_symmetric.hpp:551-566 (Mcm*Tcm) and mat_operators.hpp:104-114 (dense product)
  {
    const int i = lane / N, jx = lane % N;
    double s = 0.0;
    if (lane < N * N) {
      if (i == jx) s = s + sc->joints[i].joint_inertia;  // inertia_gen rows: Tcm = 1, Mcm = rotor inertia
#pragma unroll
      for (int b = 0; b < N; ++b) {
        if (b >= i && b >= jx) {
          const JointDev& jd = sc->joints[b];
          const double* Ti = ws.Tcm[b][i];
          const double* Tj = ws.Tcm[b][jx];
          s = s + Ti[0] * (jd.mass * Tj[0]);
          s = s + Ti[1] * (jd.mass * Tj[1]);
          s = s + Ti[2] * (jd.mass * Tj[2]);
          const d3 P = sym_mul(jd.inertia, mk3(Tj[3], Tj[4], Tj[5]));
          s = s + Ti[3] * P.x;
          s = s + Ti[4] * P.y;
          s = s + Ti[5] * P.z;
        }
      }
    }
    // mat<symmetric>(general): 0.5 * (M(j,i) + M(i,j))  (mat_alg_symmetric.hpp:183-187)
    const int tl = (lane < N * N) ? (jx * N + i) : lane;
    const double st = __shfl(s, tl, 64);
    if (lane < N * N) {
      const double m = (i == jx) ? s : ((i > jx) ? 0.5 * (st + s) : 0.5 * (s + st));
      ws.M[i][jx] = m;
      if (M_out) M_out[lane] = m;
    }
  }
  if (f_out && lane < N) {
    double fv = 0.0;
#pragma unroll
    for (int j = 0; j < N; ++j) fv = (lane == j) ? f[j] : fv;
    f_out[lane] = fv;
  }
  __syncthreads();

  // ---- linsolve_Cholesky (mat_cholesky.hpp:546-554), wave-uniform
  double L[N][N];
  bool sing = false;
#pragma unroll
  for (int i = 0; i < N; ++i) {
#pragma unroll
    for (int j = 0; j < i; ++j) {
      double v = ws.M[i][j];
#pragma unroll
      for (int k = 0; k < j; ++k) v = v - L[i][k] * L[j][k];
      L[i][j] = v / L[j][j];
    }
    double dgl = ws.M[i][i];
#pragma unroll
    for (int k = 0; k < i; ++k) dgl = dgl - L[i][k] * L[i][k];
    if (dgl < 1e-8) sing = true;
    L[i][i] = sqrt(dgl);
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
  if (sing) *singular = true;

  // pd[2j] = q_dot_j ; pd[2j+1] = qdd_j  (kte_nl_system.hpp:276-279)
  double out = __shfl(xv, lane | 1, 64);
  if (lane & 1) {
#pragma unroll
    for (int j = 0; j < N; ++j) out = (lane == 2 * j + 1) ? f[j] : out;
  }
  return out;
}

// Proximity verdict for the configuration whose joint half-angle sin/cos are (c2, s2):
// returns the minimum distance over computed pairs; with cull_positive, pairs whose bounding
// spheres are apart are skipped (they cannot make the verdict "colliding",
// proxy_query_model.cpp:386-389 culls the same way against the running minimum).
template <int N>
__device__ double proximity_min(const SceneDev* __restrict__ sc, const ShapeDev* __restrict__ env_lds,
                                const PairDev* __restrict__ pairs, int n_pairs, WaveWs<N>& ws,
                                const double (&c2)[N], const double (&s2)[N], int lane, bool cull_positive) {
  fk_pose_chain<N>(sc, c2, s2, ws, lane);
  __syncthreads();
  // robot shapes -> global pose (pose_3D::getGlobalPose, pose_3D.hpp:102-110), lane r < n_robot
  if (lane < sc->n_robot) {
    const ShapeDev& sh = sc->robot[lane];
    const int j = sh.link;
    const d3 pp = mk3(ws.Epos[j][0], ws.Epos[j][1], ws.Epos[j][2]);
    const d4 pq = d4{ws.Equat[j][0], ws.Equat[j][1], ws.Equat[j][2], ws.Equat[j][3]};
    const d3 gp = pp + qrot(pq, mk3(sh.pos[0], sh.pos[1], sh.pos[2]));
    const d4 gq = qmul(pq, d4{sh.quat[0], sh.quat[1], sh.quat[2], sh.quat[3]});
    ws.Rpos[lane][0] = gp.x; ws.Rpos[lane][1] = gp.y; ws.Rpos[lane][2] = gp.z;
    ws.Rquat[lane][0] = gq.w; ws.Rquat[lane][1] = gq.x; ws.Rquat[lane][2] = gq.y; ws.Rquat[lane][3] = gq.z;
  }
  __syncthreads();
  double dmin = INFINITY;
  for (int p0 = 0; p0 < n_pairs; p0 += 64) {
    const int p = p0 + lane;
    double d = INFINITY;
    if (p < n_pairs) {
      const PairDev pr = pairs[p];
      const ShapeDev& rs = sc->robot[pr.robot];
      const ShapeDev& es = env_lds[pr.env];
      ShapeG A, Bv;
      A.kind = rs.kind;
      A.pos = mk3(ws.Rpos[pr.robot][0], ws.Rpos[pr.robot][1], ws.Rpos[pr.robot][2]);
      A.q = d4{ws.Rquat[pr.robot][0], ws.Rquat[pr.robot][1], ws.Rquat[pr.robot][2], ws.Rquat[pr.robot][3]};
      A.d0 = rs.dims[0]; A.d1 = rs.dims[1]; A.d2 = rs.dims[2];
      Bv.kind = es.kind;
      Bv.pos = mk3(es.pos[0], es.pos[1], es.pos[2]);
      Bv.q = d4{es.quat[0], es.quat[1], es.quat[2], es.quat[3]};
      Bv.d0 = es.dims[0]; Bv.d1 = es.dims[1]; Bv.d2 = es.dims[2];
      bool skip = false;
      if (cull_positive) {
        // transformToGlobal(0) of both shapes, then |c2 - c1| - r1 - r2 (proxy_query_model.cpp:384-389)
        const d3 c1 = pr.s1_is_robot ? pose_to_parent(A.pos, A.q, mk3(0, 0, 0)) : pose_to_parent(Bv.pos, Bv.q, mk3(0, 0, 0));
        const d3 c2p = pr.s1_is_robot ? pose_to_parent(Bv.pos, Bv.q, mk3(0, 0, 0)) : pose_to_parent(A.pos, A.q, mk3(0, 0, 0));
        const double r1 = pr.s1_is_robot ? rs.brad : es.brad;
        const double r2 = pr.s1_is_robot ? es.brad : rs.brad;
        skip = (norm_2(c2p - c1) - r1 - r2 > 0.0);
      }
      if (!skip) d = pr.s1_is_robot ? pair_distance(pr.routine, A, Bv) : pair_distance(pr.routine, Bv, A);
    }
    if (d < dmin) dmin = d;
    if (cull_positive && __ballot(d < 0.0) != 0ull) break;
  }
  // wave min
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const double o = __shfl_xor(dmin, off, 64);
    if (o < dmin) dmin = o;
  }
  return dmin;
}

__device__ __forceinline__ void stage_env(const SceneDev* __restrict__ sc, ShapeDev* env_lds, int lane) {
  const int n_words = sc->n_env * int(sizeof(ShapeDev) / sizeof(double));
  const double* src = reinterpret_cast<const double*>(sc->env);
  double* dst = reinterpret_cast<double*>(env_lds);
  for (int i = lane; i < n_words; i += 64) dst[i] = src[i];
}

// ---------------------------------------------------------------------------------------------
// Kernel: steer B edges. One block (= one wave) per edge.
//   a: source state of edge e = src + src_idx[e] * src_stride (src_idx may be null -> e)
//   tgt: [B][2N] target states
// Outputs: x_out [B][2N] last free state, steps_free[B], record (optional) [B][n_steps+1][2N]
template <int N>
__global__ __launch_bounds__(64) void propagate_kernel(const SceneDev* __restrict__ sc, const PairDev* __restrict__ pairs,
                                                        int n_pairs, DynDev dyn, EdgeIO io_a, EdgeIO io_b,
                                                        uint32_t grid_a) {
  // two edge groups per launch (planner: this round's steer candidates + the previous round's goal probes)
  const bool group_b = blockIdx.x >= grid_a;
  const EdgeIO& io = group_b ? io_b : io_a;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  WaveWs<N>& ws = *reinterpret_cast<WaveWs<N>*>(smem_raw);
  ShapeDev* env_lds = reinterpret_cast<ShapeDev*>(smem_raw + ((sizeof(WaveWs<N>) + 15) / 16) * 16);
  const uint32_t B = io.d_B ? *io.d_B : io.B;
  const uint32_t e = group_b ? blockIdx.x - grid_a : blockIdx.x;
  if (e >= B) return;
  const int lane = threadIdx.x;
  constexpr int D = 2 * N;
  stage_env(sc, env_lds, lane);
  const uint32_t si = io.src_idx ? io.src_idx[e] : ((io.d_src_first ? *io.d_src_first : 0u) + e);
  const uint64_t trow = (io.d_tgt_off ? uint64_t(*io.d_tgt_off) : 0ull) + e;
  const double a_d = (lane < D) ? io.src[uint64_t(si) * io.src_stride + lane] : 0.0;
  const double b_d = (lane < D) ? io.tgt[trow * io.tgt_stride + lane] : 0.0;
  const double lo = (lane < D) ? dyn.lower[lane] : 0.0;
  const double hi = (lane < D) ? dyn.upper[lane] : 0.0;
  double* __restrict__ record = io.record;
  const int record_stride = io.record_stride;
  __syncthreads();

  double x = a_d;
  uint32_t n_free = 0;
  bool singular = false;
  if (record && lane < D) record[(uint64_t(e) * record_stride + 0) * D + lane] = x;

  for (int k = 0; k < dyn.n_steps; ++k) {
    // distance(x_current, x_goal) > goal_proximity_threshold, exact left-to-right sum
    {
      double s = 0.0;
      const double df = x - b_d;
      const double sq = df * df;
#pragma unroll
      for (int d = 0; d < D; ++d) s = s + __shfl(sq, d, 64);
      if (!(sqrt(s) > dyn.goal_tol)) break;
    }
    // PD law, zero-order hold over the step: lane j gets u_j
    double uv;
    {
      const double eq = __shfl(b_d, 2 * (lane % N), 64) - __shfl(x, 2 * (lane % N), 64);
      const double ev = __shfl(b_d, 2 * (lane % N) + 1, 64) - __shfl(x, 2 * (lane % N) + 1, 64);
      double v = dyn.kp * eq + dyn.kd * ev;
      if (v > dyn.u_max) v = dyn.u_max;
      else if (v < -dyn.u_max) v = -dyn.u_max;
      uv = v;
    }
    // runge_kutta4_integrate_impl (runge_kutta4_integrator_sys.hpp:53-97), time_step = dt.
    // One call site for f(x,u): each loop iteration of the reference evaluates f four times that
    // matter (the prime of :69 or the re-prime of :95, then :82, :86, :92); they are the stages of
    // a rolled loop here, which keeps a single copy of the dynamics in the instruction stream.
    // The re-prime after the last iteration is dead in the reference and is not evaluated.
    const double h = dyn.dt;
    double xe = x;  // end_point
    {
      double w = xe, k1 = 0.0, k2 = 0.0, k3 = 0.0;
      const int n_evals = 4 * dyn.inner[k];
#pragma unroll 1
      for (int ev = 0; ev < n_evals; ++ev) {
        const double dp = state_derivative<N>(sc, ws, xe, uv, lane, &singular, nullptr, nullptr);
        const int stage = ev & 3;
        if (stage == 0) {
          w = xe;
          k1 = h * dp;
          xe = xe + 0.5 * k1;
        } else if (stage == 1) {
          k2 = h * dp;
          xe = w + 0.5 * k2;
        } else if (stage == 2) {
          k3 = h * dp;
          xe = w + k3;
        } else {
          xe = xe + ((((1.0 / 6.0) * k1 + (2.0 / 6.0) * k2) + (h / 6.0) * dp) - (2.0 / 3.0) * k3);
        }
      }
    }
    if (singular) break;
    // is_free(x_next): hyperbox bounds (hyperbox_topology.hpp:178-189), then proximity
    bool oob = false;
    if (lane < D) {
      if (lo < hi) oob = (xe < lo) || (xe > hi);
      else oob = (xe > lo) || (xe < hi);
    }
    if (__ballot(oob) != 0ull) break;
    {
      double sn, cs;
      sincos(0.5 * xe, &sn, &cs);
      double c2[N], s2[N];
#pragma unroll
      for (int j = 0; j < N; ++j) {
        c2[j] = __shfl(cs, 2 * j, 64);
        s2[j] = __shfl(sn, 2 * j, 64);
      }
      const double dmin = proximity_min<N>(sc, env_lds, pairs, n_pairs, ws, c2, s2, lane, true);
      if (dmin < 0.0) break;
    }
    x = xe;
    ++n_free;
    if (record && lane < D) record[(uint64_t(e) * record_stride + n_free) * D + lane] = x;
  }
  if (singular && lane == 0) atomicExch(io.err_flag, int(RKH_ERR_SINGULAR));
  if (lane < D) io.x_out[uint64_t(e) * D + lane] = x;
  if (lane == 0) io.steps_free[e] = n_free;
  if (io.mode != EDGE_PLAIN) {
    // exact left-to-right euclidean metrics (vect_distance_metrics.hpp:126-137)
    double s_ar = 0.0, s_ab = 0.0, s_rb = 0.0;
    {
      const double d1 = a_d - x, d2 = a_d - b_d, d3v = x - b_d;
      const double q1 = d1 * d1, q2 = d2 * d2, q3 = d3v * d3v;
#pragma unroll
      for (int d = 0; d < D; ++d) {
        s_ar = s_ar + __shfl(q1, d, 64);
        s_ab = s_ab + __shfl(q2, d, 64);
        s_rb = s_rb + __shfl(q3, d, 64);
      }
    }
    if (io.mode == EDGE_STEER_ACCEPT) {
      // planning_visitor_base::steer_towards_position (planning_visitors.hpp:349-360)
      const double traveled = sqrt(s_ar);
      const double best_case = io.best_case ? io.best_case[e] : sqrt(s_ab);
      const bool ok = (!isinf(traveled)) && (traveled < 2.0 * best_case) && (traveled > io.steer_tol * best_case);
      if (lane == 0) io.accept[e] = ok ? 1 : 0;
    } else {
      // C_free distance used by the goal probe (MEAQR_topology.hpp:995-1003)
      const double dab = sqrt(s_ab), drb = sqrt(s_rb);
      if (lane == 0) io.goal_dist[si - 1] = (dab * 0.05 > drb) ? dab : INFINITY;
    }
  }
}

// Kernel: x' = f(x,u) for B states (one wave each); also exports M and the bias force.
template <int N>
__global__ __launch_bounds__(64) void state_derivative_kernel(const SceneDev* __restrict__ sc, const double* __restrict__ x,
                                                               const double* __restrict__ u, uint32_t B,
                                                               double* __restrict__ pd, double* __restrict__ M,
                                                               double* __restrict__ f, int* __restrict__ err_flag) {
  __shared__ WaveWs<N> ws;
  const uint32_t e = blockIdx.x;
  if (e >= B) return;
  const int lane = threadIdx.x;
  constexpr int D = 2 * N;
  const double xv = (lane < D) ? x[uint64_t(e) * D + lane] : 0.0;
  const double uv = (lane < N) ? u[uint64_t(e) * N + lane] : 0.0;
  bool singular = false;
  const double dp = state_derivative<N>(sc, ws, xv, uv, lane, &singular, M ? M + uint64_t(e) * N * N : nullptr,
                                        f ? f + uint64_t(e) * N : nullptr);
  if (lane < D) pd[uint64_t(e) * D + lane] = dp;
  if (singular && lane == 0) atomicExch(err_flag, int(RKH_ERR_SINGULAR));
}

// Kernel: exact minimum proxy-pair distance for B states (no culling).
template <int N>
__global__ __launch_bounds__(64) void min_distance_kernel(const SceneDev* __restrict__ sc, const PairDev* __restrict__ pairs,
                                                           int n_pairs, const double* __restrict__ x, uint32_t B,
                                                           double* __restrict__ dist) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  WaveWs<N>& ws = *reinterpret_cast<WaveWs<N>*>(smem_raw);
  ShapeDev* env_lds = reinterpret_cast<ShapeDev*>(smem_raw + ((sizeof(WaveWs<N>) + 15) / 16) * 16);
  const uint32_t e = blockIdx.x;
  if (e >= B) return;
  const int lane = threadIdx.x;
  constexpr int D = 2 * N;
  stage_env(sc, env_lds, lane);
  const double xv = (lane < D) ? x[uint64_t(e) * D + lane] : 0.0;
  double sn, cs;
  sincos(0.5 * xv, &sn, &cs);
  double c2[N], s2[N];
#pragma unroll
  for (int j = 0; j < N; ++j) {
    c2[j] = __shfl(cs, 2 * j, 64);
    s2[j] = __shfl(sn, 2 * j, 64);
  }
  __syncthreads();
  const double dmin = proximity_min<N>(sc, env_lds, pairs, n_pairs, ws, c2, s2, lane, false);
  if (lane == 0) dist[e] = dmin;
}

// ---- host launchers ------------------------------------------------------------------------
#define RKH_DISPATCH_N(N_, CALL)     \
  switch (N_) {                      \
    case 1: { constexpr int N = 1; CALL; } break; \
    case 2: { constexpr int N = 2; CALL; } break; \
    case 3: { constexpr int N = 3; CALL; } break; \
    case 6: { constexpr int N = 6; CALL; } break; \
    default:                         \
      set_error("propagate: chains with this number of joints are not instantiated (1,2,3,6)"); \
      return RKH_ERR_UNSUPPORTED;    \
  }

template <int N>
static size_t smem_bytes(int n_env) {
  return ((sizeof(WaveWs<N>) + 15) / 16) * 16 + size_t(n_env) * sizeof(ShapeDev);
}

rkh_status launch_propagate(hipStream_t s, int n_dof, int n_env, const SceneDev* d_scene, const void* d_pairs,
                            int n_pairs, const DynDev& dyn, const EdgeIO& io, uint32_t grid_edges, const EdgeIO* io_b,
                            uint32_t grid_b) {
  if (grid_edges + grid_b == 0) return RKH_OK;
  const EdgeIO second = io_b ? *io_b : EdgeIO();
  RKH_DISPATCH_N(n_dof, hipLaunchKernelGGL((propagate_kernel<N>), dim3(grid_edges + (io_b ? grid_b : 0u)), dim3(64),
                                           smem_bytes<N>(n_env), s, d_scene, static_cast<const PairDev*>(d_pairs),
                                           n_pairs, dyn, io, second, grid_edges));
  RKH_HIP(hipGetLastError());
  return RKH_OK;
}

rkh_status launch_state_derivative(hipStream_t s, int n_dof, const SceneDev* d_scene, const double* d_x,
                                   const double* d_u, uint32_t B, double* d_pd, double* d_M, double* d_f, int* d_err) {
  if (B == 0) return RKH_OK;
  RKH_DISPATCH_N(n_dof, hipLaunchKernelGGL((state_derivative_kernel<N>), dim3(B), dim3(64), 0, s, d_scene, d_x, d_u, B,
                                           d_pd, d_M, d_f, d_err));
  RKH_HIP(hipGetLastError());
  return RKH_OK;
}

rkh_status launch_min_distance(hipStream_t s, int n_dof, int n_env, const SceneDev* d_scene, const void* d_pairs, int n_pairs,
                               const double* d_x, uint32_t B, double* d_dist) {
  if (B == 0) return RKH_OK;
  RKH_DISPATCH_N(n_dof, hipLaunchKernelGGL((min_distance_kernel<N>), dim3(B), dim3(64), smem_bytes<N>(n_env), s, d_scene,
                                           static_cast<const PairDev*>(d_pairs), n_pairs, d_x, B, d_dist));
  RKH_HIP(hipGetLastError());
  return RKH_OK;
}

}  // namespace rkh
