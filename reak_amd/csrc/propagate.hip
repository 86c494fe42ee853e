// propagate.hip -- RK4 forward-dynamics propagation of a KTE serial chain with a proximity
// (collision) test after every step, for batches of candidate edges (gfx950, wave64).
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
// Mapping onto the wave.  A block is one wave; a wave carries 64/GL candidate edges, GL lanes each
// (GL = 64: one wavefront per candidate, lowest latency; GL = 16: four candidates per wave, used when
// many planners oversubscribe the chip).  Within an edge's lane group:
//   * state x, RK4 temporaries, bounds: lane d < 2N owns component d (one register each)
//   * sin/cos of the N joint angles (half- and full-angle): lanes 0..2N-1 in parallel
//   * base->tip kinematic sweep and tip->base force sweep: serial by nature; every lane of the group runs
//     them on group-uniform values read by LDS broadcast (no cross-lane traffic), joints unrolled (template N)
//   * Jacobian columns Tcm(body b, coord c<=b): one lane per (b,c) pair; M(i,j): one lane per entry
//   * Cholesky: lane i owns row i (divisions of a column run in parallel), same per-element operation order
//   * proximity pairs: one lane per (robot shape, obstacle) pair, ballot for "any distance < 0"
// Chain parameters, obstacle table and all per-edge intermediates live in LDS.
// Compiled with -ffp-contract=off: products and sums round exactly as in the CPU reference; only
// sin/cos (OCML vs glibc) differ by ulps (stated tolerance: 1e-10 relative on propagated states).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <unordered_set>

#include "device_math.h"
#include "proximity_device.h"
#include "proximity_planar_device.h"
#include "rkh_internal.h"

namespace rkh {

struct __attribute__((aligned(16))) JointLds {  // chain parameters of one joint group, staged in LDS
  double axis[3], joint_inertia;
  double axis_n[3], mass;
  double off_pos[3], pad0;
  double off_quat[4];
  double off_R[9], pad1;
  double inertia[6];
};

template <int N>
struct __attribute__((aligned(16))) GroupWs {  // per-edge LDS workspace
  double x[2 * N];                  // state being differentiated (q, qd interleaved)
  double b[2 * N];                  // steer target
  double u[N];                      // held input
  double tmp[2 * N];                // lane-parallel -> sequential-sum staging
  double cs[N][4];                  // c2, s2 (half angle), c1, s1 (full angle)
  double Epos[N][3], Equat[N][4];   // joint end frames (jacobian parents)
  double Lpos[N][3], Lquat[N][4];   // link end frames (inertia frames)
  double FT[N][6];                  // inertia_3D d'Alembert force / torque (to be subtracted)
  double BFT[2][6];                 // flexible beam force / torque on its two anchor frames (zero without a beam)
  double Tcm[N][N][6];              // [body][coord] jacobian column (v, w)
  double Mf[N][N];                  // Tcm^T (Mcm Tcm) before symmetrisation
  double M[N][N];                   // symmetric M, overwritten by its Cholesky factor
  double Rpos[2 * N][3], Rquat[2 * N][4];  // robot shapes, global pose
  // two waves per edge (state_derivative_duo): x' as the solving wave leaves it for the other one, the number of joints
  // whose end frames the first wave has published, the first wave's "pivot below 1e-8" verdict
  double dpx[2 * N];
  double R2[N][9], RA[N][9];               // rotation matrices of the joint rotations (half-angle quaternion / axis-angle form)
  double Wj[N][3], ALj[N][3], ACCj[N][3];  // link end frames: angular velocity, angular acceleration, acceleration
  uint32_t duo_ready, duo_sing;
};

template <int N, int GL>
struct BlockLds {
  JointLds joints[N];
  double base[10];  // pos3, quat4, acc3
  double sink[64][4];  // per-lane dummy store target: keeps the group-leader stores branch-free
  GroupWs<N> g[64 / GL];
};

// the quasi-static kernels (edge walk, distance query) only need what the proximity test touches
template <int N>
struct __attribute__((aligned(16))) GroupWsQs {
  double x[2 * N];
  double tmp[2 * N];
  double cs[N][4];
  double Epos[N][3], Equat[N][4];
  double Rpos[2 * N][3], Rquat[2 * N][4];
};
template <int N, int GL>
struct BlockLdsQs {
  JointLds joints[N];
  double base[10];
  double sink[64][4];
  GroupWsQs<N> g[64 / GL];
};

RKH_DI d3 ld3(const double* p) { return d3{p[0], p[1], p[2]}; }
RKH_DI d4 ld4(const double* p) { return d4{p[0], p[1], p[2], p[3]}; }
RKH_DI void st3(double* p, d3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }
RKH_DI void st4(double* p, d4 v) { p[0] = v.w; p[1] = v.x; p[2] = v.y; p[3] = v.z; }
RKH_DI m33 ldm(const double* p) { return m33{p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], p[8]}; }

// Chain parameters are wave-uniform.  They are packed into R vector registers (lane l of register r
// holds flat parameter r*64 + l, JointLds layout) and read back with v_readlane: no memory latency on
// the serial sweeps' critical path and no long-lived scalar registers.
template <int N>
struct CPack {
  static constexpr int R = (N * 32 + 63) / 64;
  double v[R];
};
RKH_DI double readlane_f64(double v, int src_lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
  return __hiloint2double(hi, lo);
}
template <int N>
RKH_DI double cget(const CPack<N>& cp, int idx) { return readlane_f64(cp.v[idx >> 6], idx & 63); }
template <int N>
RKH_DI d3 cget3(const CPack<N>& cp, int idx) { return d3{cget(cp, idx), cget(cp, idx + 1), cget(cp, idx + 2)}; }
template <int N>
RKH_DI CPack<N> load_cpack(const JointLds* jl, int lane) {
  CPack<N> cp;
#pragma unroll
  for (int r = 0; r < CPack<N>::R; ++r) {
    const int idx = r * 64 + lane;
    cp.v[r] = idx < N * 32 ? reinterpret_cast<const double*>(jl)[idx] : 0.0;
  }
  return cp;
}
// field offsets inside JointLds (in doubles)
enum : int { JC_AXIS = 0, JC_JIN = 3, JC_AXISN = 4, JC_MASS = 7, JC_OFFP = 8, JC_OFFQ = 12, JC_OFFR = 16, JC_INER = 26 };

// axis_angle::getRotMat (rotations_3D.hpp:2160-2180) from cos/sin of the angle and the unit axis
RKH_DI m33 axis_angle_rotmat(double ca, double sa, d3 ax) {
  const double omc = 1.0 - ca;
  const double t11 = ca + omc * ax.x * ax.x, t22 = ca + omc * ax.y * ax.y, t33 = ca + omc * ax.z * ax.z;
  const double t12 = omc * ax.x * ax.y, t13 = omc * ax.x * ax.z, t23 = omc * ax.y * ax.z;
  const double t01 = sa * ax.x, t02 = sa * ax.y, t03 = sa * ax.z;
  return m33{t11, t12 - t03, t13 + t02, t12 + t03, t22, t23 - t01, t13 - t02, t23 + t01, t33};
}

template <int N>
__device__ __forceinline__ void stage_chain(const SceneDev* __restrict__ sc, JointLds* jl, double* base, int lane) {
  for (int i = lane; i < N * 32; i += 64) {
    const int j = i >> 5, k = i & 31;
    const JointDev& J = sc->joints[j];
    double v = 0.0;
    if (k < 3) v = J.axis[k];
    else if (k == 3) v = J.joint_inertia;
    else if (k < 7) v = J.axis_n[k - 4];
    else if (k == 7) v = J.mass;
    else if (k < 11) v = J.off_pos[k - 8];
    else if (k == 11) v = 0.0;
    else if (k < 16) v = J.off_quat[k - 12];
    else if (k < 25) v = J.off_R[k - 16];
    else if (k == 25) v = 0.0;
    else v = J.inertia[k - 26];
    reinterpret_cast<double*>(&jl[j])[k] = v;
  }
  if (lane < 3) base[lane] = sc->base_pos[lane];
  else if (lane < 7) base[lane] = sc->base_quat[lane - 3];
  else if (lane < 10) base[lane] = sc->base_acc[lane - 7];
}

__device__ __forceinline__ void stage_env(const SceneDev* __restrict__ sc, ShapeDev* env_lds, int lane) {
  const int n_words = sc->n_env * int(sizeof(ShapeDev) / sizeof(double));
  const double* src = reinterpret_cast<const double*>(sc->env);
  double* dst = reinterpret_cast<double*>(env_lds);
  for (int i = lane; i < n_words; i += 64) dst[i] = src[i];
}

// x' = f(x,u) for the lane group's edge.  ws.x / ws.u hold the state and the input (already staged).
// Returns dp for lane gl (< 2N); sets *singular if a Cholesky pivot is < 1e-8.
template <int N, int GL>
__device__ double state_derivative(const SceneDev* __restrict__ sc_beam, const CPack<N>& cp,
                                   const JointLds* __restrict__ jl, const double* __restrict__ base, GroupWs<N>& ws,
                                   double* __restrict__ sink, int gl, int gb, bool* singular,
                                   unsigned long long* stamps = nullptr) {
  constexpr int D = 2 * N;
  // diagnostic builds only (rkh_diag_feval_cycles): per-phase s_memtime deltas; null in the product path
#define RKH_STAMP(i)                                        \
  if (stamps) {                                             \
    const unsigned long long t_now = __builtin_readcyclecounter(); \
    stamps[i] += t_now - t_prev;                            \
    t_prev = t_now;                                         \
  }
  unsigned long long t_prev = stamps ? __builtin_readcyclecounter() : 0ull;
  const bool lead = (gl == 0);  // the group leader's stores land in ws, everyone else's in its private sink
  // ---- sin/cos, lane-parallel: lane 2j -> half angle, lane 2j+1 -> full angle
  if (gl < D) {
    const double q = ws.x[gl & ~1];
    double sn, cs;
    sincos((gl & 1) ? q : 0.5 * q, &sn, &cs);
    ws.cs[gl >> 1][(gl & 1) * 2 + 0] = cs;
    ws.cs[gl >> 1][(gl & 1) * 2 + 1] = sn;
  }
  __syncthreads();
  RKH_STAMP(0)

  // ---- base -> tip sweep (kte_map_chain::doMotion), group-uniform
  {
    d3 pos = ld3(base);
    d4 Q = ld4(base + 3);
    d3 w = mk3(0, 0, 0), alpha = mk3(0, 0, 0);
    d3 acc = ld3(base + 7);
#pragma unroll
    for (int j = 0; j < N; ++j) {
      const int jb = j * 32;
      if (sc_beam->branch_start[j]) {  // a new branch: base frame * mount pose (rigid_link_3D::doMotion from frame 0)
        const d4 bq = ld4(base + 3);
        pos = ld3(base) + mul(rotmat(bq), ld3(sc_beam->mount_pos[j]));
        Q = qmul(bq, ld4(sc_beam->mount_quat[j]));
        w = mk3(0, 0, 0);
        alpha = mk3(0, 0, 0);
        acc = ld3(base + 7);
      }
      const d3 axis = cget3(cp, jb + JC_AXIS);
      const d3 axis_n = cget3(cp, jb + JC_AXISN);
      const double c2 = ws.cs[j][0], s2 = ws.cs[j][1];
      const double qd = ws.x[2 * j + 1];
      // revolute_joint_3D::doMotion (revolute_joint.cpp:121-148)
      const d4 tq = d4{c2, axis_n.x * s2, axis_n.y * s2, axis_n.z * s2};
      const m33 R2 = rotmat(tq);
      const d4 EQ = qmul(Q, tq);
      const d3 wb = mulT(w, R2);
      const d3 qa = qd * axis;
      const d3 Ew = wb + qa;
      const d3 Ealpha = mulT(alpha, R2) + cross(wb, qa);
      st3(lead ? ws.Epos[j] : sink, pos);
      st4(lead ? ws.Equat[j] : sink, EQ);
      // rigid_link_3D::doMotion = frame * pose (frame_3D.hpp:240-255)
      const d3 op = cget3(cp, jb + JC_OFFP);
      const m33 R = rotmat(EQ);
      pos = pos + mul(R, op);
      acc = acc + mul(R, cross(Ew, cross(Ew, op)) + cross(Ealpha, op));
      const m33 Ro = m33{cget(cp, jb + JC_OFFR + 0), cget(cp, jb + JC_OFFR + 1), cget(cp, jb + JC_OFFR + 2),
                         cget(cp, jb + JC_OFFR + 3), cget(cp, jb + JC_OFFR + 4), cget(cp, jb + JC_OFFR + 5),
                         cget(cp, jb + JC_OFFR + 6), cget(cp, jb + JC_OFFR + 7), cget(cp, jb + JC_OFFR + 8)};
      Q = qmul(EQ, d4{cget(cp, jb + JC_OFFQ), cget(cp, jb + JC_OFFQ + 1), cget(cp, jb + JC_OFFQ + 2),
                      cget(cp, jb + JC_OFFQ + 3)});
      alpha = mulT(Ealpha, Ro);
      w = mulT(Ew, Ro);
      // inertia_3D::doForce terms (inertia.cpp:111-122), applied in the backward sweep
      const double inertia[6] = {cget(cp, jb + JC_INER), cget(cp, jb + JC_INER + 1), cget(cp, jb + JC_INER + 2),
                                 cget(cp, jb + JC_INER + 3), cget(cp, jb + JC_INER + 4), cget(cp, jb + JC_INER + 5)};
      const d3 Fi = cget(cp, jb + JC_MASS) * qrot(qinv(Q), acc);
      const d3 Ti = sym_mul(inertia, alpha) + cross(w, sym_mul(inertia, w));
      st3(lead ? ws.Lpos[j] : sink, pos);
      st4(lead ? ws.Lquat[j] : sink, Q);
      st3(lead ? ws.FT[j] : sink, Fi);
      st3(lead ? ws.FT[j] + 3 : sink, Ti);
    }
  }
  __syncthreads();
  {
    // flexible_beam_3D::doForce (listed last in the chain, so it runs first in the reverse pass and its force is the
    // first term of its anchor frames' accumulators); anchors are link end frames (or a world anchor for anchor 2)
    d3 BF1 = mk3(0, 0, 0), BT1 = mk3(0, 0, 0), BF2 = mk3(0, 0, 0), BT2 = mk3(0, 0, 0);
    if (sc_beam->beam_on) {
      const int j1 = sc_beam->beam_j1, j2 = sc_beam->beam_j2;
      const d3 p1 = ld3(ws.Lpos[j1]);
      const d4 q1 = ld4(ws.Lquat[j1]);
      const d3 p2 = j2 >= 0 ? ld3(ws.Lpos[j2]) : ld3(sc_beam->beam_pos);
      const d4 q2 = j2 >= 0 ? ld4(ws.Lquat[j2]) : ld4(sc_beam->beam_quat);
      beam_force(p1, q1, p2, q2, sc_beam->beam_rest, sc_beam->beam_k, sc_beam->beam_kt, &BF1, &BT1);
      if (j2 >= 0) beam_force_anchor2(p1, q1, p2, q2, sc_beam->beam_rest, sc_beam->beam_k, sc_beam->beam_kt, &BF2, &BT2);
    }
    __syncthreads();
    st3(lead ? ws.BFT[0] : sink, BF1);
    st3(lead ? ws.BFT[0] + 3 : sink, BT1);
    st3(lead ? ws.BFT[1] : sink, BF2);
    st3(lead ? ws.BFT[1] + 3 : sink, BT2);
  }
  __syncthreads();
  RKH_STAMP(1)

  // ---- jacobian columns, one lane per (body b, coord c <= b): get_jac_relative_to
  //      (motion_jacobians.hpp:238-251) with f2 = (~F_c) * F_b (frame_3D.hpp:184-189,222-238,368-382)
  for (int p = gl; p < N * (N + 1) / 2; p += GL) {
    int b = 0, c = p;  // unrank p -> (b, c), c <= b
    while (c > b) {
      c -= b + 1;
      ++b;
    }
    if (c < sc_beam->branch_first[b]) continue;  // joint c is not upstream of body b (another branch)
    const d3 cp = ld3(ws.Epos[c]);
    const d4 cq = ld4(ws.Equat[c]);
    const d3 bp = ld3(ws.Lpos[b]);
    const d4 bq = ld4(ws.Lquat[b]);
    const m33 R = rotmat(cq);
    const d4 iq = qinv(cq);
    const d3 ipos = mulT(-cp, R);
    const m33 Ri = rotmat(iq);
    const d3 f2pos = ipos + mul(Ri, bp);
    const d4 f2q = qmul(iq, bq);
    const m33 Rf = rotmat(f2q);
    const d3 axis = ld3(jl[c].axis);
    const d3 wt = mulT(axis, Rf);
    const d3 vt = mulT(cross(axis, f2pos), Rf);
    st3(ws.Tcm[b][c], vt);
    st3(ws.Tcm[b][c] + 3, wt);
  }

  RKH_STAMP(2)
  // ---- tip -> base sweep (kte_map_chain::doForce in reverse op order), group-uniform
  double f_mine = 0.0;  // lane i < N keeps generalized force i
  {
    d3 LF = mk3(0, 0, 0), LT = mk3(0, 0, 0);
    const int bj1 = sc_beam->beam_on ? sc_beam->beam_j1 : -1, bj2 = sc_beam->beam_on ? sc_beam->beam_j2 : -1;
#pragma unroll
    for (int j = N - 1; j >= 0; --j) {
      const int jb = j * 32;
      const d3 axis = cget3(cp, jb + JC_AXIS);
      if (j + 1 < N && sc_beam->branch_start[j + 1]) {  // the joint above started another branch: this link is a tip
        LF = mk3(0, 0, 0);
        LT = mk3(0, 0, 0);
      }
      // the beam acted first on its anchor frames (reverse op order): (0 + beam) + what the child joint passes down
      if (j == bj1) {
        LF = LF + ld3(ws.BFT[0]);
        LT = LT + ld3(ws.BFT[0] + 3);
      }
      if (j == bj2) {
        LF = LF + ld3(ws.BFT[1]);
        LT = LT + ld3(ws.BFT[1] + 3);
      }
      // inertia_3D::doForce on the link end frame
      LF = LF - ld3(ws.FT[j]);
      LT = LT - ld3(ws.FT[j] + 3);
      // rigid_link_3D::doForce (rigid_link.cpp:170-178)
      const m33 Ro = m33{cget(cp, jb + JC_OFFR + 0), cget(cp, jb + JC_OFFR + 1), cget(cp, jb + JC_OFFR + 2),
                         cget(cp, jb + JC_OFFR + 3), cget(cp, jb + JC_OFFR + 4), cget(cp, jb + JC_OFFR + 5),
                         cget(cp, jb + JC_OFFR + 6), cget(cp, jb + JC_OFFR + 7), cget(cp, jb + JC_OFFR + 8)};
      const d3 op = cget3(cp, jb + JC_OFFP);
      const d3 tmp_force = mul(Ro, LF);
      const d3 ET = mul(Ro, LT) + cross(op, tmp_force);
      // revolute_joint_3D::doForce (revolute_joint.cpp:170-181)
      const m33 Ra = axis_angle_rotmat(ws.cs[j][2], ws.cs[j][3], cget3(cp, jb + JC_AXISN));
      const double ta = dot(ET, axis);
      LF = mul(Ra, tmp_force);
      LT = mul(Ra, ET - ta * axis);
      // inertia_gen::doForce: f -= q_ddot * mass with q_ddot = 0 ; driving_actuator_gen::doForce
      const double uj = ws.u[j];
      const double fj = ta + uj;
      LT = LT - uj * axis;
      f_mine = (gl == j) ? fj : f_mine;
    }
  }
  if (gl < N) ws.tmp[gl] = f_mine;  // bias force, kept for the kernel-level parity export
  __syncthreads();
  RKH_STAMP(3)

  // ---- Mf = Tcm^T (Mcm Tcm), one lane per entry (i,j), summation order of
  //      mat_alg_symmetric.hpp:551-566 (Mcm*Tcm) and mat_operators.hpp:104-114 (dense product)
  // (the loop is uniform -- lanes without an entry in the last pass recompute the last entry and do not store -- and the
  // v_readlane reads of the chain parameters sit outside the per-lane condition: a readlane must not pick a lane that
  // was masked off when a run-time-indexed copy of the parameter registers was made)
  for (int e0 = 0; e0 < N * N; e0 += GL) {
    const bool e_valid = e0 + gl < N * N;
    const int e = e_valid ? e0 + gl : N * N - 1;
    const int i = e / N, jx = e % N;
    double s = 0.0;
    if (i == jx) s = s + jl[i].joint_inertia;  // inertia_gen rows: Tcm = 1, Mcm = rotor inertia
#pragma unroll
    for (int b = 0; b < N; ++b) {
      const int bb = b * 32;
      const double mass = cget(cp, bb + JC_MASS);
      const double inertia[6] = {cget(cp, bb + JC_INER), cget(cp, bb + JC_INER + 1), cget(cp, bb + JC_INER + 2),
                                 cget(cp, bb + JC_INER + 3), cget(cp, bb + JC_INER + 4), cget(cp, bb + JC_INER + 5)};
      const int first_b = sc_beam->branch_first[b];
      if (b >= i && b >= jx && i >= first_b && jx >= first_b) {
        const double* Ti = ws.Tcm[b][i];
        const double* Tj = ws.Tcm[b][jx];
        s = s + Ti[0] * (mass * Tj[0]);
        s = s + Ti[1] * (mass * Tj[1]);
        s = s + Ti[2] * (mass * Tj[2]);
        const d3 P = sym_mul(inertia, mk3(Tj[3], Tj[4], Tj[5]));
        s = s + Ti[3] * P.x;
        s = s + Ti[4] * P.y;
        s = s + Ti[5] * P.z;
      }
    }
    if (e_valid) ws.Mf[i][jx] = s;
  }
  __syncthreads();
  // mat<symmetric>(general): 0.5 * (M(j,i) + M(i,j)), j < i  (mat_alg_symmetric.hpp:183-187)
  for (int e = gl; e < N * N; e += GL) {
    const int i = e / N, jx = e % N;
    const int lo = i < jx ? i : jx, hi = i < jx ? jx : i;
    ws.M[i][jx] = (i == jx) ? ws.Mf[i][i] : 0.5 * (ws.Mf[lo][hi] + ws.Mf[hi][lo]);
  }
  __syncthreads();

  RKH_STAMP(4)
  // ---- linsolve_Cholesky (mat_cholesky.hpp:546-554): lane i owns row i; every L(i,j) is formed by the
  //      reference's operation sequence (A(i,j), minus L(i,k) L(j,k) for k ascending, divided by L(j,j))
  const int row = gl < N ? gl : N - 1;
  double Lrow[N];
#pragma unroll
  for (int k = 0; k < N; ++k) Lrow[k] = 0.0;
  bool sing = false;
#pragma unroll
  for (int j = 0; j < N; ++j) {
    // pivot of column j, computed by every lane from row j (already final in LDS for k < j)
    double dgl = ws.M[j][j];
#pragma unroll
    for (int k = 0; k < j; ++k) {
      const double ljk = ws.M[j][k];
      dgl = dgl - ljk * ljk;
    }
    if (dgl < 1e-8) sing = true;
    const double ljj = sqrt(dgl);
    double v = ws.M[row][j];
#pragma unroll
    for (int k = 0; k < j; ++k) v = v - Lrow[k] * ws.M[j][k];
    v = v / ljj;
    Lrow[j] = (row == j) ? ljj : v;
    if (gl < N && row >= j) ws.M[row][j] = Lrow[j];
    __syncthreads();
  }
  // backsub_Cholesky_impl (mat_cholesky.hpp:160-178): L y = f, then L^T x = y
  double diag = 1.0;
#pragma unroll
  for (int k = 0; k < N; ++k) diag = (row == k) ? Lrow[k] : diag;
  double accv = f_mine;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    const double yk = __shfl(accv / diag, gb + k, 64);
    if (row == k) accv = yk;
    else if (row > k) accv = accv - Lrow[k] * yk;
  }
#pragma unroll
  for (int k = N - 1; k >= 0; --k) {
    const double xk = __shfl(accv / diag, gb + k, 64);
    if (row == k) accv = xk;
    else if (row < k) accv = accv - ws.M[k][row] * xk;
  }
  if (sing) *singular = true;

  // pd[2j] = q_dot_j ; pd[2j+1] = qdd_j  (kte_nl_system.hpp:276-279)
  const double qdd = __shfl(accv, gb + (gl >> 1), 64);
  double out = 0.0;
  if (gl < D) out = (gl & 1) ? qdd : ws.x[gl + 1];
  RKH_STAMP(5)
#undef RKH_STAMP
  return out;
}

// x' = f(x,u) by TWO waves (the latency form of the latency mapping: rounds so small that half the chip's SIMDs would
// idle even at one wave per edge -- a single problem, a graph planner's step).  One f-eval of state_derivative is ~6.6 k
// fp64 instructions in one wave's stream, 26.5 k cycles, and its phases are not all dependent:
//   wave 0: joint / link end frames (the pose half of the base -> tip sweep) -> Jacobian columns -> Mf -> M -> Cholesky factor
//   wave 1: velocities, accelerations, d'Alembert terms (the other half of the sweep, one joint behind wave 0: it waits on
//           a counter in LDS, never on a barrier) -> beam -> tip -> base force sweep -> generalized forces
// then wave 0 solves and leaves x' in LDS for both.  Every value is formed by the same operation sequence as in
// state_derivative (the two halves of the sweep only share the rotation matrix of a joint's end frame, which each
// forms from the same quaternion), so the result is the same bits.  Critical path ~15 k cycles.
RKH_DI void wave_sync() {  // LDS traffic of this wave's lanes, ordered (no block barrier: the other wave is elsewhere)
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
}
template <int N>
__device__ double state_derivative_duo(const SceneDev* __restrict__ sc_beam, const CPack<N>& cp,
                                       const JointLds* __restrict__ jl, const double* __restrict__ base, GroupWs<N>& ws,
                                       double* __restrict__ sink, int gl, int wave, bool* singular,
                                       unsigned long long* stamps = nullptr) {
  constexpr int D = 2 * N, GL = 64;
  const bool lead = (gl == 0);
  // diagnostic builds only: cycles per phase of this wave (wave 0: sincos, frames, columns, Mf + M, factor, wait, solve;
  // wave 1: sincos, velocity half, beam, force sweep, wait)
#define RKH_STAMP(i)                                                \
  if (stamps) {                                                     \
    const unsigned long long t_now = __builtin_readcyclecounter(); \
    stamps[i] += t_now - t_prev;                                    \
    t_prev = t_now;                                                 \
  }
  unsigned long long t_prev = stamps ? __builtin_readcyclecounter() : 0ull;
  uint32_t branches = 0u;  // bit j: joint j starts a new branch at the chain base (read once, ahead of the serial loops)
#pragma unroll
  for (int j = 0; j < N; ++j) branches |= sc_beam->branch_start[j] ? (1u << j) : 0u;
  if (gl < D) {  // sin / cos (both waves: the same values)
    const double q = ws.x[gl & ~1];
    double sn, cs;
    sincos((gl & 1) ? q : 0.5 * q, &sn, &cs);
    ws.cs[gl >> 1][(gl & 1) * 2 + 0] = cs;
    ws.cs[gl >> 1][(gl & 1) * 2 + 1] = sn;
  }
  if (wave == 0 && lead) {
    ws.duo_ready = 0u;
    ws.duo_sing = 0u;
  }
  __syncthreads();
  RKH_STAMP(0)
  double Lrow[N];
#pragma unroll
  for (int k = 0; k < N; ++k) Lrow[k] = 0.0;
  const int row = gl < N ? gl : N - 1;
  bool sing = false;
  if (wave == 0) {
    {  // ---- pose half of the base -> tip sweep; every joint's end frames are published as soon as they are stored
      d3 pos = ld3(base);
      d4 Q = ld4(base + 3);
#pragma unroll
      for (int j = 0; j < N; ++j) {
        const int jb = j * 32;
        if ((branches >> j) & 1u) {
          const d4 bq = ld4(base + 3);
          pos = ld3(base) + mul(rotmat(bq), ld3(sc_beam->mount_pos[j]));
          Q = qmul(bq, ld4(sc_beam->mount_quat[j]));
        }
        const d3 axis_n = cget3(cp, jb + JC_AXISN);
        const double c2 = ws.cs[j][0], s2 = ws.cs[j][1];
        const d4 tq = d4{c2, axis_n.x * s2, axis_n.y * s2, axis_n.z * s2};
        const d4 EQ = qmul(Q, tq);
        st3(lead ? ws.Epos[j] : sink, pos);
        st4(lead ? ws.Equat[j] : sink, EQ);
        const d3 op = cget3(cp, jb + JC_OFFP);
        const m33 R = rotmat(EQ);
        pos = pos + mul(R, op);
        Q = qmul(EQ, d4{cget(cp, jb + JC_OFFQ), cget(cp, jb + JC_OFFQ + 1), cget(cp, jb + JC_OFFQ + 2),
                        cget(cp, jb + JC_OFFQ + 3)});
        st3(lead ? ws.Lpos[j] : sink, pos);
        st4(lead ? ws.Lquat[j] : sink, Q);
        wave_sync();
        if (lead) __hip_atomic_store(&ws.duo_ready, uint32_t(j + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
    RKH_STAMP(1)
    // ---- jacobian columns (as in state_derivative)
    for (int p = gl; p < N * (N + 1) / 2; p += GL) {
      int b = 0, c = p;
      while (c > b) {
        c -= b + 1;
        ++b;
      }
      if (c < sc_beam->branch_first[b]) continue;
      const d3 cpos = ld3(ws.Epos[c]);
      const d4 cq = ld4(ws.Equat[c]);
      const d3 bp = ld3(ws.Lpos[b]);
      const d4 bq = ld4(ws.Lquat[b]);
      const m33 R = rotmat(cq);
      const d4 iq = qinv(cq);
      const d3 ipos = mulT(-cpos, R);
      const m33 Ri = rotmat(iq);
      const d3 f2pos = ipos + mul(Ri, bp);
      const d4 f2q = qmul(iq, bq);
      const m33 Rf = rotmat(f2q);
      const d3 axis = ld3(jl[c].axis);
      const d3 wt = mulT(axis, Rf);
      const d3 vt = mulT(cross(axis, f2pos), Rf);
      st3(ws.Tcm[b][c], vt);
      st3(ws.Tcm[b][c] + 3, wt);
    }
    wave_sync();
    RKH_STAMP(2)
    // ---- Mf = Tcm^T (Mcm Tcm), M, Cholesky factor (as in state_derivative; wave-local ordering instead of barriers)
    for (int e0 = 0; e0 < N * N; e0 += GL) {
      const bool e_valid = e0 + gl < N * N;
      const int e = e_valid ? e0 + gl : N * N - 1;
      const int i = e / N, jx = e % N;
      double sacc = 0.0;
      if (i == jx) sacc = sacc + jl[i].joint_inertia;
#pragma unroll
      for (int b = 0; b < N; ++b) {
        const int bb = b * 32;
        const double mass = cget(cp, bb + JC_MASS);
        const double inertia[6] = {cget(cp, bb + JC_INER), cget(cp, bb + JC_INER + 1), cget(cp, bb + JC_INER + 2),
                                   cget(cp, bb + JC_INER + 3), cget(cp, bb + JC_INER + 4), cget(cp, bb + JC_INER + 5)};
        const int first_b = sc_beam->branch_first[b];
        if (b >= i && b >= jx && i >= first_b && jx >= first_b) {
          const double* Ti = ws.Tcm[b][i];
          const double* Tj = ws.Tcm[b][jx];
          sacc = sacc + Ti[0] * (mass * Tj[0]);
          sacc = sacc + Ti[1] * (mass * Tj[1]);
          sacc = sacc + Ti[2] * (mass * Tj[2]);
          const d3 P = sym_mul(inertia, mk3(Tj[3], Tj[4], Tj[5]));
          sacc = sacc + Ti[3] * P.x;
          sacc = sacc + Ti[4] * P.y;
          sacc = sacc + Ti[5] * P.z;
        }
      }
      if (e_valid) ws.Mf[i][jx] = sacc;
    }
    wave_sync();
    for (int e = gl; e < N * N; e += GL) {
      const int i = e / N, jx = e % N;
      const int lo = i < jx ? i : jx, hi = i < jx ? jx : i;
      ws.M[i][jx] = (i == jx) ? ws.Mf[i][i] : 0.5 * (ws.Mf[lo][hi] + ws.Mf[hi][lo]);
    }
    wave_sync();
    RKH_STAMP(3)
#pragma unroll
    for (int j = 0; j < N; ++j) {
      double dgl = ws.M[j][j];
#pragma unroll
      for (int k = 0; k < j; ++k) {
        const double ljk = ws.M[j][k];
        dgl = dgl - ljk * ljk;
      }
      if (dgl < 1e-8) sing = true;
      const double ljj = sqrt(dgl);
      double v = ws.M[row][j];
#pragma unroll
      for (int k = 0; k < j; ++k) v = v - Lrow[k] * ws.M[j][k];
      v = v / ljj;
      Lrow[j] = (row == j) ? ljj : v;
      if (gl < N && row >= j) ws.M[row][j] = Lrow[j];
      wave_sync();
    }
    RKH_STAMP(4)
  } else {
    // ---- velocity / acceleration half of the base -> tip sweep.  Only the recurrences w, alpha, acc are serial: the
    // joint rotation matrices before them and the d'Alembert terms after them take one LANE per joint.
    if (gl < N) {
      const d3 axis_n = ld3(jl[gl].axis_n);
      const double c2 = ws.cs[gl][0], s2 = ws.cs[gl][1];
      const m33 R2 = rotmat(d4{c2, axis_n.x * s2, axis_n.y * s2, axis_n.z * s2});
      double* r2 = ws.R2[gl];
      r2[0] = R2.a11; r2[1] = R2.a12; r2[2] = R2.a13; r2[3] = R2.a21; r2[4] = R2.a22; r2[5] = R2.a23;
      r2[6] = R2.a31; r2[7] = R2.a32; r2[8] = R2.a33;
      const m33 Ra = axis_angle_rotmat(ws.cs[gl][2], ws.cs[gl][3], axis_n);  // revolute_joint_3D::doForce's rotation
      double* ra = ws.RA[gl];
      ra[0] = Ra.a11; ra[1] = Ra.a12; ra[2] = Ra.a13; ra[3] = Ra.a21; ra[4] = Ra.a22; ra[5] = Ra.a23;
      ra[6] = Ra.a31; ra[7] = Ra.a32; ra[8] = Ra.a33;
    }
    wave_sync();
    {
      d3 w = mk3(0, 0, 0), alpha = mk3(0, 0, 0);
      d3 acc = ld3(base + 7);
#pragma unroll
      for (int j = 0; j < N; ++j) {
        const int jb = j * 32;
        // joint j's end frame from wave 0 (a counter in LDS: wave 0 does not stop for this)
        while (__hip_atomic_load(&ws.duo_ready, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < uint32_t(j + 1))
          __builtin_amdgcn_s_sleep(1);
        if ((branches >> j) & 1u) {
          w = mk3(0, 0, 0);
          alpha = mk3(0, 0, 0);
          acc = ld3(base + 7);
        }
        const d3 axis = cget3(cp, jb + JC_AXIS);
        const double qd = ws.x[2 * j + 1];
        const m33 R2 = ldm(ws.R2[j]);
        const d3 wb = mulT(w, R2);
        const d3 qa = qd * axis;
        const d3 Ew = wb + qa;
        const d3 Ealpha = mulT(alpha, R2) + cross(wb, qa);
        const d3 op = cget3(cp, jb + JC_OFFP);
        const m33 R = rotmat(ld4(ws.Equat[j]));
        acc = acc + mul(R, cross(Ew, cross(Ew, op)) + cross(Ealpha, op));
        const m33 Ro = m33{cget(cp, jb + JC_OFFR + 0), cget(cp, jb + JC_OFFR + 1), cget(cp, jb + JC_OFFR + 2),
                           cget(cp, jb + JC_OFFR + 3), cget(cp, jb + JC_OFFR + 4), cget(cp, jb + JC_OFFR + 5),
                           cget(cp, jb + JC_OFFR + 6), cget(cp, jb + JC_OFFR + 7), cget(cp, jb + JC_OFFR + 8)};
        alpha = mulT(Ealpha, Ro);
        w = mulT(Ew, Ro);
        st3(lead ? ws.Wj[j] : sink, w);
        st3(lead ? ws.ALj[j] : sink, alpha);
        st3(lead ? ws.ACCj[j] : sink, acc);
      }
    }
    wave_sync();
    if (gl < N) {  // inertia_3D::doForce terms (inertia.cpp:111-122) of joint gl's link, applied in the backward sweep
      const d4 Q = ld4(ws.Lquat[gl]);
      const d3 acc = ld3(ws.ACCj[gl]), alpha = ld3(ws.ALj[gl]), w = ld3(ws.Wj[gl]);
      const double* inertia = jl[gl].inertia;
      const d3 Fi = jl[gl].mass * qrot(qinv(Q), acc);
      const d3 Ti = sym_mul(inertia, alpha) + cross(w, sym_mul(inertia, w));
      st3(ws.FT[gl], Fi);
      st3(ws.FT[gl] + 3, Ti);
    }
    wave_sync();
    RKH_STAMP(1)
    {  // flexible_beam_3D::doForce (as in state_derivative)
      d3 BF1 = mk3(0, 0, 0), BT1 = mk3(0, 0, 0), BF2 = mk3(0, 0, 0), BT2 = mk3(0, 0, 0);
      if (sc_beam->beam_on) {
        const int j1 = sc_beam->beam_j1, j2 = sc_beam->beam_j2;
        const d3 p1 = ld3(ws.Lpos[j1]);
        const d4 q1 = ld4(ws.Lquat[j1]);
        const d3 p2 = j2 >= 0 ? ld3(ws.Lpos[j2]) : ld3(sc_beam->beam_pos);
        const d4 q2 = j2 >= 0 ? ld4(ws.Lquat[j2]) : ld4(sc_beam->beam_quat);
        beam_force(p1, q1, p2, q2, sc_beam->beam_rest, sc_beam->beam_k, sc_beam->beam_kt, &BF1, &BT1);
        if (j2 >= 0) beam_force_anchor2(p1, q1, p2, q2, sc_beam->beam_rest, sc_beam->beam_k, sc_beam->beam_kt, &BF2, &BT2);
      }
      st3(lead ? ws.BFT[0] : sink, BF1);
      st3(lead ? ws.BFT[0] + 3 : sink, BT1);
      st3(lead ? ws.BFT[1] : sink, BF2);
      st3(lead ? ws.BFT[1] + 3 : sink, BT2);
    }
    wave_sync();
    RKH_STAMP(2)
    double f_mine = 0.0;
    {  // ---- tip -> base sweep (as in state_derivative)
      d3 LF = mk3(0, 0, 0), LT = mk3(0, 0, 0);
      const int bj1 = sc_beam->beam_on ? sc_beam->beam_j1 : -1, bj2 = sc_beam->beam_on ? sc_beam->beam_j2 : -1;
#pragma unroll
      for (int j = N - 1; j >= 0; --j) {
        const int jb = j * 32;
        const d3 axis = cget3(cp, jb + JC_AXIS);
        if (j + 1 < N && ((branches >> (j + 1 < N ? j + 1 : 0)) & 1u)) {
          LF = mk3(0, 0, 0);
          LT = mk3(0, 0, 0);
        }
        if (j == bj1) {
          LF = LF + ld3(ws.BFT[0]);
          LT = LT + ld3(ws.BFT[0] + 3);
        }
        if (j == bj2) {
          LF = LF + ld3(ws.BFT[1]);
          LT = LT + ld3(ws.BFT[1] + 3);
        }
        LF = LF - ld3(ws.FT[j]);
        LT = LT - ld3(ws.FT[j] + 3);
        const m33 Ro = m33{cget(cp, jb + JC_OFFR + 0), cget(cp, jb + JC_OFFR + 1), cget(cp, jb + JC_OFFR + 2),
                           cget(cp, jb + JC_OFFR + 3), cget(cp, jb + JC_OFFR + 4), cget(cp, jb + JC_OFFR + 5),
                           cget(cp, jb + JC_OFFR + 6), cget(cp, jb + JC_OFFR + 7), cget(cp, jb + JC_OFFR + 8)};
        const d3 op = cget3(cp, jb + JC_OFFP);
        const d3 tmp_force = mul(Ro, LF);
        const d3 ET = mul(Ro, LT) + cross(op, tmp_force);
        const m33 Ra = ldm(ws.RA[j]);
        const double ta = dot(ET, axis);
        LF = mul(Ra, tmp_force);
        LT = mul(Ra, ET - ta * axis);
        const double uj = ws.u[j];
        const double fj = ta + uj;
        LT = LT - uj * axis;
        f_mine = (gl == j) ? fj : f_mine;
      }
    }
    if (gl < N) ws.tmp[gl] = f_mine;
    RKH_STAMP(3)
  }
  __syncthreads();
  RKH_STAMP(5)
  if (wave == 0) {  // backsub_Cholesky_impl (mat_cholesky.hpp:160-178): L y = f, then L^T x = y
    double diag = 1.0;
#pragma unroll
    for (int k = 0; k < N; ++k) diag = (row == k) ? Lrow[k] : diag;
    double accv = (gl < N) ? ws.tmp[gl] : 0.0;
#pragma unroll
    for (int k = 0; k < N; ++k) {
      const double yk = __shfl(accv / diag, k, 64);
      if (row == k) accv = yk;
      else if (row > k) accv = accv - Lrow[k] * yk;
    }
#pragma unroll
    for (int k = N - 1; k >= 0; --k) {
      const double xk = __shfl(accv / diag, k, 64);
      if (row == k) accv = xk;
      else if (row < k) accv = accv - ws.M[k][row] * xk;
    }
    const double qdd = __shfl(accv, gl >> 1, 64);
    if (gl < D) ws.dpx[gl] = (gl & 1) ? qdd : ws.x[gl + 1];
    if (sing && lead) ws.duo_sing = 1u;
  }
  __syncthreads();
  RKH_STAMP(6)
#undef RKH_STAMP
  if (ws.duo_sing) *singular = true;
  return gl < D ? ws.dpx[gl] : 0.0;
}

// Planar scenes (SceneDev::planar): revolute_joint_2D / rigid_link_2D kinematics (revolute_joint.cpp:30-47,
// rigid_link.cpp:87-99), pose_2D::getGlobalPose of the robot shapes, then proxy_query_pair_2D::findMinimumDistance
// (proxy_query_model.cpp:163-189) replayed in finder order: every lane computes one pair's cull value and distance,
// and the sequence "skip if cull > running minimum, else take the minimum" is resolved lane by lane.
template <int N, int GL, typename WS>
__device__ double proximity_min_planar(const SceneDev* __restrict__ sc, const CPack<N>& cp, const double* __restrict__ base,
                                       const ShapeDev* __restrict__ env_lds, const PairDev* __restrict__ pairs, int n_pairs,
                                       WS& ws, int gl, int gb) {
  for (int jj = gl; jj < N; jj += GL) {
    double sn, cs;
    sincos(ws.x[2 * jj], &sn, &cs);
    ws.cs[jj][0] = cs;
    ws.cs[jj][1] = sn;
  }
  __syncthreads();
  {
    d2 pos = mk2(base[0], base[1]);
    d2 R = mk2(base[3], base[4]);
#pragma unroll
    for (int j = 0; j < N; ++j) {
      const int jb = j * 32;
      const d2 ER = rmul(R, mk2(ws.cs[j][0], ws.cs[j][1]));
      if (gl == 0) {
        ws.Epos[j][0] = pos.x;
        ws.Epos[j][1] = pos.y;
        ws.Equat[j][0] = ER.x;
        ws.Equat[j][1] = ER.y;
      }
      pos = pos + rrot(ER, mk2(cget(cp, jb + JC_OFFP), cget(cp, jb + JC_OFFP + 1)));
      R = rmul(ER, mk2(cget(cp, jb + JC_OFFQ), cget(cp, jb + JC_OFFQ + 1)));
    }
  }
  __syncthreads();
  for (int r = gl; r < sc->n_robot; r += GL) {
    const ShapeDev& sh = sc->robot[r];
    const int j = sh.link;
    const d2 pp = mk2(ws.Epos[j][0], ws.Epos[j][1]);
    const d2 pr = mk2(ws.Equat[j][0], ws.Equat[j][1]);
    const d2 gp = pp + rrot(pr, mk2(sh.pos[0], sh.pos[1]));
    const d2 gr = rmul(pr, mk2(sh.quat[0], sh.quat[1]));
    ws.Rpos[r][0] = gp.x;
    ws.Rpos[r][1] = gp.y;
    ws.Rquat[r][0] = gr.x;
    ws.Rquat[r][1] = gr.y;
  }
  __syncthreads();
  double min_dist = INFINITY;
  for (int p0 = 0; p0 < n_pairs; p0 += GL) {
    const int p = p0 + gl;
    double d = INFINITY, c = INFINITY;
    if (p < n_pairs) {
      const PairDev pr = pairs[p];
      const ShapeDev& rs = sc->robot[pr.robot];
      const ShapeDev& es = env_lds[pr.env];
      ShapeP A, Bv;
      A.kind = rs.kind;
      A.pos = mk2(ws.Rpos[pr.robot][0], ws.Rpos[pr.robot][1]);
      A.rot = mk2(ws.Rquat[pr.robot][0], ws.Rquat[pr.robot][1]);
      A.d0 = rs.dims[0]; A.d1 = rs.dims[1];
      Bv.kind = es.kind;
      Bv.pos = mk2(es.pos[0], es.pos[1]);
      Bv.rot = mk2(es.quat[0], es.quat[1]);
      Bv.d0 = es.dims[0]; Bv.d1 = es.dims[1];
      const ShapeP& s1 = pr.s1_is_robot ? A : Bv;
      const ShapeP& s2 = pr.s1_is_robot ? Bv : A;
      const double r1 = pr.s1_is_robot ? rs.brad : es.brad;
      const double r2 = pr.s1_is_robot ? es.brad : rs.brad;
      c = norm_2(to_parent(s2, mk2(0.0, 0.0)) - to_parent(s1, mk2(0.0, 0.0))) - r1 - r2;
      d = pair_distance_planar(pr.routine, s1, s2);
    }
    const int cnt = (n_pairs - p0 < GL) ? (n_pairs - p0) : GL;
    for (int l = 0; l < cnt; ++l) {
      const double cl = __shfl(c, gb + l, 64), dl = __shfl(d, gb + l, 64);
      if (p0 + l == 0) min_dist = dl;                       // the first finder is always computed
      else if (!(cl > min_dist) && dl < min_dist) min_dist = dl;
    }
  }
  return min_dist;
}

// Proximity of the configuration in ws.x (joint angles): minimum distance over computed pairs.
// With cull_positive, pairs whose bounding spheres are apart are skipped (they cannot make the verdict
// "colliding"; proxy_query_model.cpp:386-389 culls the same way against the running minimum) and the scan
// stops once every edge of the wave has met a negative distance.
// Global poses of the robot shapes at the configuration in ws.x (ws.Rpos / ws.Rquat), every lane group for its own
// point; ends on a block barrier.
template <int N, int GL, typename WS>
__device__ __forceinline__ void proximity_frames(const SceneDev* __restrict__ sc, const ShapeDev* __restrict__ robot,
                                                 const CPack<N>& cp, const double* __restrict__ base, WS& ws,
                                                 double* __restrict__ sink, int gl) {
  const bool lead = (gl == 0);
  // half-angle sin/cos, one joint per lane (strided: a 16-lane group may carry more than 16 / 2 joints)
  for (int jj = gl; jj < N; jj += GL) {
    double sn, cs;
    sincos(0.5 * ws.x[2 * jj], &sn, &cs);
    ws.cs[jj][0] = cs;
    ws.cs[jj][1] = sn;
  }
  __syncthreads();
  {  // revolute_joint_3D / rigid_link_3D kinematics, position + orientation only
    d3 pos = ld3(base);
    d4 Q = ld4(base + 3);
#pragma unroll
    for (int j = 0; j < N; ++j) {
      const int jb = j * 32;
      if (sc->branch_start[j]) {  // a new branch: base frame * mount pose (rigid_link_3D::doMotion from frame 0)
        const d4 bq = ld4(base + 3);
        pos = ld3(base) + mul(rotmat(bq), ld3(sc->mount_pos[j]));
        Q = qmul(bq, ld4(sc->mount_quat[j]));
      }
      const d3 axis_n = cget3(cp, jb + JC_AXISN);
      const double c2 = ws.cs[j][0], s2 = ws.cs[j][1];
      const d4 tq = d4{c2, axis_n.x * s2, axis_n.y * s2, axis_n.z * s2};
      const d4 EQ = qmul(Q, tq);
      st3(lead ? ws.Epos[j] : sink, pos);
      st4(lead ? ws.Equat[j] : sink, EQ);
      const m33 R = rotmat(EQ);
      pos = pos + mul(R, cget3(cp, jb + JC_OFFP));
      Q = qmul(EQ, d4{cget(cp, jb + JC_OFFQ), cget(cp, jb + JC_OFFQ + 1), cget(cp, jb + JC_OFFQ + 2),
                      cget(cp, jb + JC_OFFQ + 3)});
    }
  }
  __syncthreads();
  // robot shapes -> global pose (pose_3D::getGlobalPose, pose_3D.hpp:102-110)
  for (int r = gl; r < sc->n_robot; r += GL) {
    const ShapeDev& sh = robot[r];
    const int j = sh.link;
    const d3 pp = ld3(ws.Epos[j]);
    const d4 pq = ld4(ws.Equat[j]);
    st3(ws.Rpos[r], pp + qrot(pq, ld3(sh.pos)));
    st4(ws.Rquat[r], qmul(pq, ld4(sh.quat)));
  }
  __syncthreads();
}

template <int N, int GL, typename WS, bool GJK = true>
__device__ double proximity_min(const SceneDev* __restrict__ sc, const CPack<N>& cp,
                                const double* __restrict__ base, const ShapeDev* __restrict__ env_lds,
                                const PairDev* __restrict__ pairs, int n_pairs, WS& ws,
                                double* __restrict__ sink, int gl, int gb, bool cull_positive, bool group_done) {
  if (sc->planar) return proximity_min_planar<N, GL>(sc, cp, base, env_lds, pairs, n_pairs, ws, gl, gb);
  proximity_frames<N, GL>(sc, sc->robot, cp, base, ws, sink, gl);
  double dmin = INFINITY;
  bool hit = group_done;  // finished edges of the wave do not hold the scan open
  for (int p0 = 0; p0 < n_pairs; p0 += GL) {
    const int p = p0 + gl;
    double d = INFINITY;
    if (p < n_pairs && !hit) {
      const PairDev pr = pairs[p];
      const ShapeDev& rs = sc->robot[pr.robot];
      const ShapeDev& es = env_lds[pr.env];
      ShapeG A, Bv;
      A.kind = rs.kind;
      A.pos = ld3(ws.Rpos[pr.robot]);
      A.q = ld4(ws.Rquat[pr.robot]);
      A.d0 = rs.dims[0]; A.d1 = rs.dims[1]; A.d2 = rs.dims[2];
      Bv.kind = es.kind;
      Bv.pos = ld3(es.pos);
      Bv.q = ld4(es.quat);
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
      if (!skip) d = pr.s1_is_robot ? pair_distance<GJK>(pr.routine, A, Bv, sc->mesh_verts) : pair_distance<GJK>(pr.routine, Bv, A, sc->mesh_verts);
    }
    if (d < dmin) dmin = d;
    if (cull_positive) {
      // per-edge "any lane found a negative distance", then stop when every edge of the wave has one
      const unsigned long long m = __ballot(d < 0.0);
      const unsigned long long gm = (GL == 64) ? m : ((m >> gb) & ((1ull << (GL & 63)) - 1ull));
      hit = hit || (gm != 0ull);
      if (__all(hit)) break;
    }
  }
  // group min
#pragma unroll
  for (int off = GL / 2; off > 0; off >>= 1) {
    const double o = __shfl_xor(dmin, off, 64);
    if (o < dmin) dmin = o;
  }
  return dmin;
}

// exact left-to-right euclidean metric of a lane-distributed difference vector
// (vect_distance_metrics.hpp:126-137): stage the squares in LDS, every lane sums them in order
template <int N>
RKH_DI double group_norm(GroupWs<N>& ws, double diff, int gl) {
  if (gl < 2 * N) ws.tmp[gl] = diff * diff;
  __syncthreads();
  double s = 0.0;
#pragma unroll
  for (int d = 0; d < 2 * N; ++d) s = s + ws.tmp[d];
  __syncthreads();
  return sqrt(s);
}

template <int N, int GL>
struct SmemLayoutQs {
  static constexpr size_t block_bytes = (sizeof(BlockLdsQs<N, GL>) + 15) / 16 * 16;
  static size_t bytes(int n_env) { return block_bytes + size_t(n_env) * sizeof(ShapeDev); }
};
template <int N, int GL>
struct SmemLayout {
  static constexpr size_t block_bytes = (sizeof(BlockLds<N, GL>) + 15) / 16 * 16;
  static size_t bytes(int n_env) { return block_bytes + size_t(n_env) * sizeof(ShapeDev); }
};

struct WaveArgs {
  const SceneDev* sc;
  const PairDev* pairs;
  int n_pairs;
  DynDev dyn;
  EdgeIO io_a, io_b;
  const EdgeIO* tab_a;
  const EdgeIO* tab_b;
  uint32_t grid_a;
  KernelGate gate;
};
typedef const __attribute__((address_space(4))) WaveArgs* WaveArgP;
RKH_DI WaveArgP wave_args() {
  WaveArgP a = (WaveArgP)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(a));
  return a;
}

// ---------------------------------------------------------------------------------------------
// Kernel: steer edges.  Two edge groups per launch (planner: this round's steer candidates + the
// previous round's goal probes); 64/GL edges per wave.
// GL = 64 (one wave per edge, the latency mapping: rounds of fewer waves than the chip has SIMDs) may use the whole
// register file of its SIMD; four edges per wave (GL = 16) is the throughput form and keeps two waves per SIMD.
// DUO (GL = 64 only): two waves per edge.  Both run this body in step -- the same loads, the same RK4 glue, the same
// control flow, so every block barrier is met by both -- and differ inside state_derivative_duo; only the first wave
// writes results.
template <int N, int GL, bool GJK, bool DUO = false>
__global__ __launch_bounds__(DUO ? 128 : 64, GL == 64 ? 1 : 2) void propagate_kernel(WaveArgs) {
  static_assert(!DUO || GL == 64, "two waves per edge: one edge per block");
  // Every argument is read through the kernarg segment pointer at its point of use (by-value parameters are loaded in the
  // entry block and stay live in scalar registers; once those run out they are spilled into vector-register lanes, and
  // the EdgeIO / DynDev / KernelGate copies alone are ~170 dwords: the kernel then needed 256 registers + 185 spilled)
  const SceneDev* __restrict__ sc = wave_args()->sc;
  const PairDev* __restrict__ pairs = wave_args()->pairs;
  const int n_pairs = wave_args()->n_pairs;
  const uint32_t grid_a = wave_args()->grid_a;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  if (wave_args()->gate.count) {  // the planner's per-round choice between the kernel mappings
    const uint32_t c = *wave_args()->gate.count;
    if (c < wave_args()->gate.lo || c >= wave_args()->gate.hi) return;
  }
  constexpr int G = 64 / GL;
  BlockLds<N, GL>& lds = *reinterpret_cast<BlockLds<N, GL>*>(smem_raw);
  ShapeDev* env_lds = reinterpret_cast<ShapeDev*>(smem_raw + SmemLayout<N, GL>::block_bytes);
  // blockIdx.y selects the planning problem when per-problem EdgeIO tables are given
  bool group_b = blockIdx.x >= grid_a;
  uint32_t problem = blockIdx.y;
  uint32_t blk = group_b ? blockIdx.x - grid_a : blockIdx.x;
  if (wave_args()->gate.wave_base) {  // compact mapping (one edge per wave): block L of a 1-D grid takes working edge L
    const uint32_t L = blockIdx.x;
    if (L >= wave_args()->gate.wave_base[wave_args()->gate.n_segments]) return;
    uint32_t lo = 0, hi = wave_args()->gate.n_segments;
    while (hi - lo > 1) {
      const uint32_t mid = (lo + hi) >> 1;
      if (wave_args()->gate.wave_base[mid] <= L) lo = mid;
      else hi = mid;
    }
    problem = lo >> 1;
    group_b = (lo & 1u) != 0u;
    blk = L - wave_args()->gate.wave_base[lo];
  }
  auto edge_io = [&]() -> const EdgeIO* {
    WaveArgP A = wave_args();
    if (A->tab_a) return (group_b ? A->tab_b : A->tab_a) + problem;
    const char* ka = (const char*)(const void*)A;
    return (const EdgeIO*)(ka + (group_b ? offsetof(WaveArgs, io_b) : offsetof(WaveArgs, io_a)));
  };
  const uint32_t B = edge_io()->d_B ? *edge_io()->d_B : edge_io()->B;
  const int lane = DUO ? int(threadIdx.x & 63u) : int(threadIdx.x);
  const int wave = DUO ? int(threadIdx.x >> 6) : 0;
  const bool writer = !DUO || wave == 0;  // the wave whose results leave the block
  const int g = lane / GL, gl = lane % GL, gb = g * GL;
  const uint32_t e0 = blk * G;
  if (e0 >= B) return;
  const uint32_t e = e0 + g;
  const bool edge_valid = e < B;
  const uint32_t ec = edge_valid ? e : e0;  // idle groups shadow the wave's first edge, results discarded
  constexpr int D = 2 * N;
  stage_chain<N>(sc, lds.joints, lds.base, lane);
  stage_env(sc, env_lds, lane);
  GroupWs<N>& ws = lds.g[g];
  const uint32_t si = edge_io()->src_idx ? edge_io()->src_idx[ec] : ((edge_io()->d_src_first ? *edge_io()->d_src_first : 0u) + ec);
  const uint64_t trow = edge_io()->tgt_idx ? uint64_t(edge_io()->tgt_idx[ec]) : (edge_io()->d_tgt_off ? uint64_t(*edge_io()->d_tgt_off) : 0ull) + ec;
  const double a_d = (gl < D) ? edge_io()->src[uint64_t(si) * edge_io()->src_stride + gl] : 0.0;
  const double b_d = (gl < D) ? edge_io()->tgt[trow * edge_io()->tgt_stride + gl] : 0.0;
  const double lo = (gl < D) ? wave_args()->dyn.lower[gl] : 0.0;
  const double hi = (gl < D) ? wave_args()->dyn.upper[gl] : 0.0;
  if (gl < D) ws.b[gl] = b_d;
  double* __restrict__ record = (edge_valid && writer) ? edge_io()->record : nullptr;
  const int record_stride = edge_io()->record_stride;
  __syncthreads();
  const CPack<N> cp = load_cpack<N>(lds.joints, lane);

  double x = a_d;
  uint32_t n_free = 0;
  bool singular = false;   // a live edge met a singular mass matrix
  bool alive = true;       // this edge is still stepping
  uint32_t n_exec = 0;     // steps integrated for this edge (KernelGate::steps_exec)
  if (record && gl < D) record[(uint64_t(e) * record_stride + 0) * D + gl] = x;

  // steps of this edge: the launch's schedule, or (EdgeIO::frac) the edge's own travel fraction cut with the comparison
  // of the steer loop, current_time < fraction * (steps_per_edge * dt), current_time accumulated step by step
  int n_steps = wave_args()->dyn.n_steps;
  if (edge_io()->frac) {
    const double T_goal = edge_io()->frac[ec] * wave_args()->dyn.full_time;
    double current_time = 0.0;
    n_steps = 0;
    while (current_time < T_goal && n_steps < kMaxSteps) {
      current_time += wave_args()->dyn.dt;
      ++n_steps;
    }
  }
  if (edge_io()->mode == EDGE_POINT) {  // is_free(target): bounds, then proximity; no propagation
    n_steps = 0;
    x = b_d;
    bool oob = false;
    if (gl < D) {
      if (lo < hi) oob = (x < lo) || (x > hi);
      else oob = (x > lo) || (x < hi);
    }
    const unsigned long long m = __ballot(oob);
    const unsigned long long gm = (GL == 64) ? m : ((m >> gb) & ((1ull << (GL & 63)) - 1ull));
    bool free_pt = gm == 0ull;
    if (gl < D) ws.x[gl] = x;
    __syncthreads();
    const double dmin = proximity_min<N, GL, GroupWs<N>, GJK>(sc, cp, lds.base, env_lds, pairs, n_pairs, ws, lds.sink[lane], gl, gb, true, !free_pt);
    if (dmin < 0.0) free_pt = false;
    if (edge_valid && gl == 0 && writer) edge_io()->accept[e] = free_pt ? 1 : 0;
  }

  for (int k = 0; k < n_steps; ++k) {
    // distance(x_current, x_goal) > goal_proximity_threshold
    const double dist = group_norm<N>(ws, x - b_d, gl);
    if (!(dist > wave_args()->dyn.goal_tol)) alive = false;
    if (!__any(alive)) break;
    if (alive) ++n_exec;
    // PD law, zero-order hold over the step
    if (gl < D) ws.x[gl] = x;
    __syncthreads();
    if (gl < N) {
      double v = wave_args()->dyn.kp * (ws.b[2 * gl] - ws.x[2 * gl]) + wave_args()->dyn.kd * (ws.b[2 * gl + 1] - ws.x[2 * gl + 1]);
      if (v > wave_args()->dyn.u_max) v = wave_args()->dyn.u_max;
      else if (v < -wave_args()->dyn.u_max) v = -wave_args()->dyn.u_max;
      ws.u[gl] = v;
    }
    // runge_kutta4_integrate_impl (runge_kutta4_integrator_sys.hpp:53-97), time_step = dt.
    // One call site for f(x,u): each loop iteration of the reference evaluates f four times that
    // matter (the prime of :69 or the re-prime of :95, then :82, :86, :92); they are the stages of
    // a rolled loop here, which keeps a single copy of the dynamics in the instruction stream.
    // The re-prime after the last iteration is dead in the reference and is not evaluated.
    const double h = wave_args()->dyn.dt;
    double xe = x;  // end_point
    {
      double w = xe, k1 = 0.0, k2 = 0.0, k3 = 0.0;
      bool sing_now = false;
      const int n_evals = 4 * wave_args()->dyn.inner[k];
#pragma unroll 1
      for (int ev = 0; ev < n_evals; ++ev) {
        if (gl < D) ws.x[gl] = xe;
        __syncthreads();
        const double dp = DUO ? state_derivative_duo<N>(sc, cp, lds.joints, lds.base, ws, lds.sink[lane], gl, wave, &sing_now)
                              : state_derivative<N, GL>(sc, cp, lds.joints, lds.base, ws, lds.sink[lane], gl, gb, &sing_now);
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
      if (sing_now && alive) {
        singular = true;
        alive = false;
      }
    }
    // is_free(x_next): hyperbox bounds (hyperbox_topology.hpp:178-189), then proximity
    bool oob = false;
    if (gl < D) {
      if (lo < hi) oob = (xe < lo) || (xe > hi);
      else oob = (xe > lo) || (xe < hi);
    }
    {
      const unsigned long long m = __ballot(oob);
      const unsigned long long gm = (GL == 64) ? m : ((m >> gb) & ((1ull << (GL & 63)) - 1ull));
      if (gm != 0ull) alive = false;
    }
    if (!__any(alive)) break;
    if (gl < D) ws.x[gl] = xe;
    __syncthreads();
    const double dmin = proximity_min<N, GL, GroupWs<N>, GJK>(sc, cp, lds.base, env_lds, pairs, n_pairs, ws, lds.sink[lane], gl, gb, true, !alive);
    if (dmin < 0.0) alive = false;
    if (alive) {
      x = xe;
      ++n_free;
      if (record && gl < D) record[(uint64_t(e) * record_stride + n_free) * D + gl] = x;
    }
  }
  if (singular && gl == 0 && edge_valid && writer) atomicExch(edge_io()->err_flag, int(RKH_ERR_SINGULAR));
  if (edge_valid && gl < D && writer) edge_io()->x_out[uint64_t(e) * D + gl] = x;
  if (edge_valid && gl == 0 && writer) edge_io()->steps_free[e] = n_free;
  if (wave_args()->gate.steps_exec && edge_valid && gl == 0 && n_exec && writer)
    atomicAdd(wave_args()->gate.steps_exec, (unsigned long long)n_exec);
  if (edge_io()->mode != EDGE_PLAIN) {
    const double n_ar = group_norm<N>(ws, a_d - x, gl);
    const double n_ab = group_norm<N>(ws, a_d - b_d, gl);
    const double n_rb = group_norm<N>(ws, x - b_d, gl);
    if (edge_io()->mode == EDGE_STEER_ACCEPT) {
      // planning_visitor_base::steer_towards_position (planning_visitors.hpp:349-360)
      const double traveled = n_ar;
      const double best_case = edge_io()->best_case ? edge_io()->best_case[ec] : n_ab;
      const bool ok = (!isinf(traveled)) && (traveled < 2.0 * best_case) && (traveled > edge_io()->steer_tol * best_case);
      if (edge_valid && gl == 0 && writer) edge_io()->accept[e] = ok ? 1 : 0;
    } else if (edge_io()->mode == EDGE_CONNECT) {
      // planning_visitor_base::can_be_connected (planning_visitors.hpp:385-395); steer_tol carries the connection tolerance
      const bool ok = (!isinf(n_ar)) && (n_rb < edge_io()->steer_tol * n_ar);
      if (edge_valid && gl == 0 && writer) edge_io()->accept[e] = ok ? 1 : 0;
    } else if (edge_io()->mode == EDGE_WALK_ACCEPT) {
      // planning_visitor_base::random_walk (planning_visitors.hpp:418-421)
      const bool ok = (!isinf(n_ar)) && (n_ar > edge_io()->steer_tol * edge_io()->best_case[ec]);
      if (edge_valid && gl == 0 && writer) edge_io()->accept[e] = ok ? 1 : 0;
    } else if (edge_io()->mode == EDGE_GOAL_PROBE) {
      // C_free distance used by the goal probe (MEAQR_topology.hpp:995-1003)
      if (edge_valid && gl == 0 && writer) edge_io()->goal_dist[si - 1] = (n_ab * 0.05 > n_rb) ? n_ab : INFINITY;
    }
  }
}

// Kernel: x' = f(x,u) for B states (one wave each); also exports M and the bias force.
template <int N>
__global__ __launch_bounds__(64) void state_derivative_kernel(const SceneDev* __restrict__ sc, const double* __restrict__ x,
                                                               const double* __restrict__ u, uint32_t B,
                                                               double* __restrict__ pd, double* __restrict__ M,
                                                               double* __restrict__ f, int* __restrict__ err_flag) {
  __shared__ BlockLds<N, 64> lds;
  const uint32_t e = blockIdx.x;
  if (e >= B) return;
  const int lane = threadIdx.x;
  constexpr int D = 2 * N;
  stage_chain<N>(sc, lds.joints, lds.base, lane);
  GroupWs<N>& ws = lds.g[0];
  if (lane < D) ws.x[lane] = x[uint64_t(e) * D + lane];
  if (lane < N) ws.u[lane] = u[uint64_t(e) * N + lane];
  __syncthreads();
  const CPack<N> cp = load_cpack<N>(lds.joints, lane);
  bool singular = false;
  const double dp = state_derivative<N, 64>(sc, cp, lds.joints, lds.base, ws, lds.sink[lane], lane, 0, &singular);
  if (lane < D) pd[uint64_t(e) * D + lane] = dp;
  if (singular && lane == 0) atomicExch(err_flag, int(RKH_ERR_SINGULAR));
  // exports for the kernel-level parity tests: the symmetric M is rebuilt from Mf (ws.M now holds its
  // Cholesky factor); the bias force was parked in ws.tmp by state_derivative.
  if (M)
    for (int t = lane; t < N * N; t += 64) {
      const int i = t / N, j = t % N;
      const int lo = i < j ? i : j, hi = i < j ? j : i;
      M[uint64_t(e) * N * N + t] = (i == j) ? ws.Mf[i][i] : 0.5 * (ws.Mf[lo][hi] + ws.Mf[hi][lo]);
    }
  if (f && lane < N) f[uint64_t(e) * N + lane] = ws.tmp[lane];
}

// ---------------------------------------------------------------------------------------------
// Kernel: quasi-static edge walk interp_topo_move_position_toward_pred
// (ctrl/interpolation/interpolated_topologies.hpp:137-163) over joint positions (D = N), with
// manip_quasi_static_env::is_free (manip_free_workspace.hpp:154-156) as the predicate.
// One block of W waves per edge; its (64 / GL) W lane groups test that many consecutive interpolation points at a time
// (the points of a straight edge are independent), and the first colliding one is found from the groups' verdicts.
// dist_cur is accumulated by repeated addition exactly like the reference loop (":157 dist_cur += min_interval"),
// so the number of tested points is the same integer.  Three shapes of the same code (launch_edge_check):
//   GL = 16, W = 1  -- a launch with thousands of edges (the batch planner's rounds) fills the machine with one wave per
//                      edge, 4 points at a time, each 16-lane group running the chain kinematics once for its point;
//   GL = 16, W = 4  -- a few hundred to a few thousand edges (a step of a batch of graph planners): 16 points at a time;
//   GL = 64, W = 16 -- a few hundred edges (a step of one or a few graph planners): the machine is empty and the step
//                      waits for the longest edge, so a whole wave scans the shape pairs of ONE point (300 pairs in
//                      5 passes instead of 19) and 16 waves take 16 points at a time.  A single-problem RRT* spent 85 %
//                      of its time in this kernel at 230 us per launch with one wave per edge.
template <int N, int GL, int W>
struct BlockLdsQsW {
  JointLds joints[N];
  double base[10];
  double sink[64][4];     // dummy store targets of the non-leading lanes (never read; shared by the waves)
  GroupWsQs<N> g[(64 / GL) * W];
  double pts[(64 / GL) * W][N];   // the groups' interpolation points (space coordinates)
  uint32_t masks[W];      // per wave: bit t = group t's point lies on the edge, bit 4 + t = it passed the predicate
  // block-level verdicts (proximity_verdicts_block): per group bit 0 = its point is to be tested, bit 1 = a shape pair
  // of it is closer than 0; the (pair, group) combinations that survive the bounding-sphere cull
  static constexpr int kQueueCap = 128 * W;
  uint32_t flags[(64 / GL) * W];
  uint32_t q_cnt;
  uint32_t queue[kQueueCap];
  ShapeDev robot[2 * N];  // the scene's robot shapes (SceneDev::robot)
};
template <int N, int GL, int W>
struct SmemLayoutQsW {
  static constexpr size_t block_bytes = (sizeof(BlockLdsQsW<N, GL, W>) + 15) / 16 * 16;
  // env shapes, then (if it fits: edge_check_kernel's pairs_staged) the proxy pair list
  static size_t bytes(int n_env, int n_pairs_staged) {
    return block_bytes + size_t(n_env) * sizeof(ShapeDev) + size_t(n_pairs_staged) * sizeof(PairDev);
  }
};
// The predicate's proximity half for all G points of a pass at once (3D scenes; robot shape poses already in
// lds.g[t], lds.flags[t] = 1 for the points to test, lds.q_cnt = 0).  is_free only asks whether SOME proxy pair is
// closer than 0 (manip_free_workspace.hpp:154-156), so the pairs need no order: every thread culls (pair, point)
// combinations by the bounding spheres -- exactly findMinimumDistance's test against a running minimum of 0
// (proxy_query_model.cpp:384-389), behind a cheaper test on squares that only ever skips what that one skips -- and
// pushes the survivors into an LDS queue; then the block's threads take one surviving combination each and run its
// closed form (bit 1 of the point's flag word = some pair of it is closer than 0).  A lane group walking the pair list
// by itself (proximity_min) pays one closed form per 16 pairs as soon as any lane of the WAVE has a survivor: 9-19 in a
// row per point; here a pass costs n_pairs G / threads culls and, nearly always, ONE round of closed forms.  Should the
// survivors not fit the queue, the pair list is redone in slices that fit whatever survives.
template <int N, int GL, int W, bool GJK>
__device__ __forceinline__ void proximity_verdicts_block(const SceneDev* __restrict__ sc, const ShapeDev* __restrict__ env_lds,
                                                         const PairDev* __restrict__ pairs, int n_pairs,
                                                         BlockLdsQsW<N, GL, W>& lds, int tid) {
  constexpr int G = (64 / GL) * W, QCAP = BlockLdsQsW<N, GL, W>::kQueueCap;
  int p_lo = 0, p_step = n_pairs;
  while (p_lo < n_pairs) {
    const int p_hi = (p_lo + p_step < n_pairs) ? p_lo + p_step : n_pairs;
    for (int idx = p_lo * G + tid; idx < p_hi * G; idx += 64 * W) {
      const int t = idx % G, p = idx / G;
      if (lds.flags[t] != 1u) continue;
      const PairDev pr = pairs[p];
      const ShapeDev& rs = lds.robot[pr.robot];
      const ShapeDev& es = env_lds[pr.env];
      const d3 ca = ld3(lds.g[t].Rpos[pr.robot]), cb = ld3(es.pos);
      const d3 dc = pr.s1_is_robot ? cb - ca : ca - cb;
      const double r1 = pr.s1_is_robot ? rs.brad : es.brad, r2 = pr.s1_is_robot ? es.brad : rs.brad;
      const double sq = ((0.0 + dc.x * dc.x) + dc.y * dc.y) + dc.z * dc.z, rr = r1 + r2;
      if (sq > (rr * rr) * (1.0 + 1e-9)) continue;   // clearly apart: the exact test below skips it too
      if (sqrt(sq) - r1 - r2 > 0.0) continue;        // |c2 - c1| - r1 - r2 > 0 (transformToGlobal(0) is the shape's position)
      const uint32_t slot = atomicAdd(&lds.q_cnt, 1u);
      if (slot < uint32_t(QCAP)) lds.queue[slot] = uint32_t(t) | (uint32_t(p) << 8);
    }
    __syncthreads();
    const uint32_t cnt = lds.q_cnt;
    __syncthreads();
    if (tid == 0) lds.q_cnt = 0u;
    if (cnt > uint32_t(QCAP)) {  // (block-uniform) does not fit: slices of QCAP / G pairs always do
      p_step = QCAP / G;
      __syncthreads();
      continue;
    }
    for (uint32_t i = tid; i < cnt; i += 64 * W) {
      const uint32_t ent = lds.queue[i];
      const int t = int(ent & 255u);
      if (lds.flags[t] != 1u) continue;              // this point already has a colliding pair
      const PairDev pr = pairs[ent >> 8];
      const ShapeDev& rs = lds.robot[pr.robot];
      const ShapeDev& es = env_lds[pr.env];
      ShapeG A, Bv;
      A.kind = rs.kind;
      A.pos = ld3(lds.g[t].Rpos[pr.robot]);
      A.q = ld4(lds.g[t].Rquat[pr.robot]);
      A.d0 = rs.dims[0]; A.d1 = rs.dims[1]; A.d2 = rs.dims[2];
      Bv.kind = es.kind;
      Bv.pos = ld3(es.pos);
      Bv.q = ld4(es.quat);
      Bv.d0 = es.dims[0]; Bv.d1 = es.dims[1]; Bv.d2 = es.dims[2];
      const double d = pr.s1_is_robot ? pair_distance<GJK>(pr.routine, A, Bv, sc->mesh_verts)
                                      : pair_distance<GJK>(pr.routine, Bv, A, sc->mesh_verts);
      if (d < 0.0) atomicOr(&lds.flags[t], 2u);
    }
    __syncthreads();
    p_lo = p_hi;
  }
}

// One edge walk of edge_check_kernel (the block's groups test G consecutive points per pass).
template <int N, int GL, int W, bool GJK>
__device__ __forceinline__ void edge_walk(const SceneDev* __restrict__ sc, const PairDev* __restrict__ pairs, int n_pairs,
                                          const QsDev& qs, const EdgeIO& io, BlockLdsQsW<N, GL, W>& lds,
                                          const ShapeDev* __restrict__ env_lds, uint32_t e) {
  constexpr int GPW = 64 / GL, G = GPW * W;  // groups per wave, groups (= points per pass) of the block
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int g = tid / GL, gl = lane % GL, gb = (lane / GL) * GL;  // group of the block, lane of the group, its base lane in the wave
  GroupWsQs<N>& ws = lds.g[g];
  const uint32_t si = io.src_idx ? io.src_idx[e] : ((io.d_src_first ? *io.d_src_first : 0u) + e);
  const uint64_t trow = io.tgt_idx ? uint64_t(io.tgt_idx[e]) : ((io.d_tgt_off ? uint64_t(*io.d_tgt_off) : 0ull) + e);
  const double a_d = (gl < N) ? io.src[uint64_t(si) * io.src_stride + gl] : 0.0;
  const double b_d = (gl < N) ? io.tgt[trow * io.tgt_stride + gl] : 0.0;
  const double lo = (gl < N) ? qs.lower[gl] : 0.0;
  const double hi = (gl < N) ? qs.upper[gl] : 0.0;
  const double speed = (gl < N) ? qs.speed[gl] : 1.0;  // rate-limited space: the model sees point * speed limit
  for (int t = gl; t < 2 * N; t += GL) ws.x[t] = 0.0;  // velocities stay zero (apply_to_model writes positions only)
  __syncthreads();
  const CPack<N> cp = load_cpack<N>(lds.joints, lane);

  // exact left-to-right euclidean norm of an N-vector held by lanes gl < N of every group
  auto norm_n = [&](double diff) {
    if (gl < N) ws.tmp[gl] = diff * diff;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int d = 0; d < N; ++d) s = s + ws.tmp[d];
    __syncthreads();
    return sqrt(s);
  };
  if (io.mode == EDGE_POINT) {
    // is_free(target): hyperbox bounds, then proximity (manip_free_workspace.hpp:79-99,154-156); group 0 tests it
    if (gl < N) ws.x[2 * gl] = b_d * speed;
    bool oob = false;
    if (gl < N) {
      if (lo < hi) oob = (b_d < lo) || (b_d > hi);
      else oob = (b_d > lo) || (b_d < hi);
    }
    const unsigned long long mo = __ballot(oob);
    const bool group_oob = (GL == 64 ? mo : ((mo >> gb) & ((1ull << (GL & 63)) - 1ull))) != 0ull;
    bool is_free;
    if (!sc->planar) {
      if (gl == 0) lds.flags[g] = (g == 0 && !group_oob) ? 1u : 0u;
      if (tid == 0) lds.q_cnt = 0u;
      __syncthreads();
      proximity_frames<N, GL>(sc, lds.robot, cp, lds.base, ws, lds.sink[lane], gl);
      proximity_verdicts_block<N, GL, W, GJK>(sc, env_lds, pairs, n_pairs, lds, tid);
      is_free = (lds.flags[0] == 1u);
    } else {
      __syncthreads();
      const double dmin = proximity_min_planar<N, GL>(sc, cp, lds.base, env_lds, pairs, n_pairs, ws, gl, gb);
      is_free = !group_oob && !(dmin < 0.0);
    }
    if (g == 0 && gl < N) io.x_out[uint64_t(e) * N + gl] = b_d;
    if (tid == 0) {
      io.steps_free[e] = 1;
      io.accept[e] = is_free ? 1 : 0;
    }
    return;
  }
  const double dist_tot = norm_n(a_d - b_d);
  const double fraction = io.frac ? io.frac[e] : qs.fraction;
  double result = a_d;       // component gl of the returned point
  uint32_t n_checked = 0;
  bool completed_walk = false;  // the predicate loop ran to dist_inter without a collision (EDGE_STEER_BOTH)
  if (dist_tot == INFINITY) {
    result = a_d;
  } else if (dist_tot < qs.min_interval) {
    result = a_d + (b_d - a_d) * fraction;  // vector_topology::move_position_toward, no predicate call
  } else {
    const double dist_inter = dist_tot * fraction;
    double cur = 0.0;          // dist_cur of the last tested point (0 + min_interval == min_interval exactly)
    double last_result = a_d;  // last point that passed the predicate
    bool collided = false;
    for (;;) {
      // dist_cur of this group's point and of the block's first and last ones: cur + min_interval, repeatedly
      double my = cur, last = cur;
#pragma unroll
      for (int t = 0; t < G; ++t) {
        if (t <= g) my = my + qs.min_interval;
        last = last + qs.min_interval;
      }
      if (!((cur + qs.min_interval) < dist_inter)) break;  // not even the first point is on the edge (block-uniform)
      const bool valid = my < dist_inter;
      const double pt = a_d + (b_d - a_d) * (my / dist_tot);
      if (gl < N) {
        ws.x[2 * gl] = pt * speed;
        lds.pts[g][gl] = pt;
      }
      bool oob = false;
      if (gl < N) {
        if (lo < hi) oob = (pt < lo) || (pt > hi);
        else oob = (pt > lo) || (pt < hi);
      }
      const unsigned long long mo = __ballot(oob);
      const bool group_oob = (GL == 64 ? mo : ((mo >> gb) & ((1ull << (GL & 63)) - 1ull))) != 0ull;
      bool is_free;
      if (!sc->planar) {
        if (gl == 0) lds.flags[g] = (valid && !group_oob) ? 1u : 0u;
        if (tid == 0) lds.q_cnt = 0u;
        __syncthreads();
        proximity_frames<N, GL>(sc, lds.robot, cp, lds.base, ws, lds.sink[lane], gl);
        proximity_verdicts_block<N, GL, W, GJK>(sc, env_lds, pairs, n_pairs, lds, tid);
        is_free = (lds.flags[g] == 1u);
      } else {  // planar scenes: the verdict depends on the finder order (proximity_min_planar)
        __syncthreads();
        const double dmin = proximity_min_planar<N, GL>(sc, cp, lds.base, env_lds, pairs, n_pairs, ws, gl, gb);
        is_free = valid && !group_oob && !(dmin < 0.0);
      }
      {  // the wave's verdicts -> LDS, then every thread scans the block's groups in edge order
        const unsigned long long mv = __ballot(valid && gl == 0), mf = __ballot(is_free && gl == 0);
        uint32_t m = 0;
#pragma unroll
        for (int t = 0; t < GPW; ++t)
          m |= (uint32_t((mv >> ((t * GL) & 63)) & 1ull) << t) | (uint32_t((mf >> ((t * GL) & 63)) & 1ull) << (4 + t));
        if (lane == 0) lds.masks[wave] = m;
      }
      __syncthreads();
      int first_bad = G;  // first valid group whose point fails the predicate
      int n_valid = 0;
#pragma unroll
      for (int t = G - 1; t >= 0; --t) {
        const uint32_t m = lds.masks[t / GPW];
        const bool v = (m >> (t % GPW)) & 1u, f = (m >> (4 + (t % GPW))) & 1u;
        if (v && !f) first_bad = t;
        if (v) n_valid = n_valid > t + 1 ? n_valid : t + 1;
      }
      if (first_bad < G) {
        n_checked += uint32_t(first_bad + 1);
        if (first_bad > 0 && gl < N) last_result = lds.pts[first_bad - 1][gl];
        collided = true;
        break;
      }
      n_checked += uint32_t(n_valid);
      if (gl < N) last_result = lds.pts[n_valid - 1][gl];
      if (n_valid < G) break;  // reached dist_inter without a collision
      cur = last;
      __syncthreads();         // the points and verdicts are rewritten by the next pass
    }
    completed_walk = !collided;
    if (collided) result = last_result;
    else if (fraction == 1.0) result = b_d;  // exact end fractions (:159-162)
    else if (fraction == 0.0) result = a_d;
    else result = a_d + (b_d - a_d) * fraction;
  }
  if (g == 0 && gl < N) io.x_out[uint64_t(e) * N + gl] = result;
  if (tid == 0) io.steps_free[e] = n_checked;
  if (io.mode != EDGE_PLAIN) {
    const double n_ar = norm_n(a_d - result);
    const double n_ab = dist_tot;
    const double n_rb = norm_n(result - b_d);
    if (io.mode == EDGE_STEER_ACCEPT || io.mode == EDGE_STEER_BOTH) {
      // planning_visitor_base::steer_towards_position (planning_visitors.hpp:349-360)
      const double best_case = io.best_case ? io.best_case[e] : n_ab;
      const bool ok = (!isinf(n_ar)) && (n_ar < 2.0 * best_case) && (n_ar > io.steer_tol * best_case);
      // EDGE_STEER_BOTH: bit 1 = the walk completed.  steer_back_to_position(target, source) walks the same points
      // (move_position_back_to, interpolated_topologies.hpp:165-191) and returns the same point unless the walk
      // completes, where it returns the source itself (:185-186): its verdict is bit 0 && !bit 1.
      if (tid == 0) io.accept[e] = (ok ? 1 : 0) | ((io.mode == EDGE_STEER_BOTH && completed_walk) ? 2 : 0);
    } else if (io.mode == EDGE_GOAL_PROBE) {
      // interp_topo_get_distance_pred (interpolated_topologies.hpp:193-199)
      if (tid == 0) io.goal_dist[si - 1] = (n_rb < DBL_EPSILON) ? n_ab : INFINITY;
    } else if (io.mode == EDGE_CONNECT) {
      // planning_visitor_base::can_be_connected (planning_visitors.hpp:385-395)
      const bool ok = (!isinf(n_ar)) && (n_rb < io.steer_tol * n_ar);  // steer_tol carries the connection tolerance
      if (tid == 0) io.accept[e] = ok ? 1 : 0;
    } else if (io.mode == EDGE_WALK_ACCEPT) {
      // planning_visitor_base::random_walk (planning_visitors.hpp:418-421)
      const bool ok = (!isinf(n_ar)) && (n_ar > io.steer_tol * io.best_case[e]);
      if (tid == 0) io.accept[e] = ok ? 1 : 0;
    }
  }
}


template <int N, int GL, int W>
__device__ __forceinline__ const PairDev* edge_check_stage(const SceneDev* __restrict__ sc, const PairDev* __restrict__ pairs,
                                                           int n_pairs, int pairs_staged, BlockLdsQsW<N, GL, W>& lds,
                                                           ShapeDev* env_lds) {
  const int tid = threadIdx.x, lane = tid & 63;
  stage_chain<N>(sc, lds.joints, lds.base, lane);  // (every wave writes the same values)
  stage_env(sc, env_lds, lane);
  // robot shapes and, when the launch made room for it, the pair list: the cull loop reads both per combination
  const int n_words = sc->n_robot * int(sizeof(ShapeDev) / sizeof(double));
  for (int i = tid; i < n_words; i += 64 * W)
    reinterpret_cast<double*>(lds.robot)[i] = reinterpret_cast<const double*>(sc->robot)[i];
  if (!pairs_staged) return pairs;
  PairDev* pl = reinterpret_cast<PairDev*>(env_lds + sc->n_env);
  for (int i = tid; i < n_pairs; i += 64 * W) pl[i] = pairs[i];
  return pl;
}

template <int N, int GL, int W, bool GJK>
__global__ __launch_bounds__(64 * W) void edge_check_kernel(const SceneDev* __restrict__ sc,
                                                            const PairDev* __restrict__ pairs, int n_pairs, QsDev qs,
                                                            EdgeIO io_a, EdgeIO io_b, const EdgeIO* __restrict__ tab_a,
                                                            const EdgeIO* __restrict__ tab_b, uint32_t grid_a,
                                                            int pairs_staged) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  BlockLdsQsW<N, GL, W>& lds = *reinterpret_cast<BlockLdsQsW<N, GL, W>*>(smem_raw);
  ShapeDev* env_lds = reinterpret_cast<ShapeDev*>(smem_raw + SmemLayoutQsW<N, GL, W>::block_bytes);
  const bool group_b = blockIdx.x >= grid_a;
  const EdgeIO io = tab_a ? (group_b ? tab_b[blockIdx.y] : tab_a[blockIdx.y]) : (group_b ? io_b : io_a);
  const uint32_t B = io.d_B ? *io.d_B : io.B;
  const uint32_t e = group_b ? blockIdx.x - grid_a : blockIdx.x;
  if (e >= B) return;
  pairs = edge_check_stage<N, GL, W>(sc, pairs, n_pairs, pairs_staged, lds, env_lds);
  edge_walk<N, GL, W, GJK>(sc, pairs, n_pairs, qs, io, lds, env_lds, e);
}

// ---------------------------------------------------------------------------------------------
// The same edge walk for 3D scenes with ONE LANE PER POINT in the chain kinematics (edge_points_kernel).  In
// edge_check_kernel a point belongs to a 16-lane group whose lanes all run the serial base -> tip chain of that one
// point: ~1.2 k fp64 instructions per wave for 4 points.  Here a block of 4 waves takes G = 64 (32 for chains of more
// than 7 joints) consecutive points per pass and every phase is spread over what it is parallel in:
//   half-angle sin / cos   one (point, joint) per thread
//   joint end frames       one POINT per lane of the first wave: the serial chain, once per 64 points
//   robot shape poses      one (point, shape) per thread
//   bounding-sphere cull   one (point, pair) per thread -> survivors into the LDS queue (as proximity_verdicts_block)
//   closed forms           one surviving (point, pair) per thread
// and the first colliding point comes out of two ballots.  Per-point data sit in LDS component-major ([..][G]: lanes of
// a wave = consecutive points = consecutive addresses).  Same arithmetic per point as edge_check_kernel (the same
// device functions in the same order), so n_checked, results and verdicts are the same bits; planar scenes, whose
// verdict depends on the finder order, stay on edge_check_kernel.
struct EdgePointsCfg {
  static constexpr int T = 256;                // threads per block
  static constexpr int QCAP = 1024;            // surviving (point, pair) combinations per round of closed forms
};
// G = points per pass: 64 for the launches of few edges (the step waits for its longest walk), 32 for the launches of
// thousands of edges (half the LDS, three blocks per CU) and for chains of more than 7 joints
template <int N, int G>
struct __attribute__((aligned(16))) EdgePointsLds {
  JointLds joints[N];
  double base[10];
  ShapeDev robot[2 * N];
  double a[N], b[N], res[N];      // the edge's end points; staging of an N-vector for the ordered sums
  double pts[N][G];               // the pass's interpolation points (space coordinates)
  double cs[N][2][G];             // cos, sin of the half joint angles
  double E[N][7][G];              // joint end frames: position, quaternion
  double R[2 * N][7][G];          // robot shapes, global pose
  uint32_t flags[G];              // bit 0: to be tested (on the edge, inside the bounds), bit 1: a pair closer than 0, bit 2: on the edge
  uint32_t q_cnt;
  uint32_t scan[2];               // first point that fails the predicate, points on the edge
  uint32_t queue[EdgePointsCfg::QCAP];
};
template <int N, int G>
struct EdgePointsSmem {
  static constexpr size_t block_bytes = (sizeof(EdgePointsLds<N, G>) + 15) / 16 * 16;
  static size_t bytes(int n_env, int n_pairs_staged) {
    return block_bytes + size_t(n_env) * sizeof(ShapeDev) + size_t(n_pairs_staged) * sizeof(PairDev);
  }
};

struct EdgePointsArgs {
  const SceneDev* sc;
  const PairDev* pairs;
  int n_pairs;
  QsDev qs;
  EdgeIO io_a, io_b;
  const EdgeIO* tab_a;
  const EdgeIO* tab_b;
  uint32_t grid_a;
  int pairs_staged;
};
typedef const __attribute__((address_space(4))) EdgePointsArgs* EdgePointsArgP;
template <int N, bool GJK, int G>
__global__ __launch_bounds__(256) void edge_points_kernel(EdgePointsArgs) {
  // arguments are read through the kernarg segment pointer (a by-value record indexed at run time -- qs.speed[j], the
  // choice between io_a and io_b -- would be copied to scratch)
  EdgePointsArgP ka = (EdgePointsArgP)__builtin_amdgcn_kernarg_segment_ptr();
  const SceneDev* __restrict__ sc = ka->sc;
  const PairDev* __restrict__ pairs = ka->pairs;
  const int n_pairs = ka->n_pairs, pairs_staged = ka->pairs_staged;
  const uint32_t grid_a = ka->grid_a;
  const EdgeIO* __restrict__ tab_a = ka->tab_a;
  const EdgeIO* __restrict__ tab_b = ka->tab_b;
  const auto& qs = ka->qs;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr int T = EdgePointsCfg::T, QCAP = EdgePointsCfg::QCAP;
  EdgePointsLds<N, G>& lds = *reinterpret_cast<EdgePointsLds<N, G>*>(smem_raw);
  ShapeDev* env_lds = reinterpret_cast<ShapeDev*>(smem_raw + EdgePointsSmem<N, G>::block_bytes);
  const bool group_b = blockIdx.x >= grid_a;
  const EdgeIO* iop = tab_a ? (group_b ? &tab_b[blockIdx.y] : &tab_a[blockIdx.y]) : nullptr;
  const __attribute__((address_space(4))) EdgeIO* iok = group_b ? &ka->io_b : &ka->io_a;
#define RKH_IO(f) (iop ? iop->f : iok->f)
  const uint32_t B = RKH_IO(d_B) ? *RKH_IO(d_B) : RKH_IO(B);
  const uint32_t e = group_b ? blockIdx.x - grid_a : blockIdx.x;
  if (e >= B) return;
  const int tid = threadIdx.x, lane = tid & 63;
  stage_chain<N>(sc, lds.joints, lds.base, lane);  // (every wave writes the same values)
  stage_env(sc, env_lds, lane);
  {
    const int n_words = sc->n_robot * int(sizeof(ShapeDev) / sizeof(double));
    for (int i = tid; i < n_words; i += T) reinterpret_cast<double*>(lds.robot)[i] = reinterpret_cast<const double*>(sc->robot)[i];
    if (pairs_staged) {
      PairDev* pl = reinterpret_cast<PairDev*>(env_lds + sc->n_env);
      for (int i = tid; i < n_pairs; i += T) pl[i] = pairs[i];
      pairs = pl;
    }
  }
  const int mode = RKH_IO(mode);
  if (tid < N) {
    const uint32_t* src_idx = RKH_IO(src_idx);
    const uint32_t* d_src_first = RKH_IO(d_src_first);
    const uint32_t* tgt_idx = RKH_IO(tgt_idx);
    const uint32_t* d_tgt_off = RKH_IO(d_tgt_off);
    const uint32_t si = src_idx ? src_idx[e] : ((d_src_first ? *d_src_first : 0u) + e);
    const uint64_t trow = tgt_idx ? uint64_t(tgt_idx[e]) : ((d_tgt_off ? uint64_t(*d_tgt_off) : 0ull) + e);
    lds.a[tid] = RKH_IO(src)[uint64_t(si) * RKH_IO(src_stride) + tid];
    lds.b[tid] = RKH_IO(tgt)[trow * RKH_IO(tgt_stride) + tid];
  }
  __syncthreads();
  const CPack<N> cp = load_cpack<N>(lds.joints, lane);
  const int n_robot = sc->n_robot;

  // exact left-to-right euclidean norm of the N-vector whose component d thread d holds (vect_distance_metrics.hpp:126-137)
  auto norm_n = [&](double diff) {
    if (tid < N) lds.res[tid] = diff * diff;
    __syncthreads();
    double sacc = 0.0;
#pragma unroll
    for (int d = 0; d < N; ++d) sacc = sacc + lds.res[d];
    __syncthreads();
    return sqrt(sacc);
  };
  // is_free's proximity half for the points flagged 1 in lds.flags (their coordinates in lds.pts): sets bit 1 of the
  // flag of every point with a proxy pair closer than 0.  Block-uniform control flow throughout.
  auto test_points = [&](int n_pts) {
    for (int idx = tid; idx < G * N; idx += T) {   // half-angle sin / cos (revolute_joint_3D::doMotion)
      const int t = idx % G, j = idx / G;
      if (t >= n_pts) continue;
      double sn, cs;
      sincos(0.5 * (lds.pts[j][t] * qs.speed[j]), &sn, &cs);
      lds.cs[j][0][t] = cs;
      lds.cs[j][1][t] = sn;
    }
    __syncthreads();
    if (tid < G) {  // revolute_joint_3D / rigid_link_3D kinematics, position + orientation only: this lane's point
      const int t = tid;
      d3 pos = ld3(lds.base);
      d4 Q = ld4(lds.base + 3);
#pragma unroll
      for (int j = 0; j < N; ++j) {
        const int jb = j * 32;
        if (sc->branch_start[j]) {  // a new branch: base frame * mount pose (rigid_link_3D::doMotion from frame 0)
          const d4 bq = ld4(lds.base + 3);
          pos = ld3(lds.base) + mul(rotmat(bq), ld3(sc->mount_pos[j]));
          Q = qmul(bq, ld4(sc->mount_quat[j]));
        }
        const d3 axis_n = cget3(cp, jb + JC_AXISN);
        const double c2 = lds.cs[j][0][t], s2 = lds.cs[j][1][t];
        const d4 tq = d4{c2, axis_n.x * s2, axis_n.y * s2, axis_n.z * s2};
        const d4 EQ = qmul(Q, tq);
        lds.E[j][0][t] = pos.x; lds.E[j][1][t] = pos.y; lds.E[j][2][t] = pos.z;
        lds.E[j][3][t] = EQ.w; lds.E[j][4][t] = EQ.x; lds.E[j][5][t] = EQ.y; lds.E[j][6][t] = EQ.z;
        const m33 Rm = rotmat(EQ);
        pos = pos + mul(Rm, cget3(cp, jb + JC_OFFP));
        Q = qmul(EQ, d4{cget(cp, jb + JC_OFFQ), cget(cp, jb + JC_OFFQ + 1), cget(cp, jb + JC_OFFQ + 2),
                        cget(cp, jb + JC_OFFQ + 3)});
      }
    }
    __syncthreads();
    for (int idx = tid; idx < G * n_robot; idx += T) {  // robot shapes -> global pose (pose_3D::getGlobalPose, pose_3D.hpp:102-110)
      const int t = idx % G, r = idx / G;
      const ShapeDev& sh = lds.robot[r];
      const int j = sh.link;
      const d3 pp = d3{lds.E[j][0][t], lds.E[j][1][t], lds.E[j][2][t]};
      const d4 pq = d4{lds.E[j][3][t], lds.E[j][4][t], lds.E[j][5][t], lds.E[j][6][t]};
      const d3 gp = pp + qrot(pq, ld3(sh.pos));
      const d4 gq = qmul(pq, ld4(sh.quat));
      lds.R[r][0][t] = gp.x; lds.R[r][1][t] = gp.y; lds.R[r][2][t] = gp.z;
      lds.R[r][3][t] = gq.w; lds.R[r][4][t] = gq.x; lds.R[r][5][t] = gq.y; lds.R[r][6][t] = gq.z;
    }
    __syncthreads();
    // bounding-sphere cull of every (pair, point), survivors -> queue -> closed forms (see proximity_verdicts_block)
    int p_lo = 0, p_step = n_pairs;
    while (p_lo < n_pairs) {
      const int p_hi = (p_lo + p_step < n_pairs) ? p_lo + p_step : n_pairs;
      {  // a wave's lanes are consecutive points of ONE pair (two pairs when G = 32): the pair's constants are uniform
        constexpr int PPB = T / G;   // pairs per block iteration
        const int t = tid % G;
        const bool live = lds.flags[t] == 5u;
        auto cull = [&](int p) {
          const PairDev pr = pairs[p];
          const ShapeDev& rs = lds.robot[pr.robot];
          const ShapeDev& es = env_lds[pr.env];
          const d3 ca = d3{lds.R[pr.robot][0][t], lds.R[pr.robot][1][t], lds.R[pr.robot][2][t]}, cb = ld3(es.pos);
          const d3 dc = pr.s1_is_robot ? cb - ca : ca - cb;
          const double r1 = pr.s1_is_robot ? rs.brad : es.brad, r2 = pr.s1_is_robot ? es.brad : rs.brad;
          const double sq = ((0.0 + dc.x * dc.x) + dc.y * dc.y) + dc.z * dc.z, rr = r1 + r2;
          if (sq > (rr * rr) * (1.0 + 1e-9)) return false;   // clearly apart: the exact test below skips it too
          return !(sqrt(sq) - r1 - r2 > 0.0);                // |c2 - c1| - r1 - r2 > 0 (proxy_query_model.cpp:384-389, minimum 0)
        };
        int p = p_lo + tid / G;
        for (; p + 3 * PPB < p_hi; p += 4 * PPB) {  // four independent pairs in flight
          const bool k0 = cull(p), k1 = cull(p + PPB), k2 = cull(p + 2 * PPB), k3 = cull(p + 3 * PPB);
          if (live) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const bool k = u == 0 ? k0 : (u == 1 ? k1 : (u == 2 ? k2 : k3));
              if (k) {
                const uint32_t slot = atomicAdd(&lds.q_cnt, 1u);
                if (slot < uint32_t(QCAP)) lds.queue[slot] = uint32_t(t) | (uint32_t(p + u * PPB) << 8);
              }
            }
          }
        }
        for (; p < p_hi; p += PPB) {
          if (live && cull(p)) {
            const uint32_t slot = atomicAdd(&lds.q_cnt, 1u);
            if (slot < uint32_t(QCAP)) lds.queue[slot] = uint32_t(t) | (uint32_t(p) << 8);
          }
        }
      }
      __syncthreads();
      const uint32_t cnt = lds.q_cnt;
      __syncthreads();
      if (tid == 0) lds.q_cnt = 0u;
      if (cnt > uint32_t(QCAP)) {  // (block-uniform) does not fit: slices of QCAP / G pairs always do
        p_step = QCAP / G;
        __syncthreads();
        continue;
      }
      for (uint32_t i = tid; i < cnt; i += T) {
        const uint32_t ent = lds.queue[i];
        const int t = int(ent & 255u);
        if (lds.flags[t] != 5u) continue;              // this point already has a colliding pair
        const PairDev pr = pairs[ent >> 8];
        const ShapeDev& rs = lds.robot[pr.robot];
        const ShapeDev& es = env_lds[pr.env];
        ShapeG A, Bv;
        A.kind = rs.kind;
        A.pos = d3{lds.R[pr.robot][0][t], lds.R[pr.robot][1][t], lds.R[pr.robot][2][t]};
        A.q = d4{lds.R[pr.robot][3][t], lds.R[pr.robot][4][t], lds.R[pr.robot][5][t], lds.R[pr.robot][6][t]};
        A.d0 = rs.dims[0]; A.d1 = rs.dims[1]; A.d2 = rs.dims[2];
        Bv.kind = es.kind;
        Bv.pos = ld3(es.pos);
        Bv.q = ld4(es.quat);
        Bv.d0 = es.dims[0]; Bv.d1 = es.dims[1]; Bv.d2 = es.dims[2];
        const double d = pr.s1_is_robot ? pair_distance<GJK>(pr.routine, A, Bv, sc->mesh_verts)
                                        : pair_distance<GJK>(pr.routine, Bv, A, sc->mesh_verts);
        if (d < 0.0) atomicOr(&lds.flags[t], 2u);
      }
      __syncthreads();
      p_lo = p_hi;
    }
  };
  // hyperbox bounds of point t (manip_free_workspace.hpp:79-99; lower > upper: a wrapped coordinate)
  auto out_of_bounds = [&](int t) {
    bool oob = false;
#pragma unroll
    for (int d = 0; d < N; ++d) {
      const double pt = lds.pts[d][t], lo = qs.lower[d], hi = qs.upper[d];
      if (lo < hi) oob = oob || (pt < lo) || (pt > hi);
      else oob = oob || (pt > lo) || (pt < hi);
    }
    return oob;
  };

  const double a_d = tid < N ? lds.a[tid] : 0.0, b_d = tid < N ? lds.b[tid] : 0.0;
  if (mode == EDGE_POINT) {
    // is_free(target): hyperbox bounds, then proximity (manip_free_workspace.hpp:79-99,154-156)
    if (tid < N) lds.pts[tid][0] = b_d;
    if (tid == 0) lds.q_cnt = 0u;
    __syncthreads();
    if (tid < G) lds.flags[tid] = (tid == 0 && !out_of_bounds(0)) ? 5u : 0u;
    __syncthreads();
    test_points(1);
    if (tid < N) RKH_IO(x_out)[uint64_t(e) * N + tid] = b_d;
    if (tid == 0) {
      RKH_IO(steps_free)[e] = 1;
      RKH_IO(accept)[e] = (lds.flags[0] == 5u) ? 1 : 0;
    }
    return;
  }
  const double dist_tot = norm_n(a_d - b_d);
  const double fraction = RKH_IO(frac) ? RKH_IO(frac)[e] : qs.fraction;
  double result = a_d;       // component tid of the returned point
  uint32_t n_checked = 0;
  bool completed_walk = false;  // the predicate loop ran to dist_inter without a collision (EDGE_STEER_BOTH)
  if (dist_tot == INFINITY) {
    result = a_d;
  } else if (dist_tot < qs.min_interval) {
    result = a_d + (b_d - a_d) * fraction;  // vector_topology::move_position_toward, no predicate call
  } else {
    const double dist_inter = dist_tot * fraction;
    double cur = 0.0;          // dist_cur of the last tested point (0 + min_interval == min_interval exactly)
    double last_result = a_d;  // last point that passed the predicate
    bool collided = false;
    for (;;) {
      // dist_cur of this thread's point and of the pass's last one: cur + min_interval, repeatedly (":157 dist_cur += min_interval")
      double my = cur, last = cur;
      const int t_mine = tid < G ? tid : 0;
#pragma unroll 8
      for (int t = 0; t < G; ++t) {
        if (t <= t_mine) my = my + qs.min_interval;
        last = last + qs.min_interval;
      }
      if (!((cur + qs.min_interval) < dist_inter)) break;  // not even the first point is on the edge (block-uniform)
      if (tid < G) {
        const bool valid = my < dist_inter;
        const double f = my / dist_tot;
#pragma unroll
        for (int d = 0; d < N; ++d) lds.pts[d][tid] = lds.a[d] + (lds.b[d] - lds.a[d]) * f;
        lds.flags[tid] = valid ? (out_of_bounds(tid) ? 4u : 5u) : 0u;
      }
      if (tid == 0) lds.q_cnt = 0u;
      __syncthreads();
      test_points(G);
      if (tid < 64) {  // the first point on the edge that fails the predicate, and how many points are on the edge
        const uint32_t fl = tid < G ? lds.flags[tid] : 0u;
        const unsigned long long mv = __ballot((fl & 4u) != 0u), mf = __ballot(fl == 5u);
        const unsigned long long bad = mv & ~mf;
        if (tid == 0) {
          lds.scan[0] = bad ? uint32_t(__builtin_ctzll(bad)) : uint32_t(G);
          lds.scan[1] = uint32_t(__builtin_popcountll(mv));
        }
      }
      __syncthreads();
      const int first_bad = int(lds.scan[0]), n_valid = int(lds.scan[1]);
      if (first_bad < G) {
        n_checked += uint32_t(first_bad + 1);
        if (first_bad > 0 && tid < N) last_result = lds.pts[tid][first_bad - 1];
        collided = true;
        break;
      }
      n_checked += uint32_t(n_valid);
      if (tid < N) last_result = lds.pts[tid][n_valid - 1];
      if (n_valid < G) break;  // reached dist_inter without a collision
      cur = last;
      __syncthreads();         // the points and verdicts are rewritten by the next pass
    }
    completed_walk = !collided;
    if (collided) result = last_result;
    else if (fraction == 1.0) result = b_d;  // exact end fractions (:159-162)
    else if (fraction == 0.0) result = a_d;
    else result = a_d + (b_d - a_d) * fraction;
  }
  if (tid < N) RKH_IO(x_out)[uint64_t(e) * N + tid] = result;
  if (tid == 0) RKH_IO(steps_free)[e] = n_checked;
  if (mode != EDGE_PLAIN) {
    const double n_ar = norm_n(a_d - result);
    const double n_ab = dist_tot;
    const double n_rb = norm_n(result - b_d);
    const double steer_tol = RKH_IO(steer_tol);
    if (mode == EDGE_STEER_ACCEPT || mode == EDGE_STEER_BOTH) {
      // planning_visitor_base::steer_towards_position (planning_visitors.hpp:349-360)
      const double best_case = RKH_IO(best_case) ? RKH_IO(best_case)[e] : n_ab;
      const bool ok = (!isinf(n_ar)) && (n_ar < 2.0 * best_case) && (n_ar > steer_tol * best_case);
      // EDGE_STEER_BOTH: bit 1 = the walk completed (see edge_check_kernel)
      if (tid == 0) RKH_IO(accept)[e] = (ok ? 1 : 0) | ((mode == EDGE_STEER_BOTH && completed_walk) ? 2 : 0);
    } else if (mode == EDGE_GOAL_PROBE) {
      // interp_topo_get_distance_pred (interpolated_topologies.hpp:193-199)
      if (tid == 0) {
        const uint32_t* src_idx = RKH_IO(src_idx);
        const uint32_t* d_src_first = RKH_IO(d_src_first);
        const uint32_t si = src_idx ? src_idx[e] : ((d_src_first ? *d_src_first : 0u) + e);
        RKH_IO(goal_dist)[si - 1] = (n_rb < DBL_EPSILON) ? n_ab : INFINITY;
      }
    } else if (mode == EDGE_CONNECT) {
      // planning_visitor_base::can_be_connected (planning_visitors.hpp:385-395); steer_tol carries the connection tolerance
      const bool ok = (!isinf(n_ar)) && (n_rb < steer_tol * n_ar);
      if (tid == 0) RKH_IO(accept)[e] = ok ? 1 : 0;
    } else if (mode == EDGE_WALK_ACCEPT) {
      // planning_visitor_base::random_walk (planning_visitors.hpp:418-421)
      const bool ok = (!isinf(n_ar)) && (n_ar > steer_tol * RKH_IO(best_case)[e]);
      if (tid == 0) RKH_IO(accept)[e] = ok ? 1 : 0;
    }
  }
#undef RKH_IO
}

// Diagnostic kernel (not on the product path): `iters` back-to-back f-evals + proximity tests of one edge per
// wave, with s_memtime deltas per phase: [sincos, forward sweep, jacobian columns, force sweep, mass matrix,
// cholesky, proximity, total].
template <int N>
__global__ __launch_bounds__(64) void feval_cycles_kernel(const SceneDev* __restrict__ sc, const PairDev* __restrict__ pairs,
                                                           int n_pairs, const double* __restrict__ x,
                                                           const double* __restrict__ u, int iters,
                                                           unsigned long long* __restrict__ out, double* __restrict__ sink_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  BlockLds<N, 64>& lds = *reinterpret_cast<BlockLds<N, 64>*>(smem_raw);
  ShapeDev* env_lds = reinterpret_cast<ShapeDev*>(smem_raw + SmemLayout<N, 64>::block_bytes);
  const int lane = threadIdx.x;
  constexpr int D = 2 * N;
  stage_chain<N>(sc, lds.joints, lds.base, lane);
  stage_env(sc, env_lds, lane);
  GroupWs<N>& ws = lds.g[0];
  double xv = (lane < D) ? x[uint64_t(blockIdx.x) * D + lane] : 0.0;
  if (lane < N) ws.u[lane] = u[uint64_t(blockIdx.x) * N + lane];
  __syncthreads();
  const CPack<N> cp = load_cpack<N>(lds.joints, lane);
  unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  bool singular = false;
  double acc = 0.0;
  const unsigned long long t_begin = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (lane < D) ws.x[lane] = xv;
    __syncthreads();
    const double dp = state_derivative<N, 64>(sc, cp, lds.joints, lds.base, ws, lds.sink[lane], lane, 0, &singular, st);
    xv = xv + 1e-4 * dp;
    if (lane < D) ws.x[lane] = xv;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    acc += proximity_min<N, 64>(sc, cp, lds.base, env_lds, pairs, n_pairs, ws, lds.sink[lane], lane, 0, true, false);
    st[6] += __builtin_readcyclecounter() - t0;
  }
  st[7] = __builtin_readcyclecounter() - t_begin;
  if (lane == 0) {
    for (int i = 0; i < 8; ++i) out[blockIdx.x * 8 + i] = st[i];
    sink_out[blockIdx.x] = acc + xv + (singular ? 1.0 : 0.0);
  }
}

// Diagnostic kernel (not on the product path): state_derivative_duo, `iters` f-evals of one state per block of two
// waves; row 2 b of `out` = wave 0's stamps, row 2 b + 1 = wave 1's (8 counters each, see state_derivative_duo).
template <int N>
__global__ __launch_bounds__(128) void feval_cycles_duo_kernel(const SceneDev* __restrict__ sc, const double* __restrict__ x,
                                                                const double* __restrict__ u, int iters,
                                                                unsigned long long* __restrict__ out, double* __restrict__ sink_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  BlockLds<N, 64>& lds = *reinterpret_cast<BlockLds<N, 64>*>(smem_raw);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int D = 2 * N;
  stage_chain<N>(sc, lds.joints, lds.base, lane);
  GroupWs<N>& ws = lds.g[0];
  double xv = (lane < D) ? x[uint64_t(blockIdx.x) * D + lane] : 0.0;
  if (lane < N) ws.u[lane] = u[uint64_t(blockIdx.x) * N + lane];
  __syncthreads();
  const CPack<N> cp = load_cpack<N>(lds.joints, lane);
  unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  bool singular = false;
  const unsigned long long t_begin = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (lane < D) ws.x[lane] = xv;
    __syncthreads();
    const double dp = state_derivative_duo<N>(sc, cp, lds.joints, lds.base, ws, lds.sink[lane], lane, wave, &singular, st);
    xv = xv + 1e-4 * dp;
  }
  st[7] = __builtin_readcyclecounter() - t_begin;
  if (lane == 0) {
    for (int i = 0; i < 8; ++i) out[(uint64_t(blockIdx.x) * 2 + wave) * 8 + i] = st[i];
    if (wave == 0) sink_out[blockIdx.x] = xv + (singular ? 1.0 : 0.0);
  }
}

// Kernel: exact minimum proxy-pair distance for B states (no culling).
template <int N>
__global__ __launch_bounds__(64) void min_distance_kernel(const SceneDev* __restrict__ sc, const PairDev* __restrict__ pairs,
                                                           int n_pairs, const double* __restrict__ x, uint32_t B,
                                                           double* __restrict__ dist) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  BlockLdsQs<N, 64>& lds = *reinterpret_cast<BlockLdsQs<N, 64>*>(smem_raw);
  ShapeDev* env_lds = reinterpret_cast<ShapeDev*>(smem_raw + SmemLayoutQs<N, 64>::block_bytes);
  const uint32_t e = blockIdx.x;
  if (e >= B) return;
  const int lane = threadIdx.x;
  constexpr int D = 2 * N;
  stage_chain<N>(sc, lds.joints, lds.base, lane);
  stage_env(sc, env_lds, lane);
  GroupWsQs<N>& ws = lds.g[0];
  if (lane < D) ws.x[lane] = x[uint64_t(e) * D + lane];
  __syncthreads();
  const CPack<N> cp = load_cpack<N>(lds.joints, lane);
  const double dmin = proximity_min<N, 64>(sc, cp, lds.base, env_lds, pairs, n_pairs, ws, lds.sink[lane], lane, 0, false, false);
  if (lane == 0) dist[e] = dmin;
}

// ---- host launchers ------------------------------------------------------------------------
#define RKH_DISPATCH_N(N_, CALL)     \
  switch (N_) {                      \
    case 1: { constexpr int N = 1; CALL; } break; \
    case 2: { constexpr int N = 2; CALL; } break; \
    case 3: { constexpr int N = 3; CALL; } break; \
    case 4: { constexpr int N = 4; CALL; } break; \
    case 6: { constexpr int N = 6; CALL; } break; \
    case 7: { constexpr int N = 7; CALL; } break; \
    case 12: { constexpr int N = 12; CALL; } break; \
    default:                         \
      set_error("propagate: chains with this number of joints are not instantiated (1,2,3,4,6,7,12)"); \
      return RKH_ERR_UNSUPPORTED;    \
  }

#define RKH_DISPATCH_N_QS(N_, CALL)  \
  switch (N_) {                      \
    case 1: { constexpr int N = 1; CALL; } break; \
    case 2: { constexpr int N = 2; CALL; } break; \
    case 3: { constexpr int N = 3; CALL; } break; \
    case 4: { constexpr int N = 4; CALL; } break; \
    case 6: { constexpr int N = 6; CALL; } break; \
    case 7: { constexpr int N = 7; CALL; } break; \
    case 12: { constexpr int N = 12; CALL; } break; \
    default:                         \
      set_error("quasi-static kernels: chains with this number of joints are not instantiated (1,2,3,4,6,7,12)"); \
      return RKH_ERR_UNSUPPORTED;    \
  }

template <int N, int GL, bool GJK, bool DUO = false>
static void launch_propagate_t(hipStream_t s, int n_env, const SceneDev* d_scene, const PairDev* d_pairs, int n_pairs,
                               const DynDev& dyn, const EdgeIO& io, uint32_t edges_a, const EdgeIO& io_b,
                               uint32_t edges_b, const EdgeIO* tab_a, const EdgeIO* tab_b, uint32_t n_problems,
                               KernelGate gate) {
  constexpr uint32_t G = 64 / GL;
  const uint32_t ga = (edges_a + G - 1) / G, gbk = (edges_b + G - 1) / G;
  dim3 grid(ga + gbk, n_problems);
  if (gate.wave_base) {
    // compact mapping (G = 1): the kernel only runs while the round has fewer than gate.hi edges in total, so that many
    // blocks are enough -- instead of (bound per problem) x problems blocks that find nothing when the gate is closed
    const uint64_t all = uint64_t(ga + gbk) * n_problems;
    grid = dim3(uint32_t(std::min<uint64_t>(all, gate.hi)), 1);
  }
  WaveArgs args;
  args.sc = d_scene;
  args.pairs = d_pairs;
  args.n_pairs = n_pairs;
  args.dyn = dyn;
  args.io_a = io;
  args.io_b = io_b;
  args.tab_a = tab_a;
  args.tab_b = tab_b;
  args.grid_a = ga;
  args.gate = gate;
  hipLaunchKernelGGL((propagate_kernel<N, GL, GJK, DUO>), grid, dim3(DUO ? 128 : 64), (SmemLayout<N, GL>::bytes(n_env)), s, args);
}

// Steer `grid_edges` (+ `grid_b` of a second group) edges per problem.  Either the two EdgeIO are given by value
// (n_problems = 1) or as device tables of n_problems entries each.
rkh_status launch_propagate(hipStream_t s, int n_dof, int n_env, const SceneDev* d_scene, const void* d_pairs,
                            int n_pairs, const DynDev& dyn, const EdgeIO& io, uint32_t grid_edges, const EdgeIO* io_b,
                            uint32_t grid_b, int lanes_per_edge, const EdgeIO* tab_a, const EdgeIO* tab_b,
                            uint32_t n_problems, double* d_lane_ws, KernelGate gate) {
  const uint32_t eb = (io_b || tab_b) ? grid_b : 0u;
  if (grid_edges + eb == 0 || n_problems == 0) return RKH_OK;
  if (is_planar_scene(d_scene))  // planar chains: one lane per edge, whatever mapping was asked for (propagate_planar.hip)
    return launch_propagate_planar(s, n_dof, d_scene, d_pairs, n_pairs, dyn, io, grid_edges, io_b, grid_b, tab_a, tab_b,
                                   n_problems, gate);
  if (lanes_per_edge == 1)  // two lanes per edge, first generation (propagate_lane.hip)
    return launch_propagate_lanes(s, n_dof, d_scene, dyn, io, grid_edges, io_b, grid_b, tab_a, tab_b, n_problems, d_lane_ws,
                                  gate);
  if (lanes_per_edge == 2)  // two lanes per edge, two waves per SIMD (propagate_pair.hip)
    return launch_propagate_pairs(s, n_dof, d_scene, dyn, io, grid_edges, io_b, grid_b, tab_a, tab_b, n_problems, d_lane_ws,
                                  gate);
  const EdgeIO second = io_b ? *io_b : EdgeIO();
  const PairDev* pp = static_cast<const PairDev*>(d_pairs);
  if (lanes_per_edge == 128 && !is_mesh_scene(d_scene)) {  // two waves per edge (scenes without vertex-set shapes)
    RKH_DISPATCH_N(n_dof, (launch_propagate_t<N, 64, false, true>(s, n_env, d_scene, pp, n_pairs, dyn, io, grid_edges, second, eb,
                                                                  tab_a, tab_b, n_problems, gate)));
  } else if (lanes_per_edge == 16) {
    RKH_DISPATCH_N(n_dof, (launch_propagate_t<N, 16, true>(s, n_env, d_scene, pp, n_pairs, dyn, io, grid_edges, second, eb, tab_a,
                                                           tab_b, n_problems, gate)));
  } else if (is_mesh_scene(d_scene)) {
    RKH_DISPATCH_N(n_dof, (launch_propagate_t<N, 64, true>(s, n_env, d_scene, pp, n_pairs, dyn, io, grid_edges, second, eb, tab_a,
                                                           tab_b, n_problems, gate)));
  } else {  // no vertex-set shapes: the instantiation without the support-map query (no private segment)
    RKH_DISPATCH_N(n_dof, (launch_propagate_t<N, 64, false>(s, n_env, d_scene, pp, n_pairs, dyn, io, grid_edges, second, eb, tab_a,
                                                            tab_b, n_problems, gate)));
  }
  RKH_HIP(hipGetLastError());
  return RKH_OK;
}

rkh_status launch_state_derivative(hipStream_t s, int n_dof, const SceneDev* d_scene, const double* d_x,
                                   const double* d_u, uint32_t B, double* d_pd, double* d_M, double* d_f, int* d_err) {
  if (B == 0) return RKH_OK;
  if (is_planar_scene(d_scene)) return launch_state_derivative_planar(s, n_dof, d_scene, d_x, d_u, B, d_pd, d_M, d_f, d_err);
  RKH_DISPATCH_N(n_dof, hipLaunchKernelGGL((state_derivative_kernel<N>), dim3(B), dim3(64), 0, s, d_scene, d_x, d_u, B,
                                           d_pd, d_M, d_f, d_err));
  RKH_HIP(hipGetLastError());
  return RKH_OK;
}

static std::unordered_set<const void*>& mesh_scenes() {
  static std::unordered_set<const void*> s;
  return s;
}
void register_mesh_scene(const SceneDev* d_scene) { mesh_scenes().insert(d_scene); }
void forget_mesh_scene(const SceneDev* d_scene) { mesh_scenes().erase(d_scene); }
bool is_mesh_scene(const SceneDev* d_scene) { return mesh_scenes().count(d_scene) != 0; }

// The shape of an edge_check_kernel launch, by the number of edges and by what fits 64 KB of LDS -- 0: GL 16 x 1 wave,
// 1: GL 16 x 4 waves, 2: GL 16 x 2 waves, 3: GL 16 x 8 waves -- and whether the pair list rides in LDS too.
template <int N>
static void edge_check_shape(int n_env, int n_pairs, uint64_t n_edges, int* shape, int* staged) {
  auto fit = [&](size_t with_pairs, size_t without) { return with_pairs <= 65536 ? 2 : (without <= 65536 ? 1 : 0); };
  const int f1 = fit(SmemLayoutQsW<N, 16, 1>::bytes(n_env, n_pairs), SmemLayoutQsW<N, 16, 1>::bytes(n_env, 0));
  const int f2 = fit(SmemLayoutQsW<N, 16, 2>::bytes(n_env, n_pairs), SmemLayoutQsW<N, 16, 2>::bytes(n_env, 0));
  const int f4 = fit(SmemLayoutQsW<N, 16, 4>::bytes(n_env, n_pairs), SmemLayoutQsW<N, 16, 4>::bytes(n_env, 0));
  const int f8 = fit(SmemLayoutQsW<N, 16, 8>::bytes(n_env, n_pairs), SmemLayoutQsW<N, 16, 8>::bytes(n_env, 0));
  *shape = 0;
  *staged = f1 == 2;
  if (n_edges < 4096) {
    if (f4) *shape = 1, *staged = f4 == 2;
    else if (f2) *shape = 2, *staged = f2 == 2;
  }
  if (n_edges < 512 && f8) *shape = 3, *staged = f8 == 2;
}

template <int N, bool GJK, int G>
static rkh_status launch_edge_points_t(hipStream_t s, dim3 grid, size_t smem, const SceneDev* d_scene, const PairDev* pp,
                                       int n_pairs, const QsDev& qs, const EdgeIO& io, const EdgeIO& second,
                                       const EdgeIO* tab_a, const EdgeIO* tab_b, uint32_t grid_a, int staged) {
  auto kern = edge_points_kernel<N, GJK, G>;
  EdgePointsArgs ka;
  ka.sc = d_scene;
  ka.pairs = pp;
  ka.n_pairs = n_pairs;
  ka.qs = qs;
  ka.io_a = io;
  ka.io_b = second;
  ka.tab_a = tab_a;
  ka.tab_b = tab_b;
  ka.grid_a = grid_a;
  ka.pairs_staged = staged;
  static bool big_lds = false;  // (per instantiation) more than the default 64 KB of dynamic LDS: ask once
  if (smem > 65536 && !big_lds) {
    RKH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    big_lds = true;
  }
  hipLaunchKernelGGL(kern, grid, dim3(256), smem, s, ka);
  return RKH_OK;
}

rkh_status launch_edge_check(hipStream_t s, int n_dof, int n_env, const SceneDev* d_scene, const void* d_pairs,
                             int n_pairs, const QsDev& qs, const EdgeIO& io, uint32_t grid_edges, const EdgeIO* io_b,
                             uint32_t grid_b, const EdgeIO* tab_a, const EdgeIO* tab_b, uint32_t n_problems) {
  const uint32_t eb = (io_b || tab_b) ? grid_b : 0u;
  if (grid_edges + eb == 0 || n_problems == 0) return RKH_OK;
  const EdgeIO second = io_b ? *io_b : EdgeIO();
  const PairDev* pp = static_cast<const PairDev*>(d_pairs);
  const bool gjk = is_mesh_scene(d_scene);  // scenes without vertex-set shapes run the instantiations without GJK (no scratch)
  static const bool by_points = [] { const char* e = getenv("RKH_EDGE_POINTS"); return !e || atoi(e) != 0; }();
  if (by_points && !is_planar_scene(d_scene)) {  // 3D scenes: one lane per point in the chain kinematics
    const bool narrow = n_dof > 7 || uint64_t(grid_edges + eb) * n_problems >= 2048;  // 32 points per pass
    size_t with_pairs = 0, without = 0;
    if (narrow) {
      RKH_DISPATCH_N_QS(n_dof, (with_pairs = EdgePointsSmem<N, 32>::bytes(n_env, n_pairs), without = EdgePointsSmem<N, 32>::bytes(n_env, 0)));
    } else {
      RKH_DISPATCH_N_QS(n_dof, (with_pairs = EdgePointsSmem<N, 64>::bytes(n_env, n_pairs), without = EdgePointsSmem<N, 64>::bytes(n_env, 0)));
    }
    const size_t lds_max = 160 * 1024;
    if (without <= lds_max) {
      const int staged = with_pairs <= lds_max;
      const size_t smem = staged ? with_pairs : without;
      rkh_status st = RKH_OK;
      const dim3 grid(grid_edges + eb, n_problems);
#define RKH_POINTS_ARGS s, grid, smem, d_scene, pp, n_pairs, qs, io, second, tab_a, tab_b, grid_edges, staged
      if (narrow) {
        if (gjk) {
          RKH_DISPATCH_N_QS(n_dof, (st = launch_edge_points_t<N, true, 32>(RKH_POINTS_ARGS)));
        } else {
          RKH_DISPATCH_N_QS(n_dof, (st = launch_edge_points_t<N, false, 32>(RKH_POINTS_ARGS)));
        }
      } else {
        if (gjk) {
          RKH_DISPATCH_N_QS(n_dof, (st = launch_edge_points_t<N, true, (N <= 7 ? 64 : 32)>(RKH_POINTS_ARGS)));
        } else {
          RKH_DISPATCH_N_QS(n_dof, (st = launch_edge_points_t<N, false, (N <= 7 ? 64 : 32)>(RKH_POINTS_ARGS)));
        }
      }
#undef RKH_POINTS_ARGS
      if (st != RKH_OK) return st;
      RKH_HIP(hipGetLastError());
      return RKH_OK;
    }
  }
  // the shape of the launch (see edge_check_kernel): by the number of edges, and by what fits 64 KB of LDS
  const uint64_t n_edges = uint64_t(grid_edges + eb) * n_problems;
  int shape = 0, staged = 0;
  RKH_DISPATCH_N_QS(n_dof, (edge_check_shape<N>(n_env, n_pairs, n_edges, &shape, &staged)));
#define RKH_EDGE_LAUNCH(GL_, W_, GJK_)                                                                                          \
  RKH_DISPATCH_N_QS(n_dof, hipLaunchKernelGGL((edge_check_kernel<N, GL_, W_, GJK_>), dim3(grid_edges + eb, n_problems),   \
                                           dim3(64 * W_), (SmemLayoutQsW<N, GL_, W_>::bytes(n_env, staged ? n_pairs : 0)), s, \
                                           d_scene, pp, n_pairs, qs, io, second, tab_a, tab_b, grid_edges, staged))
  // (planar scenes, and 3D ones with RKH_EDGE_POINTS=0: one instantiation per shape, support-map query included)
  switch (shape) {
    case 1: RKH_EDGE_LAUNCH(16, 4, true); break;
    case 2: RKH_EDGE_LAUNCH(16, 2, true); break;
    case 3: RKH_EDGE_LAUNCH(16, 8, true); break;
    default: RKH_EDGE_LAUNCH(16, 1, true); break;
  }
#undef RKH_EDGE_LAUNCH
  RKH_HIP(hipGetLastError());
  return RKH_OK;
}

rkh_status launch_feval_cycles_duo(hipStream_t s, int n_dof, int n_env, const SceneDev* d_scene, const double* d_x,
                                   const double* d_u, uint32_t B, int iters, unsigned long long* d_out, double* d_sink) {
  RKH_DISPATCH_N(n_dof, hipLaunchKernelGGL((feval_cycles_duo_kernel<N>), dim3(B / 2), dim3(128), (SmemLayout<N, 64>::bytes(n_env)),
                                           s, d_scene, d_x, d_u, iters, d_out, d_sink));
  RKH_HIP(hipGetLastError());
  return RKH_OK;
}

rkh_status launch_feval_cycles(hipStream_t s, int n_dof, int n_env, const SceneDev* d_scene, const void* d_pairs,
                               int n_pairs, const double* d_x, const double* d_u, uint32_t B, int iters,
                               unsigned long long* d_out, double* d_sink) {
  RKH_DISPATCH_N(n_dof, hipLaunchKernelGGL((feval_cycles_kernel<N>), dim3(B), dim3(64), (SmemLayout<N, 64>::bytes(n_env)),
                                           s, d_scene, static_cast<const PairDev*>(d_pairs), n_pairs, d_x, d_u, iters, d_out,
                                           d_sink));
  RKH_HIP(hipGetLastError());
  return RKH_OK;
}

rkh_status launch_min_distance(hipStream_t s, int n_dof, int n_env, const SceneDev* d_scene, const void* d_pairs, int n_pairs,
                               const double* d_x, uint32_t B, double* d_dist) {
  if (B == 0) return RKH_OK;
  RKH_DISPATCH_N_QS(n_dof, hipLaunchKernelGGL((min_distance_kernel<N>), dim3(B), dim3(64), (SmemLayoutQs<N, 64>::bytes(n_env)), s,
                                           d_scene, static_cast<const PairDev*>(d_pairs), n_pairs, d_x, B, d_dist));
  RKH_HIP(hipGetLastError());
  return RKH_OK;
}

}  // namespace rkh
