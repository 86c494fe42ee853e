// propagate_lane.hip -- the throughput mapping of the steer kernel: ONE LANE PER CANDIDATE EDGE (gfx950, wave64).
//
// Same function as propagate_kernel (propagate.hip; reference citations there): RK4 forward dynamics of the KTE
// serial chain under the held PD input, proximity test after every step, accept test / goal probe at the end.
// propagate_kernel gives a whole wavefront to one edge (lowest latency, but its serial base->tip / tip->base sweeps
// run redundantly on all 64 lanes); here every lane integrates its own edge with plain serial code, so no lane
// repeats another's work and nothing crosses lanes.  The planner uses it when a round offers enough candidate edges
// to fill the chip at 64 edges per wave.
//
// What shapes the kernel is storage and code size, not arithmetic:
//   * VALU instructions address 256 architectural VGPRs = 128 doubles per lane; an unrolled f-eval of a 6-joint chain
//     wants ~200 live doubles and ~270 KB of code (measured: 33 k instructions, 4x the 64 KB instruction cache, and it
//     ran no faster than the wave-per-edge kernel).  So every loop over joints / Jacobian columns / matrix entries is
//     ROLLED (one copy of each body in the instruction stream, ~25 KB in total, one copy of sincos) and the per-lane
//     arrays they index at run time live in memory laid out [slot][lane] (conflict-free, coalesced):
//       LDS    (78 slots x 512 B = 39 KB per wave = the CU's 160 KB at one wave per SIMD): the current body's Jacobian
//              columns T[6N] and the mass matrix Mf[N*N] (read-modify-written once per body), the held input u[N];
//       global workspace (29N slots per wave, L2 resident): state / target / RK4 stage vectors, sin/cos table, joint end
//              frames (written once, read by later bodies' Jacobian columns, loads issued one column ahead) and the
//              d'Alembert forces of the links (written by the base->tip sweep, read by the tip->base sweep);
//       registers: the sweep recurrences and the Cholesky factor only.
//   * the mass matrix is accumulated body by body while the forward sweep runs (Mf(i,j) receives its terms in the same
//     ascending-body order as the reference's Tcm^T (Mcm Tcm) product), so the 6N x N Jacobian is never stored;
//   * chain parameters are wave-uniform: scalar loads with a uniform run-time index, SGPR operands;
//   * proximity: per robot shape, a uniform loop over the obstacles does the bounding-sphere cull and sets one bit
//     per surviving obstacle in a per-lane 64-bit mask; survivors are then evaluated kind by kind (sphere / box /
//     capped cylinder), so that the lanes of a wave run the same closed form.
// Every product and sum is formed in the reference's order (-ffp-contract=off), exactly as in propagate_kernel:
// both kernels return bit-identical states and verdicts (tests/test_gpu_parity.py::test_propagate_mappings_*).
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cmath>

#include "device_math.h"
#include "proximity_device.h"
#include "rkh_internal.h"

namespace rkh {

namespace {

RKH_DI d3 ldg3(const double* p) { return d3{p[0], p[1], p[2]}; }
RKH_DI d4 ldg4(const double* p) { return d4{p[0], p[1], p[2], p[3]}; }
RKH_DI m33 ldgm(const double* p) { return m33{p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], p[8]}; }

RKH_DI m33 lane_axis_angle_rotmat(double ca, double sa, d3 ax) {  // axis_angle::getRotMat (rotations_3D.hpp:2160-2180)
  const double omc = 1.0 - ca;
  const double t11 = ca + omc * ax.x * ax.x, t22 = ca + omc * ax.y * ax.y, t33 = ca + omc * ax.z * ax.z;
  const double t12 = omc * ax.x * ax.y, t13 = omc * ax.x * ax.z, t23 = omc * ax.y * ax.z;
  const double t01 = sa * ax.x, t02 = sa * ax.y, t03 = sa * ax.z;
  return m33{t11, t12 - t03, t13 + t02, t12 + t03, t22, t23 - t01, t13 - t02, t23 + t01, t33};
}

// per-lane arrays in LDS, [slot][lane]
template <int N>
struct LaneLds {
  double T[N * 6][64];   // Jacobian columns (v, w) of the current body; afterwards the generalized forces f[N]
  double Mf[N * N][64];  // Tcm^T (Mcm Tcm) before symmetrisation
  double u[N][64];       // held input
};

// global workspace of one wave, [slot][lane]; `ws` below already points at the lane's column
template <int N>
struct WsLayout {
  enum : int {
    X = 0,            // last free state
    B = 2 * N,        // steer target
    W = 4 * N,        // RK4: state at the start of the inner step
    KA = 6 * N,       // RK4: k1, then (1/6) k1 + (2/6) k2
    K3 = 8 * N,       // RK4: k3
    XE = 10 * N,      // state being differentiated / tested
    TRIG = 12 * N,    // per joint: cos, sin of the half angle, cos, sin of the full angle
    ECP = 16 * N,     // joint end frames: position
    ECQ = 19 * N,     //                   quaternion
    FT = 23 * N,      // inertia_3D d'Alembert force / torque per link
    SLOTS = 29 * N
  };
};
#define RKH_WS(slot) ws[(slot) * 64]

// x' = f(x, u) of one edge.  In: the state in WS(XE..), the input in lds.u.  Out: qdd[j] (the q components of x' are
// the qd components of x).
template <int N>
__device__ __forceinline__ void lane_state_derivative(const SceneDev* __restrict__ sc, LaneLds<N>& lds, int lane,
                                                      double* __restrict__ ws, double (&qdd)[N], bool& singular) {
  typedef WsLayout<N> L_;
  // ---- sin/cos of the half and full joint angles (one copy of sincos in the instruction stream)
#pragma unroll 1
  for (int t = 0; t < 2 * N; ++t) {
    const double q = RKH_WS(L_::XE + 2 * (t >> 1));
    double sn, cs;
    sincos((t & 1) ? q : 0.5 * q, &sn, &cs);
    RKH_WS(L_::TRIG + 2 * t) = cs;
    RKH_WS(L_::TRIG + 2 * t + 1) = sn;
  }
#pragma unroll 1
  for (int e = 0; e < N * N; ++e) {
    const int i = e / N, jx = e - i * N;
    lds.Mf[e][lane] = (i == jx) ? (0.0 + sc->joints[i].joint_inertia) : 0.0;  // inertia_gen rows: Tcm = 1
  }

  // ---- base -> tip sweep (kte_map_chain::doMotion) with the Jacobian columns and M terms of each body
  d3 pos = ldg3(sc->base_pos);
  d4 Q = ldg4(sc->base_quat);
  d3 w = mk3(0, 0, 0), alpha = mk3(0, 0, 0);
  d3 acc = ldg3(sc->base_acc);
#pragma unroll 1
  for (int j = 0; j < N; ++j) {
    const JointDev& J = sc->joints[j];
    const d3 axis = ldg3(J.axis), axis_n = ldg3(J.axis_n);
    const double c2 = RKH_WS(L_::TRIG + 4 * j), s2 = RKH_WS(L_::TRIG + 4 * j + 1);
    const double qd = RKH_WS(L_::XE + 2 * j + 1);
    // revolute_joint_3D::doMotion (revolute_joint.cpp:121-148)
    const d4 tq = d4{c2, axis_n.x * s2, axis_n.y * s2, axis_n.z * s2};
    const m33 R2 = rotmat(tq);
    const d4 EQ = qmul(Q, tq);
    const d3 wb = mulT(w, R2);
    const d3 qa = qd * axis;
    const d3 Ew = wb + qa;
    const d3 Ealpha = mulT(alpha, R2) + cross(wb, qa);
    const d3 Epos = pos;
    RKH_WS(L_::ECP + 3 * j) = pos.x; RKH_WS(L_::ECP + 3 * j + 1) = pos.y; RKH_WS(L_::ECP + 3 * j + 2) = pos.z;
    RKH_WS(L_::ECQ + 4 * j) = EQ.w; RKH_WS(L_::ECQ + 4 * j + 1) = EQ.x;
    RKH_WS(L_::ECQ + 4 * j + 2) = EQ.y; RKH_WS(L_::ECQ + 4 * j + 3) = EQ.z;
    // rigid_link_3D::doMotion = frame * pose (frame_3D.hpp:240-255)
    const d3 op = ldg3(J.off_pos);
    const m33 R = rotmat(EQ);
    pos = pos + mul(R, op);
    acc = acc + mul(R, cross(Ew, cross(Ew, op)) + cross(Ealpha, op));
    const m33 Ro = ldgm(J.off_R);
    Q = qmul(EQ, ldg4(J.off_quat));
    alpha = mulT(Ealpha, Ro);
    w = mulT(Ew, Ro);
    // inertia_3D::doForce terms (inertia.cpp:111-122), applied in the backward sweep
    const d3 Fi = J.mass * qrot(qinv(Q), acc);
    const d3 Ti = sym_mul(J.inertia, alpha) + cross(w, sym_mul(J.inertia, w));
    RKH_WS(L_::FT + 6 * j) = Fi.x; RKH_WS(L_::FT + 6 * j + 1) = Fi.y; RKH_WS(L_::FT + 6 * j + 2) = Fi.z;
    RKH_WS(L_::FT + 6 * j + 3) = Ti.x; RKH_WS(L_::FT + 6 * j + 4) = Ti.y; RKH_WS(L_::FT + 6 * j + 5) = Ti.z;
    // Jacobian columns of body j w.r.t. coords c = j, j-1, .., 0: get_jac_relative_to (motion_jacobians.hpp:238-251)
    // with f2 = (~F_c) * F_b (frame_3D.hpp:184-189,222-238,368-382).  The parent frame of the next column is loaded
    // while the current one is computed.
    d3 cp = Epos;
    d4 cq = EQ;
#pragma unroll 1
    for (int c = j; c >= 0; --c) {
      d3 ncp = cp;
      d4 ncq = cq;
      if (c > 0) {
        ncp = mk3(RKH_WS(L_::ECP + 3 * (c - 1)), RKH_WS(L_::ECP + 3 * (c - 1) + 1), RKH_WS(L_::ECP + 3 * (c - 1) + 2));
        ncq = d4{RKH_WS(L_::ECQ + 4 * (c - 1)), RKH_WS(L_::ECQ + 4 * (c - 1) + 1), RKH_WS(L_::ECQ + 4 * (c - 1) + 2),
                 RKH_WS(L_::ECQ + 4 * (c - 1) + 3)};
      }
      const m33 Rc = rotmat(cq);
      const d4 iq = qinv(cq);
      const d3 ipos = mulT(-cp, Rc);
      const m33 Ri = rotmat(iq);
      const d3 f2pos = ipos + mul(Ri, pos);
      const d4 f2q = qmul(iq, Q);
      const m33 Rf = rotmat(f2q);
      const d3 ax_c = ldg3(sc->joints[c].axis);
      const d3 wt = mulT(ax_c, Rf);
      const d3 vt = mulT(cross(ax_c, f2pos), Rf);
      lds.T[c * 6 + 0][lane] = vt.x; lds.T[c * 6 + 1][lane] = vt.y; lds.T[c * 6 + 2][lane] = vt.z;
      lds.T[c * 6 + 3][lane] = wt.x; lds.T[c * 6 + 4][lane] = wt.y; lds.T[c * 6 + 5][lane] = wt.z;
      cp = ncp;
      cq = ncq;
    }
    // Mf += Tcm_b^T (Mcm_b Tcm_b): summation order of mat_alg_symmetric.hpp:551-566 and mat_operators.hpp:104-114
#pragma unroll 1
    for (int jx = 0; jx <= j; ++jx) {
      const double m0 = J.mass * lds.T[jx * 6 + 0][lane], m1 = J.mass * lds.T[jx * 6 + 1][lane],
                   m2 = J.mass * lds.T[jx * 6 + 2][lane];
      const d3 P = sym_mul(J.inertia, mk3(lds.T[jx * 6 + 3][lane], lds.T[jx * 6 + 4][lane], lds.T[jx * 6 + 5][lane]));
#pragma unroll 1
      for (int i = 0; i <= j; ++i) {
        double s = lds.Mf[i * N + jx][lane];
        s = s + lds.T[i * 6 + 0][lane] * m0;
        s = s + lds.T[i * 6 + 1][lane] * m1;
        s = s + lds.T[i * 6 + 2][lane] * m2;
        s = s + lds.T[i * 6 + 3][lane] * P.x;
        s = s + lds.T[i * 6 + 4][lane] * P.y;
        s = s + lds.T[i * 6 + 5][lane] * P.z;
        lds.Mf[i * N + jx][lane] = s;
      }
    }
  }

  // ---- tip -> base sweep (kte_map_chain::doForce in reverse op order); f[j] lands in the (now free) T slots
  {
    d3 LF = mk3(0, 0, 0), LT = mk3(0, 0, 0);
#pragma unroll 1
    for (int j = N - 1; j >= 0; --j) {
      const JointDev& J = sc->joints[j];
      const d3 axis = ldg3(J.axis);
      LF = LF - mk3(RKH_WS(L_::FT + 6 * j), RKH_WS(L_::FT + 6 * j + 1), RKH_WS(L_::FT + 6 * j + 2));  // inertia_3D::doForce
      LT = LT - mk3(RKH_WS(L_::FT + 6 * j + 3), RKH_WS(L_::FT + 6 * j + 4), RKH_WS(L_::FT + 6 * j + 5));
      const m33 Ro = ldgm(J.off_R);                 // rigid_link_3D::doForce (rigid_link.cpp:170-178)
      const d3 op = ldg3(J.off_pos);
      const d3 tmp_force = mul(Ro, LF);
      const d3 ET = mul(Ro, LT) + cross(op, tmp_force);
      const m33 Ra = lane_axis_angle_rotmat(RKH_WS(L_::TRIG + 4 * j + 2), RKH_WS(L_::TRIG + 4 * j + 3),
                                            ldg3(J.axis_n));  // revolute_joint_3D::doForce (revolute_joint.cpp:170-181)
      const double ta = dot(ET, axis);
      LF = mul(Ra, tmp_force);
      LT = mul(Ra, ET - ta * axis);
      const double uj = lds.u[j][lane];  // inertia_gen::doForce (q_ddot = 0), driving_actuator_gen::doForce
      lds.T[j][lane] = ta + uj;
      LT = LT - uj * axis;
    }
  }

  // ---- mat<symmetric>(general): 0.5 * (M(j,i) + M(i,j)), j < i (mat_alg_symmetric.hpp:183-187), lower triangle
  double L[N][N], f[N];
#pragma unroll
  for (int i = 0; i < N; ++i) {
    f[i] = lds.T[i][lane];
#pragma unroll
    for (int j = 0; j <= i; ++j)
      L[i][j] = (i == j) ? lds.Mf[i * N + i][lane] : 0.5 * (lds.Mf[j * N + i][lane] + lds.Mf[i * N + j][lane]);
  }
  // ---- linsolve_Cholesky (mat_cholesky.hpp:63-84,546-554)
#pragma unroll
  for (int j = 0; j < N; ++j) {
    double dgl = L[j][j];
#pragma unroll
    for (int k = 0; k < j; ++k) dgl = dgl - L[j][k] * L[j][k];
    if (dgl < 1e-8) singular = true;
    const double ljj = sqrt(dgl);
    L[j][j] = ljj;
#pragma unroll
    for (int i = j + 1; i < N; ++i) {
      double v = L[i][j];
#pragma unroll
      for (int k = 0; k < j; ++k) v = v - L[i][k] * L[j][k];
      L[i][j] = v / ljj;
    }
  }
  // backsub_Cholesky_impl (mat_cholesky.hpp:160-178): L y = f, then L^T x = y
#pragma unroll
  for (int k = 0; k < N; ++k) {
    const double yk = f[k] / L[k][k];
    f[k] = yk;
#pragma unroll
    for (int r = k + 1; r < N; ++r) f[r] = f[r] - L[r][k] * yk;
  }
#pragma unroll
  for (int k = N - 1; k >= 0; --k) {
    const double xk = f[k] / L[k][k];
    f[k] = xk;
#pragma unroll
    for (int r = 0; r < k; ++r) f[r] = f[r] - L[k][r] * xk;
  }
#pragma unroll
  for (int j = 0; j < N; ++j) qdd[j] = f[j];
}

// is the configuration in WS(XE..) (joint angles) collision-free?  (manip_dk_proxy_env_impl::is_free, proximity only)
template <int N>
__device__ __forceinline__ bool lane_proximity_free(const SceneDev* __restrict__ sc, int lane, double* __restrict__ ws,
                                                    bool active) {
  typedef WsLayout<N> L_;
  bool hit = !active;  // inactive lanes take no part in the scan
  {  // joint end frames: revolute_joint_3D / rigid_link_3D kinematics, position + orientation only
    d3 pos = ldg3(sc->base_pos);
    d4 Q = ldg4(sc->base_quat);
#pragma unroll 1
    for (int j = 0; j < N; ++j) {
      const JointDev& J = sc->joints[j];
      const d3 axis_n = ldg3(J.axis_n);
      double s2, c2;
      sincos(0.5 * RKH_WS(L_::XE + 2 * j), &s2, &c2);
      const d4 tq = d4{c2, axis_n.x * s2, axis_n.y * s2, axis_n.z * s2};
      const d4 EQ = qmul(Q, tq);
      RKH_WS(L_::ECP + 3 * j) = pos.x; RKH_WS(L_::ECP + 3 * j + 1) = pos.y; RKH_WS(L_::ECP + 3 * j + 2) = pos.z;
      RKH_WS(L_::ECQ + 4 * j) = EQ.w; RKH_WS(L_::ECQ + 4 * j + 1) = EQ.x;
      RKH_WS(L_::ECQ + 4 * j + 2) = EQ.y; RKH_WS(L_::ECQ + 4 * j + 3) = EQ.z;
      const m33 R = rotmat(EQ);
      pos = pos + mul(R, ldg3(J.off_pos));
      Q = qmul(EQ, ldg4(J.off_quat));
    }
  }
  const int n_env = sc->n_env, n_robot = sc->n_robot;
#pragma unroll 1
  for (int r = 0; r < n_robot; ++r) {
    if (__all(hit)) break;
    // robot shape -> global pose (pose_3D::getGlobalPose, pose_3D.hpp:102-110)
    const ShapeDev& sh = sc->robot[r];
    const int j = sh.link;
    const d3 Epos = mk3(RKH_WS(L_::ECP + 3 * j), RKH_WS(L_::ECP + 3 * j + 1), RKH_WS(L_::ECP + 3 * j + 2));
    const d4 EQ = d4{RKH_WS(L_::ECQ + 4 * j), RKH_WS(L_::ECQ + 4 * j + 1), RKH_WS(L_::ECQ + 4 * j + 2),
                     RKH_WS(L_::ECQ + 4 * j + 3)};
    ShapeG A;
    A.kind = sh.kind;
    A.pos = Epos + qrot(EQ, ldg3(sh.pos));
    A.q = qmul(EQ, ldg4(sh.quat));
    A.d0 = sh.dims[0]; A.d1 = sh.dims[1]; A.d2 = sh.dims[2];
    const d3 ca = pose_to_parent(A.pos, A.q, mk3(0, 0, 0));
    const double ra = sh.brad;
    const bool a_sphere = (sh.kind == RKH_SHAPE_SPHERE), a_ccyl = (sh.kind == RKH_SHAPE_CCYLINDER);
#pragma unroll 1
    for (int o0 = 0; o0 < n_env; o0 += 64) {
      const int on = (n_env - o0 < 64) ? n_env - o0 : 64;
      // bounding-sphere cull (proxy_query_model.cpp:384-389), one bit per surviving obstacle; kind masks are uniform
      unsigned long long mask = 0ull, k_sphere = 0ull, k_box = 0ull, k_ccyl = 0ull;
#pragma unroll 1
      for (int i = 0; i < on; ++i) {
        const ShapeDev& es = sc->env[o0 + i];
        const int ke = es.kind;
        const unsigned long long bit = 1ull << i;
        if (ke == RKH_SHAPE_SPHERE) k_sphere |= bit;
        else if (ke == RKH_SHAPE_BOX) k_box |= bit;
        else k_ccyl |= bit;
        // shape1 is the sphere if there is one, else the capped cylinder (createProxFinderList order)
        const bool s1_is_robot = a_sphere || (a_ccyl && ke != RKH_SHAPE_SPHERE);
        if (!a_sphere && !a_ccyl && ke == RKH_SHAPE_BOX) continue;  // box-box: no finder in the reference
        const d3 cb = pose_to_parent(ldg3(es.pos), ldg4(es.quat), mk3(0, 0, 0));
        const d3 c1 = s1_is_robot ? ca : cb, c2c = s1_is_robot ? cb : ca;
        const double r1 = s1_is_robot ? ra : es.brad, r2 = s1_is_robot ? es.brad : ra;
        if (!(norm_2(c2c - c1) - r1 - r2 > 0.0)) mask |= bit;
      }
      if (hit) mask = 0ull;
      // survivors, kind by kind
#pragma unroll 1
      for (int kind = 0; kind < 3; ++kind) {
        unsigned long long m = mask & (kind == 0 ? k_sphere : (kind == 1 ? k_box : k_ccyl));
        while (__any(m != 0ull)) {
          if (m != 0ull) {
            const int i = __builtin_ctzll(m);
            m &= m - 1ull;
            const ShapeDev& es = sc->env[o0 + i];  // per-lane gather (L1 / L2 resident table)
            ShapeG Bv;
            Bv.kind = es.kind;
            Bv.pos = ldg3(es.pos);
            Bv.q = ldg4(es.quat);
            Bv.d0 = es.dims[0]; Bv.d1 = es.dims[1]; Bv.d2 = es.dims[2];
            double d;
            if (kind == 0) {
              d = a_sphere ? dist_sphere_sphere(A, Bv) : (a_ccyl ? dist_sphere_ccyl(Bv, A) : dist_sphere_box(Bv, A));
            } else if (kind == 1) {
              d = a_sphere ? dist_sphere_box(A, Bv) : dist_ccyl_box(A, Bv);
            } else {
              d = a_sphere ? dist_sphere_ccyl(A, Bv) : (a_ccyl ? dist_ccyl_ccyl(A, Bv) : dist_ccyl_box(Bv, A));
            }
            if (d < 0.0) {
              hit = true;
              m = 0ull;
              mask = 0ull;
            }
          }
        }
      }
    }
  }
  return !(hit && active);
}

}  // namespace

template <int N>
__global__ __launch_bounds__(64, 1) void propagate_lane_kernel(const SceneDev* __restrict__ sc, DynDev dyn, EdgeIO io_a,
                                                                EdgeIO io_b, const EdgeIO* __restrict__ tab_a,
                                                                const EdgeIO* __restrict__ tab_b, uint32_t grid_a,
                                                                double* __restrict__ ws_all) {
  __shared__ LaneLds<N> lds;
  typedef WsLayout<N> L_;
  constexpr int D = 2 * N;
  const bool group_b = blockIdx.x >= grid_a;
  const EdgeIO io = tab_a ? (group_b ? tab_b[blockIdx.y] : tab_a[blockIdx.y]) : (group_b ? io_b : io_a);
  const uint32_t B = io.d_B ? *io.d_B : io.B;
  const int lane = threadIdx.x;
  const uint32_t e0 = (group_b ? blockIdx.x - grid_a : blockIdx.x) * 64u;
  if (e0 >= B) return;
  const uint32_t e = e0 + lane;
  const bool edge_valid = e < B;
  const uint32_t ec = edge_valid ? e : e0;  // idle lanes shadow the wave's first edge, results discarded
  double* __restrict__ ws = ws_all + (uint64_t(blockIdx.y) * gridDim.x + blockIdx.x) * uint64_t(L_::SLOTS * 64) + lane;
  const uint32_t si = io.src_idx ? io.src_idx[ec] : ((io.d_src_first ? *io.d_src_first : 0u) + ec);
  const uint64_t trow = (io.d_tgt_off ? uint64_t(*io.d_tgt_off) : 0ull) + ec;
  const double* __restrict__ a_row = io.src + uint64_t(si) * io.src_stride;
  const double* __restrict__ b_row = io.tgt + trow * io.tgt_stride;
  double* __restrict__ record = edge_valid ? io.record : nullptr;
  const int record_stride = io.record_stride;
#pragma unroll 1
  for (int d = 0; d < D; ++d) {
    const double av = a_row[d];
    RKH_WS(L_::X + d) = av;
    RKH_WS(L_::B + d) = b_row[d];
    if (record) record[(uint64_t(e) * record_stride + 0) * D + d] = av;
  }

  uint32_t n_free = 0;
  bool singular = false;
  bool alive = edge_valid;
#pragma unroll 1
  for (int k = 0; k < dyn.n_steps; ++k) {
    // distance(x_current, x_goal) > goal_proximity_threshold (exact left-to-right sum, vect_distance_metrics.hpp:126-137)
    {
      double s = 0.0;
#pragma unroll 1
      for (int d = 0; d < D; ++d) {
        const double xv = RKH_WS(L_::X + d);
        RKH_WS(L_::XE + d) = xv;
        const double df = xv - RKH_WS(L_::B + d);
        s = s + df * df;
      }
      if (!(sqrt(s) > dyn.goal_tol)) alive = false;
    }
    if (!__any(alive)) break;
    // PD law, zero-order hold over the step
#pragma unroll 1
    for (int j = 0; j < N; ++j) {
      double v = dyn.kp * (RKH_WS(L_::B + 2 * j) - RKH_WS(L_::X + 2 * j)) +
                 dyn.kd * (RKH_WS(L_::B + 2 * j + 1) - RKH_WS(L_::X + 2 * j + 1));
      if (v > dyn.u_max) v = dyn.u_max;
      else if (v < -dyn.u_max) v = -dyn.u_max;
      lds.u[j][lane] = v;
    }
    // runge_kutta4_integrate_impl (runge_kutta4_integrator_sys.hpp:53-97): the four useful f-evals per inner step as
    // the stages of a rolled loop (one copy of the dynamics in the instruction stream)
    const double h = dyn.dt;
    bool sing_now = false;
    const int n_evals = 4 * dyn.inner[k];
#pragma unroll 1
    for (int ev = 0; ev < n_evals; ++ev) {
      double qdd[N];
      lane_state_derivative<N>(sc, lds, lane, ws, qdd, sing_now);
      const int stage = ev & 3;
#pragma unroll
      for (int j = 0; j < N; ++j) {
        // components 2j (q: derivative = qd of the differentiated state) and 2j+1 (qd: derivative = qdd)
        const double xq = RKH_WS(L_::XE + 2 * j), xqd = RKH_WS(L_::XE + 2 * j + 1);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const int d = 2 * j + half;
          const double xv = half ? xqd : xq;
          const double dp = half ? qdd[j] : xqd;
          double xn;
          if (stage == 0) {
            const double k1 = h * dp;
            RKH_WS(L_::W + d) = xv;
            RKH_WS(L_::KA + d) = k1;
            xn = xv + 0.5 * k1;
          } else if (stage == 1) {
            const double k2 = h * dp;
            const double k1 = RKH_WS(L_::KA + d);
            RKH_WS(L_::KA + d) = (1.0 / 6.0) * k1 + (2.0 / 6.0) * k2;
            xn = RKH_WS(L_::W + d) + 0.5 * k2;
          } else if (stage == 2) {
            const double k3v = h * dp;
            RKH_WS(L_::K3 + d) = k3v;
            xn = RKH_WS(L_::W + d) + k3v;
          } else {
            xn = xv + ((RKH_WS(L_::KA + d) + (h / 6.0) * dp) - (2.0 / 3.0) * RKH_WS(L_::K3 + d));
          }
          RKH_WS(L_::XE + d) = xn;
        }
      }
    }
    if (sing_now && alive) {
      singular = true;
      alive = false;
    }
    // is_free(x_next): hyperbox bounds (hyperbox_topology.hpp:178-189), then proximity
    bool oob = false;
#pragma unroll 1
    for (int d = 0; d < D; ++d) {
      const double lo = dyn.lower[d], hi = dyn.upper[d], xv = RKH_WS(L_::XE + d);
      if (lo < hi) oob = oob || (xv < lo) || (xv > hi);
      else oob = oob || (xv > lo) || (xv < hi);
    }
    if (oob) alive = false;
    if (!__any(alive)) break;
    if (!lane_proximity_free<N>(sc, lane, ws, alive)) alive = false;
    if (alive) {
      ++n_free;
#pragma unroll 1
      for (int d = 0; d < D; ++d) {
        const double xv = RKH_WS(L_::XE + d);
        RKH_WS(L_::X + d) = xv;
        if (record) record[(uint64_t(e) * record_stride + n_free) * D + d] = xv;
      }
    }
  }
  if (singular && edge_valid) atomicExch(io.err_flag, int(RKH_ERR_SINGULAR));
  double s_ar = 0.0, s_ab = 0.0, s_rb = 0.0;
#pragma unroll 1
  for (int d = 0; d < D; ++d) {
    const double xv = RKH_WS(L_::X + d), av = a_row[d], bv = RKH_WS(L_::B + d);
    if (edge_valid) io.x_out[uint64_t(e) * D + d] = xv;
    const double d_ar = av - xv, d_ab = av - bv, d_rb = xv - bv;
    s_ar = s_ar + d_ar * d_ar;
    s_ab = s_ab + d_ab * d_ab;
    s_rb = s_rb + d_rb * d_rb;
  }
  if (edge_valid) io.steps_free[e] = n_free;
  if (io.mode != EDGE_PLAIN && edge_valid) {
    const double n_ar = sqrt(s_ar), n_ab = sqrt(s_ab), n_rb = sqrt(s_rb);
    if (io.mode == EDGE_STEER_ACCEPT) {
      // planning_visitor_base::steer_towards_position (planning_visitors.hpp:349-360)
      const double best_case = io.best_case ? io.best_case[ec] : n_ab;
      const bool ok = (!isinf(n_ar)) && (n_ar < 2.0 * best_case) && (n_ar > io.steer_tol * best_case);
      io.accept[e] = ok ? 1 : 0;
    } else {
      // C_free distance used by the goal probe (MEAQR_topology.hpp:995-1003)
      io.goal_dist[si - 1] = (n_ab * 0.05 > n_rb) ? n_ab : INFINITY;
    }
  }
}
#undef RKH_WS

// bytes of workspace a launch of (edges_a + edges_b) edges per problem needs
size_t propagate_lanes_workspace_bytes(int n_dof, uint32_t edges_a, uint32_t edges_b, uint32_t n_problems) {
  const size_t waves = size_t((edges_a + 63) / 64 + (edges_b + 63) / 64) * n_problems;
  return waves * size_t(29 * n_dof) * 64 * sizeof(double);
}

template <int N>
static void launch_lane_t(hipStream_t s, const SceneDev* d_scene, const DynDev& dyn, const EdgeIO& io, uint32_t edges_a,
                          const EdgeIO& io_b, uint32_t edges_b, const EdgeIO* tab_a, const EdgeIO* tab_b,
                          uint32_t n_problems, double* d_ws) {
  const uint32_t ga = (edges_a + 63) / 64, gbk = (edges_b + 63) / 64;
  hipLaunchKernelGGL((propagate_lane_kernel<N>), dim3(ga + gbk, n_problems), dim3(64), 0, s, d_scene, dyn, io, io_b, tab_a,
                     tab_b, ga, d_ws);
}

rkh_status launch_propagate_lanes(hipStream_t s, int n_dof, const SceneDev* d_scene, const DynDev& dyn, const EdgeIO& io,
                                  uint32_t grid_edges, const EdgeIO* io_b, uint32_t grid_b, const EdgeIO* tab_a,
                                  const EdgeIO* tab_b, uint32_t n_problems, double* d_ws) {
  const uint32_t eb = (io_b || tab_b) ? grid_b : 0u;
  if (grid_edges + eb == 0 || n_problems == 0) return RKH_OK;
  if (!d_ws) {
    set_error("propagate (one lane per edge): no workspace");
    return RKH_ERR_BAD_ARG;
  }
  const EdgeIO second = io_b ? *io_b : EdgeIO();
  switch (n_dof) {
    case 1: launch_lane_t<1>(s, d_scene, dyn, io, grid_edges, second, eb, tab_a, tab_b, n_problems, d_ws); break;
    case 2: launch_lane_t<2>(s, d_scene, dyn, io, grid_edges, second, eb, tab_a, tab_b, n_problems, d_ws); break;
    case 3: launch_lane_t<3>(s, d_scene, dyn, io, grid_edges, second, eb, tab_a, tab_b, n_problems, d_ws); break;
    case 6: launch_lane_t<6>(s, d_scene, dyn, io, grid_edges, second, eb, tab_a, tab_b, n_problems, d_ws); break;
    default:
      set_error("propagate: chains with this number of joints are not instantiated (1,2,3,6)");
      return RKH_ERR_UNSUPPORTED;
  }
  RKH_HIP(hipGetLastError());
  return RKH_OK;
}

}  // namespace rkh
