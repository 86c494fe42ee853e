// propagate_lane.hip -- the throughput mapping of the steer kernel: TWO LANES PER CANDIDATE EDGE, 32 edges per wave
// (gfx950, wave64).
//
// Same function as propagate_kernel (propagate.hip; reference citations there): RK4 forward dynamics of the KTE
// serial chain under the held PD input, proximity test after every step, accept test / goal probe at the end.
// propagate_kernel gives a whole wavefront to one edge (lowest latency, but its serial base->tip / tip->base sweeps
// run redundantly on all 64 lanes).  Here lane (el, h) = (lane & 31, lane >> 5) works on edge el of the wave: the two
// halves h run the serial sweeps together and split what is independent (odd / even Jacobian columns, odd / even
// rows of the mass matrix, half / full angle sin-cos, odd / even robot shapes of the proximity test).
//
// What shapes the kernel is storage and code size, not arithmetic.  Three one-lane-per-edge versions were measured
// first (C2, 6 joints): fully unrolled with the working set in registers = 33 k instructions (270 KB, 4x the
// instruction cache) -> no faster than the wave-per-edge kernel; inner loops rolled with register arrays -> 400..1400
// spilled registers (VALU instructions address only the 256 architectural VGPRs = 128 doubles per lane); everything
// rolled with the run-time-indexed arrays in a global workspace -> 250 dependent L2 round trips per f-eval with one
// wave per SIMD to hide them, again no gain.  Hence:
//   * every loop over joints / Jacobian columns / matrix entries is ROLLED (one copy of each body and one copy of
//     sincos in the instruction stream, ~25 KB in total);
//   * every array those loops index at run time lives in LDS, [slot][edge]: joint end frames E[7N], the current body's
//     Jacobian columns T[6N], the mass matrix Mf[N*N] (read-modify-written once per body), link forces FT[6N], the
//     state being differentiated XE[2N]: 159 slots (joint 0's position is the chain base, not stored) x 32 edges x 8 B
//     = 40 704 B per wave = the CU's 160 KB at one wave per SIMD (32 edges, two lanes each: every lane in use);
//     full-angle cos/sin[2N] and the held input u[N] sit in the wave's global workspace (written and read once per
//     joint and f-eval);
//   * registers hold the sweep recurrences and the Cholesky factor only; the RK4 stage vectors, touched once per
//     f-eval with independent loads, are in a global workspace [slot][lane];
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

RKH_DI double readlane_d(double v, int src_lane) {  // wave-uniform copy of lane src_lane's value
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
  return __hiloint2double(hi, lo);
}

RKH_DI float readlane_f(float v, int src_lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src_lane));
}

RKH_DI m33 lane_axis_angle_rotmat(double ca, double sa, d3 ax) {  // axis_angle::getRotMat (rotations_3D.hpp:2160-2180)
  const double omc = 1.0 - ca;
  const double t11 = ca + omc * ax.x * ax.x, t22 = ca + omc * ax.y * ax.y, t33 = ca + omc * ax.z * ax.z;
  const double t12 = omc * ax.x * ax.y, t13 = omc * ax.x * ax.z, t23 = omc * ax.y * ax.z;
  const double t01 = sa * ax.x, t02 = sa * ax.y, t03 = sa * ax.z;
  return m33{t11, t12 - t03, t13 + t02, t12 + t03, t22, t23 - t01, t13 - t02, t23 + t01, t33};
}

constexpr int kEdgesPerWave = 32;  // six joints: 159 LDS slots x 32 edges x 8 B = 40 704 B: four waves per CU, one per SIMD

// per-edge arrays in LDS, [slot][edge]
template <int N>
struct LdsLayout {
  enum : int {
    ECP = 0,            // joint end frames: position of joints 1 .. N-1 (joint 0 sits at the chain base: a constant)
    ECQ = 3 * N - 3,    //                   quaternion
    T = 7 * N - 3,      // Jacobian columns (v, w) of the current body; afterwards the generalized forces f[N]
    MF = 13 * N - 3,    // Tcm^T (Mcm Tcm) before symmetrisation
    FT = 13 * N - 3 + N * N,  // inertia_3D d'Alembert force / torque per link
    XE = 19 * N - 3 + N * N,  // state being differentiated / tested
    // (the held input u and cos / sin of the full joint angles -- written once and read once per joint and f-eval --
    // live in the wave's global workspace, and joint 0's position is not stored: 3 N + 3 slots less than everything in
    // LDS make room for 32 edges per wave -- every lane of the wave -- instead of 28)
    SLOTS = 21 * N - 3 + N * N
  };
};
template <int N>
struct LaneLds {
  double v[LdsLayout<N>::SLOTS][kEdgesPerWave];
  double axis[N][3];  // revolute_joint_3D::mAxis of every joint: the one chain constant that is indexed per lane
};
#define RKH_LD(slot) lds.v[(slot)][el]

// global workspace of one wave, [slot][lane]; `ws` below already points at the lane's column
template <int N>
struct WsLayout {
  enum : int {
    X = 0,        // last free state
    B = 2 * N,    // steer target
    W = 4 * N,    // RK4: state at the start of the inner step
    KA = 6 * N,   // RK4: k1, then (1/6) k1 + (2/6) k2
    K3 = 8 * N,   // RK4: k3
    U = 10 * N,   // held input of the step
    C1S1 = 11 * N,  // cos, sin of the full joint angles of the state being differentiated
    SLOTS = 13 * N
  };
};
#define RKH_WS(slot) ws[(slot) * 64]

// x' = f(x, u) of one edge.  In: the state in LD(XE..), the input in LD(U..).  Out: qdd[j] (the q components of x'
// are the qd components of x).  el = the edge's LDS column, h = which of the edge's two lanes this is.
template <int N, bool DIAG = false>
__device__ __forceinline__ void lane_state_derivative(const SceneDev* __restrict__ sc, LaneLds<N>& lds, int el, int h,
                                                      double (&qdd)[N], bool& singular, double* __restrict__ u_ptr,
                                                      int u_stride, unsigned long long* stamps = nullptr) {
  // u_ptr: this lane's column of the wave's global workspace from WsLayout::U on (u[N], then cos / sin[2 N])
  typedef LdsLayout<N> L_;
  // diagnostic instantiation only (rkh_diag_feval_cycles): per-phase cycle counts
  unsigned long long t_prev = DIAG ? __builtin_readcyclecounter() : 0ull;
#define RKH_STAMP(i)                                                \
  if (DIAG) {                                                       \
    const unsigned long long t_now = __builtin_readcyclecounter(); \
    stamps[i] += t_now - t_prev;                                    \
    t_prev = t_now;                                                 \
  }
#pragma unroll 1
  for (int e = h; e < N * N; e += 2) {
    const int i = e / N, jx = e - i * N;
    RKH_LD(L_::MF + e) = (i == jx) ? (0.0 + sc->joints[i].joint_inertia) : 0.0;  // inertia_gen rows: Tcm = 1
  }

  // ---- base -> tip sweep (kte_map_chain::doMotion) with the Jacobian columns and M terms of each body
  d3 pos = ldg3(sc->base_pos);
  const d3 base_pos0 = pos;  // = joint 0's end-frame position (not kept in LDS)
  d4 Q = ldg4(sc->base_quat);
  d3 w = mk3(0, 0, 0), alpha = mk3(0, 0, 0);
  d3 acc = ldg3(sc->base_acc);
#pragma unroll 1
  for (int j = 0; j < N; ++j) {
    const JointDev& J = sc->joints[j];
    const d3 axis = ldg3(J.axis), axis_n = ldg3(J.axis_n);
    const double q = RKH_LD(L_::XE + 2 * j), qd = RKH_LD(L_::XE + 2 * j + 1);
    // one sincos per lane: half angle on lane h = 0, full angle on lane h = 1 (kept for the tip->base sweep)
    double sn, cs;
    sincos(h ? q : 0.5 * q, &sn, &cs);
    const double c2 = __shfl(cs, el, 64), s2 = __shfl(sn, el, 64);
    // every lane keeps its own copy of the full-angle pair (no cross-lane traffic through memory)
    u_ptr[(N + 2 * j) * u_stride] = __shfl(cs, el + 32, 64);
    u_ptr[(N + 2 * j + 1) * u_stride] = __shfl(sn, el + 32, 64);
    // revolute_joint_3D::doMotion (revolute_joint.cpp:121-148)
    const d4 tq = d4{c2, axis_n.x * s2, axis_n.y * s2, axis_n.z * s2};
    const m33 R2 = rotmat(tq);
    const d4 EQ = qmul(Q, tq);
    const d3 wb = mulT(w, R2);
    const d3 qa = qd * axis;
    const d3 Ew = wb + qa;
    const d3 Ealpha = mulT(alpha, R2) + cross(wb, qa);
    if (!h) {
      if (j > 0) {  // joint 0's end frame sits at the chain base
        RKH_LD(L_::ECP + 3 * j - 3) = pos.x; RKH_LD(L_::ECP + 3 * j - 2) = pos.y; RKH_LD(L_::ECP + 3 * j - 1) = pos.z;
      }
      RKH_LD(L_::ECQ + 4 * j) = EQ.w; RKH_LD(L_::ECQ + 4 * j + 1) = EQ.x;
      RKH_LD(L_::ECQ + 4 * j + 2) = EQ.y; RKH_LD(L_::ECQ + 4 * j + 3) = EQ.z;
    }
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
    RKH_STAMP(0)
    if (h) {
      RKH_LD(L_::FT + 6 * j) = Fi.x; RKH_LD(L_::FT + 6 * j + 1) = Fi.y; RKH_LD(L_::FT + 6 * j + 2) = Fi.z;
      RKH_LD(L_::FT + 6 * j + 3) = Ti.x; RKH_LD(L_::FT + 6 * j + 4) = Ti.y; RKH_LD(L_::FT + 6 * j + 5) = Ti.z;
    }
    // Jacobian columns of body j w.r.t. coords c = j-h, j-h-2, ..: get_jac_relative_to (motion_jacobians.hpp:238-251)
    // with f2 = (~F_c) * F_b (frame_3D.hpp:184-189,222-238,368-382); the edge's two lanes take alternate columns
    // two columns per iteration (independent chains); a column index below 0 is computed on column 0's frame and
    // not stored
#pragma unroll 1
    for (int ct = j - h; ct >= 0; ct -= 4) {
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int cr = ct - 2 * r;
        const int c = cr >= 0 ? cr : 0;
        const int cs3 = c > 0 ? 3 * c - 3 : 0;
        d3 cp = mk3(RKH_LD(L_::ECP + cs3), RKH_LD(L_::ECP + cs3 + 1), RKH_LD(L_::ECP + cs3 + 2));
        if (c == 0) cp = base_pos0;  // joint 0: the chain base
        const d4 cq = d4{RKH_LD(L_::ECQ + 4 * c), RKH_LD(L_::ECQ + 4 * c + 1), RKH_LD(L_::ECQ + 4 * c + 2),
                         RKH_LD(L_::ECQ + 4 * c + 3)};
        const m33 Rc = rotmat(cq);
        const d4 iq = qinv(cq);
        const d3 ipos = mulT(-cp, Rc);
        // rotmat(conj(q)) is rotmat(q) transposed bit for bit (negating x, y, z flips the sign of the w-products and
        // leaves the others unchanged), so Ri * pos is evaluated as pos * Rc: same products, same order
        const d3 f2pos = ipos + mulT(pos, Rc);
        const d4 f2q = qmul(iq, Q);
        const m33 Rf = rotmat(f2q);
        // joints[c].axis: per-lane index (the edge's two lanes work on different columns), from the wave's LDS copy
        const d3 ax_c = mk3(lds.axis[c][0], lds.axis[c][1], lds.axis[c][2]);
        const d3 wt = mulT(ax_c, Rf);
        const d3 vt = mulT(cross(ax_c, f2pos), Rf);
        if (cr >= 0) {
          RKH_LD(L_::T + c * 6 + 0) = vt.x; RKH_LD(L_::T + c * 6 + 1) = vt.y; RKH_LD(L_::T + c * 6 + 2) = vt.z;
          RKH_LD(L_::T + c * 6 + 3) = wt.x; RKH_LD(L_::T + c * 6 + 4) = wt.y; RKH_LD(L_::T + c * 6 + 5) = wt.z;
        }
      }
    }
    RKH_STAMP(1)
    // Mf += Tcm_b^T (Mcm_b Tcm_b): summation order of mat_alg_symmetric.hpp:551-566 and mat_operators.hpp:104-114;
    // the edge's two lanes take alternate rows i
    // this lane's rows i = h, h+2, h+4 (N <= 6: at most three; for longer chains a second pass): their Jacobian columns do
    // not depend on jx, so they are read from LDS once per body; rows beyond j repeat row j's operands and are not stored
#pragma unroll 1
    for (int i0 = h; i0 <= j; i0 += 6) {
      double Ti[3][6];
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const int i = (i0 + 2 * r <= j) ? i0 + 2 * r : j;
#pragma unroll
        for (int k = 0; k < 6; ++k) Ti[r][k] = RKH_LD(L_::T + i * 6 + k);
      }
#pragma unroll 1
      for (int jx = 0; jx <= j; ++jx) {
        const double m0 = J.mass * RKH_LD(L_::T + jx * 6 + 0), m1 = J.mass * RKH_LD(L_::T + jx * 6 + 1),
                     m2 = J.mass * RKH_LD(L_::T + jx * 6 + 2);
        const d3 P = sym_mul(J.inertia, mk3(RKH_LD(L_::T + jx * 6 + 3), RKH_LD(L_::T + jx * 6 + 4), RKH_LD(L_::T + jx * 6 + 5)));
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          if (i0 + 2 * r <= j) {
            const int i = i0 + 2 * r;
            double s = RKH_LD(L_::MF + i * N + jx);
            s = s + Ti[r][0] * m0;
            s = s + Ti[r][1] * m1;
            s = s + Ti[r][2] * m2;
            s = s + Ti[r][3] * P.x;
            s = s + Ti[r][4] * P.y;
            s = s + Ti[r][5] * P.z;
            RKH_LD(L_::MF + i * N + jx) = s;
          }
        }
      }
    }
    RKH_STAMP(2)
  }

  // ---- tip -> base sweep (kte_map_chain::doForce in reverse op order); f[j] lands in the (now free) T slots
  {
    d3 LF = mk3(0, 0, 0), LT = mk3(0, 0, 0);
    if (sc->beam_on) {  // flexible_beam_3D::doForce: listed last, so first in the reverse pass (first term of the sums)
      d3 BF, BT;
      beam_force(pos, Q, ldg3(sc->beam_pos), ldg4(sc->beam_quat), sc->beam_rest, sc->beam_k, sc->beam_kt, &BF, &BT);
      LF = LF + BF;
      LT = LT + BT;
    }
#pragma unroll 1
    for (int j = N - 1; j >= 0; --j) {
      const JointDev& J = sc->joints[j];
      const d3 axis = ldg3(J.axis);
      LF = LF - mk3(RKH_LD(L_::FT + 6 * j), RKH_LD(L_::FT + 6 * j + 1), RKH_LD(L_::FT + 6 * j + 2));  // inertia_3D::doForce
      LT = LT - mk3(RKH_LD(L_::FT + 6 * j + 3), RKH_LD(L_::FT + 6 * j + 4), RKH_LD(L_::FT + 6 * j + 5));
      const m33 Ro = ldgm(J.off_R);                 // rigid_link_3D::doForce (rigid_link.cpp:170-178)
      const d3 op = ldg3(J.off_pos);
      const d3 tmp_force = mul(Ro, LF);
      const d3 ET = mul(Ro, LT) + cross(op, tmp_force);
      const m33 Ra = lane_axis_angle_rotmat(u_ptr[(N + 2 * j) * u_stride], u_ptr[(N + 2 * j + 1) * u_stride],
                                            ldg3(J.axis_n));  // revolute_joint_3D::doForce (revolute_joint.cpp:170-181)
      const double ta = dot(ET, axis);
      LF = mul(Ra, tmp_force);
      LT = mul(Ra, ET - ta * axis);
      const double uj = u_ptr[j * u_stride];  // inertia_gen::doForce (q_ddot = 0), driving_actuator_gen::doForce
      if (!h) RKH_LD(L_::T + j) = ta + uj;
      LT = LT - uj * axis;
    }
  }

  RKH_STAMP(3)
  // ---- mat<symmetric>(general): 0.5 * (M(j,i) + M(i,j)), j < i (mat_alg_symmetric.hpp:183-187), lower triangle
  double L[N][N], f[N];
#pragma unroll
  for (int i = 0; i < N; ++i) {
    f[i] = RKH_LD(L_::T + i);
#pragma unroll
    for (int j = 0; j <= i; ++j)
      L[i][j] = (i == j) ? RKH_LD(L_::MF + i * N + i) : 0.5 * (RKH_LD(L_::MF + j * N + i) + RKH_LD(L_::MF + i * N + j));
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
  RKH_STAMP(4)
#undef RKH_STAMP
}

// is the configuration in LD(XE..) (joint angles) collision-free?  (manip_dk_proxy_env_impl::is_free, proximity only)
// The edge's two lanes take alternate robot shapes; the verdict is combined across them.
template <int N, bool DIAG = false>
__device__ __forceinline__ bool lane_proximity_free(const SceneDev* __restrict__ sc, LaneLds<N>& lds, int el, int h,
                                                    bool active, unsigned long long* stamps = nullptr) {
  typedef LdsLayout<N> L_;
  unsigned long long t_prev = DIAG ? __builtin_readcyclecounter() : 0ull;
#define RKH_STAMP(i)                                                \
  if (DIAG) {                                                       \
    const unsigned long long t_now = __builtin_readcyclecounter(); \
    stamps[i] += t_now - t_prev;                                    \
    t_prev = t_now;                                                 \
  }
  bool hit = !active;  // inactive lanes take no part in the scan
  // Table loads are issued one stage ahead of their use (one wave per SIMD: nothing else covers a global-load latency):
  // this lane's first robot shape and its cull record of the first obstacle chunk before the kinematics, every
  // further robot shape at the top of the iteration before the one that uses it.
  struct RobotConst {
    d3 pos; d4 q; double d0, d1, d2, brad; int kind, link;
  };
  const int n_env = sc->n_env, n_robot = sc->n_robot;
  auto load_robot = [&](int r0) {
    const int r = (r0 + h < n_robot) ? r0 + h : (r0 < n_robot ? r0 : 0);
    const ShapeDev& sh = sc->robot[r];
    return RobotConst{ldg3(sh.pos), ldg4(sh.quat), sh.dims[0], sh.dims[1], sh.dims[2], sh.brad, sh.kind, sh.link};
  };
  RobotConst nxt = load_robot(0);
  const int ol0 = (int(threadIdx.x & 63) < n_env) ? int(threadIdx.x & 63) : 0;
  const float e0x = float(sc->env_cull[ol0][0]), e0y = float(sc->env_cull[ol0][1]), e0z = float(sc->env_cull[ol0][2]),
              e0r = float(sc->env_cull[ol0][3]);
  {  // joint end frames: revolute_joint_3D / rigid_link_3D kinematics, position + orientation only
    d3 pos = ldg3(sc->base_pos);
    d4 Q = ldg4(sc->base_quat);
#pragma unroll 1
    for (int j = 0; j < N; ++j) {
      const JointDev& J = sc->joints[j];
      const d3 axis_n = ldg3(J.axis_n);
      double s2, c2;
      sincos(0.5 * RKH_LD(L_::XE + 2 * j), &s2, &c2);
      const d4 tq = d4{c2, axis_n.x * s2, axis_n.y * s2, axis_n.z * s2};
      const d4 EQ = qmul(Q, tq);
      if (!h) {
        if (j > 0) {
          RKH_LD(L_::ECP + 3 * j - 3) = pos.x; RKH_LD(L_::ECP + 3 * j - 2) = pos.y; RKH_LD(L_::ECP + 3 * j - 1) = pos.z;
        }
        RKH_LD(L_::ECQ + 4 * j) = EQ.w; RKH_LD(L_::ECQ + 4 * j + 1) = EQ.x;
        RKH_LD(L_::ECQ + 4 * j + 2) = EQ.y; RKH_LD(L_::ECQ + 4 * j + 3) = EQ.z;
      }
      const m33 R = rotmat(EQ);
      pos = pos + mul(R, ldg3(J.off_pos));
      Q = qmul(EQ, ldg4(J.off_quat));
    }
  }
  RKH_STAMP(5)
#pragma unroll 1
  for (int r0 = 0; r0 < n_robot; r0 += 2) {
    if (__all(hit)) break;
    const int r = r0 + h;
    const bool have = r < n_robot;
    const RobotConst sh = nxt;
    nxt = load_robot(r0 + 2);
    // robot shape -> global pose (pose_3D::getGlobalPose, pose_3D.hpp:102-110)
    const int j = sh.link;
    const int js3 = j > 0 ? 3 * j - 3 : 0;
    d3 Epos = mk3(RKH_LD(L_::ECP + js3), RKH_LD(L_::ECP + js3 + 1), RKH_LD(L_::ECP + js3 + 2));
    if (j == 0) Epos = ldg3(sc->base_pos);
    const d4 EQ = d4{RKH_LD(L_::ECQ + 4 * j), RKH_LD(L_::ECQ + 4 * j + 1), RKH_LD(L_::ECQ + 4 * j + 2),
                     RKH_LD(L_::ECQ + 4 * j + 3)};
    ShapeG A;
    A.kind = sh.kind;
    A.pos = Epos + qrot(EQ, sh.pos);
    A.q = qmul(EQ, sh.q);
    A.d0 = sh.d0; A.d1 = sh.d1; A.d2 = sh.d2;
    const d3 ca = pose_to_parent(A.pos, A.q, mk3(0, 0, 0));
    const double ra = sh.brad;
    const bool a_sphere = (sh.kind == RKH_SHAPE_SPHERE), a_ccyl = (sh.kind == RKH_SHAPE_CCYLINDER);
    // capped cylinder: its axis segment (for the cull below)
    const d3 a_ax = qrot(A.q, mk3(0.0, 0.0, 1.0));
    const bool a_box = !a_sphere && !a_ccyl;
    const double seg_hl = a_ccyl ? 0.5 * A.d0 : 0.0, seg_rad_m = (a_ccyl ? A.d1 : ra) + 1e-9;
#pragma unroll 1
    for (int o0 = 0; o0 < n_env; o0 += 64) {
      const int on = (n_env - o0 < 64) ? n_env - o0 : 64;
      const unsigned long long k_sphere = sc->env_kind_mask[0][o0 >> 6], k_box = sc->env_kind_mask[1][o0 >> 6],
                               k_ccyl = sc->env_kind_mask[2][o0 >> 6];
      // Cull, one bit per surviving obstacle.  Lane l fetches the cull record of obstacle o0 + l once; the uniform loop
      // over the obstacles reads it back with v_readlane (no memory latency in the loop).  A pair is dropped only if a
      // lower bound on its distance is positive -- the bounding-sphere test of proxy_query_model.cpp:384-389 or, for
      // a capped-cylinder robot shape, the distance from the obstacle's bounding sphere to the cylinder's axis segment --
      // so the verdict "some pair is closer than 0" is unchanged.
      const int ol = (o0 + int(threadIdx.x & 63) < n_env) ? o0 + int(threadIdx.x & 63) : o0;
      // The cull runs in fp32 with fused multiply-adds (it is a conservative filter, not part of the reference's
      // arithmetic): records and the shape's segment are rounded to fp32 and the reach carries a 1 mm margin, three
      // orders of magnitude above the rounding of the fp32 evaluation at these magnitudes (coordinates of a few metres).
      float ecx = e0x, ecy = e0y, ecz = e0z, ecr = e0r;
      if (o0 != 0) {  // further chunks of 64 obstacles (uniform branch)
        ecx = float(sc->env_cull[ol][0]); ecy = float(sc->env_cull[ol][1]); ecz = float(sc->env_cull[ol][2]);
        ecr = float(sc->env_cull[ol][3]);
      }
      const float cax = float(ca.x), cay = float(ca.y), caz = float(ca.z);
      const float aax = float(a_ax.x), aay = float(a_ax.y), aaz = float(a_ax.z);
      const float shl = float(seg_hl), srm = float(seg_rad_m) + 1e-3f;
      unsigned long long mask = 0ull;
      // four obstacles per iteration: four independent dependency chains (one wave per SIMD: nothing else hides the
      // latency); spare slots of the last iteration repeat obstacle o0's record and are masked off
      // returns 1 when the pair survives (branch-free: selects only)
      auto cull_one = [&](int i) -> unsigned {
        const float vx = readlane_f(ecx, i & 63) - cax, vy = readlane_f(ecy, i & 63) - cay, vz = readlane_f(ecz, i & 63) - caz;
        const float rb = readlane_f(ecr, i & 63);
        // a sphere / box robot shape is a segment of length 0 with its bounding radius
        float t = __builtin_fmaf(vz, aaz, __builtin_fmaf(vy, aay, vx * aax));
        t = __builtin_fminf(__builtin_fmaxf(t, -shl), shl);
        const float wx = __builtin_fmaf(-t, aax, vx), wy = __builtin_fmaf(-t, aay, vy), wz = __builtin_fmaf(-t, aaz, vz);
        const float w2 = __builtin_fmaf(wz, wz, __builtin_fmaf(wy, wy, wx * wx));
        const float reach = srm + rb;
        return (w2 > reach * reach) ? 0u : 1u;
      };
#pragma unroll 1
      for (int i = 0; i < on; i += 4) {
        const unsigned nib = cull_one(i) | (cull_one(i + 1) << 1) | (cull_one(i + 2) << 2) | (cull_one(i + 3) << 3);
        mask |= (unsigned long long)nib << i;
      }
      // obstacles past the end of the chunk; box-box has no finder in the reference (proxy_query_model.cpp:367)
      mask &= (on == 64) ? ~0ull : ((1ull << on) - 1ull);
      mask &= a_box ? ~k_box : ~0ull;
      if (hit || !have) mask = 0ull;
      RKH_STAMP(6)
      // survivors, kind by kind
#pragma unroll 1
      for (int kind = 0; kind < 3; ++kind) {
        unsigned long long m = mask & (kind == 0 ? k_sphere : (kind == 1 ? k_box : k_ccyl));
        if (kind == 1 && a_ccyl) {
          // capped cylinder against a box = a golden-section search along the axis (prox_fundamentals_3D.cpp:108-115,
          // ~7 k cycles that hold the whole wave while any lane runs one).  Every value that search can return is the
          // distance of SOME point of the axis segment to the box, so a lower bound over the segment that already
          // exceeds the radius settles the verdict "no collision" without it: separation along the box's own axes,
          // |c_k| - hl |t_k| - half_k, in the box frame (fp64, margin 1e-9).  First pass: drop those pairs.
          unsigned long long keep = 0ull, mm = m;
          while (__any(mm != 0ull)) {
            if (mm != 0ull) {
              const int i = __builtin_ctzll(mm);
              mm &= mm - 1ull;
              const ShapeDev& es = sc->env[o0 + i];
              const d4 bq = qinv(ldg4(es.quat));
              const d3 crel = qrot(bq, ca - ldg3(es.pos));
              const d3 trel = qrot(bq, a_ax);
              const double hl = 0.5 * A.d0;
              const double gx = fabs(crel.x) - fabs(trel.x) * hl - 0.5 * es.dims[0];
              const double gy = fabs(crel.y) - fabs(trel.y) * hl - 0.5 * es.dims[1];
              const double gz = fabs(crel.z) - fabs(trel.z) * hl - 0.5 * es.dims[2];
              if (!(fmax(gx, fmax(gy, gz)) > A.d1 + 1e-9)) keep |= 1ull << i;
            }
          }
          m = keep;
        }
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
    // What the edge's other lane found counts for both.  The exchange is its own statement: inside `hit || shfl(..)` it
    // would run only on the lanes that have not hit, and a cross-lane read of a lane that is switched off returns 0
    // (the second lane's collisions went unnoticed unless the first lane's shape collided too).
    const int other_hit = __shfl_xor(hit ? 1 : 0, 32, 64);
    hit = hit || (other_hit != 0);
    RKH_STAMP(7)
  }
  return !(hit && active);
#undef RKH_STAMP
}

}  // namespace

template <int N>
__global__ __launch_bounds__(64, 1) void propagate_lane_kernel(const SceneDev* __restrict__ sc, DynDev dyn, EdgeIO io_a,
                                                                EdgeIO io_b, const EdgeIO* __restrict__ tab_a,
                                                                const EdgeIO* __restrict__ tab_b, uint32_t grid_a,
                                                                double* __restrict__ ws_all, KernelGate gate) {
  __shared__ LaneLds<N> lds;
  typedef LdsLayout<N> L_;
  typedef WsLayout<N> W_;
  if (gate.count) {  // the planner's per-round choice between the kernel mappings
    const uint32_t c = *gate.count;
    if (c < gate.lo || c >= gate.hi) return;
  }
  constexpr int D = 2 * N;
  bool group_b = blockIdx.x >= grid_a;
  uint32_t problem = blockIdx.y;
  uint32_t wave = group_b ? blockIdx.x - grid_a : blockIdx.x;
  if (gate.wave_base) {  // compact mapping: block L of the grid (dispatch order) takes working wave L
    const uint32_t L = blockIdx.y * gridDim.x + blockIdx.x;
    if (L >= gate.wave_base[gate.n_segments]) return;
    uint32_t lo = 0, hi = gate.n_segments;  // wave_base[lo] <= L < wave_base[hi]
    while (hi - lo > 1) {
      const uint32_t mid = (lo + hi) >> 1;
      if (gate.wave_base[mid] <= L) lo = mid;
      else hi = mid;
    }
    problem = lo >> 1;
    group_b = (lo & 1u) != 0u;
    wave = L - gate.wave_base[lo];
  }
  if (threadIdx.x < 3 * N) lds.axis[threadIdx.x / 3][threadIdx.x % 3] = sc->joints[threadIdx.x / 3].axis[threadIdx.x % 3];
  __syncthreads();
  const EdgeIO io = tab_a ? (group_b ? tab_b[problem] : tab_a[problem]) : (group_b ? io_b : io_a);
  const uint32_t B = io.d_B ? *io.d_B : io.B;
  const int lane = threadIdx.x;
  const uint32_t e0 = wave * uint32_t(kEdgesPerWave);
  if (e0 >= B) return;
  const int h = lane >> 5;
  const int el_raw = lane & 31;
  const bool lane_used = el_raw < kEdgesPerWave;
  const int el = lane_used ? el_raw : 0;  // the four spare lanes mirror edge slot 0 (identical values)
  const uint32_t e = e0 + uint32_t(el);
  const bool edge_valid = e < B;
  const bool writer = edge_valid && lane_used && h == 0;  // the lane that exports the edge's results
  const uint32_t ec = edge_valid ? e : e0;  // idle slots shadow the wave's first edge, results discarded
  double* __restrict__ ws = ws_all + (uint64_t(blockIdx.y) * gridDim.x + blockIdx.x) * uint64_t(W_::SLOTS * 64) + lane;
  const uint32_t si = io.src_idx ? io.src_idx[ec] : ((io.d_src_first ? *io.d_src_first : 0u) + ec);
  const uint64_t trow = (io.d_tgt_off ? uint64_t(*io.d_tgt_off) : 0ull) + ec;
  const double* __restrict__ a_row = io.src + uint64_t(si) * io.src_stride;
  const double* __restrict__ b_row = io.tgt + trow * io.tgt_stride;
  double* __restrict__ record = writer ? io.record : nullptr;
  const int record_stride = io.record_stride;
#pragma unroll 1
  for (int d = 0; d < D; ++d) {
    const double av = a_row[d];
    RKH_WS(W_::X + d) = av;
    RKH_WS(W_::B + d) = b_row[d];
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
        const double xv = RKH_WS(W_::X + d);
        RKH_LD(L_::XE + d) = xv;
        const double df = xv - RKH_WS(W_::B + d);
        s = s + df * df;
      }
      if (!(sqrt(s) > dyn.goal_tol)) alive = false;
    }
    if (!__any(alive)) break;
    // PD law, zero-order hold over the step
#pragma unroll 1
    for (int j = 0; j < N; ++j) {
      double v = dyn.kp * (RKH_WS(W_::B + 2 * j) - RKH_WS(W_::X + 2 * j)) +
                 dyn.kd * (RKH_WS(W_::B + 2 * j + 1) - RKH_WS(W_::X + 2 * j + 1));
      if (v > dyn.u_max) v = dyn.u_max;
      else if (v < -dyn.u_max) v = -dyn.u_max;
      RKH_WS(W_::U + j) = v;
    }
    // runge_kutta4_integrate_impl (runge_kutta4_integrator_sys.hpp:53-97): the four useful f-evals per inner step as
    // the stages of a rolled loop (one copy of the dynamics in the instruction stream)
    const double h_dt = dyn.dt;
    bool sing_now = false;
    const int n_evals = 4 * dyn.inner[k];
#pragma unroll 1
    for (int ev = 0; ev < n_evals; ++ev) {
      double qdd[N];
      lane_state_derivative<N>(sc, lds, el, h, qdd, sing_now, &RKH_WS(W_::U), 64);
      const int stage = ev & 3;
#pragma unroll
      for (int j = 0; j < N; ++j) {
        // components 2j (q: derivative = qd of the differentiated state) and 2j+1 (qd: derivative = qdd)
        const double xq = RKH_LD(L_::XE + 2 * j), xqd = RKH_LD(L_::XE + 2 * j + 1);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const int d = 2 * j + half;
          const double xv = half ? xqd : xq;
          const double dp = half ? qdd[j] : xqd;
          double xn;
          if (stage == 0) {
            const double k1 = h_dt * dp;
            RKH_WS(W_::W + d) = xv;
            RKH_WS(W_::KA + d) = k1;
            xn = xv + 0.5 * k1;
          } else if (stage == 1) {
            const double k2 = h_dt * dp;
            const double k1 = RKH_WS(W_::KA + d);
            RKH_WS(W_::KA + d) = (1.0 / 6.0) * k1 + (2.0 / 6.0) * k2;
            xn = RKH_WS(W_::W + d) + 0.5 * k2;
          } else if (stage == 2) {
            const double k3v = h_dt * dp;
            RKH_WS(W_::K3 + d) = k3v;
            xn = RKH_WS(W_::W + d) + k3v;
          } else {
            xn = xv + ((RKH_WS(W_::KA + d) + (h_dt / 6.0) * dp) - (2.0 / 3.0) * RKH_WS(W_::K3 + d));
          }
          RKH_LD(L_::XE + d) = xn;  // both lanes of the edge write the same value
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
      const double lo = dyn.lower[d], hi = dyn.upper[d], xv = RKH_LD(L_::XE + d);
      if (lo < hi) oob = oob || (xv < lo) || (xv > hi);
      else oob = oob || (xv > lo) || (xv < hi);
    }
    if (oob) alive = false;
    if (!__any(alive)) break;
    if (!lane_proximity_free<N>(sc, lds, el, h, alive)) alive = false;
    if (alive) {
      ++n_free;
#pragma unroll 1
      for (int d = 0; d < D; ++d) {
        const double xv = RKH_LD(L_::XE + d);
        RKH_WS(W_::X + d) = xv;
        if (record) record[(uint64_t(e) * record_stride + n_free) * D + d] = xv;
      }
    }
  }
  if (singular && writer) atomicExch(io.err_flag, int(RKH_ERR_SINGULAR));
  double s_ar = 0.0, s_ab = 0.0, s_rb = 0.0;
#pragma unroll 1
  for (int d = 0; d < D; ++d) {
    const double xv = RKH_WS(W_::X + d), av = a_row[d], bv = RKH_WS(W_::B + d);
    if (writer) io.x_out[uint64_t(e) * D + d] = xv;
    const double d_ar = av - xv, d_ab = av - bv, d_rb = xv - bv;
    s_ar = s_ar + d_ar * d_ar;
    s_ab = s_ab + d_ab * d_ab;
    s_rb = s_rb + d_rb * d_rb;
  }
  if (writer) io.steps_free[e] = n_free;
  if (io.mode != EDGE_PLAIN && writer) {
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

// Diagnostic kernel (not on the product path): `iters` back-to-back f-evals + proximity tests of kEdgesPerWave states per wave
// with cycle counts per phase: [frames + sincos, jacobian columns, mass matrix, force sweep, cholesky,
// proximity: joint frames, cull, exact routines]
template <int N>
__global__ __launch_bounds__(64, 1) void lane_cycles_kernel(const SceneDev* __restrict__ sc, const double* __restrict__ x,
                                                             const double* __restrict__ u, uint32_t B, int iters,
                                                             unsigned long long* __restrict__ out,
                                                             double* __restrict__ sink_out, double* __restrict__ ws_all) {
  __shared__ LaneLds<N> lds;
  if (threadIdx.x < 3 * N) lds.axis[threadIdx.x / 3][threadIdx.x % 3] = sc->joints[threadIdx.x / 3].axis[threadIdx.x % 3];
  __syncthreads();
  typedef LdsLayout<N> L_;
  const int lane = threadIdx.x, h = lane >> 5, el_raw = lane & 31;
  const int el = el_raw < kEdgesPerWave ? el_raw : 0;
  uint32_t e = blockIdx.x * kEdgesPerWave + el;
  if (e >= B) e = blockIdx.x * kEdgesPerWave;
  for (int d = 0; d < 2 * N; ++d) RKH_LD(L_::XE + d) = x[uint64_t(e) * 2 * N + d];
  double* ws_u = ws_all + uint64_t(blockIdx.x) * uint64_t(3 * N * 64) + lane;  // u[N], cos / sin[2 N] of this lane
  for (int j = 0; j < N; ++j) ws_u[j * 64] = u[uint64_t(e) * N + j];
  unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  bool singular = false;
  double accv = 0.0;
  const unsigned long long t_begin = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    double qdd[N];
    lane_state_derivative<N, true>(sc, lds, el, h, qdd, singular, ws_u, 64, st);
#pragma unroll
    for (int j = 0; j < N; ++j) {
      accv += qdd[j];
      RKH_LD(L_::XE + 2 * j + 1) = RKH_LD(L_::XE + 2 * j + 1) + 1e-4 * qdd[j];
    }
    accv += lane_proximity_free<N, true>(sc, lds, el, h, true, st) ? 1.0 : 0.0;
  }
  (void)t_begin;
  if (lane == 0) {
    for (int i = 0; i < 8; ++i) out[blockIdx.x * 8 + i] = st[i];
    sink_out[blockIdx.x] = accv + (singular ? 1.0 : 0.0);
  }
}

rkh_status launch_lane_cycles(hipStream_t s, int n_dof, const SceneDev* d_scene, const double* d_x, const double* d_u,
                              uint32_t B, int iters, unsigned long long* d_out, double* d_sink) {
  const uint32_t waves = (B + kEdgesPerWave - 1) / kEdgesPerWave;
  double* d_ws = nullptr;  // per wave: u[N] + cos / sin[2 N] per lane (the kernels' global workspace slots)
  RKH_HIP(hipMalloc(&d_ws, size_t(waves) * 3 * n_dof * 64 * sizeof(double)));
  switch (n_dof) {
    case 6: hipLaunchKernelGGL((lane_cycles_kernel<6>), dim3(waves), dim3(64), 0, s, d_scene, d_x, d_u, B, iters, d_out, d_sink, d_ws); break;
    case 3: hipLaunchKernelGGL((lane_cycles_kernel<3>), dim3(waves), dim3(64), 0, s, d_scene, d_x, d_u, B, iters, d_out, d_sink, d_ws); break;
    default: (void)hipFree(d_ws); set_error("lane diagnostics: instantiated for 3 and 6 joints"); return RKH_ERR_UNSUPPORTED;
  }
  RKH_HIP(hipGetLastError());
  RKH_HIP(hipStreamSynchronize(s));  // diagnostic entry point: the scratch is released right away
  (void)hipFree(d_ws);
  return RKH_OK;
}

// bytes of workspace a launch of (edges_a + edges_b) edges per problem needs
size_t propagate_lanes_workspace_bytes(int n_dof, uint32_t edges_a, uint32_t edges_b, uint32_t n_problems) {
  const size_t waves = size_t((edges_a + kEdgesPerWave - 1) / kEdgesPerWave + (edges_b + kEdgesPerWave - 1) / kEdgesPerWave) * n_problems;
  return waves * size_t(13 * n_dof) * 64 * sizeof(double);
}

template <int N>
static void launch_lane_t(hipStream_t s, const SceneDev* d_scene, const DynDev& dyn, const EdgeIO& io, uint32_t edges_a,
                          const EdgeIO& io_b, uint32_t edges_b, const EdgeIO* tab_a, const EdgeIO* tab_b,
                          uint32_t n_problems, double* d_ws, KernelGate gate) {
  const uint32_t ga = (edges_a + kEdgesPerWave - 1) / kEdgesPerWave, gbk = (edges_b + kEdgesPerWave - 1) / kEdgesPerWave;
  hipLaunchKernelGGL((propagate_lane_kernel<N>), dim3(ga + gbk, n_problems), dim3(64), 0, s, d_scene, dyn, io, io_b, tab_a,
                     tab_b, ga, d_ws, gate);
}

rkh_status launch_propagate_lanes(hipStream_t s, int n_dof, const SceneDev* d_scene, const DynDev& dyn, const EdgeIO& io,
                                  uint32_t grid_edges, const EdgeIO* io_b, uint32_t grid_b, const EdgeIO* tab_a,
                                  const EdgeIO* tab_b, uint32_t n_problems, double* d_ws, KernelGate gate) {
  const uint32_t eb = (io_b || tab_b) ? grid_b : 0u;
  if (grid_edges + eb == 0 || n_problems == 0) return RKH_OK;
  if (!d_ws) {
    set_error("propagate (one lane per edge): no workspace");
    return RKH_ERR_BAD_ARG;
  }
  const EdgeIO second = io_b ? *io_b : EdgeIO();
  switch (n_dof) {
    case 1: launch_lane_t<1>(s, d_scene, dyn, io, grid_edges, second, eb, tab_a, tab_b, n_problems, d_ws, gate); break;
    case 2: launch_lane_t<2>(s, d_scene, dyn, io, grid_edges, second, eb, tab_a, tab_b, n_problems, d_ws, gate); break;
    case 3: launch_lane_t<3>(s, d_scene, dyn, io, grid_edges, second, eb, tab_a, tab_b, n_problems, d_ws, gate); break;
    case 4: launch_lane_t<4>(s, d_scene, dyn, io, grid_edges, second, eb, tab_a, tab_b, n_problems, d_ws, gate); break;
    case 7: launch_lane_t<7>(s, d_scene, dyn, io, grid_edges, second, eb, tab_a, tab_b, n_problems, d_ws, gate); break;
    case 6: launch_lane_t<6>(s, d_scene, dyn, io, grid_edges, second, eb, tab_a, tab_b, n_problems, d_ws, gate); break;
    default:
      set_error("propagate: chains with this number of joints are not instantiated (1,2,3,4,6,7)");
      return RKH_ERR_UNSUPPORTED;
  }
  RKH_HIP(hipGetLastError());
  return RKH_OK;
}

uint32_t lane_kernel_edges_per_wave() { return uint32_t(kEdgesPerWave); }

// resident waves per CU of the two-lanes kernel for this chain size (LDS-bound: 4 for six joints)
uint32_t lane_kernel_waves_per_cu(int n_dof) {
  int blocks = 0;
  hipError_t e = hipErrorInvalidValue;
  switch (n_dof) {
    case 1: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, propagate_lane_kernel<1>, 64, 0); break;
    case 2: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, propagate_lane_kernel<2>, 64, 0); break;
    case 3: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, propagate_lane_kernel<3>, 64, 0); break;
    case 4: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, propagate_lane_kernel<4>, 64, 0); break;
    case 6: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, propagate_lane_kernel<6>, 64, 0); break;
    case 7: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, propagate_lane_kernel<7>, 64, 0); break;
    default: break;
  }
  return (e == hipSuccess && blocks > 0) ? uint32_t(blocks) : 4u;
}

#undef RKH_LD
}  // namespace rkh
