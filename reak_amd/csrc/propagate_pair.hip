// propagate_pair.hip -- the throughput mapping of the steer kernel, second generation: TWO ADJACENT LANES PER CANDIDATE
// EDGE, 32 edges per wave, TWO WAVES PER SIMD (gfx950, wave64).
//
// Same function and same arithmetic as propagate_kernel (propagate.hip; reference citations there) and
// propagate_lane_kernel (propagate_lane.hip): RK4 forward dynamics of the KTE serial chain under the held PD input,
// proximity test after every step, accept test / goal probe at the end; every product and sum is formed in the
// reference's order (-ffp-contract=off), so all three kernels return bit-identical states and verdicts
// (tests/test_gpu_parity.py::test_propagate_mappings_*).
//
// What changed against propagate_lane_kernel (376 registers + 328 B scratch + 40.8 KB LDS per wave = one wave per SIMD,
// the VALU idle half of the time on LDS round trips):
//   * lane (el, h) = (lane >> 1, lane & 1): the edge's two lanes are neighbours, so they exchange values with a DPP
//     quad_perm move (one VALU instruction per 32 bits, no LDS, no select);
//   * the Jacobian columns of the current body live in REGISTERS: lane h computes the columns of the joints c = 2r + h
//     (r unrolled: static register indices), from the end frames of those joints, which it also keeps in registers
//     (a lane only ever needs the frames of its own columns);
//   * the mass matrix is accumulated in REGISTERS: lane h owns the rows i = 2r + h; Mcm * T of its own columns is formed
//     once per body and handed to the neighbour by DPP, so each lane has all columns' products for its rows.  Every
//     M(i, jx) receives its terms in the same ascending-body order (mat_alg_symmetric.hpp:551-566) -- no LDS
//     read-modify-write chain;
//   * LDS per edge shrinks from 159 to 75 slots (state 2N, cos / sin 2N, input N, generalized forces N, and one region
//     shared in time by the link forces (6N), the joint frames of the proximity test (7N - 3) and the assembled mass
//     matrix (N^2)) = 19.3 KB per wave: eight waves per CU;
//   * the RK4 stage vectors are split between the edge's lanes (lane h updates the joints 2r + h).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>
#include <cmath>

#include "device_math.h"
#include "proximity_device.h"
#include "rkh_internal.h"

namespace rkh {

namespace {

// The scene is read through a constant-address-space pointer: uniform addresses then always become scalar loads, also
// behind the laundering below (a laundered generic pointer would turn them into per-lane flat loads).
typedef const __attribute__((address_space(4))) SceneDev* ScenePtr;
typedef const __attribute__((address_space(4))) double* CDoubleP;
RKH_DI d3 ldg3(CDoubleP p) { return d3{p[0], p[1], p[2]}; }
RKH_DI d4 ldg4(CDoubleP p) { return d4{p[0], p[1], p[2], p[3]}; }
RKH_DI m33 ldgm(CDoubleP p) { return m33{p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], p[8]}; }
struct Sym6 {
  double t[6];
};
RKH_DI Sym6 ldsym(CDoubleP p) { return Sym6{{p[0], p[1], p[2], p[3], p[4], p[5]}}; }

RKH_DI float readlane_f(float v, int src_lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src_lane));
}

// the value the edge's other lane holds (lane ^ 1): v_mov_b32_dpp quad_perm:[1,0,3,2]
RKH_DI int xchg_i(int v) { return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true); }
RKH_DI double xchg(double v) {
  const int lo = xchg_i(__double2loint(v));
  const int hi = xchg_i(__double2hiint(v));
  return __hiloint2double(hi, lo);
}
RKH_DI d3 xchg3(d3 v) { return d3{xchg(v.x), xchg(v.y), xchg(v.z)}; }

RKH_DI m33 pair_axis_angle_rotmat(double ca, double sa, d3 ax) {  // axis_angle::getRotMat (rotations_3D.hpp:2160-2180)
  const double omc = 1.0 - ca;
  const double t11 = ca + omc * ax.x * ax.x, t22 = ca + omc * ax.y * ax.y, t33 = ca + omc * ax.z * ax.z;
  const double t12 = omc * ax.x * ax.y, t13 = omc * ax.x * ax.z, t23 = omc * ax.y * ax.z;
  const double t01 = sa * ax.x, t02 = sa * ax.y, t03 = sa * ax.z;
  return m33{t11, t12 - t03, t13 + t02, t12 + t03, t22, t23 - t01, t13 - t02, t23 + t01, t33};
}

constexpr int kPairEdges = 32;  // edges per wave: every lane in use

template <int N>
struct PairLayout {
  enum : int {
    R = (N + 1) / 2,  // joints (columns, rows) per lane
    XE = 0,           // state being differentiated / tested [2N]
    CS = 2 * N,       // cos, sin of the full joint angles [2N] (forward sweep -> backward sweep)
    U = 4 * N,        // held input of the step [N]
    F = 5 * N,        // generalized forces [N] (backward sweep -> Cholesky)
    REG = 6 * N,      // one region, three tenants in time:
    FT = REG,         //   inertia_3D d'Alembert force / torque per link [6N] (forward sweep -> backward sweep)
    MF = REG,         //   Tcm^T (Mcm Tcm) before symmetrisation [N*N] (after the backward sweep)
    ECP = REG,        //   proximity test: joint end frames, position of joints 1 .. N-1 [3N-3]
    ECQ = REG + 3 * N - 3,  //                                 quaternion [4N]
    REG_SLOTS = (6 * N > N * N ? (6 * N > 7 * N - 3 ? 6 * N : 7 * N - 3) : (N * N > 7 * N - 3 ? N * N : 7 * N - 3)),
    SLOTS = REG + REG_SLOTS
  };
};
template <int N>
struct PairLds {
  double v[PairLayout<N>::SLOTS][kPairEdges];
  double axis[N + 1][3];  // revolute_joint_3D::mAxis of every joint (indexed per lane: column 2r + h); one spare row
  // proximity test: queues of surviving (edge, robot shape, obstacle) pairs, their fill counts, one verdict per edge
  uint32_t q1[128], q2[64], qn[2], hit[kPairEdges];
};
#define RKH_LD(slot) lds.v[(slot)][el]

// global workspace of one wave, [slot][lane] (`ws` points at the lane's column): per-lane private values
template <int N>
struct PairWs {
  enum : int {
    R = (N + 1) / 2,
    W = 0,           // RK4: state at the start of the inner step (this lane's joints: 2 components each)
    KA = 2 * R,      // RK4: k1, then (1/6) k1 + (2/6) k2
    K3 = 4 * R,      // RK4: k3
    X = 6 * R,       // last free state [2N]
    B = 6 * R + 2 * N,  // steer target [2N]
    SLOTS = 6 * R + 4 * N
  };
};
// The workspace is addressed through a buffer resource: one descriptor (4 scalar registers) for the wave's region, the
// lane's byte offset in ONE vector register, the slot offset in the instruction's scalar offset.  (With flat pointers
// every slot beyond the 4 KB immediate range costs a 64-bit address pair held in vector registers across the kernel.)
typedef unsigned int rkh_u32x2 __attribute__((ext_vector_type(2)));
struct PairWsRef {
  __amdgpu_buffer_rsrc_t rsrc;
  int voff;  // lane * 8
};
RKH_DI double ws_ld(const PairWsRef& w, int slot) {
  const rkh_u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(w.rsrc, w.voff, slot * 512, 0);
  return __hiloint2double(int(v.y), int(v.x));
}
RKH_DI void ws_st(const PairWsRef& w, int slot, double d) {
  rkh_u32x2 v;
  v.x = (unsigned)__double2loint(d);
  v.y = (unsigned)__double2hiint(d);
  __builtin_amdgcn_raw_buffer_store_b64(v, w.rsrc, w.voff, slot * 512, 0);
}

// x' = f(x, u) of one edge.  In: the state in LD(XE..), the input in LD(U..).  Out: qdd[j] on both lanes (the q
// components of x' are the qd components of x).  el = the edge's LDS column, h = which of the edge's two lanes this is.
template <int N, bool DIAG = false>
__device__ __forceinline__ void pair_state_derivative(ScenePtr sc_in, PairLds<N>& lds, int el, int h,
                                                      double (&qdd)[N], bool& singular,
                                                      unsigned long long* stamps = nullptr) {
  typedef PairLayout<N> L_;
  constexpr int R = L_::R;
  // The scene pointer is laundered once per call: otherwise every scene constant this function reads (and every LDS
  // value that does not change between calls) is hoisted out of the caller's loops and held in registers across the
  // whole kernel -- the registers this kernel does not have (two waves per SIMD = 256 per lane).
  ScenePtr sc = sc_in;
  asm volatile("" : "+s"(sc), "+v"(h), "+v"(el) : : "memory");  // the lane's coordinates too: addresses derived from them
  unsigned long long t_prev = DIAG ? __builtin_readcyclecounter() : 0ull;
#define RKH_STAMP(i)                                                \
  if (DIAG) {                                                       \
    const unsigned long long t_now = __builtin_readcyclecounter(); \
    stamps[i] += t_now - t_prev;                                    \
    t_prev = t_now;                                                 \
  }
  // this lane's rows i = 2r + h of Tcm^T (Mcm Tcm): Mo[r][rr] = column 2rr + h (own parity), Mp[r][rr] = column
  // 2rr + 1 - h (the neighbour's parity).  inertia_gen rows: Tcm = 1 on the diagonal.
  double Mo[R][R], Mp[R][R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
#pragma unroll
    for (int rr = 0; rr < R; ++rr) {
      Mo[r][rr] = 0.0;
      Mp[r][rr] = 0.0;
    }
    const double ji0 = sc->joints[2 * r].joint_inertia;
    const double ji1 = (2 * r + 1 < N) ? sc->joints[2 * r + 1 < N ? 2 * r + 1 : 0].joint_inertia : 0.0;
    Mo[r][r] = 0.0 + (h ? ji1 : ji0);
  }
  // end frames of this lane's joints c = 2r + h (parents of its Jacobian columns)
  d3 Ecp[R];
  d4 Ecq[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    Ecp[r] = mk3(0.0, 0.0, 0.0);
    Ecq[r] = d4{1.0, 0.0, 0.0, 0.0};
  }

  // ---- base -> tip sweep (kte_map_chain::doMotion) with the Jacobian columns and M terms of each body
  d3 pos = ldg3(sc->base_pos);
  d4 Q = ldg4(sc->base_quat);
  d3 w = mk3(0, 0, 0), alpha = mk3(0, 0, 0);
  d3 acc = ldg3(sc->base_acc);
#pragma unroll 1
  for (int j = 0; j < N; ++j) {
    const auto& J = sc->joints[j];
    const d3 axis = ldg3(J.axis), axis_n = ldg3(J.axis_n);
    const double q = RKH_LD(L_::XE + 2 * j), qd = RKH_LD(L_::XE + 2 * j + 1);
    // one sincos per lane: half angle on lane h = 0, full angle on lane h = 1 (kept for the tip->base sweep)
    double sn, cs;
    sincos(h ? q : 0.5 * q, &sn, &cs);
    const double cs_o = xchg(cs), sn_o = xchg(sn);
    const double c2 = h ? cs_o : cs, s2 = h ? sn_o : sn;
    if (h) {
      RKH_LD(L_::CS + 2 * j) = cs;
      RKH_LD(L_::CS + 2 * j + 1) = sn;
    }
    // revolute_joint_3D::doMotion (revolute_joint.cpp:121-148)
    const d4 tq = d4{c2, axis_n.x * s2, axis_n.y * s2, axis_n.z * s2};
    const m33 R2 = rotmat(tq);
    const d4 EQ = qmul(Q, tq);
    const d3 wb = mulT(w, R2);
    const d3 qa = qd * axis;
    const d3 Ew = wb + qa;
    const d3 Ealpha = mulT(alpha, R2) + cross(wb, qa);
    {  // the joint's end frame, kept by the lane that owns column j
      const bool mine = ((j & 1) == h);
      const int rj = j >> 1;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const bool take = (rj == r) && mine;  // selects, not branches
        Ecp[r].x = take ? pos.x : Ecp[r].x; Ecp[r].y = take ? pos.y : Ecp[r].y; Ecp[r].z = take ? pos.z : Ecp[r].z;
        Ecq[r].w = take ? EQ.w : Ecq[r].w; Ecq[r].x = take ? EQ.x : Ecq[r].x;
        Ecq[r].y = take ? EQ.y : Ecq[r].y; Ecq[r].z = take ? EQ.z : Ecq[r].z;
      }
    }
    // rigid_link_3D::doMotion = frame * pose (frame_3D.hpp:240-255)
    const d3 op = ldg3(J.off_pos);
    const m33 Rm = rotmat(EQ);
    pos = pos + mul(Rm, op);
    acc = acc + mul(Rm, cross(Ew, cross(Ew, op)) + cross(Ealpha, op));
    const m33 Ro = ldgm(J.off_R);
    Q = qmul(EQ, ldg4(J.off_quat));
    alpha = mulT(Ealpha, Ro);
    w = mulT(Ew, Ro);
    // inertia_3D::doForce terms (inertia.cpp:111-122), applied in the backward sweep
    const Sym6 In6 = ldsym(J.inertia);
    const d3 Fi = J.mass * qrot(qinv(Q), acc);
    const d3 Ti = sym_mul(In6.t, alpha) + cross(w, sym_mul(In6.t, w));
    if (h) {
      RKH_LD(L_::FT + 6 * j) = Fi.x; RKH_LD(L_::FT + 6 * j + 1) = Fi.y; RKH_LD(L_::FT + 6 * j + 2) = Fi.z;
      RKH_LD(L_::FT + 6 * j + 3) = Ti.x; RKH_LD(L_::FT + 6 * j + 4) = Ti.y; RKH_LD(L_::FT + 6 * j + 5) = Ti.z;
    }
    RKH_STAMP(0)
    // Jacobian columns of body j w.r.t. this lane's coords c = 2r + h <= j: get_jac_relative_to
    // (motion_jacobians.hpp:238-251) with f2 = (~F_c) * F_b (frame_3D.hpp:184-189,222-238,368-382)
    d3 Tv[R], Tw[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      Tv[r] = mk3(0.0, 0.0, 0.0);
      Tw[r] = mk3(0.0, 0.0, 0.0);
      if (2 * r <= j) {  // uniform; a lane whose column 2r + 1 is beyond j computes on its (valid or identity) frame
        const d3 cp = Ecp[r];
        const d4 cq = Ecq[r];
        const m33 Rc = rotmat(cq);
        const d4 iq = qinv(cq);
        const d3 ipos = mulT(-cp, Rc);
        // rotmat(conj(q)) is rotmat(q) transposed bit for bit (negating x, y, z flips the sign of the w-products and
        // leaves the others unchanged), so Ri * pos is evaluated as pos * Rc: same products, same order
        const d3 f2pos = ipos + mulT(pos, Rc);
        const d4 f2q = qmul(iq, Q);
        const m33 Rf = rotmat(f2q);
        const int c = 2 * r + h;  // <= N: the spare row of lds.axis covers an odd chain's last slot
        const d3 ax_c = mk3(lds.axis[c][0], lds.axis[c][1], lds.axis[c][2]);
        const d3 wt = mulT(ax_c, Rf);
        const d3 vt = mulT(cross(ax_c, f2pos), Rf);
        // A column beyond j is an exact +0 vector: Mcm * 0 = +0 and s + (+-0) = s bit for bit (the sums start from +0
        // or from a joint inertia, so they are never -0), hence the accumulation below needs no masks.
        const bool act = (c <= j);
        Tw[r] = mk3(act ? wt.x : 0.0, act ? wt.y : 0.0, act ? wt.z : 0.0);
        Tv[r] = mk3(act ? vt.x : 0.0, act ? vt.y : 0.0, act ? vt.z : 0.0);
      }
    }
    RKH_STAMP(1)
    // Mf += Tcm_b^T (Mcm_b Tcm_b): summation order of mat_alg_symmetric.hpp:551-566 and mat_operators.hpp:104-114.
    // Column block rr: Mcm * T of this lane's column 2rr + h, and the neighbour's (column 2rr + 1 - h) by DPP.
#pragma unroll
    for (int rr = 0; rr < R; ++rr) {
      if (2 * rr <= j) {  // uniform
        const double m0 = J.mass * Tv[rr].x, m1 = J.mass * Tv[rr].y, m2 = J.mass * Tv[rr].z;
        const d3 P = sym_mul(In6.t, Tw[rr]);
        const double n0 = xchg(m0), n1 = xchg(m1), n2 = xchg(m2);
        const d3 Pn = xchg3(P);
#pragma unroll
        for (int r = 0; r < R; ++r) {
          if (2 * r <= j) {  // uniform
            double s = Mo[r][rr];
            s = s + Tv[r].x * m0;
            s = s + Tv[r].y * m1;
            s = s + Tv[r].z * m2;
            s = s + Tw[r].x * P.x;
            s = s + Tw[r].y * P.y;
            s = s + Tw[r].z * P.z;
            Mo[r][rr] = s;
            double t = Mp[r][rr];
            t = t + Tv[r].x * n0;
            t = t + Tv[r].y * n1;
            t = t + Tv[r].z * n2;
            t = t + Tw[r].x * Pn.x;
            t = t + Tw[r].y * Pn.y;
            t = t + Tw[r].z * Pn.z;
            Mp[r][rr] = t;
          }
        }
      }
    }
    RKH_STAMP(2)
  }

  // ---- tip -> base sweep (kte_map_chain::doForce in reverse op order)
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
      const auto& J = sc->joints[j];
      const d3 axis = ldg3(J.axis);
      LF = LF - mk3(RKH_LD(L_::FT + 6 * j), RKH_LD(L_::FT + 6 * j + 1), RKH_LD(L_::FT + 6 * j + 2));  // inertia_3D::doForce
      LT = LT - mk3(RKH_LD(L_::FT + 6 * j + 3), RKH_LD(L_::FT + 6 * j + 4), RKH_LD(L_::FT + 6 * j + 5));
      const m33 Ro = ldgm(J.off_R);                 // rigid_link_3D::doForce (rigid_link.cpp:170-178)
      const d3 op = ldg3(J.off_pos);
      const d3 tmp_force = mul(Ro, LF);
      const d3 ET = mul(Ro, LT) + cross(op, tmp_force);
      const m33 Ra = pair_axis_angle_rotmat(RKH_LD(L_::CS + 2 * j), RKH_LD(L_::CS + 2 * j + 1),
                                            ldg3(J.axis_n));  // revolute_joint_3D::doForce (revolute_joint.cpp:170-181)
      const double ta = dot(ET, axis);
      LF = mul(Ra, tmp_force);
      LT = mul(Ra, ET - ta * axis);
      const double uj = RKH_LD(L_::U + j);  // inertia_gen::doForce (q_ddot = 0), driving_actuator_gen::doForce
      if (!h) RKH_LD(L_::F + j) = ta + uj;
      LT = LT - uj * axis;
    }
  }
  RKH_STAMP(3)

  // ---- the two lanes' rows meet in LDS (the link forces are no longer needed)
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int i = 2 * r + h;
#pragma unroll
    for (int rr = 0; rr < R; ++rr) {
      const int co = 2 * rr + h, cn = 2 * rr + 1 - h;
      if (i < N && co < N) RKH_LD(L_::MF + i * N + co) = Mo[r][rr];
      if (i < N && cn < N) RKH_LD(L_::MF + i * N + cn) = Mp[r][rr];
    }
  }
  // ---- mat<symmetric>(general): 0.5 * (M(j,i) + M(i,j)), j < i (mat_alg_symmetric.hpp:183-187), lower triangle
  double L[N][N], f[N];
#pragma unroll
  for (int i = 0; i < N; ++i) {
    f[i] = RKH_LD(L_::F + i);
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

// ---- proximity test: work queues in LDS --------------------------------------------------------------------------
// The bounding cull runs per lane (lane h of an edge takes the robot shapes r0 + h); what survives it is NOT evaluated
// by the lane that found it but pushed, as (edge, robot shape, obstacle), into a queue in LDS and handed out one entry
// per lane: a wave spends one pass of the closed forms per 64 surviving pairs instead of one pass per pair of its
// unluckiest lane.  Pairs that need the golden-section search of findProximityBoxToLine (capped cylinder against box)
// go through a second queue, so the ~1.5 k instructions of a search are spent by full lanes only.
constexpr int kQ1Cap = 128, kQ2Cap = 64;
constexpr uint32_t kQEmpty = 0xFFFFFFFFu;
RKH_DI uint32_t q_pack(int el, int r, int o) { return (uint32_t(el) << 16) | (uint32_t(r) << 8) | uint32_t(o); }

// global pose of robot shape r of the edge in LDS column e (pose_3D::getGlobalPose, pose_3D.hpp:102-110)
template <int N>
RKH_DI ShapeG pair_robot_pose(ScenePtr sc, PairLds<N>& lds, int e, int r) {
  typedef PairLayout<N> L_;
  const auto& sh = sc->robot[r];
  const int j = sh.link;
  const int js3 = j > 0 ? 3 * j - 3 : 0;
  d3 Epos = mk3(lds.v[L_::ECP + js3][e], lds.v[L_::ECP + js3 + 1][e], lds.v[L_::ECP + js3 + 2][e]);
  if (j == 0) Epos = ldg3(sc->base_pos);
  const d4 EQ = d4{lds.v[L_::ECQ + 4 * j][e], lds.v[L_::ECQ + 4 * j + 1][e], lds.v[L_::ECQ + 4 * j + 2][e],
                   lds.v[L_::ECQ + 4 * j + 3][e]};
  ShapeG A;
  A.kind = sh.kind;
  A.pos = Epos + qrot(EQ, ldg3(sh.pos));
  A.q = qmul(EQ, ldg4(sh.quat));
  A.d0 = sh.dims[0]; A.d1 = sh.dims[1]; A.d2 = sh.dims[2];
  return A;
}
template <class EnvRef>
RKH_DI ShapeG pair_env_shape(const EnvRef& es) {
  ShapeG Bv;
  Bv.kind = es.kind;
  Bv.pos = ldg3(es.pos);
  Bv.q = ldg4(es.quat);
  Bv.d0 = es.dims[0]; Bv.d1 = es.dims[1]; Bv.d2 = es.dims[2];
  return Bv;
}

// golden-section pairs of the second queue (all lanes of the wave call this together)
template <int N>
RKH_DI void pair_drain_q2(ScenePtr sc, PairLds<N>& lds) {
  const int lane = threadIdx.x & 63;
  const uint32_t n2 = lds.qn[1] < uint32_t(kQ2Cap) ? lds.qn[1] : uint32_t(kQ2Cap);
  if (n2 == 0) return;  // uniform
  const uint32_t ent = (uint32_t(lane) < n2) ? lds.q2[lane] : kQEmpty;
  if (ent != kQEmpty) {
    const int e = int(ent >> 16), r = int((ent >> 8) & 0xFFu), o = int(ent & 0xFFu);
    if (lds.hit[e] == 0u) {
      const ShapeG A = pair_robot_pose<N>(sc, lds, e, r);
      const ShapeG Bv = pair_env_shape(sc->env[o]);
      const double d = (A.kind == RKH_SHAPE_CCYLINDER) ? dist_ccyl_box(A, Bv) : dist_ccyl_box(Bv, A);  // PR_CCYL_BOX only
      if (d < 0.0) lds.hit[e] = 1u;
    }
  }
  if (lane == 0) lds.qn[1] = 0u;
}

// the first queue: closed forms, and the separating-axis screen in front of the golden-section pairs
// n_exact / n_gold (diagnostic instantiation only): closed forms this lane evaluated, golden-section pairs it queued
template <int N>
RKH_DI void pair_drain_q1(ScenePtr sc, PairLds<N>& lds, uint32_t* n_exact = nullptr, uint32_t* n_gold = nullptr) {
  const int lane = threadIdx.x & 63;
  const uint32_t n1 = lds.qn[0] < uint32_t(kQ1Cap) ? lds.qn[0] : uint32_t(kQ1Cap);
#pragma unroll 1
  for (uint32_t k0 = 0; k0 < n1; k0 += 64u) {
    const uint32_t idx = k0 + uint32_t(lane);
    const uint32_t ent = (idx < n1) ? lds.q1[idx] : kQEmpty;
    bool golden = false;
    if (ent != kQEmpty) {
      const int e = int(ent >> 16), r = int((ent >> 8) & 0xFFu), o = int(ent & 0xFFu);
      if (lds.hit[e] == 0u) {
        const ShapeG A = pair_robot_pose<N>(sc, lds, e, r);
        const ShapeG Bv = pair_env_shape(sc->env[o]);
        bool a_first = true;
        const int rt = pair_routine(A.kind, Bv.kind, &a_first);  // createProxFinderList's cascade of kinds
        if (rt == PR_CCYL_BOX) {
          // capped cylinder against a box = a golden-section search along the axis (prox_fundamentals_3D.cpp:108-115).
          // Every value that search can return is the distance of SOME point of the axis segment to the box, so a lower
          // bound over the segment that already exceeds the radius settles "no collision" without it: separation along the
          // box's own axes, |c_k| - hl |t_k| - half_k, in the box frame (fp64, margin 1e-9).
          const ShapeG& cc = a_first ? A : Bv;
          const ShapeG& bx = a_first ? Bv : A;
          const d3 cy_c = pose_to_parent(cc.pos, cc.q, mk3(0, 0, 0));
          const d3 cy_t = qrot(cc.q, mk3(0.0, 0.0, 1.0));
          const d4 bq = qinv(bx.q);
          const d3 crel = qrot(bq, cy_c - bx.pos);
          const d3 trel = qrot(bq, cy_t);
          const double hl = 0.5 * cc.d0;
          const double gx = fabs(crel.x) - fabs(trel.x) * hl - 0.5 * bx.d0;
          const double gy = fabs(crel.y) - fabs(trel.y) * hl - 0.5 * bx.d1;
          const double gz = fabs(crel.z) - fabs(trel.z) * hl - 0.5 * bx.d2;
          golden = !(fmax(gx, fmax(gy, gz)) > cc.d1 + 1e-9);
        } else if (rt != PR_NONE) {
          const double d = a_first ? pair_distance<false>(rt, A, Bv) : pair_distance<false>(rt, Bv, A);
          if (d < 0.0) lds.hit[e] = 1u;
          if (n_exact) ++*n_exact;
        }
      }
    }
    // golden-section pairs move on to the second queue; when it is full it is drained first (uniform decisions)
    while (__any(golden)) {
      uint32_t slot = kQ2Cap;
      if (golden) slot = atomicAdd(&lds.qn[1], 1u);
      if (golden && slot < uint32_t(kQ2Cap)) {
        lds.q2[slot] = ent;
        golden = false;
        if (n_gold) ++*n_gold;
      }
      if (__any(golden)) pair_drain_q2<N>(sc, lds);  // some did not fit: empty the queue, then they try again
    }
  }
  if (lane == 0) lds.qn[0] = 0u;
}

// is the configuration in LD(XE..) (joint angles) collision-free?  (manip_dk_proxy_env_impl::is_free, proximity only)
template <int N, bool DIAG = false>
__device__ __forceinline__ bool pair_proximity_free(ScenePtr sc_in, PairLds<N>& lds, int el, int h,
                                                    bool active, unsigned long long* stamps = nullptr) {
  typedef PairLayout<N> L_;
  constexpr int R = L_::R;
  ScenePtr sc = sc_in;  // laundered, see pair_state_derivative
  asm volatile("" : "+s"(sc), "+v"(h), "+v"(el) : : "memory");
  unsigned long long t_prev = DIAG ? __builtin_readcyclecounter() : 0ull;
#define RKH_STAMP(i)                                                \
  if (DIAG) {                                                       \
    const unsigned long long t_now = __builtin_readcyclecounter(); \
    stamps[i] += t_now - t_prev;                                    \
    t_prev = t_now;                                                 \
  }
  const int lane = threadIdx.x & 63;
  const int n_env = sc->n_env, n_robot = sc->n_robot;
  uint32_t c_q1 = 0, c_exact = 0, c_gold = 0;  // (diagnostic instantiation only)
  if (lane < 2) lds.qn[lane] = 0u;
  if (lane < kPairEdges) lds.hit[lane] = 0u;
  // this lane's cull record of the first obstacle chunk, fetched ahead of the kinematics
  const int ol0 = (lane < n_env) ? lane : 0;
  const float e0x = float(sc->env_cull[ol0][0]), e0y = float(sc->env_cull[ol0][1]), e0z = float(sc->env_cull[ol0][2]),
              e0r = float(sc->env_cull[ol0][3]);
  {  // joint end frames: revolute_joint_3D / rigid_link_3D kinematics, position + orientation only.  The half-angle
     // sin / cos of the joints 2r + h are this lane's; both lanes read them back from the (idle) cos / sin slots.
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int j = 2 * r + h;
      const int jc = j < N ? j : N - 1;
      double s2, c2;
      sincos(0.5 * RKH_LD(L_::XE + 2 * jc), &s2, &c2);
      if (j < N) {
        RKH_LD(L_::CS + 2 * jc) = c2;
        RKH_LD(L_::CS + 2 * jc + 1) = s2;
      }
    }
    d3 pos = ldg3(sc->base_pos);
    d4 Q = ldg4(sc->base_quat);
#pragma unroll 1
    for (int j = 0; j < N; ++j) {
      const auto& J = sc->joints[j];
      const d3 axis_n = ldg3(J.axis_n);
      const double c2 = RKH_LD(L_::CS + 2 * j), s2 = RKH_LD(L_::CS + 2 * j + 1);
      const d4 tq = d4{c2, axis_n.x * s2, axis_n.y * s2, axis_n.z * s2};
      const d4 EQ = qmul(Q, tq);
      if (!h) {
        if (j > 0) {
          RKH_LD(L_::ECP + 3 * j - 3) = pos.x; RKH_LD(L_::ECP + 3 * j - 2) = pos.y; RKH_LD(L_::ECP + 3 * j - 1) = pos.z;
        }
        RKH_LD(L_::ECQ + 4 * j) = EQ.w; RKH_LD(L_::ECQ + 4 * j + 1) = EQ.x;
        RKH_LD(L_::ECQ + 4 * j + 2) = EQ.y; RKH_LD(L_::ECQ + 4 * j + 3) = EQ.z;
      }
      const m33 Rm = rotmat(EQ);
      pos = pos + mul(Rm, ldg3(J.off_pos));
      Q = qmul(EQ, ldg4(J.off_quat));
    }
  }
  RKH_STAMP(5)
#pragma unroll 1
  for (int r0 = 0; r0 < n_robot; r0 += 2) {
    const int r = r0 + h;
    const bool open = active && (lds.hit[el] == 0u);
    if (!__any(open)) break;  // every edge of the wave is settled
    const bool have = (r < n_robot) && open;
    const int rc = (r < n_robot) ? r : r0;
    const ShapeG A = pair_robot_pose<N>(sc, lds, el, rc);
    const d3 ca = pose_to_parent(A.pos, A.q, mk3(0, 0, 0));
    const bool a_ccyl = (A.kind == RKH_SHAPE_CCYLINDER);
    // capped cylinder: its axis segment (for the cull below)
    const d3 a_ax = qrot(A.q, mk3(0.0, 0.0, 1.0));
    const double ra = sc->robot[rc].brad;
    const double seg_hl = a_ccyl ? 0.5 * A.d0 : 0.0, seg_rad_m = (a_ccyl ? A.d1 : ra) + 1e-9;
    // static reach (SceneDev::robot_n_reach): the obstacles this lane's shape can touch at all are a prefix of the table
    const int my_reach = sc->robot_n_reach[rc];
    const int reach0 = sc->robot_n_reach[r0], reach1 = (r0 + 1 < n_robot) ? sc->robot_n_reach[r0 + 1] : 0;
    const int n_scan = reach0 > reach1 ? reach0 : reach1;  // uniform
#pragma unroll 1
    for (int o0 = 0; o0 < n_scan; o0 += 64) {
      const int on = (n_scan - o0 < 64) ? n_scan - o0 : 64;
      const unsigned long long has_finder = sc->env_finder_mask[A.kind][o0 >> 6];  // per lane (its shape's kind)
      // Cull, one bit per surviving obstacle.  Lane l fetches the cull record of obstacle o0 + l once; the uniform loop
      // over the obstacles reads it back with v_readlane (no memory latency in the loop).  A pair is dropped only if a
      // lower bound on its distance is positive -- the bounding-sphere test of proxy_query_model.cpp:384-389 or, for
      // a capped-cylinder robot shape, the distance from the obstacle's bounding sphere to the cylinder's axis segment --
      // so the verdict "some pair is closer than 0" is unchanged.  The cull runs in fp32 with fused multiply-adds (a
      // conservative filter, not part of the reference's arithmetic): records and the shape's segment are rounded to
      // fp32 and the reach carries a 1 mm margin, three orders of magnitude above the rounding of the fp32 evaluation at
      // these magnitudes (coordinates of a few metres).
      float ecx = e0x, ecy = e0y, ecz = e0z, ecr = e0r;
      if (o0 != 0) {  // further chunks of 64 obstacles (uniform branch)
        const int ol = (o0 + lane < n_env) ? o0 + lane : o0;
        ecx = float(sc->env_cull[ol][0]); ecy = float(sc->env_cull[ol][1]); ecz = float(sc->env_cull[ol][2]);
        ecr = float(sc->env_cull[ol][3]);
      }
      const float cax = float(ca.x), cay = float(ca.y), caz = float(ca.z);
      const float aax = float(a_ax.x), aay = float(a_ax.y), aaz = float(a_ax.z);
      const float shl = float(seg_hl), srm = float(seg_rad_m) + 1e-3f;
      unsigned long long mask = 0ull;
      auto cull_one = [&](int i) -> unsigned {  // 1 when the pair survives (branch-free: selects only)
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
      // obstacles past this lane's own reach; pairs of kinds the reference has no finder for
      const int mine = my_reach - o0;
      mask &= (mine >= 64) ? ~0ull : (mine <= 0 ? 0ull : ((1ull << mine) - 1ull));
      mask &= has_finder;
      if (!have) mask = 0ull;
      // survivors -> first queue.  A lane whose entries do not all fit writes placeholders into the part of its range
      // that lies inside the queue and keeps its mask for the next turn.
      while (__any(mask != 0ull)) {
        const uint32_t cnt = uint32_t(__popcll(mask));
        uint32_t base = 0u;
        if (cnt) base = atomicAdd(&lds.qn[0], cnt);
        const bool fits = base + cnt <= uint32_t(kQ1Cap);
        if (DIAG && cnt && fits) c_q1 += cnt;
        if (cnt) {
          unsigned long long mm = mask;
          for (uint32_t k = 0; k < cnt; ++k) {
            const int i = __builtin_ctzll(mm);
            mm &= mm - 1ull;
            if (base + k < uint32_t(kQ1Cap)) lds.q1[base + k] = fits ? q_pack(el, rc, o0 + i) : kQEmpty;
          }
          if (fits) mask = 0ull;
        }
        if (__any(mask != 0ull) || lds.qn[0] >= 64u) pair_drain_q1<N>(sc, lds, DIAG ? &c_exact : nullptr, DIAG ? &c_gold : nullptr);
      }
    }
    RKH_STAMP(6)
  }
  pair_drain_q1<N>(sc, lds, DIAG ? &c_exact : nullptr, DIAG ? &c_gold : nullptr);
  pair_drain_q2<N>(sc, lds);
  RKH_STAMP(7)
  if (DIAG) {  // stamps[8..10]: (robot shape, obstacle) pairs past the cull, closed forms evaluated, golden-section searches
    uint32_t v0 = c_q1, v1 = c_exact, v2 = c_gold;
    for (int off = 32; off > 0; off >>= 1) {
      v0 += __shfl_xor(v0, off, 64);
      v1 += __shfl_xor(v1, off, 64);
      v2 += __shfl_xor(v2, off, 64);
    }
    stamps[8] += v0;
    stamps[9] += v1;
    stamps[10] += v2;
  }
  const bool hit = (lds.hit[el] != 0u);
  return !(hit && active);
#undef RKH_STAMP
}

}  // namespace

// All kernel arguments travel in ONE struct and are read through the kernarg segment pointer (scalar loads at the point
// of use, through a laundered pointer).  Referencing by-value parameters directly makes the compiler load every field in
// the entry block and keep it in registers for the whole kernel; a parameter array indexed at run time is even copied to
// scratch.  This kernel has neither registers nor scratch to spare.
struct PairArgs {
  const SceneDev* sc;
  DynDev dyn;
  EdgeIO io_a, io_b;
  const EdgeIO* tab_a;
  const EdgeIO* tab_b;
  uint32_t grid_a;
  double* ws_all;
  KernelGate gate;
};
typedef const __attribute__((address_space(4))) PairArgs* PairArgP;
RKH_DI PairArgP pair_args() {
  PairArgP a = (PairArgP)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(a));
  return a;
}

template <int N>
__global__ __launch_bounds__(64, 2) void propagate_pair_kernel(PairArgs) {
  __shared__ PairLds<N> lds;
  typedef PairLayout<N> L_;
  typedef PairWs<N> W_;
  constexpr int R = L_::R;
  constexpr int D = 2 * N;
  const int lane = threadIdx.x;
  const int h = lane & 1;
  const int el = lane >> 1;
  uint32_t problem, wave;
  bool group_b;
  {
    PairArgP A = pair_args();
    if (A->gate.count) {  // the planner's per-round choice between the kernel mappings
      const uint32_t c = *A->gate.count;
      if (c < A->gate.lo || c >= A->gate.hi) return;
    }
    const uint32_t grid_a = A->grid_a;
    group_b = blockIdx.x >= grid_a;
    problem = blockIdx.y;
    wave = group_b ? blockIdx.x - grid_a : blockIdx.x;
    const uint32_t* wave_base = A->gate.wave_base;
    if (wave_base) {  // compact mapping: block L of the grid (dispatch order) takes working wave L
      const uint32_t n_segments = A->gate.n_segments;
      const uint32_t L = blockIdx.y * gridDim.x + blockIdx.x;
      if (L >= wave_base[n_segments]) return;
      uint32_t lo = 0, hi = n_segments;  // wave_base[lo] <= L < wave_base[hi]
      while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (wave_base[mid] <= L) lo = mid;
        else hi = mid;
      }
      problem = lo >> 1;
      group_b = (lo & 1u) != 0u;
      wave = L - wave_base[lo];
    }
  }
  const ScenePtr sc = (ScenePtr)pair_args()->sc;
  if (threadIdx.x < 3 * (N + 1)) {
    const int jj = threadIdx.x / 3;
    lds.axis[jj][threadIdx.x % 3] = sc->joints[jj < N ? jj : N - 1].axis[threadIdx.x % 3];
  }
  __syncthreads();
  // the launch's EdgeIO record: an entry of a device table, or one of the two by-value copies in the kernarg segment
  auto edge_io = [&]() -> const EdgeIO* {
    PairArgP A = pair_args();
    const EdgeIO* tab = group_b ? A->tab_b : A->tab_a;
    if (A->tab_a) return tab + problem;
    const char* ka = (const char*)(const void*)A;
    return (const EdgeIO*)(ka + (group_b ? offsetof(PairArgs, io_b) : offsetof(PairArgs, io_a)));
  };
  const uint32_t e0 = wave * uint32_t(kPairEdges);
  const uint32_t e = e0 + uint32_t(el);
  bool edge_valid;
  {
    const EdgeIO* io = edge_io();
    const uint32_t B = io->d_B ? *io->d_B : io->B;
    if (e0 >= B) return;
    edge_valid = e < B;
  }
  const bool writer = edge_valid && h == 0;  // the lane that exports the edge's results
  const uint32_t slot_c = edge_valid ? e : e0;  // idle slots shadow the wave's first edge, results discarded
  // the edge this slot stands for: itself, or (later phases of a split launch) an entry of the survivors' list
  const uint32_t ec = edge_io()->edge_ids ? edge_io()->edge_ids[slot_c] : slot_c;
  PairWsRef ws;
  ws.rsrc = __builtin_amdgcn_make_buffer_rsrc(
      pair_args()->ws_all + (uint64_t(blockIdx.y) * gridDim.x + blockIdx.x) * uint64_t(W_::SLOTS * 64), 0,
      W_::SLOTS * 512, 0x00020000);
  ws.voff = lane * 8;
  auto source_row = [&](const EdgeIO* io) -> uint32_t {
    return io->src_idx ? io->src_idx[ec] : ((io->d_src_first ? *io->d_src_first : 0u) + ec);
  };
  {
    const EdgeIO* io = edge_io();
    const uint32_t si = source_row(io);
    const uint64_t trow = (io->d_tgt_off ? uint64_t(*io->d_tgt_off) : 0ull) + ec;
    // a later phase of a split launch starts from the state the phase before left in its x_out
    const double* __restrict__ a_row = io->resume ? io->resume + uint64_t(ec) * D : io->src + uint64_t(si) * io->src_stride;
    const double* __restrict__ b_row = io->tgt + trow * io->tgt_stride;
    double* __restrict__ record = (writer && !io->resume) ? io->record : nullptr;
    const int record_stride = io->record_stride;
#pragma unroll 1
    for (int d = 0; d < D; ++d) {
      const double av = a_row[d];
      ws_st(ws, W_::X + d, av);
      ws_st(ws, W_::B + d, b_row[d]);
      if (record) record[(uint64_t(ec) * record_stride + 0) * D + d] = av;
    }
  }

  const int k_first = int(pair_args()->gate.step0);
  uint32_t n_free = edge_io()->resume ? uint32_t(k_first) : 0u;  // survivors of the phase before: all its steps were free
  bool singular = false;
  bool alive = edge_valid;
  uint32_t n_exec = 0;  // steps integrated for this edge (KernelGate::steps_exec)
  const int n_steps = (pair_args()->gate.step1 < uint32_t(pair_args()->dyn.n_steps)) ? int(pair_args()->gate.step1)
                                                                                   : pair_args()->dyn.n_steps;
#pragma unroll 1
  for (int k = k_first; k < n_steps; ++k) {
    // distance(x_current, x_goal) > goal_proximity_threshold (exact left-to-right sum, vect_distance_metrics.hpp:126-137)
    {
      double s = 0.0;
#pragma unroll 1
      for (int d = 0; d < D; ++d) {
        const double xv = ws_ld(ws, W_::X + d);
        RKH_LD(L_::XE + d) = xv;  // both lanes of the edge write the same value
        const double df = xv - ws_ld(ws, W_::B + d);
        s = s + df * df;
      }
      if (!(sqrt(s) > pair_args()->dyn.goal_tol)) alive = false;
    }
    if (!__any(alive)) break;
    if (alive) ++n_exec;
    // PD law, zero-order hold over the step
    {
      PairArgP A = pair_args();
      const double kp = A->dyn.kp, kd = A->dyn.kd, u_max = A->dyn.u_max;
#pragma unroll 1
      for (int j = 0; j < N; ++j) {
        double v = kp * (ws_ld(ws, W_::B + 2 * j) - ws_ld(ws, W_::X + 2 * j)) +
                   kd * (ws_ld(ws, W_::B + 2 * j + 1) - ws_ld(ws, W_::X + 2 * j + 1));
        if (v > u_max) v = u_max;
        else if (v < -u_max) v = -u_max;
        RKH_LD(L_::U + j) = v;
      }
    }
    // runge_kutta4_integrate_impl (runge_kutta4_integrator_sys.hpp:53-97): the four useful f-evals per inner step as
    // the stages of a rolled loop (one copy of the dynamics in the instruction stream)
    bool sing_now = false;
    const int n_evals = 4 * int(pair_args()->dyn.inner[k]);
#pragma unroll 1
    for (int ev = 0; ev < n_evals; ++ev) {
      double qdd[N];
      pair_state_derivative<N>(sc, lds, el, h, qdd, sing_now);
      const int stage = ev & 3;
      const double h_dt = pair_args()->dyn.dt;
      // lane h advances the joints 2 jr + h: components 2j (q: derivative = qd of the differentiated state) and
      // 2j + 1 (qd: derivative = qdd)
#pragma unroll
      for (int jr = 0; jr < R; ++jr) {
        const int j = 2 * jr + h;
        const bool jv = j < N;
        const int jc = jv ? j : 2 * jr;
        // (the odd joint's value goes through an opaque copy: a plain `h ? qdd[2jr+1] : qdd[2jr]` is turned into a
        // run-time-indexed read of the array, which puts the array into scratch)
        double qdd_odd = qdd[2 * jr + 1 < N ? 2 * jr + 1 : 2 * jr];
        asm volatile("" : "+v"(qdd_odd));
        const double qdd_j = h ? qdd_odd : qdd[2 * jr];
        const double xq = RKH_LD(L_::XE + 2 * jc), xqd = RKH_LD(L_::XE + 2 * jc + 1);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const int sl = 2 * jr + half;  // this lane's private slot
          const double xv = half ? xqd : xq;
          const double dp = half ? qdd_j : xqd;
          double xn;
          if (stage == 0) {
            const double k1 = h_dt * dp;
            ws_st(ws, W_::W + sl, xv);
            ws_st(ws, W_::KA + sl, k1);
            xn = xv + 0.5 * k1;
          } else if (stage == 1) {
            const double k2 = h_dt * dp;
            const double k1 = ws_ld(ws, W_::KA + sl);
            ws_st(ws, W_::KA + sl, (1.0 / 6.0) * k1 + (2.0 / 6.0) * k2);
            xn = ws_ld(ws, W_::W + sl) + 0.5 * k2;
          } else if (stage == 2) {
            const double k3v = h_dt * dp;
            ws_st(ws, W_::K3 + sl, k3v);
            xn = ws_ld(ws, W_::W + sl) + k3v;
          } else {
            xn = xv + ((ws_ld(ws, W_::KA + sl) + (h_dt / 6.0) * dp) - (2.0 / 3.0) * ws_ld(ws, W_::K3 + sl));
          }
          if (jv) RKH_LD(L_::XE + 2 * jc + half) = xn;
        }
      }
    }
    if (sing_now && alive) {
      singular = true;
      alive = false;
    }
    // is_free(x_next): hyperbox bounds (hyperbox_topology.hpp:178-189), then proximity
    {
      PairArgP A = pair_args();
      bool oob = false;
#pragma unroll 1
      for (int d = 0; d < D; ++d) {
        const double lo = A->dyn.lower[d], hi = A->dyn.upper[d], xv = RKH_LD(L_::XE + d);
        if (lo < hi) oob = oob || (xv < lo) || (xv > hi);
        else oob = oob || (xv > lo) || (xv < hi);
      }
      if (oob) alive = false;
    }
    if (!__any(alive)) break;
    if (!pair_proximity_free<N>(sc, lds, el, h, alive)) alive = false;
    if (alive) {
      ++n_free;
      const EdgeIO* io = edge_io();
      double* __restrict__ record = writer ? io->record : nullptr;
      const int record_stride = io->record_stride;
#pragma unroll 1
      for (int d = 0; d < D; ++d) {
        const double xv = RKH_LD(L_::XE + d);
        ws_st(ws, W_::X + d, xv);
        if (record) record[(uint64_t(ec) * record_stride + n_free) * D + d] = xv;
      }
    }
  }
  const EdgeIO* io = edge_io();
  if (singular && writer) atomicExch(io->err_flag, int(RKH_ERR_SINGULAR));
  const uint32_t si = source_row(io);
  const double* __restrict__ a_row = io->src + uint64_t(si) * io->src_stride;
  double s_ar = 0.0, s_ab = 0.0, s_rb = 0.0;
#pragma unroll 1
  for (int d = 0; d < D; ++d) {
    const double xv = ws_ld(ws, W_::X + d), av = a_row[d], bv = ws_ld(ws, W_::B + d);
    if (writer) io->x_out[uint64_t(ec) * D + d] = xv;
    const double d_ar = av - xv, d_ab = av - bv, d_rb = xv - bv;
    s_ar = s_ar + d_ar * d_ar;
    s_ab = s_ab + d_ab * d_ab;
    s_rb = s_rb + d_rb * d_rb;
  }
  if (writer) io->steps_free[ec] = n_free;
  if (pair_args()->gate.steps_exec) {
    uint32_t tot = writer ? n_exec : 0u;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) tot += __shfl_xor(tot, off, 64);
    if (lane == 0 && tot) atomicAdd(pair_args()->gate.steps_exec, (unsigned long long)tot);
  }
  if (io->mode != EDGE_PLAIN && writer) {
    const double n_ar = sqrt(s_ar), n_ab = sqrt(s_ab), n_rb = sqrt(s_rb);
    if (io->mode == EDGE_STEER_ACCEPT) {
      // planning_visitor_base::steer_towards_position (planning_visitors.hpp:349-360)
      const double best_case = io->best_case ? io->best_case[ec] : n_ab;
      const bool ok = (!isinf(n_ar)) && (n_ar < 2.0 * best_case) && (n_ar > io->steer_tol * best_case);
      io->accept[ec] = ok ? 1 : 0;
    } else {
      // C_free distance used by the goal probe (MEAQR_topology.hpp:995-1003)
      io->goal_dist[si - 1] = (n_ab * 0.05 > n_rb) ? n_ab : INFINITY;
    }
  }
}

// ---- live edges in full waves: the pool kernel and the one-step-per-launch kernel ------------------------------------
// The same edge arithmetic as propagate_pair_kernel, cut at every step of the steer loop (MEAQR_topology.hpp:503-565:
// integrate one step, test is_free, stop at the first state that is not free).  The edges of a planner round are
// short-lived (tests/diag_edge_lifetimes.py: a candidate survives 9.7 of its 20 steps on average, 21 % end in the
// first), and a wave that carries 32 edges through all of their steps keeps the lanes of the edges that ended idle:
// about half of the lane-steps of propagate_pair_kernel do nothing.  Two launch forms of ONE kernel keep the lanes busy:
//   * POOL form (PairStepArgs::pool_cursor set; the first launch of a round): a resident set of waves; whenever an edge
//     of a wave ends -- accept test / goal probe done -- its lane pair takes the next edge of the round from a global
//     cursor (one atomic per wave and step; entry i of the round is found by bisection in the exclusive prefix of the
//     segments' edge counts, round_begin_kernel's edge_base), so a wave holds edges at DIFFERENT steps side by side
//     (the step index only selects dyn.inner[k], which must be uniform for this form).  No wave ever waits for
//     another: nothing can hang.  Once the cursor is exhausted a wave that is less than half full hands its live edges
//     -- (segment, edge, step); the state is in the edge's x_out row -- to the orphan list and leaves.
//   * LIST form (the launches after it): every launch advances every entry of its input list by one step, 32 per wave
//     whatever (problem, candidates | probes) segment they belong to, and appends the survivors to the next list;
//     with `step` = kStepFromEntry each entry carries its own step (the orphans), otherwise all are at step `step`
//     (the pure step-wise form, RKH_STEER_POOL=0: one launch per step, step 0 reading the implicit list).
// Between two steps an edge lives in its x_out row (the state after its last free step).  The order of a list is
// irrelevant: edges are independent and their results are indexed by the edge.
constexpr uint32_t kStepFromEntry = 0xFFFFFFFFu;
constexpr uint32_t kEntryStepShift = 20;  // list entry .x = segment | step << 20 (segment = 2 problem + group < 2^20)
struct PairStepArgs {
  const SceneDev* sc;
  DynDev dyn;
  const EdgeIO* tab_a;        // candidates of problem p
  const EdgeIO* tab_b;        // goal probes of problem p
  double* ws_all;             // RK4 stage vectors, PairStepWs slots x 64 lanes per block of the grid
  KernelGate gate;            // count / lo / hi: the planner's per-round choice between the mappings
  const uint32_t* edge_base;  // [n_segments + 1] exclusive prefix of the edges per segment (segment 2p + g)
  uint32_t n_segments;
  uint32_t step;              // LIST form: the step this launch integrates, or kStepFromEntry
  const uint2* list_in;       // LIST form, step > 0: (segment | step << 20, edge) of the edges that are still alive
  const uint32_t* cnt_in;     //            their number
  uint2* list_out;            // survivors of this launch (LIST form) / orphans (POOL form)
  uint32_t* cnt_out;
  unsigned long long* steps_exec;  // optional: + the number of edge-steps this launch integrated
  uint32_t* pool_cursor;      // POOL form: next unclaimed entry of the round (zero when the launch starts)
  uint32_t min_live;          // POOL form: below this many live edges (cursor exhausted) a wave hands them over
};
typedef const __attribute__((address_space(4))) PairStepArgs* PairStepArgP;
RKH_DI PairStepArgP pair_step_args() {
  PairStepArgP a = (PairStepArgP)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(a));
  return a;
}
template <int N>
struct PairStepWs {  // per-lane private RK4 values (this lane's joints: 2 components each)
  enum : int { R = (N + 1) / 2, W = 0, KA = 2 * R, K3 = 4 * R, SLOTS = 6 * R };
};

template <int N>
__global__ __launch_bounds__(64, 2) void propagate_pair_step_kernel(PairStepArgs) {
  __shared__ PairLds<N> lds;
  typedef PairLayout<N> L_;
  typedef PairStepWs<N> W_;
  constexpr int R = L_::R;
  constexpr int D = 2 * N;
  const int lane = threadIdx.x;
  const int h = lane & 1;
  const int el = lane >> 1;
  uint32_t n_total;
  bool pool;
  {
    PairStepArgP A = pair_step_args();
    if (A->gate.count) {
      const uint32_t c = *A->gate.count;
      if (c < A->gate.lo || c >= A->gate.hi) return;
    }
    pool = A->pool_cursor != nullptr;
    n_total = (pool || A->step == 0u) ? A->edge_base[A->n_segments] : *A->cnt_in;
  }
  if (blockIdx.x * uint32_t(kPairEdges) >= n_total) return;
  const ScenePtr sc = (ScenePtr)pair_step_args()->sc;
  if (threadIdx.x < 3 * (N + 1)) {
    const int jj = threadIdx.x / 3;
    lds.axis[jj][threadIdx.x % 3] = sc->joints[jj < N ? jj : N - 1].axis[threadIdx.x % 3];
  }
  __syncthreads();
  PairWsRef ws;
  ws.rsrc = __builtin_amdgcn_make_buffer_rsrc(pair_step_args()->ws_all + uint64_t(blockIdx.x) * uint64_t(W_::SLOTS * 64), 0,
                                              W_::SLOTS * 512, 0x00020000);
  ws.voff = lane * 8;
  const int n_steps = pair_step_args()->dyn.n_steps;
  // entry i of the round's implicit list -> (segment, edge): bisection in the prefix of the segments' edge counts
  auto implicit_entry = [&](uint32_t i, uint32_t* seg_out, uint32_t* ec_out) {
    const uint32_t* eb = pair_step_args()->edge_base;
    uint32_t lo = 0, hi = pair_step_args()->n_segments;  // eb[lo] <= i < eb[hi]
    while (hi - lo > 1) {
      const uint32_t mid = (lo + hi) >> 1;
      if (eb[mid] <= i) lo = mid;
      else hi = mid;
    }
    *seg_out = lo;
    *ec_out = i - eb[lo];
  };
  // the lane pair's edge (both lanes of a pair hold the same values)
  bool has = false;       // POOL form: this lane pair carries an edge
  uint32_t seg = 0, ec = 0;
  int k = 0;
  bool pool_empty = false;  // uniform
  uint32_t chunk = blockIdx.x;
#pragma unroll 1
  for (;;) {
    bool edge_valid;
    if (pool) {
      // ---- refill: lane pairs without an edge take the next entries of the round
      const unsigned long long need = __ballot(!has && h == 0);
      if (!pool_empty && need) {
        PairStepArgP A = pair_step_args();
        const uint32_t cnt = uint32_t(__popcll(need));
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(A->pool_cursor, cnt);
        base = uint32_t(__builtin_amdgcn_readfirstlane(int(base)));
        const uint32_t mine = base + uint32_t(__popcll(need & ((1ull << (lane & 62)) - 1ull)));  // same on both lanes of a pair
        if (!has && mine < n_total) {
          implicit_entry(mine, &seg, &ec);
          k = 0;
          has = true;
        }
        if (base + cnt >= n_total) pool_empty = true;
      }
      const unsigned long long livem = __ballot(has && h == 0);
      if (!livem) break;
      PairStepArgP A = pair_step_args();
      if (pool_empty && A->list_out && uint32_t(__popcll(livem)) < A->min_live) {
        // less than half full and nothing left to take: the live edges go on in full waves of the launches that follow
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(A->cnt_out, uint32_t(__popcll(livem)));
        base = uint32_t(__builtin_amdgcn_readfirstlane(int(base)));
        if (has && h == 0)
          A->list_out[base + uint32_t(__popcll(livem & ((1ull << lane) - 1ull)))] =
              make_uint2(seg | (uint32_t(k) << kEntryStepShift), ec);
        break;
      }
      edge_valid = has;
      {  // idle lane pairs shadow the wave's first live edge, results discarded (uniform reads: every lane executes them)
        const int src = __builtin_ctzll(livem);
        const uint32_t s_seg = uint32_t(__builtin_amdgcn_readlane(int(seg), src));
        const uint32_t s_ec = uint32_t(__builtin_amdgcn_readlane(int(ec), src));
        const int s_k = __builtin_amdgcn_readlane(k, src);
        if (!has) {
          seg = s_seg;
          ec = s_ec;
          k = s_k;
        }
      }
    } else {
      const uint32_t i0 = chunk * uint32_t(kPairEdges);
      if (i0 >= n_total) break;
      chunk += gridDim.x;
      edge_valid = i0 + uint32_t(el) < n_total;
      const uint32_t ic = edge_valid ? i0 + uint32_t(el) : i0;  // idle slots shadow the chunk's first edge
      const uint32_t step = pair_step_args()->step;
      if (step == 0u) {
        implicit_entry(ic, &seg, &ec);
        k = 0;
      } else {
        const uint2 ent = pair_step_args()->list_in[ic];
        seg = ent.x & ((1u << kEntryStepShift) - 1u);
        ec = ent.y;
        k = step == kStepFromEntry ? int(ent.x >> kEntryStepShift) : int(step);
      }
    }
    const bool writer = edge_valid && h == 0;  // the lane that exports the edge's results
    auto edge_io = [&]() -> const EdgeIO* {
      PairStepArgP A = pair_step_args();
      return ((seg & 1u) ? A->tab_b : A->tab_a) + (seg >> 1);
    };
    auto source_row = [&](const EdgeIO* io) -> uint32_t {
      return io->src_idx ? io->src_idx[ec] : ((io->d_src_first ? *io->d_src_first : 0u) + ec);
    };
    auto target_row = [&](const EdgeIO* io) -> const double* {
      return io->tgt + ((io->d_tgt_off ? uint64_t(*io->d_tgt_off) : 0ull) + ec) * io->tgt_stride;
    };
    // the state the edge has reached (its source row, or what the step before left in x_out), the steer target, and
    // distance(x_current, x_goal) > goal_proximity_threshold (exact left-to-right sum, vect_distance_metrics.hpp:126-137)
    bool alive = edge_valid;
    {
      const EdgeIO* io = edge_io();
      const double* __restrict__ a_row = k ? io->x_out + uint64_t(ec) * D : io->src + uint64_t(source_row(io)) * io->src_stride;
      const double* __restrict__ b_row = target_row(io);
      double* __restrict__ first = (k == 0 && writer) ? io->x_out + uint64_t(ec) * D : nullptr;
      double* __restrict__ record = (k == 0 && writer) ? io->record : nullptr;
      const int record_stride = io->record_stride;
      double s = 0.0;
#pragma unroll 1
      for (int d = 0; d < D; ++d) {
        const double xv = a_row[d];
        RKH_LD(L_::XE + d) = xv;  // both lanes of the edge write the same value
        if (first) first[d] = xv;  // an edge that ends in its first step stays at its source
        if (record) record[(uint64_t(ec) * record_stride + 0) * D + d] = xv;
        const double df = xv - b_row[d];
        s = s + df * df;
      }
      if (!(sqrt(s) > pair_step_args()->dyn.goal_tol)) alive = false;
    }
    bool singular = false;
    if (__any(alive)) {
      if (pair_step_args()->steps_exec) {
        const unsigned long long m = __ballot(alive && h == 0);
        if (lane == 0) atomicAdd(pair_step_args()->steps_exec, (unsigned long long)__popcll(m));
      }
      // PD law, zero-order hold over the step
      {
        PairStepArgP A = pair_step_args();
        const double kp = A->dyn.kp, kd = A->dyn.kd, u_max = A->dyn.u_max;
        const double* __restrict__ b_row = target_row(edge_io());
#pragma unroll 1
        for (int j = 0; j < N; ++j) {
          double v = kp * (b_row[2 * j] - RKH_LD(L_::XE + 2 * j)) + kd * (b_row[2 * j + 1] - RKH_LD(L_::XE + 2 * j + 1));
          if (v > u_max) v = u_max;
          else if (v < -u_max) v = -u_max;
          RKH_LD(L_::U + j) = v;
        }
      }
      // runge_kutta4_integrate_impl (runge_kutta4_integrator_sys.hpp:53-97), see propagate_pair_kernel.  The POOL form
      // and the orphans' launches mix steps in a wave: their inner loop counts are equal (checked by the host).
      bool sing_now = false;
      const int k_uniform = __builtin_amdgcn_readfirstlane(k);
      const int n_evals = 4 * int(pair_step_args()->dyn.inner[k_uniform]);
#pragma unroll 1
      for (int ev = 0; ev < n_evals; ++ev) {
        double qdd[N];
        pair_state_derivative<N>(sc, lds, el, h, qdd, sing_now);
        const int stage = ev & 3;
        const double h_dt = pair_step_args()->dyn.dt;
#pragma unroll
        for (int jr = 0; jr < R; ++jr) {
          const int j = 2 * jr + h;
          const bool jv = j < N;
          const int jc = jv ? j : 2 * jr;
          double qdd_odd = qdd[2 * jr + 1 < N ? 2 * jr + 1 : 2 * jr];
          asm volatile("" : "+v"(qdd_odd));
          const double qdd_j = h ? qdd_odd : qdd[2 * jr];
          const double xq = RKH_LD(L_::XE + 2 * jc), xqd = RKH_LD(L_::XE + 2 * jc + 1);
#pragma unroll
          for (int half = 0; half < 2; ++half) {
            const int sl = 2 * jr + half;  // this lane's private slot
            const double xv = half ? xqd : xq;
            const double dp = half ? qdd_j : xqd;
            double xn;
            if (stage == 0) {
              const double k1 = h_dt * dp;
              ws_st(ws, W_::W + sl, xv);
              ws_st(ws, W_::KA + sl, k1);
              xn = xv + 0.5 * k1;
            } else if (stage == 1) {
              const double k2 = h_dt * dp;
              const double k1 = ws_ld(ws, W_::KA + sl);
              ws_st(ws, W_::KA + sl, (1.0 / 6.0) * k1 + (2.0 / 6.0) * k2);
              xn = ws_ld(ws, W_::W + sl) + 0.5 * k2;
            } else if (stage == 2) {
              const double k3v = h_dt * dp;
              ws_st(ws, W_::K3 + sl, k3v);
              xn = ws_ld(ws, W_::W + sl) + k3v;
            } else {
              xn = xv + ((ws_ld(ws, W_::KA + sl) + (h_dt / 6.0) * dp) - (2.0 / 3.0) * ws_ld(ws, W_::K3 + sl));
            }
            if (jv) RKH_LD(L_::XE + 2 * jc + half) = xn;
          }
        }
      }
      if (sing_now && alive) {
        singular = true;
        alive = false;
      }
      // is_free(x_next): hyperbox bounds (hyperbox_topology.hpp:178-189), then proximity
      {
        PairStepArgP A = pair_step_args();
        bool oob = false;
#pragma unroll 1
        for (int d = 0; d < D; ++d) {
          const double lo = A->dyn.lower[d], hi = A->dyn.upper[d], xv = RKH_LD(L_::XE + d);
          if (lo < hi) oob = oob || (xv < lo) || (xv > hi);
          else oob = oob || (xv > lo) || (xv < hi);
        }
        if (oob) alive = false;
      }
      if (__any(alive)) {
        if (!pair_proximity_free<N>(sc, lds, el, h, alive)) alive = false;
      }
    }
    // ---- the step's outcome: alive = step k was free (the edge stands at the new state), else it stays where it was
    const bool go_on = alive && (k + 1 < n_steps);
    const bool finished = edge_valid && !go_on;
    const uint32_t n_free = uint32_t(k) + (alive ? 1u : 0u);
    if (alive) {
      const EdgeIO* io = edge_io();
      double* __restrict__ xo = writer ? io->x_out + uint64_t(ec) * D : nullptr;
      double* __restrict__ record = writer ? io->record : nullptr;
      const int record_stride = io->record_stride;
      if (xo) {
#pragma unroll 1
        for (int d = 0; d < D; ++d) {
          const double xv = RKH_LD(L_::XE + d);
          xo[d] = xv;
          if (record) record[(uint64_t(ec) * record_stride + n_free) * D + d] = xv;
        }
      }
    }
    if (!pool) {  // survivors -> the next launch's list
      const unsigned long long m = __ballot(go_on && h == 0);
      if (m) {
        PairStepArgP A = pair_step_args();
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(A->cnt_out, uint32_t(__popcll(m)));
        base = uint32_t(__builtin_amdgcn_readfirstlane(int(base)));
        if (go_on && h == 0)
          A->list_out[base + uint32_t(__popcll(m & ((1ull << lane) - 1ull)))] =
              make_uint2(seg | (uint32_t(k + 1) << kEntryStepShift), ec);
      }
    }
    if (__any(finished)) {  // accept test / goal probe of the edges that end here
      const EdgeIO* io = edge_io();
      if (singular && writer) atomicExch(io->err_flag, int(RKH_ERR_SINGULAR));
      const uint32_t si = source_row(io);
      const double* __restrict__ a_row = io->src + uint64_t(si) * io->src_stride;
      const double* __restrict__ b_row = target_row(io);
      const double* __restrict__ x_row = io->x_out + uint64_t(ec) * D;  // written above (alive) or by an earlier step
      double s_ar = 0.0, s_ab = 0.0, s_rb = 0.0;
#pragma unroll 1
      for (int d = 0; d < D; ++d) {
        const double av = a_row[d], bv = b_row[d];
        const double xv = alive ? RKH_LD(L_::XE + d) : (k ? x_row[d] : av);
        const double d_ar = av - xv, d_ab = av - bv, d_rb = xv - bv;
        s_ar = s_ar + d_ar * d_ar;
        s_ab = s_ab + d_ab * d_ab;
        s_rb = s_rb + d_rb * d_rb;
      }
      if (finished && writer) {
        io->steps_free[ec] = n_free;
        if (io->mode != EDGE_PLAIN) {
          const double n_ar = sqrt(s_ar), n_ab = sqrt(s_ab), n_rb = sqrt(s_rb);
          if (io->mode == EDGE_STEER_ACCEPT) {
            // planning_visitor_base::steer_towards_position (planning_visitors.hpp:349-360)
            const double best_case = io->best_case ? io->best_case[ec] : n_ab;
            const bool ok = (!isinf(n_ar)) && (n_ar < 2.0 * best_case) && (n_ar > io->steer_tol * best_case);
            io->accept[ec] = ok ? 1 : 0;
          } else {
            // C_free distance used by the goal probe (MEAQR_topology.hpp:995-1003)
            io->goal_dist[si - 1] = (n_ab * 0.05 > n_rb) ? n_ab : INFINITY;
          }
        }
      }
    }
    if (pool) {  // the lane pair's next state: on to the next step, or free for a new edge
      if (edge_valid && go_on) ++k;
      else has = false;
    }
  }
}

template <int N>
static void launch_pair_step_t(hipStream_t s, const PairStepArgs& args, uint32_t blocks) {
  hipLaunchKernelGGL((propagate_pair_step_kernel<N>), dim3(blocks), dim3(64), 0, s, args);
}
static rkh_status launch_pair_step_n(hipStream_t s, int n_dof, const PairStepArgs& args, uint32_t blocks) {
  switch (n_dof) {
    case 1: launch_pair_step_t<1>(s, args, blocks); break;
    case 2: launch_pair_step_t<2>(s, args, blocks); break;
    case 3: launch_pair_step_t<3>(s, args, blocks); break;
    case 4: launch_pair_step_t<4>(s, args, blocks); break;
    case 6: launch_pair_step_t<6>(s, args, blocks); break;
    case 7: launch_pair_step_t<7>(s, args, blocks); break;
    default:
      set_error("propagate: chains with this number of joints are not instantiated (1,2,3,4,6,7)");
      return RKH_ERR_UNSUPPORTED;
  }
  return RKH_OK;
}

size_t propagate_pair_step_workspace_bytes(int n_dof, uint32_t blocks) {
  return size_t(blocks) * size_t(6 * ((n_dof + 1) / 2)) * 64 * sizeof(double);
}

// The steer launches of a round over two ping-pong lists; d_cnt[k] = entries of the list launch k reads (d_cnt[1 ..
// n_steps + 1] and *d_pool_cursor must be zero when the first launch starts: round_begin_kernel clears them).  `blocks` bounds the
// grids; a LIST launch with more chunks than blocks strides over them.  pool_blocks > 0: the POOL form first (that many
// resident waves), then n_steps - 1 LIST launches for the edges it handed over (each carries its own step; a launch
// whose list is empty returns at once); pool_blocks == 0 (or a step schedule whose inner counts differ): one LIST
// launch per step.
rkh_status launch_propagate_pair_steps(hipStream_t s, int n_dof, const SceneDev* d_scene, const DynDev& dyn,
                                       const EdgeIO* tab_a, const EdgeIO* tab_b, uint32_t n_problems,
                                       const uint32_t* d_edge_base, uint2* d_list0, uint2* d_list1, uint32_t* d_cnt,
                                       double* d_ws, uint32_t blocks, KernelGate gate, unsigned long long* d_steps_exec,
                                       uint32_t pool_blocks, uint32_t* d_pool_cursor) {
  if (blocks == 0 || n_problems == 0) return RKH_OK;
  PairStepArgs args;
  args.sc = d_scene;
  args.dyn = dyn;
  args.tab_a = tab_a;
  args.tab_b = tab_b;
  args.ws_all = d_ws;
  args.gate = gate;
  args.edge_base = d_edge_base;
  args.n_segments = 2 * n_problems;
  args.steps_exec = d_steps_exec;
  args.pool_cursor = nullptr;
  args.min_live = 0;
  bool uniform_inner = true;
  for (int k = 1; k < dyn.n_steps; ++k) uniform_inner = uniform_inner && dyn.inner[k] == dyn.inner[0];
  const bool use_pool = pool_blocks > 0 && d_pool_cursor && uniform_inner && (2u * n_problems) < (1u << kEntryStepShift);
  // (an edge the pool form hands over before its FIRST step still has n_steps steps to go: one launch more than steps)
  const int n_launches = use_pool ? dyn.n_steps + 1 : dyn.n_steps;
  for (int k = 0; k < n_launches; ++k) {
    args.list_in = (k & 1) ? d_list1 : d_list0;
    args.list_out = (k & 1) ? d_list0 : d_list1;
    args.cnt_in = d_cnt + k;
    args.cnt_out = d_cnt + k + 1;
    rkh_status st;
    if (use_pool && k == 0) {
      args.step = 0;
      args.pool_cursor = d_pool_cursor;
      args.min_live = uint32_t(kPairEdges) / 2;
      st = launch_pair_step_n(s, n_dof, args, std::min(blocks, pool_blocks));
      args.pool_cursor = nullptr;
    } else {
      args.step = use_pool ? kStepFromEntry : uint32_t(k);
      st = launch_pair_step_n(s, n_dof, args, blocks);
    }
    if (st != RKH_OK) return st;
  }
  RKH_HIP(hipGetLastError());
  return RKH_OK;
}

// Diagnostic kernel (not on the product path): `iters` back-to-back f-evals + proximity tests of kPairEdges states per
// wave with cycle counts per phase: [frames + sincos, jacobian columns, mass matrix, force sweep, assembly + cholesky,
// proximity: joint frames, cull, exact routines]
template <int N>
__global__ __launch_bounds__(64, 2) void pair_cycles_kernel(const SceneDev* __restrict__ sc, const double* __restrict__ x,
                                                             const double* __restrict__ u, uint32_t B, int iters,
                                                             unsigned long long* __restrict__ out,
                                                             double* __restrict__ sink_out) {
  __shared__ PairLds<N> lds;
  if (threadIdx.x < 3 * (N + 1)) {
    const int jj = threadIdx.x / 3;
    lds.axis[jj][threadIdx.x % 3] = sc->joints[jj < N ? jj : N - 1].axis[threadIdx.x % 3];
  }
  __syncthreads();
  typedef PairLayout<N> L_;
  const int lane = threadIdx.x, h = lane & 1, el = lane >> 1;
  uint32_t e = blockIdx.x * kPairEdges + el;
  if (e >= B) e = blockIdx.x * kPairEdges;
  for (int d = 0; d < 2 * N; ++d) RKH_LD(L_::XE + d) = x[uint64_t(e) * 2 * N + d];
  for (int j = 0; j < N; ++j) RKH_LD(L_::U + j) = u[uint64_t(e) * N + j];
  unsigned long long st[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  bool singular = false;
  double accv = 0.0;
  for (int it = 0; it < iters; ++it) {
    double qdd[N];
    pair_state_derivative<N, true>((ScenePtr)sc, lds, el, h, qdd, singular, st);
#pragma unroll
    for (int j = 0; j < N; ++j) {
      accv += qdd[j];
      RKH_LD(L_::XE + 2 * j + 1) = RKH_LD(L_::XE + 2 * j + 1) + 1e-4 * qdd[j];
    }
    accv += pair_proximity_free<N, true>((ScenePtr)sc, lds, el, h, true, st) ? 1.0 : 0.0;
  }
  if (lane == 0) {
    for (int i = 0; i < 8; ++i) out[blockIdx.x * 8 + i] = st[i];
    sink_out[blockIdx.x] = accv + (singular ? 1.0 : 0.0);
  }
}

// Diagnostic kernel (not on the product path): the proximity test of B states, 32 per wave, counting what survives each
// stage: out[0] += states tested, [1] += (robot shape, obstacle) pairs past the static reach + fp32 cull (the closed-form
// stage's input), [2] += closed forms evaluated, [3] += golden-section searches (capped cylinder / box), [4] += states
// found in collision.
template <int N>
__global__ __launch_bounds__(64, 2) void pair_counts_kernel(const SceneDev* __restrict__ sc, const double* __restrict__ x,
                                                             uint32_t B, unsigned long long* __restrict__ out) {
  __shared__ PairLds<N> lds;
  if (threadIdx.x < 3 * (N + 1)) {
    const int jj = threadIdx.x / 3;
    lds.axis[jj][threadIdx.x % 3] = sc->joints[jj < N ? jj : N - 1].axis[threadIdx.x % 3];
  }
  __syncthreads();
  typedef PairLayout<N> L_;
  const int lane = threadIdx.x, h = lane & 1, el = lane >> 1;
  const uint32_t e = blockIdx.x * kPairEdges + el;
  const bool valid = e < B;
  const uint32_t es = valid ? e : blockIdx.x * kPairEdges;
  for (int d = 0; d < 2 * N; ++d) RKH_LD(L_::XE + d) = x[uint64_t(es) * 2 * N + d];
  unsigned long long st[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const bool free_state = pair_proximity_free<N, true>((ScenePtr)sc, lds, el, h, true, st);
  const unsigned long long hits = __ballot(!free_state && valid && h == 0);
  if (lane == 0) {
    const uint32_t n_here = (B - blockIdx.x * kPairEdges) < uint32_t(kPairEdges) ? (B - blockIdx.x * kPairEdges) : uint32_t(kPairEdges);
    atomicAdd(&out[0], (unsigned long long)n_here);
    atomicAdd(&out[1], st[8]);
    atomicAdd(&out[2], st[9]);
    atomicAdd(&out[3], st[10]);
    atomicAdd(&out[4], (unsigned long long)__popcll(hits));
  }
}

rkh_status launch_pair_counts(hipStream_t s, int n_dof, const SceneDev* d_scene, const double* d_x, uint32_t B,
                              unsigned long long* d_out) {
  const uint32_t waves = (B + kPairEdges - 1) / kPairEdges;
  switch (n_dof) {
    case 6: hipLaunchKernelGGL((pair_counts_kernel<6>), dim3(waves), dim3(64), 0, s, d_scene, d_x, B, d_out); break;
    case 3: hipLaunchKernelGGL((pair_counts_kernel<3>), dim3(waves), dim3(64), 0, s, d_scene, d_x, B, d_out); break;
    default: set_error("pair diagnostics: instantiated for 3 and 6 joints"); return RKH_ERR_UNSUPPORTED;
  }
  RKH_HIP(hipGetLastError());
  return RKH_OK;
}

rkh_status launch_pair_cycles(hipStream_t s, int n_dof, const SceneDev* d_scene, const double* d_x, const double* d_u,
                              uint32_t B, int iters, unsigned long long* d_out, double* d_sink) {
  const uint32_t waves = (B + kPairEdges - 1) / kPairEdges;
  switch (n_dof) {
    case 6: hipLaunchKernelGGL((pair_cycles_kernel<6>), dim3(waves), dim3(64), 0, s, d_scene, d_x, d_u, B, iters, d_out, d_sink); break;
    case 3: hipLaunchKernelGGL((pair_cycles_kernel<3>), dim3(waves), dim3(64), 0, s, d_scene, d_x, d_u, B, iters, d_out, d_sink); break;
    default: set_error("pair diagnostics: instantiated for 3 and 6 joints"); return RKH_ERR_UNSUPPORTED;
  }
  RKH_HIP(hipGetLastError());
  RKH_HIP(hipStreamSynchronize(s));
  return RKH_OK;
}

// bytes of workspace a launch of (edges_a + edges_b) edges per problem needs
size_t propagate_pairs_workspace_bytes(int n_dof, uint32_t edges_a, uint32_t edges_b, uint32_t n_problems) {
  const size_t waves = size_t((edges_a + kPairEdges - 1) / kPairEdges + (edges_b + kPairEdges - 1) / kPairEdges) * n_problems;
  const size_t slots = size_t(6 * ((n_dof + 1) / 2) + 4 * n_dof);
  return waves * slots * 64 * sizeof(double);
}

template <int N>
static void launch_pair_t(hipStream_t s, const SceneDev* d_scene, const DynDev& dyn, const EdgeIO& io, uint32_t edges_a,
                          const EdgeIO& io_b, uint32_t edges_b, const EdgeIO* tab_a, const EdgeIO* tab_b,
                          uint32_t n_problems, double* d_ws, KernelGate gate) {
  const uint32_t ga = (edges_a + kPairEdges - 1) / kPairEdges, gbk = (edges_b + kPairEdges - 1) / kPairEdges;
  PairArgs args;
  args.sc = d_scene;
  args.dyn = dyn;
  args.io_a = io;
  args.io_b = io_b;
  args.tab_a = tab_a;
  args.tab_b = tab_b;
  args.grid_a = ga;
  args.ws_all = d_ws;
  args.gate = gate;
  hipLaunchKernelGGL((propagate_pair_kernel<N>), dim3(ga + gbk, n_problems), dim3(64), 0, s, args);
}

rkh_status launch_propagate_pairs(hipStream_t s, int n_dof, const SceneDev* d_scene, const DynDev& dyn, const EdgeIO& io,
                                  uint32_t grid_edges, const EdgeIO* io_b, uint32_t grid_b, const EdgeIO* tab_a,
                                  const EdgeIO* tab_b, uint32_t n_problems, double* d_ws, KernelGate gate) {
  const uint32_t eb = (io_b || tab_b) ? grid_b : 0u;
  if (grid_edges + eb == 0 || n_problems == 0) return RKH_OK;
  if (!d_ws) {
    set_error("propagate (two lanes per edge): no workspace");
    return RKH_ERR_BAD_ARG;
  }
  const EdgeIO second = io_b ? *io_b : EdgeIO();
  switch (n_dof) {
    case 1: launch_pair_t<1>(s, d_scene, dyn, io, grid_edges, second, eb, tab_a, tab_b, n_problems, d_ws, gate); break;
    case 2: launch_pair_t<2>(s, d_scene, dyn, io, grid_edges, second, eb, tab_a, tab_b, n_problems, d_ws, gate); break;
    case 3: launch_pair_t<3>(s, d_scene, dyn, io, grid_edges, second, eb, tab_a, tab_b, n_problems, d_ws, gate); break;
    case 4: launch_pair_t<4>(s, d_scene, dyn, io, grid_edges, second, eb, tab_a, tab_b, n_problems, d_ws, gate); break;
    case 6: launch_pair_t<6>(s, d_scene, dyn, io, grid_edges, second, eb, tab_a, tab_b, n_problems, d_ws, gate); break;
    case 7: launch_pair_t<7>(s, d_scene, dyn, io, grid_edges, second, eb, tab_a, tab_b, n_problems, d_ws, gate); break;
    default:
      set_error("propagate: chains with this number of joints are not instantiated (1,2,3,4,6,7)");
      return RKH_ERR_UNSUPPORTED;
  }
  RKH_HIP(hipGetLastError());
  return RKH_OK;
}

uint32_t pair_kernel_edges_per_wave() { return uint32_t(kPairEdges); }

// resident waves per CU of the kernel for this chain size
uint32_t pair_kernel_waves_per_cu(int n_dof) {
  int blocks = 0;
  hipError_t e = hipErrorInvalidValue;
  switch (n_dof) {
    case 1: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, propagate_pair_kernel<1>, 64, 0); break;
    case 2: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, propagate_pair_kernel<2>, 64, 0); break;
    case 3: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, propagate_pair_kernel<3>, 64, 0); break;
    case 4: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, propagate_pair_kernel<4>, 64, 0); break;
    case 6: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, propagate_pair_kernel<6>, 64, 0); break;
    case 7: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, propagate_pair_kernel<7>, 64, 0); break;
    default: break;
  }
  return (e == hipSuccess && blocks > 0) ? uint32_t(blocks) : 8u;
}

#undef RKH_LD
}  // namespace rkh
