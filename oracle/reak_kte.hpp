// ORACLE -- TEST INFRASTRUCTURE ONLY (see reak_math.hpp header).
//
// CPU restatement of ReaK's KTE-chain forward dynamics:  kte_map_chain passes, the KTE elements
// of the BASELINE chains, mass_matrix_calc, and kte_nl_system::get_state_derivative.
// The chain is an interpreted list of rkh_kte_op, executed exactly like
// kte_map_chain::{doMotion,clearForce,doForce} (ctrl/mbd_kte/kte_map_chain.hpp:71-89).
//
// Parity pin status: the 1-link pendulum of ctrl/mbd_kte/test_bm.cpp:45-77 (M = m L^2, analytic
// bias force) and a planar 2R arm closed form pin M(q) and f(q,qd,u) (tests/test_oracle_kat.py).
#ifndef REAK_ORACLE_KTE_HPP
#define REAK_ORACLE_KTE_HPP

#include <cstdint>
#include <vector>

#include "../include/rkh_types.h"
#include "reak_math.hpp"
#include "reak_planar.hpp"

namespace oracle {

// gen_coord<double> (core/kinetostatics/gen_coord.hpp)
struct GenCoord {
  double q = 0, q_dot = 0, q_ddot = 0, f = 0;
};

// jacobian_gen_3D<double> (core/kinetostatics/motion_jacobians.hpp:205-236); Parent is a frame index.
struct JacGen3D {
  int Parent = -1;
  V3 qd_vel, qd_avel, qd_acc, qd_aacc;
};

inline Pose to_pose(const rkh_pose& p) {
  Pose r;
  r.Position = V3(p.pos[0], p.pos[1], p.pos[2]);
  r.Q = Quat(p.quat[0], p.quat[1], p.quat[2], p.quat[3]);
  return r;
}

struct KteChain {
  std::vector<rkh_kte_op> ops;
  rkh_chain_base base;
  int n_coords = 0, n_frames = 0;
  std::vector<GenCoord> coords;
  std::vector<Frame> frames;
  std::vector<Frame2D> frames2; // planar chains: frame_2D per frame index
  bool planar = false;
  std::vector<double> drive;    // driving_actuator_gen::mDriveForce per coord (system input)
  std::vector<JacGen3D> jac;    // per coord: the revolute joint's mJacobian
  std::vector<int> gen_inertias, inertias_3D, inertias_2D;  // op indices, registration order of mass_matrix_calc
  std::vector<JacGen2D> jac2;   // per coord: the revolute_joint_2D's mJacobian
  uint64_t feval_flops = 0;     // not used for parity; see count in DESIGN.md

  KteChain() {}
  KteChain(const rkh_kte_op* prog, int n_ops, const rkh_chain_base& b) : ops(prog, prog + n_ops), base(b) {
    for (const auto& op : ops) {
      if (op.coord + 1 > n_coords) n_coords = op.coord + 1;
      if (op.base_frame + 1 > n_frames) n_frames = op.base_frame + 1;
      if (op.end_frame + 1 > n_frames) n_frames = op.end_frame + 1;
    }
    coords.assign(n_coords, GenCoord());
    frames.assign(n_frames, Frame());
    frames2.assign(n_frames, Frame2D());
    for (const auto& op : ops) planar = planar || op.kind == RKH_KTE_REVOLUTE_JOINT_2D;
    drive.assign(n_coords, 0.0);
    jac.assign(n_coords, JacGen3D());
    jac2.assign(n_coords, JacGen2D());
    for (int i = 0; i < n_ops; ++i) {
      if (ops[i].kind == RKH_KTE_INERTIA_GEN) gen_inertias.push_back(i);
      if (ops[i].kind == RKH_KTE_INERTIA_3D) inertias_3D.push_back(i);
      if (ops[i].kind == RKH_KTE_INERTIA_2D) inertias_2D.push_back(i);
    }
    reset_base();
  }

  void reset_base() {
    Frame f;
    f.Position = V3(base.pose.pos[0], base.pose.pos[1], base.pose.pos[2]);
    f.Q = Quat(base.pose.quat[0], base.pose.quat[1], base.pose.quat[2], base.pose.quat[3]);
    f.Acceleration = V3(base.acceleration[0], base.acceleration[1], base.acceleration[2]);
    frames[0] = f;
    frames2[0] = Frame2D();
    static_cast<Pose2&>(frames2[0]) = to_pose2(base.pose);
    frames2[0].Acceleration = V2(base.acceleration[0], base.acceleration[1]);  // gravity enters as base Acceleration
  }

  // kte_map_chain::doMotion: kte_map_chain.hpp:71-76
  void doMotion() {
    for (const auto& op : ops) {
      switch (op.kind) {
        case RKH_KTE_REVOLUTE_JOINT_3D: revolute_doMotion(op); break;
        case RKH_KTE_RIGID_LINK_3D: link_doMotion(op); break;
        case RKH_KTE_REVOLUTE_JOINT_2D: {  // revolute_joint_2D::doMotion, revolute_joint.cpp:30-56
          const Frame2D& B = frames2[op.base_frame];
          Frame2D& E = frames2[op.end_frame];
          E.Position = B.Position;
          E.Velocity = B.Velocity;
          E.Acceleration = B.Acceleration;
          E.Rotation = B.Rotation * rot_from_angle(coords[op.coord].q);
          E.AngVelocity = B.AngVelocity + coords[op.coord].q_dot;
          E.AngAcceleration = B.AngAcceleration + coords[op.coord].q_ddot;
          jac2[op.coord].Parent = op.end_frame;  // :50-55: qd_vel = 0, qd_avel = 1
          jac2[op.coord].qd_vel = V2();
          jac2[op.coord].qd_avel = 1.0;
          break;
        }
        case RKH_KTE_RIGID_LINK_2D: {  // rigid_link_2D::doMotion, rigid_link.cpp:87-99
          const Pose2 off = to_pose2(op.offset);
          const Frame2D B = frames2[op.base_frame];
          Frame2D& E = frames2[op.end_frame];
          E.Position = B.Position + B.Rotation * off.Position;
          E.Velocity = B.Velocity + B.Rotation * cross_sv(B.AngVelocity, off.Position);
          E.Acceleration = B.Acceleration + B.Rotation * ((-B.AngVelocity * B.AngVelocity) * off.Position + cross_sv(B.AngAcceleration, off.Position));
          E.Rotation = B.Rotation * off.Rotation;
          E.AngVelocity = B.AngVelocity;
          E.AngAcceleration = B.AngAcceleration;
          break;
        }
        default: break;  // inertia_*::doMotion (inertia.cpp:36-45,100-109) and actuators only store
      }
    }
  }
  // kte_map_chain::clearForce: kte_map_chain.hpp:85-89 (each element zeroes what it touches)
  void clearForce() {
    for (const auto& op : ops) {
      if (op.base_frame >= 0) { frames[op.base_frame].Force = V3(); frames[op.base_frame].Torque = V3(); }
      if (op.end_frame >= 0) { frames[op.end_frame].Force = V3(); frames[op.end_frame].Torque = V3(); }
      if (op.base_frame >= 0) { frames2[op.base_frame].Force = V2(); frames2[op.base_frame].Torque = 0.0; }
      if (op.end_frame >= 0) { frames2[op.end_frame].Force = V2(); frames2[op.end_frame].Torque = 0.0; }
      if (op.coord >= 0 && (op.kind == RKH_KTE_REVOLUTE_JOINT_3D || op.kind == RKH_KTE_REVOLUTE_JOINT_2D ||
                            op.kind == RKH_KTE_INERTIA_GEN))
        coords[op.coord].f = 0.0;
    }
  }
  // kte_map_chain::doForce: kte_map_chain.hpp:78-83 (reverse order)
  void doForce() {
    for (auto it = ops.rbegin(); it != ops.rend(); ++it) {
      const rkh_kte_op& op = *it;
      switch (op.kind) {
        case RKH_KTE_DRIVING_ACTUATOR_GEN: actuator_doForce(op); break;
        case RKH_KTE_INERTIA_GEN: inertia_gen_doForce(op); break;
        case RKH_KTE_REVOLUTE_JOINT_3D: revolute_doForce(op); break;
        case RKH_KTE_RIGID_LINK_3D: link_doForce(op); break;
        case RKH_KTE_INERTIA_3D: inertia_3D_doForce(op); break;
        case RKH_KTE_FLEXIBLE_BEAM_3D: beam_doForce(op); break;
        case RKH_KTE_REVOLUTE_JOINT_2D: {  // revolute_joint_2D::doForce, revolute_joint.cpp:58-67: the end frame's torque
          Frame2D& B = frames2[op.base_frame];  // goes to the joint coordinate only, nothing reaches the base's torque
          const Frame2D& E = frames2[op.end_frame];
          B.Force += rot_from_angle(coords[op.coord].q) * E.Force;
          coords[op.coord].f += E.Torque;
          break;
        }
        case RKH_KTE_RIGID_LINK_2D: {  // rigid_link_2D::doForce, rigid_link.cpp:104-106
          const Pose2 off = to_pose2(op.offset);
          Frame2D& B = frames2[op.base_frame];
          const Frame2D& E = frames2[op.end_frame];
          const V2 tmp_force = off.Rotation * E.Force;
          B.Force += tmp_force;
          B.Torque += E.Torque + cross_vv(off.Position, tmp_force);
          break;
        }
        case RKH_KTE_INERTIA_2D: {  // inertia_2D::doForce, inertia.cpp:73-81 (getGlobalFrame of a parentless frame = itself)
          Frame2D& F = frames2[op.end_frame];
          F.Force -= op.mass * (F.Acceleration * F.Rotation);
          F.Torque -= op.inertia[0] * F.AngAcceleration;
          break;
        }
        default: break;
      }
    }
  }
  // jacobian_gen_2D::get_jac_relative_to (motion_jacobians.hpp:138-146), velocity part.  Both frames hang off the global
  // node, so getFrameRelativeTo is (~Parent) * aFrame (frame_2D.hpp:164-167, operator~ :350-359, operator* :287-300).
  JacGen2D get_jac_relative_to_2D(const JacGen2D& J, int aFrame) const {
    const Frame2D& Pf = frames2[J.Parent];
    const Frame2D& A = frames2[aFrame];
    const V2 inv_pos = (-Pf.Position) * Pf.Rotation;
    const Rot2 inv_rot(Pf.Rotation.q[0], -Pf.Rotation.q[1]);  // invert = transpose (rotations_2D.hpp:389-391)
    const V2 f2_pos = inv_pos + inv_rot * A.Position;
    const Rot2 f2_rot = inv_rot * A.Rotation;
    JacGen2D r;
    r.Parent = aFrame;
    r.qd_vel = (cross_sv(J.qd_avel, f2_pos) + J.qd_vel) * f2_rot;
    r.qd_avel = J.qd_avel;
    return r;
  }

  // flexible_beam_3D::doForce without an object frame: flexible_beam.cpp:155-193 (:176-186)
  void beam_doForce(const rkh_kte_op& op) {
    Frame& a1 = frames[op.base_frame];
    Frame world;  // mAnchor2 fixed in the world: a parentless frame at the given pose, no motion
    if (op.end_frame < 0) {
      world.Position = V3(op.offset.pos[0], op.offset.pos[1], op.offset.pos[2]);
      world.Q = Quat(op.offset.quat[0], op.offset.quat[1], op.offset.quat[2], op.offset.quat[3]);
    }
    Frame& a2 = (op.end_frame >= 0) ? frames[op.end_frame] : world;
    const double mRestLength = op.axis[0], mStiffness = op.axis[1], mTorsionStiffness = op.axis[2];
    V3 diff = a1.Position - a2.Position;
    V3 diff_a1 = invert(a1.Q) * (-diff) - V3(mRestLength, 0.0, 0.0);
    V3 diff_a2 = invert(a2.Q) * diff + V3(mRestLength, 0.0, 0.0);
    AxisAngle angle_diff(invert(a1.Q) * a2.Q);
    a1.Force += mStiffness * diff_a1;
    a1.Torque += (mTorsionStiffness * angle_diff.mAngle) * angle_diff.mAxis;
    a2.Force += mStiffness * diff_a2;
    a2.Torque -= (mTorsionStiffness * angle_diff.mAngle) * angle_diff.mAxis;
  }

  // revolute_joint_3D::doMotion: revolute_joint.cpp:121-148
  void revolute_doMotion(const rkh_kte_op& op) {
    const Frame& mBase = frames[op.base_frame];
    Frame& mEnd = frames[op.end_frame];
    const GenCoord& mAngle = coords[op.coord];
    const V3 mAxis(op.axis[0], op.axis[1], op.axis[2]);
    mEnd.Position = mBase.Position;
    mEnd.Velocity = mBase.Velocity;
    mEnd.Acceleration = mBase.Acceleration;
    Quat tmp_quat(AxisAngle(mAngle.q, mAxis).getQuaternion());
    RotMat R2(tmp_quat.getRotMat());
    mEnd.Q = mBase.Q * tmp_quat;
    mEnd.AngVelocity = (mBase.AngVelocity * R2) + mAngle.q_dot * mAxis;
    mEnd.AngAcceleration = (mBase.AngAcceleration * R2) +
                           cross(mBase.AngVelocity * R2, mAngle.q_dot * mAxis) + mAngle.q_ddot * mAxis;
    JacGen3D& J = jac[op.coord];
    J.Parent = op.end_frame;
    J.qd_vel = V3();
    J.qd_avel = mAxis;
    J.qd_acc = V3();
    J.qd_aacc = V3();
  }
  // revolute_joint_3D::doForce: revolute_joint.cpp:170-181
  void revolute_doForce(const rkh_kte_op& op) {
    Frame& mBase = frames[op.base_frame];
    const Frame& mEnd = frames[op.end_frame];
    GenCoord& mAngle = coords[op.coord];
    const V3 mAxis(op.axis[0], op.axis[1], op.axis[2]);
    RotMat R(AxisAngle(mAngle.q, mAxis).getRotMat());
    mBase.Force += R * mEnd.Force;
    mAngle.f += dot(mEnd.Torque, mAxis);
    mBase.Torque += R * (mEnd.Torque - dot(mEnd.Torque, mAxis) * mAxis);
  }
  // revolute_joint_3D::applyReactionForce: revolute_joint.cpp:210-213
  void revolute_applyReactionForce(const rkh_kte_op& op, double aForce) {
    const V3 mAxis(op.axis[0], op.axis[1], op.axis[2]);
    frames[op.base_frame].Torque -= aForce * mAxis;
  }
  // rigid_link_3D::doMotion: rigid_link.cpp:152-156 -> frame_3D::operator*(frame, pose) :358-362
  // -> addBefore(pose) :240-255; frame_3D::operator= (:296-308) leaves mEnd's Force/Torque alone.
  void link_doMotion(const rkh_kte_op& op) {
    Frame tmp = frames[op.base_frame];
    tmp.addBefore(to_pose(op.offset));
    Frame& mEnd = frames[op.end_frame];
    tmp.Force = mEnd.Force;
    tmp.Torque = mEnd.Torque;
    mEnd = tmp;
  }
  // rigid_link_3D::doForce: rigid_link.cpp:170-178
  void link_doForce(const rkh_kte_op& op) {
    Pose mPoseOffset = to_pose(op.offset);
    Frame& mBase = frames[op.base_frame];
    const Frame& mEnd = frames[op.end_frame];
    RotMat R(mPoseOffset.Q.getRotMat());
    V3 tmp_force = R * mEnd.Force;
    mBase.Force += tmp_force;
    mBase.Torque += R * mEnd.Torque + cross(mPoseOffset.Position, tmp_force);
  }
  // symmetric 3x3 * vect: mat_alg_symmetric.hpp:646-659 (q = a11,a12,a22,a13,a23,a33)
  static V3 sym_mul(const double* t, const V3& V) {
    // t is (a11,a12,a13,a22,a23,a33) as in rkh_kte_op::inertia
    const double a11 = t[0], a12 = t[1], a13 = t[2], a22 = t[3], a23 = t[4], a33 = t[5];
    V3 result;
    result[0] += a11 * V[0];
    result[1] += a12 * V[0];
    result[0] += a12 * V[1];
    result[1] += a22 * V[1];
    result[2] += a13 * V[0];
    result[0] += a13 * V[2];
    result[2] += a23 * V[1];
    result[1] += a23 * V[2];
    result[2] += a33 * V[2];
    return result;
  }
  // inertia_3D::doForce: inertia.cpp:111-122 ; getGlobalFrame() of a parentless frame is the
  // frame itself (frame_3D.hpp:149-175).
  void inertia_3D_doForce(const rkh_kte_op& op) {
    Frame& fr = frames[op.end_frame];
    const Frame global_frame = fr;
    fr.Force -= op.mass * (invert(global_frame.Q) * global_frame.Acceleration);
    fr.Torque -= sym_mul(op.inertia, global_frame.AngAcceleration) +
                 cross(global_frame.AngVelocity, sym_mul(op.inertia, global_frame.AngVelocity));
  }
  // inertia_gen::doForce: inertia.cpp:47-54
  void inertia_gen_doForce(const rkh_kte_op& op) {
    coords[op.coord].f -= coords[op.coord].q_ddot * op.mass;
  }
  // driving_actuator_gen::doForce: driving_actuator.cpp:31-39
  void actuator_doForce(const rkh_kte_op& op) {
    coords[op.coord].f += drive[op.coord];
    const rkh_kte_op& joint = ops[op.joint_op];
    if (joint.kind == RKH_KTE_REVOLUTE_JOINT_2D)  // revolute_joint_2D::applyReactionForce: revolute_joint.cpp:112-115
      frames2[joint.base_frame].Torque -= drive[op.coord];
    else
      revolute_applyReactionForce(joint, drive[op.coord]);
  }

  // jacobian_gen_3D::get_jac_relative_to: motion_jacobians.hpp:238-251.  aFrame and the
  // jacobian's Parent are both parentless, so aFrame->getFrameRelativeTo(Parent) takes the
  // "this->Parent.expired()" branch: (~(p->getGlobalFrame())) * (*this)  (frame_3D.hpp:184-189).
  JacGen3D get_jac_relative_to(const JacGen3D& J, int aFrame) const {
    Frame f2 = frames[J.Parent].inverse();
    f2.addBefore(frames[aFrame]);
    RotMat R(f2.Q.getRotMat());
    V3 w_tmp = J.qd_avel * R;
    V3 v_tmp = (cross(J.qd_avel, f2.Position) + J.qd_vel) * R;
    JacGen3D r;
    r.Parent = aFrame;
    r.qd_vel = v_tmp;
    r.qd_avel = w_tmp;
    r.qd_acc = (cross(J.qd_avel, f2.Velocity) + cross(J.qd_aacc, f2.Position) + J.qd_acc) * R -
               cross(f2.AngVelocity, v_tmp);
    r.qd_aacc = J.qd_aacc * R - cross(f2.AngVelocity, w_tmp);
    return r;
  }

  // mass_matrix_calc::getMassMatrix: mass_matrix_calculator.cpp:80-87 with get_TMT_TdMT :100-295.
  // Returns M as a dense row-major n x n matrix already passed through the symmetric-matrix
  // conversion (0.5*(M(j,i)+M(i,j)), mat_alg_symmetric.hpp:172-200) that the assignment
  // "mat<symmetric> M = transpose(Tcm) * (Mcm * Tcm)" performs.
  void getMassMatrix(std::vector<double>& M) const {
    const int n = n_coords;
    const int m = int(gen_inertias.size()) + 3 * int(inertias_2D.size()) + 6 * int(inertias_3D.size());
    std::vector<double> Tcm(static_cast<std::size_t>(m) * n, 0.0);
    std::vector<double> Mcm(static_cast<std::size_t>(m) * m, 0.0);
    for (int i = 0; i < n; ++i) {
      int RowInd = 0;
      for (int j : gen_inertias) {  // :117-130, jacobian_gen_gen(1.0, 0.0)::write_to_matrices
        if (ops[j].upstream & (1u << i)) Tcm[RowInd * n + i] = 1.0;
        RowInd++;
      }
      for (int j : inertias_2D) {  // :130-145, three rows per inertia_2D: (v_x, v_y, omega)
        if (ops[j].upstream & (1u << i)) {
          JacGen2D r = get_jac_relative_to_2D(jac2[i], ops[j].end_frame);
          Tcm[(RowInd + 0) * n + i] = r.qd_vel[0];
          Tcm[(RowInd + 1) * n + i] = r.qd_vel[1];
          Tcm[(RowInd + 2) * n + i] = r.qd_avel;
        }
        RowInd += 3;
      }
      for (int j : inertias_3D) {  // :147-161
        if (ops[j].upstream & (1u << i)) {
          JacGen3D r = get_jac_relative_to(jac[i], ops[j].end_frame);
          for (int k = 0; k < 3; ++k) {
            Tcm[(RowInd + k) * n + i] = r.qd_vel[k];
            Tcm[(RowInd + 3 + k) * n + i] = r.qd_avel[k];
          }
        }
        RowInd += 6;
      }
    }
    int RowInd = 0;  // :262-285
    for (int j : gen_inertias) { Mcm[RowInd * m + RowInd] = ops[j].mass; RowInd++; }
    for (int j : inertias_2D) {  // :275-279
      Mcm[RowInd * m + RowInd] = ops[j].mass; RowInd++;
      Mcm[RowInd * m + RowInd] = ops[j].mass; RowInd++;
      Mcm[RowInd * m + RowInd] = ops[j].inertia[0]; RowInd++;
    }
    for (int j : inertias_3D) {
      for (int k = 0; k < 3; ++k) { Mcm[RowInd * m + RowInd] = ops[j].mass; RowInd++; }
      const double* t = ops[j].inertia;
      const double I[3][3] = {{t[0], t[1], t[2]}, {t[1], t[3], t[4]}, {t[2], t[4], t[5]}};
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) Mcm[(RowInd + a) * m + RowInd + b] = I[a][b];
      RowInd += 3;
    }
    // P = Mcm * Tcm : symmetric * rectangular, mat_alg_symmetric.hpp:551-566
    std::vector<double> P(static_cast<std::size_t>(m) * n, 0.0);
    for (int i = 0; i < m; ++i) {
      for (int l = 0; l < n; ++l) {
        for (int j = 0; j < i; ++j) {
          P[j * n + l] += Mcm[i * m + j] * Tcm[i * n + l];
          P[i * n + l] += Mcm[i * m + j] * Tcm[j * n + l];
        }
        P[i * n + l] += Mcm[i * m + i] * Tcm[i * n + l];
      }
    }
    // Mfull = transpose(Tcm) * P : dense_mat_multiply_impl, mat_operators.hpp:104-114
    std::vector<double> Mfull(static_cast<std::size_t>(n) * n, 0.0);
    for (int i = 0; i < n; ++i)
      for (int jj = 0; jj < n; ++jj) {
        double s = 0.0;
        for (int j = 0; j < m; ++j) s += Tcm[j * n + i] * P[j * n + jj];
        Mfull[i * n + jj] = s;
      }
    M.assign(static_cast<std::size_t>(n) * n, 0.0);
    for (int i = 0; i < n; ++i) {
      for (int j = 0; j < i; ++j) {
        double v = 0.5 * (Mfull[j * n + i] + Mfull[i * n + j]);
        M[i * n + j] = v;
        M[j * n + i] = v;
      }
      M[i * n + i] = Mfull[i * n + i];
    }
  }

  // kte_nl_system::apply_states_and_inputs: kte_nl_system.hpp:180-224 (dofs_gen only)
  void apply_states_and_inputs(const double* p, const double* u) {
    int i = 0;
    for (int j = 0; j < n_coords; ++j) {
      coords[j].q = p[i++];
      coords[j].q_dot = p[i++];
      coords[j].q_ddot = 0.0;
    }
    for (int j = 0; j < n_coords; ++j) drive[j] = u[j];
  }

  // kte_nl_system::get_state_derivative: kte_nl_system.hpp:239-290.  Also exports M and the
  // bias force f for the kernel-level golden vectors.  Throws singularity_error like the reference.
  void get_state_derivative(const double* p, const double* u, double* pd, double* M_out = nullptr,
                            double* f_out = nullptr) {
    apply_states_and_inputs(p, u);
    doMotion();
    clearForce();
    doForce();
    const int n = n_coords;
    std::vector<double> M;
    std::vector<double> f(n);
    for (int i = 0; i < n; ++i) f[i] = coords[i].f;
    getMassMatrix(M);
    if (M_out) for (int i = 0; i < n * n; ++i) M_out[i] = M[i];
    if (f_out) for (int i = 0; i < n; ++i) f_out[i] = f[i];
    linsolve_Cholesky(M.data(), f.data(), n);
    for (int i = 0; i < n; ++i) {
      pd[2 * i] = coords[i].q_dot;
      pd[2 * i + 1] = f[i];
    }
  }

  // manip_direct_kin_map::apply_to_model (ctrl/topologies/direct_kinematics_topomap.hpp:77-80):
  // write joint positions/velocities, then doMotion only.
  void apply_kinematics(const double* p) {
    for (int j = 0; j < n_coords; ++j) {
      coords[j].q = p[2 * j];
      coords[j].q_dot = p[2 * j + 1];
      coords[j].q_ddot = 0.0;
    }
    doMotion();
  }
};

}  // namespace oracle
#endif
