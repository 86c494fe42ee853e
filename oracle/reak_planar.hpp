// ORACLE -- TEST INFRASTRUCTURE ONLY (see reak_math.hpp header).
//
// CPU restatement of the planar (2D) pieces of the hot path, position level: rot_mat_2D / pose_2D arithmetic,
// revolute_joint_2D / rigid_link_2D kinematics, the 2D shapes' bounding radii, the closed-form 2D proximity pairs
// and proxy_query_pair_2D.  Each function cites the reference lines it follows.
//
// rkh_pose encoding of a pose_2D: pos[0..1] = Position, quat[0..1] = rot_mat_2D::q = (cos, sin); the rest is ignored.
// Parity pin status: the reference has no tests for these routines; pinned by closed forms in
// tests/test_oracle_kat.py (separated / touching / penetrating configurations worked by hand).
#ifndef REAK_ORACLE_PLANAR_HPP
#define REAK_ORACLE_PLANAR_HPP

#include <cmath>
#include <limits>
#include <vector>

#include "../include/rkh_types.h"

namespace oracle {

struct V2 {
  double v[2];
  V2() : v{0.0, 0.0} {}
  V2(double x, double y) : v{x, y} {}
  double& operator[](int i) { return v[i]; }
  double operator[](int i) const { return v[i]; }
};
inline V2 operator+(const V2& a, const V2& b) { return V2(a[0] + b[0], a[1] + b[1]); }
inline V2 operator-(const V2& a, const V2& b) { return V2(a[0] - b[0], a[1] - b[1]); }
inline V2 operator-(const V2& a) { return V2(-a[0], -a[1]); }
inline V2 operator*(double s, const V2& a) { return V2(a[0] * s, a[1] * s); }
inline V2& operator+=(V2& a, const V2& b) { a[0] += b[0]; a[1] += b[1]; return a; }
inline V2& operator-=(V2& a, const V2& b) { a[0] -= b[0]; a[1] -= b[1]; return a; }
inline double dot(const V2& a, const V2& b) { return a[0] * b[0] + a[1] * b[1]; }  // vect_alg.hpp operator*(vect,vect)
// norm_2 of a vect<double,2>: core/lin_alg/vect_alg.hpp (sum of squares left to right, sqrt)
inline double norm_2(const V2& a) {
  double s = 0.0;
  s += a[0] * a[0];
  s += a[1] * a[1];
  return std::sqrt(s);
}
// 2D cross product "scalar % vector" (vect_alg.hpp): S % V = (-V[1]*S, V[0]*S)
inline V2 cross_sv(double S, const V2& V) { return V2(-V[1] * S, V[0] * S); }

// rot_mat_2D<double>: core/kinetostatics/rotations_2D.hpp:59-
struct Rot2 {
  double q[2];
  Rot2() : q{1.0, 0.0} {}
  Rot2(double c, double s) : q{c, s} {}  // :89
};
inline Rot2 rot_from_angle(double Angle) { return Rot2(std::cos(Angle), std::sin(Angle)); }  // :108-112
inline Rot2 operator*(const Rot2& R1, const Rot2& R2) {                                      // :264-267
  return Rot2(R1.q[0] * R2.q[0] - R1.q[1] * R2.q[1], R1.q[1] * R2.q[0] + R1.q[0] * R2.q[1]);
}
inline V2 operator*(const Rot2& R, const V2& V) {  // :292-294
  return V2(V[0] * R.q[0] - V[1] * R.q[1], V[0] * R.q[1] + V[1] * R.q[0]);
}
inline V2 operator*(const V2& V, const Rot2& R) {  // :300-302 (row vector times R = R^T V)
  return V2(V[0] * R.q[0] + V[1] * R.q[1], V[1] * R.q[0] - V[0] * R.q[1]);
}

// pose_2D<double>: core/kinetostatics/pose_2D.hpp (global poses are resolved before use, so Parent is implicit)
struct Pose2 {
  V2 Position;
  Rot2 Rotation;
  V2 transformToParent(const V2& V) const { return Position + Rotation * V; }      // :177-179
  V2 transformFromParent(const V2& V) const { return (V - Position) * Rotation; }   // :191-193
  V2 rotateToParent(const V2& V) const { return Rotation * V; }                     // :149-151
  V2 rotateFromParent(const V2& V) const { return V * Rotation; }                   // :163-165
};
// 2D cross products (vect_alg.hpp:1142-1144, :1193-1198)
inline double cross_vv(const V2& a, const V2& b) { return a[0] * b[1] - a[1] * b[0]; }
// frame_2D<double>: core/kinetostatics/frame_2D.hpp (the kinematic and force fields on top of the pose)
struct Frame2D : Pose2 {
  V2 Velocity, Acceleration, Force;
  double AngVelocity = 0.0, AngAcceleration = 0.0, Torque = 0.0;
};
// jacobian_gen_2D<double>: core/kinetostatics/motion_jacobians.hpp:114-137 (velocity part); Parent is a frame index
struct JacGen2D {
  int Parent = -1;
  V2 qd_vel;
  double qd_avel = 0.0;
};

inline Pose2 to_pose2(const rkh_pose& p) {
  Pose2 r;
  r.Position = V2(p.pos[0], p.pos[1]);
  r.Rotation = Rot2(p.quat[0], p.quat[1]);
  return r;
}
// pose_2D::getGlobalPose: pose_2D.hpp:98-106
inline Pose2 global_pose2(const Pose2* parent, const Pose2& local) {
  if (!parent) return local;
  Pose2 result = *parent;
  result.Position += result.Rotation * local.Position;
  result.Rotation = result.Rotation * local.Rotation;  // operator*= :235-240, same arithmetic
  return result;
}

// proximity_record_2D: geometry/proximity/proximity_record_2D.hpp
struct ProxRecord2 {
  V2 mPoint1, mPoint2;
  double mDistance = std::numeric_limits<double>::infinity();
};

struct ShapeG2 {
  int kind;
  Pose2 g;
  double dims[2];
  // circle.cpp:31-33, rectangle.cpp:31-33, capped_rectangle.cpp:31-33 (norm_2(mDimensions) * 0.5: the caps of a
  // capped rectangle reach beyond this radius; the cull of findMinimumDistance is reproduced as written)
  double getBoundingRadius() const {
    if (kind == RKH_SHAPE_CIRCLE) return dims[0];
    return norm_2(V2(dims[0], dims[1])) * 0.5;
  }
};

// prox_circle_circle::computeProximity: prox_circle_circle.cpp:40-57
inline ProxRecord2 prox_circle_circle(const ShapeG2& s1, const ShapeG2& s2) {
  ProxRecord2 r;
  V2 c1 = s1.g.transformToParent(V2(0.0, 0.0));
  V2 c2 = s2.g.transformToParent(V2(0.0, 0.0));
  V2 diff_cc = c2 - c1;
  double dist_cc = norm_2(diff_cc);
  r.mDistance = dist_cc - s1.dims[0] - s2.dims[0];
  r.mPoint1 = c1 + (s1.dims[0] / dist_cc) * diff_cc;
  r.mPoint2 = c2 - (s2.dims[0] / dist_cc) * diff_cc;
  return r;
}

// prox_circle_crect::computeProximity: prox_circle_crect.cpp:40-84
inline ProxRecord2 prox_circle_crect(const ShapeG2& ci, const ShapeG2& cr) {
  ProxRecord2 r;
  V2 ci_c = ci.g.transformToParent(V2(0.0, 0.0));
  V2 ci_c_rel = cr.g.transformFromParent(ci_c);
  const double R = ci.dims[0];
  bool in_x_range = ((ci_c_rel[0] > -0.5 * cr.dims[0]) && (ci_c_rel[0] < 0.5 * cr.dims[0]));
  if (in_x_range) {
    if (ci_c_rel[1] > 0.0) {
      r.mPoint1 = cr.g.transformToParent(V2(ci_c_rel[0], ci_c_rel[1] - R));
      r.mPoint2 = cr.g.transformToParent(V2(ci_c_rel[0], 0.5 * cr.dims[1]));
      r.mDistance = ci_c_rel[1] - R - 0.5 * cr.dims[1];
    } else {
      r.mPoint1 = cr.g.transformToParent(V2(ci_c_rel[0], ci_c_rel[1] + R));
      r.mPoint2 = cr.g.transformToParent(V2(ci_c_rel[0], -0.5 * cr.dims[1]));
      r.mDistance = -0.5 * cr.dims[1] - ci_c_rel[1] - R;
    }
    return r;
  }
  V2 re_endc(0.0, 0.0);
  if (ci_c_rel[0] > 0.0) re_endc[0] += 0.5 * cr.dims[0];
  else re_endc[0] -= 0.5 * cr.dims[0];
  V2 diff_v_rel = ci_c_rel - re_endc;
  double diff_d_rel = norm_2(diff_v_rel);
  r.mPoint1 = cr.g.transformToParent(ci_c_rel - (R / diff_d_rel) * diff_v_rel);
  r.mPoint2 = cr.g.transformToParent(re_endc + (0.5 * cr.dims[1] / diff_d_rel) * diff_v_rel);
  r.mDistance = diff_d_rel - 0.5 * cr.dims[1] - R;
  return r;
}

// prox_circle_rectangle::computeProximity: prox_circle_rectangle.cpp:40-86
inline ProxRecord2 prox_circle_rectangle(const ShapeG2& ci, const ShapeG2& re) {
  ProxRecord2 r;
  V2 ci_c = ci.g.transformToParent(V2(0.0, 0.0));
  V2 ci_c_rel = re.g.transformFromParent(ci_c);
  bool in_x_range = ((ci_c_rel[0] > -0.5 * re.dims[0]) && (ci_c_rel[0] < 0.5 * re.dims[0]));
  bool in_y_range = ((ci_c_rel[1] > -0.5 * re.dims[1]) && (ci_c_rel[1] < 0.5 * re.dims[1]));
  if (in_x_range && in_y_range) {
    V2 bound_dists(0.5 * re.dims[0] - std::fabs(ci_c_rel[0]), 0.5 * re.dims[1] - std::fabs(ci_c_rel[1]));
    if (bound_dists[0] <= bound_dists[1]) in_x_range = false;
    else in_y_range = false;
  }
  V2 corner_pt = 0.5 * V2(re.dims[0], re.dims[1]);
  if (in_x_range) corner_pt[0] = ci_c_rel[0];
  else if (ci_c_rel[0] < 0.0) corner_pt[0] = -corner_pt[0];
  if (in_y_range) corner_pt[1] = ci_c_rel[1];
  else if (ci_c_rel[1] < 0.0) corner_pt[1] = -corner_pt[1];
  r.mPoint2 = re.g.transformToParent(corner_pt);
  V2 diff_v = r.mPoint2 - ci_c;
  double diff_d = norm_2(diff_v);
  r.mPoint1 = ci_c + (ci.dims[0] / diff_d) * diff_v;
  r.mDistance = diff_d - ci.dims[0];  // never negative: a circle centre inside the rectangle is not a collision here
  return r;
}

// prox_crect_crect::computeProximity: prox_crect_crect.cpp:40-131
inline ProxRecord2 prox_crect_crect(const ShapeG2& c1, const ShapeG2& c2) {
  ProxRecord2 r;
  V2 cr2_c = c2.g.transformToParent(V2(0.0, 0.0));
  V2 cr2_t = c2.g.rotateToParent(V2(1.0, 0.0));
  V2 cr2_c_rel = c1.g.transformFromParent(cr2_c);
  V2 cr2_t_rel = c1.g.rotateFromParent(cr2_t);
  const double L1 = c1.dims[0], W1 = c1.dims[1], L2 = c2.dims[0], W2 = c2.dims[1];
  if (std::fabs(cr2_t_rel[1]) < 1e-5) {
    // (the '||' of :58-59 is always true, as in the 3D capped-cylinder routine)
    if ((cr2_c_rel[0] + 0.5 * L2 > -0.5 * L1) || (cr2_c_rel[0] - 0.5 * L2 < 0.5 * L1)) {
      double max_x_rel = ((cr2_c_rel[0] + 0.5 * L2 < 0.5 * L1) ? (cr2_c_rel[0] + 0.5 * L2) : (0.5 * L1));
      double min_x_rel = ((cr2_c_rel[0] - 0.5 * L2 > -0.5 * L1) ? (cr2_c_rel[0] - 0.5 * L2) : (-0.5 * L1));
      double avg_x_rel = (max_x_rel + min_x_rel) * 0.5;
      V2 cr2_r_rel(0.0, 1.0);
      if (cr2_c_rel[1] < 0.0) cr2_r_rel[1] = -1.0;
      r.mPoint1 = c1.g.transformToParent(V2(avg_x_rel, 0.5 * W1 * cr2_r_rel[1]));
      r.mPoint2 = c1.g.transformToParent(V2(avg_x_rel, cr2_c_rel[1] - 0.5 * W2 * cr2_r_rel[1]));
      r.mDistance = std::fabs(cr2_c_rel[1]) - 0.5 * W1 - 0.5 * W2;
      return r;
    }
    V2 cr1_cic_rel(0.0, 0.0);
    V2 cr2_cic_rel = cr2_c_rel;
    if (cr2_c_rel[0] < 0.0) {
      cr1_cic_rel[0] -= 0.5 * L1;
      cr2_cic_rel[0] += 0.5 * L2;
    } else {
      cr1_cic_rel[0] += 0.5 * L1;
      cr2_cic_rel[0] -= 0.5 * L2;
    }
    V2 diff_v_rel = cr2_cic_rel - cr1_cic_rel;
    double dist_v_rel = norm_2(diff_v_rel);
    r.mPoint1 = c1.g.transformToParent(cr1_cic_rel + (0.5 * W1 / dist_v_rel) * diff_v_rel);
    r.mPoint2 = c1.g.transformToParent(cr2_cic_rel - (0.5 * W2 / dist_v_rel) * diff_v_rel);
    r.mDistance = dist_v_rel - 0.5 * W1 - 0.5 * W2;
    return r;
  }
  double d = dot(cr2_t_rel, cr2_c_rel);
  double denom = 1.0 - cr2_t_rel[0] * cr2_t_rel[0];
  double s_c = (cr2_t_rel[0] * cr2_c_rel[0] - d) / denom;
  double t_c = (cr2_c_rel[0] - cr2_t_rel[0] * d) / denom;
  if (s_c < -0.5 * L2) {
    s_c = -0.5 * L2;
    t_c = cr2_c_rel[0] - 0.5 * L2 * cr2_t_rel[0];
  } else if (s_c > 0.5 * L2) {
    s_c = 0.5 * L2;
    t_c = cr2_c_rel[0] + 0.5 * L2 * cr2_t_rel[0];
  }
  if (t_c < -0.5 * L1) {
    t_c = -0.5 * L1;
    s_c = -0.5 * L1 * cr2_t_rel[0] - d;
  } else if (t_c > 0.5 * L1) {
    t_c = 0.5 * L1;
    s_c = 0.5 * L1 * cr2_t_rel[0] - d;
  }
  if (s_c < -0.5 * L2) s_c = -0.5 * L2;
  else if (s_c > 0.5 * L2) s_c = 0.5 * L2;
  V2 cr1_ptc(t_c, 0.0);
  V2 cr2_ptc = cr2_c_rel + s_c * cr2_t_rel;
  V2 diff_v_rel = cr2_ptc - cr1_ptc;
  double dist_v_rel = norm_2(diff_v_rel);
  r.mPoint1 = c1.g.transformToParent(cr1_ptc + (0.5 * W1 / dist_v_rel) * diff_v_rel);
  r.mPoint2 = c1.g.transformToParent(cr2_ptc - (0.5 * W2 / dist_v_rel) * diff_v_rel);
  r.mDistance = dist_v_rel - 0.5 * W1 - 0.5 * W2;
  return r;
}

// prox_crect_rectangle::computeProximityOfLine: prox_crect_rectangle.cpp:40-181
inline void crect_rectangle_line(const ShapeG2& re, const V2& ln_c, const V2& ln_t, double half_length, ProxRecord2& result) {
  V2 ln_c_rel = re.g.transformFromParent(ln_c);
  V2 ln_t_rel = re.g.rotateFromParent(ln_t);
  const double DX = re.dims[0], DY = re.dims[1];
  if (std::fabs(ln_t_rel[0]) < 1e-5) {  // vertical line
    if ((ln_c_rel[1] + half_length > -0.5 * DY) || (ln_c_rel[1] - half_length < 0.5 * DY)) {  // always true (:51-52)
      double max_y_rel = ((ln_c_rel[1] + half_length < 0.5 * DY) ? (ln_c_rel[1] + half_length) : (0.5 * DY));
      double min_y_rel = ((ln_c_rel[1] - half_length > -0.5 * DY) ? (ln_c_rel[1] - half_length) : (-0.5 * DY));
      double avg_y_rel = (max_y_rel + min_y_rel) * 0.5;
      V2 ln_r_rel(1.0, 0.0);
      if (ln_c_rel[0] < 0.0) ln_r_rel[0] = -1.0;
      result.mPoint1 = re.g.transformToParent(V2(ln_c_rel[0], avg_y_rel));
      result.mPoint2 = re.g.transformToParent(V2(0.5 * DX * ln_r_rel[0], avg_y_rel));
      result.mDistance = std::fabs(ln_c_rel[0]) - 0.5 * DX;
      return;
    }
    V2 re_pt_rel(0.0, 0.0);
    V2 ln_pt_rel = ln_c_rel;
    if (ln_c_rel[0] < 0.0) re_pt_rel[0] -= 0.5 * DX;
    else re_pt_rel[0] += 0.5 * DX;
    if (ln_c_rel[1] < 0.0) {
      re_pt_rel[1] -= 0.5 * DY;
      ln_pt_rel[1] += half_length;
    } else {
      re_pt_rel[1] += 0.5 * DY;
      ln_pt_rel[1] -= half_length;
    }
    V2 diff_v_rel = ln_pt_rel - re_pt_rel;
    double dist_v_rel = norm_2(diff_v_rel);
    result.mPoint1 = re.g.transformToParent(ln_pt_rel);
    result.mPoint2 = re.g.transformToParent(re_pt_rel);
    result.mDistance = dist_v_rel;
    return;
  }
  if (std::fabs(ln_t_rel[1]) < 1e-5) {  // horizontal line
    if ((ln_c_rel[0] + half_length > -0.5 * DX) || (ln_c_rel[0] - half_length < 0.5 * DX)) {  // always true (:90-91)
      double max_x_rel = ((ln_c_rel[0] + half_length < 0.5 * DX) ? (ln_c_rel[0] + half_length) : (0.5 * DX));
      double min_x_rel = ((ln_c_rel[0] - half_length > -0.5 * DX) ? (ln_c_rel[0] - half_length) : (-0.5 * DX));
      double avg_x_rel = (max_x_rel + min_x_rel) * 0.5;
      V2 ln_r_rel(0.0, 1.0);
      if (ln_c_rel[1] < 0.0) ln_r_rel[1] = -1.0;
      result.mPoint1 = re.g.transformToParent(V2(avg_x_rel, ln_c_rel[1]));
      result.mPoint2 = re.g.transformToParent(V2(avg_x_rel, 0.5 * DY * ln_r_rel[1]));
      result.mDistance = std::fabs(ln_c_rel[1]) - 0.5 * DY;
      return;
    }
    V2 re_pt_rel(0.0, 0.0);
    V2 ln_pt_rel = ln_c_rel;
    if (ln_c_rel[1] < 0.0) re_pt_rel[1] -= 0.5 * DY;
    else re_pt_rel[1] += 0.5 * DY;
    if (ln_c_rel[0] < 0.0) {
      re_pt_rel[0] -= 0.5 * DX;
      ln_pt_rel[0] += half_length;
    } else {
      re_pt_rel[0] += 0.5 * DX;
      ln_pt_rel[0] -= half_length;
    }
    V2 diff_v_rel = ln_pt_rel - re_pt_rel;
    double dist_v_rel = norm_2(diff_v_rel);
    result.mPoint1 = re.g.transformToParent(ln_pt_rel);
    result.mPoint2 = re.g.transformToParent(re_pt_rel);
    result.mDistance = dist_v_rel;
    return;
  }
  // segment-point test (:126-180)
  V2 ln_n_rel = cross_sv(1.0, ln_t_rel);
  if (dot(ln_n_rel, ln_c_rel) < 0.0) ln_n_rel = -ln_n_rel;
  V2 corner_pt(-0.5 * DX, -0.5 * DY);
  if (ln_n_rel[0] > 0.0) corner_pt[0] = 0.5 * DX;
  if (ln_n_rel[1] > 0.0) corner_pt[1] = 0.5 * DY;
  V2 corner_pt_diff = (ln_c_rel - corner_pt);
  double dist_tmp = dot(corner_pt_diff, ln_n_rel);
  double t_tmp = -dot(corner_pt_diff, ln_t_rel);
  if (std::fabs(t_tmp) > half_length) {
    if (t_tmp < 0.0) t_tmp = -half_length;
    else t_tmp = half_length;
    V2 ln_pt_rel = ln_c_rel + t_tmp * ln_t_rel;
    double in_x_range = std::fabs(ln_pt_rel[0]) - 0.5 * DX;
    double in_y_range = std::fabs(ln_pt_rel[1]) - 0.5 * DY;
    if ((in_x_range < 0.0) && (in_y_range > in_x_range)) {
      corner_pt[0] = ln_pt_rel[0];
      dist_tmp = std::fabs(ln_pt_rel[1]) - 0.5 * DY;
    } else if ((in_y_range < 0.0) && (in_x_range > in_y_range)) {
      corner_pt[1] = ln_pt_rel[1];
      dist_tmp = std::fabs(ln_pt_rel[0]) - 0.5 * DX;
    } else {
      if (ln_pt_rel[0] < 0.0) corner_pt[0] = -0.5 * DX;
      else corner_pt[0] = 0.5 * DX;
      if (ln_pt_rel[1] < 0.0) corner_pt[1] = -0.5 * DY;
      else corner_pt[1] = 0.5 * DY;
      dist_tmp = norm_2(ln_pt_rel - corner_pt);
    }
    result.mPoint1 = re.g.transformToParent(ln_pt_rel);
    result.mPoint2 = re.g.transformToParent(corner_pt);
    result.mDistance = dist_tmp;
  } else {
    result.mPoint1 = re.g.transformToParent(corner_pt + dist_tmp * ln_n_rel);
    result.mPoint2 = re.g.transformToParent(corner_pt);
    result.mDistance = dist_tmp;
  }
}

// prox_crect_rectangle::computeProximity: prox_crect_rectangle.cpp:184-205
inline ProxRecord2 prox_crect_rectangle(const ShapeG2& cr, const ShapeG2& re) {
  ProxRecord2 r;
  V2 cr_c = cr.g.transformToParent(V2(0.0, 0.0));
  V2 cr_t = cr.g.rotateToParent(V2(1.0, 0.0));
  crect_rectangle_line(re, cr_c, cr_t, 0.5 * cr.dims[0], r);
  V2 diff_v = r.mPoint2 - r.mPoint1;
  double diff_d = norm_2(diff_v);
  if (r.mDistance < 0.0) r.mPoint1 -= (0.5 * cr.dims[1] / diff_d) * diff_v;
  else r.mPoint1 += (0.5 * cr.dims[1] / diff_d) * diff_v;
  r.mDistance -= 0.5 * cr.dims[1];
  return r;
}

// prox_rectangle_rectangle::computeProximityOfPoint: prox_rectangle_rectangle.cpp:40-77
inline void rectangle_point(const ShapeG2& re, const V2& aPoint, V2& aPointRec, double& aDistance) {
  V2 pt_rel = re.g.transformFromParent(aPoint);
  bool in_x_range = ((pt_rel[0] > -0.5 * re.dims[0]) && (pt_rel[0] < 0.5 * re.dims[0]));
  bool in_y_range = ((pt_rel[1] > -0.5 * re.dims[1]) && (pt_rel[1] < 0.5 * re.dims[1]));
  if (in_x_range && in_y_range) {
    V2 bound_dists(0.5 * re.dims[0] - std::fabs(pt_rel[0]), 0.5 * re.dims[1] - std::fabs(pt_rel[1]));
    if (bound_dists[0] <= bound_dists[1]) in_x_range = false;
    else in_y_range = false;
  }
  V2 corner_pt = 0.5 * V2(re.dims[0], re.dims[1]);
  if (in_x_range) corner_pt[0] = pt_rel[0];
  else if (pt_rel[0] < 0.0) corner_pt[0] = -corner_pt[0];
  if (in_y_range) corner_pt[1] = pt_rel[1];
  else if (pt_rel[1] < 0.0) corner_pt[1] = -corner_pt[1];
  aPointRec = re.g.transformToParent(corner_pt);
  aDistance = norm_2(aPointRec - aPoint);
}
// prox_rectangle_rectangle::computeProximity: prox_rectangle_rectangle.cpp:79-168 (unsigned distance: the four
// corners of each rectangle against the other, first strict minimum wins)
inline ProxRecord2 prox_rectangle_rectangle(const ShapeG2& r1, const ShapeG2& r2) {
  ProxRecord2 r;
  V2 temp_pt;
  double temp_dist;
  for (int pass = 0; pass < 2; ++pass) {
    const ShapeG2& own = pass == 0 ? r2 : r1;    // the rectangle whose corners are visited
    const ShapeG2& other = pass == 0 ? r1 : r2;  // the rectangle they are measured against
    V2 corner = 0.5 * V2(own.dims[0], own.dims[1]);
    for (int c = 0; c < 4; ++c) {
      if (c == 1 || c == 3) corner[1] = -corner[1];
      if (c == 2) corner[0] = -corner[0];
      V2 corner_gbl = own.g.transformToParent(corner);
      rectangle_point(other, corner_gbl, temp_pt, temp_dist);
      if (temp_dist < r.mDistance) {
        r.mDistance = temp_dist;
        if (pass == 0) { r.mPoint1 = temp_pt; r.mPoint2 = corner_gbl; }
        else { r.mPoint2 = temp_pt; r.mPoint1 = corner_gbl; }
      }
    }
  }
  return r;
}

// One entry of proxy_query_pair_2D::mProxFinders (createProxFinderList: proxy_query_model.cpp:75-161)
struct ProxFinder2 {
  int routine;  // 11 circle-circle, 12 circle-crect, 13 circle-rectangle, 14 crect-crect, 15 crect-rectangle, 16 rect-rect
  int s1, s2;   // (shape1, shape2) in the finder's argument order = getShape1()/getShape2()
};
inline void createProxFinderList2D(const std::vector<rkh_shape>& shapes, const std::vector<int>& model1,
                                   const std::vector<int>& model2, std::vector<ProxFinder2>& out) {
  out.clear();
  for (int i : model1)
    for (int j : model2) {
      const int ki = shapes[i].kind, kj = shapes[j].kind;
      if (ki == RKH_SHAPE_CIRCLE || kj == RKH_SHAPE_CIRCLE) {
        const int ci = (ki == RKH_SHAPE_CIRCLE) ? i : j, other = (ki == RKH_SHAPE_CIRCLE) ? j : i;
        const int ko = shapes[other].kind;
        if (ko == RKH_SHAPE_CIRCLE) out.push_back({11, ci, other});
        else if (ko == RKH_SHAPE_CRECT) out.push_back({12, ci, other});
        else if (ko == RKH_SHAPE_RECTANGLE) out.push_back({13, ci, other});
      } else if (ki == RKH_SHAPE_CRECT || kj == RKH_SHAPE_CRECT) {
        const int cr = (ki == RKH_SHAPE_CRECT) ? i : j, other = (ki == RKH_SHAPE_CRECT) ? j : i;
        const int ko = shapes[other].kind;
        if (ko == RKH_SHAPE_CRECT) out.push_back({14, cr, other});
        else if (ko == RKH_SHAPE_RECTANGLE) out.push_back({15, cr, other});
      } else if (ki == RKH_SHAPE_RECTANGLE || kj == RKH_SHAPE_RECTANGLE) {
        // (:139-155: when model1's shape is the rectangle it becomes re_geom; both are rectangles here)
        if (ki == RKH_SHAPE_RECTANGLE && kj == RKH_SHAPE_RECTANGLE) out.push_back({16, i, j});
      }
    }
}
inline ProxRecord2 computeProximity2D(const ProxFinder2& f, const std::vector<ShapeG2>& g) {
  switch (f.routine) {
    case 11: return prox_circle_circle(g[f.s1], g[f.s2]);
    case 12: return prox_circle_crect(g[f.s1], g[f.s2]);
    case 13: return prox_circle_rectangle(g[f.s1], g[f.s2]);
    case 14: return prox_crect_crect(g[f.s1], g[f.s2]);
    case 15: return prox_crect_rectangle(g[f.s1], g[f.s2]);
    case 16: return prox_rectangle_rectangle(g[f.s1], g[f.s2]);
  }
  return ProxRecord2();
}
// proxy_query_pair_2D::findMinimumDistance: proxy_query_model.cpp:163-189.  The cull against the running minimum is
// order dependent here (capped rectangles reach beyond their bounding radius), so it is restated as the sequence it is.
inline double findMinimumDistance2D(const std::vector<ProxFinder2>& finders, const std::vector<ShapeG2>& g,
                                    long* n_computed = nullptr) {
  if (finders.empty()) return std::numeric_limits<double>::infinity();
  double min_dist = computeProximity2D(finders[0], g).mDistance;
  long cnt = 1;
  for (std::size_t i = 1; i < finders.size(); ++i) {
    V2 p1 = g[finders[i].s1].g.transformToParent(V2(0.0, 0.0));
    V2 p2 = g[finders[i].s2].g.transformToParent(V2(0.0, 0.0));
    if (norm_2(p2 - p1) - g[finders[i].s1].getBoundingRadius() - g[finders[i].s2].getBoundingRadius() > min_dist)
      continue;
    double d = computeProximity2D(finders[i], g).mDistance;
    ++cnt;
    if (d < min_dist) min_dist = d;
  }
  if (n_computed) *n_computed += cnt;
  return min_dist;
}

}  // namespace oracle
#endif
