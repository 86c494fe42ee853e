// ORACLE -- TEST INFRASTRUCTURE ONLY.  Nothing in the product path (reak_amd/, include/) may
// include, link or call this.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
// leg use it, and only as the checker.
//
// CPU restatement (C++17, no Boost) of ReaK's leaf math on the planning hot path.  Every
// function cites the reference file:line (relative to /root/reference/src/ReaK/) whose fp64
// operation order it follows.  Build with -O2 -ffp-contract=off (no FMA contraction): the
// reference is built -O3 for plain x86-64 (src/CMakeLists.txt:58), i.e. without FMA.
//
// Parity pin status: quaternion / rot_mat / axis_angle identities are checked against the
// known-answer cases of core/kinetostatics/unit_test_rotations.cpp, Cholesky against
// core/lin_alg/unit_test_mat_num.cpp (tests/test_oracle_kat.py).
#ifndef REAK_ORACLE_MATH_HPP
#define REAK_ORACLE_MATH_HPP

#include <cmath>
#include <cstddef>
#include <limits>
#include <stdexcept>
#include <vector>

namespace oracle {

// ---------------------------------------------------------------- vect<double,3>
// core/lin_alg/vect_alg.hpp (fixed-size vector; operators are component-wise)
struct V3 {
  double q[3];
  V3() : q{0.0, 0.0, 0.0} {}
  V3(double x, double y, double z) : q{x, y, z} {}
  double& operator[](int i) { return q[i]; }
  const double& operator[](int i) const { return q[i]; }
};
inline V3 operator+(const V3& a, const V3& b) { return V3(a[0] + b[0], a[1] + b[1], a[2] + b[2]); }
inline V3 operator-(const V3& a, const V3& b) { return V3(a[0] - b[0], a[1] - b[1], a[2] - b[2]); }
inline V3 operator-(const V3& a) { return V3(-a[0], -a[1], -a[2]); }
inline V3 operator*(double s, const V3& a) { return V3(a[0] * s, a[1] * s, a[2] * s); }
inline V3 operator*(const V3& a, double s) { return V3(a[0] * s, a[1] * s, a[2] * s); }
inline V3& operator+=(V3& a, const V3& b) { a[0] += b[0]; a[1] += b[1]; a[2] += b[2]; return a; }
inline V3& operator-=(V3& a, const V3& b) { a[0] -= b[0]; a[1] -= b[1]; a[2] -= b[2]; return a; }
// dot product: vect_alg.hpp:2547-2555 (result(0); result += v1[i]*v2[i])
inline double dot(const V3& a, const V3& b) {
  double r = 0.0;
  for (int i = 0; i < 3; ++i) r += a[i] * b[i];
  return r;
}
// 3D cross product: vect_alg.hpp:1214-1221
inline V3 cross(const V3& a, const V3& b) {
  return V3(a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]);
}
// norm_2: vect_alg.hpp:2314-2333
inline double norm_2(const V3& v) {
  double s = 0.0;
  for (int i = 0; i < 3; ++i) s += v[i] * v[i];
  return std::sqrt(s);
}
// unit(): vect_alg.hpp:2378-2382 (result /= norm_2(result))
inline V3 unit(const V3& v) {
  double n = norm_2(v);
  return V3(v[0] / n, v[1] / n, v[2] / n);
}

// ---------------------------------------------------------------- rot_mat_3D<double>
// core/kinetostatics/rotations_3D.hpp:100-113 : column-major q[9], ctor takes row-major args.
struct RotMat {
  double q[9];
  RotMat() : q{1, 0, 0, 0, 1, 0, 0, 0, 1} {}
  RotMat(double a11, double a12, double a13, double a21, double a22, double a23, double a31,
         double a32, double a33)
      : q{a11, a21, a31, a12, a22, a32, a13, a23, a33} {}
};
// R * V : rotations_3D.hpp:372-376
inline V3 operator*(const RotMat& R, const V3& V) {
  return V3(R.q[0] * V[0] + R.q[3] * V[1] + R.q[6] * V[2],
            R.q[1] * V[0] + R.q[4] * V[1] + R.q[7] * V[2],
            R.q[2] * V[0] + R.q[5] * V[1] + R.q[8] * V[2]);
}
// V * R (= R^T V): rotations_3D.hpp:379-383
inline V3 operator*(const V3& V, const RotMat& R) {
  return V3(R.q[0] * V[0] + R.q[1] * V[1] + R.q[2] * V[2],
            R.q[3] * V[0] + R.q[4] * V[1] + R.q[5] * V[2],
            R.q[6] * V[0] + R.q[7] * V[1] + R.q[8] * V[2]);
}

// ---------------------------------------------------------------- quaternion<double>
struct Quat {
  double q[4];
  Quat() : q{1.0, 0.0, 0.0, 0.0} {}  // rotations_3D.hpp:908
  Quat(double w, double x, double y, double z) : q{w, x, y, z} {}  // private raw ctor :570
  // explicit quaternion(const Vector&): normalises (rotations_3D.hpp:916-920)
  static Quat from_vector(double w, double x, double y, double z) {
    double s = 0.0;
    s += w * w; s += x * x; s += y * y; s += z * z;
    double n = std::sqrt(s);
    return Quat(w / n, x / n, y / n, z / n);
  }
  // getRotMat: rotations_3D.hpp:986-999
  RotMat getRotMat() const {
    double t01(2.0 * q[0] * q[1]);
    double t02(2.0 * q[0] * q[2]);
    double t03(2.0 * q[0] * q[3]);
    double t11(2.0 * q[1] * q[1]);
    double t12(2.0 * q[1] * q[2]);
    double t13(2.0 * q[1] * q[3]);
    double t22(2.0 * q[2] * q[2]);
    double t23(2.0 * q[2] * q[3]);
    double t33(2.0 * q[3] * q[3]);
    return RotMat(1.0 - t22 - t33, t12 - t03, t02 + t13,
                  t12 + t03, 1.0 - t11 - t33, t23 - t01,
                  t13 - t02, t01 + t23, 1.0 - t11 - t22);
  }
};
// Q1 * Q2 : rotations_3D.hpp:1093-1098
inline Quat operator*(const Quat& Q1, const Quat& Q2) {
  return Quat(Q2.q[0] * Q1.q[0] - Q2.q[1] * Q1.q[1] - Q2.q[2] * Q1.q[2] - Q2.q[3] * Q1.q[3],
              Q2.q[0] * Q1.q[1] + Q2.q[3] * Q1.q[2] - Q2.q[2] * Q1.q[3] + Q2.q[1] * Q1.q[0],
              Q2.q[0] * Q1.q[2] - Q2.q[3] * Q1.q[1] + Q2.q[1] * Q1.q[3] + Q2.q[2] * Q1.q[0],
              Q2.q[0] * Q1.q[3] + Q2.q[2] * Q1.q[1] - Q2.q[1] * Q1.q[2] + Q2.q[3] * Q1.q[0]);
}
// Q * V : rotations_3D.hpp:1137-1151
inline V3 operator*(const Quat& Q, const V3& V) {
  double t[9];
  t[0] = Q.q[0] * Q.q[1];
  t[1] = Q.q[0] * Q.q[2];
  t[2] = Q.q[0] * Q.q[3];
  t[3] = -Q.q[1] * Q.q[1];
  t[4] = Q.q[1] * Q.q[2];
  t[5] = Q.q[1] * Q.q[3];
  t[6] = -Q.q[2] * Q.q[2];
  t[7] = Q.q[2] * Q.q[3];
  t[8] = -Q.q[3] * Q.q[3];
  return V3(2.0 * ((t[6] + t[8]) * V[0] + (t[4] - t[2]) * V[1] + (t[1] + t[5]) * V[2]) + V[0],
            2.0 * ((t[2] + t[4]) * V[0] + (t[3] + t[8]) * V[1] + (t[7] - t[0]) * V[2]) + V[1],
            2.0 * ((t[5] - t[1]) * V[0] + (t[0] + t[7]) * V[1] + (t[3] + t[6]) * V[2]) + V[2]);
}
// invert(Q): rotations_3D.hpp:1280-1282
inline Quat invert(const Quat& Q) { return Quat(Q.q[0], -Q.q[1], -Q.q[2], -Q.q[3]); }

// ---------------------------------------------------------------- axis_angle<double>
struct AxisAngle {
  double mAngle;
  V3 mAxis;
  // ctor(angle, axis): rotations_3D.hpp:1961-1974 (normalises, threshold 1e-7)
  AxisAngle(double aAngle, const V3& aAxis) : mAngle(aAngle) {
    double tmp = norm_2(aAxis);
    if (tmp > 0.0000001) {
      mAxis = V3(aAxis[0] / tmp, aAxis[1] / tmp, aAxis[2] / tmp);
    } else {
      mAxis = V3(1.0, 0.0, 0.0);
    }
  }
  // ctor(quaternion): rotations_3D.hpp:1986-2006
  explicit AxisAngle(const Quat& Q) : mAngle(0.0) {
    double v[4] = {Q.q[0], Q.q[1], Q.q[2], Q.q[3]};
    double nrm = 0.0;  // unit(v): v /= norm_2(v) (vect_alg.hpp)
    for (int i = 0; i < 4; ++i) nrm += v[i] * v[i];
    nrm = std::sqrt(nrm);
    for (int i = 0; i < 4; ++i) v[i] /= nrm;
    double tmp = std::sqrt(v[1] * v[1] + v[2] * v[2] + v[3] * v[3]);
    if (tmp > 0.0000001) {
      mAxis = V3(v[1] / tmp, v[2] / tmp, v[3] / tmp);
      if (v[0] < 0.0) {
        mAngle = 2.0 * std::acos(-v[0]);
        mAxis = V3(-mAxis[0], -mAxis[1], -mAxis[2]);
      } else {
        mAngle = 2.0 * std::acos(v[0]);
      }
    } else {
      mAxis = V3(1.0, 0.0, 0.0);
      mAngle = 0.0;
    }
  }
  // getQuaternion: rotations_3D.hpp:2107-2115
  Quat getQuaternion() const {
    double t = norm_2(mAxis);
    if (t == 0.0) return Quat(1.0, 0.0, 0.0, 0.0);
    t = std::sin(0.5 * mAngle);
    return Quat(std::cos(0.5 * mAngle), mAxis[0] * t, mAxis[1] * t, mAxis[2] * t);
  }
  // getRotMat: rotations_3D.hpp:2160-2180
  RotMat getRotMat() const {
    double ca(std::cos(mAngle));
    double one_minus_ca(1.0 - ca);
    double t11(ca + one_minus_ca * mAxis[0] * mAxis[0]);
    double t22(ca + one_minus_ca * mAxis[1] * mAxis[1]);
    double t33(ca + one_minus_ca * mAxis[2] * mAxis[2]);
    double t12(one_minus_ca * mAxis[0] * mAxis[1]);
    double t13(one_minus_ca * mAxis[0] * mAxis[2]);
    double t23(one_minus_ca * mAxis[1] * mAxis[2]);
    double sin_a(std::sin(mAngle));
    double t01(sin_a * mAxis[0]);
    double t02(sin_a * mAxis[1]);
    double t03(sin_a * mAxis[2]);
    return RotMat(t11, t12 - t03, t13 + t02, t12 + t03, t22, t23 - t01, t13 - t02, t23 + t01, t33);
  }
};

// ---------------------------------------------------------------- pose_3D<double>
// core/kinetostatics/pose_3D.hpp.  In every scene this oracle builds, a pose's Parent is either
// null (global) or a chain frame whose own Parent is null, so the parent walk has depth <= 1.
struct Pose {
  V3 Position;
  Quat Q;
  // transformToParent: pose_3D.hpp:175-177
  V3 transformToParent(const V3& V) const { return Position + Q * V; }
  // transformFromParent: pose_3D.hpp:189-191
  V3 transformFromParent(const V3& V) const { return invert(Q) * (V - Position); }
};
// getGlobalPose for a pose with one (global) parent: pose_3D.hpp:102-110
inline Pose global_pose(const Pose* parent, const Pose& local) {
  if (!parent) return local;
  Pose result = *parent;
  result.Position += result.Q * local.Position;
  result.Q = result.Q * local.Q;
  return result;
}

// ---------------------------------------------------------------- frame_3D<double>
// core/kinetostatics/frame_3D.hpp.  All chain frames are parentless (test_bm.cpp:45-49 creates
// the base frame with no Parent; revolute_joint.cpp:125 and frame_3D.hpp:258-275 propagate it).
struct Frame {
  V3 Position;
  Quat Q;
  V3 Velocity, AngVelocity, Acceleration, AngAcceleration, Force, Torque;

  // addBefore(const pose_3D&): frame_3D.hpp:240-255  (used by operator*(frame, pose) :358-362)
  Frame& addBefore(const Pose& aPose) {
    RotMat R(Q.getRotMat());
    Position += R * aPose.Position;
    Velocity += R * cross(AngVelocity, aPose.Position);
    Acceleration += R * (cross(AngVelocity, cross(AngVelocity, aPose.Position)) +
                         cross(AngAcceleration, aPose.Position));
    RotMat R2(aPose.Q.getRotMat());
    Q = Q * aPose.Q;
    AngAcceleration = (AngAcceleration * R2);
    AngVelocity = (AngVelocity * R2);
    return *this;
  }
  // addBefore(const frame_3D&): frame_3D.hpp:222-238 (used by operator*(frame, frame) :348-352)
  Frame& addBefore(const Frame& aFrame) {
    RotMat R(Q.getRotMat());
    Position += R * aFrame.Position;
    Velocity += R * (cross(AngVelocity, aFrame.Position) + aFrame.Velocity);
    Acceleration += R * (cross(AngVelocity, cross(AngVelocity, aFrame.Position)) +
                         cross(2.0 * AngVelocity, aFrame.Velocity) +
                         cross(AngAcceleration, aFrame.Position) + aFrame.Acceleration);
    RotMat R2(aFrame.Q.getRotMat());
    Q = Q * aFrame.Q;
    AngAcceleration = (AngAcceleration * R2) + cross(AngVelocity * R2, aFrame.AngVelocity) +
                      aFrame.AngAcceleration;
    AngVelocity = (AngVelocity * R2) + aFrame.AngVelocity;
    return *this;
  }
  // operator~ : frame_3D.hpp:368-382
  Frame inverse() const {
    RotMat R(Q.getRotMat());
    Frame result;
    result.Q = invert(Q);
    result.AngVelocity = R * (-AngVelocity);
    result.AngAcceleration = R * (-AngAcceleration);
    result.Position = (-Position) * R;
    result.Velocity = (-(cross(result.AngVelocity, Position) + Velocity)) * R;
    result.Acceleration = (-(cross(result.AngVelocity, cross(result.AngVelocity, Position)) +
                             cross(2.0 * result.AngVelocity, Velocity) +
                             cross(result.AngAcceleration, Position) + Acceleration)) * R;
    result.Force = (-Force) * R;
    result.Torque = R * (-Torque);
    return result;
  }
};

// ---------------------------------------------------------------- Cholesky
// core/lin_alg/mat_cholesky.hpp.  Dense row-major N x N helper.
struct singularity_error : public std::runtime_error {
  singularity_error() : std::runtime_error("A") {}
};

// decompose_Cholesky_impl: mat_cholesky.hpp:63-84 (throws if L(i,i) < NumTol before sqrt)
inline void decompose_Cholesky(const double* A, double* L, int N, double NumTol) {
  for (int i = 0; i < N; ++i) {
    for (int j = 0; j < i; ++j) {
      L[i * N + j] = A[i * N + j];
      for (int k = 0; k < j; ++k) L[i * N + j] -= L[i * N + k] * L[j * N + k];
      L[i * N + j] /= L[j * N + j];
    }
    L[i * N + i] = A[i * N + i];
    for (int k = 0; k < i; ++k) L[i * N + i] -= L[i * N + k] * L[i * N + k];
    if (L[i * N + i] < NumTol) throw singularity_error();
    L[i * N + i] = std::sqrt(L[i * N + i]);
  }
}
// backsub_Cholesky_impl: mat_cholesky.hpp:160-178 (single right-hand-side column)
inline void backsub_Cholesky(const double* L, double* B, int N) {
  for (int i = 0; i < N; ++i) {
    for (int k = 0; k < i; ++k) B[i] -= L[i * N + k] * B[k];
    B[i] /= L[i * N + i];
  }
  for (int i = N - 1; i >= 0; --i) {
    for (int k = N - 1; k > i; --k) B[i] -= L[k * N + i] * B[k];
    B[i] /= L[i * N + i];
  }
}
// linsolve_Cholesky: mat_cholesky.hpp:546-554 (NumTol = 1e-8; L zero-initialised)
inline void linsolve_Cholesky(const double* A, double* b, int N, double NumTol = 1E-8) {
  std::vector<double> L(static_cast<std::size_t>(N) * N, 0.0);
  decompose_Cholesky(A, L.data(), N, NumTol);
  backsub_Cholesky(L.data(), b, N);
}

}  // namespace oracle
#endif
