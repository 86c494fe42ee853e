// device_math.h -- fp64 leaf math for the HIP kernels (gfx950).
//
// Operation order follows ReaK's core/kinetostatics (rotations_3D.hpp, frame_3D.hpp, pose_3D.hpp)
// so that, compiled with -ffp-contract=off, every product/sum rounds like the CPU reference.
// Only sin/cos (OCML vs glibc) can differ, by ulps.
#pragma once
#include <hip/hip_runtime.h>

namespace rkh {

struct d3 {
  double x, y, z;
};
struct d4 {  // quaternion (w, x, y, z)
  double w, x, y, z;
};
struct m33 {  // row-major 3x3
  double a11, a12, a13, a21, a22, a23, a31, a32, a33;
};

#define RKH_DI __device__ __forceinline__

RKH_DI d3 mk3(double x, double y, double z) { return d3{x, y, z}; }
RKH_DI d3 operator+(d3 a, d3 b) { return d3{a.x + b.x, a.y + b.y, a.z + b.z}; }
RKH_DI d3 operator-(d3 a, d3 b) { return d3{a.x - b.x, a.y - b.y, a.z - b.z}; }
RKH_DI d3 operator-(d3 a) { return d3{-a.x, -a.y, -a.z}; }
RKH_DI d3 operator*(double s, d3 a) { return d3{a.x * s, a.y * s, a.z * s}; }
RKH_DI d3 operator*(d3 a, double s) { return d3{a.x * s, a.y * s, a.z * s}; }
// vect dot: result(0); result += a[i]*b[i]   (vect_alg.hpp:2547-2555)
RKH_DI double dot(d3 a, d3 b) { return ((0.0 + a.x * b.x) + a.y * b.y) + a.z * b.z; }
// vect cross (vect_alg.hpp:1214-1221)
RKH_DI d3 cross(d3 a, d3 b) { return d3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
RKH_DI double norm_2(d3 v) { return sqrt(((0.0 + v.x * v.x) + v.y * v.y) + v.z * v.z); }

// rot_mat_3D * V (rotations_3D.hpp:372-376)
RKH_DI d3 mul(const m33& R, d3 V) {
  return d3{R.a11 * V.x + R.a12 * V.y + R.a13 * V.z, R.a21 * V.x + R.a22 * V.y + R.a23 * V.z,
            R.a31 * V.x + R.a32 * V.y + R.a33 * V.z};
}
// V * rot_mat_3D = R^T V (rotations_3D.hpp:379-383)
RKH_DI d3 mulT(d3 V, const m33& R) {
  return d3{R.a11 * V.x + R.a21 * V.y + R.a31 * V.z, R.a12 * V.x + R.a22 * V.y + R.a32 * V.z,
            R.a13 * V.x + R.a23 * V.y + R.a33 * V.z};
}
// quaternion::getRotMat (rotations_3D.hpp:986-999)
RKH_DI m33 rotmat(d4 q) {
  double t01 = 2.0 * q.w * q.x, t02 = 2.0 * q.w * q.y, t03 = 2.0 * q.w * q.z;
  double t11 = 2.0 * q.x * q.x, t12 = 2.0 * q.x * q.y, t13 = 2.0 * q.x * q.z;
  double t22 = 2.0 * q.y * q.y, t23 = 2.0 * q.y * q.z, t33 = 2.0 * q.z * q.z;
  return m33{1.0 - t22 - t33, t12 - t03, t02 + t13, t12 + t03, 1.0 - t11 - t33, t23 - t01,
             t13 - t02, t01 + t23, 1.0 - t11 - t22};
}
// Q1 * Q2 (rotations_3D.hpp:1093-1098)
RKH_DI d4 qmul(d4 a, d4 b) {
  return d4{b.w * a.w - b.x * a.x - b.y * a.y - b.z * a.z, b.w * a.x + b.z * a.y - b.y * a.z + b.x * a.w,
            b.w * a.y - b.z * a.x + b.x * a.z + b.y * a.w, b.w * a.z + b.y * a.x - b.x * a.y + b.z * a.w};
}
RKH_DI d4 qinv(d4 q) { return d4{q.w, -q.x, -q.y, -q.z}; }
// Q * V (rotations_3D.hpp:1137-1151)
RKH_DI d3 qrot(d4 Q, d3 V) {
  double t0 = Q.w * Q.x, t1 = Q.w * Q.y, t2 = Q.w * Q.z, t3 = -Q.x * Q.x, t4 = Q.x * Q.y, t5 = Q.x * Q.z,
         t6 = -Q.y * Q.y, t7 = Q.y * Q.z, t8 = -Q.z * Q.z;
  return d3{2.0 * ((t6 + t8) * V.x + (t4 - t2) * V.y + (t1 + t5) * V.z) + V.x,
            2.0 * ((t2 + t4) * V.x + (t3 + t8) * V.y + (t7 - t0) * V.z) + V.y,
            2.0 * ((t5 - t1) * V.x + (t0 + t7) * V.y + (t3 + t6) * V.z) + V.z};
}
// pose_3D::transformToParent / transformFromParent (pose_3D.hpp:175-177,189-191)
RKH_DI d3 pose_to_parent(d3 pos, d4 q, d3 V) { return pos + qrot(q, V); }
RKH_DI d3 pose_from_parent(d3 pos, d4 q, d3 V) { return qrot(qinv(q), V - pos); }

// symmetric 3x3 (a11,a12,a13,a22,a23,a33) times vector, summation order of
// mat_alg_symmetric.hpp:646-659
RKH_DI d3 sym_mul(const double* t, d3 V) {
  d3 r;
  r.x = ((0.0 + t[0] * V.x) + t[1] * V.y) + t[2] * V.z;
  r.y = ((0.0 + t[1] * V.x) + t[3] * V.y) + t[4] * V.z;
  r.z = ((0.0 + t[2] * V.x) + t[4] * V.y) + t[5] * V.z;
  return r;
}

// flexible_beam_3D::doForce without an object frame (ctrl/mbd_kte/flexible_beam.cpp:176-186), anchor 1 only: force and
// torque the beam adds to the frame (pos1, q1) given the world anchor (pos2, q2); axis_angle(quaternion) as in
// rotations_3D.hpp:1986-2006 (unit(v) divides by the norm; threshold 1e-7)
RKH_DI void beam_force(d3 pos1, d4 q1, d3 pos2, d4 q2, double rest, double k, double kt, d3* F, d3* T) {
  const d3 diff = pos1 - pos2;
  const d3 diff_a1 = qrot(qinv(q1), -diff) - mk3(rest, 0.0, 0.0);
  const d4 dq = qmul(qinv(q1), q2);
  double v0 = dq.w, v1 = dq.x, v2 = dq.y, v3 = dq.z;
  const double nrm = sqrt(((v0 * v0 + v1 * v1) + v2 * v2) + v3 * v3);
  v0 = v0 / nrm; v1 = v1 / nrm; v2 = v2 / nrm; v3 = v3 / nrm;
  const double tmp = sqrt(v1 * v1 + v2 * v2 + v3 * v3);
  d3 axis = mk3(1.0, 0.0, 0.0);
  double angle = 0.0;
  if (tmp > 0.0000001) {
    axis = mk3(v1 / tmp, v2 / tmp, v3 / tmp);
    if (v0 < 0.0) {
      angle = 2.0 * acos(-v0);
      axis = -axis;
    } else {
      angle = 2.0 * acos(v0);
    }
  }
  *F = k * diff_a1;
  *T = (kt * angle) * axis;
}

// the same beam on its anchor 2 (a chain frame): flexible_beam.cpp:178,184-185
RKH_DI void beam_force_anchor2(d3 pos1, d4 q1, d3 pos2, d4 q2, double rest, double k, double kt, d3* F, d3* T) {
  const d3 diff = pos1 - pos2;
  const d3 diff_a2 = qrot(qinv(q2), diff) + mk3(rest, 0.0, 0.0);
  d3 F1, T1;
  beam_force(pos1, q1, pos2, q2, rest, k, kt, &F1, &T1);  // for the torsion term (angle * axis of anchor 1's frame)
  *F = k * diff_a2;
  *T = mk3(0, 0, 0) - T1;
}

// wave64 broadcast of a double from lane `src`
RKH_DI double bcast(double v, int src) { return __shfl(v, src, 64); }

}  // namespace rkh
