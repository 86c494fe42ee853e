// ORACLE -- TEST INFRASTRUCTURE ONLY (see reak_math.hpp header).
//
// CPU restatement of ReaK's 3D proximity queries on the planning hot path:
// proxy_query_pair_3D::{createProxFinderList,findMinimumDistance} and the closed-form pair
// routines for sphere / box / capped_cylinder (geometry/proximity/*.cpp).
//
// Parity pin status: the reference holds no expected outputs for these routines
// (geometry/proximity/test_nlp_proximity.cpp prints only) -> pinned by analytic cases
// (axis-aligned configurations with hand-computed distances) in tests/test_oracle_kat.py;
// otherwise "parity unpinned" beyond the line-by-line restatement.
#ifndef REAK_ORACLE_PROXIMITY_HPP
#define REAK_ORACLE_PROXIMITY_HPP

#include <algorithm>
#include <cmath>
#include <limits>
#include <vector>

#include "../include/rkh_types.h"
#include "reak_gjk.hpp"
#include "reak_kte.hpp"
#include "reak_math.hpp"

namespace oracle {

// proximity_record_3D: geometry/proximity/proximity_record_3D.hpp:47-56
struct ProxRecord {
  V3 mPoint1, mPoint2;
  double mDistance = std::numeric_limits<double>::infinity();
};

// A shape with its resolved global pose (geometry_3D::getPose().getGlobalPose()).
struct ShapeG {
  int kind;
  Pose g;  // global pose
  double dims[3];
  const double* mesh_pool = nullptr;  // RKH_SHAPE_MESH: the scene's vertex pool (dims[0] = first vertex, dims[1] = count)
  // shape_3D::getBoundingRadius: sphere.cpp:31, box.cpp:31, capped_cylinder.cpp:30
  double getBoundingRadius() const {
    switch (kind) {
      case RKH_SHAPE_SPHERE: return dims[0];
      case RKH_SHAPE_BOX: {
        double s = 0.0;
        for (int i = 0; i < 3; ++i) s += dims[i] * dims[i];
        return std::sqrt(s) * 0.5;
      }
      case RKH_SHAPE_CCYLINDER: return dims[0] * 0.5 + dims[1];
      case RKH_SHAPE_PLANE: {  // plane.cpp:31-33: norm_2(mDimensions) * 0.5
        double s = 0.0;
        for (int i = 0; i < 2; ++i) s += dims[i] * dims[i];
        return std::sqrt(s) * 0.5;
      }
      case RKH_SHAPE_CYLINDER:  // cylinder.cpp:33-35
        return std::sqrt(dims[1] * dims[1] + 0.25 * dims[0] * dims[0]);
      case RKH_SHAPE_MESH: {  // the build's convex vertex set: radius about the local origin
        double r2 = 0.0;
        for (int i = 0; i < int(dims[1]); ++i) {
          const double* v = mesh_pool + 3 * (std::size_t(dims[0]) + i);
          r2 = std::max(r2, v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        }
        return std::sqrt(r2);
      }
    }
    return 0.0;
  }
};

// findProximityBoxToPoint: geometry/proximity/prox_fundamentals_3D.cpp:35-82
inline ProxRecord findProximityBoxToPoint(const ShapeG& aBox, const V3& aPoint) {
  V3 pt_rel = aBox.g.transformFromParent(aPoint);
  const double* d = aBox.dims;
  bool in_x_range = ((pt_rel[0] > -0.5 * d[0]) && (pt_rel[0] < 0.5 * d[0]));
  bool in_y_range = ((pt_rel[1] > -0.5 * d[1]) && (pt_rel[1] < 0.5 * d[1]));
  bool in_z_range = ((pt_rel[2] > -0.5 * d[2]) && (pt_rel[2] < 0.5 * d[2]));
  bool is_inside = (in_x_range && in_y_range && in_z_range);
  if (is_inside) {
    V3 bound_dists(0.5 * d[0] - std::fabs(pt_rel[0]), 0.5 * d[1] - std::fabs(pt_rel[1]),
                   0.5 * d[2] - std::fabs(pt_rel[2]));
    if ((bound_dists[0] <= bound_dists[1]) && (bound_dists[0] <= bound_dists[2])) {
      in_x_range = false;
    } else if ((bound_dists[1] <= bound_dists[0]) && (bound_dists[1] <= bound_dists[2])) {
      in_y_range = false;
    } else {
      in_z_range = false;
    }
  }
  V3 corner_pt(0.5 * d[0], 0.5 * d[1], 0.5 * d[2]);
  if (in_x_range) corner_pt[0] = pt_rel[0];
  else if (pt_rel[0] < 0.0) corner_pt[0] = -corner_pt[0];
  if (in_y_range) corner_pt[1] = pt_rel[1];
  else if (pt_rel[1] < 0.0) corner_pt[1] = -corner_pt[1];
  if (in_z_range) corner_pt[2] = pt_rel[2];
  else if (pt_rel[2] < 0.0) corner_pt[2] = -corner_pt[2];
  ProxRecord result;
  result.mPoint1 = aBox.g.transformToParent(corner_pt);
  double diff_d = norm_2(corner_pt - pt_rel);
  result.mPoint2 = aPoint;
  result.mDistance = (is_inside ? -diff_d : diff_d);
  return result;
}

// golden_section_search_impl: core/optimization/line_search.hpp:71-95 (GoldenRatioPhi = 1.618033988).
// The reference loop has no iteration cap; RKH_GOLDEN_MAX_ITER guards the (never observed)
// non-terminating case identically here and in the HIP kernel.
static const int RKH_GOLDEN_MAX_ITER = 256;
template <typename F>
double golden_section_search(F f, double& low_bound, double& up_bound, double tol) {
  const double GoldenRatioPhi = 1.618033988;
  double mid_value = low_bound + (up_bound - low_bound) / GoldenRatioPhi;
  double mid_cost = f(mid_value);
  for (int it = 0;; ++it) {
    if (std::fabs(low_bound - up_bound) < tol || it >= RKH_GOLDEN_MAX_ITER)
      return f((low_bound + up_bound) * 0.5);
    double test_value = mid_value + (up_bound - mid_value) / GoldenRatioPhi;
    double test_cost = f(test_value);
    if (test_cost < mid_cost) {
      low_bound = mid_value;
      mid_value = test_value;
      mid_cost = test_cost;
    } else {
      up_bound = low_bound;
      low_bound = test_value;
    }
  }
}

// findProximityBoxToLine: prox_fundamentals_3D.cpp:108-115 (+ ProxBoxToLineFunctor :87-104)
inline ProxRecord findProximityBoxToLine(const ShapeG& aBox, const V3& aCenter, const V3& aTangent,
                                         double aHalfLength) {
  ProxRecord result;
  auto fct = [&](double t) {
    result = findProximityBoxToPoint(aBox, aCenter + aTangent * t);
    return result.mDistance;
  };
  double lb = -aHalfLength;
  double ub = aHalfLength;
  golden_section_search(fct, lb, ub, 1e-3 * aHalfLength);
  return result;
}

// prox_sphere_sphere::computeProximity: prox_sphere_sphere.cpp:41-58
inline ProxRecord prox_sphere_sphere(const ShapeG& s1, const ShapeG& s2) {
  ProxRecord r;
  V3 c1 = s1.g.transformToParent(V3(0, 0, 0));
  V3 c2 = s2.g.transformToParent(V3(0, 0, 0));
  V3 diff_cc = c2 - c1;
  double dist_cc = norm_2(diff_cc);
  r.mDistance = dist_cc - s1.dims[0] - s2.dims[0];
  r.mPoint1 = c1 + (s1.dims[0] / dist_cc) * diff_cc;
  r.mPoint2 = c2 - (s2.dims[0] / dist_cc) * diff_cc;
  return r;
}

// prox_sphere_box::computeProximity: prox_sphere_box.cpp:45-68
inline ProxRecord prox_sphere_box(const ShapeG& sp, const ShapeG& bx) {
  ProxRecord r;
  V3 sp_c = sp.g.transformToParent(V3(0, 0, 0));
  ProxRecord bxpt = findProximityBoxToPoint(bx, sp_c);
  V3 diff_v = bxpt.mPoint1 - bxpt.mPoint2;
  double diff_d = norm_2(diff_v);
  if (bxpt.mDistance < 0.0) r.mPoint1 = bxpt.mPoint2 - (sp.dims[0] / diff_d) * diff_v;
  else r.mPoint1 = bxpt.mPoint2 + (sp.dims[0] / diff_d) * diff_v;
  r.mPoint2 = bxpt.mPoint1;
  r.mDistance = bxpt.mDistance - sp.dims[0];
  return r;
}

// prox_sphere_ccylinder::computeProximity: prox_sphere_ccylinder.cpp:43-80
inline ProxRecord prox_sphere_ccylinder(const ShapeG& sp, const ShapeG& cc) {
  ProxRecord r;
  const double cc_len = cc.dims[0], cc_rad = cc.dims[1], sp_rad = sp.dims[0];
  V3 sp_c = sp.g.transformToParent(V3(0, 0, 0));
  V3 sp_c_rel = cc.g.transformFromParent(sp_c);
  if (std::fabs(sp_c_rel[2]) <= 0.5 * cc_len) {
    V3 sp_c_proj(sp_c_rel[0], sp_c_rel[1], 0.0);
    double sp_c_proj_d = norm_2(sp_c_proj);
    r.mPoint2 = cc.g.transformToParent(V3(0.0, 0.0, sp_c_rel[2]) + sp_c_proj * (cc_rad / sp_c_proj_d));
    r.mPoint1 = cc.g.transformToParent(sp_c_rel - sp_c_proj * (sp_rad / sp_c_proj_d));
    r.mDistance = sp_c_proj_d - sp_rad - cc_rad;
  } else {
    double fact = 1.0;
    if (sp_c_rel[2] < 0.0) fact = -1.0;
    V3 cy_c2 = cc.g.transformToParent(V3(0.0, 0.0, fact * 0.5 * cc_len));
    V3 diff_cc = cy_c2 - sp_c;
    double dist_cc = norm_2(diff_cc);
    r.mDistance = dist_cc - sp_rad - cc_rad;
    r.mPoint1 = sp_c + (sp_rad / dist_cc) * diff_cc;
    r.mPoint2 = cy_c2 - (cc_rad / dist_cc) * diff_cc;
  }
  return r;
}

// prox_ccylinder_ccylinder::computeProximity: prox_ccylinder_ccylinder.cpp:43-128
// (the parallel-branch overlap test at :61-62 is a logical OR in the reference and is kept as is)
inline ProxRecord prox_ccylinder_ccylinder(const ShapeG& c1, const ShapeG& c2) {
  ProxRecord r;
  const double L1 = c1.dims[0], R1 = c1.dims[1], L2 = c2.dims[0], R2 = c2.dims[1];
  V3 cy2_c = c2.g.transformToParent(V3(0, 0, 0));
  V3 cy2_t = c2.g.Q * V3(0.0, 0.0, 1.0);
  V3 cy2_c_rel = c1.g.transformFromParent(cy2_c);
  V3 cy2_t_rel = invert(c1.g.Q) * cy2_t;
  if (std::sqrt(cy2_t_rel[0] * cy2_t_rel[0] + cy2_t_rel[1] * cy2_t_rel[1]) < 1e-5) {
    if ((cy2_c_rel[2] + 0.5 * L2 > -0.5 * L1) || (cy2_c_rel[2] - 0.5 * L2 < 0.5 * L1)) {
      double max_z_rel = ((cy2_c_rel[2] + 0.5 * L2 < 0.5 * L1) ? (cy2_c_rel[2] + 0.5 * L2) : (0.5 * L1));
      double min_z_rel = ((cy2_c_rel[2] - 0.5 * L2 > -0.5 * L1) ? (cy2_c_rel[2] - 0.5 * L2) : (-0.5 * L1));
      double avg_z_rel = (max_z_rel + min_z_rel) * 0.5;
      V3 cy2_r_rel = unit(V3(cy2_c_rel[0], cy2_c_rel[1], 0.0));
      r.mPoint1 = c1.g.transformToParent(V3(R1 * cy2_r_rel[0], R1 * cy2_r_rel[1], avg_z_rel));
      r.mPoint2 = c1.g.transformToParent(
          V3(cy2_c_rel[0] - R2 * cy2_r_rel[0], cy2_c_rel[1] - R2 * cy2_r_rel[1], avg_z_rel));
      r.mDistance = std::sqrt(cy2_c_rel[0] * cy2_c_rel[0] + cy2_c_rel[1] * cy2_c_rel[1]) - R1 - R2;
      return r;
    }
    V3 cy1_spc_rel(0.0, 0.0, 0.0);
    V3 cy2_spc_rel = cy2_c_rel;
    if (cy2_c_rel[2] < 0.0) {
      cy1_spc_rel[2] -= 0.5 * L1;
      cy2_spc_rel[2] += 0.5 * L2;
    } else {
      cy1_spc_rel[2] += 0.5 * L1;
      cy2_spc_rel[2] -= 0.5 * L2;
    }
    V3 diff_v_rel = cy2_spc_rel - cy1_spc_rel;
    double dist_v_rel = norm_2(diff_v_rel);
    r.mPoint1 = c1.g.transformToParent(cy1_spc_rel + (R1 / dist_v_rel) * diff_v_rel);
    r.mPoint2 = c1.g.transformToParent(cy2_spc_rel - (R2 / dist_v_rel) * diff_v_rel);
    r.mDistance = dist_v_rel - R1 - R2;
    return r;
  }
  double d = dot(cy2_t_rel, cy2_c_rel);
  double denom = 1.0 - cy2_t_rel[2] * cy2_t_rel[2];
  double s_c = (cy2_t_rel[2] * cy2_c_rel[2] - d) / denom;
  double t_c = (cy2_c_rel[2] - cy2_t_rel[2] * d) / denom;
  if (s_c < -0.5 * L2) {
    s_c = -0.5 * L2;
    t_c = cy2_c_rel[2] - 0.5 * L2 * cy2_t_rel[2];
  } else if (s_c > 0.5 * L2) {
    s_c = 0.5 * L2;
    t_c = cy2_c_rel[2] + 0.5 * L2 * cy2_t_rel[2];
  }
  if (t_c < -0.5 * L1) {
    t_c = -0.5 * L1;
    s_c = -0.5 * L1 * cy2_t_rel[2] - d;
  } else if (t_c > 0.5 * L1) {
    t_c = 0.5 * L1;
    s_c = 0.5 * L1 * cy2_t_rel[2] - d;
  }
  if (s_c < -0.5 * L2) s_c = -0.5 * L2;
  else if (s_c > 0.5 * L2) s_c = 0.5 * L2;
  V3 cy1_ptc(0.0, 0.0, t_c);
  V3 cy2_ptc = cy2_c_rel + s_c * cy2_t_rel;
  V3 diff_v_rel = cy2_ptc - cy1_ptc;
  double dist_v_rel = norm_2(diff_v_rel);
  r.mPoint1 = c1.g.transformToParent(cy1_ptc + (R1 / dist_v_rel) * diff_v_rel);
  r.mPoint2 = c1.g.transformToParent(cy2_ptc - (R2 / dist_v_rel) * diff_v_rel);
  r.mDistance = dist_v_rel - R1 - R2;
  return r;
}

// prox_ccylinder_box::computeProximity: prox_ccylinder_box.cpp:45-70
inline ProxRecord prox_ccylinder_box(const ShapeG& cc, const ShapeG& bx) {
  ProxRecord r;
  V3 cy_c = cc.g.transformToParent(V3(0, 0, 0));
  V3 cy_t = cc.g.Q * V3(0.0, 0.0, 1.0);
  ProxRecord bxln = findProximityBoxToLine(bx, cy_c, cy_t, 0.5 * cc.dims[0]);
  V3 diff_v = bxln.mPoint1 - bxln.mPoint2;
  double diff_d = norm_2(diff_v);
  if (bxln.mDistance < 0.0) r.mPoint1 = bxln.mPoint2 - (cc.dims[1] / diff_d) * diff_v;
  else r.mPoint1 = bxln.mPoint2 + (cc.dims[1] / diff_d) * diff_v;
  r.mPoint2 = bxln.mPoint1;
  r.mDistance = bxln.mDistance - cc.dims[1];
  return r;
}

// ---- the plane / cylinder finders enabled in createProxFinderList (proxy_query_model.cpp:226-300) ----------------
// pose_3D::rotateToGlobal = Quat * V, rotateFromGlobal = invert(Quat) * V (pose_3D.hpp:154-170)

// prox_plane_sphere::computeProximity: prox_plane_sphere.cpp:106-122 (the infinite-plane version; the finite one is
// commented out in the reference)
inline ProxRecord prox_plane_sphere(const ShapeG& pl, const ShapeG& sp) {
  ProxRecord r;
  V3 sp_c = sp.g.transformToParent(V3(0, 0, 0));
  V3 sp_c_rel = pl.g.transformFromParent(sp_c);
  r.mPoint1 = pl.g.transformToParent(V3(sp_c_rel[0], sp_c_rel[1], 0.0));
  r.mPoint2 = pl.g.transformToParent(V3(sp_c_rel[0], sp_c_rel[1], sp_c_rel[2] - sp.dims[0]));
  r.mDistance = sp_c_rel[2] - sp.dims[0];
  return r;
}

// prox_plane_box::computeProximity: prox_plane_box.cpp:43-71.  The reference builds bx_x, bx_y and bx_z all from the
// box's local x axis (1,0,0) (:53-55); kept as is.
inline ProxRecord prox_plane_box(const ShapeG& pl, const ShapeG& bx) {
  ProxRecord r;
  V3 bx_c = bx.g.transformToParent(V3(0, 0, 0));
  V3 bx_x = invert(pl.g.Q) * (bx.g.Q * V3(1.0, 0.0, 0.0));
  V3 bx_y = invert(pl.g.Q) * (bx.g.Q * V3(1.0, 0.0, 0.0));
  V3 bx_z = invert(pl.g.Q) * (bx.g.Q * V3(1.0, 0.0, 0.0));
  if (bx_x[2] > 0.0) bx_x = -bx_x;
  if (bx_y[2] > 0.0) bx_y = -bx_y;
  if (bx_z[2] > 0.0) bx_z = -bx_z;
  V3 bx_c_rel = pl.g.transformFromParent(bx_c);
  V3 bx_pt_rel = bx_c_rel + 0.5 * (bx.dims[0] * bx_x + bx.dims[1] * bx_y + bx.dims[2] * bx_z);
  r.mPoint1 = pl.g.transformToParent(V3(bx_pt_rel[0], bx_pt_rel[1], 0.0));
  r.mPoint2 = pl.g.transformToParent(bx_pt_rel);
  r.mDistance = bx_pt_rel[2];
  return r;
}

// prox_plane_ccylinder::computeProximity: prox_plane_ccylinder.cpp:43-74
inline ProxRecord prox_plane_ccylinder(const ShapeG& pl, const ShapeG& cc) {
  ProxRecord r;
  const double L = cc.dims[0], R = cc.dims[1];
  V3 cy_c = cc.g.transformToParent(V3(0, 0, 0));
  V3 cy_t = cc.g.Q * V3(0.0, 0.0, 1.0);
  V3 cy_c_rel = pl.g.transformFromParent(cy_c);
  V3 cy_t_rel = invert(pl.g.Q) * cy_t;
  if (std::fabs(cy_t_rel[2]) < 1e-6) {
    r.mPoint1 = pl.g.transformToParent(V3(cy_c_rel[0], cy_c_rel[1], 0.0));
    r.mPoint2 = pl.g.transformToParent(V3(cy_c_rel[0], cy_c_rel[1], cy_c_rel[2] - R));
    r.mDistance = cy_c_rel[2] - R;
  } else {
    if (cy_t_rel[2] > 0.0) cy_t_rel = -cy_t_rel;
    V3 cypt_rel = cy_c_rel + (0.5 * L) * cy_t_rel + V3(0.0, 0.0, -R);
    r.mPoint1 = pl.g.transformToParent(V3(cypt_rel[0], cypt_rel[1], 0.0));
    r.mPoint2 = pl.g.transformToParent(cypt_rel);
    r.mDistance = cypt_rel[2];
  }
  return r;
}

// prox_plane_cylinder::computeProximity: prox_plane_cylinder.cpp:42-78
inline ProxRecord prox_plane_cylinder(const ShapeG& pl, const ShapeG& cy) {
  ProxRecord r;
  const double L = cy.dims[0], R = cy.dims[1];
  V3 cy_c = cy.g.transformToParent(V3(0, 0, 0));
  V3 cy_t = cy.g.Q * V3(0.0, 0.0, 1.0);
  V3 cy_c_rel = pl.g.transformFromParent(cy_c);
  V3 cy_t_rel = invert(pl.g.Q) * cy_t;
  if (std::fabs(cy_t_rel[2]) < 1e-6) {
    r.mPoint1 = pl.g.transformToParent(V3(cy_c_rel[0], cy_c_rel[1], 0.0));
    r.mPoint2 = pl.g.transformToParent(V3(cy_c_rel[0], cy_c_rel[1], cy_c_rel[2] - R));
    r.mDistance = cy_c_rel[2] - R;
  } else if (std::sqrt(cy_t_rel[0] * cy_t_rel[0] + cy_t_rel[1] * cy_t_rel[1]) < 1e-6) {
    r.mPoint1 = pl.g.transformToParent(V3(cy_c_rel[0], cy_c_rel[1], 0.0));
    r.mPoint2 = pl.g.transformToParent(V3(cy_c_rel[0], cy_c_rel[1], cy_c_rel[2] - 0.5 * L));
    r.mDistance = cy_c_rel[2] - 0.5 * L;
  } else {
    if (cy_t_rel[2] > 0.0) cy_t_rel = -cy_t_rel;
    V3 cy_r_rel = unit(V3(0.0, 0.0, -1.0) + cy_t_rel[2] * cy_t_rel);
    V3 cypt_rel = cy_c_rel + (0.5 * L) * cy_t_rel + R * cy_r_rel;
    r.mPoint1 = pl.g.transformToParent(V3(cypt_rel[0], cypt_rel[1], 0.0));
    r.mPoint2 = pl.g.transformToParent(cypt_rel);
    r.mDistance = cypt_rel[2];
  }
  return r;
}

// prox_plane_plane::computeProximityOfPoint: prox_plane_plane.cpp:43-95 (here the plane IS finite)
inline void plane_proximity_of_point(const ShapeG& pl, const V3& aPoint, V3& aPointRec, double& aDistance) {
  V3 pt_rel = pl.g.transformFromParent(aPoint);
  const double hx = 0.5 * pl.dims[0], hy = 0.5 * pl.dims[1];
  if ((pt_rel[0] > -hx) && (pt_rel[0] < hx) && (pt_rel[1] > -hy) && (pt_rel[1] < hy)) {
    double fact = 1.0;
    if (pt_rel[2] < 0.0) fact = -1.0;
    aPointRec = pl.g.transformToParent(V3(pt_rel[0], pt_rel[1], 0.0));
    aDistance = fact * pt_rel[2];
  } else {
    if ((pt_rel[0] > -hx) && (pt_rel[0] < hx)) {
      double fact = 1.0;
      if (pt_rel[1] < 0.0) fact = -1.0;
      aPointRec = pl.g.transformToParent(V3(pt_rel[0], fact * 0.5 * pl.dims[1], 0.0));
    } else if ((pt_rel[1] > -hy) && (pt_rel[1] < hy)) {
      double fact = 1.0;
      if (pt_rel[0] < 0.0) fact = -1.0;
      aPointRec = pl.g.transformToParent(V3(fact * 0.5 * pl.dims[0], pt_rel[1], 0.0));
    } else {
      V3 rim_pt(0.5 * pl.dims[0], 0.5 * pl.dims[1], 0.0);
      if (pt_rel[0] < 0.0) rim_pt[0] = -rim_pt[0];
      if (pt_rel[1] < 0.0) rim_pt[1] = -rim_pt[1];
      aPointRec = pl.g.transformToParent(rim_pt);
    }
    aDistance = norm_2(aPointRec - aPoint);
  }
}

// prox_plane_plane::computeProximity: prox_plane_plane.cpp:98-183 (the four corners of plane 2 against plane 1, then
// the four corners of plane 1 against plane 2; corner order (+,+), (+,-), (-,-), (-,+))
inline ProxRecord prox_plane_plane(const ShapeG& p1, const ShapeG& p2) {
  ProxRecord r;  // mDistance = +inf
  V3 temp_pt;
  double temp_dist;
  for (int side = 0; side < 2; ++side) {
    const ShapeG& of = side == 0 ? p2 : p1;       // whose corners
    const ShapeG& against = side == 0 ? p1 : p2;  // tested against
    V3 corner(0.5 * of.dims[0], 0.5 * of.dims[1], 0.0);
    for (int k = 0; k < 4; ++k) {
      if (k == 1 || k == 3) corner[1] = -corner[1];
      if (k == 2) corner[0] = -corner[0];
      V3 corner_gbl = of.g.transformToParent(corner);
      plane_proximity_of_point(against, corner_gbl, temp_pt, temp_dist);
      if (temp_dist < r.mDistance) {
        r.mDistance = temp_dist;
        if (side == 0) { r.mPoint1 = temp_pt; r.mPoint2 = corner_gbl; }
        else { r.mPoint2 = temp_pt; r.mPoint1 = corner_gbl; }
      }
    }
  }
  return r;
}

// prox_sphere_cylinder::computeProximity: prox_sphere_cylinder.cpp:43-92
inline ProxRecord prox_sphere_cylinder(const ShapeG& sp, const ShapeG& cy) {
  ProxRecord r;
  const double L = cy.dims[0], R = cy.dims[1], sr = sp.dims[0];
  V3 sp_c = sp.g.transformToParent(V3(0, 0, 0));
  V3 sp_c_rel = cy.g.transformFromParent(sp_c);
  double sp_c_rel_rad = std::sqrt(sp_c_rel[0] * sp_c_rel[0] + sp_c_rel[1] * sp_c_rel[1]);
  if (std::fabs(sp_c_rel[2]) <= 0.5 * L) {
    V3 sp_c_proj(sp_c_rel[0], sp_c_rel[1], 0.0);
    double sp_c_proj_d = norm_2(sp_c_proj);
    r.mPoint2 = cy.g.transformToParent(V3(0.0, 0.0, sp_c_rel[2]) + sp_c_proj * (R / sp_c_proj_d));
    r.mPoint1 = cy.g.transformToParent(sp_c_rel - sp_c_proj * (sr / sp_c_proj_d));
    r.mDistance = sp_c_proj_d - sr - R;
  } else if (sp_c_rel_rad < R) {
    double fact = 1.0;
    if (sp_c_rel[2] < 0.0) fact = -1.0;
    r.mPoint2 = cy.g.transformToParent(V3(sp_c_rel[0], sp_c_rel[1], fact * 0.5 * L));
    r.mPoint1 = cy.g.transformToParent(V3(sp_c_rel[0], sp_c_rel[1], sp_c_rel[2] - fact * sr));
    r.mDistance = fact * sp_c_rel[2] - 0.5 * L - sr;
  } else {
    V3 sp_c_proj(sp_c_rel[0], sp_c_rel[1], 0.0);
    double sp_c_proj_d = norm_2(sp_c_proj);
    double fact = 1.0;
    if (sp_c_rel[2] < 0.0) fact = -1.0;
    V3 rim_pt = (R / sp_c_proj_d) * sp_c_proj + V3(0.0, 0.0, fact * 0.5 * L);
    r.mPoint2 = cy.g.transformToParent(rim_pt);
    sp_c_proj = r.mPoint2 - sp_c;
    sp_c_proj_d = norm_2(sp_c_proj);
    r.mPoint1 = sp_c + (sr / sp_c_proj_d) * sp_c_proj;
    r.mDistance = sp_c_proj_d - sr;
  }
  return r;
}

// One entry of proxy_query_pair_3D::mProxFinders: which closed form, and (shape1, shape2) in the
// finder's own argument order (createProxFinderList: proxy_query_model.cpp:215-374).
struct ProxFinder {
  // 1 sphere-sphere, 2 sphere-ccyl, 3 sphere-box, 4 ccyl-ccyl, 5 ccyl-box,
  // 6 plane-plane, 7 plane-sphere, 8 plane-ccyl, 9 plane-cylinder, 10 plane-box, 11 sphere-cylinder,
  // 12 GJK (a convex vertex set against sphere / ccyl / box / vertex set: the build's query, reak_gjk.hpp)
  int routine;
  int s1, s2;   // indices into the combined shape table
};

// createProxFinderList: i-major / j-minor over (model1 shapes, model2 shapes), the reference's cascade of kinds (plane,
// then sphere, then capped cylinder, then cylinder, then box: proxy_query_model.cpp:225-370); pairs without an enabled
// routine (ccyl-cylinder :317-320, cylinder-cylinder :341-344, cylinder-box :346-349, box-box :366-369) produce no
// finder.
inline void createProxFinderList(const std::vector<rkh_shape>& shapes, const std::vector<int>& model1,
                                 const std::vector<int>& model2, std::vector<ProxFinder>& out) {
  out.clear();
  for (int i : model1)
    for (int j : model2) {
      const int ki = shapes[i].kind, kj = shapes[j].kind;
      if (ki == RKH_SHAPE_MESH || kj == RKH_SHAPE_MESH) {  // not a reference pair; shape1 = model 1's shape
        const int ko = (ki == RKH_SHAPE_MESH) ? kj : ki;
        if (ko == RKH_SHAPE_SPHERE || ko == RKH_SHAPE_CCYLINDER || ko == RKH_SHAPE_BOX || ko == RKH_SHAPE_MESH)
          out.push_back({12, i, j});
      } else if (ki == RKH_SHAPE_PLANE || kj == RKH_SHAPE_PLANE) {
        int pl = (ki == RKH_SHAPE_PLANE) ? i : j;
        int other = (ki == RKH_SHAPE_PLANE) ? j : i;
        int ko = shapes[other].kind;
        if (ko == RKH_SHAPE_PLANE) out.push_back({6, pl, other});
        else if (ko == RKH_SHAPE_SPHERE) out.push_back({7, pl, other});
        else if (ko == RKH_SHAPE_CCYLINDER) out.push_back({8, pl, other});
        else if (ko == RKH_SHAPE_CYLINDER) out.push_back({9, pl, other});
        else if (ko == RKH_SHAPE_BOX) out.push_back({10, pl, other});
      } else if (ki == RKH_SHAPE_SPHERE || kj == RKH_SHAPE_SPHERE) {
        int sp = (ki == RKH_SHAPE_SPHERE) ? i : j;
        int other = (ki == RKH_SHAPE_SPHERE) ? j : i;
        int ko = shapes[other].kind;
        if (ko == RKH_SHAPE_SPHERE) out.push_back({1, sp, other});
        else if (ko == RKH_SHAPE_CCYLINDER) out.push_back({2, sp, other});
        else if (ko == RKH_SHAPE_CYLINDER) out.push_back({11, sp, other});
        else if (ko == RKH_SHAPE_BOX) out.push_back({3, sp, other});
      } else if (ki == RKH_SHAPE_CCYLINDER || kj == RKH_SHAPE_CCYLINDER) {
        int cc = (ki == RKH_SHAPE_CCYLINDER) ? i : j;
        int other = (ki == RKH_SHAPE_CCYLINDER) ? j : i;
        int ko = shapes[other].kind;
        if (ko == RKH_SHAPE_CCYLINDER) out.push_back({4, cc, other});
        else if (ko == RKH_SHAPE_BOX) out.push_back({5, cc, other});
      }
      // cylinder-cylinder, cylinder-box, box-box: no finder
    }
}

inline GjkShape to_gjk(const ShapeG& s) {
  GjkShape r;
  r.kind = s.kind;
  r.g = s.g;
  for (int k = 0; k < 3; ++k) r.dims[k] = s.dims[k];
  if (s.kind == RKH_SHAPE_MESH) {
    r.verts = s.mesh_pool + 3 * std::size_t(s.dims[0]);
    r.nv = int(s.dims[1]);
  }
  return r;
}

inline ProxRecord computeProximity(const ProxFinder& f, const std::vector<ShapeG>& g) {
  if (f.routine == 12) {
    ProxRecord r;
    r.mDistance = gjk_distance(to_gjk(g[f.s1]), to_gjk(g[f.s2]));
    return r;
  }
  switch (f.routine) {
    case 1: return prox_sphere_sphere(g[f.s1], g[f.s2]);
    case 2: return prox_sphere_ccylinder(g[f.s1], g[f.s2]);
    case 3: return prox_sphere_box(g[f.s1], g[f.s2]);
    case 4: return prox_ccylinder_ccylinder(g[f.s1], g[f.s2]);
    case 5: return prox_ccylinder_box(g[f.s1], g[f.s2]);
    case 6: return prox_plane_plane(g[f.s1], g[f.s2]);
    case 7: return prox_plane_sphere(g[f.s1], g[f.s2]);
    case 8: return prox_plane_ccylinder(g[f.s1], g[f.s2]);
    case 9: return prox_plane_cylinder(g[f.s1], g[f.s2]);
    case 10: return prox_plane_box(g[f.s1], g[f.s2]);
    case 11: return prox_sphere_cylinder(g[f.s1], g[f.s2]);
  }
  return ProxRecord();
}

// proxy_query_pair_3D::findMinimumDistance: proxy_query_model.cpp:376-402.
// Returns the minimum distance (+inf if there is no finder); *n_computed counts computeProximity calls.
inline double findMinimumDistance(const std::vector<ProxFinder>& finders, const std::vector<ShapeG>& g,
                                  long* n_computed = nullptr) {
  if (finders.empty()) return std::numeric_limits<double>::infinity();
  double min_dist = computeProximity(finders[0], g).mDistance;
  long cnt = 1;
  for (std::size_t i = 1; i < finders.size(); ++i) {
    V3 p1 = g[finders[i].s1].g.transformToParent(V3(0, 0, 0));
    V3 p2 = g[finders[i].s2].g.transformToParent(V3(0, 0, 0));
    if (norm_2(p2 - p1) - g[finders[i].s1].getBoundingRadius() - g[finders[i].s2].getBoundingRadius() >
        min_dist)
      continue;
    double d = computeProximity(finders[i], g).mDistance;
    ++cnt;
    if (min_dist > d) min_dist = d;
  }
  if (n_computed) *n_computed += cnt;
  return min_dist;
}

// Collision environment of one manipulator: manip_dk_proxy_env_impl::is_free
// (ctrl/topologies/manip_free_workspace.hpp:79-99) with a single proxy_query_pair_3D
// (robot model = chain-anchored shapes, environment model = world shapes).
struct ProxyEnv {
  std::vector<rkh_shape> shapes;
  std::vector<double> mesh_pool;  // vertex pool of the RKH_SHAPE_MESH shapes
  std::vector<int> robot, env;
  std::vector<ProxFinder> finders;
  std::vector<ProxFinder2> finders2;  // planar scenes: proxy_query_pair_2D
  bool planar = false;
  long n_pair_tests = 0;

  ProxyEnv() {}
  ProxyEnv(const rkh_shape* s, int n, const double* verts = nullptr, int n_verts = 0)
      : shapes(s, s + n), mesh_pool(verts, verts + 3 * std::size_t(n_verts)) {
    for (int i = 0; i < n; ++i) (shapes[i].anchor >= 0 ? robot : env).push_back(i);
    for (int i = 0; i < n; ++i) planar = planar || (shapes[i].kind >= RKH_SHAPE_CIRCLE && shapes[i].kind <= RKH_SHAPE_CRECT);
    if (planar) createProxFinderList2D(shapes, robot, env, finders2);
    else createProxFinderList(shapes, robot, env, finders);
  }
  void resolve2(const KteChain& chain, std::vector<ShapeG2>& g) const {
    g.resize(shapes.size());
    for (std::size_t i = 0; i < shapes.size(); ++i) {
      g[i].kind = shapes[i].kind;
      for (int k = 0; k < 2; ++k) g[i].dims[k] = shapes[i].dims[k];
      const Pose2 local = to_pose2(shapes[i].pose);
      g[i].g = shapes[i].anchor >= 0 ? global_pose2(&chain.frames2[shapes[i].anchor], local) : local;
    }
  }
  // Resolve every shape's global pose from the chain frames (pose_3D::getGlobalPose, pose_3D.hpp:102-110)
  void resolve(const KteChain& chain, std::vector<ShapeG>& g) const {
    g.resize(shapes.size());
    for (std::size_t i = 0; i < shapes.size(); ++i) {
      g[i].kind = shapes[i].kind;
      g[i].mesh_pool = mesh_pool.data();
      for (int k = 0; k < 3; ++k) g[i].dims[k] = shapes[i].dims[k];
      Pose local = to_pose(shapes[i].pose);
      if (shapes[i].anchor >= 0) {
        Pose parent;
        parent.Position = chain.frames[shapes[i].anchor].Position;
        parent.Q = chain.frames[shapes[i].anchor].Q;
        g[i].g = global_pose(&parent, local);
      } else {
        g[i].g = local;
      }
    }
  }
  double min_distance(const KteChain& chain) {
    if (planar) {
      std::vector<ShapeG2> g2;
      resolve2(chain, g2);
      return findMinimumDistance2D(finders2, g2, &n_pair_tests);
    }
    std::vector<ShapeG> g;
    resolve(chain, g);
    return findMinimumDistance(finders, g, &n_pair_tests);
  }
  // is_free: "(tmp) && (mDistance < 0.0) -> false"
  bool is_free(const KteChain& chain) { return !(min_distance(chain) < 0.0); }
};

}  // namespace oracle
#endif
