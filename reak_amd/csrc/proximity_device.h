// proximity_device.h -- closed-form pair distance routines of ReaK's geometry/proximity on the device.
//
// Same branch structure and fp64 operation order as the reference routines so the sign of the
// distance (the collision verdict of manip_dk_proxy_env_impl::is_free,
// ctrl/topologies/manip_free_workspace.hpp:85-95) matches the CPU planner:
//   prox_sphere_sphere.cpp:41-58, prox_sphere_ccylinder.cpp:43-80, prox_sphere_box.cpp:45-68,
//   prox_ccylinder_ccylinder.cpp:43-128 (incl. the always-true OR of the parallel branch, :61-62),
//   prox_ccylinder_box.cpp:45-70 + findProximityBoxToLine/Point (prox_fundamentals_3D.cpp:35-115)
//   with golden_section_search_impl (core/optimization/line_search.hpp:71-95).
// Only the distance is produced (mPoint1/mPoint2 are not needed for the verdict).
#pragma once
#include "device_math.h"

namespace rkh {

struct ShapeG {  // shape with resolved global pose
  int kind;
  d3 pos;
  d4 q;
  double d0, d1, d2;  // dims
};

enum PairRoutine : int { PR_SPHERE_SPHERE = 1, PR_SPHERE_CCYL = 2, PR_SPHERE_BOX = 3, PR_CCYL_CCYL = 4, PR_CCYL_BOX = 5 };

// findProximityBoxToPoint (prox_fundamentals_3D.cpp:35-82): signed distance only
RKH_DI double box_point_distance(const ShapeG& bx, d3 pt) {
  const d3 p = pose_from_parent(bx.pos, bx.q, pt);
  const double hx = 0.5 * bx.d0, hy = 0.5 * bx.d1, hz = 0.5 * bx.d2;
  bool in_x = (p.x > -hx) && (p.x < hx);
  bool in_y = (p.y > -hy) && (p.y < hy);
  bool in_z = (p.z > -hz) && (p.z < hz);
  const bool inside = in_x && in_y && in_z;
  if (inside) {
    const double bx_ = hx - fabs(p.x), by_ = hy - fabs(p.y), bz_ = hz - fabs(p.z);
    if ((bx_ <= by_) && (bx_ <= bz_)) in_x = false;
    else if ((by_ <= bx_) && (by_ <= bz_)) in_y = false;
    else in_z = false;
  }
  d3 c = mk3(hx, hy, hz);
  if (in_x) c.x = p.x;
  else if (p.x < 0.0) c.x = -c.x;
  if (in_y) c.y = p.y;
  else if (p.y < 0.0) c.y = -c.y;
  if (in_z) c.z = p.z;
  else if (p.z < 0.0) c.z = -c.z;
  const double diff_d = norm_2(c - p);
  return inside ? -diff_d : diff_d;
}

// findProximityBoxToLine (prox_fundamentals_3D.cpp:108-115): the distance the functor saw last,
// i.e. f((low+up)/2) of golden_section_search_impl (line_search.hpp:74-95).
#define RKH_GOLDEN_MAX_ITER 256
RKH_DI double box_line_distance(const ShapeG& bx, d3 center, d3 tangent, double half_len) {
  const double phi = 1.618033988;
  const double tol = 1e-3 * half_len;
  double low = -half_len, up = half_len;
  double mid = low + (up - low) / phi;
  double mid_cost = box_point_distance(bx, center + tangent * mid);
  for (int it = 0;; ++it) {
    if (fabs(low - up) < tol || it >= RKH_GOLDEN_MAX_ITER) return box_point_distance(bx, center + tangent * ((low + up) * 0.5));
    const double test = mid + (up - mid) / phi;
    const double test_cost = box_point_distance(bx, center + tangent * test);
    if (test_cost < mid_cost) {
      low = mid;
      mid = test;
      mid_cost = test_cost;
    } else {
      up = low;
      low = test;
    }
  }
}

RKH_DI double dist_sphere_sphere(const ShapeG& s1, const ShapeG& s2) {
  const d3 c1 = pose_to_parent(s1.pos, s1.q, mk3(0, 0, 0));
  const d3 c2 = pose_to_parent(s2.pos, s2.q, mk3(0, 0, 0));
  return norm_2(c2 - c1) - s1.d0 - s2.d0;
}

RKH_DI double dist_sphere_box(const ShapeG& sp, const ShapeG& bx) {
  const d3 c = pose_to_parent(sp.pos, sp.q, mk3(0, 0, 0));
  return box_point_distance(bx, c) - sp.d0;
}

RKH_DI double dist_sphere_ccyl(const ShapeG& sp, const ShapeG& cc) {
  const double len = cc.d0, rad = cc.d1, sr = sp.d0;
  const d3 sp_c = pose_to_parent(sp.pos, sp.q, mk3(0, 0, 0));
  const d3 rel = pose_from_parent(cc.pos, cc.q, sp_c);
  if (fabs(rel.z) <= 0.5 * len) {
    const double proj_d = norm_2(mk3(rel.x, rel.y, 0.0));
    return proj_d - sr - rad;
  }
  double fact = 1.0;
  if (rel.z < 0.0) fact = -1.0;
  const d3 cy_c2 = pose_to_parent(cc.pos, cc.q, mk3(0.0, 0.0, fact * 0.5 * len));
  return norm_2(cy_c2 - sp_c) - sr - rad;
}

RKH_DI double dist_ccyl_ccyl(const ShapeG& c1, const ShapeG& c2) {
  const double L1 = c1.d0, R1 = c1.d1, L2 = c2.d0, R2 = c2.d1;
  const d3 cy2_c = pose_to_parent(c2.pos, c2.q, mk3(0, 0, 0));
  const d3 cy2_t = qrot(c2.q, mk3(0.0, 0.0, 1.0));
  const d3 cr = pose_from_parent(c1.pos, c1.q, cy2_c);
  const d3 tr = qrot(qinv(c1.q), cy2_t);
  if (sqrt(tr.x * tr.x + tr.y * tr.y) < 1e-5) {
    if ((cr.z + 0.5 * L2 > -0.5 * L1) || (cr.z - 0.5 * L2 < 0.5 * L1)) {
      return sqrt(cr.x * cr.x + cr.y * cr.y) - R1 - R2;
    }
    d3 s1 = mk3(0.0, 0.0, 0.0), s2 = cr;
    if (cr.z < 0.0) {
      s1.z -= 0.5 * L1;
      s2.z += 0.5 * L2;
    } else {
      s1.z += 0.5 * L1;
      s2.z -= 0.5 * L2;
    }
    return norm_2(s2 - s1) - R1 - R2;
  }
  const double d = dot(tr, cr);
  const double denom = 1.0 - tr.z * tr.z;
  double s_c = (tr.z * cr.z - d) / denom;
  double t_c = (cr.z - tr.z * d) / denom;
  if (s_c < -0.5 * L2) {
    s_c = -0.5 * L2;
    t_c = cr.z - 0.5 * L2 * tr.z;
  } else if (s_c > 0.5 * L2) {
    s_c = 0.5 * L2;
    t_c = cr.z + 0.5 * L2 * tr.z;
  }
  if (t_c < -0.5 * L1) {
    t_c = -0.5 * L1;
    s_c = -0.5 * L1 * tr.z - d;
  } else if (t_c > 0.5 * L1) {
    t_c = 0.5 * L1;
    s_c = 0.5 * L1 * tr.z - d;
  }
  if (s_c < -0.5 * L2) s_c = -0.5 * L2;
  else if (s_c > 0.5 * L2) s_c = 0.5 * L2;
  const d3 p1 = mk3(0.0, 0.0, t_c);
  const d3 p2 = cr + s_c * tr;
  return norm_2(p2 - p1) - R1 - R2;
}

RKH_DI double dist_ccyl_box(const ShapeG& cc, const ShapeG& bx) {
  const d3 cy_c = pose_to_parent(cc.pos, cc.q, mk3(0, 0, 0));
  const d3 cy_t = qrot(cc.q, mk3(0.0, 0.0, 1.0));
  return box_line_distance(bx, cy_c, cy_t, 0.5 * cc.d0) - cc.d1;
}

// (shape1, shape2) are already in the routine's own argument order
RKH_DI double pair_distance(int routine, const ShapeG& s1, const ShapeG& s2) {
  switch (routine) {
    case PR_SPHERE_SPHERE: return dist_sphere_sphere(s1, s2);
    case PR_SPHERE_CCYL: return dist_sphere_ccyl(s1, s2);
    case PR_SPHERE_BOX: return dist_sphere_box(s1, s2);
    case PR_CCYL_CCYL: return dist_ccyl_ccyl(s1, s2);
    case PR_CCYL_BOX: return dist_ccyl_box(s1, s2);
  }
  return INFINITY;
}

}  // namespace rkh
