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
#include "../../include/rkh_types.h"
#include "device_math.h"

#include "gjk_device.h"

namespace rkh {

struct ShapeG {  // shape with resolved global pose
  int kind;
  d3 pos;
  d4 q;
  double d0, d1, d2;  // dims
};

enum PairRoutine : int {
  PR_NONE = 0,  // the reference has no finder for the pair (proxy_query_model.cpp:317-320,341-349,366-369)
  PR_SPHERE_SPHERE = 1, PR_SPHERE_CCYL = 2, PR_SPHERE_BOX = 3, PR_CCYL_CCYL = 4, PR_CCYL_BOX = 5,
  PR_PLANE_PLANE = 6, PR_PLANE_SPHERE = 7, PR_PLANE_CCYL = 8, PR_PLANE_CYL = 9, PR_PLANE_BOX = 10, PR_SPHERE_CYL = 11,
  PR_GJK = 12  // a convex vertex set against a sphere / capped cylinder / box / vertex set (gjk_device.h); shape1 = model 1's
};

// createProxFinderList's cascade of kinds (proxy_query_model.cpp:225-370) for one pair: the routine, and whether the
// FIRST shape is the routine's shape1 (plane before sphere before capped cylinder; equal kinds: model 1's shape).
RKH_DI int pair_routine(int ka, int kb, bool* a_is_shape1) {
  auto other = [&](int first_kind) { return (ka == first_kind) ? kb : ka; };
  if (ka == RKH_SHAPE_MESH || kb == RKH_SHAPE_MESH) {  // not a reference pair: the build's GJK query
    *a_is_shape1 = true;
    const int ko = other(RKH_SHAPE_MESH);
    return (ko == RKH_SHAPE_SPHERE || ko == RKH_SHAPE_CCYLINDER || ko == RKH_SHAPE_BOX || ko == RKH_SHAPE_MESH) ? PR_GJK : PR_NONE;
  }
  if (ka == RKH_SHAPE_PLANE || kb == RKH_SHAPE_PLANE) {
    *a_is_shape1 = (ka == RKH_SHAPE_PLANE);
    switch (other(RKH_SHAPE_PLANE)) {
      case RKH_SHAPE_PLANE: return PR_PLANE_PLANE;
      case RKH_SHAPE_SPHERE: return PR_PLANE_SPHERE;
      case RKH_SHAPE_CCYLINDER: return PR_PLANE_CCYL;
      case RKH_SHAPE_CYLINDER: return PR_PLANE_CYL;
      case RKH_SHAPE_BOX: return PR_PLANE_BOX;
    }
    return PR_NONE;
  }
  if (ka == RKH_SHAPE_SPHERE || kb == RKH_SHAPE_SPHERE) {
    *a_is_shape1 = (ka == RKH_SHAPE_SPHERE);
    switch (other(RKH_SHAPE_SPHERE)) {
      case RKH_SHAPE_SPHERE: return PR_SPHERE_SPHERE;
      case RKH_SHAPE_CCYLINDER: return PR_SPHERE_CCYL;
      case RKH_SHAPE_CYLINDER: return PR_SPHERE_CYL;
      case RKH_SHAPE_BOX: return PR_SPHERE_BOX;
    }
    return PR_NONE;
  }
  if (ka == RKH_SHAPE_CCYLINDER || kb == RKH_SHAPE_CCYLINDER) {
    *a_is_shape1 = (ka == RKH_SHAPE_CCYLINDER);
    switch (other(RKH_SHAPE_CCYLINDER)) {
      case RKH_SHAPE_CCYLINDER: return PR_CCYL_CCYL;
      case RKH_SHAPE_BOX: return PR_CCYL_BOX;
    }
    return PR_NONE;
  }
  *a_is_shape1 = true;
  return PR_NONE;  // cylinder-cylinder, cylinder-box, box-box
}

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

// ---- plane / cylinder finders (prox_plane_{sphere,box,ccylinder,cylinder,plane}.cpp, prox_sphere_cylinder.cpp) -------
// The enabled prox_plane_* routines treat the plane as infinite (normal = local z); only prox_plane_plane looks at its
// extents.  pose_3D::rotateToGlobal = Quat * V, rotateFromGlobal = invert(Quat) * V (pose_3D.hpp:154-170).

RKH_DI double dist_plane_sphere(const ShapeG& pl, const ShapeG& sp) {  // prox_plane_sphere.cpp:106-122
  const d3 sp_c = pose_to_parent(sp.pos, sp.q, mk3(0, 0, 0));
  const d3 rel = pose_from_parent(pl.pos, pl.q, sp_c);
  return rel.z - sp.d0;
}

// prox_plane_box.cpp:43-71: bx_x, bx_y and bx_z are ALL built from the box's local x axis in the reference (:53-55)
RKH_DI double dist_plane_box(const ShapeG& pl, const ShapeG& bx) {
  const d3 bx_c = pose_to_parent(bx.pos, bx.q, mk3(0, 0, 0));
  d3 bx_x = qrot(qinv(pl.q), qrot(bx.q, mk3(1.0, 0.0, 0.0)));
  if (bx_x.z > 0.0) bx_x = -bx_x;
  const d3 bx_y = bx_x, bx_z = bx_x;
  const d3 c_rel = pose_from_parent(pl.pos, pl.q, bx_c);
  const d3 pt = c_rel + 0.5 * (bx.d0 * bx_x + bx.d1 * bx_y + bx.d2 * bx_z);
  return pt.z;
}

RKH_DI double dist_plane_ccyl(const ShapeG& pl, const ShapeG& cc) {  // prox_plane_ccylinder.cpp:43-74
  const d3 cy_c = pose_to_parent(cc.pos, cc.q, mk3(0, 0, 0));
  const d3 cy_t = qrot(cc.q, mk3(0.0, 0.0, 1.0));
  const d3 c_rel = pose_from_parent(pl.pos, pl.q, cy_c);
  d3 t_rel = qrot(qinv(pl.q), cy_t);
  if (fabs(t_rel.z) < 1e-6) return c_rel.z - cc.d1;
  if (t_rel.z > 0.0) t_rel = -t_rel;
  const d3 pt = c_rel + (0.5 * cc.d0) * t_rel + mk3(0.0, 0.0, -cc.d1);
  return pt.z;
}

RKH_DI double dist_plane_cyl(const ShapeG& pl, const ShapeG& cy) {  // prox_plane_cylinder.cpp:42-78
  const d3 cy_c = pose_to_parent(cy.pos, cy.q, mk3(0, 0, 0));
  const d3 cy_t = qrot(cy.q, mk3(0.0, 0.0, 1.0));
  const d3 c_rel = pose_from_parent(pl.pos, pl.q, cy_c);
  d3 t_rel = qrot(qinv(pl.q), cy_t);
  if (fabs(t_rel.z) < 1e-6) return c_rel.z - cy.d1;
  if (sqrt(t_rel.x * t_rel.x + t_rel.y * t_rel.y) < 1e-6) return c_rel.z - 0.5 * cy.d0;
  if (t_rel.z > 0.0) t_rel = -t_rel;
  const d3 v = mk3(0.0, 0.0, -1.0) + t_rel.z * t_rel;
  const double n = norm_2(v);
  const d3 r_rel = mk3(v.x / n, v.y / n, v.z / n);  // unit(): vect_alg.hpp:2378-2382
  const d3 pt = c_rel + (0.5 * cy.d0) * t_rel + cy.d1 * r_rel;
  return pt.z;
}

// prox_plane_plane::computeProximityOfPoint (prox_plane_plane.cpp:43-95): here the plane is finite
RKH_DI double plane_point_distance(const ShapeG& pl, d3 pt) {
  const d3 p = pose_from_parent(pl.pos, pl.q, pt);
  const double hx = 0.5 * pl.d0, hy = 0.5 * pl.d1;
  const bool in_x = (p.x > -hx) && (p.x < hx), in_y = (p.y > -hy) && (p.y < hy);
  if (in_x && in_y) {
    double fact = 1.0;
    if (p.z < 0.0) fact = -1.0;
    return fact * p.z;
  }
  d3 rim;
  if (in_x) {
    double fact = 1.0;
    if (p.y < 0.0) fact = -1.0;
    rim = mk3(p.x, fact * 0.5 * pl.d1, 0.0);
  } else if (in_y) {
    double fact = 1.0;
    if (p.x < 0.0) fact = -1.0;
    rim = mk3(fact * 0.5 * pl.d0, p.y, 0.0);
  } else {
    rim = mk3(0.5 * pl.d0, 0.5 * pl.d1, 0.0);
    if (p.x < 0.0) rim.x = -rim.x;
    if (p.y < 0.0) rim.y = -rim.y;
  }
  return norm_2(pose_to_parent(pl.pos, pl.q, rim) - pt);
}

// prox_plane_plane.cpp:98-183: the corners of plane 2 against plane 1, then those of plane 1 against plane 2
RKH_DI double dist_plane_plane(const ShapeG& p1, const ShapeG& p2) {
  double best = INFINITY;
#pragma unroll 1
  for (int side = 0; side < 2; ++side) {
    const ShapeG& of = side == 0 ? p2 : p1;
    const ShapeG& against = side == 0 ? p1 : p2;
    d3 corner = mk3(0.5 * of.d0, 0.5 * of.d1, 0.0);
#pragma unroll 1
    for (int k = 0; k < 4; ++k) {
      if (k == 1 || k == 3) corner.y = -corner.y;
      if (k == 2) corner.x = -corner.x;
      const double d = plane_point_distance(against, pose_to_parent(of.pos, of.q, corner));
      if (d < best) best = d;
    }
  }
  return best;
}

RKH_DI double dist_sphere_cyl(const ShapeG& sp, const ShapeG& cy) {  // prox_sphere_cylinder.cpp:43-92
  const double L = cy.d0, R = cy.d1, sr = sp.d0;
  const d3 sp_c = pose_to_parent(sp.pos, sp.q, mk3(0, 0, 0));
  const d3 rel = pose_from_parent(cy.pos, cy.q, sp_c);
  const double rel_rad = sqrt(rel.x * rel.x + rel.y * rel.y);
  if (fabs(rel.z) <= 0.5 * L) return norm_2(mk3(rel.x, rel.y, 0.0)) - sr - R;
  double fact = 1.0;
  if (rel.z < 0.0) fact = -1.0;
  if (rel_rad < R) return fact * rel.z - 0.5 * L - sr;
  const d3 proj = mk3(rel.x, rel.y, 0.0);
  const double proj_d = norm_2(proj);
  const d3 rim = (R / proj_d) * proj + mk3(0.0, 0.0, fact * 0.5 * L);
  const d3 p2 = pose_to_parent(cy.pos, cy.q, rim);
  return norm_2(p2 - sp_c) - sr;
}

RKH_DI GjkShape to_gjk(const ShapeG& s, const double* mesh_pool) {
  GjkShape g;
  g.kind = s.kind;
  g.pos = s.pos;
  g.R = rotmat(s.q);
  g.d0 = s.d0; g.d1 = s.d1; g.d2 = s.d2;
  g.verts = nullptr;
  g.nv = 0;
  if (s.kind == RKH_SHAPE_MESH) {
    g.verts = mesh_pool + 3 * int(s.d0);
    g.nv = int(s.d1);
  }
  return g;
}

// (shape1, shape2) are already in the routine's own argument order; mesh_pool: the scene's vertex pool (GJK pairs only)
// GJK = false: kernels that never see mesh scenes (scene_fits_lane_kernel) leave the support-map query out -- its
// run-time-indexed simplex arrays are the only thing in this header that needs a private (scratch) segment
template <bool GJK = true>
RKH_DI double pair_distance(int routine, const ShapeG& s1, const ShapeG& s2, const double* mesh_pool = nullptr) {
  if (GJK) {
    if (routine == PR_GJK) return gjk_distance(to_gjk(s1, mesh_pool), to_gjk(s2, mesh_pool));
  }
  switch (routine) {
    case PR_SPHERE_SPHERE: return dist_sphere_sphere(s1, s2);
    case PR_SPHERE_CCYL: return dist_sphere_ccyl(s1, s2);
    case PR_SPHERE_BOX: return dist_sphere_box(s1, s2);
    case PR_CCYL_CCYL: return dist_ccyl_ccyl(s1, s2);
    case PR_CCYL_BOX: return dist_ccyl_box(s1, s2);
    case PR_PLANE_PLANE: return dist_plane_plane(s1, s2);
    case PR_PLANE_SPHERE: return dist_plane_sphere(s1, s2);
    case PR_PLANE_CCYL: return dist_plane_ccyl(s1, s2);
    case PR_PLANE_CYL: return dist_plane_cyl(s1, s2);
    case PR_PLANE_BOX: return dist_plane_box(s1, s2);
    case PR_SPHERE_CYL: return dist_sphere_cyl(s1, s2);
  }
  return INFINITY;
}

}  // namespace rkh
