// proximity_planar_device.h -- the reference's planar (2D) pose arithmetic and closed-form pair distances on the device.
//
// Same branch structure and fp64 operation order as
//   rot_mat_2D / pose_2D (core/kinetostatics/rotations_2D.hpp:264-302, pose_2D.hpp:98-106,177-193),
//   prox_circle_circle.cpp:40-57, prox_circle_crect.cpp:40-84, prox_circle_rectangle.cpp:40-86,
//   prox_crect_crect.cpp:40-131, prox_crect_rectangle.cpp:40-205, prox_rectangle_rectangle.cpp:40-168.
// Only the distance is produced (and, where the distance depends on them, the two points).
#pragma once
#include "device_math.h"

namespace rkh {

struct d2 {
  double x, y;
};
RKH_DI d2 mk2(double x, double y) { return d2{x, y}; }
RKH_DI d2 operator+(d2 a, d2 b) { return d2{a.x + b.x, a.y + b.y}; }
RKH_DI d2 operator-(d2 a, d2 b) { return d2{a.x - b.x, a.y - b.y}; }
RKH_DI d2 operator-(d2 a) { return d2{-a.x, -a.y}; }
RKH_DI d2 operator*(double s, d2 a) { return d2{a.x * s, a.y * s}; }
RKH_DI double dot(d2 a, d2 b) { return a.x * b.x + a.y * b.y; }
RKH_DI double norm_2(d2 v) { return sqrt((0.0 + v.x * v.x) + v.y * v.y); }
// rot_mat_2D (c, s): R1 * R2, R * V, V * R (= R^T V)
RKH_DI d2 rmul(d2 a, d2 b) { return d2{a.x * b.x - a.y * b.y, a.y * b.x + a.x * b.y}; }
RKH_DI d2 rrot(d2 R, d2 V) { return d2{V.x * R.x - V.y * R.y, V.x * R.y + V.y * R.x}; }
RKH_DI d2 rrotT(d2 V, d2 R) { return d2{V.x * R.x + V.y * R.y, V.y * R.x - V.x * R.y}; }

struct ShapeP {  // planar shape with resolved global pose
  int kind;
  d2 pos;
  d2 rot;  // (cos, sin)
  double d0, d1;
};
RKH_DI d2 to_parent(const ShapeP& s, d2 V) { return s.pos + rrot(s.rot, V); }
RKH_DI d2 from_parent(const ShapeP& s, d2 V) { return rrotT(V - s.pos, s.rot); }

enum PairRoutinePlanar : int {
  PR_CIRCLE_CIRCLE = 11, PR_CIRCLE_CRECT = 12, PR_CIRCLE_RECT = 13, PR_CRECT_CRECT = 14, PR_CRECT_RECT = 15, PR_RECT_RECT = 16
};

RKH_DI double dist_circle_circle(const ShapeP& s1, const ShapeP& s2) {
  const d2 c1 = to_parent(s1, mk2(0.0, 0.0)), c2 = to_parent(s2, mk2(0.0, 0.0));
  return norm_2(c2 - c1) - s1.d0 - s2.d0;
}

RKH_DI double dist_circle_crect(const ShapeP& ci, const ShapeP& cr) {
  const d2 rel = from_parent(cr, to_parent(ci, mk2(0.0, 0.0)));
  const double R = ci.d0;
  if ((rel.x > -0.5 * cr.d0) && (rel.x < 0.5 * cr.d0)) {
    if (rel.y > 0.0) return rel.y - R - 0.5 * cr.d1;
    return -0.5 * cr.d1 - rel.y - R;
  }
  d2 endc = mk2(0.0, 0.0);
  if (rel.x > 0.0) endc.x += 0.5 * cr.d0;
  else endc.x -= 0.5 * cr.d0;
  return norm_2(rel - endc) - 0.5 * cr.d1 - R;
}

// closest boundary point of a rectangle to a point, in the rectangle's frame (shared by prox_circle_rectangle.cpp:54-78
// and prox_rectangle_rectangle.cpp:46-74)
RKH_DI d2 rect_corner_point(const ShapeP& re, d2 rel) {
  bool in_x = (rel.x > -0.5 * re.d0) && (rel.x < 0.5 * re.d0);
  bool in_y = (rel.y > -0.5 * re.d1) && (rel.y < 0.5 * re.d1);
  if (in_x && in_y) {
    const double bx = 0.5 * re.d0 - fabs(rel.x), by = 0.5 * re.d1 - fabs(rel.y);
    if (bx <= by) in_x = false;
    else in_y = false;
  }
  d2 c = mk2(re.d0 * 0.5, re.d1 * 0.5);
  if (in_x) c.x = rel.x;
  else if (rel.x < 0.0) c.x = -c.x;
  if (in_y) c.y = rel.y;
  else if (rel.y < 0.0) c.y = -c.y;
  return c;
}

RKH_DI double dist_circle_rect(const ShapeP& ci, const ShapeP& re) {
  const d2 ci_c = to_parent(ci, mk2(0.0, 0.0));
  const d2 p2 = to_parent(re, rect_corner_point(re, from_parent(re, ci_c)));
  return norm_2(p2 - ci_c) - ci.d0;
}

RKH_DI double dist_crect_crect(const ShapeP& c1, const ShapeP& c2) {
  const d2 c2c = to_parent(c2, mk2(0.0, 0.0));
  const d2 c2t = rrot(c2.rot, mk2(1.0, 0.0));
  const d2 cr = from_parent(c1, c2c);
  const d2 tr = rrotT(c2t, c1.rot);
  const double L1 = c1.d0, W1 = c1.d1, L2 = c2.d0, W2 = c2.d1;
  if (fabs(tr.y) < 1e-5) {
    if ((cr.x + 0.5 * L2 > -0.5 * L1) || (cr.x - 0.5 * L2 < 0.5 * L1)) return fabs(cr.y) - 0.5 * W1 - 0.5 * W2;
    d2 a = mk2(0.0, 0.0), b = cr;
    if (cr.x < 0.0) {
      a.x -= 0.5 * L1;
      b.x += 0.5 * L2;
    } else {
      a.x += 0.5 * L1;
      b.x -= 0.5 * L2;
    }
    return norm_2(b - a) - 0.5 * W1 - 0.5 * W2;
  }
  const double d = dot(tr, cr);
  const double denom = 1.0 - tr.x * tr.x;
  double s_c = (tr.x * cr.x - d) / denom;
  double t_c = (cr.x - tr.x * d) / denom;
  if (s_c < -0.5 * L2) {
    s_c = -0.5 * L2;
    t_c = cr.x - 0.5 * L2 * tr.x;
  } else if (s_c > 0.5 * L2) {
    s_c = 0.5 * L2;
    t_c = cr.x + 0.5 * L2 * tr.x;
  }
  if (t_c < -0.5 * L1) {
    t_c = -0.5 * L1;
    s_c = -0.5 * L1 * tr.x - d;
  } else if (t_c > 0.5 * L1) {
    t_c = 0.5 * L1;
    s_c = 0.5 * L1 * tr.x - d;
  }
  if (s_c < -0.5 * L2) s_c = -0.5 * L2;
  else if (s_c > 0.5 * L2) s_c = 0.5 * L2;
  const d2 p1 = mk2(t_c, 0.0);
  const d2 p2 = cr + s_c * tr;
  return norm_2(p2 - p1) - 0.5 * W1 - 0.5 * W2;
}

// prox_crect_rectangle: the centre line against the rectangle (computeProximityOfLine), then the circle sweep
RKH_DI double dist_crect_rect(const ShapeP& cr, const ShapeP& re) {
  const d2 ln_c = to_parent(cr, mk2(0.0, 0.0));
  const d2 ln_t = rrot(cr.rot, mk2(1.0, 0.0));
  const double half_length = 0.5 * cr.d0;
  const d2 c = from_parent(re, ln_c);
  const d2 t = rrotT(ln_t, re.rot);
  const double DX = re.d0, DY = re.d1;
  double dist;
  if (fabs(t.x) < 1e-5) {
    if ((c.y + half_length > -0.5 * DY) || (c.y - half_length < 0.5 * DY)) {
      dist = fabs(c.x) - 0.5 * DX;
    } else {
      d2 re_pt = mk2(0.0, 0.0), ln_pt = c;
      if (c.x < 0.0) re_pt.x -= 0.5 * DX;
      else re_pt.x += 0.5 * DX;
      if (c.y < 0.0) {
        re_pt.y -= 0.5 * DY;
        ln_pt.y += half_length;
      } else {
        re_pt.y += 0.5 * DY;
        ln_pt.y -= half_length;
      }
      dist = norm_2(ln_pt - re_pt);
    }
  } else if (fabs(t.y) < 1e-5) {
    if ((c.x + half_length > -0.5 * DX) || (c.x - half_length < 0.5 * DX)) {
      dist = fabs(c.y) - 0.5 * DY;
    } else {
      d2 re_pt = mk2(0.0, 0.0), ln_pt = c;
      if (c.y < 0.0) re_pt.y -= 0.5 * DY;
      else re_pt.y += 0.5 * DY;
      if (c.x < 0.0) {
        re_pt.x -= 0.5 * DX;
        ln_pt.x += half_length;
      } else {
        re_pt.x += 0.5 * DX;
        ln_pt.x -= half_length;
      }
      dist = norm_2(ln_pt - re_pt);
    }
  } else {
    d2 n = mk2(-t.y * 1.0, t.x * 1.0);  // 1.0 % ln_t_rel
    if (dot(n, c) < 0.0) n = -n;
    d2 corner = mk2(-0.5 * DX, -0.5 * DY);
    if (n.x > 0.0) corner.x = 0.5 * DX;
    if (n.y > 0.0) corner.y = 0.5 * DY;
    const d2 diff = c - corner;
    dist = dot(diff, n);
    double t_tmp = -dot(diff, t);
    if (fabs(t_tmp) > half_length) {
      if (t_tmp < 0.0) t_tmp = -half_length;
      else t_tmp = half_length;
      const d2 ln_pt = c + t_tmp * t;
      const double in_x = fabs(ln_pt.x) - 0.5 * DX;
      const double in_y = fabs(ln_pt.y) - 0.5 * DY;
      if ((in_x < 0.0) && (in_y > in_x)) {
        dist = fabs(ln_pt.y) - 0.5 * DY;
      } else if ((in_y < 0.0) && (in_x > in_y)) {
        dist = fabs(ln_pt.x) - 0.5 * DX;
      } else {
        if (ln_pt.x < 0.0) corner.x = -0.5 * DX;
        else corner.x = 0.5 * DX;
        if (ln_pt.y < 0.0) corner.y = -0.5 * DY;
        else corner.y = 0.5 * DY;
        dist = norm_2(ln_pt - corner);
      }
    }
  }
  return dist - 0.5 * cr.d1;
}

RKH_DI double dist_rect_rect(const ShapeP& r1, const ShapeP& r2) {
  double best = INFINITY;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const ShapeP& own = pass == 0 ? r2 : r1;
    const ShapeP& other = pass == 0 ? r1 : r2;
    d2 corner = mk2(own.d0 * 0.5, own.d1 * 0.5);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (c == 1 || c == 3) corner.y = -corner.y;
      if (c == 2) corner.x = -corner.x;
      const d2 g = to_parent(own, corner);
      const d2 rec = to_parent(other, rect_corner_point(other, from_parent(other, g)));
      const double dd = norm_2(rec - g);
      if (dd < best) best = dd;
    }
  }
  return best;
}

RKH_DI double pair_distance_planar(int routine, const ShapeP& s1, const ShapeP& s2) {
  switch (routine) {
    case PR_CIRCLE_CIRCLE: return dist_circle_circle(s1, s2);
    case PR_CIRCLE_CRECT: return dist_circle_crect(s1, s2);
    case PR_CIRCLE_RECT: return dist_circle_rect(s1, s2);
    case PR_CRECT_CRECT: return dist_crect_crect(s1, s2);
    case PR_CRECT_RECT: return dist_crect_rect(s1, s2);
    case PR_RECT_RECT: return dist_rect_rect(s1, s2);
  }
  return INFINITY;
}

}  // namespace rkh
