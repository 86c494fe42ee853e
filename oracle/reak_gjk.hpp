// ORACLE -- TEST INFRASTRUCTURE ONLY (see reak_math.hpp header).
//
// Distance between convex shapes through support maps (Gilbert-Johnson-Keerthi), for pairs that involve a convex
// vertex set ("mesh").  NOT a restatement: the reference's proximity module has closed forms only and no mesh shape
// (TODO_list.txt:230).  BASELINE config C4 asks for convex-mesh obstacles, so the build defines the query (see
// reak_amd/csrc/gjk_device.h for the definition) and this file is its CPU twin, written independently in the oracle's
// own types.  What pins it: on sphere / capped-cylinder / box pairs it must reproduce the reference's closed forms
// (restated in reak_proximity.hpp) while the cores are apart, and a box given as the mesh of its eight corners must
// behave like the box (tests/test_oracle_kat.py); beyond that, "parity unpinned".
//
// Shape = convex core swept by a radius: sphere = point + r, capped cylinder = axis segment + r, box / mesh = themselves.
// distance = |closest point of (coreA - coreB) to the origin| - rA - rB; intersecting cores: -(rA + rB) - 1e-9.
#ifndef REAK_ORACLE_GJK_HPP
#define REAK_ORACLE_GJK_HPP

#include <cmath>
#include <limits>
#include <vector>

#include "../include/rkh_types.h"
#include "reak_math.hpp"

namespace oracle {

struct GjkShape {
  int kind = 0;
  Pose g;
  double dims[3] = {0, 0, 0};
  const double* verts = nullptr;  // mesh: local vertices
  int nv = 0;
  double radius() const { return kind == RKH_SHAPE_SPHERE ? dims[0] : (kind == RKH_SHAPE_CCYLINDER ? dims[1] : 0.0); }
  // support point of the core in world direction d
  V3 support(const V3& d) const {
    const RotMat R = g.Q.getRotMat();
    const V3 dl = d * R;  // R^T d
    V3 p(0.0, 0.0, 0.0);
    if (kind == RKH_SHAPE_CCYLINDER) {
      p = V3(0.0, 0.0, dl[2] >= 0.0 ? 0.5 * dims[0] : -0.5 * dims[0]);
    } else if (kind == RKH_SHAPE_BOX) {
      p = V3(dl[0] >= 0.0 ? 0.5 * dims[0] : -0.5 * dims[0], dl[1] >= 0.0 ? 0.5 * dims[1] : -0.5 * dims[1],
             dl[2] >= 0.0 ? 0.5 * dims[2] : -0.5 * dims[2]);
    } else if (kind == RKH_SHAPE_CYLINDER) {  // flat-ended cylinder (dims: length, radius; axis = local z): rim point
      const double rho = std::sqrt(dl[0] * dl[0] + dl[1] * dl[1]);
      p = V3(rho > 0.0 ? (dims[1] * dl[0]) / rho : 0.0, rho > 0.0 ? (dims[1] * dl[1]) / rho : 0.0,
             dl[2] >= 0.0 ? 0.5 * dims[0] : -0.5 * dims[0]);
    } else if (kind == RKH_SHAPE_MESH) {
      double best = -std::numeric_limits<double>::infinity();
      for (int i = 0; i < nv; ++i) {
        const V3 v(verts[3 * i], verts[3 * i + 1], verts[3 * i + 2]);
        const double t = dot(dl, v);
        if (t > best) {
          best = t;
          p = v;
        }
      }
    }
    return g.Position + R * p;
  }
};

namespace gjk_detail {
inline V3 O() { return V3(0.0, 0.0, 0.0); }
// closest point to the origin on a simplex of 1..3 points; reduces the simplex to the supporting face
inline V3 closest3(std::vector<V3>& W) {
  if (W.size() == 1) return W[0];
  if (W.size() == 2) {
    const V3 a = W[0], b = W[1], ab = b - a;
    const double t = dot(O() - a, ab), den = dot(ab, ab);
    if (t <= 0.0 || den <= 0.0) { W = {a}; return a; }
    if (t >= den) { W = {b}; return b; }
    return a + (t / den) * ab;
  }
  const V3 a = W[0], b = W[1], c = W[2];
  const V3 ab = b - a, ac = c - a, ap = O() - a;
  const double d1 = dot(ab, ap), d2 = dot(ac, ap);
  if (d1 <= 0.0 && d2 <= 0.0) { W = {a}; return a; }
  const V3 bp = O() - b;
  const double d3 = dot(ab, bp), d4 = dot(ac, bp);
  if (d3 >= 0.0 && d4 <= d3) { W = {b}; return b; }
  const double vc = d1 * d4 - d3 * d2;
  if (vc <= 0.0 && d1 >= 0.0 && d3 <= 0.0) {
    const double t = d1 / (d1 - d3);
    W = {a, b};
    return a + t * ab;
  }
  const V3 cp = O() - c;
  const double d5 = dot(ab, cp), d6 = dot(ac, cp);
  if (d6 >= 0.0 && d5 <= d6) { W = {c}; return c; }
  const double vb = d5 * d2 - d1 * d6;
  if (vb <= 0.0 && d2 >= 0.0 && d6 <= 0.0) {
    const double t = d2 / (d2 - d6);
    W = {a, c};
    return a + t * ac;
  }
  const double va = d3 * d6 - d5 * d4;
  if (va <= 0.0 && (d4 - d3) >= 0.0 && (d5 - d6) >= 0.0) {
    const double t = (d4 - d3) / ((d4 - d3) + (d5 - d6));
    W = {b, c};
    return b + t * (c - b);
  }
  const double denom = 1.0 / (va + vb + vc);
  const double vv = vb * denom, ww = vc * denom;
  return a + vv * ab + ww * ac;
}
// the same for up to 4 points; false = the origin is inside the tetrahedron
inline bool closest(std::vector<V3>& W, V3& v) {
  if (W.size() < 4) {
    v = closest3(W);
    return true;
  }
  const V3 P[4] = {W[0], W[1], W[2], W[3]};
  const int F[4][4] = {{0, 1, 2, 3}, {0, 2, 3, 1}, {0, 3, 1, 2}, {1, 3, 2, 0}};
  double best = std::numeric_limits<double>::infinity();
  std::vector<V3> bestW;
  V3 bestV;
  bool outside_any = false;
  for (int f = 0; f < 4; ++f) {
    const V3 a = P[F[f][0]], b = P[F[f][1]], c = P[F[f][2]], dd = P[F[f][3]];
    const V3 nrm = cross(b - a, c - a);
    const double so = dot(O() - a, nrm), sd = dot(dd - a, nrm);
    if (so * sd < 0.0 || sd == 0.0) {
      outside_any = true;
      std::vector<V3> T = {a, b, c};
      const V3 tv = closest3(T);
      const double q = dot(tv, tv);
      if (q < best) {
        best = q;
        bestV = tv;
        bestW = T;
      }
    }
  }
  if (!outside_any) return false;
  W = bestW;
  v = bestV;
  return true;
}
}  // namespace gjk_detail

inline double gjk_distance(const GjkShape& A, const GjkShape& B) {
  const double rsum = A.radius() + B.radius();
  V3 v = A.g.Position - B.g.Position;
  if (dot(v, v) == 0.0) v = V3(1.0, 0.0, 0.0);
  std::vector<V3> W;
  for (int it = 0; it < 64; ++it) {
    const V3 w = A.support(-v) - B.support(v);
    const double vv = dot(v, v), vw = dot(v, w);
    if (!W.empty() && (vv - vw) <= 1e-14 * vv) break;
    bool dup = false;
    for (const V3& p : W) dup = dup || (p[0] == w[0] && p[1] == w[1] && p[2] == w[2]);
    if (dup) break;
    W.push_back(w);
    if (!gjk_detail::closest(W, v)) return -rsum - 1e-9;
    if (dot(v, v) <= 1e-30) return -rsum - 1e-9;
  }
  return norm_2(v) - rsum;
}

}  // namespace oracle
#endif
