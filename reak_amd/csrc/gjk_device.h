// gjk_device.h -- GJK distance between convex shapes given by support maps (fp64), for the pairs the reference has no
// closed form for: everything that involves a CONVEX VERTEX SET ("mesh", RKH_SHAPE_MESH).
//
// The reference's proximity module has closed-form pairs only (no mesh shape, no GJK: TODO_list.txt:230); BASELINE
// config C4 asks for convex-mesh obstacles through "batched GJK / support-mapping distance queries".  This is the
// build's definition of that query, the same in the oracle (oracle/reak_gjk.hpp) and on the device:
//   * a shape = a convex CORE swept by a radius: sphere = point + r, capped cylinder = axis segment + r, box, mesh and
//     the flat-ended cylinder = themselves (r = 0; the cylinder's support map is curved, so its pairs end on the
//     relative tolerance or the iteration cap instead of a repeated vertex).  The radii are handled analytically (distance = core distance - rA - rB), so pairs of
//     spheres / capped cylinders / boxes reproduce the reference's closed forms to rounding while the cores are apart;
//   * core distance = Gilbert-Johnson-Keerthi on the Minkowski difference with polytope supports (finite termination),
//     closest point of the simplex by Voronoi-region tests (Ericson, Real-Time Collision Detection 5.1);
//   * intersecting cores: the penetration depth is not computed; the result is -(rA + rB) - kGjkOverlap (negative =
//     collision, which is all manip_dk_proxy_env_impl::is_free looks at, manip_free_workspace.hpp:85-95).
// Closed forms stay the default for primitive pairs (they ARE the reference); GJK runs for mesh pairs, and for any
// pair through the diagnostic entry point that checks it against the closed forms.
#pragma once
#include "../../include/rkh_types.h"
#include "device_math.h"

namespace rkh {

constexpr double kGjkOverlap = 1e-9;
constexpr int kGjkMaxIter = 64;

struct GjkShape {
  int kind;
  d3 pos;
  m33 R;  // rotmat(q): local -> world
  double d0, d1, d2;
  const double* verts;  // mesh: nv local vertices (x, y, z), else unused
  int nv;
};

RKH_DI double gjk_radius(const GjkShape& s) {
  return s.kind == RKH_SHAPE_SPHERE ? s.d0 : (s.kind == RKH_SHAPE_CCYLINDER ? s.d1 : 0.0);
}

// support point of the shape's CORE in world direction dir
RKH_DI d3 gjk_support(const GjkShape& s, d3 dir) {
  const d3 dl = mulT(dir, s.R);  // R^T dir
  d3 p;
  if (s.kind == RKH_SHAPE_SPHERE) {
    p = mk3(0.0, 0.0, 0.0);
  } else if (s.kind == RKH_SHAPE_CCYLINDER) {
    p = mk3(0.0, 0.0, dl.z >= 0.0 ? 0.5 * s.d0 : -0.5 * s.d0);
  } else if (s.kind == RKH_SHAPE_BOX) {
    p = mk3(dl.x >= 0.0 ? 0.5 * s.d0 : -0.5 * s.d0, dl.y >= 0.0 ? 0.5 * s.d1 : -0.5 * s.d1,
            dl.z >= 0.0 ? 0.5 * s.d2 : -0.5 * s.d2);
  } else if (s.kind == RKH_SHAPE_CYLINDER) {  // flat-ended cylinder (d0 = length, d1 = radius; axis = local z): rim point
    const double rho = sqrt(dl.x * dl.x + dl.y * dl.y);
    p = mk3(rho > 0.0 ? (s.d1 * dl.x) / rho : 0.0, rho > 0.0 ? (s.d1 * dl.y) / rho : 0.0,
            dl.z >= 0.0 ? 0.5 * s.d0 : -0.5 * s.d0);
  } else {  // mesh: first maximum wins
    double best = -INFINITY;
    p = mk3(0.0, 0.0, 0.0);
#pragma unroll 1
    for (int i = 0; i < s.nv; ++i) {
      const d3 v = mk3(s.verts[3 * i], s.verts[3 * i + 1], s.verts[3 * i + 2]);
      const double t = dot(dl, v);
      if (t > best) {
        best = t;
        p = v;
      }
    }
  }
  return s.pos + mul(s.R, p);
}

// Closest point to the origin on the simplex W[0..n), n = 1..3; the simplex is reduced to the face that carries it.
RKH_DI void gjk_closest3(d3* W, int& n, d3& v) {
  if (n == 1) {
    v = W[0];
    return;
  }
  if (n == 2) {
    const d3 a = W[0], b = W[1], ab = b - a;
    const double t = dot(mk3(0, 0, 0) - a, ab), den = dot(ab, ab);
    if (t <= 0.0 || den <= 0.0) { n = 1; W[0] = a; v = a; return; }
    if (t >= den) { n = 1; W[0] = b; v = b; return; }
    v = a + (t / den) * ab;
    return;
  }
  if (n == 3) {  // Ericson 5.1.5, query point = origin
    const d3 a = W[0], b = W[1], c = W[2];
    const d3 ab = b - a, ac = c - a, ap = mk3(0, 0, 0) - a;
    const double d1 = dot(ab, ap), d2 = dot(ac, ap);
    if (d1 <= 0.0 && d2 <= 0.0) { n = 1; W[0] = a; v = a; return; }
    const d3 bp = mk3(0, 0, 0) - b;
    const double d3_ = dot(ab, bp), d4 = dot(ac, bp);
    if (d3_ >= 0.0 && d4 <= d3_) { n = 1; W[0] = b; v = b; return; }
    const double vc = d1 * d4 - d3_ * d2;
    if (vc <= 0.0 && d1 >= 0.0 && d3_ <= 0.0) {
      const double t = d1 / (d1 - d3_);
      n = 2; W[0] = a; W[1] = b; v = a + t * ab;
      return;
    }
    const d3 cp = mk3(0, 0, 0) - c;
    const double d5 = dot(ab, cp), d6 = dot(ac, cp);
    if (d6 >= 0.0 && d5 <= d6) { n = 1; W[0] = c; v = c; return; }
    const double vb = d5 * d2 - d1 * d6;
    if (vb <= 0.0 && d2 >= 0.0 && d6 <= 0.0) {
      const double t = d2 / (d2 - d6);
      n = 2; W[0] = a; W[1] = c; v = a + t * ac;
      return;
    }
    const double va = d3_ * d6 - d5 * d4;
    if (va <= 0.0 && (d4 - d3_) >= 0.0 && (d5 - d6) >= 0.0) {
      const double t = (d4 - d3_) / ((d4 - d3_) + (d5 - d6));
      n = 2; W[0] = b; W[1] = c; v = b + t * (c - b);
      return;
    }
    const double denom = 1.0 / (va + vb + vc);
    const double vv = vb * denom, ww = vc * denom;
    v = a + vv * ab + ww * ac;
    return;
  }
}

// The same for n = 1..4.  Returns false when the origin is inside the tetrahedron (the sets intersect).
RKH_DI bool gjk_closest(d3* W, int& n, d3& v) {
  if (n < 4) {
    gjk_closest3(W, n, v);
    return true;
  }
  // n == 4: test the four faces the origin can be outside of (Ericson 5.1.6); keep the closest
  const d3 P[4] = {W[0], W[1], W[2], W[3]};
  const int F[4][4] = {{0, 1, 2, 3}, {0, 2, 3, 1}, {0, 3, 1, 2}, {1, 3, 2, 0}};  // face (i,j,k), opposite vertex l
  double best = INFINITY;
  d3 bestW[3];
  int bestN = 0;
  d3 bestV = mk3(0, 0, 0);
  bool outside_any = false;
#pragma unroll 1
  for (int f = 0; f < 4; ++f) {
    const d3 a = P[F[f][0]], b = P[F[f][1]], c = P[F[f][2]], dd = P[F[f][3]];
    const d3 nrm = cross(b - a, c - a);
    const double so = dot(mk3(0, 0, 0) - a, nrm), sd = dot(dd - a, nrm);
    // the origin is outside this face if it lies on the other side than the opposite vertex (degenerate: count as outside)
    if (so * sd < 0.0 || sd == 0.0) {
      outside_any = true;
      d3 T[3] = {a, b, c};
      int tn = 3;
      d3 tv;
      gjk_closest3(T, tn, tv);
      const double q = dot(tv, tv);
      if (q < best) {
        best = q;
        bestN = tn;
        bestV = tv;
        bestW[0] = T[0]; bestW[1] = T[1]; bestW[2] = T[2];
      }
    }
  }
  if (!outside_any) return false;  // inside all four faces
  n = bestN;
  for (int i = 0; i < bestN; ++i) W[i] = bestW[i];
  v = bestV;
  return true;
}

// distance between the two swept shapes (see the header comment)
RKH_DI double gjk_distance(const GjkShape& A, const GjkShape& B) {
  const double rsum = gjk_radius(A) + gjk_radius(B);
  d3 v = A.pos - B.pos;
  if (dot(v, v) == 0.0) v = mk3(1.0, 0.0, 0.0);
  d3 W[4];
  int n = 0;
#pragma unroll 1
  for (int it = 0; it < kGjkMaxIter; ++it) {
    const d3 w = gjk_support(A, -v) - gjk_support(B, v);
    const double vv = dot(v, v), vw = dot(v, w);
    // no support point closer to the origin than v along v: v is the closest point of A - B (first pass: n == 0)
    if (n > 0 && (vv - vw) <= 1e-14 * vv) break;
    bool dup = false;
    for (int i = 0; i < n; ++i) dup = dup || (W[i].x == w.x && W[i].y == w.y && W[i].z == w.z);
    if (dup) break;
    W[n++] = w;
    if (!gjk_closest(W, n, v)) return -rsum - kGjkOverlap;
    if (dot(v, v) <= 1e-30) return -rsum - kGjkOverlap;  // the origin lies on the simplex
  }
  return norm_2(v) - rsum;
}

}  // namespace rkh
