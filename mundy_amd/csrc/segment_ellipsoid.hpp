// segment_ellipsoid.hpp -- the rod - ellipsoid class (R-E) of the mixed-shape narrow phase.  BUILD EXTENSION, parity
// unpinned: the reference has no segment - ellipsoid distance (mundy_geom/distance/LineSegmentEllipsoid.hpp:21-33 is an
// empty stub), and its point - ellipsoid distance (PointEllipsoid.hpp:94-135) is a nine-start L-BFGS over the surface
// normal good to 1e-4 (UnitTestEllipsoidEllipsoid.cpp:52-53).  Rounds 1-2 reused that minimiser for this class --
// 1 280 objective evaluations per pair, 69 % of the narrow phase of BASELINE configs[4] -- although nothing pins R-E to
// it.  Round 3 computes the same quantity in closed form:
//   * the SIGNED distance s(y) of a point to an ellipsoid and its closest surface point, exactly: in the body frame the
//     closest point is x_i = e_i^2 y_i / (tau + e_i^2) with tau the root of sum_i (e_i y_i / (tau + e_i^2))^2 = 1 in
//     tau > -e_min^2 (Lagrange condition, solved for u = tau / e_min^2 + 1 > 0; Eberly, "Distance from a point to an ellipse, an ellipsoid, or a
//     hyperellipsoid": first octant, axes sorted descending, the points on the symmetry planes treated separately).
//     The root is found by Newton on the secular-equation form 1 - 1 / sqrt(sum) -- convex, decreasing and nearly
//     linear, so the iteration climbs monotonically from the left end of the interval: no bracketing, 3-6 steps;
//   * s is a convex function (signed distance to a convex body), so along the rod's centreline p(t), t in [0, 1], its
//     derivative n(t) . (p1 - p0) -- n the outward normal at the closest point -- is monotone: bisection on its sign
//     (kBisections halvings) finds the closest approach; the derivative's sign at the ends decides the end-point cases.
//   separation = s(p(t*)) - r_rod, contact points p(t*) and the closest surface point, normal = -n (rod -> ellipsoid),
//   as before.  Where the centreline stays outside the ellipsoid this is the quantity the old definition minimised
//   (checked against a scan of the pinned point - ellipsoid distance along the centreline, tests); where it enters the
//   ellipsoid the old definition stopped at the surface crossing (distance 0), this one reports the deepest point.
// The oracle restates the same sequence of IEEE operations (oracle/mundy_oracle.hpp, segment_ellipsoid): the GPU results
// are bit-identical to it.  ~10^4 instructions per pair against 1.5 * 10^6.
#pragma once
#include "ellipsoid_device.hpp"

namespace mhip {
namespace segell {

constexpr int kNewtonMax = 64;    // (3-6 in practice; the loop ends when the iterate stops moving)
constexpr int kBisections = 48;   // interval 2^-48 of the centreline
constexpr double kTiny = 1e-290;
constexpr double kRelTiny = 1e-100;  // a coordinate this far below the point's largest one is zero for the case analysis

// root of Q(u) = (r0 z0 / (u + r0 - 1))^2 + (r1 z1 / (u + r1 - 1))^2 + (z2 / u)^2 = 1 on u > 0 (Eberly's s = u - 1:
// the unknown is kept as the distance from the pole at s = -1, which a point next to the plane of the two long axes
// approaches to within its z2 -- s itself could not resolve that next to -1).  Newton on 1 - 1 / sqrt(Q), the
// secular-equation form: 1 / sqrt(Q) is concave and nearly linear in u, also next to the pole, where Newton on Q - 1
// itself crawls: u+ = u + Q (sqrt(Q) - 1) / (-Q' / 2), monotone from the left end u = z2, where Q >= 1.
// m0 = r0 - 1, m1 = r1 - 1 (>= 0)
__device__ inline double root3(double r0, double r1, double m0, double m1, double z0, double z1, double z2) {
  double u = z2;
  for (int it = 0; it < kNewtonMax; ++it) {
    const double d0 = u + m0, d1 = u + m1;
    const double q0 = r0 * z0 / d0, q1 = r1 * z1 / d1, q2 = z2 / u;
    const double Q = q0 * q0 + q1 * q1 + q2 * q2;
    if (!(Q > 1.0)) break;
    const double dg = q0 * q0 / d0 + q1 * q1 / d1 + q2 * q2 / u;  // -Q'(u) / 2
    const double un = u + Q * (sqrt(Q) - 1.0) / dg;
    if (!(un > u)) break;
    u = un;
  }
  return u;
}
__device__ inline double root2(double r0, double m0, double z0, double z1) {
  double u = z1;
  for (int it = 0; it < kNewtonMax; ++it) {
    const double d0 = u + m0;
    const double q0 = r0 * z0 / d0, q1 = z1 / u;
    const double Q = q0 * q0 + q1 * q1;
    if (!(Q > 1.0)) break;
    const double dg = q0 * q0 / d0 + q1 * q1 / u;
    const double un = u + Q * (sqrt(Q) - 1.0) / dg;
    if (!(un > u)) break;
    u = un;
  }
  return u;
}

// closest point (x0, x1) of the ellipse (x/e0)^2 + (y/e1)^2 = 1, e0 >= e1, to (y0, y1) >= 0; returns the distance
__device__ inline double ellipse2(double e0, double e1, double y0, double y1, double& x0, double& x1) {
  if (y1 > 0.0) {
    if (y0 > 0.0) {
      const double z0 = y0 / e0, z1 = y1 / e1;
      const double g = z0 * z0 + z1 * z1 - 1.0;
      if (g != 0.0) {
        const double q = e0 / e1;
        const double r0 = q * q, m0 = (q - 1.0) * (q + 1.0);
        const double u = root2(r0, m0, z0, z1);
        x0 = r0 * y0 / (u + m0);
        x1 = y1 / u;
        const double a = x0 - y0, b = x1 - y1;
        return sqrt(a * a + b * b);
      }
      x0 = y0;
      x1 = y1;
      return 0.0;
    }
    x0 = 0.0;
    x1 = e1;
    return fabs(y1 - e1);
  }
  const double numer0 = e0 * y0, denom0 = e0 * e0 - e1 * e1;
  if (numer0 < denom0) {
    const double xde0 = numer0 / denom0;
    x0 = e0 * xde0;
    x1 = e1 * sqrt(1.0 - xde0 * xde0);
    const double a = x0 - y0;
    return sqrt(a * a + x1 * x1);
  }
  x0 = e0;
  x1 = 0.0;
  return fabs(y0 - e0);
}

// closest point x of the ellipsoid with semi-axes e0 >= e1 >= e2 to y >= 0 (first octant); returns the distance
__device__ inline double ellipsoid3(double e0, double e1, double e2, double y0, double y1, double y2, double& x0,
                                    double& x1, double& x2) {
  if (y2 > 0.0) {
    if (y1 > 0.0) {
      if (y0 > 0.0) {
        const double z0 = y0 / e0, z1 = y1 / e1, z2 = y2 / e2;
        const double g = z0 * z0 + z1 * z1 + z2 * z2 - 1.0;
        if (g != 0.0) {
          const double q0 = e0 / e2, q1 = e1 / e2;
          const double r0 = q0 * q0, r1 = q1 * q1;
          const double m0 = (q0 - 1.0) * (q0 + 1.0), m1 = (q1 - 1.0) * (q1 + 1.0);
          const double u = root3(r0, r1, m0, m1, z0, z1, z2);
          x0 = r0 * y0 / (u + m0);
          x1 = r1 * y1 / (u + m1);
          x2 = y2 / u;
          const double a = x0 - y0, b = x1 - y1, c = x2 - y2;
          return sqrt(a * a + b * b + c * c);
        }
        x0 = y0;
        x1 = y1;
        x2 = y2;
        return 0.0;
      }
      x0 = 0.0;
      return ellipse2(e1, e2, y1, y2, x1, x2);
    }
    if (y0 > 0.0) {
      x1 = 0.0;
      return ellipse2(e0, e2, y0, y2, x0, x2);
    }
    x0 = 0.0;
    x1 = 0.0;
    x2 = e2;
    return fabs(y2 - e2);
  }
  const double denom0 = e0 * e0 - e2 * e2, denom1 = e1 * e1 - e2 * e2;
  const double numer0 = e0 * y0, numer1 = e1 * y1;
  if (numer0 < denom0 && numer1 < denom1) {
    const double xde0 = numer0 / denom0, xde1 = numer1 / denom1;
    const double discr = 1.0 - xde0 * xde0 - xde1 * xde1;
    if (discr > 0.0) {
      x0 = e0 * xde0;
      x1 = e1 * xde1;
      x2 = e2 * sqrt(discr);
      const double a = x0 - y0, b = x1 - y1;
      return sqrt(a * a + b * b + x2 * x2);
    }
  }
  x2 = 0.0;
  return ellipse2(e0, e1, y0, y1, x0, x1);
}

struct PointResult {
  double sdist;  // signed: negative inside
  V3 x, n;       // closest surface point and the outward unit normal there (body frame)
};
// (conditional exchanges of named scalars: no run-time indexed arrays, nothing goes to scratch)
#define MHIP_SEGELL_SWAP(c, a, b)  \
  {                                \
    const double ta_ = (c) ? b : a; \
    b = (c) ? a : b;               \
    a = ta_;                       \
  }
__device__ inline PointResult point_ellipsoid_body(V3 y, V3 e) {
  const double sx = y.x < 0.0 ? -1.0 : 1.0, sy = y.y < 0.0 ? -1.0 : 1.0, sz = y.z < 0.0 ? -1.0 : 1.0;
  double e0 = e.x, e1 = e.y, e2 = e.z;
  double a0 = sx * y.x, a1 = sy * y.y, a2 = sz * y.z;
  // (a coordinate that a division by a semi-axis could flush to zero IS zero for the case analysis below)
  // ... and so is one more than 100 decades below the point's largest coordinate: with equal semi-axes (m = 0, the
  // pole term d = u) Newton starts at u = z2 and Q = (r z / u)^2 would overflow; the symmetry-plane branches are exact
  // for such a point to 1e-100 relative
  double amax = a0 < a1 ? a1 : a0;
  amax = amax < a2 ? a2 : amax;
  const double floor_ = kRelTiny * amax < kTiny ? kTiny : kRelTiny * amax;
  a0 = a0 < floor_ ? 0.0 : a0;
  a1 = a1 < floor_ ? 0.0 : a1;
  a2 = a2 < floor_ ? 0.0 : a2;
  double i0 = 0.0, i1 = 1.0, i2 = 2.0;  // which lab axis sits in which sorted place
  {
    const bool c = e0 < e1;
    MHIP_SEGELL_SWAP(c, e0, e1) MHIP_SEGELL_SWAP(c, a0, a1) MHIP_SEGELL_SWAP(c, i0, i1)
  }
  {
    const bool c = e1 < e2;
    MHIP_SEGELL_SWAP(c, e1, e2) MHIP_SEGELL_SWAP(c, a1, a2) MHIP_SEGELL_SWAP(c, i1, i2)
  }
  {
    const bool c = e0 < e1;
    MHIP_SEGELL_SWAP(c, e0, e1) MHIP_SEGELL_SWAP(c, a0, a1) MHIP_SEGELL_SWAP(c, i0, i1)
  }
  double x0, x1, x2;
  const double dist = ellipsoid3(e0, e1, e2, a0, a1, a2, x0, x1, x2);
  const double w0 = a0 / e0, w1 = a1 / e1, w2 = a2 / e2;
  const bool inside = w0 * w0 + w1 * w1 + w2 * w2 < 1.0;
  double m0 = x0 / (e0 * e0), m1 = x1 / (e1 * e1), m2 = x2 / (e2 * e2);
  const double inv = 1.0 / sqrt(m0 * m0 + m1 * m1 + m2 * m2);
  m0 *= inv;
  m1 *= inv;
  m2 *= inv;
  PointResult r;
  r.sdist = inside ? -dist : dist;
  r.x = V3{sx * (i0 == 0.0 ? x0 : (i1 == 0.0 ? x1 : x2)), sy * (i0 == 1.0 ? x0 : (i1 == 1.0 ? x1 : x2)),
           sz * (i0 == 2.0 ? x0 : (i1 == 2.0 ? x1 : x2))};
  r.n = V3{sx * (i0 == 0.0 ? m0 : (i1 == 0.0 ? m1 : m2)), sy * (i0 == 1.0 ? m0 : (i1 == 1.0 ? m1 : m2)),
           sz * (i0 == 2.0 ? m0 : (i1 == 2.0 ? m1 : m2))};
  return r;
}
#undef MHIP_SEGELL_SWAP

struct SegmentResult {
  double sdist, t;   // signed distance of the centreline's closest approach, its parameter in [0, 1]
  V3 p, x, n;        // the centreline point, the closest surface point, the outward unit normal there (lab frame)
};
__device__ inline SegmentResult segment_ellipsoid(V3 p0, V3 p1, const EllipsoidD& el) {
  const Quat qc{el.q.w, -el.q.x, -el.q.y, -el.q.z};
  const V3 y0 = qrot(qc, p0 - el.c), y1 = qrot(qc, p1 - el.c);
  const V3 dy = y1 - y0;
  double t = 0.0;
  PointResult best = point_ellipsoid_body(y0, el.r);
  if (dot(best.n, dy) < 0.0) {  // the distance still falls at t = 0
    const PointResult end = point_ellipsoid_body(y1, el.r);
    if (!(dot(end.n, dy) > 0.0)) {  // ... and at t = 1
      best = end;
      t = 1.0;
    } else {
      double lo = 0.0, hi = 1.0;
      for (int it = 0; it < kBisections; ++it) {
        const double mid = 0.5 * (lo + hi);
        const PointResult pm = point_ellipsoid_body(y0 + mid * dy, el.r);
        const double h = dot(pm.n, dy);
        if (h < 0.0) lo = mid; else hi = mid;
      }
      t = 0.5 * (lo + hi);
      best = point_ellipsoid_body(y0 + t * dy, el.r);
    }
  }
  SegmentResult r;
  r.sdist = best.sdist;
  r.t = t;
  r.p = p0 + t * (p1 - p0);
  r.x = qrot(el.q, best.x) + el.c;
  r.n = qrot(el.q, best.n);
  return r;
}

}  // namespace segell
}  // namespace mhip
