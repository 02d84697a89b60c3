// mundy_oracle.hpp -- CPU restatement of MuNDy's contact hot path.
//
// TEST INFRASTRUCTURE ONLY.  Nothing in the product path (mundy_amd/, include/) may include, link or call this
// file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker / CPU baseline.
//
// Every function restates, in plain scalar C++ and with the reference's operation order, one function of the
// reference (cited as path:line relative to /root/reference).  No reference source is copied: the reference is
// Kokkos-templated C++20 over accessor/ownership types; this is flat structs and loops.
//
// Parity pinning: the reference cannot be built in this image (it needs Kokkos, KokkosKernels, STK, OpenRAND, GTest,
// none of which exist here, and hand-written stand-ins for them are not allowed), so this oracle is pinned by the
// reference's own known-answer tests and analytic/manufactured test cases, restated in tests/test_oracle_*.py:
//   mundy/geom/tests/unit_tests/UnitTestSegmentSegment.cpp:417-472 (two 17-digit KATs) and :74-350 (generators)
//   mundy/geom/tests/unit_tests/UnitTestComputeAABB.cpp:137-262, UnitTestComputeBoundingRadius.cpp:120-242
//   mundy/math/tests/unit_tests/UnitTestConvex.cpp:46-143,608-625
//   mundy/math/tests/unit_tests/UnitTestZMorton.cpp:217-380, UnitTestHilbert.cpp:48-387
//   mundy/geom/tests/unit_tests/UnitTestPeriodicity.cpp:623-948 (properties)
// The neighbour-search predicate is third-party in the reference (stk::search / ArborX, absent): "parity unpinned"
// for pair sets; the predicate is defined here from MuNDy's own geom::intersects / bounding-sphere code.
//
// Build with -ffp-contract=off so that a*b+c is two roundings, as in the device code.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

namespace moracle {

// ---------------------------------------------------------------------------------------------------------------
// mundy::math -- Vector3 / Quaternion subset
// ---------------------------------------------------------------------------------------------------------------
struct V3 {
  double x, y, z;
  double& operator[](int i) { return (&x)[i]; }
  const double& operator[](int i) const { return (&x)[i]; }
};
struct Quat {
  double w, x, y, z;
};

inline V3 operator+(const V3& a, const V3& b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(const V3& a, const V3& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(double s, const V3& a) { return {s * a.x, s * a.y, s * a.z}; }
inline V3 operator*(const V3& a, double s) { return {a.x * s, a.y * s, a.z * s}; }

// mundy/math/src/mundy_math/impl/VectorImpl.hpp:339-344: unary right fold, a0*b0 + (a1*b1 + a2*b2).
inline double dot(const V3& a, const V3& b) { return a.x * b.x + (a.y * b.y + a.z * b.z); }
// mundy/math/src/mundy_math/Vector.hpp:1150-1156: two_norm = std::sqrt(dot(v, v)).
inline double norm(const V3& a) { return std::sqrt(dot(a, a)); }
// mundy/math/src/mundy_math/Vector3.hpp:93-105.
inline V3 cross(const V3& a, const V3& b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

// mundy/math/src/mundy_math/impl/QuaternionImpl.hpp:142-164.
inline Quat qmul(const Quat& q, const Quat& o) {
  Quat r;
  r.w = q.w * o.w - q.x * o.x - q.y * o.y - q.z * o.z;
  r.x = q.w * o.x + q.x * o.w + q.y * o.z - q.z * o.y;
  r.y = q.w * o.y - q.x * o.z + q.y * o.w + q.z * o.x;
  r.z = q.w * o.z + q.x * o.y - q.y * o.x + q.z * o.w;
  return r;
}
// mundy/math/src/mundy_math/Quaternion.hpp:1208-1219.
inline Quat conjugate(const Quat& q) { return {q.w, -q.x, -q.y, -q.z}; }
// mundy/math/src/mundy_math/Quaternion.hpp:1221-1229: conjugate(q) * (1 / |q|^2), plain left-to-right sum.
inline Quat inverse(const Quat& q) {
  const double inv_norm_squared = 1.0 / (q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
  const Quat c = conjugate(q);
  return {c.w * inv_norm_squared, c.x * inv_norm_squared, c.y * inv_norm_squared, c.z * inv_norm_squared};
}
// mundy/math/src/mundy_math/impl/QuaternionImpl.hpp:184-202: (q * (0, v)) * inverse(q).
inline V3 qrot(const Quat& q, const V3& v) {
  const Quat vq{0.0, v.x, v.y, v.z};
  const Quat qi = inverse(q);
  const Quat r = qmul(qmul(q, vq), qi);
  return {r.x, r.y, r.z};
}
// mundy/math/src/mundy_math/Quaternion.hpp:1489-1505.
inline Quat quat_from_parallel_transport(const V3& from, const V3& to) {
  const double dp = dot(from, to);
  const V3 cp = cross(from, to);
  const double sqrt_term = std::sqrt(0.5 * (1.0 + dp));
  // vec = 0.5 * cross / sqrt_term  ==  (0.5 * cross) / sqrt_term
  const V3 half_cp = 0.5 * cp;
  return {sqrt_term, half_cp.x / sqrt_term, half_cp.y / sqrt_term, half_cp.z / sqrt_term};
}

// mundy/math/src/mundy_math/Tolerance.hpp:38-70.
constexpr double kZeroTol = 1e-15;
constexpr double kRelaxedZeroTol = 1e-8;

// ---------------------------------------------------------------------------------------------------------------
// mundy::geom -- AABB / bounding radius
// ---------------------------------------------------------------------------------------------------------------
struct AABB {
  double lo[3], hi[3];
};

// mundy/geom/src/mundy_geom/compute_aabb.hpp:72-80 (center -/+ ones * radius).
inline AABB compute_aabb_sphere(const V3& c, double r) {
  AABB b;
  for (int k = 0; k < 3; ++k) {
    b.lo[k] = c[k] - 1.0 * r;
    b.hi[k] = c[k] + 1.0 * r;
  }
  return b;
}
// The centreline half vector shared by compute_aabb(Spherocylinder) (compute_aabb.hpp:115-117) and the rod contact
// assembly:  scaled_dir = 0.5 * length * (orientation * z_axis).
inline V3 spherocylinder_half_axis(const Quat& q, double length) {
  const V3 zaxis{0.0, 0.0, 1.0};
  return (0.5 * length) * qrot(q, zaxis);
}
// mundy/geom/src/mundy_geom/compute_aabb.hpp:105-127.
inline AABB compute_aabb_spherocylinder(const V3& c, const Quat& q, double r, double length) {
  const V3 d = spherocylinder_half_axis(q, length);
  const V3 p0 = c - d, p1 = c + d;
  AABB b;
  for (int k = 0; k < 3; ++k) {
    b.lo[k] = std::min(p0[k], p1[k]) - r;
    b.hi[k] = std::max(p0[k], p1[k]) + r;
  }
  return b;
}
// mundy/geom/src/mundy_geom/compute_aabb.hpp:82-103 (exact only for axis-aligned rotations -- quirk kept).
inline AABB compute_aabb_ellipsoid(const V3& c, const Quat& q, const V3& radii) {
  const V3 rr = qrot(q, radii);
  const V3 p0 = c - rr, p1 = c + rr;
  AABB b;
  for (int k = 0; k < 3; ++k) {
    b.lo[k] = std::min(p0[k], p1[k]);
    b.hi[k] = std::max(p0[k], p1[k]);
  }
  return b;
}
// BUILD EXTENSION (flagged option of SURVEY row a7, no reference implementation): the tight, conservative box of a
// rotated ellipsoid.  With the body axes a_j = q * e_j the support of the ellipsoid along lab axis k is
// sqrt(sum_j (r_j a_j[k])^2); the reference's box above is exact only for axis-aligned rotations and can miss contacts.
inline AABB compute_aabb_ellipsoid_conservative(const V3& c, const Quat& q, const V3& radii) {
  const V3 a0 = qrot(q, V3{1.0, 0.0, 0.0}), a1 = qrot(q, V3{0.0, 1.0, 0.0}), a2 = qrot(q, V3{0.0, 0.0, 1.0});
  AABB b;
  for (int k = 0; k < 3; ++k) {
    const double t0 = radii.x * a0[k], t1 = radii.y * a1[k], t2 = radii.z * a2[k];
    const double e = std::sqrt(t0 * t0 + (t1 * t1 + t2 * t2));
    b.lo[k] = c[k] - e;
    b.hi[k] = c[k] + e;
  }
  return b;
}
// mundy/geom/src/mundy_geom/compute_aabb.hpp:129-143 (SpherocylinderSegment).
inline AABB compute_aabb_segment(const V3& p0, const V3& p1, double r) {
  AABB b;
  for (int k = 0; k < 3; ++k) {
    b.lo[k] = std::min(p0[k], p1[k]) - r;
    b.hi[k] = std::max(p0[k], p1[k]) + r;
  }
  return b;
}
// mundy/geom/src/mundy_geom/primitives/AABB.hpp:420-431 (closed test: touching boxes intersect).
inline bool intersects(const AABB& a, const AABB& b) {
  if (a.hi[0] < b.lo[0] || a.hi[1] < b.lo[1] || a.hi[2] < b.lo[2]) return false;
  const bool disjoint2 = b.hi[0] < a.lo[0] || b.hi[1] < a.lo[1] || b.hi[2] < a.lo[2];
  return !disjoint2;
}
// mundy/geom/src/mundy_geom/compute_bounding_radius.hpp:61-81.
inline double bounding_radius_sphere(double r) { return r; }
inline double bounding_radius_ellipsoid(const V3& radii) { return std::max(radii.x, std::max(radii.y, radii.z)); }
inline double bounding_radius_spherocylinder(double r, double length) { return 0.5 * length + r; }
inline double bounding_radius_segment(const V3& p0, const V3& p1, double r) { return 0.5 * norm(p1 - p0) + r; }

// ---------------------------------------------------------------------------------------------------------------
// sin / cos.  The reference calls Kokkos::sin / Kokkos::cos, i.e. the platform's libm (glibc on its CPU builds, the
// device math library on GPU builds): values that differ between platforms in the last ulp, which the L-BFGS line
// search of the ellipsoid distances turns into a different branch -- and occasionally another local minimum -- for a
// fraction of a percent of the pairs.
//   kTrigLibm    std::sin / std::cos: what the reference's host build executes (default; the reference KATs run here)
//   kTrigShared  one fixed sequence of IEEE double operations (no FMA contraction): round x 2/pi to the nearest integer,
//                subtract that multiple of pi/2 in two-part pieces (Cody & Waite), then the usual degree-13 / degree-14
//                minimax polynomials on [-pi/4, pi/4] with the reduction's tail carried through (the construction of
//                fdlibm's kernels; error < 1 ulp for the |x| < 10^5 that occur here).  The device path evaluates exactly
//                this sequence, so in this mode the oracle and the device agree bit for bit wherever only + - x / sqrt
//                are involved -- the mode the GPU parity tests run in.
// ---------------------------------------------------------------------------------------------------------------
enum TrigMode : int { kTrigLibm = 0, kTrigShared = 1 };
inline int& trig_mode() {
  static int mode = kTrigLibm;
  return mode;
}
inline void shared_sincos(double x, double& s, double& c) {
  const double fn = std::rint(x * 6.36619772367581382433e-01);  // x * 2/pi to the nearest integer (ties to even)
  const int n = static_cast<int>(fn);
  // pi/2 = p1 + p1t (p1: 33 bits, so fn * p1 is exact), then p1t = p2 + p2t for the second pass
  double r = x - fn * 1.57079632673412561417e+00;
  double w = fn * 6.07710050650619224932e-11;
  {
    const double t = r;
    w = fn * 6.07710050630396597660e-11;
    r = t - w;
    w = fn * 2.02226624879595063154e-21 - ((t - r) - w);
  }
  const double y0 = r - w;
  const double y1 = (r - y0) - w;  // tail of the reduced argument
  const double z = y0 * y0;
  // sin on [-pi/4, pi/4]
  const double ps = -1.98412698298579493134e-04 +
                    z * (2.75573137070700676789e-06 + z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10));
  const double rs = 8.33333333332248946124e-03 + z * ps;
  const double v = z * y0;
  const double sv = y0 - ((z * (0.5 * y1 - v * rs) - y1) - v * -1.66666666666666324348e-01);
  // cos on [-pi/4, pi/4]
  const double pc = 2.48015872894767294178e-05 +
                    z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11));
  const double rc = z * (4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * pc));
  const double hz = 0.5 * z;
  const double wc = 1.0 - hz;
  const double cv = wc + (((1.0 - wc) - hz) + (z * rc - y0 * y1));
  switch (n & 3) {
    case 0: s = sv; c = cv; break;
    case 1: s = cv; c = -sv; break;
    case 2: s = -sv; c = -cv; break;
    default: s = -cv; c = sv; break;
  }
}
inline void sincos_mode(double x, double& s, double& c) {
  if (trig_mode() == kTrigShared) {
    shared_sincos(x, s, c);
  } else {
    s = std::sin(x);
    c = std::cos(x);
  }
}

// mundy/math/src/mundy_math/Quaternion.hpp:1366-1383 (rotate_quaternion: Delong 2015 App. A eq. 1, then normalize()
// :409-417 with norm :1233-1235, a left-to-right sum of squares).
inline Quat rotate_quaternion(const Quat& q, const V3& omega, double dt) {
  const double w = norm(omega);
  if (w < kZeroTol) return q;
  const double winv = 1.0 / w;
  double sw, cw;
  sincos_mode(0.5 * w * dt, sw, cw);
  const double s = q.w;
  const V3 p{q.x, q.y, q.z};
  const V3 cr = cross(omega, p);
  const double a = s * sw, b = sw * winv;
  const V3 xyz{a * omega.x * winv + cw * p.x + b * cr.x, a * omega.y * winv + cw * p.y + b * cr.y,
               a * omega.z * winv + cw * p.z + b * cr.z};
  const double qw = s * cw - dot(omega, p) * sw * winv;
  const double inv = 1.0 / std::sqrt(qw * qw + xyz.x * xyz.x + xyz.y * xyz.y + xyz.z * xyz.z);
  return Quat{qw * inv, xyz.x * inv, xyz.y * inv, xyz.z * inv};
}

// ---------------------------------------------------------------------------------------------------------------
// mundy::geom -- periodicity (PeriodicScaledMetric only)
// ---------------------------------------------------------------------------------------------------------------
struct PeriodicScaledMetric {
  V3 scale, scale_inv;
  // mundy/geom/src/mundy_geom/periodicity.hpp:756-759.
  explicit PeriodicScaledMetric(const V3& cell) : scale(cell), scale_inv{1.0 / cell.x, 1.0 / cell.y, 1.0 / cell.z} {}
  // :785-788 / :793-796 (elementwise_mul(scale_inv, p)).
  V3 to_fractional(const V3& p) const { return {scale_inv.x * p.x, scale_inv.y * p.y, scale_inv.z * p.z}; }
  V3 from_fractional(const V3& f) const { return {scale.x * f.x, scale.y * f.y, scale.z * f.z}; }
  // :798-803: x - (double)(int64)round(x).
  static double min_image1(double x) { return x - static_cast<double>(static_cast<int64_t>(std::round(x))); }
  // :140-150 impl::safe_unit_mod1<int64_t>.
  static double unit_mod1(double s) {
    const double k = static_cast<double>(static_cast<int64_t>(std::floor(s)));
    double t = s - k;
    if (std::fabs(t - 1.0) < kZeroTol) t = 0.0;
    return t;
  }
  // :812-816.
  V3 sep(const V3& p1, const V3& p2) const {
    const V3 f = to_fractional(p2 - p1);
    return from_fractional({min_image1(f.x), min_image1(f.y), min_image1(f.z)});
  }
  // :818-823.
  V3 wrap(const V3& p) const {
    const V3 f = to_fractional(p);
    return from_fractional({unit_mod1(f.x), unit_mod1(f.y), unit_mod1(f.z)});
  }
};

// Triclinic cell: mundy/geom/src/mundy_geom/periodicity.hpp:233-332 (PeriodicMetric).  h holds the lattice vectors as
// columns, row-major storage h[3*i+j] = h(i,j).  h_inv = math::inverse(h) = adjugate / determinant
// (mundy/math/src/mundy_math/Matrix.hpp:1596-1601) with the Laplace-expansion determinant and cofactors of
// mundy/math/src/mundy_math/impl/MatrixImpl.hpp:481-506 (unary right folds; the +/-1 factors are exact).
struct PeriodicMetric {
  double h[9], h_inv[9];
  static double det2(double a, double b, double c, double d) { return a * d + (-(b * c)); }  // MatrixImpl.hpp:485
  // minor<r, c> of a 3x3 stored row-major, then its 2x2 determinant
  static double minor_det(const double* m, int r, int c) {
    int rows[2], cols[2], k = 0;
    for (int i = 0; i < 3; ++i)
      if (i != r) rows[k++] = i;
    k = 0;
    for (int j = 0; j < 3; ++j)
      if (j != c) cols[k++] = j;
    return det2(m[3 * rows[0] + cols[0]], m[3 * rows[0] + cols[1]], m[3 * rows[1] + cols[0]], m[3 * rows[1] + cols[1]]);
  }
  static double determinant(const double* m) {
    const double t0 = m[0] * minor_det(m, 0, 0), t1 = -(m[1] * minor_det(m, 0, 1)), t2 = m[2] * minor_det(m, 0, 2);
    return t0 + (t1 + t2);
  }
  static void inverse(const double* m, double* out) {
    const double det = determinant(m);
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        // adjugate(i, j) = cofactor(j, i); cofactor sign by flat index parity (MatrixImpl.hpp:505)
        const double cof = minor_det(m, j, i) * (((3 * j + i) % 2 == 0) ? 1.0 : -1.0);
        out[3 * i + j] = cof / det;
      }
  }
  explicit PeriodicMetric(const double* cell) {
    for (int i = 0; i < 9; ++i) h[i] = cell[i];
    inverse(h, h_inv);
  }
  // Matrix * vector = per-row dot (MatrixImpl.hpp:348-355)
  static V3 matvec(const double* m, const V3& v) {
    return {dot(V3{m[0], m[1], m[2]}, v), dot(V3{m[3], m[4], m[5]}, v), dot(V3{m[6], m[7], m[8]}, v)};
  }
  V3 to_fractional(const V3& p) const { return matvec(h_inv, p); }    // :279-281
  V3 from_fractional(const V3& f) const { return matvec(h, f); }      // :286-288
  // :304-307
  V3 sep(const V3& p1, const V3& p2) const {
    const V3 f = to_fractional(p2 - p1);
    return from_fractional({PeriodicScaledMetric::min_image1(f.x), PeriodicScaledMetric::min_image1(f.y),
                            PeriodicScaledMetric::min_image1(f.z)});
  }
  // :312-314
  V3 wrap(const V3& p) const {
    const V3 f = to_fractional(p);
    return from_fractional({PeriodicScaledMetric::unit_mod1(f.x), PeriodicScaledMetric::unit_mod1(f.y),
                            PeriodicScaledMetric::unit_mod1(f.z)});
  }
  // :323-327 translate(point, h * num_images)
  V3 shift_image(const V3& p, const int* n) const {
    return p + matvec(h, V3{static_cast<double>(n[0]), static_cast<double>(n[1]), static_cast<double>(n[2])});
  }
};

// ---------------------------------------------------------------------------------------------------------------
// mundy::geom -- distances
// ---------------------------------------------------------------------------------------------------------------
// mundy/geom/src/mundy_geom/distance/PointPoint.hpp:43-62.
inline double distance_point_point(const V3& p1, const V3& p2, V3* sep = nullptr) {
  const V3 s = p2 - p1;
  if (sep) *sep = s;
  return norm(s);
}
// mundy/geom/src/mundy_geom/distance/SphereSphere.hpp:54-59.
inline double distance_sphere_sphere(const V3& c1, double r1, const V3& c2, double r2) {
  return distance_point_point(c1, c2) - r1 - r2;
}
// mundy/geom/src/mundy_geom/distance/SphereSphere.hpp:66-76 (sep rescaled to surface-to-surface; NaN if coincident).
inline double distance_sphere_sphere(const V3& c1, double r1, const V3& c2, double r2, V3& sep) {
  const double cc = distance_point_point(c1, c2, &sep);
  const double surface_distance = cc - r1 - r2;
  const double f = surface_distance / cc;
  sep = sep * f;
  return surface_distance;
}

// mundy/geom/src/mundy_geom/distance/PointLineSegment.hpp:128-172.  Note arch_length is left unclamped in
// cases 3.1/3.2 (:158-163).
inline double distance_point_segment(const V3& point, const V3& p1, const V3& p2, V3& closest, double& t, V3& sep) {
  const V3 p21 = p2 - p1;
  const double num = dot(p21, point - p1);
  if ((num < kZeroTol) & (num > -kZeroTol)) {
    closest = p1;
    t = 0.0;
  } else {
    const double denom = dot(p21, p21);
    if (denom < kZeroTol) {
      closest = p1;
      t = 0.0;
    } else {
      t = num / denom;
      if (t < 0.0) {
        closest = p1;
      } else if (t > 1.0) {
        closest = p2;
      } else {
        closest = p1 + t * p21;
      }
    }
  }
  return distance_point_point(point, closest, &sep);
}

// mundy/geom/src/mundy_geom/distance/PointSphere.hpp:57-62 (no sep) and :69-79 (sep rescaled to the surface; NaN when
// the point is the centre).
inline double distance_point_sphere(const V3& point, const V3& c, double r) { return distance_point_point(point, c) - r; }
inline double distance_point_sphere(const V3& point, const V3& c, double r, V3& sep) {
  const double center_point_distance = distance_point_point(point, c, &sep);
  const double surface_distance = center_point_distance - r;
  sep = sep * (surface_distance / center_point_distance);
  return surface_distance;
}
// mundy/geom/src/mundy_geom/distance/LineSegmentSphere.hpp:88-100: distance(sphere.center(), line_segment, cp, t, sep)
// minus the radius, sep rescaled (it is the separation PointLineSegment returns: centre -> closest point).
inline double distance_segment_sphere(const V3& p1, const V3& p2, const V3& c, double r, V3& closest, double& t, V3& sep) {
  const double line_center_distance = distance_point_segment(c, p1, p2, closest, t, sep);
  const double surface_distance = line_center_distance - r;
  sep = sep * (surface_distance / line_center_distance);
  return surface_distance;
}

// mundy/geom/src/mundy_geom/distance/LineSegmentLineSegment.hpp:189-318.
inline double distance_segment_segment(const V3& l0, const V3& l1, const V3& m0, const V3& m1, V3& cp1, V3& cp2,
                                       double& s, double& t, V3& sep) {
  const V3 u = l1 - l0;
  const V3 v = m1 - m0;
  const V3 w = l0 - m0;
  const double a = dot(u, u);
  const double b = dot(u, v);
  const double c = dot(v, v);
  const double d = dot(u, w);
  const double e = dot(v, w);
  const double D = a * c - b * b;

  if (D < std::sqrt(kZeroTol)) {
    // colinear: 4 point-segment distances, first exact match of the minimum wins (:236-265)
    V3 c1, c2, c3, c4, s1, s2, s3, s4;
    double t1, t2, t3, t4;
    const double dist1 = distance_point_segment(l0, m0, m1, c1, t1, s1);
    const double dist2 = distance_point_segment(l1, m0, m1, c2, t2, s2);
    const double dist3 = distance_point_segment(m0, l0, l1, c3, t3, s3);
    const double dist4 = distance_point_segment(m1, l0, l1, c4, t4, s4);
    const double min_distance = std::min(std::min(dist1, dist2), std::min(dist3, dist4));
    if (min_distance == dist1) {
      s = 0.0; t = t1; cp1 = l0; cp2 = c1; sep = s1;
    } else if (min_distance == dist2) {
      s = 1.0; t = t2; cp1 = l1; cp2 = c2; sep = s2;
    } else if (min_distance == dist3) {
      s = t3; t = 0.0; cp1 = c3; cp2 = m0; sep = s3;
    } else {
      s = t4; t = 1.0; cp1 = c4; cp2 = m1; sep = s4;
    }
    return min_distance;
  }

  double sN = b * e - c * d;
  double tN = a * e - b * d;
  double sD = D;
  double tD = D;
  if (sN < 0.0) {
    sN = 0.0; tN = e; tD = c;
  } else if (sN > sD) {
    sN = sD; tN = e + b; tD = c;
  }
  if (tN < 0.0) {
    tN = 0.0;
    if (-d < 0.0) {
      sN = 0.0;
    } else if (-d > a) {
      sN = sD;
    } else {
      sN = -d; sD = a;
    }
  } else if (tN > tD) {
    tN = tD;
    if ((-d + b) < 0.0) {
      sN = 0.0;
    } else if ((-d + b) > a) {
      sN = sD;
    } else {
      sN = (-d + b); sD = a;
    }
  }
  s = (std::fabs(sN) < kZeroTol) ? 0.0 : sN / sD;
  t = (std::fabs(tN) < kZeroTol) ? 0.0 : tN / tD;
  cp1 = l0 + s * u;
  cp2 = m0 + t * v;
  return distance_point_point(cp1, cp2, &sep);
}

// Spherocylinder-spherocylinder contact.  Not a reference function: assembled as the deprecated linker
// kernel does (scrap/parameter_interface/linkers/.../SpherocylinderSpherocylinderLinker.cpp:207-247) but with the
// live tree's z reference axis (compute_aabb.hpp:115-117):
//   endpoints = c -/+ 0.5*L*(q*zhat); dist = seg-seg; sep = dist - (r1 + r2); n = (cp2 - cp1) * (1/dist);
//   contact points = the centreline closest points.
struct RodContact {
  double sep;
  V3 normal, cp1, cp2;
  double s, t;
};
inline RodContact contact_segments(const V3& a0, const V3& a1, double ra, const V3& b0, const V3& b1, double rb) {
  RodContact out;
  V3 sepv;
  const double dist = distance_segment_segment(a0, a1, b0, b1, out.cp1, out.cp2, out.s, out.t, sepv);
  const double radius_sum = ra + rb;
  out.sep = dist - radius_sum;
  const double inv = 1.0 / dist;
  // the scrap linker takes left_to_right = cp2 - cp1, NOT the distance function's `sep` output: in the colinear
  // cases 1.3/1.4 (LineSegmentLineSegment.hpp:251-264) `sep` is the point->segment vector of m0/m1 onto segment 1,
  // i.e. it points from segment 2 to segment 1 (reference quirk, reproduced by distance_segment_segment above).
  (void)sepv;
  out.normal = (out.cp2 - out.cp1) * inv;
  return out;
}

// Sphere-sphere contact = distance(Sphere,Sphere) (SphereSphere.hpp:54-59) for the signed separation and the scrap
// app's normal  n = (xj - xi) * (1/|xj - xi|)  (scrap/lcp_spheres/NgpLcp.cpp:360-372).
inline double contact_spheres(const V3& c1, double r1, const V3& c2, double r2, V3& normal) {
  V3 s;
  const double cc = distance_point_point(c1, c2, &s);
  const double inv = 1.0 / cc;
  normal = s * inv;
  return cc - r1 - r2;
}
// Periodic variant: the centre separation comes from PeriodicScaledMetric::sep (periodicity.hpp:812-816).
inline double contact_spheres_periodic(const PeriodicScaledMetric& m, const V3& c1, double r1, const V3& c2,
                                       double r2, V3& normal) {
  const V3 s = m.sep(c1, c2);
  const double cc = norm(s);
  const double inv = 1.0 / cc;
  normal = s * inv;
  return cc - r1 - r2;
}

// ---------------------------------------------------------------------------------------------------------------
// mundy::math minimize: allocation-free L-BFGS + Fletcher line search + central differences
// (mundy/math/src/mundy_math/impl/minimize_impl.hpp:46-605, mundy_math/minimize.hpp:42-51)
// ---------------------------------------------------------------------------------------------------------------
namespace minimize {
constexpr double kEps = 2.220446049250313e-16;  // Kokkos::Experimental::epsilon_v<double>

inline double put_in_range(double mn, double mx, double v) { return (v < mn) ? mn : (v > mx) ? mx : v; }

// minimize_impl.hpp:57-87
inline double poly_min_extrap(double f0, double d0, double f1, double d1, double limit = 1) {
  const double n = 3 * (f1 - f0) - 2 * d0 - d1;
  const double e = d0 + d1 - 2 * (f1 - f0);
  const double temp_sqr = std::max(n * n - 3 * e * d0, 0.0);
  if (temp_sqr < 0) return 0.5;
  if (std::fabs(e) <= kEps) return 0.5;
  const double temp = std::sqrt(temp_sqr);
  const double x1 = (temp - n) / (3 * e);
  const double x2 = -(temp + n) / (3 * e);
  const double y1 = f0 + d0 * x1 + n * x1 * x1 + e * x1 * x1 * x1;
  const double y2 = f0 + d0 * x2 + n * x2 * x2 + e * x2 * x2 * x2;
  const double x = (y1 < y2) ? x1 : x2;
  return put_in_range(0, limit, x);
}

template <size_t N>
struct Vec {
  double v[N];
  double& operator[](size_t i) { return v[i]; }
  const double& operator[](size_t i) const { return v[i]; }
};
template <size_t N>
inline double vdot(const Vec<N>& a, const Vec<N>& b) {  // right fold, as mundy::math::dot
  double acc = a[N - 1] * b[N - 1];
  for (size_t i = N - 1; i-- > 0;) acc = a[i] * b[i] + acc;
  return acc;
}

// central_differences (minimize_impl.hpp:194-230)
template <size_t N, class F>
inline Vec<N> central_diff(const F& f, const Vec<N>& x, double eps) {
  Vec<N> der, e = x;
  for (size_t i = 0; i < N; ++i) {
    const double old_val = e[i];
    e[i] += eps;
    const double delta_plus = f(e);
    e[i] = old_val - eps;
    const double delta_minus = f(e);
    der[i] = (delta_plus - delta_minus) / ((old_val + eps) - (old_val - eps));
    e[i] = old_val;
  }
  return der;
}

// line_search (minimize_impl.hpp:233-405); f and der are functions of the scalar step
template <class F, class D>
inline double line_search(const F& f, double f0, const D& der, double d0, double rho, double sigma, double min_f,
                          size_t max_iter) {
  const double tau1a = 1.4, tau1b = 9, tau2 = 1.0 / 10.0, tau3 = 1.0 / 2.0;
  if (std::fabs(d0) <= std::fabs(f0) * kEps) return 0;
  if (f0 <= min_f) return 0;
  const double mu = (min_f - f0) / (rho * d0);
  double alpha = 1;
  if (mu < 0) alpha = -alpha;
  alpha = put_in_range(0, 0.65 * mu, alpha);
  double last_alpha = 0, last_val = f0, last_val_der = d0;
  double a, b, a_val, b_val, a_val_der, b_val_der;
  const double thresh = std::fabs(sigma * d0);
  size_t itr = 0;
  while (true) {
    ++itr;
    const double val = f(alpha);
    const double val_der = der(alpha);
    if (val <= min_f) return alpha;
    if (val > f0 + rho * alpha * d0 || val >= last_val) {
      a_val = last_val; a_val_der = last_val_der; b_val = val; b_val_der = val_der;
      a = last_alpha; b = alpha;
      break;
    }
    if (std::fabs(val_der) <= thresh) return alpha;
    if (last_alpha == alpha || itr >= max_iter) return alpha;
    if (val_der >= 0) {
      a_val = val; a_val_der = val_der; b_val = last_val; b_val_der = last_val_der;
      a = alpha; b = last_alpha;
      break;
    }
    const double temp = alpha;
    double first, last;
    if (mu > 0) {
      first = std::min(mu, alpha + tau1a * (alpha - last_alpha));
      last = std::min(mu, alpha + tau1b * (alpha - last_alpha));
    } else {
      first = std::max(mu, alpha + tau1a * (alpha - last_alpha));
      last = std::max(mu, alpha + tau1b * (alpha - last_alpha));
    }
    if (last_alpha < alpha) {
      alpha = last_alpha + (alpha - last_alpha) * poly_min_extrap(last_val, last_val_der, val, val_der, 1e10);
    } else {
      alpha = alpha + (last_alpha - alpha) * poly_min_extrap(val, val_der, last_val, last_val_der, 1e10);
    }
    alpha = put_in_range(first, last, alpha);
    last_alpha = temp;
    last_val = val;
    last_val_der = val_der;
  }
  while (true) {
    ++itr;
    const double first = a + tau2 * (b - a);
    const double last = b - tau3 * (b - a);
    alpha = a + (b - a) * poly_min_extrap(a_val, a_val_der, b_val, b_val_der);
    alpha = put_in_range(first, last, alpha);
    const double val = f(alpha);
    const double val_der = der(alpha);
    if (val <= min_f || itr >= max_iter) return alpha;
    if (a == first || b == last) return b;
    const double max_possible_alpha = std::max(std::fabs(a), std::fabs(b));
    if (std::fabs(max_possible_alpha * d0) <= std::fabs(f0) * kEps) return alpha;
    if (val > f0 + rho * alpha * d0 || val >= a_val) {
      b = alpha; b_val = val; b_val_der = val_der;
    } else {
      if (std::fabs(val_der) <= thresh) return alpha;
      if ((b - a) * val_der >= 0) {
        b = a; b_val = a_val; b_val_der = a_val_der;
      }
      a = alpha; a_val = val; a_val_der = val_der;
    }
  }
}

// lbfgs_search_strategy (minimize_impl.hpp:407-566), M = history size
template <size_t M, size_t N>
struct Lbfgs {
  struct Item {
    Vec<N> s, y;
    double rho;
  };
  Item data[M];
  double alpha[M];
  bool been_used = false;
  size_t current_size = 0;
  Vec<N> prev_x, prev_derivative, prev_direction;

  const Vec<N>& next_direction(const Vec<N>& x, const Vec<N>& g) {
    for (size_t k = 0; k < N; ++k) prev_direction[k] = -g[k];
    if (!been_used) {
      been_used = true;
    } else {
      Item t;
      for (size_t k = 0; k < N; ++k) {
        t.s[k] = x[k] - prev_x[k];
        t.y[k] = g[k] - prev_derivative[k];
      }
      const double temp = vdot(t.s, t.y);
      if (std::fabs(temp) > kEps) {
        t.rho = 1.0 / temp;
        if (current_size < M) {
          data[current_size++] = t;
        } else {
          for (size_t i = 1; i < M; ++i) data[i - 1] = data[i];  // rotate_data: drop the oldest (:556-565)
          data[M - 1] = t;
        }
      } else {
        current_size = 0;
      }
      if (current_size > 0) {
        for (size_t i = 0; i < M; ++i) alpha[i] = 0.0;
        for (size_t i = current_size; i-- > 0;) {
          alpha[i] = data[i].rho * vdot(data[i].s, prev_direction);
          for (size_t k = 0; k < N; ++k) prev_direction[k] = prev_direction[k] - alpha[i] * data[i].y[k];
        }
        double H_0 = 1.0 / data[current_size - 1].rho / vdot(data[current_size - 1].y, data[current_size - 1].y);
        H_0 = put_in_range(0.001, 1000.0, H_0);
        for (size_t k = 0; k < N; ++k) prev_direction[k] = H_0 * prev_direction[k];
        for (size_t i = 0; i < current_size; ++i) {
          const double beta = data[i].rho * vdot(data[i].y, prev_direction);
          for (size_t k = 0; k < N; ++k) prev_direction[k] = prev_direction[k] + (alpha[i] - beta) * data[i].s[k];
        }
      }
    }
    prev_x = x;
    prev_derivative = g;
    return prev_direction;
  }
};

// find_min_using_approximate_derivatives (minimize.hpp:42-51 + minimize_impl.hpp:568-599).  NB the reference's callers
// pass their "min_objective_delta" as the THIRD argument, which is min_allowable_cost.
template <size_t M, size_t N, class F>
inline double find_min(const F& cost_func, Vec<N>& x, double min_allowable_cost = -std::numeric_limits<double>::infinity(),
                       double min_objective_delta = 1e-7, double derivative_eps = 1e-7) {
  Lbfgs<M, N> strat;
  // objective_delta_stop_strategy (minimize_impl.hpp:151-191)
  bool stop_used = false;
  double prev_funct_value = 0;
  auto should_continue = [&](double funct_value) {
    if (stop_used && std::fabs(funct_value - prev_funct_value) < min_objective_delta) return false;
    stop_used = true;
    prev_funct_value = funct_value;
    return true;
  };
  double cost = cost_func(x);
  Vec<N> g = central_diff(cost_func, x, derivative_eps);
  while (should_continue(cost) && cost > min_allowable_cost) {
    const Vec<N> s = strat.next_direction(x, g);
    auto phi = [&](double a) {
      Vec<N> p;
      for (size_t k = 0; k < N; ++k) p[k] = x[k] + a * s[k];
      return cost_func(p);
    };
    auto dphi = [&](double a) { return (phi(a + derivative_eps) - phi(a - derivative_eps)) /
                                       ((a + derivative_eps) - (a - derivative_eps)); };
    const double alpha = line_search(phi, cost, dphi, vdot(g, s), 0.01, 0.9, min_allowable_cost, 100);
    for (size_t k = 0; k < N; ++k) x[k] = alpha * s[k] + x[k];
    g = central_diff(cost_func, x, derivative_eps);
    cost = cost_func(x);
  }
  return cost;
}
}  // namespace minimize

// ---------------------------------------------------------------------------------------------------------------
// Ellipsoids (mundy_geom/primitives/Ellipsoid.hpp:420-468, distance/EllipsoidEllipsoid.hpp:62-151,
// distance/PointEllipsoid.hpp:61-135)
// ---------------------------------------------------------------------------------------------------------------
struct Ellipsoid {
  V3 center;
  Quat q;
  V3 radii;
};
// Ellipsoid.hpp:420-460
inline V3 map_body_frame_normal_to_ellipsoid(const V3& nhat, const Ellipsoid& el) {
  const double r1 = el.radii.x, r2 = el.radii.y, r3 = el.radii.z;
  const double sign0 = std::copysign(1.0, nhat.x), sign1 = std::copysign(1.0, nhat.y), sign2 = std::copysign(1.0, nhat.z);
  double alpha1, alpha2;
  if (sign0 * nhat.x > kZeroTol) {
    const double tmp0 = 1.0 / (r1 * nhat.x);
    const double tmp1 = tmp0 * r2 * nhat.y;
    const double tmp2 = tmp0 * r3 * nhat.z;
    alpha1 = 1.0 / (1.0 + tmp1 * tmp1);
    alpha2 = 1.0 / (1.0 + tmp2 * tmp2 * alpha1);
  } else if (sign1 * nhat.y > kZeroTol) {
    const double tmp = r3 * nhat.z / (r2 * nhat.y);
    alpha1 = 0.0;
    alpha2 = 1.0 / (1.0 + tmp * tmp);
  } else {
    alpha1 = 0.0;
    alpha2 = 0.0;
  }
  const double sa1 = std::sqrt(alpha1), sa2 = std::sqrt(alpha2);
  const double x = 0.5 * sign0 * ((1.0 + sign0) * r1 + (1.0 - sign0) * r1) * sa1 * sa2;
  const double y = 0.5 * sign1 * ((1.0 + sign1) * r2 + (1.0 - sign1) * r2) * std::sqrt(1.0 - alpha1) * sa2;
  const double z = 0.5 * sign2 * ((1.0 + sign2) * r3 + (1.0 - sign2) * r3) * std::sqrt(1.0 - alpha2);
  return {x, y, z};
}
// Ellipsoid.hpp:462-468
inline V3 map_surface_normal_to_foot_point(const V3& lab_nhat, const Ellipsoid& el) {
  const V3 body_nhat = qrot(conjugate(el.q), lab_nhat);
  const V3 foot = map_body_frame_normal_to_ellipsoid(body_nhat, el);
  return qrot(el.q, foot) + el.center;
}
struct EllipsoidPairResult {
  double dist;
  V3 cp1, cp2, n1, n2;
};
// EllipsoidEllipsoid.hpp:62-151
inline EllipsoidPairResult distance_ellipsoid_ellipsoid(const Ellipsoid& e1, const Ellipsoid& e2) {
  EllipsoidPairResult r;
  auto objective = [&](const minimize::Vec<2>& tp) {
    double st, ct, sp, cp;
    sincos_mode(tp[0], st, ct);
    sincos_mode(tp[1], sp, cp);
    r.n1 = {st * cp, st * sp, ct};
    r.n2 = {-r.n1.x, -r.n1.y, -r.n1.z};
    r.cp1 = map_surface_normal_to_foot_point(r.n1, e1);
    r.cp2 = map_surface_normal_to_foot_point(r.n2, e2);
    return distance_point_point(r.cp1, r.cp2);
  };
  constexpr double pi = 3.141592653589793;
  const double half_pi = 0.5 * pi, one_third_pi = pi / 3.0, five_third_pi = 5.0 * one_third_pi;
  const double theta_guesses[3] = {0.0, half_pi, pi};
  const double phi_guesses[3] = {one_third_pi, pi, five_third_pi};
  double global_dist = std::numeric_limits<double>::infinity();
  minimize::Vec<2> best{{0.0, 0.0}};
  for (int t = 0; t < 3; ++t)
    for (int p = 0; p < 3; ++p) {
      minimize::Vec<2> tp{{theta_guesses[t], phi_guesses[p]}};
      const double d = minimize::find_min<10, 2>(objective, tp, kRelaxedZeroTol);
      if (d < global_dist) {
        global_dist = d;
        best = tp;
      }
    }
  objective(best);
  r.dist = dot(r.cp2 - r.cp1, r.n1);
  return r;
}
// PointEllipsoid.hpp:94-135
inline double distance_point_ellipsoid(const V3& point, const Ellipsoid& el, V3& closest, V3& normal) {
  auto objective = [&](const minimize::Vec<2>& tp) {
    double st, ct, sp, cp;
    sincos_mode(tp[0], st, ct);
    sincos_mode(tp[1], sp, cp);
    normal = {st * cp, st * sp, ct};
    closest = map_surface_normal_to_foot_point(normal, el);
    return distance_point_point(closest, point);
  };
  constexpr double pi = 3.141592653589793;
  const double half_pi = 0.5 * pi, one_third_pi = pi / 3.0, five_third_pi = 5.0 * one_third_pi;
  const double theta_guesses[3] = {0.0, half_pi, pi};
  const double phi_guesses[3] = {one_third_pi, pi, five_third_pi};
  double global_dist = std::numeric_limits<double>::infinity();
  minimize::Vec<2> best{{0.0, 0.0}};
  for (int t = 0; t < 3; ++t)
    for (int p = 0; p < 3; ++p) {
      minimize::Vec<2> tp{{theta_guesses[t], phi_guesses[p]}};
      const double d = minimize::find_min<10, 2>(objective, tp, kRelaxedZeroTol);
      if (d < global_dist) {
        global_dist = d;
        best = tp;
      }
    }
  objective(best);
  return dot(point - closest, normal);
}

// ---------------------------------------------------------------------------------------------------------------
// Rod - ellipsoid (R-E): NO reference function exists (LineSegmentEllipsoid.hpp:21-33 is an empty stub).  BUILD
// EXTENSION, PARITY UNPINNED.  Round 3 definition (see mundy_amd/csrc/segment_ellipsoid.hpp for the derivation): the
// closest approach of the rod's centreline to the ellipsoid, from
//   (a) the exact signed distance of a point to an ellipsoid -- closest point x_i = e_i^2 y_i / (tau + e_i^2), tau the
//       root of the Lagrange condition (Eberly's first-octant / sorted-axes case analysis; Newton on the
//       secular-equation form of the condition), and
//   (b) bisection on the sign of the derivative n(t) . (p1 - p0) of that (convex) distance along the centreline.
// This CPU restatement keeps the product's arithmetic (the same IEEE operations on the same values in the same order,
// so that the GPU can be compared bit for bit) but not its code: the axes are sorted through an index permutation
// here, by conditional exchanges of registers there.  What pins the DEFINITION is independent of both: the scan of the
// reference-pinned point - ellipsoid distance along the centreline (tests/test_oracle_ellipsoid_kat.py).
// ---------------------------------------------------------------------------------------------------------------
namespace segell {
constexpr int kNewtonMax = 64, kBisections = 48;
constexpr double kTiny = 1e-290;
constexpr double kRelTiny = 1e-100;  // a coordinate this far below the point's largest one is zero for the case analysis
// root in u > 0 of sum_k (r[k] z[k] / (u + m[k]))^2 = 1 with m[k] = r[k] - 1, r[N - 1] = 1 (u = Eberly's s + 1: the
// distance from the pole, which can be as small as z[N - 1]): Newton on 1 - 1 / sqrt(sum) from u = z[N - 1]
template <int N>
inline double secular_root(const double (&r)[N], const double (&m)[N], const double (&z)[N]) {
  double u = z[N - 1];
  for (int it = 0; it < kNewtonMax; ++it) {
    double Q = 0.0, dg = 0.0;
    for (int k = 0; k < N; ++k) {
      const double d = (k == N - 1) ? u : u + m[k];
      const double q = (k == N - 1) ? z[k] / d : r[k] * z[k] / d;
      Q = (k == 0) ? q * q : Q + q * q;
      dg = (k == 0) ? q * q / d : dg + q * q / d;
    }
    if (!(Q > 1.0)) break;
    const double un = u + Q * (std::sqrt(Q) - 1.0) / dg;
    if (!(un > u)) break;
    u = un;
  }
  return u;
}
// closest point of the ellipse with semi-axes e0 >= e1 to (y0, y1) >= 0
inline double closest_on_ellipse(double e0, double e1, double y0, double y1, double& x0, double& x1) {
  if (y1 > 0.0) {
    if (y0 > 0.0) {
      const double z[2] = {y0 / e0, y1 / e1};
      const double g = z[0] * z[0] + z[1] * z[1] - 1.0;
      if (g == 0.0) {
        x0 = y0; x1 = y1;
        return 0.0;
      }
      const double ratio = e0 / e1;
      const double r[2] = {ratio * ratio, 1.0}, m[2] = {(ratio - 1.0) * (ratio + 1.0), 0.0};
      const double u = secular_root<2>(r, m, z);
      x0 = r[0] * y0 / (u + m[0]);
      x1 = y1 / u;
      const double a = x0 - y0, b = x1 - y1;
      return std::sqrt(a * a + b * b);
    }
    x0 = 0.0; x1 = e1;
    return std::fabs(y1 - e1);
  }
  const double numer0 = e0 * y0, denom0 = e0 * e0 - e1 * e1;
  if (numer0 < denom0) {
    const double xde0 = numer0 / denom0;
    x0 = e0 * xde0;
    x1 = e1 * std::sqrt(1.0 - xde0 * xde0);
    const double a = x0 - y0;
    return std::sqrt(a * a + x1 * x1);
  }
  x0 = e0; x1 = 0.0;
  return std::fabs(y0 - e0);
}
// closest point of the ellipsoid with semi-axes e[0] >= e[1] >= e[2] to y >= 0
inline double closest_on_ellipsoid(const double (&e)[3], const double (&y)[3], double (&x)[3]) {
  if (y[2] > 0.0) {
    if (y[1] > 0.0) {
      if (y[0] > 0.0) {
        const double z[3] = {y[0] / e[0], y[1] / e[1], y[2] / e[2]};
        const double g = z[0] * z[0] + z[1] * z[1] + z[2] * z[2] - 1.0;
        if (g == 0.0) {
          x[0] = y[0]; x[1] = y[1]; x[2] = y[2];
          return 0.0;
        }
        const double ratio0 = e[0] / e[2], ratio1 = e[1] / e[2];
        const double r[3] = {ratio0 * ratio0, ratio1 * ratio1, 1.0};
        const double m[3] = {(ratio0 - 1.0) * (ratio0 + 1.0), (ratio1 - 1.0) * (ratio1 + 1.0), 0.0};
        const double u = secular_root<3>(r, m, z);
        x[0] = r[0] * y[0] / (u + m[0]);
        x[1] = r[1] * y[1] / (u + m[1]);
        x[2] = y[2] / u;
        const double a = x[0] - y[0], b = x[1] - y[1], c = x[2] - y[2];
        return std::sqrt(a * a + b * b + c * c);
      }
      x[0] = 0.0;
      return closest_on_ellipse(e[1], e[2], y[1], y[2], x[1], x[2]);
    }
    if (y[0] > 0.0) {
      x[1] = 0.0;
      return closest_on_ellipse(e[0], e[2], y[0], y[2], x[0], x[2]);
    }
    x[0] = 0.0; x[1] = 0.0; x[2] = e[2];
    return std::fabs(y[2] - e[2]);
  }
  const double denom0 = e[0] * e[0] - e[2] * e[2], denom1 = e[1] * e[1] - e[2] * e[2];
  const double numer0 = e[0] * y[0], numer1 = e[1] * y[1];
  if (numer0 < denom0 && numer1 < denom1) {
    const double xde0 = numer0 / denom0, xde1 = numer1 / denom1;
    const double discr = 1.0 - xde0 * xde0 - xde1 * xde1;
    if (discr > 0.0) {
      x[0] = e[0] * xde0;
      x[1] = e[1] * xde1;
      x[2] = e[2] * std::sqrt(discr);
      const double a = x[0] - y[0], b = x[1] - y[1];
      return std::sqrt(a * a + b * b + x[2] * x[2]);
    }
  }
  x[2] = 0.0;
  return closest_on_ellipse(e[0], e[1], y[0], y[1], x[0], x[1]);
}
struct PointResult {
  double sdist;
  V3 x, n;
};
// signed distance (negative inside), closest surface point and outward unit normal, all in the ellipsoid's body frame
inline PointResult point_ellipsoid_body(const V3& yv, const V3& radii) {
  const double yy[3] = {yv.x, yv.y, yv.z}, rr[3] = {radii.x, radii.y, radii.z};
  double sgn[3];
  for (int k = 0; k < 3; ++k) sgn[k] = yy[k] < 0.0 ? -1.0 : 1.0;
  // places 0, 1, 2 of the sorted order hold the axes perm[0..2]: the exchange network (0,1), (1,2), (0,1) on strict <
  int perm[3] = {0, 1, 2};
  if (rr[perm[0]] < rr[perm[1]]) std::swap(perm[0], perm[1]);
  if (rr[perm[1]] < rr[perm[2]]) std::swap(perm[1], perm[2]);
  if (rr[perm[0]] < rr[perm[1]]) std::swap(perm[0], perm[1]);
  double e[3], y[3], x[3];
  // (a coordinate a division by a semi-axis could flush to zero IS zero below, and so is one more than 100 decades
  // below the point's largest: with equal semi-axes the Newton start u = z2 would overflow Q -- the device's rule)
  double amax = sgn[0] * yy[0] < sgn[1] * yy[1] ? sgn[1] * yy[1] : sgn[0] * yy[0];
  amax = amax < sgn[2] * yy[2] ? sgn[2] * yy[2] : amax;
  const double floor_ = kRelTiny * amax < kTiny ? kTiny : kRelTiny * amax;
  for (int k = 0; k < 3; ++k) {
    e[k] = rr[perm[k]];
    y[k] = sgn[perm[k]] * yy[perm[k]];
    if (y[k] < floor_) y[k] = 0.0;
  }
  const double dist = closest_on_ellipsoid(e, y, x);
  const double w0 = y[0] / e[0], w1 = y[1] / e[1], w2 = y[2] / e[2];
  const bool inside = w0 * w0 + w1 * w1 + w2 * w2 < 1.0;
  double m[3];
  for (int k = 0; k < 3; ++k) m[k] = x[k] / (e[k] * e[k]);
  const double inv = 1.0 / std::sqrt(m[0] * m[0] + m[1] * m[1] + m[2] * m[2]);
  double xo[3], no[3];
  for (int k = 0; k < 3; ++k) {
    xo[perm[k]] = sgn[perm[k]] * x[k];
    no[perm[k]] = sgn[perm[k]] * (m[k] * inv);
  }
  return {inside ? -dist : dist, {xo[0], xo[1], xo[2]}, {no[0], no[1], no[2]}};
}
struct SegmentResult {
  double sdist, t;
  V3 p, x, n;  // centreline point, closest surface point, outward unit normal there (lab frame)
};
inline SegmentResult segment_ellipsoid(const V3& p0, const V3& p1, const Ellipsoid& el) {
  const Quat qc = conjugate(el.q);
  const V3 y0 = qrot(qc, p0 - el.center), y1 = qrot(qc, p1 - el.center);
  const V3 dy = y1 - y0;
  double t = 0.0;
  PointResult best = point_ellipsoid_body(y0, el.radii);
  if (dot(best.n, dy) < 0.0) {
    const PointResult end = point_ellipsoid_body(y1, el.radii);
    if (!(dot(end.n, dy) > 0.0)) {
      best = end;
      t = 1.0;
    } else {
      double lo = 0.0, hi = 1.0;
      for (int it = 0; it < kBisections; ++it) {
        const double mid = 0.5 * (lo + hi);
        const double h = dot(point_ellipsoid_body(y0 + mid * dy, el.radii).n, dy);
        if (h < 0.0) lo = mid; else hi = mid;
      }
      t = 0.5 * (lo + hi);
      best = point_ellipsoid_body(y0 + t * dy, el.radii);
    }
  }
  return {best.sdist, t, p0 + t * (p1 - p0), qrot(el.q, best.x) + el.center, qrot(el.q, best.n)};
}
}  // namespace segell

// ---------------------------------------------------------------------------------------------------------------
// Mixed-shape contact generation (BASELINE configs[4]).  Body kinds: 0 sphere, 1 spherocylinder, 2 ellipsoid;
// shape = (r, -, -) / (r, L, -) / (r1, r2, r3).  Per pair class:
//   S-S  contact_spheres                               (SphereSphere.hpp:54-59, NgpLcp.cpp:360-372)
//   R-R  contact_segments                              (LineSegmentLineSegment.hpp:189-318 + linker assembly)
//   E-E  distance_ellipsoid_ellipsoid                  (EllipsoidEllipsoid.hpp:106-151)
//   S-R  point-segment distance minus (r_s + r_rod), normal = (closest - centre)/dist, contact points = sphere centre
//        and centreline point (scrap/.../SphereSpherocylinderLinker.cpp:210-239; LineSegmentSphere.hpp:56-61)
//   S-E  signed distance of the sphere's centre to the ellipsoid - r_s, normal = -ellipsoid normal (the reference's
//        SphereEllipsoid.hpp:21-33 is an empty stub: a build-side definition, PARITY UNPINNED).  Route 0 (default): the
//        exact distance (segell, as a rod of zero length); route 1: distance(Point, Ellipsoid), the reference's own
//        nine-start L-BFGS (SURVEY 8f.4's routing) -- the two agree to that routine's 1e-4
//   R-E  NO reference function exists (LineSegmentEllipsoid.hpp:21-33 is an empty stub).  Build extension, PARITY
//        UNPINNED: segell::segment_ellipsoid above -- the closest approach of the rod's centreline to the ellipsoid
//        (exact signed point - ellipsoid distance, minimised along the centreline), minus the rod radius, as the
//        sphere classes do; contact points = the centreline point and the closest surface point, normal = rod ->
//        ellipsoid.  A rod of zero length gives the exact S-E.  (Rounds 1-2 ran distance(Point, Ellipsoid)'s
//        nine-start L-BFGS with the centreline's closest point as the point: the same quantity for a centreline
//        outside the ellipsoid, to that minimiser's 1e-4, at 1 280 objective evaluations per pair.)
// The pair is evaluated in canonical class order (lower kind first) and flipped back if the list order is the reverse.
// ---------------------------------------------------------------------------------------------------------------
enum BodyKind : int { kSphere = 0, kRod = 1, kEllipsoid = 2 };
// S-E route: 0 = exact (segell, a rod of zero length), 1 = distance_point_ellipsoid (the reference's minimiser)
inline int& sphere_ellipsoid_route() {
  static int route = 0;
  return route;
}
struct MixedBody {
  int kind;
  V3 c;
  Quat q;
  V3 shape;
};
struct MixedContact {
  double sep;
  V3 normal, cp1, cp2;
};
inline MixedContact contact_mixed_canonical(const MixedBody& A, const MixedBody& B) {
  MixedContact o;
  if (A.kind == kSphere && B.kind == kSphere) {
    o.sep = contact_spheres(A.c, A.shape.x, B.c, B.shape.x, o.normal);
    o.cp1 = A.c;
    o.cp2 = B.c;
  } else if (A.kind == kRod && B.kind == kRod) {
    const V3 da = spherocylinder_half_axis(A.q, A.shape.y), db = spherocylinder_half_axis(B.q, B.shape.y);
    const RodContact rc = contact_segments(A.c - da, A.c + da, A.shape.x, B.c - db, B.c + db, B.shape.x);
    o.sep = rc.sep; o.normal = rc.normal; o.cp1 = rc.cp1; o.cp2 = rc.cp2;
  } else if (A.kind == kEllipsoid && B.kind == kEllipsoid) {
    const EllipsoidPairResult r = distance_ellipsoid_ellipsoid({A.c, A.q, A.shape}, {B.c, B.q, B.shape});
    o.sep = r.dist; o.normal = r.n1; o.cp1 = r.cp1; o.cp2 = r.cp2;
  } else if (A.kind == kSphere && B.kind == kRod) {
    const V3 d = spherocylinder_half_axis(B.q, B.shape.y);
    V3 closest, sepv;
    double t;
    const double dist = distance_point_segment(A.c, B.c - d, B.c + d, closest, t, sepv);
    const double radius_sum = A.shape.x + B.shape.x;
    o.sep = dist - radius_sum;
    const double inv = 1.0 / dist;
    o.normal = (closest - A.c) * inv;
    o.cp1 = A.c;
    o.cp2 = closest;
  } else if (A.kind == kSphere && B.kind == kEllipsoid && sphere_ellipsoid_route() == 0) {
    const segell::SegmentResult r = segell::segment_ellipsoid(A.c, A.c, Ellipsoid{B.c, B.q, B.shape});
    o.sep = r.sdist - A.shape.x;
    o.normal = {-r.n.x, -r.n.y, -r.n.z};
    o.cp1 = r.p;
    o.cp2 = r.x;
  } else if (A.kind == kSphere && B.kind == kEllipsoid) {
    V3 closest, ne;
    const double d = distance_point_ellipsoid(A.c, {B.c, B.q, B.shape}, closest, ne);
    o.sep = d - A.shape.x;
    o.normal = {-ne.x, -ne.y, -ne.z};
    o.cp1 = A.c;
    o.cp2 = closest;
  } else {  // rod - ellipsoid (extension)
    const V3 hd = spherocylinder_half_axis(A.q, A.shape.y);
    const segell::SegmentResult r = segell::segment_ellipsoid(A.c - hd, A.c + hd, Ellipsoid{B.c, B.q, B.shape});
    o.sep = r.sdist - A.shape.x;
    o.normal = {-r.n.x, -r.n.y, -r.n.z};
    o.cp1 = r.p;
    o.cp2 = r.x;
  }
  return o;
}
inline MixedContact contact_mixed(const MixedBody& bi, const MixedBody& bj) {
  if (bi.kind <= bj.kind) return contact_mixed_canonical(bi, bj);
  MixedContact o = contact_mixed_canonical(bj, bi);
  std::swap(o.cp1, o.cp2);
  o.normal = {-o.normal.x, -o.normal.y, -o.normal.z};
  return o;
}
inline AABB compute_aabb_mixed(const MixedBody& b) {
  if (b.kind == kSphere) return compute_aabb_sphere(b.c, b.shape.x);
  if (b.kind == kRod) return compute_aabb_spherocylinder(b.c, b.q, b.shape.x, b.shape.y);
  return compute_aabb_ellipsoid(b.c, b.q, b.shape);
}
inline double bounding_radius_mixed(const MixedBody& b) {
  if (b.kind == kSphere) return bounding_radius_sphere(b.shape.x);
  if (b.kind == kRod) return bounding_radius_spherocylinder(b.shape.x, b.shape.y);
  return bounding_radius_ellipsoid(b.shape);
}

// ---------------------------------------------------------------------------------------------------------------
// mundy::math::convex -- spaces, residual policies, BB step, BBPGD (convex.hpp)
// ---------------------------------------------------------------------------------------------------------------
enum SpaceKind : int { kUnconstrained = 0, kLowerBound = 1, kUpperBound = 2, kBounded = 3 };
// mundy/math/src/mundy_math/convex.hpp:46-115.
struct Space {
  int kind;
  double lo, hi;
  double project(double x) const {
    switch (kind) {
      case kLowerBound: return std::max(x, lo);
      case kUpperBound: return std::min(x, hi);
      case kBounded: return std::min(std::max(x, lo), hi);
      default: return x;
    }
  }
};
enum ResidualKind : int { kProjectedDiff = 0, kProjectedGradient = 1 };

// ---------------------------------------------------------------------------------------------------------------
// Summation mode of every reduction whose rounding feeds the Barzilai-Borwein step (diff_dot x2, the per-body force /
// torque sums of the contact operators, the rows of the dense operator).
//   kSumSerial       plain left-to-right double sums: what Kokkos-Serial executes (the reference; on a parallel
//                    backend its own order -- and, with atomics, its iteration count -- changes from run to run).
//   kSumCompensated  the same sums carried as unevaluated pairs hi + lo (TwoSum cascade, ~106 bits) and rounded once:
//                    the correctly rounded exact sum unless that sum lies within ~n 2^-106 of a rounding boundary, i.e.
//                    independent of the summation order.  The device path sums this way, so in this mode the oracle's
//                    iterates, and hence its ITERATION COUNT, are comparable bit for bit with a tree-ordered device
//                    sum -- BB steps amplify rounding differences of plain sums into +/-15 % of the count.
// Same algorithm either way; the mode only fixes how the rounding of a sum is defined.
// ---------------------------------------------------------------------------------------------------------------
enum SumMode : int { kSumSerial = 0, kSumCompensated = 1 };
inline int& sum_mode() {
  static int mode = kSumSerial;
  return mode;
}
struct Acc {
  double hi = 0.0, lo = 0.0;
  void add(double b) {
    if (sum_mode() == kSumSerial) {
      hi += b;
      return;
    }
    const double s = hi + b;
    const double bb = s - hi;
    lo += (hi - (s - bb)) + (b - bb);  // hi + b == s + error exactly (Knuth TwoSum)
    hi = s;
  }
  double value() const { return (sum_mode() == kSumSerial || !(hi - hi == 0.0)) ? hi : hi + lo; }
};

// KokkosBackend vector kernels, serial order (convex.hpp:201-284).
inline void axpby(double alpha, const double* x, double beta, double* y, size_t n) {
  const bool az = std::fabs(alpha) < kZeroTol, bz = std::fabs(beta) < kZeroTol;
  if (!az && !bz) {
    for (size_t i = 0; i < n; ++i) y[i] = alpha * x[i] + beta * y[i];
  } else if (az && !bz) {
    for (size_t i = 0; i < n; ++i) y[i] *= beta;
  } else if (!az && bz) {
    for (size_t i = 0; i < n; ++i) y[i] = alpha * x[i];
  } else {
    for (size_t i = 0; i < n; ++i) y[i] = 0.0;
  }
}
inline void wrapped_axpbyz(double alpha, const double* x, double beta, const double* y, double* z, size_t n,
                           const Space& sp) {
  const bool az = std::fabs(alpha) < kZeroTol, bz = std::fabs(beta) < kZeroTol;
  if (!az && !bz) {
    for (size_t i = 0; i < n; ++i) z[i] = sp.project(alpha * x[i] + beta * y[i]);
  } else if (az && !bz) {
    for (size_t i = 0; i < n; ++i) z[i] = sp.project(beta * y[i]);
  } else if (!az && bz) {
    for (size_t i = 0; i < n; ++i) z[i] = sp.project(alpha * x[i]);
  } else {
    for (size_t i = 0; i < n; ++i) z[i] = sp.project(0.0);
  }
}
inline double diff_dot(const double* x, const double* y, size_t n) {
  Acc sum;
  for (size_t i = 0; i < n; ++i) {
    const double diff = x[i] - y[i];
    sum.add(diff * diff);
  }
  return sum.value();
}
inline double diff_dot(const double* x1, const double* x2, const double* y1, const double* y2, size_t n) {
  Acc sum;
  for (size_t i = 0; i < n; ++i) {
    const double xd = x1[i] - x2[i];
    const double yd = y1[i] - y2[i];
    sum.add(xd * yd);
  }
  return sum.value();
}
// convex.hpp:434-496.  Kokkos::Max<double> starts from the lowest finite double.
inline double residual(int kind, const double* x, const double* g, size_t n, const Space& sp) {
  double mx = std::numeric_limits<double>::lowest();
  if (kind == kProjectedGradient) {
    for (size_t i = 0; i < n; ++i) {
      const double v = (x[i] < kZeroTol) ? std::max(0.0, g[i]) : std::fabs(g[i]);
      if (v > mx) mx = v;
    }
    return mx;
  }
  constexpr double small_step_size = 1e-6;
  for (size_t i = 0; i < n; ++i) {
    const double xp = sp.project(x[i] - small_step_size * g[i]);
    const double v = std::fabs(x[i] - xp);
    if (v > mx) mx = v;
  }
  return mx / small_step_size;
}
// convex.hpp:498-516.
inline double bb_step(const double* x_old, const double* g_old, const double* x, const double* g, size_t n) {
  const double num = diff_dot(x, x_old, n);
  double denom = diff_dot(x, x_old, g, g_old, n);
  constexpr double eps = kZeroTol * 10;
  denom += eps * (std::fabs(denom) < eps);
  return num / denom;
}

struct SolveResult {
  unsigned num_iters;
  double residual;
  int converged;
};

// convex.hpp:614-666, 789-797.  Op is any callable  void(const double* x, double* y).
template <class Op>
SolveResult solve_cqpp(const Op& A, const double* q, const Space& sp, int resid_kind, unsigned max_iters, double tol,
                       size_t n, double* x, double* g, double* x_tmp, double* g_tmp) {
  // initialize
  std::copy(x, x + n, x_tmp);
  A(x_tmp, g_tmp);
  axpby(1.0, q, 1.0, g_tmp, n);
  double res = residual(resid_kind, x_tmp, g_tmp, n, sp);
  double step = 1.0 / res;
  unsigned iter = 0;
  bool converged = (res <= tol);
  if (converged) std::copy(g_tmp, g_tmp + n, g);
  // iterate
  while (!(converged || iter >= max_iters)) {
    wrapped_axpbyz(1.0, x_tmp, -step, g_tmp, x, n, sp);
    A(x, g);
    axpby(1.0, q, 1.0, g, n);
    res = residual(resid_kind, x, g, n, sp);
    if (res <= tol) {
      converged = true;
      break;
    }
    step = bb_step(x_tmp, g_tmp, x, g, n);
    std::copy(x, x + n, x_tmp);
    std::copy(g, g + n, g_tmp);
    ++iter;
  }
  return {iter, res, converged ? 1 : 0};
}

// MundyMathBackend<Scalar, N> (convex.hpp:288-350): the same solver on fixed-size Vector/Matrix inside a kernel.
// Differences from KokkosBackend that change bits: dot products and matrix rows are RIGHT folds
// (impl/VectorImpl.hpp:339-344, impl/MatrixImpl.hpp:348-355), axpby / wrapped_axpbyz have no |alpha|,|beta| < 1e-15
// branches (:314-322), reduce_max starts from -infinity (:346).
inline double rfold_dot(const double* a, const double* b, size_t n) {
  double acc = a[n - 1] * b[n - 1];
  for (size_t i = n - 1; i-- > 0;) acc = a[i] * b[i] + acc;
  return acc;
}
inline SolveResult solve_cqpp_small(size_t n, const double* A, const double* q, const Space& sp, int resid_kind,
                                    unsigned max_iters, double tol, double* x, double* g) {
  std::vector<double> xt(n), gt(n), d1(n), d2(n);
  auto apply = [&](const double* in, double* out) {
    for (size_t i = 0; i < n; ++i) out[i] = rfold_dot(A + i * n, in, n);
  };
  auto resid = [&](const double* xx, const double* gg) {
    double mx = -std::numeric_limits<double>::infinity();
    for (size_t i = 0; i < n; ++i) {
      double v;
      if (resid_kind == kProjectedGradient)
        v = (xx[i] < kZeroTol) ? std::max(0.0, gg[i]) : std::fabs(gg[i]);
      else
        v = std::fabs(xx[i] - sp.project(xx[i] - 1e-6 * gg[i]));
      if (v > mx) mx = v;
    }
    return resid_kind == kProjectedGradient ? mx : mx / 1e-6;
  };
  for (size_t i = 0; i < n; ++i) xt[i] = x[i];
  apply(xt.data(), gt.data());
  for (size_t i = 0; i < n; ++i) gt[i] = 1.0 * q[i] + 1.0 * gt[i];
  double res = resid(xt.data(), gt.data());
  double step = 1.0 / res;
  unsigned iter = 0;
  bool converged = res <= tol;
  if (converged)
    for (size_t i = 0; i < n; ++i) g[i] = gt[i];
  while (!(converged || iter >= max_iters)) {
    for (size_t i = 0; i < n; ++i) x[i] = sp.project(1.0 * xt[i] + (-step) * gt[i]);
    apply(x, g);
    for (size_t i = 0; i < n; ++i) g[i] = 1.0 * q[i] + 1.0 * g[i];
    res = resid(x, g);
    if (res <= tol) {
      converged = true;
      break;
    }
    for (size_t i = 0; i < n; ++i) {
      d1[i] = x[i] - xt[i];
      d2[i] = g[i] - gt[i];
    }
    const double num = rfold_dot(d1.data(), d1.data(), n);
    double den = rfold_dot(d1.data(), d2.data(), n);
    constexpr double eps = kZeroTol * 10;
    den += eps * (std::fabs(den) < eps);
    step = num / den;
    for (size_t i = 0; i < n; ++i) {
      xt[i] = x[i];
      gt[i] = g[i];
    }
    ++iter;
  }
  return {iter, res, converged ? 1 : 0};
}

// Dense operator, row-major n x n (KokkosBlas::gemv "N", convex.hpp:168-174).
struct DenseOp {
  const double* A;
  size_t n;
  void operator()(const double* x, double* y) const {
    for (size_t i = 0; i < n; ++i) {
      Acc acc;
      for (size_t j = 0; j < n; ++j) acc.add(A[i * n + j] * x[j]);
      y[i] = acc.value();
    }
  }
};

// Matrix-free contact operator  y = dt * D^T M D x  (scrap/lcp_spheres/NgpLcp.cpp:442-548: K21 scatter, K22 dry
// mobility, K23 gather).  Rigid bodies: force f and torque tq per body, diagonal mobilities mt (translation) and
// mr (rotation; 0 for the translation-only sphere app).  Lever arms ra, rb = contact point - body centre; pass
// nullptr for spheres (no torque).  Scrap sign conventions: F_src += -lam*n, F_tgt += +lam*n (:467-472);
// sdot = -n.(U_src - U_tgt) (:526-528).  Body sums are taken in constraint order (Kokkos-Serial order).
struct ContactOp {
  const int32_t* pairs;  // [C][2]
  const double* normal;  // [C][3]
  const double* ra;      // [C][3] or null
  const double* rb;      // [C][3] or null
  const double* mt;      // [N]
  const double* mr;      // [N] or null
  double dt;
  size_t C, N;
  mutable std::vector<double> F, T, U, W;
  mutable std::vector<Acc> Fa, Ta;
  void operator()(const double* x, double* y) const {
    Fa.assign(3 * N, Acc{});
    F.resize(3 * N);
    U.resize(3 * N);
    const bool rot = (ra && rb && mr);
    if (rot) {
      Ta.assign(3 * N, Acc{});
      T.resize(3 * N);
      W.resize(3 * N);
    }
    for (size_t c = 0; c < C; ++c) {
      const int32_t i = pairs[2 * c], j = pairs[2 * c + 1];
      const double lam = x[c];
      const V3 n{normal[3 * c], normal[3 * c + 1], normal[3 * c + 2]};
      const V3 f{lam * n.x, lam * n.y, lam * n.z};
      for (int k = 0; k < 3; ++k) {
        Fa[3 * i + k].add(-f[k]);
        Fa[3 * j + k].add(f[k]);
      }
      if (rot) {
        const V3 a{ra[3 * c], ra[3 * c + 1], ra[3 * c + 2]}, b{rb[3 * c], rb[3 * c + 1], rb[3 * c + 2]};
        const V3 ta = cross(a, f), tb = cross(b, f);
        for (int k = 0; k < 3; ++k) {
          Ta[3 * i + k].add(-ta[k]);
          Ta[3 * j + k].add(tb[k]);
        }
      }
    }
    for (size_t b = 0; b < N; ++b)
      for (int k = 0; k < 3; ++k) {
        F[3 * b + k] = Fa[3 * b + k].value();
        U[3 * b + k] = mt[b] * F[3 * b + k];
        if (rot) {
          T[3 * b + k] = Ta[3 * b + k].value();
          W[3 * b + k] = mr[b] * T[3 * b + k];
        }
      }
    for (size_t c = 0; c < C; ++c) {
      const int32_t i = pairs[2 * c], j = pairs[2 * c + 1];
      const V3 n{normal[3 * c], normal[3 * c + 1], normal[3 * c + 2]};
      V3 vi{U[3 * i], U[3 * i + 1], U[3 * i + 2]}, vj{U[3 * j], U[3 * j + 1], U[3 * j + 2]};
      if (rot) {
        const V3 a{ra[3 * c], ra[3 * c + 1], ra[3 * c + 2]}, b{rb[3 * c], rb[3 * c + 1], rb[3 * c + 2]};
        const V3 wi{W[3 * i], W[3 * i + 1], W[3 * i + 2]}, wj{W[3 * j], W[3 * j + 1], W[3 * j + 2]};
        vi = vi + cross(wi, a);
        vj = vj + cross(wj, b);
      }
      const double sdot = -n.x * (vi.x - vj.x) - n.y * (vi.y - vj.y) - n.z * (vi.z - vj.z);
      y[c] = dt * sdot;
    }
  }
};

// The same operator for spherocylinders with the lever arms written through the rod axis: a rod's contact point lies
// on its centreline, cp = c + (s - 1/2) u with u = p1 - p0 (endpoints as compute_aabb.hpp:115-117 places them), so
//   r x f = (s - 1/2) (u x f)   ->   T_b = u_b x S_b,  S_b = sum (s - 1/2) f      (one cross product per body)
//   W x r = (s - 1/2) (W x u)   ->   v_contact = U_b + (s - 1/2) Z_b,  Z_b = W_b x u_b
// Algebraically ContactOp with ra = (s - 1/2) u_i, rb = (t - 1/2) u_j; the association differs, so the two agree to
// rounding (one ulp of the arm), not bitwise.  This is the association the device path evaluates (its half-edge
// records hold (n, s - 1/2)); with kSumCompensated the two produce the same bits.  Like the 6-DOF form of ContactOp it
// has no counterpart in the reference (its LCP app is spheres only): parity unpinned.
//
// s, t are the arclengths OF THE CONTACT POINTS.  distance(Point, LineSegment) clamps the closest point but leaves its
// parameter unclamped (PointLineSegment.hpp:156-166) and the colinear branch of segment-segment hands that parameter
// back (LineSegmentLineSegment.hpp:236-265), while the contact points of the assembly are the clamped closest points
// (SpherocylinderSpherocylinderLinker.cpp:246-247): the arm coefficient is clamp(s, 0, 1) - 1/2, so that this form
// equals ContactOp with ra = cp1 - c_i, rb = cp2 - c_j on aligned rods too (round-2 review: the unclamped parameter
// put the contact point of an end-to-end pair off the rod).
inline double contact_arclength(double t) { return t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t); }
struct ContactOpRod {
  const int32_t* pairs;   // [C][2]
  const double* normal;   // [C][3]
  const double* arc_s;    // [C] arclength parameter of the closest point on the first rod, in [0, 1]
  const double* arc_t;    // [C] ... on the second rod
  const double* seg;      // [N][8] segment records (p0, p1, radius, -)
  const double* mt;       // [N]
  const double* mr;       // [N]
  double dt;
  size_t C, N;
  mutable std::vector<double> U, Z, Wv;
  mutable std::vector<Acc> Fa, Sa;
  void operator()(const double* x, double* y) const {
    Fa.assign(3 * N, Acc{});
    Sa.assign(3 * N, Acc{});
    U.resize(3 * N);
    Z.resize(3 * N);
    Wv.resize(3 * N);
    for (size_t c = 0; c < C; ++c) {
      const int32_t i = pairs[2 * c], j = pairs[2 * c + 1];
      const double lam = x[c];
      const V3 n{normal[3 * c], normal[3 * c + 1], normal[3 * c + 2]};
      const V3 f{lam * n.x, lam * n.y, lam * n.z};
      const double ci = contact_arclength(arc_s[c]) - 0.5, cj = contact_arclength(arc_t[c]) - 0.5;
      for (int k = 0; k < 3; ++k) {
        Fa[3 * i + k].add(-f[k]);
        Fa[3 * j + k].add(f[k]);
        Sa[3 * i + k].add(ci * (-f[k]));
        Sa[3 * j + k].add(cj * f[k]);
      }
    }
    for (size_t b = 0; b < N; ++b) {
      const double* r = seg + 8 * b;
      const V3 u{r[3] - r[0], r[4] - r[1], r[5] - r[2]};
      const V3 Fb{Fa[3 * b].value(), Fa[3 * b + 1].value(), Fa[3 * b + 2].value()};
      const V3 Sb{Sa[3 * b].value(), Sa[3 * b + 1].value(), Sa[3 * b + 2].value()};
      const V3 tq = cross(u, Sb);
      const V3 w{mr[b] * tq.x, mr[b] * tq.y, mr[b] * tq.z};
      const V3 z = cross(w, u);
      for (int k = 0; k < 3; ++k) {
        U[3 * b + k] = mt[b] * Fb[k];
        Wv[3 * b + k] = w[k];
        Z[3 * b + k] = z[k];
      }
    }
    for (size_t c = 0; c < C; ++c) {
      const int32_t i = pairs[2 * c], j = pairs[2 * c + 1];
      const V3 n{normal[3 * c], normal[3 * c + 1], normal[3 * c + 2]};
      const double ci = contact_arclength(arc_s[c]) - 0.5, cj = contact_arclength(arc_t[c]) - 0.5;
      const V3 zi{Z[3 * i], Z[3 * i + 1], Z[3 * i + 2]}, zj{Z[3 * j], Z[3 * j + 1], Z[3 * j + 2]};
      const V3 vi = V3{U[3 * i], U[3 * i + 1], U[3 * i + 2]} + zi * ci;
      const V3 vj = V3{U[3 * j], U[3 * j + 1], U[3 * j + 2]} + zj * cj;
      const double sdot = -n.x * (vi.x - vj.x) - n.y * (vi.y - vj.y) - n.z * (vi.z - vj.z);
      y[c] = dt * sdot;
    }
  }
};

// ---------------------------------------------------------------------------------------------------------------
// BUILD EXTENSION (no reference implementation; SURVEY F2: the reference has no frictional solver, so everything in
// this block is "parity unpinned" -- it checks the GPU extension against an independent serial statement of the SAME
// algorithm, not against MuNDy).  Coulomb friction as a cone complementarity problem solved by the reference's BBPGD
// iteration (convex.hpp:614-666) with a per-contact cone projection:
//   p_c in R^3 (world-frame impulse), K_c = { |p - (p.n) n| <= mu (p.n) }, forces -p / +p at the contact points,
//   g_c = dt [(U_j + W_j x rb) - (U_i + W_i x ra)] + sep_c n_c.
// ---------------------------------------------------------------------------------------------------------------
inline V3 project_cone(const V3& v, const V3& n, double mu) {
  const double a = dot(v, n);
  const V3 b = v - n * a;
  const double bn = norm(b);
  if (a >= 0.0 && bn <= mu * a) return v;  // a >= 0 matters only for mu = 0 (v anti-parallel to n: bn = 0 = mu a)
  if (mu * bn <= -a) return V3{0.0, 0.0, 0.0};
  const double an = (mu * bn + a) / (mu * mu + 1.0);
  if (!(bn > 0.0)) return n * an;
  return n * an + b * ((mu * an) / bn);
}
struct FrictionOp {
  const int32_t* pairs;
  const double *normal, *ra, *rb, *mt, *mr, *sep;
  double dt;
  size_t C, N;
  mutable std::vector<double> F, T;
  mutable std::vector<Acc> Fa, Ta;
  // g = A p + q, serial scatter / mobility / gather
  void gradient(const double* p, double* g) const {
    Fa.assign(3 * N, Acc{});
    Ta.assign(3 * N, Acc{});
    F.resize(3 * N);
    T.resize(3 * N);
    for (size_t c = 0; c < C; ++c) {
      const int32_t i = pairs[2 * c], j = pairs[2 * c + 1];
      const V3 f{p[3 * c], p[3 * c + 1], p[3 * c + 2]};
      const V3 a{ra[3 * c], ra[3 * c + 1], ra[3 * c + 2]}, b{rb[3 * c], rb[3 * c + 1], rb[3 * c + 2]};
      const V3 ta = cross(a, f), tb = cross(b, f);
      for (int k = 0; k < 3; ++k) {
        Fa[3 * i + k].add(-f[k]);
        Fa[3 * j + k].add(f[k]);
        Ta[3 * i + k].add(-ta[k]);
        Ta[3 * j + k].add(tb[k]);
      }
    }
    for (size_t k = 0; k < 3 * N; ++k) {
      F[k] = Fa[k].value();
      T[k] = Ta[k].value();
    }
    for (size_t c = 0; c < C; ++c) {
      const int32_t i = pairs[2 * c], j = pairs[2 * c + 1];
      const V3 n{normal[3 * c], normal[3 * c + 1], normal[3 * c + 2]};
      const V3 a{ra[3 * c], ra[3 * c + 1], ra[3 * c + 2]}, b{rb[3 * c], rb[3 * c + 1], rb[3 * c + 2]};
      const V3 ui{mt[i] * F[3 * i], mt[i] * F[3 * i + 1], mt[i] * F[3 * i + 2]};
      const V3 uj{mt[j] * F[3 * j], mt[j] * F[3 * j + 1], mt[j] * F[3 * j + 2]};
      const V3 wi{mr[i] * T[3 * i], mr[i] * T[3 * i + 1], mr[i] * T[3 * i + 2]};
      const V3 wj{mr[j] * T[3 * j], mr[j] * T[3 * j + 1], mr[j] * T[3 * j + 2]};
      const V3 vi = ui + cross(wi, a), vj = uj + cross(wj, b);
      for (int k = 0; k < 3; ++k) g[3 * c + k] = dt * (vj[k] - vi[k]) + sep[c] * n[k];
    }
  }
};
inline SolveResult solve_friction_contact(const FrictionOp& op, double mu, unsigned max_iters, double tol, double* p,
                                          double* g) {
  const size_t C = op.C;
  std::vector<double> pt(p, p + 3 * C), gt(3 * C);
  auto nrm = [&](size_t c) { return V3{op.normal[3 * c], op.normal[3 * c + 1], op.normal[3 * c + 2]}; };
  auto residual = [&](const double* x, const double* gr) {
    double r = -1.7976931348623157e308;
    for (size_t c = 0; c < C; ++c) {
      const V3 pc{x[3 * c], x[3 * c + 1], x[3 * c + 2]}, gc{gr[3 * c], gr[3 * c + 1], gr[3 * c + 2]};
      const V3 w = project_cone(pc - gc * 1e-6, nrm(c), mu);
      for (int k = 0; k < 3; ++k) r = std::max(r, std::fabs(pc[k] - w[k]));
    }
    return r / 1e-6;
  };
  op.gradient(pt.data(), gt.data());
  SolveResult out{0, residual(pt.data(), gt.data()), false};
  auto finish = [&](const std::vector<double>& x, const std::vector<double>& gr) {
    std::copy(x.begin(), x.end(), p);
    std::copy(gr.begin(), gr.end(), g);
  };
  if (out.residual <= tol) {
    out.converged = true;
    finish(pt, gt);
    return out;
  }
  double step = 1.0 / out.residual;
  std::vector<double> pn(3 * C), gn(3 * C);
  while (out.num_iters < max_iters) {
    for (size_t c = 0; c < C; ++c) {
      V3 v;
      for (int k = 0; k < 3; ++k) v[k] = (std::fabs(-step) < kZeroTol) ? pt[3 * c + k] : pt[3 * c + k] + (-step) * gt[3 * c + k];
      const V3 w = project_cone(v, nrm(c), mu);
      for (int k = 0; k < 3; ++k) pn[3 * c + k] = w[k];
    }
    op.gradient(pn.data(), gn.data());
    out.residual = residual(pn.data(), gn.data());
    if (out.residual <= tol) {
      out.converged = true;
      finish(pn, gn);
      return out;
    }
    Acc numa, dena;  // one term per contact: the right-fold dot products of its three components
    for (size_t c = 0; c < C; ++c) {
      const V3 dp{pn[3 * c] - pt[3 * c], pn[3 * c + 1] - pt[3 * c + 1], pn[3 * c + 2] - pt[3 * c + 2]};
      const V3 dg{gn[3 * c] - gt[3 * c], gn[3 * c + 1] - gt[3 * c + 1], gn[3 * c + 2] - gt[3 * c + 2]};
      numa.add(dot(dp, dp));
      dena.add(dot(dp, dg));
    }
    const double num = numa.value();
    double den = dena.value();
    den += 1e-14 * (std::fabs(den) < 1e-14 ? 1.0 : 0.0);
    step = num / den;
    pt.swap(pn);
    gt.swap(gn);
    ++out.num_iters;
  }
  finish(pt, gt);
  return out;
}

// BUILD EXTENSION (parity unpinned): the same cone complementarity problem by APGD (Mazhar, Heyn, Negrut, Tasora 2015),
// as the device runs it (csrc/convex.hip, k_constraint_friction_apgd): one operator application per sweep,
//   y = p_k + beta (p_k - p_{k-1}),  g_y = g_k + beta (g_k - g_{k-1}),  p+ = Proj_K(y - g_y / L),  g+ = N p+ + q,
//   accepted iff (p+ - y).(g+ - g_y) <= L |p+ - y|^2 (else L <- 2 L and again), converged iff the projected-difference
//   residual of (p+, g+) <= tol, theta / beta by Nesterov's recursion, restart iff g_y.(p+ - p_k) > 0, L <- 0.9 L,
//   L_0 = the initial residual.  num_iters counts every sweep.
inline SolveResult solve_friction_contact_apgd(const FrictionOp& op, double mu, unsigned max_iters, double tol, double* p,
                                               double* g) {
  const size_t C = op.C;
  auto nrm = [&](size_t c) { return V3{op.normal[3 * c], op.normal[3 * c + 1], op.normal[3 * c + 2]}; };
  auto residual = [&](const double* x, const double* gr) {
    double r = -1.7976931348623157e308;
    for (size_t c = 0; c < C; ++c) {
      const V3 pc{x[3 * c], x[3 * c + 1], x[3 * c + 2]}, gc{gr[3 * c], gr[3 * c + 1], gr[3 * c + 2]};
      const V3 w = project_cone(pc - gc * 1e-6, nrm(c), mu);
      for (int k = 0; k < 3; ++k) r = std::max(r, std::fabs(pc[k] - w[k]));
    }
    return r / 1e-6;
  };
  std::vector<double> P[3], G[3];
  for (int b = 0; b < 3; ++b) {
    P[b].assign(3 * C, 0.0);
    G[b].assign(3 * C, 0.0);
  }
  std::copy(p, p + 3 * C, P[0].begin());
  op.gradient(P[0].data(), G[0].data());
  SolveResult out{0, residual(P[0].data(), G[0].data()), false};
  auto finish = [&](int b) {
    std::copy(P[b].begin(), P[b].end(), p);
    std::copy(G[b].begin(), G[b].end(), g);
  };
  if (out.residual <= tol || max_iters == 0) {
    out.converged = out.residual <= tol;
    finish(0);
    return out;
  }
  double L = out.residual, theta = 1.0, beta = 0.0;
  if (!(L > 0.0)) L = 1.0;
  int cur = 0, prev = 0, nxt = 1;
  std::vector<double> y(3 * C), gy(3 * C);
  while (out.num_iters < max_iters) {
    const double t = 1.0 / L;
    for (size_t c = 0; c < C; ++c) {
      V3 v;
      for (int k = 0; k < 3; ++k) {
        const size_t q = 3 * c + k;
        y[q] = P[cur][q] + beta * (P[cur][q] - P[prev][q]);
        gy[q] = G[cur][q] + beta * (G[cur][q] - G[prev][q]);
        v[k] = y[q] + (-t) * gy[q];
      }
      const V3 w = project_cone(v, nrm(c), mu);
      for (int k = 0; k < 3; ++k) P[nxt][3 * c + k] = w[k];
    }
    op.gradient(P[nxt].data(), G[nxt].data());
    ++out.num_iters;
    Acc sa, sb, sr;
    for (size_t c = 0; c < C; ++c) {
      V3 d, dg, gyc, dk;
      for (int k = 0; k < 3; ++k) {
        const size_t q = 3 * c + k;
        d[k] = P[nxt][q] - y[q];
        dg[k] = G[nxt][q] - gy[q];
        gyc[k] = gy[q];
        dk[k] = P[nxt][q] - P[cur][q];
      }
      sa.add(dot(d, dg));
      sb.add(dot(d, d));
      sr.add(dot(gyc, dk));
    }
    if (sa.value() > L * sb.value()) {
      L *= 2.0;
      continue;
    }
    out.residual = residual(P[nxt].data(), G[nxt].data());
    if (out.residual <= tol) {
      out.converged = true;
      finish(nxt);
      return out;
    }
    double th1 = (-(theta * theta) + theta * std::sqrt(theta * theta + 4.0)) / 2.0;
    beta = theta * (1.0 - theta) / (theta * theta + th1);
    if (sr.value() > 0.0) {
      beta = 0.0;
      th1 = 1.0;
    }
    theta = th1;
    L *= 0.9;
    const int old_cur = cur;
    prev = old_cur;
    cur = nxt;
    nxt = 3 - old_cur - cur;
  }
  finish(cur);
  return out;
}

// The scrap app's own matrix-free BBPGD (scrap/lcp_spheres/NgpLcp.cpp:558-759, DRY mobility), serial order.
// Differs from convex.hpp's PGDStrategy: Dai-Fletcher residual with a 1e-12 active-set test (:376-405), strict `<`
// convergence test, BB1/BB2 alternating by the parity of ite_count with `|b| < 1e-12 -> b += 1e-12` (:716-731),
// ite_count counts started iterations, and the first projected step uses signed_sep_dot (still zero) rather than
// signed_sep_dot_tmp (:639) -- all reproduced as written.
struct ScrapResult {
  double max_abs_projected_sep;
  int ite_count;
  double max_speed;
};
template <class Op>  // ContactOp or ContactOpRod
ScrapResult scrap_resolve_collisions(const Op& A, const double* sep, double max_allowable_overlap,
                                            int max_col_iterations, double* lam, double* lam_tmp, double* sep_dot_dt,
                                            double* sep_dot_dt_tmp) {
  // sep_dot_dt holds dt * signed_sep_dot (the operator returns dt * sdot); the scrap code keeps sdot and multiplies by
  // dt at each use: sep_new = sep + dt*sdot, gkdiff = dt*(sdot - sdot_tmp).  Same products, formed once here.
  const size_t C = A.C;
  int ite_count = 0;
  std::copy(lam, lam + C, lam_tmp);
  std::fill(sep_dot_dt, sep_dot_dt + C, 0.0);
  std::fill(sep_dot_dt_tmp, sep_dot_dt_tmp + C, 0.0);
  A(lam_tmp, sep_dot_dt_tmp);
  auto residual = [&](const double* x, const double* gdt) {
    double mx = std::numeric_limits<double>::lowest();
    for (size_t i = 0; i < C; ++i) {
      const double sep_new = sep[i] + gdt[i];
      const double v = (x[i] < 1e-12) ? std::fabs(std::min(sep_new, 0.0)) : std::fabs(sep_new);
      if (v > mx) mx = v;
    }
    return mx;
  };
  double res = residual(lam_tmp, sep_dot_dt_tmp);
  if (!(res < max_allowable_overlap)) {
    double alpha = 1.0 / res;
    while (ite_count < max_col_iterations) {
      ++ite_count;
      for (size_t i = 0; i < C; ++i) lam[i] = std::max(lam_tmp[i] - alpha * (sep[i] + sep_dot_dt[i]), 0.0);
      A(lam, sep_dot_dt);
      res = residual(lam, sep_dot_dt);
      if (res < max_allowable_overlap) break;
      Acc sxx, sxg, sgg;
      for (size_t i = 0; i < C; ++i) {
        const double xd = lam[i] - lam_tmp[i];
        const double gd = sep_dot_dt[i] - sep_dot_dt_tmp[i];
        sxx.add(xd * xd);
        sxg.add(xd * gd);
        sgg.add(gd * gd);
      }
      const double xx = sxx.value(), xg = sxg.value(), gg = sgg.value();
      double a, b;
      if (ite_count % 2 == 0) {
        a = xx; b = xg;
      } else {
        a = xg; b = gg;
      }
      if (std::fabs(b) < 1e-12) b += 1e-12;
      alpha = a / b;
      std::copy(lam, lam + C, lam_tmp);
      std::copy(sep_dot_dt, sep_dot_dt + C, sep_dot_dt_tmp);
    }
  }
  double max_speed = 0.0;
  for (size_t b = 0; b < A.N; ++b) {
    const double v = std::sqrt(A.U[3 * b] * A.U[3 * b] + A.U[3 * b + 1] * A.U[3 * b + 1] + A.U[3 * b + 2] * A.U[3 * b + 2]);
    if (v > max_speed) max_speed = v;
  }
  return {res, ite_count, max_speed};
}

// ---------------------------------------------------------------------------------------------------------------
// Neighbour search oracle.  The reference's search is stk::search::coarse_search (Trilinos 16.0.0, absent) at
// mundy/mesh/src/mundy_mesh/GenNeighborLinkers.hpp:658 -- PARITY UNPINNED; predicate defined from MuNDy's own code:
//   kSearchSpheres: bounding spheres (c, R + buffer) as built at GenNeighborLinkers.hpp:579-583, closed test
//                   |c_j - c_i|^2 <= ((R_i + buffer) + (R_j + buffer))^2
//   kSearchAABB:    compute_aabb(body) grown by buffer on every face, geom::intersects (AABB.hpp:420-431, closed).
// Output: pairs (i, j) sorted by (i, j); unique i<j (scrap/lcp_spheres/NgpLcp.cpp:287-295) or symmetric i!=j
// (GenNeighborLinkers.hpp:655 + ExcludeSelfInteractions :185-200).
// Periodic boxes use the minimum image of the centre separation (PeriodicScaledMetric::sep).
// ---------------------------------------------------------------------------------------------------------------
enum SearchKind : int { kSearchSpheres = 0, kSearchAABB = 1 };

template <class Metric>  // PeriodicScaledMetric (orthorhombic box) or PeriodicMetric (triclinic cell): both have sep()
inline bool search_overlap(int kind, const double* lo_i, const double* hi_i, const double* lo_j, const double* hi_j,
                           const V3& ci, double Ri, const V3& cj, double Rj, const Metric* pm) {
  if (kind == kSearchSpheres) {
    const V3 s = pm ? pm->sep(ci, cj) : (cj - ci);
    const double d2 = dot(s, s);
    const double rs = Ri + Rj;
    return d2 <= rs * rs;
  }
  if (!pm) {
    AABB a, b;
    for (int k = 0; k < 3; ++k) {
      a.lo[k] = lo_i[k]; a.hi[k] = hi_i[k]; b.lo[k] = lo_j[k]; b.hi[k] = hi_j[k];
    }
    return intersects(a, b);
  }
  // periodic AABB test: shift box j by the lattice image that brings its centre closest to box i's centre
  // (centres = box midpoints), then the closed interval test.
  V3 mi, mj;
  for (int k = 0; k < 3; ++k) {
    mi[k] = 0.5 * (lo_i[k] + hi_i[k]);
    mj[k] = 0.5 * (lo_j[k] + hi_j[k]);
  }
  const V3 s = pm->sep(mi, mj);
  for (int k = 0; k < 3; ++k) {
    const double shift = (mi[k] + s[k]) - mj[k];
    const double blo = lo_j[k] + shift, bhi = hi_j[k] + shift;
    if (hi_i[k] < blo || bhi < lo_i[k]) return false;
  }
  return true;
}

// O(N^2) reference search over grown boxes [lo, hi] ([N][3] each), centres c ([N][3]) and grown radii R ([N]).
inline void search_bruteforce(int kind, size_t n, const double* lo, const double* hi, const double* c,
                              const double* R, const double* box /*null or [3]*/, bool symmetric,
                              std::vector<int32_t>& pairs) {
  pairs.clear();
  PeriodicScaledMetric pmv(box ? V3{box[0], box[1], box[2]} : V3{1, 1, 1});
  const PeriodicScaledMetric* pm = box ? &pmv : nullptr;
  for (size_t i = 0; i < n; ++i) {
    for (size_t j = symmetric ? 0 : i + 1; j < n; ++j) {
      if (j == i) continue;
      // the predicate is always evaluated with the lower index first, so (i,j) and (j,i) agree bit for bit
      const size_t a = std::min(i, j), b = std::max(i, j);
      const V3 ca{c[3 * a], c[3 * a + 1], c[3 * a + 2]}, cb{c[3 * b], c[3 * b + 1], c[3 * b + 2]};
      if (search_overlap(kind, lo + 3 * a, hi + 3 * a, lo + 3 * b, hi + 3 * b, ca, R[a], cb, R[b], pm)) {
        pairs.push_back(static_cast<int32_t>(i));
        pairs.push_back(static_cast<int32_t>(j));
      }
    }
  }
}

// The same in a TRICLINIC cell: the predicate with PeriodicMetric::sep (periodicity.hpp:304-307: minimum
// image of the fractional coordinates) in the place of PeriodicScaledMetric::sep.  Brute force only: it is the checker.
inline void search_bruteforce_triclinic(int kind, size_t n, const double* lo, const double* hi, const double* c,
                                        const double* R, const double* cell /*[9]*/, bool symmetric,
                                        std::vector<int32_t>& pairs) {
  pairs.clear();
  const PeriodicMetric pm(cell);
  for (size_t i = 0; i < n; ++i) {
    for (size_t j = symmetric ? 0 : i + 1; j < n; ++j) {
      if (j == i) continue;
      const size_t a = std::min(i, j), b = std::max(i, j);
      const V3 ca{c[3 * a], c[3 * a + 1], c[3 * a + 2]}, cb{c[3 * b], c[3 * b + 1], c[3 * b + 2]};
      if (search_overlap(kind, lo + 3 * a, hi + 3 * a, lo + 3 * b, hi + 3 * b, ca, R[a], cb, R[b], &pm)) {
        pairs.push_back(static_cast<int32_t>(i));
        pairs.push_back(static_cast<int32_t>(j));
      }
    }
  }
}

// Cell-list search with the same predicate (evaluated with the lower index first, so it is orientation
// independent and identical to the brute-force result).  This is also the CPU-baseline broad phase.
inline void search_celllist(int kind, size_t n, const double* lo, const double* hi, const double* c, const double* R,
                            const double* box, bool symmetric, std::vector<int32_t>& pairs) {
  pairs.clear();
  if (n == 0) return;
  PeriodicScaledMetric pmv(box ? V3{box[0], box[1], box[2]} : V3{1, 1, 1});
  const PeriodicScaledMetric* pm = box ? &pmv : nullptr;
  // binning point = box midpoint (AABB mode) or centre (sphere mode); reach = max half extent / radius
  std::vector<double> p(3 * n);
  double reach = 0;
  for (size_t i = 0; i < n; ++i)
    for (int k = 0; k < 3; ++k) {
      if (kind == kSearchAABB) {
        p[3 * i + k] = 0.5 * (lo[3 * i + k] + hi[3 * i + k]);
        reach = std::max(reach, 0.5 * (hi[3 * i + k] - lo[3 * i + k]));
      } else {
        p[3 * i + k] = c[3 * i + k];
        reach = std::max(reach, R[i]);
      }
    }
  double h = 2.0 * reach * (1.0 + 1e-12);
  if (!(h > 0)) h = 1.0;
  double origin[3], ext[3];
  int nc[3];
  for (int k = 0; k < 3; ++k) {
    if (pm) {
      origin[k] = 0.0;
      ext[k] = box[k];
    } else {
      double mn = p[k], mx = p[k];
      for (size_t i = 1; i < n; ++i) {
        mn = std::min(mn, p[3 * i + k]);
        mx = std::max(mx, p[3 * i + k]);
      }
      origin[k] = mn;
      ext[k] = mx - mn;
    }
    nc[k] = std::max(1, std::min(1024, static_cast<int>(std::floor(ext[k] / h))));
  }
  if (pm)
    for (size_t i = 0; i < n; ++i) {
      const V3 w = pm->wrap({p[3 * i], p[3 * i + 1], p[3 * i + 2]});
      p[3 * i] = w.x; p[3 * i + 1] = w.y; p[3 * i + 2] = w.z;
    }
  auto cell_of = [&](size_t i, int k) {
    if (ext[k] <= 0) return 0;
    const int ci = static_cast<int>(std::floor((p[3 * i + k] - origin[k]) / ext[k] * nc[k]));
    return std::max(0, std::min(nc[k] - 1, ci));
  };
  const size_t ncell = static_cast<size_t>(nc[0]) * nc[1] * nc[2];
  std::vector<int32_t> head(ncell + 1, 0), cellid(n), order(n);
  for (size_t i = 0; i < n; ++i) {
    cellid[i] = (cell_of(i, 2) * nc[1] + cell_of(i, 1)) * nc[0] + cell_of(i, 0);
    head[cellid[i] + 1]++;
  }
  for (size_t k = 0; k < ncell; ++k) head[k + 1] += head[k];
  {
    std::vector<int32_t> cur(head.begin(), head.end() - 1);
    for (size_t i = 0; i < n; ++i) order[cur[cellid[i]]++] = static_cast<int32_t>(i);
  }
  // rows of consecutive bodies in chunks, the chunks on the OpenMP threads (the CPU baseline runs this on all host
  // cores); every chunk keeps its own pair list and the lists are joined in chunk order: the same list as a serial walk
  constexpr size_t kChunk = 2048;
  const size_t nchunks = (n + kChunk - 1) / kChunk;
  std::vector<std::vector<int32_t>> chunk_pairs(nchunks);
#pragma omp parallel for schedule(dynamic, 1)
  for (size_t ch = 0; ch < nchunks; ++ch) {
  std::vector<int32_t> row;
  std::vector<int32_t>& pairs = chunk_pairs[ch];  // (shadows the output list inside the chunk)
  for (size_t i = ch * kChunk; i < std::min(n, (ch + 1) * kChunk); ++i) {
    row.clear();
    const int cx = cellid[i] % nc[0], cy = (cellid[i] / nc[0]) % nc[1], cz = cellid[i] / (nc[0] * nc[1]);
    int seen[27];
    int nseen = 0;
    for (int dz = -1; dz <= 1; ++dz)
      for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
          int x = cx + dx, y = cy + dy, z = cz + dz;
          if (pm) {
            x = (x + nc[0]) % nc[0]; y = (y + nc[1]) % nc[1]; z = (z + nc[2]) % nc[2];
          } else if (x < 0 || y < 0 || z < 0 || x >= nc[0] || y >= nc[1] || z >= nc[2]) {
            continue;
          }
          const int cid = (z * nc[1] + y) * nc[0] + x;
          bool dup = false;
          for (int q = 0; q < nseen; ++q) dup |= (seen[q] == cid);
          if (dup) continue;
          seen[nseen++] = cid;
          for (int32_t s = head[cid]; s < head[cid + 1]; ++s) {
            const size_t j = order[s];
            if (j == i || (!symmetric && j < i)) continue;
            const size_t a = std::min(i, j), b2 = std::max(i, j);
            const V3 ca{c[3 * a], c[3 * a + 1], c[3 * a + 2]}, cb{c[3 * b2], c[3 * b2 + 1], c[3 * b2 + 2]};
            if (search_overlap(kind, lo + 3 * a, hi + 3 * a, lo + 3 * b2, hi + 3 * b2, ca, R[a], cb, R[b2], pm))
              row.push_back(static_cast<int32_t>(j));
          }
        }
    std::sort(row.begin(), row.end());
    for (int32_t j : row) {
      pairs.push_back(static_cast<int32_t>(i));
      pairs.push_back(j);
    }
  }
  }
  size_t total = 0;
  for (const auto& cp : chunk_pairs) total += cp.size();
  pairs.reserve(total);
  for (const auto& cp : chunk_pairs) pairs.insert(pairs.end(), cp.begin(), cp.end());
}

// ---------------------------------------------------------------------------------------------------------------
// zorder_knn / zmorton_less (mundy/math/src/mundy_math/zmort.hpp)
// ---------------------------------------------------------------------------------------------------------------
// zmort.hpp:95-107.
inline int float_exp(uint64_t xi) {
  uint64_t uxi = xi & 0x7fffffffffffffffull;
  if (uxi == 0 || uxi >= 0x7ff0000000000000ull) return 0;
  uxi >>= 52;
  return (uxi == 0) ? -1022 : static_cast<int>(uxi) - 1023;
}
inline int float_exp(uint32_t xi) {
  uint32_t uxi = xi & 0x7fffffffu;
  if (uxi == 0 || uxi >= 0x7f800000u) return 0;
  uxi >>= 23;
  return (uxi == 0) ? -126 : static_cast<int>(uxi) - 127;
}
// zmort.hpp:109-115.
inline uint64_t float_sig(uint64_t xi) { return xi & 0x000fffffffffffffull; }
inline uint32_t float_sig(uint32_t xi) { return xi & 0x007fffffu; }
// zmort.hpp:129-151 (table lookup there; the value is floor(log2 x)).
inline int uint_log_base2(uint64_t x) {
  int l = -1;
  while (x) {
    x >>= 1;
    ++l;
  }
  return l;
}
inline uint64_t to_uint(double x) {
  uint64_t u;
  std::memcpy(&u, &x, 8);
  return u;
}
inline uint32_t to_uint(float x) {
  uint32_t u;
  std::memcpy(&u, &x, 4);
  return u;
}
// zmort.hpp:166-188.
template <class S>
inline int float_xor_msb(S p, S q) {
  constexpr int nbits = sizeof(S) == 8 ? 52 : 23;
  if (p == q || p == -q) return std::numeric_limits<int>::min();
  const auto pui = to_uint(p), qui = to_uint(q);
  const int pe = float_exp(pui), qe = float_exp(qui);
  if (pe == qe) {
    const auto x = float_sig(pui) ^ float_sig(qui);
    if (x > 0) return pe + uint_log_base2(static_cast<uint64_t>(x)) - nbits;
    return pe;
  }
  return std::max(pe, qe);
}
// zmort.hpp:195-220 (zorder_knn::Less<Point, d>): axes scanned d-1 .. 0, strict <, sign mismatch decides at once.
template <class S>
inline bool zorder_less(const S* p, const S* q, int d) {
  int x = std::numeric_limits<int>::min();
  int k = 0;
  for (int j = d; j-- > 0;) {
    if ((p[j] < S(0)) != (q[j] < S(0))) return p[j] < q[j];
    const int y = float_xor_msb(p[j], q[j]);
    if (x < y) {
      x = y;
      k = j;
    }
  }
  return p[k] < q[k];
}
// zmort.hpp:228-265 (mundy::math::zmorton_less): axes scanned 0,1,2; sign mismatches resolved z, y, x.
inline bool zmorton_less(const double* p, const double* q) {
  int sl[3];
  for (int k = 0; k < 3; ++k) sl[k] = ((p[k] < 0.0) != (q[k] < 0.0)) ? (p[k] < q[k]) : -1;
  int x = std::numeric_limits<int>::min();
  int k = 0;
  for (int j = 0; j < 3; ++j) {
    const int y = float_xor_msb(p[j], q[j]);
    if (x < y) {
      x = y;
      k = j;
    }
  }
  return (sl[2] != -1) ? sl[2] : ((sl[1] != -1) ? sl[1] : ((sl[0] != -1) ? sl[0] : (p[k] < q[k])));
}

// ---------------------------------------------------------------------------------------------------------------
// Hilbert curve generator (mundy/math/src/mundy_math/Hilbert.hpp:48-128)
// ---------------------------------------------------------------------------------------------------------------
inline size_t hilbert_3d(size_t s, size_t i, std::vector<V3>& pos, V3 cur, V3 dr1, V3 dr2, V3 dr3) {
  if (s == 1) {
    pos[i] = cur;
    return i + 1;
  }
  const size_t snew = s / 2;
  const double sn = static_cast<double>(snew);
  V3 cn = cur;
  for (const V3& dr : {dr1, dr2, dr3}) {
    const V3 st{dr.x < 0.0 ? 1.0 : 0.0, dr.y < 0.0 ? 1.0 : 0.0, dr.z < 0.0 ? 1.0 : 0.0};
    cn = cn - sn * V3{st.x * dr.x, st.y * dr.y, st.z * dr.z};
  }
  const V3 m1 = -1.0 * dr1, m2 = -1.0 * dr2, m3 = -1.0 * dr3;
  i = hilbert_3d(snew, i, pos, cn, dr2, dr3, dr1);
  i = hilbert_3d(snew, i, pos, cn + sn * dr1, dr3, dr1, dr2);
  i = hilbert_3d(snew, i, pos, cn + sn * (dr1 + dr2), dr3, dr1, dr2);
  i = hilbert_3d(snew, i, pos, cn + sn * dr2, m1, m2, dr3);
  i = hilbert_3d(snew, i, pos, cn + sn * (dr2 + dr3), m1, m2, dr3);
  i = hilbert_3d(snew, i, pos, cn + sn * ((dr1 + dr2) + dr3), m3, dr1, m2);
  i = hilbert_3d(snew, i, pos, cn + sn * (dr1 + dr3), m3, dr1, m2);
  i = hilbert_3d(snew, i, pos, cn + sn * dr3, dr2, m3, m1);
  return i;
}
// Hilbert.hpp:86-128.
inline void create_hilbert_positions_and_directors(size_t num_points, V3 orientation, double side_length,
                                                   std::vector<V3>& positions, std::vector<V3>& directors) {
  size_t ns = 2;
  while (ns * ns * ns < num_points) ns *= 2;
  positions.assign(ns * ns * ns, V3{0, 0, 0});
  const V3 zhat{0, 0, 1};
  V3 d1 = orientation;
  d1 = d1 * (1.0 / 1.0);
  {
    const double nn = norm(d1);
    d1 = {d1.x / nn, d1.y / nn, d1.z / nn};
  }
  V3 d2 = cross(zhat, d1);
  {
    const double nn = norm(d2);
    d2 = {d2.x / nn, d2.y / nn, d2.z / nn};
  }
  V3 d3 = cross(d1, d2);
  {
    const double nn = norm(d3);
    d3 = {d3.x / nn, d3.y / nn, d3.z / nn};
  }
  hilbert_3d(ns, 0, positions, V3{0, 0, 0}, side_length * d1, side_length * d2, side_length * d3);
  directors.resize(positions.size() - 1);
  for (size_t i = 0; i < directors.size(); ++i) {
    V3 d = positions[(i + 1) % positions.size()] - positions[i];
    const double nn = norm(d);
    directors[i] = {d.x / nn, d.y / nn, d.z / nn};
  }
}

}  // namespace moracle
