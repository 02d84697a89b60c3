// geom_device.hpp -- per-object / per-pair device functions of the narrow phase.
// Each function states which MundyGeom routine it computes (path:line under the MuNDy tree); arithmetic order follows
// the reference so results are bit-identical to a scalar evaluation (build: -ffp-contract=off).
#pragma once
#include "mhip_internal.hpp"

namespace mhip {

struct Box {
  V3 lo, hi;
};

__device__ inline double dmin(double a, double b) { return (b < a) ? b : a; }
__device__ inline double dmax(double a, double b) { return (a < b) ? b : a; }

// sin and cos as ONE fixed sequence of IEEE double operations (the build never contracts a*b+c): x 2/pi to the nearest
// integer, Cody-Waite subtraction of that multiple of pi/2 in two-part pieces, then the usual minimax polynomials on
// [-pi/4, pi/4] with the reduction's tail carried through (the construction of fdlibm's kernels; < 1 ulp for the
// |x| < 10^5 that occur here).  The device math library's sincos and a host libm differ in the last ulp here and there,
// which the L-BFGS line search of the ellipsoid distances turns into another branch; a CPU evaluation of THIS sequence
// lands on the same bits (the oracle's kTrigShared mode), and it is shorter than the library routine, which also
// handles huge arguments.
__device__ inline void det_sincos(double x, double& s, double& c) {
  const double fn = rint(x * 6.36619772367581382433e-01);
  const int n = static_cast<int>(fn);
  double r = x - fn * 1.57079632673412561417e+00;
  double w = fn * 6.07710050650619224932e-11;
  {
    const double t = r;
    w = fn * 6.07710050630396597660e-11;
    r = t - w;
    w = fn * 2.02226624879595063154e-21 - ((t - r) - w);
  }
  const double y0 = r - w;
  const double y1 = (r - y0) - w;
  const double z = y0 * y0;
  const double ps = -1.98412698298579493134e-04 +
                    z * (2.75573137070700676789e-06 + z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10));
  const double rs = 8.33333333332248946124e-03 + z * ps;
  const double v = z * y0;
  const double sv = y0 - ((z * (0.5 * y1 - v * rs) - y1) - v * -1.66666666666666324348e-01);
  const double pc = 2.48015872894767294178e-05 +
                    z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11));
  const double rc = z * (4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * pc));
  const double hz = 0.5 * z;
  const double wc = 1.0 - hz;
  const double cv = wc + (((1.0 - wc) - hz) + (z * rc - y0 * y1));
  const int q = n & 3;
  s = (q == 0) ? sv : (q == 1) ? cv : (q == 2) ? -sv : -cv;
  c = (q == 0) ? cv : (q == 1) ? -sv : (q == 2) ? -cv : sv;
}

// compute_aabb(Sphere): centre -/+ ones*radius  (mundy_geom/compute_aabb.hpp:72-80)
__device__ inline Box aabb_sphere(V3 c, double r) {
  const double e = 1.0 * r;
  return {{c.x - e, c.y - e, c.z - e}, {c.x + e, c.y + e, c.z + e}};
}
// centreline half vector 0.5*L*(q*zhat)  (mundy_geom/compute_aabb.hpp:115-117)
__device__ inline V3 rod_half_axis(Quat q, double length) { return (0.5 * length) * qrot(q, V3{0.0, 0.0, 1.0}); }
// compute_aabb(SpherocylinderSegment)  (mundy_geom/compute_aabb.hpp:129-143); with endpoints c -/+ half axis it is
// compute_aabb(Spherocylinder) (:105-127)
__device__ inline Box aabb_segment(V3 p0, V3 p1, double r) {
  return {{dmin(p0.x, p1.x) - r, dmin(p0.y, p1.y) - r, dmin(p0.z, p1.z) - r},
          {dmax(p0.x, p1.x) + r, dmax(p0.y, p1.y) + r, dmax(p0.z, p1.z) + r}};
}
// compute_aabb(Ellipsoid): min/max of centre -/+ q*radii (mundy_geom/compute_aabb.hpp:82-103) -- exact only for
// axis-aligned rotations; the reference's behaviour is kept.
__device__ inline Box aabb_ellipsoid(V3 c, Quat q, V3 radii) {
  const V3 rr = qrot(q, radii);
  const V3 a = c - rr, b = c + rr;
  return {{dmin(a.x, b.x), dmin(a.y, b.y), dmin(a.z, b.z)}, {dmax(a.x, b.x), dmax(a.y, b.y), dmax(a.z, b.z)}};
}

// BUILD EXTENSION (the flagged option of SURVEY row a7): tight conservative box, half extent along lab axis k =
// sqrt(sum_j (r_j (q * e_j)[k])^2).
__device__ inline Box aabb_ellipsoid_conservative(V3 c, Quat q, V3 radii) {
  const V3 a0 = qrot(q, V3{1.0, 0.0, 0.0}), a1 = qrot(q, V3{0.0, 1.0, 0.0}), a2 = qrot(q, V3{0.0, 0.0, 1.0});
  V3 e;
  {
    const double t0 = radii.x * a0.x, t1 = radii.y * a1.x, t2 = radii.z * a2.x;
    e.x = sqrt(t0 * t0 + (t1 * t1 + t2 * t2));
  }
  {
    const double t0 = radii.x * a0.y, t1 = radii.y * a1.y, t2 = radii.z * a2.y;
    e.y = sqrt(t0 * t0 + (t1 * t1 + t2 * t2));
  }
  {
    const double t0 = radii.x * a0.z, t1 = radii.y * a1.z, t2 = radii.z * a2.z;
    e.z = sqrt(t0 * t0 + (t1 * t1 + t2 * t2));
  }
  return {c - e, c + e};
}

// distance(Point, Point, sep)  (mundy_geom/distance/PointPoint.hpp:54-62)
__device__ inline double dist_point_point(V3 p1, V3 p2, V3& sep) {
  sep = p2 - p1;
  return norm(sep);
}

// distance(Point, LineSegment, closest, t, sep)  (mundy_geom/distance/PointLineSegment.hpp:128-172).
// t is NOT clamped when the closest point is an endpoint (:158-163).
__device__ inline double dist_point_segment(V3 p, V3 a, V3 b, V3& closest, double& t, V3& sep) {
  const V3 ab = b - a;
  const double num = dot(ab, p - a);
  if ((num < kZeroTol) & (num > -kZeroTol)) {
    closest = a;
    t = 0.0;
  } else {
    const double den = dot(ab, ab);
    if (den < kZeroTol) {
      closest = a;
      t = 0.0;
    } else {
      t = num / den;
      if (t < 0.0) {
        closest = a;
      } else if (t > 1.0) {
        closest = b;
      } else {
        closest = a + t * ab;
      }
    }
  }
  return dist_point_point(p, closest, sep);
}

struct SegSeg {
  double dist, s, t;
  V3 cp1, cp2, sep;
};

// distance(LineSegment, LineSegment, cp1, cp2, s, t, sep)  (mundy_geom/distance/LineSegmentLineSegment.hpp:189-318).
__device__ inline SegSeg dist_segment_segment(V3 l0, V3 l1, V3 m0, V3 m1) {
  SegSeg o;
  const V3 u = l1 - l0, v = m1 - m0, w = l0 - m0;
  const double a = dot(u, u), b = dot(u, v), c = dot(v, v), d = dot(u, w), e = dot(v, w);
  const double D = a * c - b * b;
  if (D < 3.162277660168379e-08 /* == sqrt(1e-15) correctly rounded, :215 */) {
    // colinear: best of the four endpoint-to-segment distances; ties resolved in the order 1,2,3,4 (:236-265)
    V3 c1, c2, c3, c4, s1, s2, s3, s4;
    double t1, t2, t3, t4;
    const double d1 = dist_point_segment(l0, m0, m1, c1, t1, s1);
    const double d2 = dist_point_segment(l1, m0, m1, c2, t2, s2);
    const double d3 = dist_point_segment(m0, l0, l1, c3, t3, s3);
    const double d4 = dist_point_segment(m1, l0, l1, c4, t4, s4);
    const double dm = dmin(dmin(d1, d2), dmin(d3, d4));
    if (dm == d1) {
      o.s = 0.0; o.t = t1; o.cp1 = l0; o.cp2 = c1; o.sep = s1;
    } else if (dm == d2) {
      o.s = 1.0; o.t = t2; o.cp1 = l1; o.cp2 = c2; o.sep = s2;
    } else if (dm == d3) {
      o.s = t3; o.t = 0.0; o.cp1 = c3; o.cp2 = m0; o.sep = s3;
    } else {
      o.s = t4; o.t = 1.0; o.cp1 = c4; o.cp2 = m1; o.sep = s4;
    }
    o.dist = dm;
    return o;
  }
  double sN = b * e - c * d, tN = a * e - b * d, sD = D, tD = D;
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
    const double bd = -d + b;
    if (bd < 0.0) {
      sN = 0.0;
    } else if (bd > a) {
      sN = sD;
    } else {
      sN = bd; sD = a;
    }
  }
  o.s = (fabs(sN) < kZeroTol) ? 0.0 : sN / sD;
  o.t = (fabs(tN) < kZeroTol) ? 0.0 : tN / tD;
  o.cp1 = l0 + o.s * u;
  o.cp2 = m0 + o.t * v;
  o.dist = dist_point_point(o.cp1, o.cp2, o.sep);
  return o;
}

// PeriodicScaledMetric::sep  (mundy_geom/periodicity.hpp:785-816): scale * (f - (double)(int64)round(f)),
// f = scale_inv * (p2 - p1)
struct Periodic {
  V3 scale, scale_inv;
};
__host__ __device__ inline Periodic make_periodic(const double* box) {
  return {{box[0], box[1], box[2]}, {1.0 / box[0], 1.0 / box[1], 1.0 / box[2]}};
}
__device__ inline double min_image1(double f) { return f - static_cast<double>(static_cast<long long>(round(f))); }
__device__ inline V3 periodic_sep(const Periodic& pm, V3 p1, V3 p2) {
  const V3 d = p2 - p1;
  const V3 f{pm.scale_inv.x * d.x, pm.scale_inv.y * d.y, pm.scale_inv.z * d.z};
  return {pm.scale.x * min_image1(f.x), pm.scale.y * min_image1(f.y), pm.scale.z * min_image1(f.z)};
}
// PeriodicScaledMetric::wrap with impl::safe_unit_mod1 (periodicity.hpp:140-150, :818-823)
__device__ inline double unit_mod1(double s) {
  const double k = static_cast<double>(static_cast<long long>(floor(s)));
  double t = s - k;
  if (fabs(t - 1.0) < kZeroTol) t = 0.0;
  return t;
}
__device__ inline V3 periodic_wrap(const Periodic& pm, V3 p) {
  return {pm.scale.x * unit_mod1(pm.scale_inv.x * p.x), pm.scale.y * unit_mod1(pm.scale_inv.y * p.y),
          pm.scale.z * unit_mod1(pm.scale_inv.z * p.z)};
}

// Triclinic cell: PeriodicMetric (mundy_geom/periodicity.hpp:233-332).  h holds the lattice vectors as columns
// (row-major storage); h_inv = math::inverse(h) = adjugate / determinant with the Laplace expansion and right folds of
// mundy_math/impl/MatrixImpl.hpp:481-506, mundy_math/Matrix.hpp:1596-1601 (the +/-1 cofactor factors are exact).
struct Triclinic {
  double h[9], hi[9];
};
__host__ __device__ inline double det2(double a, double b, double c, double d) { return a * d + (-(b * c)); }
__host__ __device__ inline double minor_det3(const double* m, int r, int c) {
  const int r0 = (r == 0) ? 1 : 0, r1 = (r == 2) ? 1 : 2, c0 = (c == 0) ? 1 : 0, c1 = (c == 2) ? 1 : 2;
  return det2(m[3 * r0 + c0], m[3 * r0 + c1], m[3 * r1 + c0], m[3 * r1 + c1]);
}
__host__ __device__ inline double determinant3(const double* m) {
  const double t0 = m[0] * minor_det3(m, 0, 0), t1 = -(m[1] * minor_det3(m, 0, 1)), t2 = m[2] * minor_det3(m, 0, 2);
  return t0 + (t1 + t2);
}
__host__ __device__ inline void inverse3(const double* m, double* out) {
  const double det = determinant3(m);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)  // adjugate(i, j) = cofactor(j, i), sign by flat-index parity
      out[3 * i + j] = (minor_det3(m, j, i) * (((3 * j + i) % 2 == 0) ? 1.0 : -1.0)) / det;
}
__host__ inline Triclinic make_triclinic(const double* h) {
  Triclinic t;
  for (int i = 0; i < 9; ++i) t.h[i] = h[i];
  inverse3(t.h, t.hi);
  return t;
}
__host__ __device__ inline V3 matvec3(const double* m, V3 v) {  // per-row dot (MatrixImpl.hpp:348-355)
  return {dot(V3{m[0], m[1], m[2]}, v), dot(V3{m[3], m[4], m[5]}, v), dot(V3{m[6], m[7], m[8]}, v)};
}
__device__ inline V3 periodic_sep(const Triclinic& pm, V3 p1, V3 p2) {  // periodicity.hpp:304-307
  const V3 f = matvec3(pm.hi, p2 - p1);
  return matvec3(pm.h, V3{min_image1(f.x), min_image1(f.y), min_image1(f.z)});
}
__device__ inline V3 periodic_wrap(const Triclinic& pm, V3 p) {  // periodicity.hpp:312-314
  const V3 f = matvec3(pm.hi, p);
  return matvec3(pm.h, V3{unit_mod1(f.x), unit_mod1(f.y), unit_mod1(f.z)});
}

}  // namespace mhip
