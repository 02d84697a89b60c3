// ellipsoid_device.hpp -- the pieces of the ellipsoid shared-normal distance that every lane evaluates: the foot-point
// maps of mundy_geom/primitives/Ellipsoid.hpp:420-468 and the two scalar helpers of the line search
// (mundy_math/impl/minimize_impl.hpp:57-87).  The minimiser itself -- the 3x3 multistart L-BFGS(m = 10) over (theta, phi)
// of mundy_geom/distance/EllipsoidEllipsoid.hpp:62-151 / PointEllipsoid.hpp:61-135 -- runs as a per-lane state machine
// (ellipsoid_lockstep.hpp).  This kernel family is compute/latency bound (~2*10^6 fp64 instructions per 160-B pair),
// not HBM bound.
// Parity: sin/cos are det_sincos (geom_device.hpp), one fixed sequence of IEEE operations that the oracle evaluates too,
// so the iterates -- and the local minimum a pair settles in -- are the oracle's bit for bit; against the reference's
// libm the bar stays its own 1e-4 (UnitTestEllipsoidEllipsoid.cpp:52-53).
#pragma once
#include "geom_device.hpp"

namespace mhip {
namespace lbfgs {

constexpr double kEps = 2.220446049250313e-16;
constexpr int M = 10;  // lbfgs_max_memory_size (EllipsoidEllipsoid.hpp:116)

struct V2 {
  double a, b;
};
__device__ inline double dot2(V2 x, V2 y) { return x.a * y.a + x.b * y.b; }
__device__ inline double clampd(double mn, double mx, double v) { return (v < mn) ? mn : (v > mx) ? mx : v; }

// minimize_impl.hpp:57-87
__device__ inline double poly_min_extrap(double f0, double d0, double f1, double d1, double limit) {
  const double n = 3 * (f1 - f0) - 2 * d0 - d1;
  const double e = d0 + d1 - 2 * (f1 - f0);
  const double t2 = dmax(n * n - 3 * e * d0, 0.0);
  if (fabs(e) <= kEps) return 0.5;
  const double temp = sqrt(t2);
  const double x1 = (temp - n) / (3 * e);
  const double x2 = -(temp + n) / (3 * e);
  const double y1 = f0 + d0 * x1 + n * x1 * x1 + e * x1 * x1 * x1;
  const double y2 = f0 + d0 * x2 + n * x2 * x2 + e * x2 * x2 * x2;
  return clampd(0.0, limit, (y1 < y2) ? x1 : x2);
}

}  // namespace lbfgs


struct EllipsoidD {
  V3 c;
  Quat q;
  V3 r;
};

// map_body_frame_normal_to_ellipsoid (mundy_geom/primitives/Ellipsoid.hpp:420-460)
__device__ inline V3 body_normal_to_foot(V3 nh, const EllipsoidD& el) {
  const double r1 = el.r.x, r2 = el.r.y, r3 = el.r.z;
  const double s0 = copysign(1.0, nh.x), s1 = copysign(1.0, nh.y), s2 = copysign(1.0, nh.z);
  double alpha1, alpha2;
  if (s0 * nh.x > kZeroTol) {
    const double tmp0 = 1.0 / (r1 * nh.x);
    const double tmp1 = tmp0 * r2 * nh.y;
    const double tmp2 = tmp0 * r3 * nh.z;
    alpha1 = 1.0 / (1.0 + tmp1 * tmp1);
    alpha2 = 1.0 / (1.0 + tmp2 * tmp2 * alpha1);
  } else if (s1 * nh.y > kZeroTol) {
    const double tmp = r3 * nh.z / (r2 * nh.y);
    alpha1 = 0.0;
    alpha2 = 1.0 / (1.0 + tmp * tmp);
  } else {
    alpha1 = 0.0;
    alpha2 = 0.0;
  }
  const double sa1 = sqrt(alpha1), sa2 = sqrt(alpha2);
  return {0.5 * s0 * ((1.0 + s0) * r1 + (1.0 - s0) * r1) * sa1 * sa2,
          0.5 * s1 * ((1.0 + s1) * r2 + (1.0 - s1) * r2) * sqrt(1.0 - alpha1) * sa2,
          0.5 * s2 * ((1.0 + s2) * r3 + (1.0 - s2) * r3) * sqrt(1.0 - alpha2)};
}
// map_surface_normal_to_foot_point_on_ellipsoid (Ellipsoid.hpp:462-468)
__device__ inline V3 normal_to_foot_point(V3 lab_n, const EllipsoidD& el) {
  const Quat qc{el.q.w, -el.q.x, -el.q.y, -el.q.z};
  const V3 body_n = qrot(qc, lab_n);
  return qrot(el.q, body_normal_to_foot(body_n, el)) + el.c;
}

}  // namespace mhip
