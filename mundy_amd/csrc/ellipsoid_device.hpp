// ellipsoid_device.hpp -- shared-normal signed distance between ellipsoids (and point-ellipsoid) on the device:
// the 3x3 multistart L-BFGS(m = 10) over (theta, phi) of mundy_geom/distance/EllipsoidEllipsoid.hpp:62-151 /
// PointEllipsoid.hpp:61-135, with the allocation-free minimiser of mundy_math/impl/minimize_impl.hpp:46-605
// (Fletcher line search, central differences) specialised to two unknowns.  This kernel family is compute/latency
// bound (~10^4-10^5 fp64 flops per 160-B pair), not HBM bound.
// Parity: sin/cos come from the device math library, so iterates differ from the host in the last bits and the
// line-search branches can diverge; the bar is the reference's own 1e-4 (UnitTestEllipsoidEllipsoid.cpp:52-53).
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

// F: double operator()(V2) -- the objective.  Everything below is the reference algorithm for N = 2.
template <class F>
__device__ inline V2 central_diff(const F& f, V2 x, double eps) {
  V2 der;
  {
    const double old = x.a;
    V2 e = x;
    e.a = old + eps;
    const double fp = f(e);
    e.a = old - eps;
    const double fm = f(e);
    der.a = (fp - fm) / ((old + eps) - (old - eps));
  }
  {
    const double old = x.b;
    V2 e = x;
    e.b = old + eps;
    const double fp = f(e);
    e.b = old - eps;
    const double fm = f(e);
    der.b = (fp - fm) / ((old + eps) - (old - eps));
  }
  return der;
}

// line_search (minimize_impl.hpp:233-405) along x + alpha * s
template <class F>
__device__ inline double line_search(const F& f, V2 x, V2 s, double f0, double d0, double rho, double sigma,
                                     double min_f, int max_iter, double eps) {
  auto phi = [&](double a) { return f(V2{x.a + a * s.a, x.b + a * s.b}); };
  auto dphi = [&](double a) { return (phi(a + eps) - phi(a - eps)) / ((a + eps) - (a - eps)); };
  const double tau1a = 1.4, tau1b = 9, tau2 = 1.0 / 10.0, tau3 = 1.0 / 2.0;
  if (fabs(d0) <= fabs(f0) * kEps) return 0;
  if (f0 <= min_f) return 0;
  const double mu = (min_f - f0) / (rho * d0);
  double alpha = 1;
  if (mu < 0) alpha = -alpha;
  alpha = clampd(0.0, 0.65 * mu, alpha);
  double last_alpha = 0, last_val = f0, last_val_der = d0;
  double a, b, a_val, b_val, a_val_der, b_val_der;
  const double thresh = fabs(sigma * d0);
  int itr = 0;
  while (true) {
    ++itr;
    const double val = phi(alpha);
    const double val_der = dphi(alpha);
    if (val <= min_f) return alpha;
    if (val > f0 + rho * alpha * d0 || val >= last_val) {
      a_val = last_val; a_val_der = last_val_der; b_val = val; b_val_der = val_der;
      a = last_alpha; b = alpha;
      break;
    }
    if (fabs(val_der) <= thresh) return alpha;
    if (last_alpha == alpha || itr >= max_iter) return alpha;
    if (val_der >= 0) {
      a_val = val; a_val_der = val_der; b_val = last_val; b_val_der = last_val_der;
      a = alpha; b = last_alpha;
      break;
    }
    const double temp = alpha;
    double first, last;
    if (mu > 0) {
      first = dmin(mu, alpha + tau1a * (alpha - last_alpha));
      last = dmin(mu, alpha + tau1b * (alpha - last_alpha));
    } else {
      first = dmax(mu, alpha + tau1a * (alpha - last_alpha));
      last = dmax(mu, alpha + tau1b * (alpha - last_alpha));
    }
    if (last_alpha < alpha)
      alpha = last_alpha + (alpha - last_alpha) * poly_min_extrap(last_val, last_val_der, val, val_der, 1e10);
    else
      alpha = alpha + (last_alpha - alpha) * poly_min_extrap(val, val_der, last_val, last_val_der, 1e10);
    alpha = clampd(first, last, alpha);
    last_alpha = temp;
    last_val = val;
    last_val_der = val_der;
  }
  while (true) {
    ++itr;
    const double first = a + tau2 * (b - a);
    const double last = b - tau3 * (b - a);
    alpha = a + (b - a) * poly_min_extrap(a_val, a_val_der, b_val, b_val_der, 1.0);
    alpha = clampd(first, last, alpha);
    const double val = phi(alpha);
    const double val_der = dphi(alpha);
    if (val <= min_f || itr >= max_iter) return alpha;
    if (a == first || b == last) return b;
    const double max_possible_alpha = dmax(fabs(a), fabs(b));
    if (fabs(max_possible_alpha * d0) <= fabs(f0) * kEps) return alpha;
    if (val > f0 + rho * alpha * d0 || val >= a_val) {
      b = alpha; b_val = val; b_val_der = val_der;
    } else {
      if (fabs(val_der) <= thresh) return alpha;
      if ((b - a) * val_der >= 0) {
        b = a; b_val = a_val; b_val_der = a_val_der;
      }
      a = alpha; a_val = val; a_val_der = val_der;
    }
  }
}

// find_min_using_approximate_derivatives<10>(f, x, min_allowable_cost) with the defaults min_objective_delta = 1e-7,
// derivative_eps = 1e-7 (minimize.hpp:42-51; the callers' third argument binds to min_allowable_cost).
template <class F>
__device__ inline double find_min(const F& f, V2& x, double min_allowable_cost) {
  const double min_delta = 1e-7, eps = 1e-7;
  // lbfgs_search_strategy state (minimize_impl.hpp:407-566)
  V2 hs[M], hy[M];
  double hrho[M], halpha[M];
  int current_size = 0;
  bool been_used = false, stop_used = false;
  V2 prev_x{0, 0}, prev_g{0, 0};
  double prev_val = 0;
  double cost = f(x);
  V2 g = central_diff(f, x, eps);
  while (true) {
    // objective_delta_stop_strategy::should_continue_search (minimize_impl.hpp:164-183)
    if (stop_used && fabs(cost - prev_val) < min_delta) break;
    stop_used = true;
    prev_val = cost;
    if (!(cost > min_allowable_cost)) break;
    // get_next_direction
    V2 dir{-g.a, -g.b};
    if (!been_used) {
      been_used = true;
    } else {
      const V2 s{x.a - prev_x.a, x.b - prev_x.b}, y{g.a - prev_g.a, g.b - prev_g.b};
      const double temp = dot2(s, y);
      if (fabs(temp) > kEps) {
        if (current_size < M) {
          hs[current_size] = s; hy[current_size] = y; hrho[current_size] = 1.0 / temp;
          ++current_size;
        } else {
          for (int i = 1; i < M; ++i) {
            hs[i - 1] = hs[i]; hy[i - 1] = hy[i]; hrho[i - 1] = hrho[i];
          }
          hs[M - 1] = s; hy[M - 1] = y; hrho[M - 1] = 1.0 / temp;
        }
      } else {
        current_size = 0;
      }
      if (current_size > 0) {
        for (int i = current_size - 1; i >= 0; --i) {
          halpha[i] = hrho[i] * dot2(hs[i], dir);
          dir = V2{dir.a - halpha[i] * hy[i].a, dir.b - halpha[i] * hy[i].b};
        }
        double H0 = 1.0 / hrho[current_size - 1] / dot2(hy[current_size - 1], hy[current_size - 1]);
        H0 = clampd(0.001, 1000.0, H0);
        dir = V2{H0 * dir.a, H0 * dir.b};
        for (int i = 0; i < current_size; ++i) {
          const double beta = hrho[i] * dot2(hy[i], dir);
          dir = V2{dir.a + (halpha[i] - beta) * hs[i].a, dir.b + (halpha[i] - beta) * hs[i].b};
        }
      }
    }
    prev_x = x;
    prev_g = g;
    const double alpha = line_search(f, x, dir, cost, dot2(g, dir), 0.01, 0.9, min_allowable_cost, 100, eps);
    x = V2{alpha * dir.a + x.a, alpha * dir.b + x.b};
    g = central_diff(f, x, eps);
    cost = f(x);
  }
  return cost;
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

struct EllipsoidPair {
  double dist;
  V3 cp1, cp2, n1;
};

__device__ inline EllipsoidPair dist_ellipsoid_ellipsoid(const EllipsoidD& e1, const EllipsoidD& e2) {
  auto eval = [&](lbfgs::V2 tp, V3& n1, V3& f1, V3& f2) {
    double st, ct, sp, cp;  // one argument reduction per angle
    sincos(tp.a, &st, &ct);
    sincos(tp.b, &sp, &cp);
    n1 = V3{st * cp, st * sp, ct};
    f1 = normal_to_foot_point(n1, e1);
    f2 = normal_to_foot_point(V3{-n1.x, -n1.y, -n1.z}, e2);
    V3 sep;
    return dist_point_point(f1, f2, sep);
  };
  auto objective = [&](lbfgs::V2 tp) {
    V3 n1, f1, f2;
    return eval(tp, n1, f1, f2);
  };
  const double pi = 3.141592653589793;
  const double theta_guesses[3] = {0.0, 0.5 * pi, pi};
  const double phi_guesses[3] = {pi / 3.0, pi, 5.0 * (pi / 3.0)};
  double best = __builtin_huge_val();
  lbfgs::V2 best_tp{0.0, 0.0};
  for (int t = 0; t < 3; ++t)
    for (int p = 0; p < 3; ++p) {
      lbfgs::V2 tp{theta_guesses[t], phi_guesses[p]};
      const double d = lbfgs::find_min(objective, tp, 1e-8 /* get_relaxed_zero_tolerance<double>() */);
      if (d < best) {
        best = d;
        best_tp = tp;
      }
    }
  EllipsoidPair r;
  eval(best_tp, r.n1, r.cp1, r.cp2);
  r.dist = dot(r.cp2 - r.cp1, r.n1);
  return r;
}

// distance(SharedNormalSigned, Point, Ellipsoid, closest, normal) (PointEllipsoid.hpp:94-135)
__device__ inline double dist_point_ellipsoid(V3 point, const EllipsoidD& el, V3& closest, V3& normal) {
  auto eval = [&](lbfgs::V2 tp, V3& n, V3& f) {
    double st, ct, sp, cp;  // one argument reduction per angle
    sincos(tp.a, &st, &ct);
    sincos(tp.b, &sp, &cp);
    n = V3{st * cp, st * sp, ct};
    f = normal_to_foot_point(n, el);
    V3 sep;
    return dist_point_point(f, point, sep);
  };
  auto objective = [&](lbfgs::V2 tp) {
    V3 n, f;
    return eval(tp, n, f);
  };
  const double pi = 3.141592653589793;
  const double theta_guesses[3] = {0.0, 0.5 * pi, pi};
  const double phi_guesses[3] = {pi / 3.0, pi, 5.0 * (pi / 3.0)};
  double best = __builtin_huge_val();
  lbfgs::V2 best_tp{0.0, 0.0};
  for (int t = 0; t < 3; ++t)
    for (int p = 0; p < 3; ++p) {
      lbfgs::V2 tp{theta_guesses[t], phi_guesses[p]};
      const double d = lbfgs::find_min(objective, tp, 1e-8);
      if (d < best) {
        best = d;
        best_tp = tp;
      }
    }
  eval(best_tp, normal, closest);
  return dot(point - closest, normal);
}

}  // namespace mhip
