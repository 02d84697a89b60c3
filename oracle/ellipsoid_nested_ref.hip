// ellipsoid_nested_ref.hip -- TEST INFRASTRUCTURE (not part of libmundy_hip.so): the ellipsoid distances with the
// reference's minimiser written as ordinary nested loops, one thread per pair -- find_min / line_search / central
// differences of mundy_math/impl/minimize_impl.hpp:151-605 specialised to two unknowns, called nine times per pair as
// mundy_geom/distance/EllipsoidEllipsoid.hpp:106-151 and PointEllipsoid.hpp:94-135 do.  It is a close restatement of
// that algorithm (like oracle/, and for the same reason: to be a checker) and exists for ONE purpose: the production
// kernels run the same arithmetic as a per-lane state machine (mundy_amd/csrc/ellipsoid_lockstep.hpp), and the tests
// require every output of the two forms to agree bit for bit.  Built by oracle/ellipsoid_nested.py into
// oracle/libellipsoid_nested_ref.so; shares only the per-evaluation device functions with the product
// (foot-point maps, det_sincos, poly_min_extrap).
#include "../mundy_amd/csrc/ellipsoid_device.hpp"

namespace mhip {
namespace lbfgs {

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

struct EllipsoidPair {
  double dist;
  V3 cp1, cp2, n1;
};

__device__ inline EllipsoidPair dist_ellipsoid_ellipsoid(const EllipsoidD& e1, const EllipsoidD& e2) {
  auto eval = [&](lbfgs::V2 tp, V3& n1, V3& f1, V3& f2) {
    double st, ct, sp, cp;  // one argument reduction per angle
    det_sincos(tp.a, st, ct);
    det_sincos(tp.b, sp, cp);
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
    det_sincos(tp.a, st, ct);
    det_sincos(tp.b, sp, cp);
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


struct BodyD {
  int kind;
  V3 c;
  Quat q;
  V3 s;
};
__device__ inline EllipsoidD load_ellipsoid(const double* c, const double* q, const double* r, size_t i) {
  return {load3(c, i), load4q(q, i), load3(r, i)};
}

__global__ void __launch_bounds__(64)
    k_ref_dist_ellipsoids(size_t n, const double* c1, const double* q1, const double* r1, const double* c2,
                          const double* q2, const double* r2, double* dist, double* cp1, double* cp2, double* n1,
                          double* n2) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const EllipsoidPair r = dist_ellipsoid_ellipsoid(load_ellipsoid(c1, q1, r1, i), load_ellipsoid(c2, q2, r2, i));
  dist[i] = r.dist;
  store3(cp1, i, r.cp1);
  store3(cp2, i, r.cp2);
  store3(n1, i, r.n1);
  store3(n2, i, V3{-r.n1.x, -r.n1.y, -r.n1.z});
}
__global__ void __launch_bounds__(64)
    k_ref_dist_point_ellipsoid(size_t n, const double* p, const double* c, const double* q, const double* r, double* dist,
                               double* cp, double* nrm) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  V3 closest, normal;
  dist[i] = dist_point_ellipsoid(load3(p, i), load_ellipsoid(c, q, r, i), closest, normal);
  store3(cp, i, closest);
  store3(nrm, i, normal);
}
// rod (centre, quaternion, radius, length) against ellipsoid: the point - ellipsoid minimisation with the closest point
// of the rod's centreline as the point (the R-E class of mundy_amd/csrc/mixed.hip): normal = rod -> ellipsoid
__global__ void __launch_bounds__(64)
    k_ref_rod_ellipsoid(size_t n, const double* rc, const double* rq, const double* rshape, const double* ec,
                        const double* eq, const double* er, double* sep, double* normal, double* cp1, double* cp2) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const BodyD A{1, load3(rc, i), load4q(rq, i), load3(rshape, i)};
  const EllipsoidD el = load_ellipsoid(ec, eq, er, i);
  const V3 hd = rod_half_axis(A.q, A.s.y);
  const V3 p0 = A.c - hd, p1 = A.c + hd;
  auto eval = [&](lbfgs::V2 tp, V3& ne, V3& foot, V3& closest) {
    double st, ct, sp, cp;
    det_sincos(tp.a, st, ct);
    det_sincos(tp.b, sp, cp);
    ne = V3{st * cp, st * sp, ct};
    foot = normal_to_foot_point(ne, el);
    V3 sv;
    double tt;
    return dist_point_segment(foot, p0, p1, closest, tt, sv);
  };
  auto objective = [&](lbfgs::V2 tp) {
    V3 ne, foot, closest;
    return eval(tp, ne, foot, closest);
  };
  const double pi = 3.141592653589793;
  const double tg[3] = {0.0, 0.5 * pi, pi}, pg[3] = {pi / 3.0, pi, 5.0 * (pi / 3.0)};
  double best = __builtin_huge_val();
  lbfgs::V2 btp{0.0, 0.0};
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) {
      lbfgs::V2 tp{tg[a], pg[b]};
      const double d = lbfgs::find_min(objective, tp, 1e-8);
      if (d < best) {
        best = d;
        btp = tp;
      }
    }
  V3 ne, foot, closest;
  eval(btp, ne, foot, closest);
  sep[i] = dot(closest - foot, ne) - A.s.x;
  store3(normal, i, V3{-ne.x, -ne.y, -ne.z});
  store3(cp1, i, closest);
  store3(cp2, i, foot);
}

}  // namespace mhip

using namespace mhip;

extern "C" {
// all pointers are device pointers; every launch is followed by a device synchronisation; returns the hipError_t
int ref_distance_ellipsoid_ellipsoid(size_t n, const double* c1, const double* q1, const double* r1, const double* c2,
                                     const double* q2, const double* r2, double* dist, double* cp1, double* cp2,
                                     double* n1, double* n2) {
  if (n == 0) return 0;
  k_ref_dist_ellipsoids<<<static_cast<unsigned>((n + 63) / 64), 64>>>(n, c1, q1, r1, c2, q2, r2, dist, cp1, cp2, n1, n2);
  return static_cast<int>(hipDeviceSynchronize());
}
int ref_distance_point_ellipsoid(size_t n, const double* p, const double* c, const double* q, const double* r,
                                 double* dist, double* cp, double* nrm) {
  if (n == 0) return 0;
  k_ref_dist_point_ellipsoid<<<static_cast<unsigned>((n + 63) / 64), 64>>>(n, p, c, q, r, dist, cp, nrm);
  return static_cast<int>(hipDeviceSynchronize());
}
int ref_contact_rod_ellipsoid(size_t n, const double* rc, const double* rq, const double* rshape, const double* ec,
                              const double* eq, const double* er, double* sep, double* normal, double* cp1,
                              double* cp2) {
  if (n == 0) return 0;
  k_ref_rod_ellipsoid<<<static_cast<unsigned>((n + 63) / 64), 64>>>(n, rc, rq, rshape, ec, eq, er, sep, normal, cp1, cp2);
  return static_cast<int>(hipDeviceSynchronize());
}
}  // extern "C"
