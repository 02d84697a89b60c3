// oracle_capi.cpp -- extern "C" batch wrappers over mundy_oracle.hpp so tests/ and bench.py (cpu_baseline leg) can
// drive the CPU restatement through ctypes.  TEST INFRASTRUCTURE ONLY (see mundy_oracle.hpp header).
//
// The "_mt" entry points are the CPU baseline: the same per-element functions run under OpenMP, the way the
// reference's Kokkos-OpenMP backend would run them (parallel_for over bodies/pairs, parallel_reduce for the BBPGD
// reductions, atomics for the force scatter at scrap/lcp_spheres/NgpLcp.cpp:467-472).
#include <omp.h>

#include <numeric>

#include "mundy_oracle.hpp"

using namespace moracle;

namespace {
inline V3 ld3(const double* p, size_t i) { return {p[3 * i], p[3 * i + 1], p[3 * i + 2]}; }
inline void st3(double* p, size_t i, const V3& v) {
  p[3 * i] = v.x;
  p[3 * i + 1] = v.y;
  p[3 * i + 2] = v.z;
}
inline Quat ldq(const double* p, size_t i) { return {p[4 * i], p[4 * i + 1], p[4 * i + 2], p[4 * i + 3]}; }
inline void st_aabb(double* out, size_t i, const AABB& b) {
  for (int k = 0; k < 3; ++k) {
    out[6 * i + k] = b.lo[k];
    out[6 * i + 3 + k] = b.hi[k];
  }
}
std::vector<int32_t> g_pairs;
}  // namespace

extern "C" {

int o_num_threads() { return omp_get_max_threads(); }
void o_set_num_threads(int n) { omp_set_num_threads(n > 0 ? n : 1); }

// ---- per-body geometry ------------------------------------------------------------------------------------------
void o_compute_aabb_spheres(size_t n, const double* center, const double* radius, double* out) {
#pragma omp parallel for
  for (size_t i = 0; i < n; ++i) st_aabb(out, i, compute_aabb_sphere(ld3(center, i), radius[i]));
}
void o_compute_aabb_spherocylinders(size_t n, const double* center, const double* quat, const double* radius,
                                    const double* length, double* out) {
#pragma omp parallel for
  for (size_t i = 0; i < n; ++i)
    st_aabb(out, i, compute_aabb_spherocylinder(ld3(center, i), ldq(quat, i), radius[i], length[i]));
}
void o_compute_aabb_ellipsoids(size_t n, const double* center, const double* quat, const double* radii,
                               double* out) {
#pragma omp parallel for
  for (size_t i = 0; i < n; ++i) st_aabb(out, i, compute_aabb_ellipsoid(ld3(center, i), ldq(quat, i), ld3(radii, i)));
}
void o_compute_aabb_ellipsoids_conservative(size_t n, const double* center, const double* quat, const double* radii,
                                            double* out) {
#pragma omp parallel for
  for (size_t i = 0; i < n; ++i)
    st_aabb(out, i, compute_aabb_ellipsoid_conservative(ld3(center, i), ldq(quat, i), ld3(radii, i)));
}
void o_compute_aabb_segments(size_t n, const double* p0, const double* p1, const double* radius, double* out) {
#pragma omp parallel for
  for (size_t i = 0; i < n; ++i) st_aabb(out, i, compute_aabb_segment(ld3(p0, i), ld3(p1, i), radius[i]));
}
void o_bounding_radius_spherocylinders(size_t n, const double* radius, const double* length, double* out) {
  for (size_t i = 0; i < n; ++i) out[i] = bounding_radius_spherocylinder(radius[i], length[i]);
}
void o_bounding_radius_ellipsoids(size_t n, const double* radii, double* out) {
  for (size_t i = 0; i < n; ++i) out[i] = bounding_radius_ellipsoid(ld3(radii, i));
}
void o_bounding_radius_segments(size_t n, const double* p0, const double* p1, const double* radius, double* out) {
  for (size_t i = 0; i < n; ++i) out[i] = bounding_radius_segment(ld3(p0, i), ld3(p1, i), radius[i]);
}
// seg record: p0(3) p1(3) radius pad  = 8 doubles
void o_spherocylinder_segments(size_t n, const double* center, const double* quat, const double* radius,
                               const double* length, double* seg) {
#pragma omp parallel for
  for (size_t i = 0; i < n; ++i) {
    const V3 c = ld3(center, i);
    const V3 d = spherocylinder_half_axis(ldq(quat, i), length[i]);
    const V3 p0 = c - d, p1 = c + d;
    double* s = seg + 8 * i;
    s[0] = p0.x; s[1] = p0.y; s[2] = p0.z; s[3] = p1.x; s[4] = p1.y; s[5] = p1.z; s[6] = radius[i]; s[7] = 0.0;
  }
}
void o_quat_rotate(size_t n, const double* quat, const double* v, double* out) {
  for (size_t i = 0; i < n; ++i) st3(out, i, qrot(ldq(quat, i), ld3(v, i)));
}
void o_quat_from_parallel_transport(size_t n, const double* from, const double* to, double* out) {
  for (size_t i = 0; i < n; ++i) {
    const Quat q = quat_from_parallel_transport(ld3(from, i), ld3(to, i));
    out[4 * i] = q.w; out[4 * i + 1] = q.x; out[4 * i + 2] = q.y; out[4 * i + 3] = q.z;
  }
}

// ---- distances ----------------------------------------------------------------------------------------------------
void o_distance_point_segment(size_t n, const double* p, const double* a0, const double* a1, double* dist,
                              double* cp, double* t, double* sep) {
  for (size_t i = 0; i < n; ++i) {
    V3 c, s;
    dist[i] = distance_point_segment(ld3(p, i), ld3(a0, i), ld3(a1, i), c, t[i], s);
    st3(cp, i, c);
    st3(sep, i, s);
  }
}
void o_distance_segment_segment(size_t n, const double* a0, const double* a1, const double* b0, const double* b1,
                                double* dist, double* cp1, double* cp2, double* s, double* t, double* sep) {
#pragma omp parallel for
  for (size_t i = 0; i < n; ++i) {
    V3 c1, c2, sv;
    dist[i] = distance_segment_segment(ld3(a0, i), ld3(a1, i), ld3(b0, i), ld3(b1, i), c1, c2, s[i], t[i], sv);
    st3(cp1, i, c1);
    st3(cp2, i, c2);
    st3(sep, i, sv);
  }
}
void o_distance_sphere_sphere(size_t n, const double* c1, const double* r1, const double* c2, const double* r2,
                              double* dist, double* sep) {
  for (size_t i = 0; i < n; ++i) {
    V3 s;
    dist[i] = distance_sphere_sphere(ld3(c1, i), r1[i], ld3(c2, i), r2[i], s);
    st3(sep, i, s);
  }
}
void o_distance_point_sphere(size_t n, const double* p, const double* c, const double* r, double* dist, double* sep) {
  for (size_t i = 0; i < n; ++i) {
    V3 s;
    dist[i] = distance_point_sphere(ld3(p, i), ld3(c, i), r[i], s);
    st3(sep, i, s);
  }
}
void o_distance_segment_sphere(size_t n, const double* a0, const double* a1, const double* c, const double* r,
                               double* dist, double* cp, double* t, double* sep) {
  for (size_t i = 0; i < n; ++i) {
    V3 cl, s;
    dist[i] = distance_segment_sphere(ld3(a0, i), ld3(a1, i), ld3(c, i), r[i], cl, t[i], s);
    st3(cp, i, cl);
    st3(sep, i, s);
  }
}
void o_contact_spheres(size_t C, const int32_t* pairs, const double* center, const double* radius,
                       const double* box, double* sep, double* normal) {
  const PeriodicScaledMetric pm(box ? V3{box[0], box[1], box[2]} : V3{1, 1, 1});
#pragma omp parallel for
  for (size_t c = 0; c < C; ++c) {
    const int32_t i = pairs[2 * c], j = pairs[2 * c + 1];
    V3 n;
    sep[c] = box ? contact_spheres_periodic(pm, ld3(center, i), radius[i], ld3(center, j), radius[j], n)
                 : contact_spheres(ld3(center, i), radius[i], ld3(center, j), radius[j], n);
    st3(normal, c, n);
  }
}
// seg = [N][8] records from o_spherocylinder_segments; ra/rb = closest point - body centre (lever arms).
// box (null or [3]): rod j at the lattice image whose centre is nearest to rod i's centre -- shift = (c_i + sep(c_i,
// c_j)) - c_j with PeriodicScaledMetric::sep (periodicity.hpp:812-816), a rigid translation as wrap_rigid moves a
// spherocylinder (:1094-1113).
void o_contact_spherocylinders(size_t C, const int32_t* pairs, const double* seg, const double* center,
                               const double* box, double* sep, double* normal, double* cp1, double* cp2, double* ra,
                               double* rb, double* s, double* t) {
  const PeriodicScaledMetric pm(box ? V3{box[0], box[1], box[2]} : V3{1, 1, 1});
#pragma omp parallel for
  for (size_t c = 0; c < C; ++c) {
    const int32_t i = pairs[2 * c], j = pairs[2 * c + 1];
    const double* si = seg + 8 * i;
    const double* sj = seg + 8 * j;
    V3 shift{0.0, 0.0, 0.0};
    V3 q0{sj[0], sj[1], sj[2]}, q1{sj[3], sj[4], sj[5]};
    if (box) {
      const V3 ci = ld3(center, i), cj = ld3(center, j);
      shift = (ci + pm.sep(ci, cj)) - cj;
      q0 = q0 + shift;
      q1 = q1 + shift;
    }
    const RodContact rc = contact_segments({si[0], si[1], si[2]}, {si[3], si[4], si[5]}, si[6], q0, q1, sj[6]);
    sep[c] = rc.sep;
    st3(normal, c, rc.normal);
    if (cp1) st3(cp1, c, rc.cp1);
    if (cp2) st3(cp2, c, rc.cp2);
    if (ra) st3(ra, c, rc.cp1 - ld3(center, i));
    if (rb) st3(rb, c, rc.cp2 - (box ? ld3(center, j) + shift : ld3(center, j)));
    // arclengths of the contact points (the clamped closest points the linker uses,
    // SpherocylinderSpherocylinderLinker.cpp:246-247), not the raw line parameters of the colinear branch
    if (s) s[c] = contact_arclength(rc.s);
    if (t) t[c] = contact_arclength(rc.t);
  }
}

// ---- ellipsoids / minimize ------------------------------------------------------------------------------------------
void o_distance_ellipsoid_ellipsoid(size_t n, const double* c1, const double* q1, const double* r1, const double* c2,
                                    const double* q2, const double* r2, double* dist, double* cp1, double* cp2,
                                    double* n1, double* n2) {
#pragma omp parallel for schedule(dynamic, 16)
  for (size_t i = 0; i < n; ++i) {
    const EllipsoidPairResult r = distance_ellipsoid_ellipsoid({ld3(c1, i), ldq(q1, i), ld3(r1, i)},
                                                               {ld3(c2, i), ldq(q2, i), ld3(r2, i)});
    dist[i] = r.dist;
    if (cp1) st3(cp1, i, r.cp1);
    if (cp2) st3(cp2, i, r.cp2);
    if (n1) st3(n1, i, r.n1);
    if (n2) st3(n2, i, r.n2);
  }
}
void o_distance_point_ellipsoid(size_t n, const double* p, const double* c, const double* q, const double* r,
                                double* dist, double* cp, double* nrm) {
#pragma omp parallel for schedule(dynamic, 16)
  for (size_t i = 0; i < n; ++i) {
    V3 closest, normal;
    dist[i] = distance_point_ellipsoid(ld3(p, i), {ld3(c, i), ldq(q, i), ld3(r, i)}, closest, normal);
    if (cp) st3(cp, i, closest);
    if (nrm) st3(nrm, i, normal);
  }
}
// UnitTestMinimize.cpp:45-105 problems: kind 0 quadratic1, 1 quadratic2 (N = 2), 2 rosenbrock (N = 42)
double o_minimize_test(int kind, double* x) {
  using namespace minimize;
  if (kind == 2) {
    Vec<42> v;
    for (int i = 0; i < 42; ++i) v[i] = x[i];
    auto rosen = [](const Vec<42>& y) {
      double sum = 0.0;
      for (size_t i = 0; i < 41; ++i) sum += 2.0 * std::pow(y[i + 1] - y[i] * y[i], 2.0) + std::pow(1.0 - y[i], 2.0);
      return sum;
    };
    const double c = find_min<10, 42>(rosen, v, 1e-7);
    for (int i = 0; i < 42; ++i) x[i] = v[i];
    return c;
  }
  Vec<2> v{{x[0], x[1]}};
  double c;
  if (kind == 0)
    c = find_min<10, 2>([](const Vec<2>& y) { return y[0] * y[0] + y[1] * y[1]; }, v, 1e-7);
  else
    c = find_min<10, 2>([](const Vec<2>& y) { return (y[0] - 2.0) * (y[0] - 2.0) + (y[1] + 1.0) * (y[1] + 1.0); }, v,
                        1e-7);
  x[0] = v[0];
  x[1] = v[1];
  return c;
}

// ---- mixed shapes ---------------------------------------------------------------------------------------------------
static MixedBody ld_body(const int32_t* kind, const double* c, const double* q, const double* shape, size_t i) {
  return {kind[i], ld3(c, i), ldq(q, i), ld3(shape, i)};
}
void o_aabb_mixed(size_t n, const int32_t* kind, const double* c, const double* q, const double* shape, double* aabb,
                  double* brad) {
#pragma omp parallel for
  for (size_t i = 0; i < n; ++i) {
    const MixedBody b = ld_body(kind, c, q, shape, i);
    st_aabb(aabb, i, compute_aabb_mixed(b));
    brad[i] = bounding_radius_mixed(b);
  }
}
// box (null or [3]): body j at the nearest lattice image of its centre, c_j' = c_i + sep(c_i, c_j)
void o_contact_mixed(size_t C, const int32_t* pairs, const int32_t* kind, const double* c, const double* q,
                     const double* shape, const double* box, double* sep, double* normal, double* cp1, double* cp2,
                     double* ra, double* rb) {
  const PeriodicScaledMetric pm(box ? V3{box[0], box[1], box[2]} : V3{1, 1, 1});
#pragma omp parallel for schedule(dynamic, 64)
  for (size_t k = 0; k < C; ++k) {
    const int32_t i = pairs[2 * k], j = pairs[2 * k + 1];
    const MixedBody bi = ld_body(kind, c, q, shape, i);
    MixedBody bj = ld_body(kind, c, q, shape, j);
    if (box) bj.c = bi.c + pm.sep(bi.c, bj.c);
    const MixedContact m = contact_mixed(bi, bj);
    sep[k] = m.sep;
    st3(normal, k, m.normal);
    st3(cp1, k, m.cp1);
    st3(cp2, k, m.cp2);
    st3(ra, k, m.cp1 - bi.c);
    st3(rb, k, m.cp2 - bj.c);
  }
}

// ---- time integration (scrap/lcp_spheres/NgpLcp.cpp:898 + Quaternion.hpp:1366-1383) -------------------------------
void o_integrate_euler(size_t n, double dt, const double* vel, double* center, double* quat) {
  for (size_t i = 0; i < n; ++i) {
    for (int k = 0; k < 3; ++k) center[3 * i + k] = dt * vel[6 * i + k] + 1.0 * center[3 * i + k];
    if (quat) {
      const Quat q = rotate_quaternion(ldq(quat, i), V3{vel[6 * i + 3], vel[6 * i + 4], vel[6 * i + 5]}, dt);
      quat[4 * i] = q.w; quat[4 * i + 1] = q.x; quat[4 * i + 2] = q.y; quat[4 * i + 3] = q.z;
    }
  }
}

// ---- periodicity --------------------------------------------------------------------------------------------------
void o_periodic_sep(size_t n, const double* box, const double* p1, const double* p2, double* out) {
  const PeriodicScaledMetric pm(V3{box[0], box[1], box[2]});
  for (size_t i = 0; i < n; ++i) st3(out, i, pm.sep(ld3(p1, i), ld3(p2, i)));
}
void o_periodic_wrap(size_t n, const double* box, const double* p, double* out) {
  const PeriodicScaledMetric pm(V3{box[0], box[1], box[2]});
  for (size_t i = 0; i < n; ++i) st3(out, i, pm.wrap(ld3(p, i)));
}
// triclinic PeriodicMetric (periodicity.hpp:233-332); h row-major 3x3, lattice vectors as columns
void o_unit_cell_inverse(const double* h, double* h_inv) { PeriodicMetric::inverse(h, h_inv); }
void o_periodic_sep_triclinic(size_t n, const double* h, const double* p1, const double* p2, double* out) {
  const PeriodicMetric pm(h);
  for (size_t i = 0; i < n; ++i) st3(out, i, pm.sep(ld3(p1, i), ld3(p2, i)));
}
void o_periodic_wrap_triclinic(size_t n, const double* h, const double* p, double* out) {
  const PeriodicMetric pm(h);
  for (size_t i = 0; i < n; ++i) st3(out, i, pm.wrap(ld3(p, i)));
}
void o_shift_image_triclinic(size_t n, const double* h, const double* p, const int* images, double* out) {
  const PeriodicMetric pm(h);
  for (size_t i = 0; i < n; ++i) st3(out, i, pm.shift_image(ld3(p, i), images + 3 * i));
}

// ---- neighbour search ---------------------------------------------------------------------------------------------
// method 0 = O(N^2), 1 = cell list.  Returns the pair count; fetch with o_search_fetch.
size_t o_search(int kind, int method, size_t n, const double* lo, const double* hi, const double* c, const double* R,
                const double* box, int symmetric) {
  if (method == 0)
    search_bruteforce(kind, n, lo, hi, c, R, box, symmetric != 0, g_pairs);
  else
    search_celllist(kind, n, lo, hi, c, R, box, symmetric != 0, g_pairs);
  return g_pairs.size() / 2;
}
size_t o_search_triclinic(int kind, size_t n, const double* lo, const double* hi, const double* c, const double* R,
                          const double* cell, int symmetric) {
  search_bruteforce_triclinic(kind, n, lo, hi, c, R, cell, symmetric != 0, g_pairs);
  return g_pairs.size() / 2;
}
void o_search_fetch(int32_t* out) { std::copy(g_pairs.begin(), g_pairs.end(), out); }
// rebuild test, mundy/mesh/src/mundy_mesh/GenNeighborLinkers.hpp:603-615: any |c_new - c_old| > 0.5 * buffer.
int o_moved_too_much(size_t n, const double* c_new, const double* c_old, double buffer) {
  bool moved = false;
  for (size_t i = 0; i < n; ++i) {
    const double dx = c_new[3 * i] - c_old[3 * i], dy = c_new[3 * i + 1] - c_old[3 * i + 1],
                 dz = c_new[3 * i + 2] - c_old[3 * i + 2];
    const double disp = std::sqrt(dx * dx + dy * dy + dz * dz);
    moved = moved || (disp > 0.5 * buffer);
  }
  return moved ? 1 : 0;
}

// ---- convex: vector kernels and solvers ----------------------------------------------------------------------------
void o_axpby(size_t n, double alpha, const double* x, double beta, double* y) { axpby(alpha, x, beta, y, n); }
void o_wrapped_axpbyz(size_t n, double alpha, const double* x, double beta, const double* y, double* z, int kind,
                      double lo, double hi) {
  wrapped_axpbyz(alpha, x, beta, y, z, n, Space{kind, lo, hi});
}
double o_diff_dot2(size_t n, const double* x, const double* y) { return diff_dot(x, y, n); }
double o_diff_dot4(size_t n, const double* x1, const double* x2, const double* y1, const double* y2) {
  return diff_dot(x1, x2, y1, y2, n);
}
double o_residual(size_t n, int resid_kind, const double* x, const double* g, int kind, double lo, double hi) {
  return residual(resid_kind, x, g, n, Space{kind, lo, hi});
}
double o_bb_step(size_t n, const double* x_old, const double* g_old, const double* x, const double* g) {
  return bb_step(x_old, g_old, x, g, n);
}
void o_gemv(size_t n, const double* A, const double* x, double* y) { DenseOp{A, n}(x, y); }

void o_solve_cqpp_dense(size_t n, const double* A, const double* q, int kind, double lo, double hi, int resid_kind,
                        unsigned max_iters, double tol, double* x, double* g, double* x_tmp, double* g_tmp,
                        unsigned* num_iters, double* res, int* converged) {
  const SolveResult r =
      solve_cqpp(DenseOp{A, n}, q, Space{kind, lo, hi}, resid_kind, max_iters, tol, n, x, g, x_tmp, g_tmp);
  *num_iters = r.num_iters;
  *res = r.residual;
  *converged = r.converged;
}

void o_solve_small_cqpp_batch(size_t batch, size_t n, const double* A, const double* q, int kind, double lo, double hi,
                              int resid_kind, unsigned max_iters, double tol, double* x, double* g, unsigned* iters,
                              double* res, int* conv) {
  for (size_t b = 0; b < batch; ++b) {
    const SolveResult r = solve_cqpp_small(n, A + b * n * n, q + b * n, Space{kind, lo, hi}, resid_kind, max_iters, tol,
                                           x + b * n, g + b * n);
    iters[b] = r.num_iters;
    res[b] = r.residual;
    conv[b] = r.converged;
  }
}

void o_contact_op_apply(size_t C, size_t N, const int32_t* pairs, const double* normal, const double* ra,
                        const double* rb, const double* mt, const double* mr, double dt, const double* x, double* y) {
  ContactOp op{pairs, normal, ra, rb, mt, mr, dt, C, N, {}, {}, {}, {}, {}, {}};
  op(x, y);
}

void o_solve_cqpp_contact(size_t C, size_t N, const int32_t* pairs, const double* normal, const double* ra,
                          const double* rb, const double* mt, const double* mr, double dt, const double* q, int kind,
                          double lo, double hi, int resid_kind, unsigned max_iters, double tol, double* x, double* g,
                          double* x_tmp, double* g_tmp, unsigned* num_iters, double* res, int* converged) {
  ContactOp op{pairs, normal, ra, rb, mt, mr, dt, C, N, {}, {}, {}, {}, {}, {}};
  const SolveResult r = solve_cqpp(op, q, Space{kind, lo, hi}, resid_kind, max_iters, tol, C, x, g, x_tmp, g_tmp);
  *num_iters = r.num_iters;
  *res = r.residual;
  *converged = r.converged;
}

// sin / cos used by the ellipsoid objective and rotate_quaternion (mundy_oracle.hpp, TrigMode)
void o_set_sphere_ellipsoid_route(int route) { sphere_ellipsoid_route() = route ? 1 : 0; }
void o_set_trig_mode(int mode) { trig_mode() = (mode == kTrigShared) ? kTrigShared : kTrigLibm; }
void o_shared_sincos(size_t n, const double* x, double* s, double* c) {
  for (size_t i = 0; i < n; ++i) shared_sincos(x[i], s[i], c[i]);
}
// summation mode of the BB-step reductions and the per-body sums (mundy_oracle.hpp, SumMode)
void o_set_sum_mode(int mode) { sum_mode() = (mode == kSumCompensated) ? kSumCompensated : kSumSerial; }
int o_get_sum_mode() { return sum_mode(); }

// the spherocylinder operator in rod-axis form (ContactOpRod)
void o_contact_op_apply_rod(size_t C, size_t N, const int32_t* pairs, const double* normal, const double* arc_s,
                            const double* arc_t, const double* seg, const double* mt, const double* mr, double dt,
                            const double* x, double* y, double* body_velocity /* [N][6] = (U, W) or null */) {
  ContactOpRod op{pairs, normal, arc_s, arc_t, seg, mt, mr, dt, C, N, {}, {}, {}, {}, {}};
  op(x, y);
  if (body_velocity)
    for (size_t b = 0; b < N; ++b)
      for (int k = 0; k < 3; ++k) {
        body_velocity[6 * b + k] = op.U[3 * b + k];
        body_velocity[6 * b + 3 + k] = op.Wv[3 * b + k];
      }
}
void o_solve_cqpp_contact_rod(size_t C, size_t N, const int32_t* pairs, const double* normal, const double* arc_s,
                              const double* arc_t, const double* seg, const double* mt, const double* mr, double dt,
                              const double* q, int kind, double lo, double hi, int resid_kind, unsigned max_iters,
                              double tol, double* x, double* g, double* x_tmp, double* g_tmp, unsigned* num_iters,
                              double* res, int* converged) {
  ContactOpRod op{pairs, normal, arc_s, arc_t, seg, mt, mr, dt, C, N, {}, {}, {}, {}, {}};
  const SolveResult r = solve_cqpp(op, q, Space{kind, lo, hi}, resid_kind, max_iters, tol, C, x, g, x_tmp, g_tmp);
  *num_iters = r.num_iters;
  *res = r.residual;
  *converged = r.converged;
}
void o_scrap_resolve_collisions_rod(size_t C, size_t N, const int32_t* pairs, const double* normal,
                                    const double* arc_s, const double* arc_t, const double* seg, const double* mt,
                                    const double* mr, double dt, const double* sep, double max_allowable_overlap,
                                    int max_iters, double* lam, double* g, double* res, int* ite_count,
                                    double* max_speed) {
  ContactOpRod op{pairs, normal, arc_s, arc_t, seg, mt, mr, dt, C, N, {}, {}, {}, {}, {}};
  std::vector<double> lam_tmp(C), gdt(C), gdt_tmp(C);
  const ScrapResult r = scrap_resolve_collisions(op, sep, max_allowable_overlap, max_iters, lam, lam_tmp.data(),
                                                 gdt.data(), gdt_tmp.data());
  for (size_t i = 0; i < C; ++i) g[i] = sep[i] + gdt[i];
  *res = r.max_abs_projected_sep;
  *ite_count = r.ite_count;
  *max_speed = r.max_speed;
}

// build extension (parity unpinned): frictional cone complementarity solve, serial
void o_solve_friction_contact(size_t C, size_t N, const int32_t* pairs, const double* normal, const double* ra,
                              const double* rb, const double* mt, const double* mr, double dt, const double* sep,
                              double mu, unsigned max_iters, double tol, double* p, double* g, unsigned* num_iters,
                              double* res, int* converged) {
  FrictionOp op{pairs, normal, ra, rb, mt, mr, sep, dt, C, N, {}, {}, {}, {}};
  const SolveResult r = solve_friction_contact(op, mu, max_iters, tol, p, g);
  *num_iters = r.num_iters;
  *res = r.residual;
  *converged = r.converged;
}
void o_solve_friction_contact_apgd(size_t C, size_t N, const int32_t* pairs, const double* normal, const double* ra,
                                   const double* rb, const double* mt, const double* mr, double dt, const double* sep,
                                   double mu, unsigned max_iters, double tol, double* p, double* g, unsigned* num_iters,
                                   double* res, int* converged) {
  FrictionOp op{pairs, normal, ra, rb, mt, mr, sep, dt, C, N, {}, {}, {}, {}};
  const SolveResult r = solve_friction_contact_apgd(op, mu, max_iters, tol, p, g);
  *num_iters = r.num_iters;
  *res = r.residual;
  *converged = r.converged;
}
void o_project_cone(size_t n, const double* v, const double* nrm, double mu, double* out) {
  for (size_t i = 0; i < n; ++i) st3(out, i, project_cone(ld3(v, i), ld3(nrm, i), mu));
}

void o_scrap_resolve_collisions(size_t C, size_t N, const int32_t* pairs, const double* normal, const double* ra,
                                const double* rb, const double* mt, const double* mr, double dt, const double* sep,
                                double max_allowable_overlap, int max_iters, double* lam, double* g,
                                double* res, int* ite_count, double* max_speed) {
  ContactOp op{pairs, normal, ra, rb, mt, mr, dt, C, N, {}, {}, {}, {}, {}, {}};
  std::vector<double> lam_tmp(C), gdt(C), gdt_tmp(C);
  const ScrapResult r = scrap_resolve_collisions(op, sep, max_allowable_overlap, max_iters, lam, lam_tmp.data(),
                                                 gdt.data(), gdt_tmp.data());
  for (size_t i = 0; i < C; ++i) g[i] = sep[i] + gdt[i];
  *res = r.max_abs_projected_sep;
  *ite_count = r.ite_count;
  *max_speed = r.max_speed;
}

// OpenMP version of the same unfused BBPGD (CPU baseline): identical kernel structure -- projection pass, operator
// apply as scatter(atomics)/mobility/gather, +q pass, residual reduce, two BB reduces, two copies -- each its own
// parallel loop, exactly what PGDStrategy::iterate (convex.hpp:638-666) launches on Kokkos-OpenMP.
void o_solve_cqpp_contact_mt(size_t C, size_t N, const int32_t* pairs, const double* normal, const double* ra,
                             const double* rb, const double* mt, const double* mr, double dt, const double* q,
                             int kind, double lo, double hi, int resid_kind, unsigned max_iters, double tol,
                             double* x, double* g, double* x_tmp, double* g_tmp, unsigned* num_iters, double* res_out,
                             int* converged_out) {
  const Space sp{kind, lo, hi};
  const bool rot = (ra && rb && mr);
  std::vector<double> F(3 * N), T(rot ? 3 * N : 0), U(3 * N), W(rot ? 3 * N : 0);
  auto apply = [&](const double* xin, double* yout) {
#pragma omp parallel for
    for (size_t b = 0; b < 3 * N; ++b) {
      F[b] = 0.0;
      if (rot) T[b] = 0.0;
    }
#pragma omp parallel for
    for (size_t c = 0; c < C; ++c) {
      const int32_t i = pairs[2 * c], j = pairs[2 * c + 1];
      const double lam = xin[c];
      const V3 f{lam * normal[3 * c], lam * normal[3 * c + 1], lam * normal[3 * c + 2]};
      for (int k = 0; k < 3; ++k) {
#pragma omp atomic
        F[3 * i + k] += -f[k];
#pragma omp atomic
        F[3 * j + k] += f[k];
      }
      if (rot) {
        const V3 ta = cross(ld3(ra, c), f), tb = cross(ld3(rb, c), f);
        for (int k = 0; k < 3; ++k) {
#pragma omp atomic
          T[3 * i + k] += -ta[k];
#pragma omp atomic
          T[3 * j + k] += tb[k];
        }
      }
    }
#pragma omp parallel for
    for (size_t b = 0; b < N; ++b)
      for (int k = 0; k < 3; ++k) {
        U[3 * b + k] = mt[b] * F[3 * b + k];
        if (rot) W[3 * b + k] = mr[b] * T[3 * b + k];
      }
#pragma omp parallel for
    for (size_t c = 0; c < C; ++c) {
      const int32_t i = pairs[2 * c], j = pairs[2 * c + 1];
      const V3 n = ld3(normal, c);
      V3 vi = ld3(U.data(), i), vj = ld3(U.data(), j);
      if (rot) {
        vi = vi + cross(ld3(W.data(), i), ld3(ra, c));
        vj = vj + cross(ld3(W.data(), j), ld3(rb, c));
      }
      const double sdot = -n.x * (vi.x - vj.x) - n.y * (vi.y - vj.y) - n.z * (vi.z - vj.z);
      yout[c] = dt * sdot;
    }
  };
  auto resid = [&](const double* xx, const double* gg) {
    double mx = std::numeric_limits<double>::lowest();
    if (resid_kind == kProjectedGradient) {
#pragma omp parallel for reduction(max : mx)
      for (size_t i = 0; i < C; ++i) {
        const double v = (xx[i] < kZeroTol) ? std::max(0.0, gg[i]) : std::fabs(gg[i]);
        if (v > mx) mx = v;
      }
      return mx;
    }
#pragma omp parallel for reduction(max : mx)
    for (size_t i = 0; i < C; ++i) {
      const double v = std::fabs(xx[i] - sp.project(xx[i] - 1e-6 * gg[i]));
      if (v > mx) mx = v;
    }
    return mx / 1e-6;
  };
  auto copy = [&](const double* s, double* d) {
#pragma omp parallel for
    for (size_t i = 0; i < C; ++i) d[i] = s[i];
  };
  auto addq = [&](double* gg) {
#pragma omp parallel for
    for (size_t i = 0; i < C; ++i) gg[i] = 1.0 * q[i] + 1.0 * gg[i];
  };

  copy(x, x_tmp);
  apply(x_tmp, g_tmp);
  addq(g_tmp);
  double res = resid(x_tmp, g_tmp);
  double step = 1.0 / res;
  unsigned iter = 0;
  bool converged = res <= tol;
  if (converged) copy(g_tmp, g);
  while (!(converged || iter >= max_iters)) {
    const double beta = -step;
    const bool bz = std::fabs(beta) < kZeroTol;
#pragma omp parallel for
    for (size_t i = 0; i < C; ++i) x[i] = sp.project(bz ? 1.0 * x_tmp[i] : 1.0 * x_tmp[i] + beta * g_tmp[i]);
    apply(x, g);
    addq(g);
    res = resid(x, g);
    if (res <= tol) {
      converged = true;
      break;
    }
    double num = 0, den = 0;
#pragma omp parallel for reduction(+ : num)
    for (size_t i = 0; i < C; ++i) {
      const double d = x[i] - x_tmp[i];
      num += d * d;
    }
#pragma omp parallel for reduction(+ : den)
    for (size_t i = 0; i < C; ++i) den += (x[i] - x_tmp[i]) * (g[i] - g_tmp[i]);
    constexpr double eps = kZeroTol * 10;
    den += eps * (std::fabs(den) < eps);
    step = num / den;
    copy(x, x_tmp);
    copy(g, g_tmp);
    ++iter;
  }
  *num_iters = iter;
  *res_out = res;
  *converged_out = converged ? 1 : 0;
}

// ---- zmorton -------------------------------------------------------------------------------------------------------
int o_float_exp_f64(double x) { return float_exp(to_uint(x)); }
int o_float_exp_f32(float x) { return float_exp(to_uint(x)); }
uint64_t o_float_sig_f64(double x) { return float_sig(to_uint(x)); }
uint32_t o_float_sig_f32(float x) { return float_sig(to_uint(x)); }
int o_uint_log_base2(uint64_t x) { return uint_log_base2(x); }
int o_float_xor_msb_f64(double p, double q) { return float_xor_msb(p, q); }
int o_float_xor_msb_f32(float p, float q) { return float_xor_msb(p, q); }
int o_zorder_less_f64(const double* p, const double* q, int d) { return zorder_less(p, q, d) ? 1 : 0; }
int o_zorder_less_f32(const float* p, const float* q, int d) { return zorder_less(p, q, d) ? 1 : 0; }
int o_zmorton_less(const double* p, const double* q) { return zmorton_less(p, q) ? 1 : 0; }
// argsort of n points of dimension d with zorder_knn::Less (std::sort as UnitTestZMorton.cpp:181-194 does)
void o_zorder_argsort_f64(size_t n, int d, const double* pts, int64_t* order) {
  std::iota(order, order + n, int64_t{0});
  std::sort(order, order + n, [&](int64_t a, int64_t b) { return zorder_less(pts + a * d, pts + b * d, d); });
}
void o_zorder_argsort_f32(size_t n, int d, const float* pts, int64_t* order) {
  std::iota(order, order + n, int64_t{0});
  std::sort(order, order + n, [&](int64_t a, int64_t b) { return zorder_less(pts + a * d, pts + b * d, d); });
}

// ---- hilbert --------------------------------------------------------------------------------------------------------
// positions must hold ns^3 points (ns = smallest power of two >= 2 with ns^3 >= num_points), directors ns^3 - 1.
size_t o_hilbert_num_positions(size_t num_points) {
  size_t ns = 2;
  while (ns * ns * ns < num_points) ns *= 2;
  return ns * ns * ns;
}
void o_hilbert_positions_and_directors(size_t num_points, const double* orientation, double side, double* positions,
                                       double* directors) {
  std::vector<V3> pos, dir;
  create_hilbert_positions_and_directors(num_points, {orientation[0], orientation[1], orientation[2]}, side, pos,
                                         dir);
  for (size_t i = 0; i < pos.size(); ++i) st3(positions, i, pos[i]);
  for (size_t i = 0; i < dir.size(); ++i) st3(directors, i, dir[i]);
}

// raw recursion as UnitTestHilbert.cpp:48-66 calls it: s lattice points per side, unit axes dr1, dr2, dr3
void o_hilbert_3d(size_t s, const double* cur, const double* dr1, const double* dr2, const double* dr3,
                  double* positions) {
  std::vector<V3> pos(s * s * s);
  hilbert_3d(s, 0, pos, ld3(cur, 0), ld3(dr1, 0), ld3(dr2, 0), ld3(dr3, 0));
  for (size_t i = 0; i < pos.size(); ++i) st3(positions, i, pos[i]);
}

}  // extern "C"
