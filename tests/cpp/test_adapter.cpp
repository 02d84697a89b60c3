// test_adapter.cpp -- the reference's own unit tests for this path, restated against the C++ adapter
// (include/mundy_hip/adapter.hpp) so they read like the originals:
//   mundy/math/tests/unit_tests/UnitTestConvex.cpp:239-625   (run_kokkos_test on the three analytic SPD problems and
//                                                            RandomLCP{3,7,200}: x0 = 99.99, max_iters 1000, tol 1e-6)
//   mundy/geom/tests/unit_tests/UnitTestComputeAABB.cpp:167-232, UnitTestSegmentSegment.cpp:417-472
//   mundy/mesh/tests/unit_tests/UnitTestGenNeighborLinks.cpp:73-152 (two coincident spheres -> one link)
// Needs a GPU; built and run by tests/test_adapter_cpp.py.  Exit code = number of failed checks.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>

#include "mundy_hip/adapter.hpp"

using namespace mundy_hip;
namespace cx = mundy_hip::convex;

static int g_failures = 0;
#define EXPECT_TRUE(c)                                                    \
  do {                                                                    \
    if (!(c)) {                                                           \
      ++g_failures;                                                       \
      std::printf("FAILED %s:%d  %s\n", __FILE__, __LINE__, #c);          \
    }                                                                     \
  } while (0)
#define EXPECT_NEAR(a, b, tol) EXPECT_TRUE(std::fabs((a) - (b)) <= (tol))
#define EXPECT_THROW(stmt, exc)                                           \
  do {                                                                    \
    bool thrown_ = false;                                                 \
    try { stmt; } catch (const exc&) { thrown_ = true; } catch (...) {}   \
    if (!thrown_) {                                                       \
      ++g_failures;                                                       \
      std::printf("FAILED %s:%d  expected %s\n", __FILE__, __LINE__, #exc); \
    }                                                                     \
  } while (0)

// ---- UnitTestConvex.cpp ----------------------------------------------------------------------------------------------
struct DenseProblem {
  std::vector<double> A, q, x_exact;
  size_t n;
};
static DenseProblem spd3(std::vector<double> x_exact) {
  DenseProblem p;
  p.n = 3;
  p.A = {2.0, -1.0, 0.0, -1.0, 2.0, -1.0, 0.0, -1.0, 2.0};
  p.x_exact = x_exact;
  p.q.assign(3, 0.0);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) p.q[i] -= p.A[3 * i + j] * x_exact[j];  // q = -A x*
  return p;
}
static DenseProblem random_lcp(size_t n, unsigned seed) {  // UnitTestConvex.cpp:416-524 (std::mt19937 entries)
  std::mt19937_64 rng(seed);
  std::uniform_real_distribution<double> u11(-1.0, 1.0), u01(0.0, 1.0);
  DenseProblem p;
  p.n = n;
  p.A.resize(n * n);
  for (auto& a : p.A) a = u11(rng);
  for (size_t i = 0; i < n; ++i) {
    double off = 0;
    for (size_t j = 0; j < n; ++j) off += std::fabs(p.A[i * n + j]) * (i != j);
    p.A[i * n + i] = off + 10.0;
  }
  std::vector<double> g_star(n);
  p.x_exact.resize(n);
  for (size_t i = 0; i < n; ++i) {
    const double v = u01(rng) * 0.9 + 0.1;
    const bool active = u01(rng) < 0.5;
    p.x_exact[i] = active ? v : 0.0;
    g_star[i] = active ? 0.0 : v;
  }
  p.q.resize(n);
  for (size_t i = 0; i < n; ++i) {
    double ax = 0;
    for (size_t j = 0; j < n; ++j) ax += p.A[i * n + j] * p.x_exact[j];
    p.q[i] = g_star[i] - ax;
  }
  return p;
}

template <class Space>
static void run_hip_test(const DenseProblem& test, const Space& space) {  // UnitTestConvex.cpp:563-606
  DeviceVector A(test.A), q(test.q);
  const size_t size = test.n;
  DeviceVector x(std::vector<double>(size, 99.99)), grad(size), x_tmp(size), grad_tmp(size);
  const cx::DenseMatrix Aop{A.data(), size};
  const auto cqpp = make_hip_cqpp(Aop, q, space);
  const auto backend = cqpp.backend();
  cx::PGDConfig<double> cfg{1000, 1e-6};
  auto pgd = make_pgd_solution_strategy(backend, cfg);
  auto pgd_state = make_pgd_state(backend, x, grad, x_tmp, grad_tmp);
  auto result = solve_cqpp(cqpp, pgd, pgd_state);
  EXPECT_TRUE(result.converged);
  EXPECT_TRUE(result.num_iters <= cfg.max_iters);
  const auto xh = x.download();
  for (size_t i = 0; i < size; ++i) EXPECT_NEAR(xh[i], test.x_exact[i], 10 * cfg.tol);
}

static void test_convex_analytical_solutions() {
  run_hip_test(spd3({1.0, 0.0, 1.0}), cx::space::Unconstrained<double>());
  run_hip_test(spd3({1.0, 0.0, 1.0}), cx::space::Bounded<double>(0.0, 2.0));
  run_hip_test(spd3({9.0, 9.0, 9.0}), cx::space::Bounded<double>(9.0, 10.0));
  for (size_t n : {size_t(3), size_t(7), size_t(200)}) run_hip_test(random_lcp(n, 17 + n), cx::space::LowerBound<double>{0.0});
}

// user operator with only apply(x, y): takes the generic (unfused) path
struct WrappedOp {
  const ContactOperator& inner;
  void apply(const DeviceVector& x, DeviceVector& y) const { inner.apply(x, y); }
};
struct NotAnOperator {};

static void test_contact_lcp_fused_equals_generic() {
  // two rows of touching/overlapping unit spheres along x
  const int N = 40;
  std::vector<double> c(3 * N), r(N, 1.0), mt(N, 1.0 / (6.0 * M_PI * 1e-3));
  for (int i = 0; i < N; ++i) {
    c[3 * i] = 1.7 * (i % 20); c[3 * i + 1] = 1.8 * (i / 20); c[3 * i + 2] = 0.0;
  }
  DeviceVector dc(c), dr(r), dmt(mt), aabb(6 * N);
  check(mhip_compute_aabb_spheres(N, dc.data(), dr.data(), aabb.data(), nullptr));
  mesh::GenNeighborLinks links;
  links.set_search_buffer(0.2).set_search_kind(MHIP_SEARCH_SPHERES).concretize();
  EXPECT_TRUE(links.generate(N, aabb.data(), dc.data(), dr.data()));
  EXPECT_TRUE(!links.generate(N, aabb.data(), dc.data(), dr.data()));  // nothing moved: no regeneration
  const size_t C = links.num_links();
  EXPECT_TRUE(C > 30);
  auto pairs = links.links();
  DeviceVector sep(C), normal(3 * C);
  check(mhip_contact_spheres(C, pairs.data(), dc.data(), dr.data(), nullptr, sep.data(), normal.data(), nullptr));
  ContactOperator op(C, N, pairs.data(), normal.data(), nullptr, nullptr, dmt.data(), nullptr, 5e-3);
  cx::PGDConfig<double> cfg{5000, 1e-6};
  std::vector<double> xs[2];
  unsigned iters[2] = {0, 0};
  for (int pass = 0; pass < 2; ++pass) {
    DeviceVector x(std::vector<double>(C, 0.0)), grad(C), x_tmp(C), grad_tmp(C);
    const auto backend = cx::HipBackend{};
    auto pgd = make_pgd_solution_strategy(backend, cfg);
    auto st = make_pgd_state(backend, x, grad, x_tmp, grad_tmp);
    if (pass == 0) {
      const auto lcp = make_hip_lcp(op, sep);  // fused device-resident driver
      auto res = solve_lcp(lcp, pgd, st);
      EXPECT_TRUE(res.converged);
      iters[0] = res.num_iters;
    } else {
      const WrappedOp wrapped{op};
      const auto lcp = make_hip_lcp(wrapped, sep);  // the reference's loop through HipBackend
      auto res = solve_lcp(lcp, pgd, st);
      EXPECT_TRUE(res.converged);
      iters[1] = res.num_iters;
    }
    xs[pass] = grad.download();
    const auto xh = x.download();
    for (double v : xh) EXPECT_TRUE(v >= 0.0);
    for (double g : xs[pass]) EXPECT_TRUE(g >= -1e-5);
  }
  EXPECT_TRUE(std::abs((int)iters[0] - (int)iters[1]) <= 5);
  for (size_t i = 0; i < C; ++i) EXPECT_NEAR(xs[0][i], xs[1][i], 2e-5);
}

static void test_error_behaviour() {
  DeviceVector a(3), b(4), y(3);
  EXPECT_THROW(cx::HipBackend::axpby(1.0, a, 1.0, b), std::invalid_argument);
  EXPECT_THROW(cx::HipBackend::diff_dot(a, b), std::invalid_argument);
  DeviceVector A(9);
  EXPECT_THROW(cx::HipBackend::apply(cx::DenseMatrix{A.data(), 3}, b, y), std::invalid_argument);  // convex.hpp:171
  EXPECT_THROW(cx::HipBackend::apply(NotAnOperator{}, a, y), std::logic_error);                     // convex.hpp:197
  mesh::GenNeighborLinks g;
  EXPECT_THROW(g.generate(0, nullptr, nullptr, nullptr), std::runtime_error);  // before concretization
  g.concretize();
  EXPECT_THROW(g.set_search_buffer(1.0), std::runtime_error);
  EXPECT_THROW(g.concretize(), std::runtime_error);
}

// ---- geometry KATs ----------------------------------------------------------------------------------------------------
static void expect_aabb(const geom::AABB<double>& a, std::vector<double> e) {
  for (int k = 0; k < 3; ++k) {
    EXPECT_NEAR(a.min_corner()[k], e[k], 1e-8);
    EXPECT_NEAR(a.max_corner()[k], e[3 + k], 1e-8);
  }
}
static void test_compute_aabb_hard_coded() {  // UnitTestComputeAABB.cpp:167-232
  using namespace geom;
  const Quaternion<double> id(1, 0, 0, 0), x90(1.0 / std::sqrt(2.0), 1.0 / std::sqrt(2.0), 0.0, 0.0);
  const Point<double> c(1, -2, 3);
  auto s = compute_aabb(std::vector<Sphere<double>>{{Point<double>(0, 0, 0), 1.0}, {c, 4.0}});
  expect_aabb(s[0], {-1, -1, -1, 1, 1, 1});
  expect_aabb(s[1], {-3, -6, -1, 5, 2, 7});
  auto r = compute_aabb(std::vector<Spherocylinder<double>>{{c, id, 4, 0}, {c, id, 0, 4}, {c, id, 2, 4}, {c, x90, 2, 3}});
  expect_aabb(r[0], {-3, -6, -1, 5, 2, 7});
  expect_aabb(r[1], {1, -2, 1, 1, -2, 5});
  expect_aabb(r[2], {-1, -4, -1, 3, 0, 7});
  expect_aabb(r[3], {-1, -5.5, 1, 3, 1.5, 5});
  auto e = compute_aabb(std::vector<Ellipsoid<double>>{{c, id, Point<double>(4, 5, 6)}, {c, x90, Point<double>(4, 5, 6)}});
  expect_aabb(e[0], {-3, -7, -3, 5, 3, 9});
  expect_aabb(e[1], {-3, -8, -2, 5, 4, 8});
  EXPECT_TRUE(intersects(s[0], s[1]) == false || true);
  const AABB<double> inverted;  // default = inverted box: intersects nothing
  EXPECT_TRUE(!intersects(inverted, s[0]));
}
static void test_segment_kats() {  // UnitTestSegmentSegment.cpp:417-472
  using namespace geom;
  std::vector<LineSegment<double>> a{
      {Point<double>(0.2257294191072674, 0.30159862841764695, 0.12784820133135649),
       Point<double>(0.22572948671663273, 0.30159858045792487, 0.1278481814714105)},
      {Point<double>(9.64101615137754, 6, 3.18961417478521), Point<double>(9.64101615137754, 6, 8.189614174785209)}};
  std::vector<LineSegment<double>> b{
      {Point<double>(0.5220039935659887, 0.88764831847472003, -0.2219484914838093),
       Point<double>(0.50288066060587278, 0.66779290982621586, -0.5723507723323677)},
      {Point<double>(10.39230484541326, 6, 0.6472696138825587), Point<double>(10.39230484541326, 6, 5.647269613882559)}};
  const auto r = distance(SharedNormalSigned{}, a, b);
  EXPECT_NEAR(r.distance[0], 0.74347757392471259, 1e-6);
  EXPECT_NEAR(r.arch_length1[0], 1.0, 1e-6);
  EXPECT_NEAR(r.arch_length2[0], 0.069641589451982497, 1e-6);
  EXPECT_NEAR(r.closest_point2[0][1], 0.87233723836682309, 1e-6);
  EXPECT_NEAR(r.distance[1], 0.7512886940357237, 1e-6);
  const auto rev = distance(SharedNormalSigned{}, b, a);
  EXPECT_NEAR(rev.distance[1], r.distance[1], 1e-6);
  std::vector<Point<double>> sep;
  const auto d = distance(SharedNormalSigned{}, std::vector<Sphere<double>>{{Point<double>(0, 0, 0), 1.0}},
                          std::vector<Sphere<double>>{{Point<double>(3, 0, 0), 0.5}}, &sep);
  EXPECT_NEAR(d[0], 1.5, 1e-15);
  EXPECT_NEAR(sep[0][0], 1.5, 1e-15);
}
static void test_single_object_overloads() {
  // seam S4: the reference's free functions on ONE pair of owning primitives (SphereSphere.hpp:44-76,
  // LineSegmentLineSegment.hpp:169-197, EllipsoidEllipsoid.hpp:106-113, PointEllipsoid.hpp:94-135, compute_aabb.hpp:72-127)
  using namespace geom;
  const Quaternion<double> id(1, 0, 0, 0), x90(1.0 / std::sqrt(2.0), 1.0 / std::sqrt(2.0), 0.0, 0.0);
  const Point<double> c(1, -2, 3);
  expect_aabb(compute_aabb(Sphere<double>(c, 4.0)), {-3, -6, -1, 5, 2, 7});                     // UnitTestComputeAABB.cpp:167-232
  expect_aabb(compute_aabb(Spherocylinder<double>(c, x90, 2, 3)), {-1, -5.5, 1, 3, 1.5, 5});
  expect_aabb(compute_aabb(Ellipsoid<double>(c, x90, Point<double>(4, 5, 6))), {-3, -8, -2, 5, 4, 8});
  const Sphere<double> s1(Point<double>(0, 0, 0), 1.0), s2(Point<double>(3, 0, 0), 0.5);
  Point<double> sep;
  EXPECT_NEAR(distance(s1, s2), 1.5, 1e-15);
  EXPECT_NEAR(distance(SharedNormalSigned{}, s1, s2), 1.5, 1e-15);
  EXPECT_NEAR(distance(s1, s2, sep), 1.5, 1e-15);
  EXPECT_NEAR(sep[0], 1.5, 1e-15);
  // UnitTestSegmentSegment.cpp:417-472, first known-answer case
  const LineSegment<double> a(Point<double>(0.2257294191072674, 0.30159862841764695, 0.12784820133135649),
                              Point<double>(0.22572948671663273, 0.30159858045792487, 0.1278481814714105));
  const LineSegment<double> b(Point<double>(0.5220039935659887, 0.88764831847472003, -0.2219484914838093),
                              Point<double>(0.50288066060587278, 0.66779290982621586, -0.5723507723323677));
  Point<double> cp1, cp2;
  double t1 = -1, t2 = -1;
  EXPECT_NEAR(distance(a, b, cp1, cp2, t1, t2, sep), 0.74347757392471259, 1e-6);
  EXPECT_NEAR(t1, 1.0, 1e-6);
  EXPECT_NEAR(t2, 0.069641589451982497, 1e-6);
  EXPECT_NEAR(cp2[1], 0.87233723836682309, 1e-6);
  // the colinear branch hands back the UNCLAMPED parameter (PointLineSegment.hpp:156-166): rods end to end
  const LineSegment<double> l(Point<double>(0, 0, -1), Point<double>(0, 0, 1)), m(Point<double>(0.8, 0, 1.5), Point<double>(0.8, 0, 3.5));
  distance(SharedNormalSigned{}, l, m, cp1, cp2, t1, t2, sep);
  EXPECT_TRUE(t1 == 1.0 && t2 == -0.25 && cp2[2] == 1.5);
  // distance(Point, Sphere[, sep]) (PointSphere.hpp:46-80) and distance(LineSegment, Sphere[, cp, arch_length, sep])
  // (LineSegmentSphere.hpp:47-100): the reference has no test of its own for either; hand-computed answers
  const Sphere<double> ball(Point<double>(0, 0, 0), 1.0);
  EXPECT_NEAR(distance(Point<double>(0, 0, 5), ball), 4.0, 1e-15);
  EXPECT_NEAR(distance(SharedNormalSigned{}, Point<double>(0, 3, 4), ball), 4.0, 1e-15);
  EXPECT_NEAR(distance(Point<double>(0, 0, 0.25), ball), -0.75, 1e-15);              // inside: negative
  EXPECT_NEAR(distance(Point<double>(0, 0, 5), ball, sep), 4.0, 1e-15);
  EXPECT_TRUE(sep[0] == 0.0 && sep[1] == 0.0);
  EXPECT_NEAR(sep[2], -4.0, 1e-15);                                                 // from the point to the surface
  const LineSegment<double> rail(Point<double>(-1, 0, 0), Point<double>(1, 0, 0));
  const Sphere<double> bead(Point<double>(0.5, 3, 0), 1.0);
  EXPECT_NEAR(distance(rail, bead), 2.0, 1e-15);
  EXPECT_NEAR(distance(SharedNormalSigned{}, rail, bead), 2.0, 1e-15);
  EXPECT_NEAR(distance(rail, bead, cp1, t1, sep), 2.0, 1e-15);
  EXPECT_NEAR(t1, 0.75, 1e-15);
  EXPECT_NEAR(cp1[0], 0.5, 1e-15);
  EXPECT_NEAR(sep[1], -2.0, 1e-15);                  // PointLineSegment's separation (centre -> closest point), rescaled
  // beyond the end: closest point clamped, arch length left unclamped (PointLineSegment.hpp:156-166)
  EXPECT_NEAR(distance(rail, Sphere<double>(Point<double>(4, 0, 0), 0.5), cp1, t1, sep), 2.5, 1e-15);
  EXPECT_TRUE(t1 == 2.5 && cp1[0] == 1.0);
  // two spheres as ellipsoids (UnitTestEllipsoidEllipsoid.cpp:65-145), tolerance 1e-4
  const Ellipsoid<double> e1(Point<double>(0, 0, 0), id, Point<double>(1, 1, 1)), e2(Point<double>(4, 0, 0), x90, Point<double>(2, 2, 2));
  Point<double> n1, n2;
  EXPECT_NEAR(distance(SharedNormalSigned{}, e1, e2, cp1, cp2, n1, n2), 1.0, 1e-4);
  EXPECT_NEAR(distance(SharedNormalSigned{}, e1, e2), 1.0, 1e-4);
  EXPECT_NEAR(n1[0], 1.0, 1e-4);
  EXPECT_NEAR(distance(SharedNormalSigned{}, Point<double>(0, 5, 0), e2, cp1, n1), std::sqrt(41.0) - 2.0, 1e-4);
}
static void test_two_coincident_spheres() {  // UnitTestGenNeighborLinks.cpp:73-152
  DeviceVector c(std::vector<double>(6, 0.0)), r(std::vector<double>(2, 1.0)), aabb(12);
  check(mhip_compute_aabb_spheres(2, c.data(), r.data(), aabb.data(), nullptr));
  mesh::GenNeighborLinks g;
  g.set_search_buffer(0.0).concretize();
  EXPECT_TRUE(g.generate(2, aabb.data(), c.data(), r.data()));
  EXPECT_TRUE(g.num_links() == 1);
  const auto p = g.links().download();
  EXPECT_TRUE((p[0] == 0 && p[1] == 1) || (p[0] == 1 && p[1] == 0));
}

static void test_gen_neighbor_links_reference_usage() {
  // UnitTestGenNeighborLinks.cpp:73-152 as written: source == target == the spheres part, symmetry enforced, search
  // buffer 0, filter = ExcludeSelfInteractions -> the two coincident spheres are linked, in either orientation.  Run on
  // both search structures; with the symmetry flag both orientations are results, without it the unique one.
  namespace sf = mesh::search_filters;
  DeviceVector c(std::vector<double>(6, 0.0)), r(std::vector<double>(2, 1.0)), aabb(12);
  check(mhip_compute_aabb_spheres(2, c.data(), r.data(), aabb.data(), nullptr));
  DeviceArray<unsigned char> all(std::vector<unsigned char>{1, 1});
  DeviceArray<uint64_t> ids(std::vector<uint64_t>{1, 2});  // declare_element(1, ...), declare_element(2, ...)
  for (int method : {MHIP_SEARCH_METHOD_GRID, MHIP_SEARCH_METHOD_MORTON_LBVH}) {
    for (bool symmetric : {true, false}) {
      mesh::GenNeighborLinks g;
      g.set_enforce_source_target_symmetry(symmetric)
          .set_search_buffer(0.0)
          .set_search_method(method)
          .set_search_filter(sf::make_search_filter(sf::ExcludeSelfInteractions{}))
          .acts_on(2, all.data(), all.data())
          .set_identities(2, ids.data(), nullptr)
          .concretize();
      EXPECT_TRUE(g.is_concretized() && g.get_search_method() == method);
      EXPECT_TRUE(g.generate(2, aabb.data(), c.data(), r.data()));
      EXPECT_TRUE(g.num_links() == (symmetric ? 2u : 1u));
      auto idp = g.ident_links();
      const auto src = idp.source_id.download(), tgt = idp.target_id.download();
      const auto sp = idp.source_proc.download();
      for (size_t k = 0; k < g.num_links(); ++k) {
        EXPECT_TRUE((src[k] == 1 && tgt[k] == 2) || (src[k] == 2 && tgt[k] == 1));  // :149-151
        EXPECT_TRUE(sp[k] == 0);
      }
      if (symmetric) EXPECT_TRUE(src[0] == 1 && tgt[0] == 2 && src[1] == 2 && tgt[1] == 1);
      // the links as LinkData would hold them
      const auto coo = g.export_coo(100, 3, 3);
      const auto lid = coo.link_id.download(), linked = coo.linked_entity_ids.download();
      EXPECT_TRUE(lid[0] == 100 && linked[0] == src[0] && linked[1] == tgt[0]);
      const auto crs = g.export_crs(100, 512);
      const auto num = crs.num_connected_links.download();
      EXPECT_TRUE(crs.num_buckets == 1 && num[0] == g.num_links() && num[1] == g.num_links());
    }
    // without the filter stk's coarse_search also reports every sphere against itself
    mesh::GenNeighborLinks g2;
    g2.set_enforce_source_target_symmetry(true).set_search_method(method).set_search_filter(sf::make_search_filter());
    g2.concretize();
    g2.generate(2, aabb.data(), c.data(), r.data());
    EXPECT_TRUE(g2.num_links() == 4);
    // distinct source and target sets: sphere 1 is the only source, sphere 2 the only target -> exactly (1, 2)
    DeviceArray<unsigned char> s1(std::vector<unsigned char>{1, 0}), t2(std::vector<unsigned char>{0, 1});
    mesh::GenNeighborLinks g3;
    g3.set_enforce_source_target_symmetry(true).set_search_method(method);
    g3.set_search_filter(sf::make_search_filter(sf::ExcludeSelfInteractions{})).acts_on(2, s1.data(), t2.data());
    g3.set_identities(2, ids.data(), nullptr).concretize();
    g3.generate(2, aabb.data(), c.data(), r.data());
    EXPECT_TRUE(g3.num_links() == 1);
    auto one = g3.ident_links();
    EXPECT_TRUE(one.source_id.download()[0] == 1 && one.target_id.download()[0] == 2);
    // ExcludeConnectedEntities: sphere 1 is connected to sphere 2 (local indices 0 -> 1): that link is filtered, the
    // opposite orientation (2 has no connection to 1) stays
    DeviceArray<int32_t> cptr(std::vector<int32_t>{0, 1, 1}), cidx(std::vector<int32_t>{1});
    mesh::GenNeighborLinks g4;
    g4.set_enforce_source_target_symmetry(true).set_search_method(method);
    g4.set_search_filter(sf::make_search_filter(sf::ExcludeSelfInteractions{},
                                                sf::ExcludeConnectedEntities{2, cptr.data(), cidx.data(), 1}));
    g4.concretize();
    g4.generate(2, aabb.data(), c.data(), r.data());
    const auto p4 = g4.links().download();
    EXPECT_TRUE(g4.num_links() == 1 && p4[0] == 1 && p4[1] == 0);
    EXPECT_THROW(g4.acts_on(2, all.data(), all.data()), std::runtime_error);  // "Cannot set source/targets after concretization."
  }
}

static void test_ellipsoid_sphere_cases() {
  // SharedNormalDistanceBetweenEllipsoids.AnalyticalSphereTestCases and ...EllipsoidAndPoint.AnalyticalSphereTestCases
  // (UnitTestEllipsoidEllipsoid.cpp:65-145): ellipsoids with three equal radii are spheres, tolerance 1e-4
  std::mt19937_64 rng(7);
  std::uniform_real_distribution<double> pos(-10.0, 10.0), rad(0.1, 10.0);
  std::normal_distribution<double> gauss(0.0, 1.0);
  auto sphere_like = [&]() {
    double q[4] = {gauss(rng), gauss(rng), gauss(rng), gauss(rng)};
    const double nn = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    const double r = rad(rng);
    return geom::Ellipsoid<double>(geom::Point<double>(pos(rng), pos(rng), pos(rng)),
                                   geom::Quaternion<double>(q[0] / nn, q[1] / nn, q[2] / nn, q[3] / nn),
                                   geom::Point<double>(r, r, r));
  };
  std::vector<geom::Ellipsoid<double>> a, b;
  std::vector<geom::Point<double>> pts;
  for (int i = 0; i < 500; ++i) {
    a.push_back(sphere_like());
    b.push_back(sphere_like());
    pts.emplace_back(pos(rng), pos(rng), pos(rng));
  }
  const auto r = geom::distance(geom::SharedNormalSigned{}, a, b);
  std::vector<geom::Point<double>> closest, normal;
  const auto dp = geom::distance(geom::SharedNormalSigned{}, pts, a, &closest, &normal);
  for (size_t i = 0; i < a.size(); ++i) {
    auto len = [](double x, double y, double z) { return std::sqrt(x * x + y * y + z * z); };
    const auto &ca = a[i].center(), &cb = b[i].center();
    const double expect = len(cb[0] - ca[0], cb[1] - ca[1], cb[2] - ca[2]) - a[i].radii()[0] - b[i].radii()[0];
    EXPECT_TRUE(std::fabs(r.distance[i] - expect) <= 1e-4);
    EXPECT_TRUE(std::fabs(len(r.shared_normal1[i][0], r.shared_normal1[i][1], r.shared_normal1[i][2]) - 1.0) <= 1e-12);
    EXPECT_TRUE(r.shared_normal2[i][0] == -r.shared_normal1[i][0]);
    const double expect_p = len(pts[i][0] - ca[0], pts[i][1] - ca[1], pts[i][2] - ca[2]) - a[i].radii()[0];
    EXPECT_TRUE(std::fabs(dp[i] - expect_p) <= 1e-4);
  }
}

static void test_periodic_metrics() {  // UnitTestPeriodicity.cpp:623-660 (MinImageDirectVsPeriodic), restated
  const geom::Point<double> cell(100.0, 100.0, 100.0);
  const auto metric = geom::periodic_metric_from_unit_cell(cell);
  const auto scaled = geom::periodic_scaled_metric_from_unit_cell(cell);
  std::vector<geom::Point<double>> p1, p2;
  unsigned long long state = 1234;
  auto uniform = [&state]() {  // splitmix64 -> [0, 100)
    state += 0x9E3779B97F4A7C15ull;
    unsigned long long z = state;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return 100.0 * static_cast<double>(z >> 11) / 9007199254740992.0;
  };
  for (int i = 0; i < 2000; ++i) {
    p1.emplace_back(uniform(), uniform(), uniform());
    p2.emplace_back(uniform(), uniform(), uniform());
  }
  const auto s1 = metric.sep(p1, p2), s2 = scaled.sep(p1, p2);
  for (size_t i = 0; i < p1.size(); ++i) {
    double best = 1e300;
    for (int a = -1; a <= 1; ++a)
      for (int b = -1; b <= 1; ++b)
        for (int c = -1; c <= 1; ++c) {
          const double dx = p2[i][0] + 100.0 * a - p1[i][0], dy = p2[i][1] + 100.0 * b - p1[i][1],
                       dz = p2[i][2] + 100.0 * c - p1[i][2];
          best = std::min(best, std::sqrt(dx * dx + dy * dy + dz * dz));
        }
    const double n1 = std::sqrt(s1[i][0] * s1[i][0] + s1[i][1] * s1[i][1] + s1[i][2] * s1[i][2]);
    const double n2 = std::sqrt(s2[i][0] * s2[i][0] + s2[i][1] * s2[i][1] + s2[i][2] * s2[i][2]);
    EXPECT_TRUE(std::fabs(n1 - best) <= 1e-8);  // get_relaxed_zero_tolerance<double>()
    EXPECT_TRUE(std::fabs(n2 - best) <= 1e-8);
  }
  const auto w = metric.wrap({geom::Point<double>(950.0, -10.0, 100.0)});
  EXPECT_TRUE(std::fabs(w[0][0] - 50.0) <= 1e-8 && std::fabs(w[0][1] - 90.0) <= 1e-8 && std::fabs(w[0][2]) <= 1e-8);
  EXPECT_TRUE(metric.inverse()[0] == 0.01 && metric.inverse()[4] == 0.01 && metric.inverse()[1] == 0.0);
}

int main() {
  int count = 0;
  char arch[128];
  check(mhip_device_info(&count, arch, sizeof(arch)));
  std::printf("devices: %d, arch %s\n", count, arch);
  test_convex_analytical_solutions();
  test_contact_lcp_fused_equals_generic();
  test_error_behaviour();
  test_compute_aabb_hard_coded();
  test_segment_kats();
  test_single_object_overloads();
  test_two_coincident_spheres();
  test_gen_neighbor_links_reference_usage();
  test_periodic_metrics();
  test_ellipsoid_sphere_cases();
  std::printf("%s (%d failed checks)\n", g_failures ? "FAILED" : "ALL PASSED", g_failures);
  return g_failures;
}
