// adapter.hpp -- header-only C++17 host layer over the C ABI of mundy_hip.h that presents MuNDy's own vocabulary, so a
// MuNDy-style timestep loop can switch this path in:
//
//   mundy_hip::geom     Point, Sphere, Spherocylinder, Ellipsoid, LineSegment, AABB (accessor names of
//                       mundy_geom/primitives/*.hpp), compute_aabb(...), distance(SharedNormalSigned{}, a, b, outs...)
//                       as batch overloads over std::vector of primitives (AoS -> SoA pack, one kernel launch)
//   mundy_hip::mesh     GenNeighborLinks builder: set_search_buffer / set_enforce_source_target_symmetry / concretize /
//                       generate (mundy_mesh/GenNeighborLinkers.hpp:294-543)
//   mundy_hip::convex   space::{Unconstrained,LowerBound,UpperBound,Bounded}, HipBackend (the static interface of
//                       convex::KokkosBackend, convex.hpp:141-285), CQPPProblem, LCPProblem, to_cqpp, PGDConfig,
//                       SolveResult, PGDState, PGDStrategy, BBStepStrategy, LinfNormProjected{Diff,Gradient}Residual,
//                       make_*, solve_cqpp, solve_lcp -- same names, argument meaning and error behaviour
//   mundy_hip::ContactOperator   the matrix-free LinearOp with `void apply(x, y) const` (convex.hpp:133-136)
//
// Functors cannot cross a C ABI, so the convex spaces and residual policies are tag types that map to enums; a
// user-defined LinearOp only needs `void apply(const DeviceVector&, DeviceVector&) const` on device vectors.
// Status codes are mapped back to the exception types MUNDY_THROW_REQUIRE throws (throw_assert.hpp:135-203).
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <limits>
#include <memory>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "../mundy_hip.h"

namespace mundy_hip {

// ---- errors ---------------------------------------------------------------------------------------------------------
inline void check(int status) {
  if (status == MHIP_SUCCESS) return;
  const std::string msg = mhip_last_error();
  switch (status) {
    case MHIP_ERR_INVALID_ARGUMENT: throw std::invalid_argument(msg);
    case MHIP_ERR_LOGIC: throw std::logic_error(msg);
    default: throw std::runtime_error(msg);
  }
}

// ---- device memory ----------------------------------------------------------------------------------------------------
template <class T>
class DeviceArray {
 public:
  DeviceArray() = default;
  explicit DeviceArray(size_t n) : n_(n) {
    void* p = nullptr;
    check(mhip_malloc(&p, n * sizeof(T)));
    p_ = static_cast<T*>(p);
  }
  explicit DeviceArray(const std::vector<T>& host) : DeviceArray(host.size()) { upload(host); }
  DeviceArray(const DeviceArray&) = delete;
  DeviceArray& operator=(const DeviceArray&) = delete;
  DeviceArray(DeviceArray&& o) noexcept : p_(o.p_), n_(o.n_) { o.p_ = nullptr; o.n_ = 0; }
  DeviceArray& operator=(DeviceArray&& o) noexcept {
    if (this != &o) {
      release();
      p_ = o.p_; n_ = o.n_; o.p_ = nullptr; o.n_ = 0;
    }
    return *this;
  }
  ~DeviceArray() { release(); }
  size_t size() const { return n_; }
  size_t extent(int) const { return n_; }  // Kokkos::View spelling
  T* data() { return p_; }
  const T* data() const { return p_; }
  void upload(const std::vector<T>& host, mhip_stream_t s = nullptr) {
    if (host.size() != n_) throw std::invalid_argument("DeviceArray::upload: size mismatch");
    check(mhip_memcpy_h2d(p_, host.data(), n_ * sizeof(T), s));
  }
  std::vector<T> download(mhip_stream_t s = nullptr) const {
    std::vector<T> host(n_);
    check(mhip_memcpy_d2h(host.data(), p_, n_ * sizeof(T), s));
    return host;
  }

 private:
  void release() {
    if (p_) mhip_free(p_);
    p_ = nullptr;
  }
  T* p_ = nullptr;
  size_t n_ = 0;
};
using DeviceVector = DeviceArray<double>;

// ---- geometry -----------------------------------------------------------------------------------------------------------
namespace geom {

struct SharedNormalSigned {};  // mundy_geom/distance/Types.hpp
struct Euclidean {};

template <class S = double>
struct Point {
  S v[3]{S(0), S(0), S(0)};
  Point() = default;
  Point(S x, S y, S z) : v{x, y, z} {}
  S& operator[](int i) { return v[i]; }
  const S& operator[](int i) const { return v[i]; }
};
template <class S = double>
struct Quaternion {
  S q[4]{S(1), S(0), S(0), S(0)};  // (w, x, y, z); default identity (Spherocylinder.hpp default)
  Quaternion() = default;
  Quaternion(S w, S x, S y, S z) : q{w, x, y, z} {}
  S& w() { return q[0]; }
  S& x() { return q[1]; }
  S& y() { return q[2]; }
  S& z() { return q[3]; }
  const S& operator[](int i) const { return q[i]; }
};
template <class S = double>
class Sphere {  // primitives/Sphere.hpp:40-231; default radius -1
 public:
  using scalar_t = S;
  Sphere() = default;
  Sphere(const Point<S>& c, S r) : center_(c), radius_(r) {}
  const Point<S>& center() const { return center_; }
  Point<S>& center() { return center_; }
  const S& radius() const { return radius_; }
  S& radius() { return radius_; }

 private:
  Point<S> center_;
  S radius_{S(-1)};
};
template <class S = double>
class Spherocylinder {  // primitives/Spherocylinder.hpp:40-321; defaults q = identity, r = L = -1
 public:
  using scalar_t = S;
  Spherocylinder() = default;
  Spherocylinder(const Point<S>& c, const Quaternion<S>& q, S r, S l) : center_(c), orient_(q), radius_(r), length_(l) {}
  const Point<S>& center() const { return center_; }
  const Quaternion<S>& orientation() const { return orient_; }
  const S& radius() const { return radius_; }
  const S& length() const { return length_; }
  Point<S>& center() { return center_; }
  Quaternion<S>& orientation() { return orient_; }

 private:
  Point<S> center_;
  Quaternion<S> orient_;
  S radius_{S(-1)}, length_{S(-1)};
};
template <class S = double>
class Ellipsoid {  // primitives/Ellipsoid.hpp:42-371
 public:
  using scalar_t = S;
  Ellipsoid() = default;
  Ellipsoid(const Point<S>& c, const Quaternion<S>& q, const Point<S>& radii) : center_(c), orient_(q), radii_(radii) {}
  const Point<S>& center() const { return center_; }
  const Quaternion<S>& orientation() const { return orient_; }
  const Point<S>& radii() const { return radii_; }

 private:
  Point<S> center_;
  Quaternion<S> orient_;
  Point<S> radii_;
};
template <class S = double>
class LineSegment {
 public:
  LineSegment() = default;
  LineSegment(const Point<S>& a, const Point<S>& b) : a_(a), b_(b) {}
  const Point<S>& start() const { return a_; }
  const Point<S>& end() const { return b_; }

 private:
  Point<S> a_, b_;
};
template <class S = double>
class AABB {  // primitives/AABB.hpp:41; default is the inverted box (+max, -max)
 public:
  AABB() = default;
  AABB(S x0, S y0, S z0, S x1, S y1, S z1) : lo_(x0, y0, z0), hi_(x1, y1, z1) {}
  const Point<S>& min_corner() const { return lo_; }
  const Point<S>& max_corner() const { return hi_; }

 private:
  static constexpr S kMax = std::numeric_limits<S>::max();
  Point<S> lo_{kMax, kMax, kMax}, hi_{-kMax, -kMax, -kMax};
};
// geom::intersects (AABB.hpp:420-431): closed test
template <class S>
bool intersects(const AABB<S>& a, const AABB<S>& b) {
  for (int k = 0; k < 3; ++k)
    if (a.max_corner()[k] < b.min_corner()[k]) return false;
  for (int k = 0; k < 3; ++k)
    if (b.max_corner()[k] < a.min_corner()[k]) return false;
  return true;
}

namespace detail {
inline std::vector<AABB<double>> unpack_aabb(const std::vector<double>& h) {
  std::vector<AABB<double>> out(h.size() / 6);
  for (size_t i = 0; i < out.size(); ++i)
    out[i] = AABB<double>(h[6 * i], h[6 * i + 1], h[6 * i + 2], h[6 * i + 3], h[6 * i + 4], h[6 * i + 5]);
  return out;
}
template <class V>
void push3(std::vector<double>& dst, const V& p) {
  dst.push_back(p[0]); dst.push_back(p[1]); dst.push_back(p[2]);
}
}  // namespace detail

// compute_aabb batch overloads (compute_aabb.hpp:72-127)
inline std::vector<AABB<double>> compute_aabb(const std::vector<Sphere<double>>& s) {
  std::vector<double> c, r;
  for (const auto& x : s) { detail::push3(c, x.center()); r.push_back(x.radius()); }
  DeviceVector dc(c), dr(r), out(6 * s.size());
  check(mhip_compute_aabb_spheres(s.size(), dc.data(), dr.data(), out.data(), nullptr));
  return detail::unpack_aabb(out.download());
}
inline std::vector<AABB<double>> compute_aabb(const std::vector<Spherocylinder<double>>& s) {
  std::vector<double> c, q, r, l;
  for (const auto& x : s) {
    detail::push3(c, x.center());
    for (int k = 0; k < 4; ++k) q.push_back(x.orientation()[k]);
    r.push_back(x.radius()); l.push_back(x.length());
  }
  DeviceVector dc(c), dq(q), dr(r), dl(l), out(6 * s.size());
  check(mhip_compute_aabb_spherocylinders(s.size(), dc.data(), dq.data(), dr.data(), dl.data(), out.data(), nullptr));
  return detail::unpack_aabb(out.download());
}
inline std::vector<AABB<double>> compute_aabb(const std::vector<Ellipsoid<double>>& s) {
  std::vector<double> c, q, r;
  for (const auto& x : s) {
    detail::push3(c, x.center());
    for (int k = 0; k < 4; ++k) q.push_back(x.orientation()[k]);
    detail::push3(r, x.radii());
  }
  DeviceVector dc(c), dq(q), dr(r), out(6 * s.size());
  check(mhip_compute_aabb_ellipsoids(s.size(), dc.data(), dq.data(), dr.data(), out.data(), nullptr));
  return detail::unpack_aabb(out.download());
}

// distance(SharedNormalSigned, Sphere, Sphere, sep) (SphereSphere.hpp:66-76), element-wise over two equal-length lists
inline std::vector<double> distance(SharedNormalSigned, const std::vector<Sphere<double>>& a,
                                    const std::vector<Sphere<double>>& b, std::vector<Point<double>>* sep = nullptr) {
  if (a.size() != b.size()) throw std::invalid_argument("distance: list sizes differ");
  std::vector<double> c1, r1, c2, r2;
  for (size_t i = 0; i < a.size(); ++i) {
    detail::push3(c1, a[i].center()); r1.push_back(a[i].radius());
    detail::push3(c2, b[i].center()); r2.push_back(b[i].radius());
  }
  DeviceVector d1(c1), e1(r1), d2(c2), e2(r2), dist(a.size()), dsep(3 * a.size());
  check(mhip_distance_sphere_sphere(a.size(), d1.data(), e1.data(), d2.data(), e2.data(), dist.data(), dsep.data(),
                                    nullptr));
  if (sep) {
    const auto h = dsep.download();
    sep->resize(a.size());
    for (size_t i = 0; i < a.size(); ++i) (*sep)[i] = Point<double>(h[3 * i], h[3 * i + 1], h[3 * i + 2]);
  }
  return dist.download();
}

// distance(SharedNormalSigned, Point, Sphere[, sep]) (PointSphere.hpp:57-79), element-wise over two equal-length lists
inline std::vector<double> distance(SharedNormalSigned, const std::vector<Point<double>>& p,
                                    const std::vector<Sphere<double>>& s, std::vector<Point<double>>* sep = nullptr) {
  if (p.size() != s.size()) throw std::invalid_argument("distance: list sizes differ");
  const size_t n = p.size();
  std::vector<double> pp, c, r;
  for (size_t i = 0; i < n; ++i) {
    detail::push3(pp, p[i]); detail::push3(c, s[i].center()); r.push_back(s[i].radius());
  }
  DeviceVector dp(pp), dc(c), dr(r), dist(n), dsep(3 * n);
  check(mhip_distance_point_sphere(n, dp.data(), dc.data(), dr.data(), dist.data(), dsep.data(), nullptr));
  if (sep) {
    const auto h = dsep.download();
    sep->resize(n);
    for (size_t i = 0; i < n; ++i) (*sep)[i] = Point<double>(h[3 * i], h[3 * i + 1], h[3 * i + 2]);
  }
  return dist.download();
}

struct SegmentSphereResult {
  std::vector<double> distance, arch_length;
  std::vector<Point<double>> closest_point, sep;
};
// distance(SharedNormalSigned, LineSegment, Sphere, closest_point, arch_length, sep) (LineSegmentSphere.hpp:88-100)
inline SegmentSphereResult distance(SharedNormalSigned, const std::vector<LineSegment<double>>& a,
                                    const std::vector<Sphere<double>>& s) {
  if (a.size() != s.size()) throw std::invalid_argument("distance: list sizes differ");
  const size_t n = a.size();
  std::vector<double> a0, a1, c, r;
  for (size_t i = 0; i < n; ++i) {
    detail::push3(a0, a[i].start()); detail::push3(a1, a[i].end());
    detail::push3(c, s[i].center()); r.push_back(s[i].radius());
  }
  DeviceVector da0(a0), da1(a1), dc(c), dr(r), dist(n), cp(3 * n), t(n), dsep(3 * n);
  check(mhip_distance_segment_sphere(n, da0.data(), da1.data(), dc.data(), dr.data(), dist.data(), cp.data(), t.data(),
                                     dsep.data(), nullptr));
  SegmentSphereResult out;
  out.distance = dist.download(); out.arch_length = t.download();
  auto to_pts = [n](const std::vector<double>& h) {
    std::vector<Point<double>> q(n);
    for (size_t i = 0; i < n; ++i) q[i] = Point<double>(h[3 * i], h[3 * i + 1], h[3 * i + 2]);
    return q;
  };
  out.closest_point = to_pts(cp.download()); out.sep = to_pts(dsep.download());
  return out;
}

struct EllipsoidEllipsoidResult {
  std::vector<double> distance;
  std::vector<Point<double>> closest_point1, closest_point2, shared_normal1, shared_normal2;
};
// distance(SharedNormalSigned, Ellipsoid, Ellipsoid, cp1, cp2, n1, n2) (EllipsoidEllipsoid.hpp:62-151), element-wise
inline EllipsoidEllipsoidResult distance(SharedNormalSigned, const std::vector<Ellipsoid<double>>& a,
                                         const std::vector<Ellipsoid<double>>& b) {
  if (a.size() != b.size()) throw std::invalid_argument("distance: list sizes differ");
  const size_t n = a.size();
  std::vector<double> c1, q1, r1, c2, q2, r2;
  for (size_t i = 0; i < n; ++i) {
    detail::push3(c1, a[i].center()); detail::push3(r1, a[i].radii());
    detail::push3(c2, b[i].center()); detail::push3(r2, b[i].radii());
    for (int k = 0; k < 4; ++k) { q1.push_back(a[i].orientation()[k]); q2.push_back(b[i].orientation()[k]); }
  }
  DeviceVector dc1(c1), dq1(q1), dr1(r1), dc2(c2), dq2(q2), dr2(r2), dist(n), cp1(3 * n), cp2(3 * n), n1(3 * n), n2(3 * n);
  check(mhip_distance_ellipsoid_ellipsoid(n, dc1.data(), dq1.data(), dr1.data(), dc2.data(), dq2.data(), dr2.data(),
                                          dist.data(), cp1.data(), cp2.data(), n1.data(), n2.data(), nullptr));
  auto to_pts = [n](const std::vector<double>& h) {
    std::vector<Point<double>> p(n);
    for (size_t i = 0; i < n; ++i) p[i] = Point<double>(h[3 * i], h[3 * i + 1], h[3 * i + 2]);
    return p;
  };
  EllipsoidEllipsoidResult r;
  r.distance = dist.download();
  r.closest_point1 = to_pts(cp1.download()); r.closest_point2 = to_pts(cp2.download());
  r.shared_normal1 = to_pts(n1.download()); r.shared_normal2 = to_pts(n2.download());
  return r;
}
// distance(SharedNormalSigned, Point, Ellipsoid, closest, normal) (PointEllipsoid.hpp:94-135), element-wise
inline std::vector<double> distance(SharedNormalSigned, const std::vector<Point<double>>& p,
                                    const std::vector<Ellipsoid<double>>& e,
                                    std::vector<Point<double>>* closest = nullptr,
                                    std::vector<Point<double>>* normal = nullptr) {
  if (p.size() != e.size()) throw std::invalid_argument("distance: list sizes differ");
  const size_t n = p.size();
  std::vector<double> pp, c, q, r;
  for (size_t i = 0; i < n; ++i) {
    detail::push3(pp, p[i]); detail::push3(c, e[i].center()); detail::push3(r, e[i].radii());
    for (int k = 0; k < 4; ++k) q.push_back(e[i].orientation()[k]);
  }
  DeviceVector dp(pp), dc(c), dq(q), dr(r), dist(n), cp(3 * n), nrm(3 * n);
  check(mhip_distance_point_ellipsoid(n, dp.data(), dc.data(), dq.data(), dr.data(), dist.data(), cp.data(),
                                      nrm.data(), nullptr));
  auto fill = [n](std::vector<Point<double>>* dst, const std::vector<double>& h) {
    if (!dst) return;
    dst->resize(n);
    for (size_t i = 0; i < n; ++i) (*dst)[i] = Point<double>(h[3 * i], h[3 * i + 1], h[3 * i + 2]);
  };
  fill(closest, cp.download());
  fill(normal, nrm.download());
  return dist.download();
}

struct SegmentSegmentResult {
  std::vector<double> distance, arch_length1, arch_length2;
  std::vector<Point<double>> closest_point1, closest_point2, sep;
};
// distance(SharedNormalSigned, LineSegment, LineSegment, cp1, cp2, s, t, sep) (LineSegmentLineSegment.hpp:189-318)
inline SegmentSegmentResult distance(SharedNormalSigned, const std::vector<LineSegment<double>>& a,
                                     const std::vector<LineSegment<double>>& b) {
  if (a.size() != b.size()) throw std::invalid_argument("distance: list sizes differ");
  const size_t n = a.size();
  std::vector<double> a0, a1, b0, b1;
  for (size_t i = 0; i < n; ++i) {
    detail::push3(a0, a[i].start()); detail::push3(a1, a[i].end());
    detail::push3(b0, b[i].start()); detail::push3(b1, b[i].end());
  }
  DeviceVector da0(a0), da1(a1), db0(b0), db1(b1), dist(n), cp1(3 * n), cp2(3 * n), s(n), t(n), sep(3 * n);
  check(mhip_distance_segment_segment(n, da0.data(), da1.data(), db0.data(), db1.data(), dist.data(), cp1.data(),
                                      cp2.data(), s.data(), t.data(), sep.data(), nullptr));
  SegmentSegmentResult r;
  r.distance = dist.download(); r.arch_length1 = s.download(); r.arch_length2 = t.download();
  auto to_pts = [n](const std::vector<double>& h) {
    std::vector<Point<double>> p(n);
    for (size_t i = 0; i < n; ++i) p[i] = Point<double>(h[3 * i], h[3 * i + 1], h[3 * i + 2]);
    return p;
  };
  r.closest_point1 = to_pts(cp1.download()); r.closest_point2 = to_pts(cp2.download()); r.sep = to_pts(sep.download());
  return r;
}

// ---- single-object overloads (seam S4) ---------------------------------------------------------------------------------
// The reference's free functions take ONE pair of owning primitives (they are called per thread inside user kernels:
// distance/SphereSphere.hpp:44-76, LineSegmentLineSegment.hpp:169-197, EllipsoidEllipsoid.hpp:106-113,
// PointEllipsoid.hpp:94-135, compute_aabb.hpp:72-127).  Host code that holds single objects gets the same signatures
// here; each forwards to a one-element batch of the device routine (same arithmetic, one launch and one round trip
// per call -- for a handful of objects; anything in a loop belongs in the batch overloads above).
inline AABB<double> compute_aabb(const Sphere<double>& sphere) {
  return compute_aabb(std::vector<Sphere<double>>{sphere})[0];
}
inline AABB<double> compute_aabb(const Spherocylinder<double>& spherocylinder) {
  return compute_aabb(std::vector<Spherocylinder<double>>{spherocylinder})[0];
}
inline AABB<double> compute_aabb(const Ellipsoid<double>& ellipsoid) {
  return compute_aabb(std::vector<Ellipsoid<double>>{ellipsoid})[0];
}
inline double distance(SharedNormalSigned t, const Sphere<double>& sphere1, const Sphere<double>& sphere2) {
  return distance(t, std::vector<Sphere<double>>{sphere1}, std::vector<Sphere<double>>{sphere2})[0];
}
inline double distance(const Sphere<double>& sphere1, const Sphere<double>& sphere2) {  // SphereSphere.hpp:44-48
  return distance(SharedNormalSigned{}, sphere1, sphere2);
}
inline double distance(SharedNormalSigned t, const Sphere<double>& sphere1, const Sphere<double>& sphere2,
                       Point<double>& sep) {
  std::vector<Point<double>> s;
  const double d = distance(t, std::vector<Sphere<double>>{sphere1}, std::vector<Sphere<double>>{sphere2}, &s)[0];
  sep = s[0];
  return d;
}
inline double distance(const Sphere<double>& sphere1, const Sphere<double>& sphere2, Point<double>& sep) {
  return distance(SharedNormalSigned{}, sphere1, sphere2, sep);  // SphereSphere.hpp:66-76
}
inline double distance(SharedNormalSigned t, const Point<double>& point, const Sphere<double>& sphere) {
  return distance(t, std::vector<Point<double>>{point}, std::vector<Sphere<double>>{sphere})[0];
}
inline double distance(const Point<double>& point, const Sphere<double>& sphere) {  // PointSphere.hpp:46-50
  return distance(SharedNormalSigned{}, point, sphere);
}
inline double distance(const Point<double>& point, const Sphere<double>& sphere, Point<double>& sep) {
  std::vector<Point<double>> s;   // PointSphere.hpp:69-79
  const double d = distance(SharedNormalSigned{}, std::vector<Point<double>>{point}, std::vector<Sphere<double>>{sphere}, &s)[0];
  sep = s[0];
  return d;
}
inline double distance(SharedNormalSigned t, const LineSegment<double>& line_segment, const Sphere<double>& sphere,
                       Point<double>& closest_point, double& arch_length, Point<double>& sep) {
  const SegmentSphereResult r = distance(t, std::vector<LineSegment<double>>{line_segment},
                                         std::vector<Sphere<double>>{sphere});   // LineSegmentSphere.hpp:88-100
  closest_point = r.closest_point[0];
  arch_length = r.arch_length[0];
  sep = r.sep[0];
  return r.distance[0];
}
inline double distance(const LineSegment<double>& line_segment, const Sphere<double>& sphere,
                       Point<double>& closest_point, double& arch_length, Point<double>& sep) {
  return distance(SharedNormalSigned{}, line_segment, sphere, closest_point, arch_length, sep);  // :71-78
}
inline double distance(SharedNormalSigned t, const LineSegment<double>& line_segment, const Sphere<double>& sphere) {
  Point<double> cp, sep;   // LineSegmentSphere.hpp:58-62
  double al;
  return distance(t, line_segment, sphere, cp, al, sep);
}
inline double distance(const LineSegment<double>& line_segment, const Sphere<double>& sphere) {
  return distance(SharedNormalSigned{}, line_segment, sphere);   // LineSegmentSphere.hpp:47-51
}
inline double distance(SharedNormalSigned t, const LineSegment<double>& line_segment1,
                       const LineSegment<double>& line_segment2, Point<double>& closest_point1,
                       Point<double>& closest_point2, double& arch_length1, double& arch_length2, Point<double>& sep) {
  const SegmentSegmentResult r = distance(t, std::vector<LineSegment<double>>{line_segment1},
                                          std::vector<LineSegment<double>>{line_segment2});
  closest_point1 = r.closest_point1[0];
  closest_point2 = r.closest_point2[0];
  arch_length1 = r.arch_length1[0];   // the reference's raw parameters: unclamped in the colinear branch
  arch_length2 = r.arch_length2[0];
  sep = r.sep[0];
  return r.distance[0];
}
inline double distance(const LineSegment<double>& line_segment1, const LineSegment<double>& line_segment2,
                       Point<double>& closest_point1, Point<double>& closest_point2, double& arch_length1,
                       double& arch_length2, Point<double>& sep) {  // LineSegmentLineSegment.hpp:169-178
  return distance(SharedNormalSigned{}, line_segment1, line_segment2, closest_point1, closest_point2, arch_length1,
                  arch_length2, sep);
}
inline double distance(SharedNormalSigned t, const Ellipsoid<double>& ellipsoid1, const Ellipsoid<double>& ellipsoid2,
                       Point<double>& closest_point1, Point<double>& closest_point2, Point<double>& shared_normal1,
                       Point<double>& shared_normal2) {  // EllipsoidEllipsoid.hpp:106-113
  const EllipsoidEllipsoidResult r = distance(t, std::vector<Ellipsoid<double>>{ellipsoid1},
                                              std::vector<Ellipsoid<double>>{ellipsoid2});
  closest_point1 = r.closest_point1[0];
  closest_point2 = r.closest_point2[0];
  shared_normal1 = r.shared_normal1[0];
  shared_normal2 = r.shared_normal2[0];
  return r.distance[0];
}
inline double distance(SharedNormalSigned t, const Ellipsoid<double>& ellipsoid1, const Ellipsoid<double>& ellipsoid2) {
  Point<double> a, b, c, d;
  return distance(t, ellipsoid1, ellipsoid2, a, b, c, d);
}
inline double distance(SharedNormalSigned t, const Point<double>& point, const Ellipsoid<double>& ellipsoid,
                       Point<double>& closest_point, Point<double>& ellipsoid_normal) {  // PointEllipsoid.hpp:94-135
  std::vector<Point<double>> cp, nrm;
  const double d = distance(t, std::vector<Point<double>>{point}, std::vector<Ellipsoid<double>>{ellipsoid}, &cp, &nrm)[0];
  closest_point = cp[0];
  ellipsoid_normal = nrm[0];
  return d;
}

// ---- periodic metrics (mundy_geom/periodicity.hpp) -------------------------------------------------------------------
namespace detail {
inline std::vector<double> flatten(const std::vector<Point<double>>& p) {
  std::vector<double> h;
  h.reserve(3 * p.size());
  for (const auto& x : p) push3(h, x);
  return h;
}
inline std::vector<Point<double>> to_points(const std::vector<double>& h) {
  std::vector<Point<double>> p(h.size() / 3);
  for (size_t i = 0; i < p.size(); ++i) p[i] = Point<double>(h[3 * i], h[3 * i + 1], h[3 * i + 2]);
  return p;
}
}  // namespace detail

/// PeriodicScaledMetric (periodicity.hpp:743-841): orthorhombic cell of edge lengths `cell`; batch sep / wrap.
class PeriodicScaledMetric {
 public:
  explicit PeriodicScaledMetric(const Point<double>& cell) : cell_{cell[0], cell[1], cell[2]} {}
  std::vector<Point<double>> sep(const std::vector<Point<double>>& p1, const std::vector<Point<double>>& p2) const {
    if (p1.size() != p2.size()) throw std::invalid_argument("sep: list sizes differ");
    DeviceVector a(detail::flatten(p1)), b(detail::flatten(p2)), out(3 * p1.size());
    check(mhip_periodic_sep(p1.size(), cell_, a.data(), b.data(), out.data(), nullptr));
    return detail::to_points(out.download());
  }
  std::vector<Point<double>> wrap(const std::vector<Point<double>>& p) const {
    DeviceVector a(detail::flatten(p));
    check(mhip_wrap_rigid(p.size(), cell_, a.data(), nullptr));
    return detail::to_points(a.download());
  }
  const double* cell() const { return cell_; }

 private:
  double cell_[3];
};
/// PeriodicMetric (periodicity.hpp:233-332): general unit cell h (row-major 3x3, lattice vectors as columns).
class PeriodicMetric {
 public:
  explicit PeriodicMetric(const double (&h)[9]) {
    for (int i = 0; i < 9; ++i) h_[i] = h[i];
    check(mhip_unit_cell_inverse(h_, h_inv_));
  }
  std::vector<Point<double>> sep(const std::vector<Point<double>>& p1, const std::vector<Point<double>>& p2) const {
    if (p1.size() != p2.size()) throw std::invalid_argument("sep: list sizes differ");
    DeviceVector a(detail::flatten(p1)), b(detail::flatten(p2)), out(3 * p1.size());
    check(mhip_periodic_sep_triclinic(p1.size(), h_, a.data(), b.data(), out.data(), nullptr));
    return detail::to_points(out.download());
  }
  std::vector<Point<double>> wrap(const std::vector<Point<double>>& p) const {
    DeviceVector a(detail::flatten(p));
    check(mhip_wrap_rigid_triclinic(p.size(), h_, a.data(), nullptr));
    return detail::to_points(a.download());
  }
  const double* direct_lattice_vectors() const { return h_; }
  const double* inverse() const { return h_inv_; }

 private:
  double h_[9], h_inv_[9];
};
/// periodic_metric_from_unit_cell / periodic_scaled_metric_from_unit_cell (periodicity.hpp:848-855, :871-875)
inline PeriodicMetric periodic_metric_from_unit_cell(const Point<double>& cell) {
  const double h[9] = {cell[0], 0, 0, 0, cell[1], 0, 0, 0, cell[2]};
  return PeriodicMetric(h);
}
inline PeriodicScaledMetric periodic_scaled_metric_from_unit_cell(const Point<double>& cell) {
  return PeriodicScaledMetric(cell);
}

}  // namespace geom

// ---- neighbour links ------------------------------------------------------------------------------------------------------
namespace mesh {

/// search_filters (GenNeighborLinkers.hpp:139-283).  The reference composes arbitrary predicates over the search results
/// (SearchFilter<ExecSpace, Predicates...>); functors cannot cross the C ABI, so the two predicates it ships are values
/// here and make_search_filter combines them.
namespace search_filters {
struct ExcludeSelfInteractions {};  // :185-200
/// :202-236 -- a source is not linked to the targets it is connected to.  The connectivity is a CSR over the bodies
/// (device arrays; it plays the part of ngp_mesh.get_connected_entities).
struct ExcludeConnectedEntities {
  size_t num_bodies;
  const int32_t* conn_ptr;  // [num_bodies + 1]
  const int32_t* conn_idx;
  size_t num_entries;
};
struct SearchFilter {
  bool exclude_self = false;
  bool has_connected = false;
  ExcludeConnectedEntities connected{0, nullptr, nullptr, 0};
};
inline void add_predicate(SearchFilter& f, const ExcludeSelfInteractions&) { f.exclude_self = true; }
inline void add_predicate(SearchFilter& f, const ExcludeConnectedEntities& c) {
  f.has_connected = true;
  f.connected = c;
}
template <class... Predicates>
std::shared_ptr<SearchFilter> make_search_filter(const Predicates&... predicates) {  // :269-272
  auto f = std::make_shared<SearchFilter>();
  (add_predicate(*f, predicates), ...);
  return f;
}
}  // namespace search_filters

/// one row of the result in stk::search's vocabulary: IdentProcIntersection (GenNeighborLinkers.hpp:118-121)
struct IdentProcPairs {
  DeviceArray<uint64_t> source_id, target_id;
  DeviceArray<int32_t> source_proc, target_proc;
};
/// the list in LinkCOOData's per-link fields (LinkMetaData.hpp:102-106)
struct LinkCOO {
  DeviceArray<uint64_t> link_id, linked_entity_ids;  // [P], [P][2]
  DeviceArray<unsigned char> linked_entity_ranks;    // [P][2]
};
/// entity -> connected links as LinkCRSBucketConn keeps it per bucket (LinkCRSBucketConn.hpp:183-191)
struct LinkCRS {
  unsigned bucket_capacity = 0;
  size_t num_buckets = 0;
  DeviceArray<unsigned> num_connected_links, sparse_connectivity_offsets;  // [n], [num_buckets][capacity + 1]
  DeviceArray<uint64_t> sparse_connectivity, bucket_begin;                 // [2 P], [num_buckets + 1]
};

class GenNeighborLinks {  // mundy_mesh/GenNeighborLinkers.hpp:294-866 (builder + generate), device arrays instead of STK
 public:
  GenNeighborLinks() { check(mhip_broadphase_create(&h_)); }
  ~GenNeighborLinks() { mhip_broadphase_destroy(h_); }
  GenNeighborLinks(const GenNeighborLinks&) = delete;
  GenNeighborLinks& operator=(const GenNeighborLinks&) = delete;

  GenNeighborLinks& set_search_buffer(double b) { guard("search buffer"); cfg_.buffer = b; return *this; }
  GenNeighborLinks& set_enforce_source_target_symmetry(bool v) {
    guard("enforce source-target symmetry"); cfg_.symmetric = v ? 1 : 0; return *this;
  }
  GenNeighborLinks& set_search_kind(int kind) { guard("search kind"); cfg_.search_kind = kind; return *this; }
  GenNeighborLinks& set_periodic_box(double lx, double ly, double lz) {
    guard("periodic box"); cfg_.periodic = 1; cfg_.box[0] = lx; cfg_.box[1] = ly; cfg_.box[2] = lz; return *this;
  }
  /// triclinic unit cell (PeriodicMetric, periodicity.hpp:233-332): row-major 3 x 3, lattice vectors as columns
  GenNeighborLinks& set_periodic_cell(const double cell[9]) {
    guard("periodic cell"); cfg_.periodic = 2;
    for (int k = 0; k < 9; ++k) cfg_.cell[k] = cell[k];
    return *this;
  }
  /// stk::search::SearchMethod (:443-447; the reference's default is MORTON_LBVH): MHIP_SEARCH_METHOD_*
  GenNeighborLinks& set_search_method(int method) { guard("search method"); cfg_.method = method; return *this; }
  int get_search_method() const { return cfg_.method; }
  /// :449-455.  Without a filter every body also meets itself, as stk's coarse_search reports it.
  GenNeighborLinks& set_search_filter(std::shared_ptr<search_filters::SearchFilter> filter) {
    guard("search filter");
    filter_ = std::move(filter);
    cfg_.include_self = (filter_ && filter_->exclude_self) ? 0 : 1;
    if (filter_ && filter_->has_connected) {
      const auto& c = filter_->connected;
      check(mhip_broadphase_set_exclusions(h_, c.num_bodies, c.conn_ptr, c.conn_idx, c.num_entries, nullptr));
    }
    return *this;
  }
  /// acts_on(source_selector, target_selector, ...) (:486-507): byte masks over the bodies (device arrays, nullptr = all)
  GenNeighborLinks& acts_on(size_t n, const unsigned char* source_mask, const unsigned char* target_mask) {
    guard("source/targets");
    check(mhip_broadphase_set_sets(h_, n, source_mask, target_mask, nullptr));
    return *this;
  }
  /// (stk::mesh::EntityId, owner rank) of every body (:575-584); nullptr id = the local index, nullptr owner = 0
  GenNeighborLinks& set_identities(size_t n, const uint64_t* entity_id, const int32_t* owner_rank) {
    check(mhip_broadphase_set_identities(h_, n, entity_id, owner_rank, nullptr));
    return *this;
  }
  /// the already-linked neighbours (get_linked_neighbors_set, :91-113), dropped from the results when duplicate links
  /// are not allowed (set_allow_duplicate_links(false), :422-427): same CSR form as ExcludeConnectedEntities, and it
  /// replaces that filter's list -- merge the two lists when both are wanted
  GenNeighborLinks& set_existing_linked_neighbors(size_t n, const int32_t* ptr, const int32_t* idx, size_t entries) {
    check(mhip_broadphase_set_exclusions(h_, n, ptr, idx, entries, nullptr));
    generated_ = false;
    return *this;
  }
  void concretize() {
    if (concretized_) throw std::runtime_error("Cannot concretize more than once.");
    concretized_ = true;
  }
  bool is_concretized() const { return concretized_; }
  /// \return true if the search was performed, false if no regeneration was necessary (:510-543)
  bool generate(size_t n, const double* aabb, const double* center, const double* bounding_radius,
                mhip_stream_t stream = nullptr, bool force = false) {
    if (!concretized_) throw std::runtime_error("Cannot generate links before concretization.");
    if (generated_ && !force) {
      int flag = 0;
      check(mhip_broadphase_needs_rebuild(h_, n, center, &flag, stream));
      if (!flag) return false;
    }
    check(mhip_broadphase_build(h_, &cfg_, n, aabb, center, bounding_radius, &num_pairs_, stream));
    n_ = n;
    generated_ = true;
    return true;
  }
  /// the rebuild test on its own (:603-615): has any body moved more than half the search buffer since the last build?
  bool objects_moved_too_much(size_t n, const double* center, mhip_stream_t stream = nullptr) const {
    int flag = 1;
    check(mhip_broadphase_needs_rebuild(h_, n, center, &flag, stream));
    return flag != 0;
  }
  /// the body numbering changed (reordering, migration): the next generate() builds a new list whatever the rebuild
  /// rule says (the reference re-collects its entity indices when they change, GenNeighborLinkers.hpp:745-800)
  void invalidate() { generated_ = false; }
  size_t num_links() const { return num_pairs_; }
  /// (source, target) pairs, sorted by (source, target)
  DeviceArray<int32_t> links(mhip_stream_t stream = nullptr) const {
    DeviceArray<int32_t> p(2 * num_pairs_);
    check(mhip_broadphase_get_pairs(h_, p.data(), nullptr, nullptr, stream));
    return p;
  }
  /// the links as IdentProcIntersection rows
  IdentProcPairs ident_links(mhip_stream_t stream = nullptr) const {
    IdentProcPairs r{DeviceArray<uint64_t>(num_pairs_), DeviceArray<uint64_t>(num_pairs_),
                     DeviceArray<int32_t>(num_pairs_), DeviceArray<int32_t>(num_pairs_)};
    check(mhip_broadphase_get_ident_pairs(h_, r.source_id.data(), r.source_proc.data(), r.target_id.data(),
                                          r.target_proc.data(), stream));
    return r;
  }
  /// the links in MuNDy's link layout, ready for LinkData without the host-serial request_link loop (:714-738)
  LinkCOO export_coo(uint64_t first_link_id, int source_rank, int target_rank, mhip_stream_t stream = nullptr) const {
    LinkCOO r{DeviceArray<uint64_t>(num_pairs_), DeviceArray<uint64_t>(2 * num_pairs_),
              DeviceArray<unsigned char>(2 * num_pairs_)};
    check(mhip_links_export_coo(h_, first_link_id, source_rank, target_rank, r.link_id.data(),
                                r.linked_entity_ids.data(), r.linked_entity_ranks.data(), stream));
    return r;
  }
  LinkCRS export_crs(uint64_t first_link_id, unsigned bucket_capacity = 512, mhip_stream_t stream = nullptr) const {
    LinkCRS r;
    r.bucket_capacity = bucket_capacity;
    r.num_buckets = (n_ + bucket_capacity - 1) / bucket_capacity;
    r.num_connected_links = DeviceArray<unsigned>(n_);
    r.sparse_connectivity_offsets = DeviceArray<unsigned>(r.num_buckets * (bucket_capacity + 1));
    r.sparse_connectivity = DeviceArray<uint64_t>(2 * num_pairs_);
    r.bucket_begin = DeviceArray<uint64_t>(r.num_buckets + 1);
    check(mhip_links_export_crs(h_, first_link_id, bucket_capacity, r.num_connected_links.data(),
                                r.sparse_connectivity_offsets.data(), r.sparse_connectivity.data(),
                                r.bucket_begin.data(), stream));
    return r;
  }
  /// the same into a caller-owned buffer that only grows (a time loop then stops allocating); returns its data()
  int32_t* links_into(DeviceArray<int32_t>& out, mhip_stream_t stream = nullptr) const {
    if (out.size() < 2 * num_pairs_) out = DeviceArray<int32_t>(2 * num_pairs_ + num_pairs_ / 4 + 16);
    check(mhip_broadphase_get_pairs(h_, out.data(), nullptr, nullptr, stream));
    return out.data();
  }

 private:
  void guard(const char* what) const {
    if (concretized_) throw std::runtime_error(std::string("Cannot set ") + what + " after concretization.");
  }
  mhip_broadphase_t h_ = nullptr;
  // no filter set: the historical behaviour of this adapter (self pairs excluded) is kept until set_search_filter is
  // called, after which the filter decides as in the reference
  mhip_broadphase_config cfg_{MHIP_SEARCH_SPHERES, 0, 0.0, 0, {0, 0, 0}, MHIP_SEARCH_METHOD_AUTO, 0, {0, 0, 0, 0, 0, 0, 0, 0, 0}};
  std::shared_ptr<search_filters::SearchFilter> filter_;
  bool concretized_ = false, generated_ = false;
  size_t num_pairs_ = 0, n_ = 0;
};

}  // namespace mesh

// ---- contact operator (a LinearOp with apply(x, y)) ---------------------------------------------------------------------
class ContactOperator {
 public:
  ContactOperator(size_t num_constraints, size_t num_bodies, const int32_t* pairs, const double* normal,
                  const double* ra, const double* rb, const double* mob_trans, const double* mob_rot, double dt,
                  mhip_stream_t stream = nullptr, const double* priority = nullptr) {
    check(mhip_contact_op_create(&h_, num_constraints, num_bodies, pairs, normal, ra, rb, mob_trans, mob_rot, dt,
                                 priority, stream));
  }
  /// Spherocylinders: lever arms as arclengths (s, t) along the rods' segments (mhip_contact_op_create_rods).
  struct Rods {
    const double* arc_s;
    const double* arc_t;
    const double* segments;  // [num_bodies][8] from mhip_spherocylinder_segments
  };
  ContactOperator(size_t num_constraints, size_t num_bodies, const int32_t* pairs, const double* normal, Rods rods,
                  const double* mob_trans, const double* mob_rot, double dt, mhip_stream_t stream = nullptr,
                  const double* priority = nullptr) {
    check(mhip_contact_op_create_rods(&h_, num_constraints, num_bodies, pairs, normal, rods.arc_s, rods.arc_t,
                                      rods.segments, mob_trans, mob_rot, dt, priority, stream));
  }
  ~ContactOperator() { mhip_contact_op_destroy(h_); }
  ContactOperator(const ContactOperator&) = delete;
  ContactOperator& operator=(const ContactOperator&) = delete;
  /// same pairs, new geometry (a step that reuses the neighbour list): the incidence index stays
  void refresh(const double* normal, const double* ra, const double* rb, mhip_stream_t stream = nullptr) {
    check(mhip_contact_op_refresh(h_, normal, ra, rb, stream));
  }
  void refresh(const double* normal, Rods rods, mhip_stream_t stream = nullptr) {
    check(mhip_contact_op_refresh_rods(h_, normal, rods.arc_s, rods.arc_t, rods.segments, stream));
  }
  void apply(const DeviceVector& x, DeviceVector& y) const { check(mhip_contact_op_apply(h_, x.data(), y.data(), nullptr)); }
  mhip_contact_op_t handle() const { return h_; }

 private:
  mhip_contact_op_t h_ = nullptr;
};

// ---- convex -------------------------------------------------------------------------------------------------------------
namespace convex {

namespace space {  // convex.hpp:46-115
template <class S = double>
struct Unconstrained {
  using scalar_t = S;
  S project(const S& x) const { return x; }
  S operator()(const S& x) const { return project(x); }
  mhip_space c_space() const { return {MHIP_SPACE_UNCONSTRAINED, 0.0, 0.0}; }
};
template <class S = double>
struct LowerBound {
  using scalar_t = S;
  S lower_bound;
  S project(const S& x) const { return x < lower_bound ? lower_bound : x; }
  S operator()(const S& x) const { return project(x); }
  mhip_space c_space() const { return {MHIP_SPACE_LOWER_BOUND, lower_bound, 0.0}; }
};
template <class S = double>
struct UpperBound {
  using scalar_t = S;
  S upper_bound;
  S project(const S& x) const { return upper_bound < x ? upper_bound : x; }
  S operator()(const S& x) const { return project(x); }
  mhip_space c_space() const { return {MHIP_SPACE_UPPER_BOUND, 0.0, upper_bound}; }
};
template <class S = double>
struct Bounded {
  using scalar_t = S;
  S lower_bound, upper_bound;
  Bounded(S lo, S hi) : lower_bound(lo), upper_bound(hi) {}
  S project(const S& x) const {
    const S m = x < lower_bound ? lower_bound : x;
    return upper_bound < m ? upper_bound : m;
  }
  S operator()(const S& x) const { return project(x); }
  mhip_space c_space() const { return {MHIP_SPACE_BOUNDED, lower_bound, upper_bound}; }
};
}  // namespace space

/// Dense row-major n x n operator (the rank-2 Kokkos::View path, convex.hpp:168-174)
struct DenseMatrix {
  const double* data;
  size_t n;
  size_t extent(int) const { return n; }
};

/// Backend for device vectors on the HIP library: the static interface of convex::KokkosBackend (convex.hpp:141-285)
struct HipBackend {
  using scalar_t = double;
  using vector_t = DeviceVector;

  static size_t vector_size(const vector_t& x) { return x.size(); }
  static void deep_copy(vector_t& dst, const vector_t& src) {
    if (dst.size() != src.size()) throw std::invalid_argument("deep_copy: size mismatch");
    check(mhip_deep_copy(src.size(), dst.data(), src.data(), nullptr));
  }
  template <class Op, class = void>
  struct has_apply_member : std::false_type {};
  template <class Op>
  struct has_apply_member<Op, std::void_t<decltype(std::declval<const Op&>().apply(std::declval<const vector_t&>(),
                                                                                  std::declval<vector_t&>()))>>
      : std::true_type {};
  // Path 1: dense matrix -> gemv; Path 2: op.apply(x, y); Path 3: neither -> std::logic_error (convex.hpp:166-199)
  template <class LinearOp>
  static void apply(const LinearOp& op, const vector_t& x, vector_t& y) {
    if constexpr (std::is_same_v<LinearOp, DenseMatrix>) {
      if (op.extent(1) != x.extent(0)) throw std::invalid_argument("gemv: dimension mismatch A(:,1) vs x");
      if (op.extent(0) != y.extent(0)) throw std::invalid_argument("gemv: dimension mismatch A(0,:) vs y");
      check(mhip_gemv(op.n, op.data, x.data(), y.data(), nullptr));
    } else if constexpr (has_apply_member<LinearOp>::value) {
      op.apply(x, y);
    } else {
      throw std::logic_error("HipBackend::apply: op must be a DenseMatrix or provide void apply(x,y).");
    }
  }
  static void axpby(scalar_t alpha, const vector_t& x, scalar_t beta, vector_t& y) {
    if (x.size() != y.size()) throw std::invalid_argument("x and y must have the same size.");
    check(mhip_axpby(x.size(), alpha, x.data(), beta, y.data(), nullptr));
  }
  template <class Space>
  static void wrapped_axpbyz(scalar_t alpha, const vector_t& x, scalar_t beta, const vector_t& y, vector_t& z,
                             const Space& space) {
    if (x.size() != y.size() || x.size() != z.size())
      throw std::invalid_argument("x, y, and z must have the same size.");
    const mhip_space sp = space.c_space();
    check(mhip_wrapped_axpbyz(x.size(), alpha, x.data(), beta, y.data(), z.data(), &sp, nullptr));
  }
  static scalar_t diff_dot(const vector_t& x, const vector_t& y) {
    if (x.size() != y.size()) throw std::invalid_argument("x and y must have the same size.");
    double r = 0;
    check(mhip_diff_dot2(x.size(), x.data(), y.data(), &r, nullptr));
    return r;
  }
  static scalar_t diff_dot(const vector_t& x1, const vector_t& x2, const vector_t& y1, const vector_t& y2) {
    if (x1.size() != x2.size() || x1.size() != y1.size() || x1.size() != y2.size())
      throw std::invalid_argument("x1, x2, y1, and y2 must have the same size.");
    double r = 0;
    check(mhip_diff_dot4(x1.size(), x1.data(), x2.data(), y1.data(), y2.data(), &r, nullptr));
    return r;
  }
};

template <class Backend, class LinearOp, class ConvexSpace>
class CQPPProblem {  // convex.hpp:363-388: holds references, owns nothing
 public:
  using backend_t = Backend;
  using vector_t = typename Backend::vector_t;
  CQPPProblem(Backend, const LinearOp& A, const vector_t& q, const ConvexSpace& space) : A_(A), q_(q), space_(space) {}
  Backend backend() const { return Backend{}; }
  const LinearOp& A() const { return A_; }
  const vector_t& q() const { return q_; }
  const ConvexSpace& space() const { return space_; }

 private:
  const LinearOp& A_;
  const vector_t& q_;
  const ConvexSpace& space_;
};
template <class Backend, class LinearOp>
class LCPProblem {  // convex.hpp:401-422
 public:
  using backend_t = Backend;
  using vector_t = typename Backend::vector_t;
  LCPProblem(Backend, const LinearOp& A, const vector_t& q) : A_(A), q_(q) {}
  Backend backend() const { return Backend{}; }
  const LinearOp& A() const { return A_; }
  const vector_t& q() const { return q_; }

 private:
  const LinearOp& A_;
  const vector_t& q_;
};
template <class Backend, class LinearOp>
auto to_cqpp(const LCPProblem<Backend, LinearOp>& P) {  // convex.hpp:424-428
  static const space::LowerBound<double> Rn_plus{0.0};
  return CQPPProblem<Backend, LinearOp, space::LowerBound<double>>(P.backend(), P.A(), P.q(), Rn_plus);
}

struct LinfNormProjectedGradientResidual {  // convex.hpp:434-466
  static constexpr int kind = MHIP_RESIDUAL_PROJECTED_GRADIENT;
};
struct LinfNormProjectedDiffResidual {  // convex.hpp:468-496
  static constexpr int kind = MHIP_RESIDUAL_PROJECTED_DIFF;
};
template <class Policy, class Space>
double evaluate_residual(Policy, const DeviceVector& x, const DeviceVector& grad, const Space& space) {
  const mhip_space sp = space.c_space();
  double r = 0;
  check(mhip_residual(x.size(), Policy::kind, x.data(), grad.data(), &sp, &r, nullptr));
  return r;
}
struct BBStepStrategy {  // convex.hpp:498-516
  double operator()(HipBackend, const DeviceVector& x_old, const DeviceVector& g_old, const DeviceVector& x,
                    const DeviceVector& g) const {
    double r = 0;
    check(mhip_bb_step(x.size(), x_old.data(), g_old.data(), x.data(), g.data(), &r, nullptr));
    return r;
  }
};

template <class Scalar = double>
struct PGDConfig {  // convex.hpp:519-525
  unsigned max_iters{1000};
  Scalar tol{1e-8};
};
template <class Scalar = double>
struct SolveResult {  // convex.hpp:527-534
  unsigned num_iters{0};
  Scalar residual{0};
  bool converged{false};
};

template <class Backend>
class PGDState {  // convex.hpp:543-590: four caller-owned vectors by reference
 public:
  using vector_t = typename Backend::vector_t;
  PGDState(const Backend&, vector_t& x, vector_t& g, vector_t& x_tmp, vector_t& g_tmp)
      : x_(x), g_(g), x_tmp_(x_tmp), g_tmp_(g_tmp) {}
  vector_t& x() { return x_; }
  vector_t& grad() { return g_; }
  vector_t& x_tmp() { return x_tmp_; }
  vector_t& grad_tmp() { return g_tmp_; }
  unsigned& iter() { return iter_; }
  bool& converged() { return converged_; }
  double& residual() { return residual_; }
  double& step_size() { return step_size_; }
  unsigned iter() const { return iter_; }
  bool converged() const { return converged_; }
  double residual() const { return residual_; }

 private:
  vector_t &x_, &g_, &x_tmp_, &g_tmp_;
  unsigned iter_{0};
  bool converged_{false};
  double residual_{0}, step_size_{1};
};

template <class Backend, class StepPolicy, class ResidualPolicy>
class PGDStrategy {  // convex.hpp:592-681, kernel by kernel through the backend
 public:
  using config_t = PGDConfig<double>;
  using state_t = PGDState<Backend>;
  using result_t = SolveResult<double>;
  using residual_policy_t = ResidualPolicy;
  PGDStrategy(Backend, StepPolicy step, ResidualPolicy resid, config_t cfg = {}) : step_(step), resid_(resid), cfg_(cfg) {}
  const config_t& config() const { return cfg_; }

  template <class Problem>
  void initialize(const Problem& prob, state_t& state) const {
    Backend::deep_copy(state.x_tmp(), state.x());
    Backend::apply(prob.A(), state.x_tmp(), state.grad_tmp());
    Backend::axpby(1.0, prob.q(), 1.0, state.grad_tmp());
    state.residual() = evaluate_residual(resid_, state.x_tmp(), state.grad_tmp(), prob.space());
    state.step_size() = 1.0 / state.residual();
    state.iter() = 0;
    state.converged() = (state.residual() <= cfg_.tol);
    if (state.converged()) Backend::deep_copy(state.grad(), state.grad_tmp());
  }
  template <class Problem>
  bool iterate(const Problem& prob, state_t& state) const {
    if (state.converged() || state.iter() >= cfg_.max_iters) return state.converged();
    Backend::wrapped_axpbyz(1.0, state.x_tmp(), -state.step_size(), state.grad_tmp(), state.x(), prob.space());
    Backend::apply(prob.A(), state.x(), state.grad());
    Backend::axpby(1.0, prob.q(), 1.0, state.grad());
    state.residual() = evaluate_residual(resid_, state.x(), state.grad(), prob.space());
    if (state.residual() <= cfg_.tol) {
      state.converged() = true;
      return true;
    }
    state.step_size() = step_(Backend{}, state.x_tmp(), state.grad_tmp(), state.x(), state.grad());
    Backend::deep_copy(state.x_tmp(), state.x());
    Backend::deep_copy(state.grad_tmp(), state.grad());
    ++state.iter();
    return false;
  }
  bool done(const state_t& state) const { return state.converged() || state.iter() >= cfg_.max_iters; }
  result_t result(const state_t& state) const { return {state.iter(), state.residual(), state.converged()}; }

 private:
  StepPolicy step_;
  ResidualPolicy resid_;
  config_t cfg_;
};

}  // namespace convex

// ---- make_* / solve_* (convex.hpp:722-845) -------------------------------------------------------------------------------
template <class LinearOp, class ConvexSpace>
auto make_hip_cqpp(const LinearOp& A, const DeviceVector& q, const ConvexSpace& space) {
  return convex::CQPPProblem<convex::HipBackend, LinearOp, ConvexSpace>(convex::HipBackend{}, A, q, space);
}
template <class LinearOp>
auto make_hip_lcp(const LinearOp& A, const DeviceVector& q) {
  return convex::LCPProblem<convex::HipBackend, LinearOp>(convex::HipBackend{}, A, q);
}
template <class Backend, class StepPolicy, class ResidualPolicy>
auto make_pgd_solution_strategy(const Backend& b, const StepPolicy& s, const ResidualPolicy& r,
                                const convex::PGDConfig<double>& cfg = {}) {
  return convex::PGDStrategy<Backend, StepPolicy, ResidualPolicy>(b, s, r, cfg);
}
template <class Backend>
auto make_pgd_solution_strategy(const Backend& b, const convex::PGDConfig<double>& cfg = {}) {
  return convex::PGDStrategy<Backend, convex::BBStepStrategy, convex::LinfNormProjectedDiffResidual>(
      b, convex::BBStepStrategy{}, convex::LinfNormProjectedDiffResidual{}, cfg);
}
template <class Backend>
auto make_pgd_state(const Backend& b, typename Backend::vector_t& x, typename Backend::vector_t& grad,
                    typename Backend::vector_t& x_tmp, typename Backend::vector_t& grad_tmp) {
  return convex::PGDState<Backend>(b, x, grad, x_tmp, grad_tmp);
}

/// solve_cqpp (convex.hpp:789-797).  Generic operators run the reference's loop through the backend; a
/// ContactOperator with the default BB step takes the fused device-resident driver (same algorithm, 3 launches per
/// iteration, no host round trip).
template <class Problem, class Strategy>
auto solve_cqpp(const Problem& prob, const Strategy& strat, typename Strategy::state_t& state) ->
    typename Strategy::result_t {
  using op_t = std::decay_t<decltype(prob.A())>;
  if constexpr (std::is_same_v<op_t, ContactOperator>) {
    const mhip_space sp = prob.space().c_space();
    const mhip_pgd_config cfg{strat.config().max_iters, strat.config().tol, Strategy::residual_policy_t::kind};
    mhip_solve_result r{};
    check(mhip_bbpgd_solve_contact(prob.A().handle(), prob.q().data(), &sp, &cfg, state.x().data(), state.grad().data(),
                                   state.x_tmp().data(), state.grad_tmp().data(), &r, nullptr));
    state.iter() = r.num_iters;
    state.residual() = r.residual;
    state.converged() = r.converged != 0;
    return {r.num_iters, r.residual, r.converged != 0};
  } else {
    strat.initialize(prob, state);
    while (!strat.done(state)) {
      if (strat.iterate(prob, state)) break;
    }
    return strat.result(state);
  }
}
/// solve_lcp (convex.hpp:839-845)
template <class Problem, class Strategy>
auto solve_lcp(const Problem& prob, const Strategy& strat, typename Strategy::state_t& state) ->
    typename Strategy::result_t {
  auto cqpp_prob = convex::to_cqpp(prob);
  return solve_cqpp(cqpp_prob, strat, state);
}

}  // namespace mundy_hip
