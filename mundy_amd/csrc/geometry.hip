// geometry.hip -- per-body AABB / segment-record kernels and the narrow-phase (contact generation) kernels.
// All kernels are HBM-bound streaming or gather kernels: one thread per body / per pair, grid-stride, AoS rows of
// 3/4/6/8 doubles read as contiguous runs so a wavefront's accesses cover whole cache lines.
#include "geom_device.hpp"

namespace mhip {

__device__ inline void store_box(double* aabb, size_t i, const Box& b) {
  double2* o = reinterpret_cast<double2*>(aabb + 6 * i);  // 48-byte rows: 16-byte aligned
  o[0] = make_double2(b.lo.x, b.lo.y);
  o[1] = make_double2(b.lo.z, b.hi.x);
  o[2] = make_double2(b.hi.y, b.hi.z);
}

__global__ void __launch_bounds__(kBlock) k_aabb_spheres(size_t n, const double* __restrict__ center,
                                                        const double* __restrict__ radius,
                                                        double* __restrict__ aabb) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    store_box(aabb, i, aabb_sphere(load3(center, i), radius[i]));
}

__global__ void __launch_bounds__(kBlock)
    k_aabb_spherocylinders(size_t n, const double* __restrict__ center, const double* __restrict__ quat,
                           const double* __restrict__ radius, const double* __restrict__ length,
                           double* __restrict__ aabb) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const V3 c = load3(center, i);
    const V3 d = rod_half_axis(load4q(quat, i), length[i]);
    store_box(aabb, i, aabb_segment(c - d, c + d, radius[i]));
  }
}

__global__ void __launch_bounds__(kBlock)
    k_aabb_ellipsoids(size_t n, const double* __restrict__ center, const double* __restrict__ quat,
                      const double* __restrict__ radii, double* __restrict__ aabb) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    store_box(aabb, i, aabb_ellipsoid(load3(center, i), load4q(quat, i), load3(radii, i)));
}

__global__ void __launch_bounds__(kBlock)
    k_aabb_ellipsoids_conservative(size_t n, const double* __restrict__ center, const double* __restrict__ quat,
                                   const double* __restrict__ radii, double* __restrict__ aabb) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    store_box(aabb, i, aabb_ellipsoid_conservative(load3(center, i), load4q(quat, i), load3(radii, i)));
}

struct SegRec {
  V3 p0, p1;
  double r;
};
__device__ inline SegRec load_seg(const double* seg, size_t i) {
  const double2* s = reinterpret_cast<const double2*>(seg + 8 * i);  // 64-byte records
  const double2 a = s[0], b = s[1], c = s[2], d = s[3];
  return {{a.x, a.y, b.x}, {b.y, c.x, c.y}, d.x};
}

__global__ void __launch_bounds__(kBlock) k_aabb_segments(size_t n, const double* __restrict__ seg,
                                                         double* __restrict__ aabb) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const SegRec s = load_seg(seg, i);
    store_box(aabb, i, aabb_segment(s.p0, s.p1, s.r));
  }
}

__global__ void __launch_bounds__(kBlock)
    k_bounding_radius_rods(size_t n, const double* __restrict__ radius, const double* __restrict__ length,
                           double* __restrict__ out) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    out[i] = 0.5 * length[i] + radius[i];  // compute_bounding_radius.hpp:82-90
}
__global__ void __launch_bounds__(kBlock)
    k_bounding_radius_ellipsoids(size_t n, const double* __restrict__ radii, double* __restrict__ out) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const V3 r = load3(radii, i);
    out[i] = dmax(r.x, dmax(r.y, r.z));  // compute_bounding_radius.hpp:74-79
  }
}

__global__ void __launch_bounds__(kBlock)
    k_rod_segments(size_t n, const double* __restrict__ center, const double* __restrict__ quat,
                   const double* __restrict__ radius, const double* __restrict__ length, double* __restrict__ seg) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const V3 c = load3(center, i);
    const V3 d = rod_half_axis(load4q(quat, i), length[i]);
    const V3 p0 = c - d, p1 = c + d;
    double2* o = reinterpret_cast<double2*>(seg + 8 * i);
    o[0] = make_double2(p0.x, p0.y);
    o[1] = make_double2(p0.z, p1.x);
    o[2] = make_double2(p1.y, p1.z);
    o[3] = make_double2(radius[i], 0.0);
  }
}

// ---- element-wise distance batches ----------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock)
    k_dist_sphere_sphere(size_t n, const double* __restrict__ c1, const double* __restrict__ r1,
                         const double* __restrict__ c2, const double* __restrict__ r2, double* __restrict__ dist,
                         double* __restrict__ sep) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    // distance(Sphere, Sphere, sep): SphereSphere.hpp:66-76
    V3 s;
    const double cc = dist_point_point(load3(c1, i), load3(c2, i), s);
    const double d = cc - r1[i] - r2[i];
    if (dist) dist[i] = d;
    if (sep) store3(sep, i, s * (d / cc));
  }
}

__global__ void __launch_bounds__(kBlock)
    k_dist_point_sphere(size_t n, const double* __restrict__ p, const double* __restrict__ c,
                        const double* __restrict__ r, double* __restrict__ dist, double* __restrict__ sep) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    // distance(Point, Sphere, sep): PointSphere.hpp:69-79 (sep from the point to the surface; NaN when the point is
    // the centre, as in the reference: 0 * (d / 0))
    V3 s;
    const double cc = dist_point_point(load3(p, i), load3(c, i), s);
    const double d = cc - r[i];
    if (dist) dist[i] = d;
    if (sep) store3(sep, i, s * (d / cc));
  }
}

__global__ void __launch_bounds__(kBlock)
    k_dist_segment_sphere(size_t n, const double* __restrict__ a0, const double* __restrict__ a1,
                          const double* __restrict__ c, const double* __restrict__ r, double* __restrict__ dist,
                          double* __restrict__ cp, double* __restrict__ t, double* __restrict__ sep) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    // distance(LineSegment, Sphere, cp, t, sep): LineSegmentSphere.hpp:88-100 -- distance(sphere.center(), segment,
    // ...) rescaled to the surface; the separation is the one that routine hands back (centre -> closest point)
    V3 cl, s;
    double tt;
    const double lc = dist_point_segment(load3(c, i), load3(a0, i), load3(a1, i), cl, tt, s);
    const double d = lc - r[i];
    if (dist) dist[i] = d;
    if (cp) store3(cp, i, cl);
    if (t) t[i] = tt;
    if (sep) store3(sep, i, s * (d / lc));
  }
}

__global__ void __launch_bounds__(kBlock)
    k_dist_point_segment(size_t n, const double* __restrict__ p, const double* __restrict__ a0,
                         const double* __restrict__ a1, double* __restrict__ dist, double* __restrict__ cp,
                         double* __restrict__ t, double* __restrict__ sep) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    V3 c, s;
    double tt;
    const double d = dist_point_segment(load3(p, i), load3(a0, i), load3(a1, i), c, tt, s);
    if (dist) dist[i] = d;
    if (cp) store3(cp, i, c);
    if (t) t[i] = tt;
    if (sep) store3(sep, i, s);
  }
}

__global__ void __launch_bounds__(kBlock)
    k_dist_segment_segment(size_t n, const double* __restrict__ a0, const double* __restrict__ a1,
                           const double* __restrict__ b0, const double* __restrict__ b1, double* __restrict__ dist,
                           double* __restrict__ cp1, double* __restrict__ cp2, double* __restrict__ s,
                           double* __restrict__ t, double* __restrict__ sep) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const SegSeg r = dist_segment_segment(load3(a0, i), load3(a1, i), load3(b0, i), load3(b1, i));
    if (dist) dist[i] = r.dist;
    if (cp1) store3(cp1, i, r.cp1);
    if (cp2) store3(cp2, i, r.cp2);
    if (s) s[i] = r.s;
    if (t) t[i] = r.t;
    if (sep) store3(sep, i, r.sep);
  }
}

// ---- contact generation over a neighbour list -------------------------------------------------------------------------
// spheres: algorithmic bytes per contact = pair 8 + 2 x (centre 24 + radius 8) gathered + sep 8 + normal 24 = 104 B
template <bool PERIODIC, class Metric>
__global__ void __launch_bounds__(kBlock)
    k_contact_spheres(size_t nc, const int2* __restrict__ pairs, const double* __restrict__ center,
                      const double* __restrict__ radius, Metric pm, double* __restrict__ sep,
                      double* __restrict__ normal) {
  for (size_t c = blockIdx.x * (size_t)blockDim.x + threadIdx.x; c < nc; c += (size_t)gridDim.x * blockDim.x) {
    const int2 ij = pairs[c];
    const V3 ci = load3(center, ij.x), cj = load3(center, ij.y);
    const V3 d = PERIODIC ? periodic_sep(pm, ci, cj) : (cj - ci);
    const double cc = norm(d);
    const double inv = 1.0 / cc;  // NgpLcp.cpp:369-372
    if (sep) sep[c] = cc - radius[ij.x] - radius[ij.y];  // SphereSphere.hpp:58
    if (normal) store3(normal, c, d * inv);
  }
}

// rods: algorithmic bytes per contact = pair 8 + 2 x 64 B segment records + 2 x 24 B centres gathered
//       + sep 8 + normal 24 + lever arms 48 (+ optional cp 48, s/t 16)
// PERIODIC: the second rod is taken at the lattice image whose centre is nearest to the first rod's centre
// (PeriodicScaledMetric::sep of the centres, periodicity.hpp:812-816; rigid translation as wrap_rigid moves a
// spherocylinder, :1094-1113): shift = (c_i + sep(c_i, c_j)) - c_j is added to both of its endpoints, and its lever arm
// is taken from the shifted centre.  Contact points come out in the first rod's image.
template <bool PERIODIC>
__global__ void __launch_bounds__(kBlock)
    k_contact_rods(size_t nc, const int2* __restrict__ pairs, const double* __restrict__ seg,
                   const double* __restrict__ center, Periodic pm, double* __restrict__ sep,
                   double* __restrict__ normal, double* __restrict__ cp1, double* __restrict__ cp2,
                   double* __restrict__ ra, double* __restrict__ rb, double* __restrict__ s, double* __restrict__ t) {
  for (size_t c = blockIdx.x * (size_t)blockDim.x + threadIdx.x; c < nc; c += (size_t)gridDim.x * blockDim.x) {
    const int2 ij = pairs[c];
    const SegRec a = load_seg(seg, ij.x);
    SegRec b = load_seg(seg, ij.y);
    V3 shift{0.0, 0.0, 0.0};
    if (PERIODIC) {
      const V3 ci = load3(center, ij.x), cj = load3(center, ij.y);
      shift = (ci + periodic_sep(pm, ci, cj)) - cj;
      b.p0 = b.p0 + shift;
      b.p1 = b.p1 + shift;
    }
    const SegSeg r = dist_segment_segment(a.p0, a.p1, b.p0, b.p1);
    const double radius_sum = a.r + b.r;
    const double inv = 1.0 / r.dist;
    if (sep) sep[c] = r.dist - radius_sum;
    // left-to-right vector is cp2 - cp1 (linker kernel :226), not the distance routine's `sep` (see oracle note)
    if (normal) store3(normal, c, (r.cp2 - r.cp1) * inv);
    if (cp1) store3(cp1, c, r.cp1);
    if (cp2) store3(cp2, c, r.cp2);
    if (ra) store3(ra, c, r.cp1 - load3(center, ij.x));
    if (rb) store3(rb, c, r.cp2 - (PERIODIC ? load3(center, ij.y) + shift : load3(center, ij.y)));
    // arclengths of the contact points cp1 / cp2, i.e. of the CLAMPED closest points: in the colinear branch the
    // distance routine's own s / t are unclamped line parameters (mhip_distance_segment_segment returns those raw)
    if (s) s[c] = contact_arclength(r.s);
    if (t) t[c] = contact_arclength(r.t);
  }
}

}  // namespace mhip

using namespace mhip;

#define REQ_PTR(p) MHIP_REQUIRE((p) != nullptr || n == 0, MHIP_ERR_INVALID_ARGUMENT, "%s: %s is null", __func__, #p)

extern "C" {

int mhip_compute_aabb_spheres(size_t n, const double* center, const double* radius, double* aabb,
                              mhip_stream_t stream) {
  REQ_PTR(center); REQ_PTR(radius); REQ_PTR(aabb);
  if (n == 0) return MHIP_SUCCESS;
  k_aabb_spheres<<<grid_for(n), kBlock, 0, as_stream(stream)>>>(n, center, radius, aabb);
  MHIP_LAUNCH_CHECK();
  return MHIP_SUCCESS;
}

int mhip_compute_aabb_spherocylinders(size_t n, const double* center, const double* quat, const double* radius,
                                      const double* length, double* aabb, mhip_stream_t stream) {
  TraceRange trace_range("compute_aabb");
  REQ_PTR(center); REQ_PTR(quat); REQ_PTR(radius); REQ_PTR(length); REQ_PTR(aabb);
  if (n == 0) return MHIP_SUCCESS;
  k_aabb_spherocylinders<<<grid_for(n), kBlock, 0, as_stream(stream)>>>(n, center, quat, radius, length, aabb);
  MHIP_LAUNCH_CHECK();
  return MHIP_SUCCESS;
}

int mhip_compute_aabb_ellipsoids(size_t n, const double* center, const double* quat, const double* radii,
                                 double* aabb, mhip_stream_t stream) {
  REQ_PTR(center); REQ_PTR(quat); REQ_PTR(radii); REQ_PTR(aabb);
  if (n == 0) return MHIP_SUCCESS;
  k_aabb_ellipsoids<<<grid_for(n), kBlock, 0, as_stream(stream)>>>(n, center, quat, radii, aabb);
  MHIP_LAUNCH_CHECK();
  return MHIP_SUCCESS;
}

int mhip_compute_aabb_ellipsoids_conservative(size_t n, const double* center, const double* quat, const double* radii,
                                              double* aabb, mhip_stream_t stream) {
  REQ_PTR(center); REQ_PTR(quat); REQ_PTR(radii); REQ_PTR(aabb);
  if (n == 0) return MHIP_SUCCESS;
  k_aabb_ellipsoids_conservative<<<grid_for(n), kBlock, 0, as_stream(stream)>>>(n, center, quat, radii, aabb);
  MHIP_LAUNCH_CHECK();
  return MHIP_SUCCESS;
}

int mhip_compute_aabb_segments(size_t n, const double* seg, double* aabb, mhip_stream_t stream) {
  REQ_PTR(seg); REQ_PTR(aabb);
  if (n == 0) return MHIP_SUCCESS;
  k_aabb_segments<<<grid_for(n), kBlock, 0, as_stream(stream)>>>(n, seg, aabb);
  MHIP_LAUNCH_CHECK();
  return MHIP_SUCCESS;
}

int mhip_bounding_radius_spherocylinders(size_t n, const double* radius, const double* length, double* out,
                                         mhip_stream_t stream) {
  REQ_PTR(radius); REQ_PTR(length); REQ_PTR(out);
  if (n == 0) return MHIP_SUCCESS;
  k_bounding_radius_rods<<<grid_for(n), kBlock, 0, as_stream(stream)>>>(n, radius, length, out);
  MHIP_LAUNCH_CHECK();
  return MHIP_SUCCESS;
}

int mhip_bounding_radius_ellipsoids(size_t n, const double* radii, double* out, mhip_stream_t stream) {
  REQ_PTR(radii); REQ_PTR(out);
  if (n == 0) return MHIP_SUCCESS;
  k_bounding_radius_ellipsoids<<<grid_for(n), kBlock, 0, as_stream(stream)>>>(n, radii, out);
  MHIP_LAUNCH_CHECK();
  return MHIP_SUCCESS;
}

int mhip_spherocylinder_segments(size_t n, const double* center, const double* quat, const double* radius,
                                 const double* length, double* seg, mhip_stream_t stream) {
  REQ_PTR(center); REQ_PTR(quat); REQ_PTR(radius); REQ_PTR(length); REQ_PTR(seg);
  MHIP_REQUIRE((reinterpret_cast<uintptr_t>(seg) & 15) == 0, MHIP_ERR_INVALID_ARGUMENT, "seg must be 16-byte aligned");
  if (n == 0) return MHIP_SUCCESS;
  k_rod_segments<<<grid_for(n), kBlock, 0, as_stream(stream)>>>(n, center, quat, radius, length, seg);
  MHIP_LAUNCH_CHECK();
  return MHIP_SUCCESS;
}

int mhip_distance_sphere_sphere(size_t n, const double* c1, const double* r1, const double* c2, const double* r2,
                                double* dist, double* sep, mhip_stream_t stream) {
  REQ_PTR(c1); REQ_PTR(r1); REQ_PTR(c2); REQ_PTR(r2);
  if (n == 0) return MHIP_SUCCESS;
  k_dist_sphere_sphere<<<grid_for(n), kBlock, 0, as_stream(stream)>>>(n, c1, r1, c2, r2, dist, sep);
  MHIP_LAUNCH_CHECK();
  return MHIP_SUCCESS;
}

int mhip_distance_point_sphere(size_t n, const double* p, const double* c, const double* r, double* dist, double* sep,
                               mhip_stream_t stream) {
  REQ_PTR(p); REQ_PTR(c); REQ_PTR(r);
  if (n == 0) return MHIP_SUCCESS;
  k_dist_point_sphere<<<grid_for(n), kBlock, 0, as_stream(stream)>>>(n, p, c, r, dist, sep);
  MHIP_LAUNCH_CHECK();
  return MHIP_SUCCESS;
}

int mhip_distance_segment_sphere(size_t n, const double* a0, const double* a1, const double* c, const double* r,
                                 double* dist, double* cp, double* t, double* sep, mhip_stream_t stream) {
  REQ_PTR(a0); REQ_PTR(a1); REQ_PTR(c); REQ_PTR(r);
  if (n == 0) return MHIP_SUCCESS;
  k_dist_segment_sphere<<<grid_for(n), kBlock, 0, as_stream(stream)>>>(n, a0, a1, c, r, dist, cp, t, sep);
  MHIP_LAUNCH_CHECK();
  return MHIP_SUCCESS;
}

int mhip_distance_point_segment(size_t n, const double* p, const double* a0, const double* a1, double* dist,
                                double* cp, double* t, double* sep, mhip_stream_t stream) {
  REQ_PTR(p); REQ_PTR(a0); REQ_PTR(a1);
  if (n == 0) return MHIP_SUCCESS;
  k_dist_point_segment<<<grid_for(n), kBlock, 0, as_stream(stream)>>>(n, p, a0, a1, dist, cp, t, sep);
  MHIP_LAUNCH_CHECK();
  return MHIP_SUCCESS;
}

int mhip_distance_segment_segment(size_t n, const double* a0, const double* a1, const double* b0, const double* b1,
                                  double* dist, double* cp1, double* cp2, double* s, double* t, double* sep,
                                  mhip_stream_t stream) {
  REQ_PTR(a0); REQ_PTR(a1); REQ_PTR(b0); REQ_PTR(b1);
  if (n == 0) return MHIP_SUCCESS;
  k_dist_segment_segment<<<grid_for(n), kBlock, 0, as_stream(stream)>>>(n, a0, a1, b0, b1, dist, cp1, cp2, s, t,
                                                                         sep);
  MHIP_LAUNCH_CHECK();
  return MHIP_SUCCESS;
}

int mhip_contact_spheres(size_t c, const int32_t* pairs, const double* center, const double* radius,
                         const double* box, double* sep, double* normal, mhip_stream_t stream) {
  TraceRange trace_range("contact_spheres");
  const size_t n = c;
  REQ_PTR(pairs); REQ_PTR(center); REQ_PTR(radius);
  if (c == 0) return MHIP_SUCCESS;
  const int2* p2 = reinterpret_cast<const int2*>(pairs);
  if (box) {
    MHIP_REQUIRE(box[0] > 0 && box[1] > 0 && box[2] > 0, MHIP_ERR_INVALID_ARGUMENT, "periodic box must be positive");
    k_contact_spheres<true><<<grid_for(c), kBlock, 0, as_stream(stream)>>>(c, p2, center, radius, make_periodic(box),
                                                                          sep, normal);
  } else {
    const double one[3] = {1, 1, 1};
    k_contact_spheres<false><<<grid_for(c), kBlock, 0, as_stream(stream)>>>(c, p2, center, radius,
                                                                           make_periodic(one), sep, normal);
  }
  MHIP_LAUNCH_CHECK();
  return MHIP_SUCCESS;
}

int mhip_contact_spheres_triclinic(size_t c, const int32_t* pairs, const double* center, const double* radius,
                                   const double* cell, double* sep, double* normal, mhip_stream_t stream) {
  const size_t n = c;
  REQ_PTR(pairs); REQ_PTR(center); REQ_PTR(radius);
  MHIP_REQUIRE(cell != nullptr && determinant3(cell) != 0.0, MHIP_ERR_INVALID_ARGUMENT, "unit cell matrix is singular");
  if (c == 0) return MHIP_SUCCESS;
  k_contact_spheres<true><<<grid_for(c), kBlock, 0, as_stream(stream)>>>(c, reinterpret_cast<const int2*>(pairs), center,
                                                                        radius, make_triclinic(cell), sep, normal);
  MHIP_LAUNCH_CHECK();
  return MHIP_SUCCESS;
}

static int contact_rods(size_t c, const int32_t* pairs, const double* seg, const double* center, const double* box,
                        double* sep, double* normal, double* cp1, double* cp2, double* ra, double* rb, double* s,
                        double* t, mhip_stream_t stream) {
  TraceRange trace_range("contact_spherocylinders");
  const size_t n = c;
  REQ_PTR(pairs); REQ_PTR(seg);
  MHIP_REQUIRE(center != nullptr || (ra == nullptr && rb == nullptr && box == nullptr), MHIP_ERR_INVALID_ARGUMENT,
               "center is required when lever arms or a periodic box are requested");
  MHIP_REQUIRE((reinterpret_cast<uintptr_t>(seg) & 15) == 0, MHIP_ERR_INVALID_ARGUMENT, "seg must be 16-byte aligned");
  if (box)
    MHIP_REQUIRE(box[0] > 0 && box[1] > 0 && box[2] > 0, MHIP_ERR_INVALID_ARGUMENT, "periodic box must be positive");
  if (c == 0) return MHIP_SUCCESS;
  const int2* p2 = reinterpret_cast<const int2*>(pairs);
  const double one[3] = {1, 1, 1};
  if (box)
    k_contact_rods<true><<<grid_for(c), kBlock, 0, as_stream(stream)>>>(c, p2, seg, center, make_periodic(box), sep,
                                                                        normal, cp1, cp2, ra, rb, s, t);
  else
    k_contact_rods<false><<<grid_for(c), kBlock, 0, as_stream(stream)>>>(c, p2, seg, center, make_periodic(one), sep,
                                                                         normal, cp1, cp2, ra, rb, s, t);
  MHIP_LAUNCH_CHECK();
  return MHIP_SUCCESS;
}

int mhip_contact_spherocylinders(size_t c, const int32_t* pairs, const double* seg, const double* center, double* sep,
                                 double* normal, double* cp1, double* cp2, double* ra, double* rb, double* s,
                                 double* t, mhip_stream_t stream) {
  return contact_rods(c, pairs, seg, center, nullptr, sep, normal, cp1, cp2, ra, rb, s, t, stream);
}

int mhip_contact_spherocylinders_periodic(size_t c, const int32_t* pairs, const double* seg, const double* center,
                                          const double* box, double* sep, double* normal, double* cp1, double* cp2,
                                          double* ra, double* rb, double* s, double* t, mhip_stream_t stream) {
  MHIP_REQUIRE(box != nullptr, MHIP_ERR_INVALID_ARGUMENT, "box is null");
  return contact_rods(c, pairs, seg, center, box, sep, normal, cp1, cp2, ra, rb, s, t, stream);
}

}  // extern "C"
