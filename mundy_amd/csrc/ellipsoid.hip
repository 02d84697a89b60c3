// ellipsoid.hip -- ellipsoid narrow phase: batch and neighbour-list entry points over
// ellipsoid_device.hpp.  One thread per pair; 64-thread workgroups keep the divergent L-BFGS loops of different
// pairs on as many SIMDs as possible.  Compute/latency bound (fp64 vector + transcendental), priced against the
// fp64 vector peak, not HBM.
#include "ellipsoid_device.hpp"

namespace mhip {

constexpr int kEllBlock = 64;
// Register budget: unconstrained, the compiler takes 314-370 VGPRs + AGPRs (one wave per SIMD) and still spills the
// L-BFGS history (dynamically indexed) to scratch; two waves per SIMD (256 registers) measured +22 % pairs/s, three
// and four waves spill too much (MI355X, 4*10^5 pairs: 4.6 / 5.6 / 5.3 / 4.5 *10^6 pairs/s for 1 / 2 / 3 / 4 waves).
#ifndef ELL_WAVES
#define ELL_WAVES 2
#endif
#define ELL_OCC __attribute__((amdgpu_waves_per_eu(ELL_WAVES)))

__device__ inline EllipsoidD load_ellipsoid(const double* c, const double* q, const double* r, size_t i) {
  return {load3(c, i), load4q(q, i), load3(r, i)};
}

__global__ void __launch_bounds__(kEllBlock) ELL_OCC
    k_dist_ellipsoids(size_t n, const double* __restrict__ c1, const double* __restrict__ q1,
                      const double* __restrict__ r1, const double* __restrict__ c2, const double* __restrict__ q2,
                      const double* __restrict__ r2, double* __restrict__ dist, double* __restrict__ cp1,
                      double* __restrict__ cp2, double* __restrict__ n1, double* __restrict__ n2) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const EllipsoidPair r = dist_ellipsoid_ellipsoid(load_ellipsoid(c1, q1, r1, i), load_ellipsoid(c2, q2, r2, i));
  if (dist) dist[i] = r.dist;
  if (cp1) store3(cp1, i, r.cp1);
  if (cp2) store3(cp2, i, r.cp2);
  if (n1) store3(n1, i, r.n1);
  if (n2) store3(n2, i, V3{-r.n1.x, -r.n1.y, -r.n1.z});
}

__global__ void __launch_bounds__(kEllBlock) ELL_OCC
    k_dist_point_ellipsoid(size_t n, const double* __restrict__ p, const double* __restrict__ c,
                           const double* __restrict__ q, const double* __restrict__ r, double* __restrict__ dist,
                           double* __restrict__ cp, double* __restrict__ nrm) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  V3 closest, normal;
  const double d = dist_point_ellipsoid(load3(p, i), load_ellipsoid(c, q, r, i), closest, normal);
  if (dist) dist[i] = d;
  if (cp) store3(cp, i, closest);
  if (nrm) store3(nrm, i, normal);
}

// contact generation over a neighbour list: sep = shared-normal signed distance, normal = n1 (outward normal of the
// source ellipsoid), contact points = the two foot points, lever arms about the body centres.
__global__ void __launch_bounds__(kEllBlock) ELL_OCC
    k_contact_ellipsoids(size_t nc, const int2* __restrict__ pairs, const double* __restrict__ center,
                         const double* __restrict__ quat, const double* __restrict__ radii, double* __restrict__ sep,
                         double* __restrict__ normal, double* __restrict__ cp1, double* __restrict__ cp2,
                         double* __restrict__ ra, double* __restrict__ rb) {
  const size_t c = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (c >= nc) return;
  const int2 ij = pairs[c];
  const EllipsoidD e1 = load_ellipsoid(center, quat, radii, ij.x), e2 = load_ellipsoid(center, quat, radii, ij.y);
  const EllipsoidPair r = dist_ellipsoid_ellipsoid(e1, e2);
  if (sep) sep[c] = r.dist;
  if (normal) store3(normal, c, r.n1);
  if (cp1) store3(cp1, c, r.cp1);
  if (cp2) store3(cp2, c, r.cp2);
  if (ra) store3(ra, c, r.cp1 - e1.c);
  if (rb) store3(rb, c, r.cp2 - e2.c);
}

}  // namespace mhip

using namespace mhip;

extern "C" {

int mhip_distance_ellipsoid_ellipsoid(size_t n, const double* c1, const double* q1, const double* r1,
                                      const double* c2, const double* q2, const double* r2, double* dist, double* cp1,
                                      double* cp2, double* n1, double* n2, mhip_stream_t stream) {
  if (n == 0) return MHIP_SUCCESS;
  MHIP_REQUIRE(c1 && q1 && r1 && c2 && q2 && r2, MHIP_ERR_INVALID_ARGUMENT, "ellipsoid arrays must not be null");
  k_dist_ellipsoids<<<grid_exact(n, kEllBlock), kEllBlock, 0, as_stream(stream)>>>(n, c1, q1, r1, c2, q2, r2, dist,
                                                                                   cp1, cp2, n1, n2);
  MHIP_LAUNCH_CHECK();
  return MHIP_SUCCESS;
}

int mhip_distance_point_ellipsoid(size_t n, const double* p, const double* c, const double* q, const double* r,
                                  double* dist, double* cp, double* normal, mhip_stream_t stream) {
  if (n == 0) return MHIP_SUCCESS;
  MHIP_REQUIRE(p && c && q && r, MHIP_ERR_INVALID_ARGUMENT, "point / ellipsoid arrays must not be null");
  k_dist_point_ellipsoid<<<grid_exact(n, kEllBlock), kEllBlock, 0, as_stream(stream)>>>(n, p, c, q, r, dist, cp,
                                                                                        normal);
  MHIP_LAUNCH_CHECK();
  return MHIP_SUCCESS;
}

int mhip_contact_ellipsoids(size_t c, const int32_t* pairs, const double* center, const double* quat,
                            const double* radii, double* sep, double* normal, double* cp1, double* cp2, double* ra,
                            double* rb, mhip_stream_t stream) {
  if (c == 0) return MHIP_SUCCESS;
  MHIP_REQUIRE(pairs && center && quat && radii, MHIP_ERR_INVALID_ARGUMENT, "pairs / ellipsoid arrays must not be null");
  k_contact_ellipsoids<<<grid_exact(c, kEllBlock), kEllBlock, 0, as_stream(stream)>>>(
      c, reinterpret_cast<const int2*>(pairs), center, quat, radii, sep, normal, cp1, cp2, ra, rb);
  MHIP_LAUNCH_CHECK();
  return MHIP_SUCCESS;
}

}  // extern "C"
