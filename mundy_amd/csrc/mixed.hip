// mixed.hip -- bodies of mixed shape (BASELINE configs[4]: sphere / spherocylinder / ellipsoid): kind-dispatched AABBs
// and contact generation with the pairs binned by shape class, so that each launch runs ONE distance routine and
// wavefronts do not diverge between a 30-flop sphere test and a 10^5-flop ellipsoid minimisation.
//   classes (lower kind first): S-S, S-R, S-E, R-R, R-E, E-E
//   1. k_pair_class        class id per pair
//   2. 6 x (flag, scan, scatter)   stable counting sort of pair indices by class (device-resident class offsets)
//   3. one kernel per class over its index range, results scattered back to the pairs' own slots
// S-E and R-E have no reference implementation (empty stubs SphereEllipsoid.hpp / LineSegmentEllipsoid.hpp): build
// extensions, parity unpinned (see the oracle).  S-E = the signed distance of the sphere's centre to the ellipsoid minus
// the radius: by default the exact distance in closed form (segment_ellipsoid.hpp; a rod of zero length), on request
// (mhip_contact_mixed_set_sphere_ellipsoid_route(1)) the reference's own point - ellipsoid routine, the nine-start
// L-BFGS of PointEllipsoid.hpp:94-135 (SURVEY 8f.4's routing), which the closed form matches to that routine's 1e-4;
// R-E = the closest approach of the rod's centreline to the ellipsoid in closed form, minus the radius.  kind: 0 sphere, 1 spherocylinder,
// 2 ellipsoid; shape [n][3] = (r,-,-) / (r,L,-) / (r1,r2,r3).
#include <atomic>

#include "ellipsoid_lockstep.hpp"
#include "segment_ellipsoid.hpp"

// This file is compiled twice.  mixed.hip itself: everything, under the library's -ffp-contract=off (every a*b+c two
// roundings: results bit-identical to the scalar reference order).  mixed_fma.hip (which only includes this file with
// MHIP_MIXED_FMA_TU defined): the three MINIMISATION classes alone, under -ffp-contract=fast -- a labelled build of the
// compute-bound kernels for callers that take the reference's own 1e-4 tolerance of the ellipsoid distances
// (UnitTestEllipsoidEllipsoid.cpp:53) instead of bit parity with the oracle; selected at run time by
// mhip_contact_mixed_set_contraction, never the default.
#ifdef MHIP_MIXED_FMA_TU
#define MHIP_LOCKSTEP_KERNEL k_contact_class_lockstep_fma
#define MHIP_LOCKSTEP_LAUNCH launch_contact_classes_lockstep_fma
#else
#define MHIP_LOCKSTEP_KERNEL k_contact_class_lockstep
#define MHIP_LOCKSTEP_LAUNCH launch_contact_classes_lockstep
#endif

namespace mhip {

struct BodyD {
  int kind;
  V3 c;
  Quat q;
  V3 s;
};
__device__ inline BodyD load_body(const int32_t* kind, const double* c, const double* q, const double* shape,
                                  size_t i) {
  return {kind[i], load3(c, i), load4q(q, i), load3(shape, i)};
}

#ifndef MHIP_MIXED_FMA_TU
template <bool CONSERVATIVE>
__global__ void __launch_bounds__(kBlock)
    k_aabb_mixed(size_t n, const int32_t* __restrict__ kind, const double* __restrict__ center,
                 const double* __restrict__ quat, const double* __restrict__ shape, double* __restrict__ aabb,
                 double* __restrict__ brad) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const BodyD b = load_body(kind, center, quat, shape, i);
    Box bx;
    double R;
    if (b.kind == 0) {
      bx = aabb_sphere(b.c, b.s.x);
      R = b.s.x;
    } else if (b.kind == 1) {
      const V3 d = rod_half_axis(b.q, b.s.y);
      bx = aabb_segment(b.c - d, b.c + d, b.s.x);
      R = 0.5 * b.s.y + b.s.x;
    } else {
      bx = CONSERVATIVE ? aabb_ellipsoid_conservative(b.c, b.q, b.s) : aabb_ellipsoid(b.c, b.q, b.s);
      R = dmax(b.s.x, dmax(b.s.y, b.s.z));
    }
    double* o = aabb + 6 * i;
    o[0] = bx.lo.x; o[1] = bx.lo.y; o[2] = bx.lo.z; o[3] = bx.hi.x; o[4] = bx.hi.y; o[5] = bx.hi.z;
    if (brad) brad[i] = R;
  }
}

__device__ inline int class_of(int ka, int kb) {  // ka <= kb -> 0..5
  const int t = ka * 3 + kb;                      // 0 SS, 1 SR, 2 SE, 4 RR, 5 RE, 8 EE
  return t < 3 ? t : (t < 6 ? t - 1 : 5);
}

__global__ void __launch_bounds__(kBlock) k_pair_class(size_t nc, const int2* __restrict__ pairs,
                                                      const int32_t* __restrict__ kind, int32_t* __restrict__ cls) {
  for (size_t k = blockIdx.x * (size_t)blockDim.x + threadIdx.x; k < nc; k += (size_t)gridDim.x * blockDim.x) {
    const int2 ij = pairs[k];
    const int a = kind[ij.x], b = kind[ij.y];
    cls[k] = class_of(a < b ? a : b, a < b ? b : a);
  }
}
__global__ void __launch_bounds__(kBlock) k_class_flags(size_t nc, const int32_t* __restrict__ cls, int which,
                                                       int32_t* __restrict__ flags) {
  for (size_t k = blockIdx.x * (size_t)blockDim.x + threadIdx.x; k < nc; k += (size_t)gridDim.x * blockDim.x)
    flags[k] = (cls[k] == which) ? 1 : 0;
}
__global__ void __launch_bounds__(kBlock)
    k_class_scatter(size_t nc, const int32_t* __restrict__ flags, const int32_t* __restrict__ pos, int which,
                    int32_t* __restrict__ class_start, int32_t* __restrict__ order) {
  const int32_t base = class_start[which];
  for (size_t k = blockIdx.x * (size_t)blockDim.x + threadIdx.x; k < nc; k += (size_t)gridDim.x * blockDim.x)
    if (flags[k]) order[base + pos[k]] = static_cast<int32_t>(k);
  if (blockIdx.x == 0 && threadIdx.x == 0) class_start[which + 1] = base + pos[nc];
}

#endif  // !MHIP_MIXED_FMA_TU

struct MixedOut {
  double *sep, *normal, *cp1, *cp2, *ra, *rb;
  // periodic box: body j is taken at the lattice image whose centre is nearest to body i's centre
  // (c_j' = c_i + PeriodicScaledMetric::sep(c_i, c_j); rigid translation, periodicity.hpp:1088-1160)
  int periodic;
  Periodic pm;
};
__device__ inline void nearest_image(const MixedOut& o, const BodyD& bi, BodyD& bj) {
  if (o.periodic) bj.c = bi.c + periodic_sep(o.pm, bi.c, bj.c);
}
__device__ inline void store_contact(const MixedOut& o, size_t k, bool swapped, double sep, V3 n, V3 cpA, V3 cpB, V3 ci,
                                     V3 cj) {
  // canonical (A, B) = (lower kind, higher kind); the list's (i, j) may be (B, A)
  const V3 c1 = swapped ? cpB : cpA, c2 = swapped ? cpA : cpB;
  if (swapped) n = V3{-n.x, -n.y, -n.z};
  if (o.sep) o.sep[k] = sep;
  if (o.normal) store3(o.normal, k, n);
  if (o.cp1) store3(o.cp1, k, c1);
  if (o.cp2) store3(o.cp2, k, c2);
  if (o.ra) store3(o.ra, k, c1 - ci);
  if (o.rb) store3(o.rb, k, c2 - cj);
}

#ifndef MHIP_MIXED_FMA_TU
// the closed-form classes: sphere - sphere, sphere - rod, rod - rod, rod - ellipsoid, sphere - ellipsoid (default route)
template <int CLS, int BLOCK>
__global__ void __launch_bounds__(BLOCK)
    k_contact_class(const int32_t* __restrict__ class_start, const int32_t* __restrict__ order,
                    const int2* __restrict__ pairs, const int32_t* __restrict__ kind, const double* __restrict__ center,
                    const double* __restrict__ quat, const double* __restrict__ shape, MixedOut out) {
  const int32_t beg = class_start[CLS], end = class_start[CLS + 1];
  for (int32_t t = beg + blockIdx.x * blockDim.x + threadIdx.x; t < end; t += gridDim.x * blockDim.x) {
    const size_t k = static_cast<size_t>(order[t]);
    const int2 ij = pairs[k];
    const BodyD bi = load_body(kind, center, quat, shape, ij.x);
    BodyD bj = load_body(kind, center, quat, shape, ij.y);
    nearest_image(out, bi, bj);
    const bool swapped = bi.kind > bj.kind;
    const BodyD& A = swapped ? bj : bi;
    const BodyD& B = swapped ? bi : bj;
    if (CLS == 0) {  // sphere - sphere
      V3 d;
      const double cc = dist_point_point(A.c, B.c, d);
      const double inv = 1.0 / cc;
      store_contact(out, k, swapped, cc - A.s.x - B.s.x, d * inv, A.c, B.c, bi.c, bj.c);
    } else if (CLS == 1) {  // sphere - rod (SphereSpherocylinderLinker.cpp:210-239)
      const V3 hd = rod_half_axis(B.q, B.s.y);
      V3 closest, sepv;
      double tt;
      const double dist = dist_point_segment(A.c, B.c - hd, B.c + hd, closest, tt, sepv);
      const double radius_sum = A.s.x + B.s.x;
      const double inv = 1.0 / dist;
      store_contact(out, k, swapped, dist - radius_sum, (closest - A.c) * inv, A.c, closest, bi.c, bj.c);
    } else if (CLS == 3) {  // rod - rod
      const V3 da = rod_half_axis(A.q, A.s.y), db = rod_half_axis(B.q, B.s.y);
      const SegSeg r = dist_segment_segment(A.c - da, A.c + da, B.c - db, B.c + db);
      const double radius_sum = A.s.x + B.s.x;
      const double inv = 1.0 / r.dist;
      store_contact(out, k, swapped, r.dist - radius_sum, (r.cp2 - r.cp1) * inv, r.cp1, r.cp2, bi.c, bj.c);
    } else if (CLS == 2) {  // sphere - ellipsoid, exact (segment_ellipsoid.hpp): what a rod of zero length gets
      const segell::SegmentResult r = segell::segment_ellipsoid(A.c, A.c, EllipsoidD{B.c, B.q, B.s});
      store_contact(out, k, swapped, r.sdist - A.s.x, V3{-r.n.x, -r.n.y, -r.n.z}, r.p, r.x, bi.c, bj.c);
    } else if (CLS == 4) {  // rod - ellipsoid (segment_ellipsoid.hpp)
      const V3 hd = rod_half_axis(A.q, A.s.y);
      const segell::SegmentResult r = segell::segment_ellipsoid(A.c - hd, A.c + hd, EllipsoidD{B.c, B.q, B.s});
      store_contact(out, k, swapped, r.sdist - A.s.x, V3{-r.n.x, -r.n.y, -r.n.z}, r.p, r.x, bi.c, bj.c);
    }
  }
}

#endif  // !MHIP_MIXED_FMA_TU

// The two minimisation classes (S-E, E-E) in lockstep form (ellipsoid_lockstep.hpp): persistent wavefronts, one
// lane per pair of the class, objective evaluations converged, lanes refilled from a per-class counter.  Same
// arithmetic per lane as the nested-loop form the tests build as their checker (tests/cpp/ellipsoid_nested_ref.hip).
// Two waves per SIMD: 20 KB of LDS history per wave (ellipsoid_lockstep.hpp).
template <int CLS>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2)))
    MHIP_LOCKSTEP_KERNEL(const int32_t* __restrict__ class_start, const int32_t* __restrict__ order,
                             const int2* __restrict__ pairs, const int32_t* __restrict__ kind,
                             const double* __restrict__ center, const double* __restrict__ quat,
                             const double* __restrict__ shape, MixedOut out, unsigned long long* __restrict__ counter,
                             lockstep::StartBoard board) {
  static_assert(CLS == 2 || CLS == 5, "lockstep kernels exist for the minimisation classes only");
  const int32_t beg = class_start[CLS], end = class_start[CLS + 1];
  const size_t n = static_cast<size_t>(end - beg);
  const int lane = threadIdx.x & 63;
  __shared__ double history_tile[lockstep::kHistorySlots][64];  // one column per lane (workgroup = one wave)
  const lockstep::History hist{&history_tile[0][threadIdx.x & 63]};
  lockstep::Machine m;
  m.phase = lockstep::PH_IDLE;
  m.evals = 0;
  const int spu = lockstep::starts_per_unit(n), units = 9 / spu;  // (by the size of the CLASS, known here only)
  BodyD A{}, B{};
  lockstep::Frame frA{}, frB{};  // per-pair constants of the objective (see ellipsoid_lockstep.hpp)
  bool swapped = false;
  size_t k = 0, slot = 0;  // the pair's place in the caller's list, and in this class (its row of the start board)
  bool active = false, need = true;
  for (unsigned round = 0;; ++round) {
    const unsigned long long want = __ballot(need);
    if (want) {
      const int leader = __ffsll(static_cast<long long>(want)) - 1;
      unsigned long long base = 0;
      if (lane == leader) base = atomicAdd(counter, static_cast<unsigned long long>(__popcll(want)));
      base = __shfl(base, leader, 64);
      if (need) {
        const size_t unit = base + __popcll(want & ((1ull << lane) - 1ull));  // (pair of the class, third of its starts)
        slot = unit / units;
        need = false;
        active = slot < n;
        if (active) {
          k = static_cast<size_t>(order[beg + slot]);
          const int2 ij = pairs[k];
          const BodyD bi = load_body(kind, center, quat, shape, ij.x);
          BodyD bj = load_body(kind, center, quat, shape, ij.y);
          nearest_image(out, bi, bj);
          swapped = bi.kind > bj.kind;
          A = swapped ? bj : bi;
          B = swapped ? bi : bj;
          frB = lockstep::make_frame(B.q);
          if (CLS == 5) frA = lockstep::make_frame(A.q);
          lockstep::begin_item(m, static_cast<int>(unit % units), spu);
        }
      }
    }
    if (!__any(active)) break;
    V3 n1{0, 0, 0}, f1{0, 0, 0}, f2{0, 0, 0};
    double fv = 0.0;
    const bool evaluate = active && lockstep::wants_evaluation(m);  // parked lanes sit the round out
    if (evaluate) {
      const lbfgs::V2 tp = lockstep::query_point(m);
      double st, ct, sp, cp;
      det_sincos(tp.a, st, ct);
      det_sincos(tp.b, sp, cp);
      n1 = V3{st * cp, st * sp, ct};
      const EllipsoidD elB{B.c, B.q, B.s};
      V3 sv;
      if (CLS == 2) {         // point - ellipsoid: n1 is the ellipsoid's outward normal, f1 its foot point
        f1 = lockstep::normal_to_foot_point_framed(n1, elB, frB);
        fv = dist_point_point(f1, A.c, sv);
      } else {                // ellipsoid - ellipsoid
        f1 = lockstep::normal_to_foot_point_framed(n1, EllipsoidD{A.c, A.q, A.s}, frA);
        f2 = lockstep::normal_to_foot_point_framed(V3{-n1.x, -n1.y, -n1.z}, elB, frB);
        fv = dist_point_point(f1, f2, sv);
      }
    }
    if (evaluate && lockstep::take_value(m, fv)) {  // that was the evaluation at the best of the nine starts
      const V3 ci = swapped ? B.c : A.c, cj = swapped ? A.c : B.c;  // centres in the list's (i, j) order
      if (CLS == 2) {
        const double d = dot(A.c - f1, n1);
        store_contact(out, k, swapped, d - A.s.x, V3{-n1.x, -n1.y, -n1.z}, A.c, f1, ci, cj);
      } else {
        store_contact(out, k, swapped, dot(f2 - f1, n1), n1, f1, f2, ci, cj);
      }
      active = false;
      need = true;
    }
    lockstep::scheduled_transitions(m, hist, active, round);
    if (active && m.phase == lockstep::PH_START_DONE && !lockstep::post_start(m, board, slot)) {
      active = false;
      need = true;
    }
  }
  if (m.evals) atomicAdd(counter + 4, static_cast<unsigned long long>(m.evals));  // objective evaluations of the class
}

// the two launches (S-E, E-E); cnt[k] = next pair of class k, cnt[4 + k] = its objective evaluations (k = 1 was R-E,
// which runs in closed form since round 3: its words stay zero)
int MHIP_LOCKSTEP_LAUNCH(unsigned grid, const int32_t* start, const int32_t* order, const int2* pairs,
                         const int32_t* kind, const double* center, const double* quat, const double* shape,
                         const MixedOut& out, unsigned long long* cnt, bool sphere_ellipsoid,
                         const lockstep::StartBoard& board_se, const lockstep::StartBoard& board_ee, hipStream_t s) {
  if (sphere_ellipsoid)
    MHIP_LOCKSTEP_KERNEL<2><<<grid, 64, 0, s>>>(start, order, pairs, kind, center, quat, shape, out, cnt + 0, board_se);
  MHIP_LOCKSTEP_KERNEL<5><<<grid, 64, 0, s>>>(start, order, pairs, kind, center, quat, shape, out, cnt + 2, board_ee);
  MHIP_LAUNCH_CHECK();
  return MHIP_SUCCESS;
}

#ifndef MHIP_MIXED_FMA_TU
// (mixed_fma.hip)
int launch_contact_classes_lockstep_fma(unsigned grid, const int32_t* start, const int32_t* order, const int2* pairs,
                                        const int32_t* kind, const double* center, const double* quat,
                                        const double* shape, const MixedOut& out, unsigned long long* cnt,
                                        bool sphere_ellipsoid, const lockstep::StartBoard& board_se,
                                        const lockstep::StartBoard& board_ee, hipStream_t s);
std::atomic<int> g_mixed_contraction{0};
std::atomic<int> g_mixed_se_route{0};  // 0: S-E in closed form; 1: through the reference's point - ellipsoid minimiser

struct MixedScratch {
  DeviceBuffer cls, flags, pos, order, start, scanws, counters, board;
  int32_t* host = nullptr;
};
MixedScratch& mixed_scratch() {
  thread_local MixedScratch s;
  return s;
}

#endif  // !MHIP_MIXED_FMA_TU

}  // namespace mhip

#ifndef MHIP_MIXED_FMA_TU
using namespace mhip;

extern "C" {

int mhip_compute_aabb_mixed(size_t n, const int32_t* kind, const double* center, const double* quat,
                            const double* shape, double* aabb, double* bounding_radius, mhip_stream_t stream) {
  if (n == 0) return MHIP_SUCCESS;
  MHIP_REQUIRE(kind && center && quat && shape && aabb, MHIP_ERR_INVALID_ARGUMENT, "null argument");
  k_aabb_mixed<false><<<grid_for(n), kBlock, 0, as_stream(stream)>>>(n, kind, center, quat, shape, aabb,
                                                                      bounding_radius);
  MHIP_LAUNCH_CHECK();
  return MHIP_SUCCESS;
}

int mhip_compute_aabb_mixed_conservative(size_t n, const int32_t* kind, const double* center, const double* quat,
                                         const double* shape, double* aabb, double* bounding_radius,
                                         mhip_stream_t stream) {
  if (n == 0) return MHIP_SUCCESS;
  MHIP_REQUIRE(kind && center && quat && shape && aabb, MHIP_ERR_INVALID_ARGUMENT, "null argument");
  k_aabb_mixed<true><<<grid_for(n), kBlock, 0, as_stream(stream)>>>(n, kind, center, quat, shape, aabb,
                                                                     bounding_radius);
  MHIP_LAUNCH_CHECK();
  return MHIP_SUCCESS;
}

static int contact_mixed_impl(size_t c, const int32_t* pairs, const int32_t* kind, const double* center,
                              const double* quat, const double* shape, const double* box, double* sep, double* normal,
                              double* cp1, double* cp2, double* ra, double* rb, size_t* class_counts,
                              mhip_stream_t stream);

int mhip_contact_mixed(size_t c, const int32_t* pairs, const int32_t* kind, const double* center, const double* quat,
                       const double* shape, double* sep, double* normal, double* cp1, double* cp2, double* ra,
                       double* rb, size_t* class_counts, mhip_stream_t stream) {
  return contact_mixed_impl(c, pairs, kind, center, quat, shape, nullptr, sep, normal, cp1, cp2, ra, rb, class_counts,
                            stream);
}

int mhip_contact_mixed_periodic(size_t c, const int32_t* pairs, const int32_t* kind, const double* center,
                                const double* quat, const double* shape, const double* box, double* sep, double* normal,
                                double* cp1, double* cp2, double* ra, double* rb, size_t* class_counts,
                                mhip_stream_t stream) {
  MHIP_REQUIRE(box != nullptr && box[0] > 0 && box[1] > 0 && box[2] > 0, MHIP_ERR_INVALID_ARGUMENT,
               "periodic box must be positive");
  return contact_mixed_impl(c, pairs, kind, center, quat, shape, box, sep, normal, cp1, cp2, ra, rb, class_counts,
                            stream);
}

static int contact_mixed_impl(size_t c, const int32_t* pairs, const int32_t* kind, const double* center,
                              const double* quat, const double* shape, const double* box, double* sep, double* normal,
                              double* cp1, double* cp2, double* ra, double* rb, size_t* class_counts,
                              mhip_stream_t stream) {
  TraceRange trace_range("contact_mixed");
  if (class_counts)
    for (int k = 0; k < 6; ++k) class_counts[k] = 0;
  if (c == 0) return MHIP_SUCCESS;
  MHIP_REQUIRE(pairs && kind && center && quat && shape, MHIP_ERR_INVALID_ARGUMENT, "null argument");
  MHIP_REQUIRE(c < (1u << 31), MHIP_ERR_RUNTIME, "too many pairs");
  hipStream_t s = as_stream(stream);
  MixedScratch& ms = mixed_scratch();
  for (DeviceBuffer* b : {&ms.cls, &ms.flags, &ms.pos, &ms.order})
    if (int e = b->reserve((c + 2) * sizeof(int32_t))) return e;
  if (int e = ms.start.reserve(16 * sizeof(int32_t))) return e;
  if (int e = ms.scanws.reserve(scan_workspace_bytes(c + 2) + 64)) return e;
  if (!ms.host) MHIP_HIP(hipHostMalloc(reinterpret_cast<void**>(&ms.host), 64));
  const int2* p2 = reinterpret_cast<const int2*>(pairs);
  int32_t* start = ms.start.as<int32_t>();
  MHIP_HIP(hipMemsetAsync(start, 0, 16 * sizeof(int32_t), s));
  const unsigned g = grid_for(c);
  k_pair_class<<<g, kBlock, 0, s>>>(c, p2, kind, ms.cls.as<int32_t>());
  MHIP_LAUNCH_CHECK();
  for (int k = 0; k < 6; ++k) {
    k_class_flags<<<g, kBlock, 0, s>>>(c, ms.cls.as<int32_t>(), k, ms.flags.as<int32_t>());
    MHIP_LAUNCH_CHECK();
    if (int e = exclusive_scan_i32(ms.flags.as<int32_t>(), ms.pos.as<int32_t>(), c, ms.scanws.ptr, s)) return e;
    k_class_scatter<<<g, kBlock, 0, s>>>(c, ms.flags.as<int32_t>(), ms.pos.as<int32_t>(), k, start,
                                         ms.order.as<int32_t>());
    MHIP_LAUNCH_CHECK();
  }
  const double unit_box[3] = {1, 1, 1};
  const MixedOut out{sep, normal, cp1, cp2, ra, rb, box ? 1 : 0, make_periodic(box ? box : unit_box)};
  const int32_t* order = ms.order.as<int32_t>();
#define CLASS(K, BLK, GRID) \
  k_contact_class<K, BLK><<<GRID, BLK, 0, s>>>(start, order, p2, kind, center, quat, shape, out)
  const unsigned gs = g;  // closed-form classes: grid-stride over <= c
  CLASS(0, kBlock, gs);
  CLASS(1, kBlock, gs);
  CLASS(3, kBlock, gs);
  CLASS(4, kBlock, gs);
  const bool se_minimiser = g_mixed_se_route.load() != 0;
  if (!se_minimiser) CLASS(2, kBlock, gs);
  {
    if (int e = ms.counters.reserve(64)) return e;
    unsigned long long* cnt = ms.counters.as<unsigned long long>();  // [k]: next pair of class k; [4 + k]: its evaluations
    MHIP_HIP(hipMemsetAsync(cnt, 0, 8 * sizeof(unsigned long long), s));
    const unsigned gl = static_cast<unsigned>(9 * c / 64 + 1 > 2048 ? 2048 : 9 * c / 64 + 1);  // persistent waves (up to nine units per pair)
    // start boards of the minimisation classes (ellipsoid_lockstep.hpp), one row per pair of the class.  The class
    // sizes stay on the device, so each board is sized for every pair of the list (the S-E one only when that class
    // is routed through the minimiser)
    const size_t boards = se_minimiser ? 2 : 1;
    const size_t rec_doubles = lockstep::board_doubles(c);
    if (int e = ms.board.reserve(boards * (rec_doubles * sizeof(double) + c * sizeof(unsigned) + 64))) return e;
    double* rec = ms.board.as<double>();
    unsigned* arr = reinterpret_cast<unsigned*>(rec + boards * rec_doubles);
    MHIP_HIP(hipMemsetAsync(arr, 0, boards * c * sizeof(unsigned), s));
    const lockstep::StartBoard board_ee{rec, arr};
    const lockstep::StartBoard board_se{rec + (boards - 1) * rec_doubles, arr + (boards - 1) * c};
    if (g_mixed_contraction.load() != 0) {
      if (int e = launch_contact_classes_lockstep_fma(gl, start, order, p2, kind, center, quat, shape, out, cnt, se_minimiser, board_se, board_ee, s)) return e;
    } else {
      if (int e = launch_contact_classes_lockstep(gl, start, order, p2, kind, center, quat, shape, out, cnt, se_minimiser, board_se, board_ee, s)) return e;
    }
  }
#undef CLASS
  MHIP_LAUNCH_CHECK();
  if (class_counts) {
    MHIP_HIP(hipMemcpyAsync(ms.host, start, 8 * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    MHIP_HIP(hipStreamSynchronize(s));
    for (int k = 0; k < 6; ++k) class_counts[k] = static_cast<size_t>(ms.host[k + 1] - ms.host[k]);
  }
  return MHIP_SUCCESS;
}

int mhip_contact_mixed_set_sphere_ellipsoid_route(int route) {
  MHIP_REQUIRE(route == 0 || route == 1, MHIP_ERR_INVALID_ARGUMENT, "route must be 0 (closed form) or 1 (the reference's minimiser), got %d", route);
  g_mixed_se_route.store(route);
  return MHIP_SUCCESS;
}

int mhip_contact_mixed_set_contraction(int on) {
  MHIP_REQUIRE(on == 0 || on == 1, MHIP_ERR_INVALID_ARGUMENT, "contraction must be 0 or 1, got %d", on);
  g_mixed_contraction.store(on);
  return MHIP_SUCCESS;
}

/* objective evaluations of the S-E, R-E and E-E classes in the last mhip_contact_mixed* call on this host thread
 * (synchronises the stream); R-E is evaluated in closed form and reports 0 */
int mhip_contact_mixed_last_evaluations(unsigned long long evaluations[3], mhip_stream_t stream) {
  MHIP_REQUIRE(evaluations != nullptr, MHIP_ERR_INVALID_ARGUMENT, "evaluations is null");
  MixedScratch& ms = mixed_scratch();
  evaluations[0] = evaluations[1] = evaluations[2] = 0;
  if (!ms.counters.ptr) return MHIP_SUCCESS;
  unsigned long long host[8];
  MHIP_HIP(hipMemcpyAsync(host, ms.counters.ptr, sizeof(host), hipMemcpyDeviceToHost, as_stream(stream)));
  MHIP_HIP(hipStreamSynchronize(as_stream(stream)));
  for (int k = 0; k < 3; ++k) evaluations[k] = host[4 + k];
  return MHIP_SUCCESS;
}

}  // extern "C"
#endif  // !MHIP_MIXED_FMA_TU
