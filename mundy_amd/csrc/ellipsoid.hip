// ellipsoid.hip -- ellipsoid narrow phase: batch and neighbour-list entry points.
// One lane per pair; the multistart L-BFGS runs as a per-lane state machine so that the objective evaluations of a
// wavefront stay converged and lanes refill from a global counter (ellipsoid_lockstep.hpp).  Compute bound (fp64
// vector + transcendental), priced against the fp64 vector peak, not HBM.
#include "ellipsoid_lockstep.hpp"

namespace mhip {

constexpr int kEllBlock = 64;
// Resources of the lockstep kernel: ~220 VGPRs, no scratch, 20 KB of LDS per wave for the L-BFGS history ring -> two
// waves per SIMD (eight per CU).  History of the form (10^6 pairs, MI355X): one wave per SIMD with a 30 KB ring 1.2 * 10^7
// pairs/s; the history in registers under compile-time indices 9.9 * 10^6 at one wave per SIMD, 8.9 at two (236 B of
// spills), 3.3 at three; the plain nested-loop form (now the tests' checker) 4.6 / 5.6 / 5.3 / 4.5 * 10^6 at 1 - 4 waves.
#define ELL_OCC __attribute__((amdgpu_waves_per_eu(2)))

__device__ inline EllipsoidD load_ellipsoid(const double* c, const double* q, const double* r, size_t i) {
  return {load3(c, i), load4q(q, i), load3(r, i)};
}

// ---- lockstep form (ellipsoid_lockstep.hpp): persistent wavefronts, one lane per pair, objective evaluations converged
struct EEInput {  // where a lane finds pair k
  const int2* pairs;                            // neighbour-list form: both ellipsoids come from one body table ...
  const double *c1, *q1, *r1, *c2, *q2, *r2;    // ... or element-wise over two tables (pairs == nullptr)
  const double* point;                          // POINT problems: point k against ellipsoid k of (c1, q1, r1)
};
struct EEOutput {
  double *dist, *cp1, *cp2, *n1, *n2, *ra, *rb;  // any may be null
};
// POINT = false: ellipsoid-ellipsoid (EllipsoidEllipsoid.hpp:106-151); true: point-ellipsoid (PointEllipsoid.hpp:94-135)
template <bool POINT>
__global__ void __launch_bounds__(kEllBlock) ELL_OCC
    k_ellipsoid_pairs_lockstep(size_t n, EEInput in, EEOutput out, unsigned long long* __restrict__ counter,
                               lockstep::StartBoard board) {
  const int lane = threadIdx.x & 63;
  __shared__ double history_tile[lockstep::kHistorySlots][64];  // one column per lane (workgroup = one wave)
  const lockstep::History hist{&history_tile[0][threadIdx.x & 63]};
  lockstep::Machine m;
  m.phase = lockstep::PH_IDLE;
  m.evals = 0;
  const int spu = lockstep::starts_per_unit(n), units = 9 / spu;
  EllipsoidD e1{}, e2{};
  lockstep::Frame fr1{}, fr2{};  // per-pair constants of the foot-point maps
  size_t k = 0;
  bool active = false, need = true;
  for (unsigned round = 0;; ++round) {
    // lanes without work take the next units -- (pair, third of its starts) -- from the global counter (one atomic per wave)
    const unsigned long long want = __ballot(need);
    if (want) {
      const int leader = __ffsll(static_cast<long long>(want)) - 1;
      unsigned long long base = 0;
      if (lane == leader) base = atomicAdd(counter, static_cast<unsigned long long>(__popcll(want)));
      base = __shfl(base, leader, 64);
      if (need) {
        const size_t unit = base + __popcll(want & ((1ull << lane) - 1ull));
        k = unit / units;
        need = false;
        active = k < n;
        if (active) {
          if (POINT) {
            e1 = load_ellipsoid(in.c1, in.q1, in.r1, k);
            e2.c = load3(in.point, k);  // the point rides in e2's centre
          } else if (in.pairs) {
            const int2 ij = in.pairs[k];
            e1 = load_ellipsoid(in.c1, in.q1, in.r1, ij.x);
            e2 = load_ellipsoid(in.c1, in.q1, in.r1, ij.y);
          } else {
            e1 = load_ellipsoid(in.c1, in.q1, in.r1, k);
            e2 = load_ellipsoid(in.c2, in.q2, in.r2, k);
          }
          fr1 = lockstep::make_frame(e1.q);
          if (!POINT) fr2 = lockstep::make_frame(e2.q);
          lockstep::begin_item(m, static_cast<int>(unit % units), spu);
        }
      }
    }
    if (!__any(active)) break;  // every lane of the wave has run out of pairs
    // the objective for all lanes at once (EllipsoidEllipsoid.hpp:118-129): the converged part of the loop
    V3 n1{0, 0, 0}, f1{0, 0, 0}, f2{0, 0, 0};
    double fv = 0.0;
    const bool evaluate = active && lockstep::wants_evaluation(m);  // parked lanes sit the round out
    if (evaluate) {
      const lbfgs::V2 tp = lockstep::query_point(m);
      double st, ct, sp, cp;
      det_sincos(tp.a, st, ct);
      det_sincos(tp.b, sp, cp);
      n1 = V3{st * cp, st * sp, ct};
      f1 = lockstep::normal_to_foot_point_framed(n1, e1, fr1);
      f2 = POINT ? e2.c : lockstep::normal_to_foot_point_framed(V3{-n1.x, -n1.y, -n1.z}, e2, fr2);
      V3 sep;
      fv = dist_point_point(f1, f2, sep);
    }
    // each lane's minimiser takes its value (bookkeeping); the diverging logic runs on the wave's schedule below
    if (evaluate && lockstep::take_value(m, fv)) {  // that was the evaluation at the best of the nine starts
      if (out.dist) out.dist[k] = dot(f2 - f1, n1);  // POINT: dot(point - closest, normal)
      if (out.n1) store3(out.n1, k, n1);
      if (out.n2) store3(out.n2, k, V3{-n1.x, -n1.y, -n1.z});
      if (out.cp1) store3(out.cp1, k, f1);
      if (out.cp2) store3(out.cp2, k, f2);
      if (out.ra) store3(out.ra, k, f1 - e1.c);
      if (out.rb) store3(out.rb, k, f2 - e2.c);
      active = false;
      need = true;
    }
    lockstep::scheduled_transitions(m, hist, active, round);
    // a start has returned: on the pair's board with it; the ninth arrival goes on to the final evaluation
    if (active && m.phase == lockstep::PH_START_DONE && !lockstep::post_start(m, board, k)) {
      active = false;
      need = true;
    }
  }
  if (m.evals) atomicAdd(counter + 1, static_cast<unsigned long long>(m.evals));  // objective evaluations, for the fp64 roofline
}

struct EllipsoidScratch {
  DeviceBuffer counter, board;
};
EllipsoidScratch& ellipsoid_scratch() {
  thread_local EllipsoidScratch s;
  return s;
}
// persistent grid: every CU gets its waves, each lane keeps pulling pairs until the counter passes n
int launch_ellipsoid_lockstep(size_t n, const EEInput& in, const EEOutput& out, hipStream_t s) {
  EllipsoidScratch& es = ellipsoid_scratch();
  if (int e = es.counter.reserve(64)) return e;
  MHIP_HIP(hipMemsetAsync(es.counter.ptr, 0, 2 * sizeof(unsigned long long), s));
  // the start board: the records of every pair, then the arrival counters (zeroed)
  const size_t rec_doubles = lockstep::board_doubles(n);
  if (int e = es.board.reserve(rec_doubles * sizeof(double) + n * sizeof(unsigned) + 64)) return e;
  const lockstep::StartBoard board{es.board.as<double>(), reinterpret_cast<unsigned*>(es.board.as<double>() + rec_doubles)};
  MHIP_HIP(hipMemsetAsync(board.arrived, 0, n * sizeof(unsigned), s));
  const size_t waves = ((9 / lockstep::starts_per_unit(n)) * n + 63) / 64;
#ifndef MHIP_ELL_GRID
#define MHIP_ELL_GRID 2048   // 256 CUs x 4 SIMDs x 2 waves: all resident (A/B: 1024 = one wave per SIMD)
#endif
  const unsigned grid = static_cast<unsigned>(waves < MHIP_ELL_GRID ? waves : MHIP_ELL_GRID);
  if (in.point)
    k_ellipsoid_pairs_lockstep<true><<<grid, kEllBlock, 0, s>>>(n, in, out, es.counter.as<unsigned long long>(), board);
  else
    k_ellipsoid_pairs_lockstep<false><<<grid, kEllBlock, 0, s>>>(n, in, out, es.counter.as<unsigned long long>(), board);
  MHIP_LAUNCH_CHECK();
  return MHIP_SUCCESS;
}

}  // namespace mhip

using namespace mhip;

extern "C" {

int mhip_distance_ellipsoid_ellipsoid(size_t n, const double* c1, const double* q1, const double* r1,
                                      const double* c2, const double* q2, const double* r2, double* dist, double* cp1,
                                      double* cp2, double* n1, double* n2, mhip_stream_t stream) {
  if (n == 0) return MHIP_SUCCESS;
  MHIP_REQUIRE(c1 && q1 && r1 && c2 && q2 && r2, MHIP_ERR_INVALID_ARGUMENT, "ellipsoid arrays must not be null");
  return launch_ellipsoid_lockstep(n, EEInput{nullptr, c1, q1, r1, c2, q2, r2, nullptr},
                                   EEOutput{dist, cp1, cp2, n1, n2, nullptr, nullptr}, as_stream(stream));
}

int mhip_distance_point_ellipsoid(size_t n, const double* p, const double* c, const double* q, const double* r,
                                  double* dist, double* cp, double* normal, mhip_stream_t stream) {
  if (n == 0) return MHIP_SUCCESS;
  MHIP_REQUIRE(p && c && q && r, MHIP_ERR_INVALID_ARGUMENT, "point / ellipsoid arrays must not be null");
  // closest point = the ellipsoid's foot point (cp1 slot), normal = its outward normal there (n1 slot)
  return launch_ellipsoid_lockstep(n, EEInput{nullptr, c, q, r, nullptr, nullptr, nullptr, p},
                                   EEOutput{dist, cp, nullptr, normal, nullptr, nullptr, nullptr}, as_stream(stream));
}

int mhip_contact_ellipsoids(size_t c, const int32_t* pairs, const double* center, const double* quat,
                            const double* radii, double* sep, double* normal, double* cp1, double* cp2, double* ra,
                            double* rb, mhip_stream_t stream) {
  TraceRange trace_range("distance(Ellipsoid, Ellipsoid)");
  if (c == 0) return MHIP_SUCCESS;
  MHIP_REQUIRE(pairs && center && quat && radii, MHIP_ERR_INVALID_ARGUMENT, "pairs / ellipsoid arrays must not be null");
  return launch_ellipsoid_lockstep(c, EEInput{reinterpret_cast<const int2*>(pairs), center, quat, radii, nullptr,
                                              nullptr, nullptr},
                                   EEOutput{sep, cp1, cp2, normal, nullptr, ra, rb}, as_stream(stream));
}

/* objective evaluations of the last ellipsoid distance call on this host thread (synchronises the stream): with ~2.3 *
 * 10^3 fp64 instructions per evaluation this prices the kernel against the fp64 vector peak */
int mhip_ellipsoid_last_evaluations(unsigned long long* evaluations, mhip_stream_t stream) {
  MHIP_REQUIRE(evaluations != nullptr, MHIP_ERR_INVALID_ARGUMENT, "evaluations is null");
  EllipsoidScratch& es = ellipsoid_scratch();
  *evaluations = 0;
  if (!es.counter.ptr) return MHIP_SUCCESS;
  unsigned long long host[2] = {0, 0};
  MHIP_HIP(hipMemcpyAsync(host, es.counter.ptr, sizeof(host), hipMemcpyDeviceToHost, as_stream(stream)));
  MHIP_HIP(hipStreamSynchronize(as_stream(stream)));
  *evaluations = host[1];
  return MHIP_SUCCESS;
}

}  // extern "C"
