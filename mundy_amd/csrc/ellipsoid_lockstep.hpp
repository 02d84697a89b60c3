// ellipsoid_lockstep.hpp -- the multistart L-BFGS of the ellipsoid distances (mundy_math/impl/minimize_impl.hpp:151-605
// as called by mundy_geom/distance/EllipsoidEllipsoid.hpp:106-151) run in lockstep across a wavefront.
//
// One lane still owns one pair, and performs exactly the arithmetic of the reference's find_min / line_search in exactly
// their order -- but the control flow is turned inside out.  Each lane carries the minimiser as an explicit state machine
// whose only externally visible act is "evaluate the objective at this point".  The wave's loop is then
//     evaluate the objective for all lanes at once  ->  every lane feeds the value to its own machine
// so the expensive part (two sincos, four quaternion rotations, two foot-point maps: ~850 instructions) runs with all
// lanes converged whatever line-search branch, L-BFGS iteration, start point or pair each lane is in; only the
// minimiser logic diverges, and the wave runs its heavy pieces on a schedule rather than whenever a lane gets there
// ("Scheduling" below).  The unit of work a lane takes from the global counter is a THIRD of a pair -- three of its
// nine starts (round 3; it was the whole pair): the nine minimisations of a pair are independent until the best of
// them is chosen, and a launch rarely has more than a few pairs per lane of the machine (10^6 mixed bodies:
// 2.5 * 10^5 E-E pairs on 1.3 * 10^5 lanes) -- with whole pairs the launch lasted as long as its unluckiest lane's two
// or three pairs, and an eighth of the system (one rank of eight) would have lasted exactly as long as the whole.
// Each unit leaves its best (cost, point) on a per-pair board; the lane whose unit is the last to finish picks the
// best in the reference's order (first of equals) and runs the final evaluation ("Start board" below).  Single
// starts as units are finer still, but the hand-over -- stores, drain, ticket -- then sits in every second round of a
// wave: they pay only where the launch has less than a pair per two lanes.  E-E pairs, ms per launch, units of 9 / 3 / 1
// starts (profiles/r03_lockstep_units.txt): 3 * 10^4 pairs 7.6 / 3.8 / 2.7; 10^5 10.2 / 6.1 / 6.9; 2.5 * 10^5
// 15.9 / 11.4 / 16.0; 10^6 42.1 / 38.9 / 60.9 -- so the unit is one start up to kSingleStartBelow pairs, three beyond.  (One thread per pair with the minimiser called
// as ordinary nested loops spends most of its time with lanes masked off: pairs need 577...1640 objective
// evaluations, and lanes are in different loops at any moment.)
//
// Results are bit-identical to the nested-loop form: the same evaluations at the same points in the same order per
// lane (the tests build that form as their own checker, tests/cpp/ellipsoid_nested_ref.hip).
#pragma once
#include "ellipsoid_device.hpp"

namespace mhip {
namespace lockstep {

using lbfgs::kEps;
using lbfgs::M;
using lbfgs::V2;

enum Phase : int {
  PH_COST_INIT = 0,  // f(x) at the start point
  PH_G0P, PH_G0M, PH_G1P, PH_G1M,  // central differences: x.a +/- eps, x.b +/- eps
  PH_COST_AFTER,     // f(x) after a line search
  PH_LS1_VAL, PH_LS1_DP, PH_LS1_DM,  // bracketing loop of the line search: phi(a), phi(a + eps), phi(a - eps)
  PH_LS2_VAL, PH_LS2_DP, PH_LS2_DM,  // sectioning loop
  PH_FINAL,          // re-evaluation at the best point of the nine starts (outputs)
  // phases without an evaluation: the lane has the values its next decision needs and waits for the wave's next pass
  // over that piece of minimiser logic (see "Scheduling" below)
  PH_WAIT_LS1, PH_WAIT_LS2,  // the tests at the bottom of the bracketing / sectioning loop
  PH_WAIT_HEAD,              // head of find_min's loop: stop tests, L-BFGS direction, line-search set-up
  PH_START_DONE,             // this start's find_min has returned (cost, x): to be posted on the pair's board
  PH_IDLE
};

// Scheduling.  The minimiser logic between two evaluations is cheap per lane but expensive per wave: a wave executes a
// piece of it whenever ANY of its lanes is there, and with 64 lanes in 64 different places every piece is needed every
// round -- the L-BFGS two-loop recursion (~600 instructions) and the line-search tests (~300) against ~850 for the
// objective itself.  So the two heavy pieces are not run every round: a lane that reaches one parks in a PH_WAIT phase
// (it takes no evaluation meanwhile) and the wave runs the line-search tests every kLsEvery-th round and the loop head
// every kHeadEvery-th, for all lanes parked there at once -- or at once when no lane of the wave could evaluate
// otherwise.  A parked lane costs 1/64 of an evaluation round per round it waits; the piece it waits for costs the
// whole wave.  Nothing a lane computes changes: same evaluations, same points, same order.
// kLsEvery = 3 is the length of a line-search trial (phi(a), phi(a + eps), phi(a - eps)): lanes fall into step with
// the schedule, a lane that needs another trial is back at the tests exactly when they run next, and a lane that
// leaves the line search takes its five gradient / cost evaluations and parks for ONE round before a head pass.
// Measured (MI355X, 4 * 10^5 E-E pairs; mixed 3 * 10^5-body narrow phase): every round 1.57 * 10^7 pairs/s, 56.6 ms;
// (2, 4) 1.66, 50.3; (3, 3) 1.94, 43.5; (3, 6) 1.90, 42.7; (4, 4) 1.64, 51.2; (6, 6) 1.60, 50.7; (8, 8) 1.56, 51.2
// (scripts/ab_lockstep_schedule.sh).  Holding the head pass back until 6 ... 28 lanes wait for it only loses
// (1.84 ... 1.53): a parked lane also falls out of step.
#ifndef MHIP_LOCKSTEP_LS_EVERY
#define MHIP_LOCKSTEP_LS_EVERY 3
#endif
#ifndef MHIP_LOCKSTEP_HEAD_EVERY
#define MHIP_LOCKSTEP_HEAD_EVERY 3
#endif
constexpr unsigned kLsEvery = MHIP_LOCKSTEP_LS_EVERY, kHeadEvery = MHIP_LOCKSTEP_HEAD_EVERY;

struct Machine {
  // find_min (minimize_impl.hpp:407-605 with the defaults of minimize.hpp:42-51)
  V2 x, g, prev_x, prev_g, dir;
  double cost, prev_val;
  bool been_used, stop_used, after_ls;
  int current_size, ring_head;  // the L-BFGS history itself lives in LDS (History below), as a ring buffer
  unsigned evals;               // objective evaluations of this pair (profiling: the fp64 roofline of the class)
  // line_search (minimize_impl.hpp:233-405)
  double f0, d0, mu, alpha, last_alpha, last_val, last_val_der, a, b, a_val, b_val, a_val_der, b_val_der, thresh;
  double ls_first, ls_last, val, fp, fm;
  int itr;
  // multistart (EllipsoidEllipsoid.hpp:106-151)
  int start, starts_per_unit;
  double best;
  V2 best_tp;
  int phase;
};

// The lane's column of the workgroup's history tile: M entries of (s.a, s.b, y.a, y.b), laid out [slot][lane] so a
// wave's access to one slot is conflict-free.  Kept in LDS rather than registers because it is indexed at run time (a
// ring buffer): with compile-time indices under predicates the bookkeeping cost ~1000 instructions per round, executed
// by the wave whenever ANY lane was in that phase.  20 KB per wave: EIGHT waves share a CU's 160 KB, two per SIMD.
// What the reference also stores per entry is derived instead: rho_i = 1 / (s_i . y_i) is recomputed from the stored
// s_i, y_i (the same division of the same operands: the same bits), and the alpha_i of the two-loop recursion live in
// registers under compile-time indices (only ten scalars, written in one unrolled loop and read in the next).
constexpr int kHistorySlots = 4 * M;
struct History {
  double* col;  // &tile[0][lane]
  __device__ double& sa(int i) const { return col[(0 * M + i) * 64]; }
  __device__ double& sb(int i) const { return col[(1 * M + i) * 64]; }
  __device__ double& ya(int i) const { return col[(2 * M + i) * 64]; }
  __device__ double& yb(int i) const { return col[(3 * M + i) * 64]; }
};

// Per-pair constants of the objective, hoisted out of the ~900 evaluations a pair needs.  qrot(q, v) computes
// 1 / |q|^2 and the inverse quaternion on every call; both depend on the body only.  The values below are produced by
// the very expressions qrot uses, so the framed foot-point map returns the same bits as normal_to_foot_point.
struct Frame {
  Quat q, q_inv;    // lab <- body:  (q (0, v)) q_inv
  Quat qc, qc_inv;  // body <- lab:  (conj(q) (0, v)) conj(q)_inv
};
__device__ inline Frame make_frame(const Quat& q) {
  const double inv_n2 = 1.0 / (q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
  const Quat qc{q.w, -q.x, -q.y, -q.z};
  const double inv_n2c = 1.0 / (qc.w * qc.w + qc.x * qc.x + qc.y * qc.y + qc.z * qc.z);
  return Frame{q, Quat{q.w * inv_n2, -q.x * inv_n2, -q.y * inv_n2, -q.z * inv_n2}, qc,
               Quat{qc.w * inv_n2c, -qc.x * inv_n2c, -qc.y * inv_n2c, -qc.z * inv_n2c}};
}
__device__ inline V3 qrot_framed(const Quat& q, const Quat& q_inv, V3 v) {
  const Quat vq{0.0, v.x, v.y, v.z};
  const Quat r = qmul(qmul(q, vq), q_inv);
  return {r.x, r.y, r.z};
}
// map_surface_normal_to_foot_point_on_ellipsoid (Ellipsoid.hpp:462-468) with the frame precomputed
__device__ inline V3 normal_to_foot_point_framed(V3 lab_n, const EllipsoidD& el, const Frame& f) {
  const V3 body_n = qrot_framed(f.qc, f.qc_inv, lab_n);
  return qrot_framed(f.q, f.q_inv, body_normal_to_foot(body_n, el)) + el.c;
}

constexpr double kMinDelta = 1e-7, kDerivEps = 1e-7, kMinCost = 1e-8 /* get_relaxed_zero_tolerance<double>() */;
constexpr double kRho = 0.01, kSigma = 0.9;
constexpr int kLsMaxIter = 100;

__device__ inline V2 start_point(int s) {
  const double pi = 3.141592653589793;
  const double theta_guesses[3] = {0.0, 0.5 * pi, pi};
  const double phi_guesses[3] = {pi / 3.0, pi, 5.0 * (pi / 3.0)};
  return V2{theta_guesses[s / 3], phi_guesses[s % 3]};
}

__device__ inline void begin_start(Machine& m) {
  m.x = start_point(m.start);
  m.current_size = 0;
  m.ring_head = 0;
  m.been_used = false;
  m.stop_used = false;
  m.after_ls = false;
  m.prev_x = V2{0, 0};
  m.prev_g = V2{0, 0};
  m.prev_val = 0;
  m.phase = PH_COST_INIT;
}
// a lane's next unit of work: starts [unit * spu, (unit + 1) * spu) of some pair, spu = 1, 3 or 9 starts per unit
// (m.evals keeps counting over the lane's units)
#ifndef MHIP_LOCKSTEP_STARTS_PER_UNIT
#define MHIP_LOCKSTEP_STARTS_PER_UNIT 0  // 0: by the size of the launch (starts_per_unit below); 1 / 3 / 9 force it (A/B)
#endif
constexpr size_t kSingleStartBelow = 60000;
__host__ __device__ inline int starts_per_unit(size_t pairs) {
  if (MHIP_LOCKSTEP_STARTS_PER_UNIT) return MHIP_LOCKSTEP_STARTS_PER_UNIT;
  return pairs <= kSingleStartBelow ? 1 : 3;
}
__device__ inline void begin_item(Machine& m, int unit, int spu) {
  m.starts_per_unit = spu;
  m.start = unit * spu;
  m.best = __builtin_huge_val();
  m.best_tp = V2{0.0, 0.0};
  begin_start(m);
}

// the point the lane wants evaluated in its current phase
__device__ inline V2 query_point(const Machine& m) {
  const double eps = kDerivEps;
  switch (m.phase) {
    case PH_COST_INIT:
    case PH_COST_AFTER: return m.x;
    case PH_G0P: return V2{m.x.a + eps, m.x.b};
    case PH_G0M: return V2{m.x.a - eps, m.x.b};
    case PH_G1P: return V2{m.x.a, m.x.b + eps};
    case PH_G1M: return V2{m.x.a, m.x.b - eps};
    case PH_LS1_VAL:
    case PH_LS2_VAL: return V2{m.x.a + m.alpha * m.dir.a, m.x.b + m.alpha * m.dir.b};
    case PH_LS1_DP:
    case PH_LS2_DP: return V2{m.x.a + (m.alpha + eps) * m.dir.a, m.x.b + (m.alpha + eps) * m.dir.b};
    case PH_LS1_DM:
    case PH_LS2_DM: return V2{m.x.a + (m.alpha - eps) * m.dir.a, m.x.b + (m.alpha - eps) * m.dir.b};
    case PH_FINAL: return m.best_tp;
    default: return V2{0.0, 0.0};
  }
}

// the line search returned `alpha`: take the step, then gradient and cost at the new point
__device__ inline void line_search_done(Machine& m, double alpha) {
  m.x = V2{alpha * m.dir.a + m.x.a, alpha * m.dir.b + m.x.b};
  m.after_ls = true;
  m.phase = PH_G0P;
}

// one start is finished (find_min's return value is m.cost, its point m.x): kept if it is the best of the unit so far
// -- `if (d < global_dist)`, EllipsoidEllipsoid.hpp:141-145 -- then the unit's next start, or the board (post_start)
__device__ inline void finish_start(Machine& m) {
  if (m.cost < m.best) {
    m.best = m.cost;
    m.best_tp = m.x;
  }
  ++m.start;
  if (m.start % m.starts_per_unit != 0)
    begin_start(m);
  else
    m.phase = PH_START_DONE;
}

// Start board.  Per pair of a launch: one (cost, theta, phi) record per unit and an arrival counter (zeroed by the host
// before the launch).  A lane posts its unit's best with write-through stores, drains them, takes a ticket; the last
// arrival reads the records past its L1 and chooses as the reference's loop over the starts does -- `if (d <
// global_dist)` in start order, within a unit and between units: the first of equal costs
// (EllipsoidEllipsoid.hpp:131-147).  Same hand-off as k_fold_finalize
// (convex.hip): agent-scope relaxed atomics, no fence that writes back or invalidates an L2.
struct StartBoard {
  double* records;   // [pairs][9 / starts per unit][3]
  unsigned* arrived; // [pairs]
};
// doubles of the records of a launch of up to `pairs` pairs whatever unit the kernel picks for the pairs it finds
__host__ __device__ inline size_t board_doubles(size_t pairs) {
  if (MHIP_LOCKSTEP_STARTS_PER_UNIT) return pairs * 3 * (9 / MHIP_LOCKSTEP_STARTS_PER_UNIT);
  const size_t small = pairs < kSingleStartBelow ? pairs : kSingleStartBelow;
  return small * 27 > pairs * 9 ? small * 27 : pairs * 9;
}
// returns true when this lane's unit was the last of its pair: the machine then holds the best point and wants the
// final evaluation (PH_FINAL); false: the lane is free for its next unit
__device__ inline bool post_start(Machine& m, const StartBoard& board, size_t pair) {
  const int units = 9 / m.starts_per_unit;
  const int unit = m.start / m.starts_per_unit - 1;  // (m.start has moved past the unit's last start)
  double* rec = board.records + (pair * units + static_cast<size_t>(unit)) * 3;
  __hip_atomic_store(rec + 0, m.best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(rec + 1, m.best_tp.a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(rec + 2, m.best_tp.b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned ticket = __hip_atomic_fetch_add(board.arrived + pair, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (ticket != static_cast<unsigned>(units - 1)) {
    m.phase = PH_IDLE;
    return false;
  }
  const double* all = board.records + pair * units * 3;
  double best = __builtin_huge_val();
  V2 best_tp{0.0, 0.0};
  for (int s = 0; s < units; ++s) {
    const double d = __hip_atomic_load(all + 3 * s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const double a = __hip_atomic_load(all + 3 * s + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const double b = __hip_atomic_load(all + 3 * s + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (d < best) {
      best = d;
      best_tp = V2{a, b};
    }
  }
  m.best = best;
  m.best_tp = best_tp;
  m.phase = PH_FINAL;
  return true;
}

// head of find_min's loop: stop tests, L-BFGS direction, line-search set-up (no objective evaluation in here).
// m.g arrives as the two central-difference numerators f(x + eps e_i) - f(x - eps e_i); the divisions happen here.
__device__ inline void iteration_head(Machine& m, const History& h) {
  const double eps = kDerivEps;
  m.g = V2{m.g.a / ((m.x.a + eps) - (m.x.a - eps)), m.g.b / ((m.x.b + eps) - (m.x.b - eps))};
  if (m.stop_used && fabs(m.cost - m.prev_val) < kMinDelta) return finish_start(m);
  m.stop_used = true;
  m.prev_val = m.cost;
  if (!(m.cost > kMinCost)) return finish_start(m);
  V2 dir{-m.g.a, -m.g.b};
  if (!m.been_used) {
    m.been_used = true;
  } else {
    const V2 s{m.x.a - m.prev_x.a, m.x.b - m.prev_x.b}, y{m.g.a - m.prev_g.a, m.g.b - m.prev_g.b};
    const double temp = lbfgs::dot2(s, y);
    // logical entry i (0 = oldest) sits in ring slot (ring_head + i) mod M
    if (fabs(temp) > kEps) {
      int slot;
      if (m.current_size < M) {
        slot = m.ring_head + m.current_size;
        if (slot >= M) slot -= M;
        ++m.current_size;
      } else {  // full: the oldest entry is overwritten and the ring advances (the reference shifts the arrays)
        slot = m.ring_head;
        m.ring_head = (m.ring_head + 1 == M) ? 0 : m.ring_head + 1;
      }
      h.sa(slot) = s.a; h.sb(slot) = s.b; h.ya(slot) = y.a; h.yb(slot) = y.b;  // rho = 1 / temp: see History
    } else {
      m.current_size = 0;
      m.ring_head = 0;
    }
    if (m.current_size > 0) {
      double alpha_i[M], rho_i[M];  // compile-time indices only (both loops are fully unrolled): registers, no scratch
      double rho_last = 0.0;
      V2 y_last{0.0, 0.0};
#pragma unroll
      for (int i = M - 1; i >= 0; --i) {
        if (i < m.current_size) {
          int p = m.ring_head + i;
          if (p >= M) p -= M;
          const V2 sp{h.sa(p), h.sb(p)}, yp{h.ya(p), h.yb(p)};
          const double rho = 1.0 / lbfgs::dot2(sp, yp);
          if (i == m.current_size - 1) {
            rho_last = rho;
            y_last = yp;
          }
          const double al = rho * lbfgs::dot2(sp, dir);
          alpha_i[i] = al;
          rho_i[i] = rho;
          dir = V2{dir.a - al * yp.a, dir.b - al * yp.b};
        }
      }
      double H0 = 1.0 / rho_last / lbfgs::dot2(y_last, y_last);
      H0 = lbfgs::clampd(0.001, 1000.0, H0);
      dir = V2{H0 * dir.a, H0 * dir.b};
#pragma unroll
      for (int i = 0; i < M; ++i) {
        if (i < m.current_size) {
          int p = m.ring_head + i;
          if (p >= M) p -= M;
          const V2 sp{h.sa(p), h.sb(p)}, yp{h.ya(p), h.yb(p)};
          const double beta = rho_i[i] * lbfgs::dot2(yp, dir);
          const double al = alpha_i[i];
          dir = V2{dir.a + (al - beta) * sp.a, dir.b + (al - beta) * sp.b};
        }
      }
    }
  }
  m.dir = dir;
  m.prev_x = m.x;
  m.prev_g = m.g;
  // line_search(f, x, dir, cost, dot(g, dir), 0.01, 0.9, min_allowable_cost, 100, eps): the part before its first loop
  m.f0 = m.cost;
  m.d0 = lbfgs::dot2(m.g, dir);
  if (fabs(m.d0) <= fabs(m.f0) * kEps) return line_search_done(m, 0);
  if (m.f0 <= kMinCost) return line_search_done(m, 0);
  m.mu = (kMinCost - m.f0) / (kRho * m.d0);
  double alpha = 1;
  if (m.mu < 0) alpha = -alpha;
  m.alpha = lbfgs::clampd(0.0, 0.65 * m.mu, alpha);
  m.last_alpha = 0;
  m.last_val = m.f0;
  m.last_val_der = m.d0;
  m.thresh = fabs(kSigma * m.d0);
  m.itr = 1;  // ++itr at the top of the first loop
  m.phase = PH_LS1_VAL;
}

__device__ inline bool wants_evaluation(const Machine& m) { return m.phase <= PH_FINAL; }

// the lane's objective value at query_point(m) goes into the machine: bookkeeping only (a store and the next phase);
// returns true when the pair is complete (that was the evaluation at the best of the nine starts)
__device__ inline bool take_value(Machine& m, double fv) {
  ++m.evals;
  switch (m.phase) {
    case PH_COST_INIT:
      m.cost = fv;
      m.phase = PH_G0P;
      return false;
    case PH_G0P:
    case PH_G1P:
    case PH_LS1_DP:
    case PH_LS2_DP:
      m.fp = fv;
      m.phase += 1;
      return false;
    case PH_G0M:
      m.g.a = m.fp - fv;  // numerator of the central difference (divided in iteration_head)
      m.phase = PH_G1P;
      return false;
    case PH_G1M:
      m.g.b = m.fp - fv;
      m.phase = m.after_ls ? PH_COST_AFTER : PH_WAIT_HEAD;
      return false;
    case PH_COST_AFTER:
      m.cost = fv;
      m.phase = PH_WAIT_HEAD;
      return false;
    case PH_LS1_VAL:
    case PH_LS2_VAL:
      m.val = fv;
      m.phase += 1;
      return false;
    case PH_LS1_DM:
      m.fm = fv;
      m.phase = PH_WAIT_LS1;
      return false;
    case PH_LS2_DM:
      m.fm = fv;
      m.phase = PH_WAIT_LS2;
      return false;
    case PH_FINAL:
      m.phase = PH_IDLE;
      return true;
    default:
      return false;
  }
}

// bottom of the line search's two loops (minimize_impl.hpp:300-405) for the lanes parked in PH_WAIT_LS1 / PH_WAIT_LS2.
// The three ways on -- return alpha / go (on) sectioning / extrapolate the bracket -- are decided first and carried out
// once below, so the cubic fit (poly_min_extrap: a square root and two divisions) exists once in the code.
__device__ inline void line_search_tests(Machine& m) {
  const bool l1 = m.phase == PH_WAIT_LS1, l2 = m.phase == PH_WAIT_LS2;
  if (!(l1 || l2)) return;
  const double eps = kDerivEps;
  // (members a lane may return are read once, up here: a load per branch would end up as a load through a selected
  // pointer, which keeps the members in scratch memory)
  const double val = m.val, alpha = m.alpha, a_in = m.a, b_in = m.b;
  const double val_der = (m.fp - m.fm) / ((alpha + eps) - (alpha - eps));
  enum { DONE, SECTION, EXTRAPOLATE } way;
  double done_alpha = alpha;
  if (l1) {
    if (val <= kMinCost) {
      way = DONE;
    } else if (val > m.f0 + kRho * alpha * m.d0 || val >= m.last_val) {
      m.a_val = m.last_val; m.a_val_der = m.last_val_der; m.b_val = val; m.b_val_der = val_der;
      m.a = m.last_alpha; m.b = alpha;
      way = SECTION;
    } else if (fabs(val_der) <= m.thresh) {
      way = DONE;
    } else if (m.last_alpha == alpha || m.itr >= kLsMaxIter) {
      way = DONE;
    } else if (val_der >= 0) {
      m.a_val = val; m.a_val_der = val_der; m.b_val = m.last_val; m.b_val_der = m.last_val_der;
      m.a = alpha; m.b = m.last_alpha;
      way = SECTION;
    } else {
      way = EXTRAPOLATE;
    }
  } else {
    if (val <= kMinCost || m.itr >= kLsMaxIter) {
      way = DONE;
    } else if (a_in == m.ls_first || b_in == m.ls_last) {
      way = DONE;
      done_alpha = b_in;
    } else if (fabs(dmax(fabs(a_in), fabs(b_in)) * m.d0) <= fabs(m.f0) * kEps) {
      way = DONE;
    } else if (val > m.f0 + kRho * alpha * m.d0 || val >= m.a_val) {
      m.b = alpha; m.b_val = val; m.b_val_der = val_der;
      way = SECTION;
    } else if (fabs(val_der) <= m.thresh) {
      way = DONE;
    } else {
      if ((b_in - a_in) * val_der >= 0) {
        m.b = a_in; m.b_val = m.a_val; m.b_val_der = m.a_val_der;
      }
      m.a = alpha; m.a_val = val; m.a_val_der = val_der;
      way = SECTION;
    }
  }
  if (way == DONE) return line_search_done(m, done_alpha);
  // SECTION: top of the sectioning loop, next trial alpha inside [a, b]; EXTRAPOLATE: next trial of the bracketing loop
  const bool fwd = m.last_alpha < alpha;
  const bool sec = way == SECTION;
  double pf0, pd0, pf1, pd1;
  if (sec) {
    pf0 = m.a_val; pd0 = m.a_val_der; pf1 = m.b_val; pd1 = m.b_val_der;
  } else if (fwd) {
    pf0 = m.last_val; pd0 = m.last_val_der; pf1 = val; pd1 = val_der;
  } else {
    pf0 = val; pd0 = val_der; pf1 = m.last_val; pd1 = m.last_val_der;
  }
  const double pm = lbfgs::poly_min_extrap(pf0, pd0, pf1, pd1, sec ? 1.0 : 1e10);
  ++m.itr;
  if (sec) {
    const double tau2 = 1.0 / 10.0, tau3 = 1.0 / 2.0;
    m.ls_first = m.a + tau2 * (m.b - m.a);
    m.ls_last = m.b - tau3 * (m.b - m.a);
    m.alpha = lbfgs::clampd(m.ls_first, m.ls_last, m.a + (m.b - m.a) * pm);
    m.phase = PH_LS2_VAL;
  } else {
    const double tau1a = 1.4, tau1b = 9;
    double first, last;
    if (m.mu > 0) {
      first = dmin(m.mu, alpha + tau1a * (alpha - m.last_alpha));
      last = dmin(m.mu, alpha + tau1b * (alpha - m.last_alpha));
    } else {
      first = dmax(m.mu, alpha + tau1a * (alpha - m.last_alpha));
      last = dmax(m.mu, alpha + tau1b * (alpha - m.last_alpha));
    }
    const double na = fwd ? m.last_alpha + (alpha - m.last_alpha) * pm : alpha + (m.last_alpha - alpha) * pm;
    m.alpha = lbfgs::clampd(first, last, na);
    m.last_alpha = alpha;
    m.last_val = val;
    m.last_val_der = val_der;
    m.phase = PH_LS1_VAL;
  }
}

// One round of the wave after its evaluation: the minimiser logic that is due (see "Scheduling").  `live` = the lane
// holds a pair.  All branch conditions are wave-uniform.
__device__ inline void scheduled_transitions(Machine& m, const History& h, bool live, unsigned round) {
  if (round % kLsEvery == 0 || !__any(live && wants_evaluation(m))) line_search_tests(m);
  if (round % kHeadEvery == 0 || !__any(live && wants_evaluation(m))) {
    if (m.phase == PH_WAIT_HEAD) iteration_head(m, h);
  }
}

}  // namespace lockstep
}  // namespace mhip
