// ellipsoid_lockstep.hpp -- the multistart L-BFGS of the ellipsoid distances (mundy_math/impl/minimize_impl.hpp:151-605
// as called by mundy_geom/distance/EllipsoidEllipsoid.hpp:106-151) run in lockstep across a wavefront.
//
// One lane still owns one pair, and performs exactly the arithmetic of the reference's find_min / line_search in exactly
// their order -- but the control flow is turned inside out.  Each lane carries the minimiser as an explicit state machine
// whose only externally visible act is "evaluate the objective at this point".  The wave's loop is then
//     evaluate the objective for all lanes at once  ->  every lane feeds the value to its own machine
// so the expensive part (two sincos, four quaternion rotations, two foot-point maps: ~95 % of the instructions) runs
// with all lanes converged whatever line-search branch, L-BFGS iteration, start point or pair each lane is in; only
// the few dozen flops of minimiser logic diverge.  A lane that finishes a pair takes the next one from a global
// counter, so no lane waits for the slowest pair of its wave either.  (One thread per pair with the minimiser called
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
  PH_IDLE
};

struct Machine {
  // find_min (minimize_impl.hpp:407-605 with the defaults of minimize.hpp:42-51)
  V2 x, g, prev_x, prev_g, dir;
  double cost, prev_val;
  bool been_used, stop_used, after_ls;
  int current_size, ring_head;  // the L-BFGS history itself lives in LDS (History below), as a ring buffer
  unsigned evals;               // objective evaluations of this pair (profiling: the fp64 roofline of the class)
  // line_search (minimize_impl.hpp:233-405)
  double f0, d0, mu, alpha, last_alpha, last_val, last_val_der, a, b, a_val, b_val, a_val_der, b_val_der, thresh;
  double ls_first, ls_last, val, fp;
  int itr;
  // multistart (EllipsoidEllipsoid.hpp:106-151)
  int start;
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
__device__ inline void begin_pair(Machine& m) {
  m.evals = 0;
  m.start = 0;
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

// top of the sectioning loop (minimize_impl.hpp line_search, second while): next trial alpha
__device__ inline void sectioning_top(Machine& m) {
  const double tau2 = 1.0 / 10.0, tau3 = 1.0 / 2.0;
  ++m.itr;
  m.ls_first = m.a + tau2 * (m.b - m.a);
  m.ls_last = m.b - tau3 * (m.b - m.a);
  m.alpha = m.a + (m.b - m.a) * lbfgs::poly_min_extrap(m.a_val, m.a_val_der, m.b_val, m.b_val_der, 1.0);
  m.alpha = lbfgs::clampd(m.ls_first, m.ls_last, m.alpha);
  m.phase = PH_LS2_VAL;
}

// the line search returned `alpha`: take the step, then gradient and cost at the new point
__device__ inline void line_search_done(Machine& m, double alpha) {
  m.x = V2{alpha * m.dir.a + m.x.a, alpha * m.dir.b + m.x.b};
  m.after_ls = true;
  m.phase = PH_G0P;
}

// one start is finished: keep it if it is the best so far, go to the next start or to the final evaluation
__device__ inline void finish_start(Machine& m) {
  if (m.cost < m.best) {
    m.best = m.cost;
    m.best_tp = m.x;
  }
  ++m.start;
  if (m.start < 9)
    begin_start(m);
  else
    m.phase = PH_FINAL;
}

// head of find_min's loop: stop tests, L-BFGS direction, line-search set-up (no objective evaluation in here)
__device__ inline void iteration_head(Machine& m, const History& h) {
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
      double alpha_i[M];  // compile-time indices only (both loops are fully unrolled): registers, no scratch
#pragma unroll
      for (int i = M - 1; i >= 0; --i) {
        if (i < m.current_size) {
          int p = m.ring_head + i;
          if (p >= M) p -= M;
          const V2 sp{h.sa(p), h.sb(p)}, yp{h.ya(p), h.yb(p)};
          const double rho = 1.0 / lbfgs::dot2(sp, yp);
          const double al = rho * lbfgs::dot2(sp, dir);
          alpha_i[i] = al;
          dir = V2{dir.a - al * yp.a, dir.b - al * yp.b};
        }
      }
      int pl = m.ring_head + m.current_size - 1;
      if (pl >= M) pl -= M;
      const V2 s_last{h.sa(pl), h.sb(pl)}, y_last{h.ya(pl), h.yb(pl)};
      const double rho_last = 1.0 / lbfgs::dot2(s_last, y_last);
      double H0 = 1.0 / rho_last / lbfgs::dot2(y_last, y_last);
      H0 = lbfgs::clampd(0.001, 1000.0, H0);
      dir = V2{H0 * dir.a, H0 * dir.b};
#pragma unroll
      for (int i = 0; i < M; ++i) {
        if (i < m.current_size) {
          int p = m.ring_head + i;
          if (p >= M) p -= M;
          const V2 sp{h.sa(p), h.sb(p)}, yp{h.ya(p), h.yb(p)};
          const double rho = 1.0 / lbfgs::dot2(sp, yp);
          const double beta = rho * lbfgs::dot2(yp, dir);
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

// feeds the objective value at query_point(m) to the machine; returns true when the pair is complete (PH_FINAL done)
__device__ inline bool advance(Machine& m, const History& h, double fv) {
  const double eps = kDerivEps;
  ++m.evals;
  switch (m.phase) {
    case PH_COST_INIT:
      m.cost = fv;
      m.phase = PH_G0P;
      return false;
    case PH_G0P:
      m.fp = fv;
      m.phase = PH_G0M;
      return false;
    case PH_G0M:
      m.g.a = (m.fp - fv) / ((m.x.a + eps) - (m.x.a - eps));
      m.phase = PH_G1P;
      return false;
    case PH_G1P:
      m.fp = fv;
      m.phase = PH_G1M;
      return false;
    case PH_G1M:
      m.g.b = (m.fp - fv) / ((m.x.b + eps) - (m.x.b - eps));
      if (m.after_ls) {
        m.phase = PH_COST_AFTER;
      } else {
        iteration_head(m, h);
      }
      return false;
    case PH_COST_AFTER:
      m.cost = fv;
      iteration_head(m, h);
      return false;
    case PH_LS1_VAL:
    case PH_LS2_VAL:
      m.val = fv;
      m.phase += 1;
      return false;
    case PH_LS1_DP:
    case PH_LS2_DP:
      m.fp = fv;
      m.phase += 1;
      return false;
    case PH_LS1_DM: {
      const double val = m.val, alpha = m.alpha;
      const double val_der = (m.fp - fv) / ((alpha + eps) - (alpha - eps));
      const double tau1a = 1.4, tau1b = 9;
      if (val <= kMinCost) {
        line_search_done(m, alpha);
        return false;
      }
      if (val > m.f0 + kRho * alpha * m.d0 || val >= m.last_val) {
        m.a_val = m.last_val; m.a_val_der = m.last_val_der; m.b_val = val; m.b_val_der = val_der;
        m.a = m.last_alpha; m.b = alpha;
        sectioning_top(m);
        return false;
      }
      if (fabs(val_der) <= m.thresh) {
        line_search_done(m, alpha);
        return false;
      }
      if (m.last_alpha == alpha || m.itr >= kLsMaxIter) {
        line_search_done(m, alpha);
        return false;
      }
      if (val_der >= 0) {
        m.a_val = val; m.a_val_der = val_der; m.b_val = m.last_val; m.b_val_der = m.last_val_der;
        m.a = alpha; m.b = m.last_alpha;
        sectioning_top(m);
        return false;
      }
      const double temp = alpha;
      double first, last;
      if (m.mu > 0) {
        first = dmin(m.mu, alpha + tau1a * (alpha - m.last_alpha));
        last = dmin(m.mu, alpha + tau1b * (alpha - m.last_alpha));
      } else {
        first = dmax(m.mu, alpha + tau1a * (alpha - m.last_alpha));
        last = dmax(m.mu, alpha + tau1b * (alpha - m.last_alpha));
      }
      double na;
      if (m.last_alpha < alpha)
        na = m.last_alpha + (alpha - m.last_alpha) * lbfgs::poly_min_extrap(m.last_val, m.last_val_der, val, val_der, 1e10);
      else
        na = alpha + (m.last_alpha - alpha) * lbfgs::poly_min_extrap(val, val_der, m.last_val, m.last_val_der, 1e10);
      m.alpha = lbfgs::clampd(first, last, na);
      m.last_alpha = temp;
      m.last_val = val;
      m.last_val_der = val_der;
      ++m.itr;
      m.phase = PH_LS1_VAL;
      return false;
    }
    case PH_LS2_DM: {
      const double val = m.val, alpha = m.alpha;
      const double val_der = (m.fp - fv) / ((alpha + eps) - (alpha - eps));
      if (val <= kMinCost || m.itr >= kLsMaxIter) {
        line_search_done(m, alpha);
        return false;
      }
      if (m.a == m.ls_first || m.b == m.ls_last) {
        line_search_done(m, m.b);
        return false;
      }
      const double max_possible_alpha = dmax(fabs(m.a), fabs(m.b));
      if (fabs(max_possible_alpha * m.d0) <= fabs(m.f0) * kEps) {
        line_search_done(m, alpha);
        return false;
      }
      if (val > m.f0 + kRho * alpha * m.d0 || val >= m.a_val) {
        m.b = alpha; m.b_val = val; m.b_val_der = val_der;
      } else {
        if (fabs(val_der) <= m.thresh) {
          line_search_done(m, alpha);
          return false;
        }
        if ((m.b - m.a) * val_der >= 0) {
          m.b = m.a; m.b_val = m.a_val; m.b_val_der = m.a_val_der;
        }
        m.a = alpha; m.a_val = val; m.a_val_der = val_der;
      }
      sectioning_top(m);
      return false;
    }
    case PH_FINAL:
      m.phase = PH_IDLE;
      return true;
    default:
      return false;
  }
}

}  // namespace lockstep
}  // namespace mhip
