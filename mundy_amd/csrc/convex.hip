// convex.hip -- the BBPGD solver backend (convex.hpp restated for gfx950): S1 vector kernels, the matrix-free contact
// operator (S2) and the fused BBPGD driver.
//
// Everything here is HBM-bandwidth bound (SpMV-like gathers and streaming vector passes): no MFMA.
//
// Fused iteration (mhip_bbpgd_solve_contact), 3-4 launches, no host round trip:
//   k_body        per body:       x_c = Proj(xt_c - step*gt_c) for the incident constraints (fixed order),
//                                 F = sum -/+ x_c n_c, T = sum -/+ r x (x_c n_c), (U, W) = (mt F, mr T)   [K24+K21+K22]
//   k_constraint  per constraint: x_c again (bitwise the same), g_c = dt * sdot(U, W) + q_c, store x, g;
//                                 block partials of max residual, sum dx^2, sum dx dg                       [K23+K14+K17+K16]
//   k_fold_partials (only above 4096 block partials) 64 workgroups fold contiguous slices of the partials
//   k_finalize    one workgroup:  ordered reduction of the partials, residual test, BB1 step, iteration count,
//                                 all kept in a device-resident state block                                  [a26/a27]
// The iterate lives packed as (x, g) pairs in an operator-owned ping-pong pair (K20's deep_copies become a parity
// flip); mhip_bbpgd_solve_* writes the caller's x, g, x_tmp, g_tmp once at the end with the reference's
// post-conditions (which array holds what).
#include <cstdlib>
#include <mutex>
#include <vector>

#include "mhip_internal.hpp"

namespace mhip {

constexpr unsigned kProfileStride = 8;         // profiling brackets every 8th iteration (event records cost ~10 us)
constexpr double kSmallStep = 1e-6;            // convex.hpp:479
constexpr double kBBEps = 1e-15 * 10;          // convex.hpp:511
constexpr double kLowest = -1.7976931348623157e308;  // Kokkos::Max<double> identity

// ------------------------------------------------------------------------------------------------------------------
// S1 vector kernels (KokkosBackend, convex.hpp:201-284).  The |alpha|,|beta| < 1e-15 branches are resolved on the
// host exactly as the reference does and select the kernel variant.
// ------------------------------------------------------------------------------------------------------------------
// Streaming kernels move 16 bytes per lane per access when the vectors are 16-byte aligned (VEC: elements in pairs,
// the odd tail element by one thread); element-wise arithmetic is unchanged, so results do not depend on VEC.
#ifndef MHIP_STREAM_UNROLL
#define MHIP_STREAM_UNROLL 4
#endif
// one tile per workgroup up to this many workgroups (measured, 2^27 doubles: capped at 2048 / 8192 workgroups with a
// grid-stride loop 5.3-5.4 / 5.4-5.8 TB/s, uncapped 5.6-6.0; profiles/r03_stream_ceiling.txt)
#ifndef MHIP_STREAM_GRID
#define MHIP_STREAM_GRID 1048576
#endif
constexpr int kStreamUnroll = MHIP_STREAM_UNROLL;
// a workgroup streams CONTIGUOUS tiles of kStreamUnroll x 256 16-byte elements (lane -> element tid + u * 256 of the
// tile): first element of this thread in its workgroup's first tile
__device__ inline size_t stream_first() { return blockIdx.x * (size_t)(kBlock * kStreamUnroll) + threadIdx.x; }
template <int MODE>  // 0: a*x+b*y   1: b*y   2: a*x   3: 0
__device__ inline double axpby_value(double alpha, double x, double beta, double y) {
  if (MODE == 0) return alpha * x + beta * y;
  if (MODE == 1) return y * beta;
  if (MODE == 2) return alpha * x;
  return 0.0;
}
template <int MODE, bool VEC>
__global__ void __launch_bounds__(kBlock) k_axpby(size_t n, double alpha, const double* __restrict__ x, double beta,
                                                 double* __restrict__ y) {
  const size_t tid = blockIdx.x * (size_t)blockDim.x + threadIdx.x, nth = (size_t)gridDim.x * blockDim.x;
  if (VEC) {
    const double2* x2 = reinterpret_cast<const double2*>(x);
    double2* y2 = reinterpret_cast<double2*>(y);
    // kStreamUnroll 16-byte accesses per lane and array in flight (all loads of a round issued before its first store)
    const size_t n2 = n / 2;
    for (size_t i0 = stream_first(); i0 < n2; i0 += nth * kStreamUnroll) {
      double2 a[kStreamUnroll], b[kStreamUnroll];
#pragma unroll
      for (int u = 0; u < kStreamUnroll; ++u) {
        const size_t i = i0 + u * kBlock;
        a[u] = ((MODE == 0 || MODE == 2) && i < n2) ? x2[i] : make_double2(0.0, 0.0);
        b[u] = ((MODE == 0 || MODE == 1) && i < n2) ? y2[i] : make_double2(0.0, 0.0);
      }
#pragma unroll
      for (int u = 0; u < kStreamUnroll; ++u) {
        const size_t i = i0 + u * kBlock;
        if (i < n2)
          y2[i] = make_double2(axpby_value<MODE>(alpha, a[u].x, beta, b[u].x), axpby_value<MODE>(alpha, a[u].y, beta, b[u].y));
      }
    }
    if ((n & 1) && tid == 0) y[n - 1] = axpby_value<MODE>(alpha, MODE == 1 || MODE == 3 ? 0.0 : x[n - 1], beta, y[n - 1]);
  } else {
    for (size_t i = tid; i < n; i += nth)
      y[i] = axpby_value<MODE>(alpha, (MODE == 0 || MODE == 2) ? x[i] : 0.0, beta, (MODE == 0 || MODE == 1) ? y[i] : 0.0);
  }
}
template <int MODE, bool VEC>
__global__ void __launch_bounds__(kBlock) k_wrapped_axpbyz(size_t n, double alpha, const double* __restrict__ x,
                                                          double beta, const double* __restrict__ y,
                                                          double* __restrict__ z, Space sp) {
  const size_t tid = blockIdx.x * (size_t)blockDim.x + threadIdx.x, nth = (size_t)gridDim.x * blockDim.x;
  if (VEC) {
    const double2* x2 = reinterpret_cast<const double2*>(x);
    const double2* y2 = reinterpret_cast<const double2*>(y);
    double2* z2 = reinterpret_cast<double2*>(z);
    const size_t n2 = n / 2;
    for (size_t i0 = stream_first(); i0 < n2; i0 += nth * kStreamUnroll) {
      double2 a[kStreamUnroll], b[kStreamUnroll];
#pragma unroll
      for (int u = 0; u < kStreamUnroll; ++u) {
        const size_t i = i0 + u * kBlock;
        a[u] = ((MODE == 0 || MODE == 2) && i < n2) ? x2[i] : make_double2(0.0, 0.0);
        b[u] = ((MODE == 0 || MODE == 1) && i < n2) ? y2[i] : make_double2(0.0, 0.0);
      }
#pragma unroll
      for (int u = 0; u < kStreamUnroll; ++u) {
        const size_t i = i0 + u * kBlock;
        if (i < n2)
          z2[i] = make_double2(sp.project(axpby_value<MODE>(alpha, a[u].x, beta, b[u].x)),
                               sp.project(axpby_value<MODE>(alpha, a[u].y, beta, b[u].y)));
      }
    }
    if ((n & 1) && tid == 0)
      z[n - 1] = sp.project(axpby_value<MODE>(alpha, (MODE == 0 || MODE == 2) ? x[n - 1] : 0.0, beta,
                                              (MODE == 0 || MODE == 1) ? y[n - 1] : 0.0));
  } else {
    for (size_t i = tid; i < n; i += nth)
      z[i] = sp.project(axpby_value<MODE>(alpha, (MODE == 0 || MODE == 2) ? x[i] : 0.0, beta,
                                          (MODE == 0 || MODE == 1) ? y[i] : 0.0));
  }
}
template <bool VEC>
__global__ void __launch_bounds__(kBlock) k_copy(size_t n, double* __restrict__ dst, const double* __restrict__ src) {
  const size_t tid = blockIdx.x * (size_t)blockDim.x + threadIdx.x, nth = (size_t)gridDim.x * blockDim.x;
  if (VEC) {
    const size_t n2 = n / 2;
    for (size_t i0 = stream_first(); i0 < n2; i0 += nth * kStreamUnroll) {
      double2 a[kStreamUnroll];
#pragma unroll
      for (int u = 0; u < kStreamUnroll; ++u)
        a[u] = (i0 + u * kBlock < n2) ? reinterpret_cast<const double2*>(src)[i0 + u * kBlock] : make_double2(0.0, 0.0);
#pragma unroll
      for (int u = 0; u < kStreamUnroll; ++u)
        if (i0 + u * kBlock < n2) reinterpret_cast<double2*>(dst)[i0 + u * kBlock] = a[u];
    }
    if ((n & 1) && tid == 0) dst[n - 1] = src[n - 1];
  } else {
    for (size_t i = tid; i < n; i += nth) dst[i] = src[i];
  }
}
__global__ void __launch_bounds__(kBlock) k_fill(size_t n, double* __restrict__ dst, double v) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    dst[i] = v;
}
inline bool aligned16(const void* a, const void* b = nullptr, const void* c = nullptr) {
  return ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c)) & 15u) == 0;
}
// streaming grids: enough workgroups to keep every CU's queue full, capped so the tail wave is short
inline unsigned stream_grid(size_t work_items) {
  const size_t g = (work_items + (size_t)kBlock * kStreamUnroll - 1) / ((size_t)kBlock * kStreamUnroll);
  return static_cast<unsigned>(g == 0 ? 1 : (g > MHIP_STREAM_GRID ? MHIP_STREAM_GRID : g));
}

// residual term of one unknown (policies convex.hpp:434-496)
__device__ inline double residual_term(int kind, double x, double g, const Space& sp) {
  if (kind == MHIP_RESIDUAL_PROJECTED_GRADIENT) return (x < kZeroTol) ? ((0.0 < g) ? g : 0.0) : fabs(g);
  return fabs(x - sp.project(x - kSmallStep * g));
}

// reductions: partials[block] then a single-workgroup final pass.  OP: 0 = sum (x-y)^2, 1 = sum (x1-x2)(y1-y2),
// 2 = max resid.  The sums are double-double (see mhip_internal.hpp): partials are (hi, lo) planes kMaxGrid apart.
template <int OP>
__global__ void __launch_bounds__(kBlock)
    k_reduce_partials(size_t n, const double* __restrict__ a, const double* __restrict__ b,
                      const double* __restrict__ c, const double* __restrict__ d, int resid_kind, Space sp,
                      double* __restrict__ partials) {
  __shared__ double scratch[2 * kBlock / 64];
  double mx = kLowest;
  DD acc{0.0, 0.0};
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    if (OP == 0) {
      const double df = a[i] - b[i];
      dd_add(acc, df * df);
    } else if (OP == 1) {
      dd_add(acc, (a[i] - b[i]) * (c[i] - d[i]));
    } else {
      const double v = residual_term(resid_kind, a[i], b[i], sp);
      if (v > mx) mx = v;
    }
  }
  if (OP == 2) {
    const double r = block_max(mx, scratch);
    if (threadIdx.x == 0) partials[blockIdx.x] = r;
  } else {
    const DD r = block_sum(acc, scratch);
    if (threadIdx.x == 0) {
      partials[blockIdx.x] = r.hi;
      partials[kMaxGrid + blockIdx.x] = r.lo;
    }
  }
}
template <int OP>
__global__ void __launch_bounds__(kBlock) k_reduce_final(int nparts, const double* __restrict__ partials,
                                                        double* __restrict__ out) {
  __shared__ double scratch[2 * kBlock / 64];
  double mx = kLowest;
  DD acc{0.0, 0.0};
  for (int i = threadIdx.x; i < nparts; i += blockDim.x) {
    if (OP == 2) {
      if (partials[i] > mx) mx = partials[i];
    } else {
      dd_add(acc, DD{partials[i], partials[kMaxGrid + i]});
    }
  }
  if (OP == 2) {
    const double r = block_max(mx, scratch);
    if (threadIdx.x == 0) *out = r;
  } else {
    const DD r = block_sum(acc, scratch);
    if (threadIdx.x == 0) *out = dd_value(r);
  }
}

// y = A x, row-major n x n: one wavefront per row (KokkosBlas::gemv "N"), double-double row sums
__global__ void __launch_bounds__(kBlock) k_gemv(size_t n, const double* __restrict__ A, const double* __restrict__ x,
                                                double* __restrict__ y) {
  const int lane = threadIdx.x & 63;
  const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6;
  const size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
  for (size_t row = wave; row < n; row += nwaves) {
    const double* Ar = A + row * n;
    DD acc{0.0, 0.0};
    for (size_t j = lane; j < n; j += 64) dd_add(acc, Ar[j] * x[j]);
    acc = wave_sum(acc);
    if (lane == 0) y[row] = dd_value(acc);
  }
}

// ------------------------------------------------------------------------------------------------------------------
// MundyMathBackend<Scalar, N> (convex.hpp:288-350): the whole BBPGD solve inside one thread on a small dense problem.
// Right-fold dots (as mundy::math::dot / Matrix * Vector), no |alpha|,|beta| < 1e-15 branches, reduce_max from -inf:
// bit-identical to a scalar evaluation of that backend.
// ------------------------------------------------------------------------------------------------------------------
constexpr int kSmallMax = 16;

__device__ inline double rfold_dot(const double* a, const double* b, int n) {
  double acc = a[n - 1] * b[n - 1];
  for (int i = n - 2; i >= 0; --i) acc = a[i] * b[i] + acc;
  return acc;
}

__global__ void __launch_bounds__(64)
    k_small_cqpp(size_t batch, int n, const double* __restrict__ A, const double* __restrict__ q, Space sp,
                 int resid_kind, unsigned max_iters, double tol, double* __restrict__ xout, double* __restrict__ gout,
                 unsigned* __restrict__ iters, double* __restrict__ resout, int* __restrict__ conv) {
  const size_t p = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (p >= batch) return;
  const double* Ap = A + p * (size_t)n * n;
  const double* qp = q + p * (size_t)n;
  double x[kSmallMax], g[kSmallMax], xt[kSmallMax], gt[kSmallMax], d1[kSmallMax], d2[kSmallMax];
  auto apply = [&](const double* in, double* out) {
    for (int i = 0; i < n; ++i) out[i] = rfold_dot(Ap + (size_t)i * n, in, n);
  };
  auto resid = [&](const double* xx, const double* gg) {
    double mx = -__builtin_huge_val();
    for (int i = 0; i < n; ++i) {
      const double v = residual_term(resid_kind, xx[i], gg[i], sp);
      if (v > mx) mx = v;
    }
    return resid_kind == MHIP_RESIDUAL_PROJECTED_GRADIENT ? mx : mx / kSmallStep;
  };
  for (int i = 0; i < n; ++i) {
    x[i] = xout[p * (size_t)n + i];
    xt[i] = x[i];
    g[i] = 0.0;
  }
  apply(xt, gt);
  for (int i = 0; i < n; ++i) gt[i] = 1.0 * qp[i] + 1.0 * gt[i];
  double res = resid(xt, gt);
  double step = 1.0 / res;
  unsigned iter = 0;
  bool converged = res <= tol;
  if (converged)
    for (int i = 0; i < n; ++i) g[i] = gt[i];
  while (!(converged || iter >= max_iters)) {
    for (int i = 0; i < n; ++i) x[i] = sp.project(1.0 * xt[i] + (-step) * gt[i]);
    apply(x, g);
    for (int i = 0; i < n; ++i) g[i] = 1.0 * qp[i] + 1.0 * g[i];
    res = resid(x, g);
    if (res <= tol) {
      converged = true;
      break;
    }
    for (int i = 0; i < n; ++i) {
      d1[i] = x[i] - xt[i];
      d2[i] = g[i] - gt[i];
    }
    const double num = rfold_dot(d1, d1, n);
    double den = rfold_dot(d1, d2, n);
    den += kBBEps * (fabs(den) < kBBEps ? 1.0 : 0.0);
    step = num / den;
    for (int i = 0; i < n; ++i) {
      xt[i] = x[i];
      gt[i] = g[i];
    }
    ++iter;
  }
  for (int i = 0; i < n; ++i) {
    xout[p * (size_t)n + i] = x[i];
    gout[p * (size_t)n + i] = g[i];
  }
  iters[p] = iter;
  resout[p] = res;
  conv[p] = converged ? 1 : 0;
}

// thread-local scratch for the blocking S1 reductions (the stream is synchronised before they return)
struct ReduceScratch {
  DeviceBuffer dev;  // 2 planes of kMaxGrid partials + 1 result
  double* host = nullptr;
  int ensure() {
    if (int e = dev.reserve((2 * kMaxGrid + 8) * sizeof(double))) return e;
    if (!host) MHIP_HIP(hipHostMalloc(reinterpret_cast<void**>(&host), 8 * sizeof(double)));
    return MHIP_SUCCESS;
  }
};
ReduceScratch& reduce_scratch() {
  thread_local ReduceScratch s;
  return s;
}

template <int OP>
int reduce_to_host(size_t n, const double* a, const double* b, const double* c, const double* d, int resid_kind,
                   Space sp, double* result, hipStream_t stream) {
  ReduceScratch& rs = reduce_scratch();
  if (int e = rs.ensure()) return e;
  double* partials = rs.dev.as<double>();
  double* out = partials + 2 * kMaxGrid;
  const unsigned grid = grid_for(n);
  k_reduce_partials<OP><<<grid, kBlock, 0, stream>>>(n, a, b, c, d, resid_kind, sp, partials);
  MHIP_LAUNCH_CHECK();
  k_reduce_final<OP><<<1, kBlock, 0, stream>>>(n == 0 ? 0 : static_cast<int>(grid), partials, out);
  MHIP_LAUNCH_CHECK();
  MHIP_HIP(hipMemcpyAsync(rs.host, out, sizeof(double), hipMemcpyDeviceToHost, stream));
  MHIP_HIP(hipStreamSynchronize(stream));
  *result = rs.host[0];
  return MHIP_SUCCESS;
}

int launch_axpby(size_t n, double alpha, const double* x, double beta, double* y, hipStream_t s) {
  if (n == 0) return MHIP_SUCCESS;
  const bool az = fabs(alpha) < kZeroTol, bz = fabs(beta) < kZeroTol;
  const int mode = (!az && !bz) ? 0 : ((az && !bz) ? 1 : ((!az && bz) ? 2 : 3));
  const bool vec = aligned16(x, y);
  const unsigned g = stream_grid(vec ? (n + 1) / 2 : n);
#define AXPBY(M)                                                        \
  do {                                                                  \
    if (vec) k_axpby<M, true><<<g, kBlock, 0, s>>>(n, alpha, x, beta, y); \
    else k_axpby<M, false><<<g, kBlock, 0, s>>>(n, alpha, x, beta, y);  \
  } while (0)
  if (mode == 0) AXPBY(0); else if (mode == 1) AXPBY(1); else if (mode == 2) AXPBY(2); else AXPBY(3);
#undef AXPBY
  MHIP_LAUNCH_CHECK();
  return MHIP_SUCCESS;
}
int launch_wrapped_axpbyz(size_t n, double alpha, const double* x, double beta, const double* y, double* z, Space sp,
                          hipStream_t s) {
  if (n == 0) return MHIP_SUCCESS;
  const bool az = fabs(alpha) < kZeroTol, bz = fabs(beta) < kZeroTol;
  const int mode = (!az && !bz) ? 0 : ((az && !bz) ? 1 : ((!az && bz) ? 2 : 3));
  const bool vec = aligned16(x, y, z);
  const unsigned g = stream_grid(vec ? (n + 1) / 2 : n);
#define WAXPBYZ(M)                                                                 \
  do {                                                                             \
    if (vec) k_wrapped_axpbyz<M, true><<<g, kBlock, 0, s>>>(n, alpha, x, beta, y, z, sp); \
    else k_wrapped_axpbyz<M, false><<<g, kBlock, 0, s>>>(n, alpha, x, beta, y, z, sp);    \
  } while (0)
  if (mode == 0) WAXPBYZ(0); else if (mode == 1) WAXPBYZ(1); else if (mode == 2) WAXPBYZ(2); else WAXPBYZ(3);
#undef WAXPBYZ
  MHIP_LAUNCH_CHECK();
  return MHIP_SUCCESS;
}
int launch_copy(size_t n, double* dst, const double* src, hipStream_t s) {
  if (n == 0 || dst == src) return MHIP_SUCCESS;
  const bool vec = aligned16(dst, src);
  const unsigned g = stream_grid(vec ? (n + 1) / 2 : n);
  if (vec) k_copy<true><<<g, kBlock, 0, s>>>(n, dst, src);
  else k_copy<false><<<g, kBlock, 0, s>>>(n, dst, src);
  MHIP_LAUNCH_CHECK();
  return MHIP_SUCCESS;
}

// ------------------------------------------------------------------------------------------------------------------
// contact operator
// ------------------------------------------------------------------------------------------------------------------
// Streams that are read once per launch (entries, records, contact geometry) are loaded non-temporally so that they do
// not push the GATHERED tables (packed iterates, body rows) out of L2 / Infinity Cache between the two sweeps.
// MHIP_NT: bit 0 = the body sweep's compact streams, bit 1 = the constraint sweep's per-contact streams.  Measured on the
// 10^6-rod solve (profiles/r03_ab_nt.txt): k_body 0.0943 -> 0.0904 ms, k_constraint 0.0768 -> 0.0730 ms, step 148.5 ->
// 141.7 ms.  (The per-body rows -- row pointers, masks, mobilities, axes -- loaded that way cost k_body 4 %: neighbouring
// lanes share their lines.)
#ifndef MHIP_NT
#define MHIP_NT 3
#endif
typedef double mhip_d2v __attribute__((ext_vector_type(2)));
typedef int mhip_i2v __attribute__((ext_vector_type(2)));
template <bool NT> __device__ inline double ld_s(const double* p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT> __device__ inline int32_t ld_s(const int32_t* p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT> __device__ inline double2 ld_s(const double2* p) {
  if (!NT) return *p;
  const mhip_d2v v = __builtin_nontemporal_load(reinterpret_cast<const mhip_d2v*>(p));
  return make_double2(v.x, v.y);
}
template <bool NT> __device__ inline int2 ld_s(const int2* p) {
  if (!NT) return *p;
  const mhip_i2v v = __builtin_nontemporal_load(reinterpret_cast<const mhip_i2v*>(p));
  return make_int2(v.x, v.y);
}
template <bool NT> __device__ inline V3 ld_s3(const double* p, size_t i) {
  return {ld_s<NT>(p + 3 * i), ld_s<NT>(p + 3 * i + 1), ld_s<NT>(p + 3 * i + 2)};
}

struct SolverState {  // device resident
  double step, residual, num, den;
  unsigned iter;   // reported iteration count
  unsigned flips;  // completed non-terminal iterations = parity of the x/x_tmp ping-pong
  int converged, done, converged_at_init;
};

enum XMode { X_APPLY = 0, X_INIT = 1, X_SOLVE = 2 };
enum Kinematics { KIN_TRANS = 0, KIN_RIGID = 1, KIN_ROD = 2 };  // spheres | explicit lever arms | rod-compressed

struct OpView {
  size_t C, N;
  const int2* pairs;
  const double *normal, *ra, *rb, *mt, *mr;
  const int32_t *inc_ptr, *inc;  // body -> incident constraints, entry = (c << 1) | side, ascending
  const double* half;            // per incidence entry: (n_c, r_side), 6 doubles (3 when translation only)
  double* vel;                   // [N][6]
  double dt;
  // domain decomposition: bodies [body_first, body_first + body_count) are owned by this rank and swept
  // by k_body; the other rows of vel are ghosts filled by the halo exchange.  counted[c] != 0 marks the contacts this
  // rank contributes to the global reductions (a contact duplicated on two ranks is counted by the owner of its
  // lower body).  Defaults: all bodies, all contacts.
  size_t body_first, body_count;
  const unsigned char* counted;
  // rod-compressed kinematics (KIN_ROD): a rod's contact point lies on its centreline, cp = c + (s - 1/2) u with
  // u = p1 - p0, so the lever arm is ONE scalar per contact side plus a per-body axis (an L2-resident table):
  //   torque  T_b = u_b x sum_e coef_e f_e          velocity at the contact  v = U_b + coef (W_b x u_b) = U_b + coef Z_b
  // The sweeps then stream (s, t) [16 B] instead of (ra, rb) [48 B], 32-byte half-edge records instead of 48-byte
  // ones, and gather the same 48-byte body rows, which now hold (U, Z).  W is kept in `omega` for the integrator.
  const double *arc_s, *arc_t;  // [C] arclength parameters of the two closest points
  const double* axis;           // [N][3] u = p1 - p0
  double* omega;                // [N][3] angular velocity W
  int xcd_aware;  // T of xcd_tile(): consecutive tiles per XCD within a window (performance only; 0 = off)
  // constraint sweeps over a sub-range [c_first, c_end) (the staged solver sweeps interior contacts while the halo of
  // the boundary ones is in flight); its block partials go to slots part_offset + blockIdx of planes part_stride apart
  // (part_stride = 0: one sweep over everything, planes gridDim apart)
  size_t c_first, c_end;
  unsigned part_offset, part_stride;
  // Activity masks (packed LCP solves): bit p of body_mask[b] is 0 iff the contact behind the p-th incidence entry of
  // body b has x == 0 and 0 <= g < inf in the iterate the constraint sweep wrote last.  For any finite step >= 0 such
  // a contact's next iterate is Proj(0 - step g) = 0 exactly, i.e. it adds +/-0 to the body's sums -- the body sweep
  // walks only the set bits and never touches the entry, iterate or record of the others (two thirds of the list at
  // 10^6 rods).  The constraint sweep keeps the masks current: it knows the old state (from the iterate it read) and
  // the new one (from the iterate it writes) and flips the two bits with atomicXor when they differ, which after the
  // first few iterations is rare.  A set bit only means "evaluate": stale ones are harmless, and a clear bit is never
  // stale because every contact is re-evaluated every iteration.  Entries beyond the 64th of a body are always walked.
  unsigned long long* body_mask;  // [N]
  const unsigned char* pos;       // [2C]: slot of (c, side) in its body's incidence list, 255 if >= 64
  // Active lists.  The masks say which third of a body's entries matters, but those entries are strewn over its list
  // (overlap at the start of the step predicts only ~60 % of them), so the masked sweep touches sectors of which it
  // uses a fraction (PMC: 689 MB per launch against 346 MB it needs).  Every time the host polls for convergence the
  // flagged entries and their records are copied, body by body, into compact arrays (k_active_*): the sweep then
  // streams exactly what it uses.  Between two snapshots the masks keep changing a little: a contact that became active
  // since (mask & ~snap_mask) is walked through the full list as before, one that became inactive still sits in the
  // compact list and evaluates to lambda = 0 -- it adds nothing.  Same terms, and every sum is a double-double pair
  // rounded once: the iterates do not notice.  null = no snapshot yet (the first iterations walk the masks).
  const int32_t* aptr;                  // [N + 1] body -> its compact range
  const int32_t* aent;                  // entries (c << 1 | side), snapshot order
  const double* arec;                   // their records, same layout as `half`
  const unsigned long long* snap_mask;  // [N] the masks the snapshot was taken from
  // Tiered solves (see "Cold tier" below): the body rows ping-pong between vel (even parity) and vel_alt like the
  // iterate does, so that the rows of the last TWO iterates exist when the solve ends; drift [N] accumulates, per body,
  // an upper bound of how far any of its contact-point velocities has moved (times dt) since the bookkeeping began.
  // null = off.
  double* vel_alt;
  double* drift;
  const double* arm_max;  // [N] (vector-arm operator, tiered solves) longest lever arm of each body's contacts
  // tiered solves: a body whose drift reaches fire_at[b] has a sleeping contact to wake: the body sweep lists it in
  // fired (tier_counters[1] = length).  null = nobody sleeps.
  const double* fire_at;
  int32_t* fired;
  unsigned long long* tier_counters;
};

// XCD-aware work mapping (MI355X: 8 XCDs, each with a private 4 MiB L2; workgroups are dealt round-robin over the XCDs,
// so blockIdx % 8 labels the XCD).  Work is cut into tiles of 256 items.  Within every window of 8 T consecutive tiles
// the tiles are permuted so that one XCD gets T CONSECUTIVE tiles: with Z-ordered bodies the rows its workgroups
// gather (body rows, the iterate of neighbouring contacts) then come from a compact range that is fetched into ONE L2
// instead of all eight, while the eight XCDs together still sweep the arrays front to back (the window is what is
// resident on the chip at one time, so the HBM streams stay sequential).  T = 0 disables the permutation.
// Placement is a performance assumption only: any mapping gives the same results.
__device__ inline size_t xcd_tile(size_t lin, size_t ntiles, unsigned T) {
  if (T == 0) return lin;
  const size_t W = 8 * (size_t)T;
  const size_t g = lin / W, r = lin % W;
  if ((g + 1) * W > ntiles) return lin;  // the last, partial window keeps the identity
  return g * W + (r % 8) * T + r / 8;
}

// the iterate a constraint carries in a given mode; bitwise identical wherever it is evaluated.
// PACKED: the fused and staged solvers keep (x, g) of one iterate interleaved, 16 bytes per constraint in an
// operator-owned ping-pong pair, so the body sweep's gather of a half edge's iterate is ONE 16-byte access instead of
// two 8-byte accesses into separate arrays (the gathers, not the streams, are what k_body waits on).  xt then points
// at the packed array of the current iterate; in X_INIT it is the caller's plain x in both layouts.
// (two steps, so that a sweep can issue the loads of several contacts before it uses the first)
template <int MODE, bool PACKED, bool NT = false>
__device__ inline double2 iterate_load(size_t c, const double* __restrict__ xt, const double* __restrict__ gt) {
  if (MODE == X_SOLVE) {
    if (PACKED) return ld_s<NT>(reinterpret_cast<const double2*>(xt) + c);
    return make_double2(xt[c], gt[c]);
  }
  return make_double2(xt[c], 0.0);
}
template <int MODE>
__device__ inline double iterate_value(double2 p, double step, bool step_is_zero, const Space& sp) {
  if (MODE == X_SOLVE) {
    // wrapped_axpbyz(1, x_tmp, -step, g_tmp, x, space) with its beta ~ 0 branch (convex.hpp:228-247, :647)
    const double v = step_is_zero ? 1.0 * p.x : 1.0 * p.x + (-step) * p.y;
    return sp.project(v);
  }
  return p.x;
}
template <int MODE, bool PACKED, bool NT = false>
__device__ inline double iterate_x(size_t c, const double* __restrict__ xt, const double* __restrict__ gt,
                                   double step, bool step_is_zero, const Space& sp, double* x_old = nullptr,
                                   double* g_old = nullptr) {
  const double2 p = iterate_load<MODE, PACKED, NT>(c, xt, gt);
  if (MODE == X_SOLVE) {
    if (x_old) *x_old = p.x;
    if (g_old) *g_old = p.y;
  }
  return iterate_value<MODE>(p, step, step_is_zero, sp);
}

// Body sweep.  G lanes cooperate on one body: lane `sub` walks the body's half-edge records sub, sub+G, ... (a
// contiguous, coalesced stream: records are stored in incidence order), U of them at a time, then a G-lane butterfly
// adds the partial sums.  The order is fixed (no atomics), so results are bitwise reproducible run to run.
//   half-edge record e, written once at operator creation: (n_c, r_side) 48 B | (n_c, s - 1/2) 32 B for rods |
//   n_c 24 B translation-only
// compulsory bytes (rods): per half edge 4 (entry) + 32 (record), 16 per contact for its iterate (gathered by both
// of its half edges); per body 4 (row pointer) + 16 (mobilities) + 24 (axis) + 48 (velocity row) + 24 (omega).
#ifndef MHIP_KBODY_WAVES
#define MHIP_KBODY_WAVES 1
#endif
// entries per lane and chunk of the flat stream over the compact active lists (0: per-body chains there too).
// Measured (profiles/r03_ab_nt.txt, 10^6 rods): 0 -> 0.0986 ms, 1 -> 0.1028, 2 -> 0.0944, 4 -> 0.1071 (48 KB of LDS)
#ifndef MHIP_KBODY_FLAT
#define MHIP_KBODY_FLAT 2
#endif
// 1: the sweep's early-fetched last words wait in LDS instead of registers (rods: 110 -> 96 VGPRs, 24 + 8 KB of LDS: FIVE
// workgroups per CU with the chain still four levels long).  Measured, same box, three runs each
// (profiles/r04_ab_experiments.txt): 0.0807 ms against 0.0772 -- a fifth resident workgroup makes the sweep SLOWER.  Off.
#ifndef MHIP_KCON_DYN_LDS    // A/B only: the same for the packed constraint sweep
#define MHIP_KCON_DYN_LDS 0
#endif
#ifndef MHIP_KBODY_DYN_LDS   // A/B only: unused dynamic LDS per workgroup of the flat sweep (caps the workgroups per CU)
#define MHIP_KBODY_DYN_LDS 0
#endif
#ifndef MHIP_KBODY_STASH
#define MHIP_KBODY_STASH 0
#endif
#ifndef MHIP_KBODY_EARLY_FROM   // chunk size (x 256 entries) from which the sweep's last words are fetched at its top
#define MHIP_KBODY_EARLY_FROM 3
#endif
#ifndef MHIP_KBODY_FLAT3_ABOVE   // share of a 2 x 256-entry chunk a workgroup's lists may fill before the chunk is 3 x 256
#define MHIP_KBODY_FLAT3_ABOVE 0.9
#endif
// FLATP > 0 (packed solves with a snapshot of the active lists): the compact lists are not walked body by body --
// the workgroup's 256 threads STREAM the entries of its 256 / G bodies flat, FLATP x 256 at a time (every lane has the
// same work whatever its body's share of the list; entry, record and the gather of the iterate are FLATP independent
// chains per lane; no lane waits for its body's row pointer first), leave (n, coef, +/-lambda, +/-dlambda) of every
// entry in LDS, and the G lanes of a body then add up their body's slice of that image -- the double-double sums see
// the same terms (rounded once: any order gives the same bits).  What the snapshot does not cover (entries that became
// active since, lists beyond 64 entries, steps outside [0, finite]) takes the per-body chains as before.
#ifdef MHIP_EXP_COUNT_MM
__device__ unsigned long long g_dbg[4];
#endif
// TRACK (tiered solves): 0 = no drift bookkeeping; 1 = the drift is the difference of the body's new row and its row of
// the previous iterate (an extra 48-byte read per body: free while the row tables live in the Infinity Cache, i.e. up
// to ~1.7 * 10^6 bodies); 2 = the change of the force is accumulated beside the sums from +/-(lam - x_old) n of the
// entries walked (rounds 2-3: no extra read, but 18 VGPRs and a (n, coef, +/-lam, +/-dlam) image of 48 B per entry) --
// what systems whose row tables outgrow the cache take: 4 * 10^6 / 16 * 10^6 rods 0.329 / 1.303 ms with 1, 0.30 / 1.17
// with 2 (profiles/r04_large_sizes.txt).  Chosen by op_drift_source().
template <int MODE, int KIN, int G, int U, bool PACKED, int TRACK = 0, int FLATP = 0>
__global__ void __launch_bounds__(kBlock, MHIP_KBODY_WAVES)
    k_body(OpView op, const SolverState* __restrict__ st, const double* __restrict__ X0, const double* __restrict__ X1,
           const double* __restrict__ G0, const double* __restrict__ G1, Space sp) {
  constexpr bool FLAT = FLATP > 0 && MODE == X_SOLVE && PACKED;
  // the image holds PRODUCTS: the entry's force f = +/-lambda n and, with it, what the torque sum takes -- the arclength
  // coefficient (rods: S += coef f) or r x f itself (vector arms): 32 / 48 bytes per entry (round 3 stored (n, arm,
  // +/-lambda, +/-dlambda): 48 / 64), i.e. 24 / 36 KB per workgroup at 3 x 256 entries
  constexpr bool kRegTrack = TRACK == 2 && MODE == X_SOLVE && PACKED;   // (image of rounds 2-3: see TRACK)
  constexpr int kFlatPlanes = kRegTrack ? ((KIN == KIN_RIGID) ? 4 : 3) : ((KIN == KIN_RIGID) ? 3 : 2);
  __shared__ double2 flat_img[FLAT ? kFlatPlanes * FLATP * kBlock : 1];
  // one tile = one workgroup's worth of bodies (see xcd_tile), the grid covers them once
  const size_t tile = xcd_tile(blockIdx.x, gridDim.x, op.xcd_aware);
  const size_t t = tile * (size_t)blockDim.x + threadIdx.x;
  const int sub = static_cast<int>(t % G);
  const bool has_body = t / G < op.body_count;
  if (!FLAT && !has_body) return;  // whole groups leave together (G divides the wave size)
  // (FLAT: lanes without a body still stream entries and meet the workgroup's barriers; b is clamped for them)
  const size_t b = op.body_first + (has_body ? t / G : op.body_count - 1);
  // FLAT: the tile's share [E0, E1) of the compact arrays (uniform over the workgroup: two scalar loads) is asked for
  // HERE, together with the solver state below -- round 3 read `done` first, alone, and everything else behind the
  // branch on it: one full memory round trip at the head of every workgroup's chain of dependent accesses
  int32_t E0 = 0, E1 = 0;
  if constexpr (FLAT) {
    if (op.aptr != nullptr) {
      const size_t tb0r = tile * (size_t)(kBlock / G);
      const size_t tb0 = (tb0r < op.body_count) ? tb0r : op.body_count;  // (the grid is rounded up to whole XCD rounds)
      const size_t tb1 = (tb0 + kBlock / G < op.body_count) ? tb0 + kBlock / G : op.body_count;
      E0 = op.aptr[op.body_first + tb0];
      E1 = op.aptr[op.body_first + tb1];
    }
  }
  const double* xt = X0;
  const double* gt = G0;
  double step = 0.0;
  double* vel_new = op.vel;  // rows this sweep writes (tiered solves: the buffer of the new iterate's parity)
  if (MODE == X_SOLVE) {
    const int done = st->done;
    const bool odd = st->flips & 1u;
    step = st->step;
    if (done) return;
    if (odd) {
      xt = X1;
      gt = G1;
    }
    if (op.vel_alt) vel_new = odd ? op.vel : op.vel_alt;
  }
  const bool step_is_zero = fabs(-step) < kZeroTol;
  constexpr int HW = (KIN == KIN_RIGID) ? 6 : (KIN == KIN_ROD ? 4 : 3);
  // force and torque sums in double-double: their rounded values do not depend on the order of the list, on G or on U
  DD3 Fdd{{0.0, 0.0}, {0.0, 0.0}, {0.0, 0.0}}, Tdd{{0.0, 0.0}, {0.0, 0.0}, {0.0, 0.0}};
  const int32_t beg = op.inc_ptr[b], end = op.inc_ptr[b + 1];
  // the body's own constants, needed only after the sums: fetched now, by the lane that will use them, so that they are
  // not one more dependent memory level at the end of the chain
  double mt = 0.0, mr = 0.0;
  V3 axis{0.0, 0.0, 0.0};
  // drift bookkeeping of tiered solves: how far this body's row moves in this sweep.  Round 4: taken from the ROWS --
  // the row of the previous iterate is still there when the new one is written (the other buffer of the ping-pong, or
  // the row about to be overwritten), so |dU|_1, |dZ|_1 are differences of two rows; rounds 2-3 accumulated
  // F_new - F_old = sum +/- (lam - x_old) n beside the sums, which cost the sweep 18 VGPRs, a fourth LDS plane and the
  // cross products of phase B.  A bound only: its value never reaches an iterate.
  // (a template parameter: carried as a run-time flag the bookkeeping cost the untracked sweep 6 % in registers)
  constexpr bool track = TRACK != 0 && MODE == X_SOLVE && PACKED;
  constexpr bool track_regs = kRegTrack;             // TRACK == 2
  constexpr bool track_rows = track && !track_regs;  // TRACK == 1
  V3 dF{0.0, 0.0, 0.0}, dS{0.0, 0.0, 0.0};           // (TRACK == 2) F_new - F_old, S_new - S_old: plain sums, a bound only
  // With 3 x 256 entries per chunk the words the END of the sweep needs (drift, firing threshold) are fetched now, beside
  // the others, instead of costing one more memory round trip in the life of every workgroup (round 3, when the 36 KB
  // image capped the sweep at four workgroups per CU and 128 VGPRs were free; round 4: 24 KB image and 96 VGPRs -- five
  // workgroups -- with these words still in); with 2 x 256 entries they stay where they are used.
  constexpr bool kEarlyTail = FLAT && FLATP >= MHIP_KBODY_EARLY_FROM && track;
  constexpr bool kEarlySnap = FLAT && FLATP >= MHIP_KBODY_EARLY_FROM;
  double drift_old = 0.0, fire_thr = 0.0;
  unsigned long long snap_early = 0ull;
  if (kEarlySnap) {
    // (an UNCONDITIONAL load from a pointer chosen on the scalar side: inside a branch the compiler complemented the
    // word on the spot, i.e. waited for it -- a full memory round trip in front of everything else.  Without a
    // snapshot the word read is never used; the body rows are always there to be read.)
    const unsigned long long* src = (op.aptr != nullptr && op.body_mask != nullptr)
                                        ? op.snap_mask
                                        : reinterpret_cast<const unsigned long long*>(op.vel);
    snap_early = src[b];
  }
  unsigned long long mask_early = 0ull;  // (the body's activity mask: used after the flat stream, asked for now)
  if (kEarlySnap) {
    const unsigned long long* src = (op.body_mask != nullptr) ? op.body_mask
                                                              : reinterpret_cast<const unsigned long long*>(op.vel);
    mask_early = src[b];
  }
  // tracked sweeps: the row of the previous iterate (the drift is the difference of the two rows), with the early words
  double2 o0 = make_double2(0.0, 0.0), o1 = o0, o2 = o0;
  if (sub == 0) {
    mt = op.mt[b];
    if (KIN != KIN_TRANS) mr = op.mr[b];
    if (KIN == KIN_ROD) axis = load3(op.axis, b);
    if (kEarlyTail) {
      drift_old = op.drift[b];
      if (op.fire_at != nullptr) fire_thr = op.fire_at[b];
      if (track_rows) {
        const double* vel_old = op.vel_alt ? ((vel_new == op.vel) ? op.vel_alt : op.vel) : op.vel;
        const double2* vo = reinterpret_cast<const double2*>(vel_old + 6 * b);
        o0 = vo[0];
        o1 = vo[1];
        o2 = vo[2];
      }
    }
  }
#if MHIP_KBODY_STASH
  // (A/B only, see MHIP_KBODY_STASH) ... parked in LDS until the end of the sweep: 64 B per body, 8 KB per workgroup
  // (not with vector arms: their 36 KB image plus the stash would leave three workgroups per CU where there are four)
  constexpr bool kStash = kEarlyTail && track_rows && KIN != KIN_RIGID;
  __shared__ double2 stash[kStash ? 4 * (kBlock / G) : 1];
  if (kStash && sub == 0) {
    double2* mine = stash + 4 * (threadIdx.x / G);
    mine[0] = o0;
    mine[1] = o1;
    mine[2] = o2;
    mine[3] = make_double2(drift_old, fire_thr);
  }
#endif
  // The sweep is a chain of dependent accesses (row pointer -> incidence entry -> iterate of that contact -> record),
  // so what it waits on is latency, not bytes: each lane keeps U independent chains in flight, every level's U loads
  // issued back to back before the first use.  kk[u] = incidence slot or -1.
  // ent / rec: the arrays kk indexes (the full lists, or the compact active lists).  EAGER: the records are fetched
  // together with the entries instead of after the iterate is known -- one dependent level less; used where nearly
  // every entry walked carries an impulse (flagged entries), not where two thirds do not (unmasked walks).
  auto process = [&](const int32_t* __restrict__ ent, const double* __restrict__ rec, const int32_t* kk,
                     const bool eager) {
    int32_t e[U];
    double lam[U], xo[U];
    double2 h0[U], h1[U], h2[U];
    auto fetch = [&](int u) {
      const size_t k = static_cast<size_t>(kk[u]);
      if (KIN == KIN_TRANS) {
        const double* H = rec + k * HW;
        h0[u] = make_double2(H[0], H[1]);
        h1[u] = make_double2(H[2], 0.0);
      } else {  // 48-byte (n, r) / 32-byte (n, s - 1/2) records, 16-byte aligned
        const double2* H2 = reinterpret_cast<const double2*>(rec + k * HW);
        h0[u] = H2[0];
        h1[u] = H2[1];
        if (KIN == KIN_RIGID) h2[u] = H2[2];
      }
    };
#pragma unroll
    for (int u = 0; u < U; ++u) {
      e[u] = (kk[u] >= 0) ? ent[kk[u]] : -1;
      h0[u] = h1[u] = h2[u] = make_double2(0.0, 0.0);
      if (eager && kk[u] >= 0) fetch(u);
    }
    // (the iterate of an empty slot is read from contact 0 and dropped: an unconditional load can be issued next to
    // the others, a conditional one waits for the one before it)
    double2 pit[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
      pit[u] = iterate_load<MODE, PACKED>(e[u] >= 0 ? static_cast<size_t>(e[u] >> 1) : 0, xt, gt);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      xo[u] = (track_regs && e[u] >= 0) ? pit[u].x : 0.0;
      lam[u] = (e[u] >= 0) ? iterate_value<MODE>(pit[u], step, step_is_zero, sp) : 0.0;
    }
    // an inactive contact (lam == 0) adds +/-0 to the sums, which leaves them bit for bit unchanged -- so (when the
    // records are not fetched eagerly) its record is never fetched (TRACK == 2: unless its multiplier just dropped to
    // zero and the drift bookkeeping wants the change)
    if (!eager) {
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (lam[u] != 0.0 || (track_regs && xo[u] != 0.0)) fetch(u);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (lam[u] == 0.0 && !(track_regs && xo[u] != 0.0)) continue;
      const V3 n{h0[u].x, h0[u].y, h1[u].x};
      if (track_regs) {  // change of this body's force (and of S) against the previous iterate
        const double dl = (e[u] & 1) ? lam[u] - xo[u] : xo[u] - lam[u];
        const V3 df{dl * n.x, dl * n.y, dl * n.z};
        dF = dF + df;
        if (KIN == KIN_ROD) dS = dS + h1[u].y * df;
        if (KIN == KIN_RIGID) dS = dS + cross(V3{h1[u].y, h2[u].x, h2[u].y}, df);  // change of the torque
      }
      if (lam[u] == 0.0) continue;
      V3 f{lam[u] * n.x, lam[u] * n.y, lam[u] * n.z};
      if (!(e[u] & 1)) f = V3{-f.x, -f.y, -f.z};  // F_src += -lam n, F_tgt += +lam n  (NgpLcp.cpp:467-472)
      dd_add(Fdd, f);
      if (KIN == KIN_RIGID) dd_add(Tdd, cross(V3{h1[u].y, h2[u].x, h2[u].y}, f));  // torque r x (+/- lam n)
      if (KIN == KIN_ROD) dd_add(Tdd, h1[u].y * f);  // S = sum coef f; the torque is u x S, formed once per body below
    }
  };
  // activity masks apply when Proj(0 - step g) == 0 is guaranteed for g >= 0: LCP space, finite non-negative step
  const bool masked = PACKED && MODE == X_SOLVE && op.body_mask != nullptr && sp.kind == MHIP_SPACE_LOWER_BOUND &&
                      sp.lo == 0.0 && step >= 0.0 && step <= 1.7976931348623157e308;
  int32_t full_from = beg;  // incidence slots from here on are walked unconditionally
  if (masked) {
    // (the compact lists first: nothing in there waits for the body's own rows -- masks, row pointers -- loaded above)
    if (op.aptr != nullptr) {  // the snapshot's active entries, streamed; what became active since stays in mm
      const int32_t ab = op.aptr[b], ae = op.aptr[b + 1];
      if constexpr (FLAT) {
        constexpr int kChunk = FLATP * kBlock;
        constexpr bool NTS = (MHIP_NT & 1) != 0;
        double2* const pl0 = flat_img;                               // (f.x, f.y)
        double2* const pl1 = flat_img + kChunk;                      // (f.z, coef)  |  (f.z, (r x f).x)
        double2* const pl2 = flat_img + (kFlatPlanes - 1) * kChunk;  // vector arms: ((r x f).y, (r x f).z)
        for (int32_t base = E0; base < E1; base += kChunk) {
          // phase A: FLATP entries per lane, all their loads in flight together
          int32_t fe[FLATP];
          double2 r0[FLATP], r1[FLATP], r2[FLATP];
          // (the entries of all passes first: the gathers wait for them alone while the records are still on their way;
          // a slot past the end of the range reads the range's last entry and is dropped -- unconditional loads keep the
          // code straight-line, so the wait counts are exact)
          bool live[FLATP];
#pragma unroll
          for (int p = 0; p < FLATP; ++p) {
            const int32_t k = base + p * kBlock + static_cast<int32_t>(threadIdx.x);
            live[p] = k < E1;
            fe[p] = ld_s<NTS>(op.aent + (live[p] ? k : E1 - 1));
          }
#pragma unroll
          for (int p = 0; p < FLATP; ++p) {
            const int32_t k0 = base + p * kBlock + static_cast<int32_t>(threadIdx.x);
            const size_t k = static_cast<size_t>(live[p] ? k0 : E1 - 1);
            r2[p] = make_double2(0.0, 0.0);
            if (KIN == KIN_TRANS) {
              const double* H = op.arec + k * HW;
              r0[p] = make_double2(ld_s<NTS>(H), ld_s<NTS>(H + 1));
              r1[p] = make_double2(ld_s<NTS>(H + 2), 0.0);
            } else {
              const double2* H2 = reinterpret_cast<const double2*>(op.arec + k * HW);
              r0[p] = ld_s<NTS>(H2);
              r1[p] = ld_s<NTS>(H2 + 1);
              if (KIN == KIN_RIGID) r2[p] = ld_s<NTS>(H2 + 2);
            }
          }
#pragma unroll
          for (int p = 0; p < FLATP; ++p)
            if (!live[p]) fe[p] = -1;
          double2 pit[FLATP];
#pragma unroll
          for (int p = 0; p < FLATP; ++p)
            pit[p] = iterate_load<MODE, PACKED>(fe[p] >= 0 ? static_cast<size_t>(fe[p] >> 1) : 0, xt, gt);
#ifdef MHIP_EXP_EXTRA_GATHER   // TIMING EXPERIMENT ONLY: a second gather of the same pattern (the other parity's pairs)
          double2 pit2[FLATP];
#pragma unroll
          for (int p = 0; p < FLATP; ++p)
            pit2[p] = iterate_load<MODE, PACKED>(fe[p] >= 0 ? static_cast<size_t>(fe[p] >> 1) : 0, (xt == X0) ? X1 : X0, gt);
#endif
#pragma unroll
          for (int p = 0; p < FLATP; ++p) {
            // (the iterate is USED unconditionally and the dead slot's value dropped by a select: with the use inside
            // `fe >= 0` the compiler sinks the gather into that branch and waits for it there with vmcnt(0))
            double lam_live = iterate_value<MODE>(pit[p], step, step_is_zero, sp);
#ifdef MHIP_EXP_EXTRA_GATHER
            if (pit2[p].x == 1.2345e300) lam_live = 0.0;
#endif
            const double lam = (fe[p] >= 0) ? lam_live : 0.0;
            // F_src += -lam n, F_tgt += +lam n  (NgpLcp.cpp:467-472): the products the per-body chains form, formed
            // here by the lane that holds the record (the same multiplications: the same bits)
            const double sl = (fe[p] & 1) ? lam : -lam;
            const int slot = p * kBlock + static_cast<int>(threadIdx.x);
            if constexpr (track_regs) {  // the image of rounds 2-3: (n, arm, +/-lam, +/-dlam)
              const double xo = (fe[p] >= 0) ? pit[p].x : 0.0;
              pl0[slot] = r0[p];
              pl1[slot] = r1[p];
              if (KIN == KIN_RIGID) flat_img[2 * kChunk + slot] = r2[p];
              pl2[slot] = make_double2(sl, (fe[p] & 1) ? lam - xo : xo - lam);
              continue;
            }
            const V3 f{sl * r0[p].x, sl * r0[p].y, sl * r1[p].x};
            pl0[slot] = make_double2(f.x, f.y);
            if (KIN == KIN_RIGID) {
              const V3 tq = cross(V3{r1[p].y, r2[p].x, r2[p].y}, f);  // torque r x (+/- lam n)
              pl1[slot] = make_double2(f.z, tq.x);
              pl2[slot] = make_double2(tq.y, tq.z);
            } else {
              pl1[slot] = make_double2(f.z, r1[p].y);  // rods: the arclength coefficient (spheres: 0)
            }
          }
          __syncthreads();
          // phase B: a body's lanes add up its slice of the image (an entry whose multiplier is zero holds +/-0: adding
          // it leaves every sum as it is, bit for bit)
          if (has_body) {
            const int32_t lo = (ab > base) ? ab : base;
            const int32_t hi = (ae < base + kChunk) ? ae : base + kChunk;
            for (int32_t k = lo + sub; k < hi; k += G) {
              const int slot = k - base;
              const double2 a0 = pl0[slot], a1 = pl1[slot];
              if constexpr (track_regs) {
                const double2 a3 = pl2[slot];   // (+/-lam, +/-dlam)
                const V3 n{a0.x, a0.y, a1.x};
                V3 arm{0.0, 0.0, 0.0};
                if (KIN == KIN_RIGID) {
                  const double2 a2 = flat_img[2 * kChunk + slot];
                  arm = V3{a1.y, a2.x, a2.y};
                }
                if (a3.y != 0.0) {
                  const V3 df{a3.y * n.x, a3.y * n.y, a3.y * n.z};
                  dF = dF + df;
                  if (KIN == KIN_ROD) dS = dS + a1.y * df;
                  if (KIN == KIN_RIGID) dS = dS + cross(arm, df);
                }
                if (a3.x == 0.0) continue;
                const V3 fr{a3.x * n.x, a3.x * n.y, a3.x * n.z};
                dd_add(Fdd, fr);
                if (KIN == KIN_RIGID) dd_add(Tdd, cross(arm, fr));
                if (KIN == KIN_ROD) dd_add(Tdd, a1.y * fr);
                continue;
              }
              const V3 f{a0.x, a0.y, a1.x};
#ifdef MHIP_EXP_PLAIN_SUMS   // TIMING EXPERIMENT ONLY (what the double-double sums of phase B cost): plain sums
              Fdd.x.hi += f.x; Fdd.y.hi += f.y; Fdd.z.hi += f.z;
              if (KIN == KIN_ROD) { Tdd.x.hi += a1.y * f.x; Tdd.y.hi += a1.y * f.y; Tdd.z.hi += a1.y * f.z; }
              if (KIN == KIN_RIGID) { const double2 a2 = pl2[slot]; Tdd.x.hi += a1.y; Tdd.y.hi += a2.x; Tdd.z.hi += a2.y; }
#else
              dd_add(Fdd, f);
              if (KIN == KIN_RIGID) {
                const double2 a2 = pl2[slot];
                dd_add(Tdd, V3{a1.y, a2.x, a2.y});
              }
              if (KIN == KIN_ROD) dd_add(Tdd, a1.y * f);  // S = sum coef f
#endif
            }
          }
          __syncthreads();
        }
      } else {
        for (int32_t k0 = ab + sub; k0 < ae; k0 += G * U) {
          int32_t kc[U];
#pragma unroll
          for (int u = 0; u < U; ++u) kc[u] = (k0 + u * G < ae) ? k0 + u * G : -1;
          process(op.aent, op.arec, kc, true);
        }
      }
    }
    const int32_t head = (end - beg < 64) ? end - beg : 64;
    unsigned long long mm = kEarlySnap ? mask_early : op.body_mask[b];
    if (head < 64) mm &= (1ull << head) - 1ull;
    if (op.aptr != nullptr) mm &= ~(kEarlySnap ? snap_early : op.snap_mask[b]);
    if (FLAT && !has_body) mm = 0ull;
#ifdef MHIP_EXP_COUNT_MM   // DIAGNOSTIC BUILD ONLY: how many waves / lanes / entries take the per-body chains behind a snapshot
    if (op.aptr != nullptr && FLAT) {
      const bool any_lane = __any(mm != 0ull);
      if ((threadIdx.x & 63) == 0) {
        atomicAdd(&g_dbg[2], 1ull);
        if (any_lane) atomicAdd(&g_dbg[0], 1ull);
      }
      if (mm != 0ull) {
        atomicAdd(&g_dbg[1], 1ull);
        atomicAdd(&g_dbg[3], (unsigned long long)__popcll(mm));
      }
    }
#endif
    int32_t kk[U];
#pragma unroll
    for (int u = 0; u < U; ++u) kk[u] = -1;
    int rank = 0, nm = 0;
    while (mm) {  // the lane takes every G-th set bit (in slot order), U at a time
      const int bit = __ffsll(static_cast<long long>(mm)) - 1;
      mm &= mm - 1ull;
      if ((rank++ & (G - 1)) != sub) continue;
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (nm == u) kk[u] = beg + bit;
      if (++nm == U) {
        process(op.inc, op.half, kk, true);
        nm = 0;
#pragma unroll
        for (int u = 0; u < U; ++u) kk[u] = -1;
      }
    }
    if (nm) process(op.inc, op.half, kk, true);
    full_from = beg + head;
  }
  if (FLAT && !has_body) return;  // (no barrier follows; whole groups leave together)
  for (int32_t k0 = full_from + sub; k0 < end; k0 += G * U) {
    int32_t kk[U];
#pragma unroll
    for (int u = 0; u < U; ++u) kk[u] = (k0 + u * G < end) ? k0 + u * G : -1;
    process(op.inc, op.half, kk, false);
  }
#pragma unroll
  for (int off = G / 2; off > 0; off >>= 1) {
    dd_add(Fdd.x, dd_shfl_xor(Fdd.x, off));
    dd_add(Fdd.y, dd_shfl_xor(Fdd.y, off));
    dd_add(Fdd.z, dd_shfl_xor(Fdd.z, off));
    if (KIN != KIN_TRANS) {
      dd_add(Tdd.x, dd_shfl_xor(Tdd.x, off));
      dd_add(Tdd.y, dd_shfl_xor(Tdd.y, off));
      dd_add(Tdd.z, dd_shfl_xor(Tdd.z, off));
    }
  }
  if (track_regs) {
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) {
      dF = dF + V3{__shfl_xor(dF.x, off, 64), __shfl_xor(dF.y, off, 64), __shfl_xor(dF.z, off, 64)};
      if (KIN != KIN_TRANS) dS = dS + V3{__shfl_xor(dS.x, off, 64), __shfl_xor(dS.y, off, 64), __shfl_xor(dS.z, off, 64)};
    }
  }
  if (sub != 0) return;
#if MHIP_KBODY_STASH
  if (kStash) {
    const double2* mine = stash + 4 * (threadIdx.x / G);
    o0 = mine[0];
    o1 = mine[1];
    o2 = mine[2];
    drift_old = mine[3].x;
    fire_thr = mine[3].y;
  }
#endif
  // (where it was not fetched at the top: asked for before the last arithmetic of the sweep)
  if (track_rows && !kEarlyTail) {
    const double* vel_old = op.vel_alt ? ((vel_new == op.vel) ? op.vel_alt : op.vel) : op.vel;
    const double2* vo = reinterpret_cast<const double2*>(vel_old + 6 * b);
    o0 = vo[0];
    o1 = vo[1];
    o2 = vo[2];
  }
  const V3 F = dd_value(Fdd), T = dd_value(Tdd);  // the one rounding of each sum
  double2* v = reinterpret_cast<double2*>(vel_new + 6 * b);
  V3 W{0.0, 0.0, 0.0};
  if (KIN == KIN_RIGID) W = V3{mr * T.x, mr * T.y, mr * T.z};
  if (KIN == KIN_ROD) {
    const V3 tq = cross(axis, T);  // T holds S = sum coef f
    const V3 w{mr * tq.x, mr * tq.y, mr * tq.z};
    // (the fused / staged solvers -- the packed X_SOLVE sweeps -- need W only for the final iterate: they end with one
    // X_APPLY sweep of it, and the 24 bytes per body are not written 770 times: k_body 0.103 -> 0.100 ms at 10^6 rods)
    if (!(MODE == X_SOLVE && PACKED)) store3(op.omega, b, w);
    W = cross(w, axis);  // the row carries Z = W x u: the contact-point velocity is U + coef Z
  }
  const V3 Ub{mt * F.x, mt * F.y, mt * F.z};  // U = F / (6 pi r mu)  (NgpLcp.cpp:484-486)
  v[0] = make_double2(Ub.x, Ub.y);
  v[1] = make_double2(Ub.z, W.x);
  v[2] = make_double2(W.y, W.z);
  if (track) {
    // |change of n . (U + coef Z)| <= |dU|_1 + |coef| |dZ|_1 with |coef| <= 1/2 (rods; spheres carry no Z): what any
    // contact of this body can have moved by, times dt as the gradient sees it.  The row holds (U, Z) resp. (U, W).
    double d;
    if constexpr (track_regs) {  // dU = mt dF, dZ = (mr (u x dS)) x u; vector arms: dW = mr dT
      d = mt * (fabs(dF.x) + fabs(dF.y) + fabs(dF.z));
      if (KIN == KIN_ROD) {
        const V3 tq = cross(axis, dS);
        const V3 dZ = cross(V3{mr * tq.x, mr * tq.y, mr * tq.z}, axis);
        d += 0.5 * (fabs(dZ.x) + fabs(dZ.y) + fabs(dZ.z));
      }
      if (KIN == KIN_RIGID) d += op.arm_max[b] * mr * (fabs(dS.x) + fabs(dS.y) + fabs(dS.z));
    } else {
      d = fabs(Ub.x - o0.x) + fabs(Ub.y - o0.y) + fabs(Ub.z - o1.x);
      const double dw = fabs(W.x - o1.y) + fabs(W.y - o2.x) + fabs(W.z - o2.y);
      if (KIN == KIN_ROD) d += 0.5 * dw;
      // vector arms: the contact point at arm r moves by dU + dW x r, |dW x r| <= |dW|_1 |r|
      if (KIN == KIN_RIGID) d += op.arm_max[b] * dw;
    }
    const double D = (kEarlyTail ? drift_old : op.drift[b]) + op.dt * d;
    op.drift[b] = D;
    if (op.fire_at != nullptr && !(D < (kEarlyTail ? fire_thr : op.fire_at[b])))  // one of its sleeping contacts has used up its share of slack
      op.fired[atomicAdd(&op.tier_counters[1], 1ull)] = static_cast<int32_t>(b);
  }
}

// one reduction record = kRed doubles (max residual term; sum dx^2 and sum dx dg as double-double pairs), stored as
// planes `stride` apart
constexpr int kRed = MHIP_BBPGD_REDUCTION_WIDTH;
static_assert(kRed == 5, "record layout: max, num.hi, num.lo, den.hi, den.lo");
__device__ inline void store_partial(double* __restrict__ p, size_t stride, size_t slot, double rmax, DD num, DD den) {
  p[slot] = rmax;
  p[stride + slot] = num.hi;
  p[2 * stride + slot] = num.lo;
  p[3 * stride + slot] = den.hi;
  p[4 * stride + slot] = den.lo;
}

// dt * sdot of contact c from the body rows `vel`: sdot = -n . (v_src - v_tgt) at the contact points (NgpLcp.cpp:526-528)
template <int KIN>
__device__ inline double contact_dt_sdot(const OpView& op, const double* __restrict__ vel, size_t c, int2 ij) {
  const V3 n = load3(op.normal, c);
  const double2* vi2 = reinterpret_cast<const double2*>(vel + 6 * (size_t)ij.x);
  const double2* vj2 = reinterpret_cast<const double2*>(vel + 6 * (size_t)ij.y);
  const double2 a0 = vi2[0], a1 = vi2[1], b0 = vj2[0], b1 = vj2[1];
  V3 vi{a0.x, a0.y, a1.x}, vj{b0.x, b0.y, b1.x};
  if (KIN == KIN_RIGID) {
    const double2 a2 = vi2[2], b2 = vj2[2];
    vi = vi + cross(V3{a1.y, a2.x, a2.y}, load3(op.ra, c));
    vj = vj + cross(V3{b1.y, b2.x, b2.y}, load3(op.rb, c));
  }
  if (KIN == KIN_ROD) {
    const double2 a2 = vi2[2], b2 = vj2[2];
    const double ci = rod_arm_coef(op.arc_s[c]), cj = rod_arm_coef(op.arc_t[c]);
    vi = vi + ci * V3{a1.y, a2.x, a2.y};
    vj = vj + cj * V3{b1.y, b2.x, b2.y};
  }
  const double sdot = -n.x * (vi.x - vj.x) - n.y * (vi.y - vj.y) - n.z * (vi.z - vj.z);
  return op.dt * sdot;
}

// Cold tail of a tiered solve ("Cold tier" below).  A sleeping contact (i, j) holds two thresholds, thr[0] for drift[i]
// and thr[1] for drift[j] (each body gets half the contact's slack); fire_at[b] is the smallest threshold among b's
// sleeping contacts.  The body sweep lists the bodies whose drift has reached fire_at.  The first `blocks` workgroups
// of the constraint sweep's grid ("service" workgroups) do the tail's work in the shadow of the hot sweep: they evaluate
// the contacts that were awake when the iteration started (list[0, counters[2])), then walk the fired bodies'
// incidence lists, wake the contacts whose threshold is reached -- listed, and evaluated on the spot by the lane that
// woke them -- and reset fire_at to the smallest threshold left.  A handful of bodies per iteration: nothing scans the
// tail, and the tail costs no launch of its own.
struct TierCheck {
  size_t H, I;                   // cold tail = contacts [H, I)
  double* wake;                  // [C - H][2] thresholds; thr[0] = -inf: awake
  int32_t* list;                 // awake contacts of the tail
  unsigned long long* counters;  // [0] = length of list, [1] = length of fired, [2] = length of list when the iteration began
  const int32_t* fired;          // bodies listed by the body sweep
  double* fire_at;               // [N]
  unsigned blocks;               // workgroups at the front of the grid that do this instead of sweeping
};

// one contact of the tail evaluated as the sweep evaluates a hot one (same expressions: same bits); a sleeper's stale
// pair says x = 0, g > 0 -- all an evaluation uses of an inactive contact
template <int KIN>
__device__ __forceinline__ void tier_evaluate(const OpView& op, const double* __restrict__ vel,
                                              const double* __restrict__ xt, double* __restrict__ xn,
                                              const double* __restrict__ q, const Space& sp, double step,
                                              bool step_is_zero, int resid_kind, size_t c, double& rmax, DD& num,
                                              DD& den) {
  const int2 ij = op.pairs[c];
  double x_old = 0.0, g_old = 0.0;
  const double xc = iterate_x<X_SOLVE, true>(c, xt, nullptr, step, step_is_zero, sp, &x_old, &g_old);
  const double y = contact_dt_sdot<KIN>(op, vel, c, ij);
  const double g = 1.0 * q[c] + 1.0 * y;  // axpby(1, q, 1, grad)  (convex.hpp:623, :651)
  reinterpret_cast<double2*>(xn)[c] = make_double2(xc, g);
  if (op.body_mask != nullptr && sp.kind == MHIP_SPACE_LOWER_BOUND && sp.lo == 0.0) {
    const bool was = !(x_old == 0.0 && g_old >= 0.0 && g_old <= 1.7976931348623157e308);
    const bool now = !(xc == 0.0 && g >= 0.0 && g <= 1.7976931348623157e308);
    if (was != now) {
      const unsigned pi = op.pos[2 * c], pj = op.pos[2 * c + 1];
      if (pi < 64u) atomicXor(&op.body_mask[ij.x], 1ull << pi);
      if (pj < 64u) atomicXor(&op.body_mask[ij.y], 1ull << pj);
    }
  }
  if (op.counted == nullptr || op.counted[c]) {
    const double r = residual_term(resid_kind, xc, g, sp);
    if (r > rmax) rmax = r;
    const double dx = xc - x_old;
    dd_add(num, dx * dx);            // diff_dot(x, x_old)              (convex.hpp:507)
    dd_add(den, dx * (g - g_old));   // diff_dot(x, x_old, g, g_old)    (convex.hpp:508)
  }
}

// what a service workgroup does (block = its index among the tc.blocks of them); its partial record goes to `slot`.
// Waking and evaluating alternate in rounds -- a lane wakes at most kTierRoundClaims contacts into the workgroup's LDS
// queue, then the workgroup evaluates the queue -- so that the walk of the incidence lists and the evaluation do not
// hold their registers at the same time (evaluating on the spot took 82 VGPRs against the sweep's 66).
constexpr int kTierRoundClaims = 8;
constexpr int kTierQueue = kTierRoundClaims * kBlock;
template <int KIN>
__device__ __forceinline__ void tier_service(const OpView& op, const SolverState* __restrict__ st, double* __restrict__ X0,
                                             double* __restrict__ X1, const double* __restrict__ q, const Space& sp,
                                             int resid_kind, double* __restrict__ partials, const TierCheck& tc,
                                             unsigned block, size_t slot, double* scratch, int32_t* queue,
                                             unsigned* queued) {
  if (st->done) return;
  const double* xt = X0;
  double* xn = X1;
  if (st->flips & 1u) {
    xt = X1; xn = X0;
  }
  const double step = st->step;
  const double* vel = op.vel;  // the rows the body sweep of this iteration wrote
  if (op.vel_alt && !(st->flips & 1u)) vel = op.vel_alt;
  const bool step_is_zero = fabs(-step) < kZeroTol;
  double rmax = 0.0;  // (sleeping contacts add exact zeros, among them 0 to the max)
  DD num{0.0, 0.0}, den{0.0, 0.0};
  // this workgroup's share of the contacts woken in earlier iterations: evaluated in the first round below
  const size_t awake = static_cast<size_t>(tc.counters[2]);
  const size_t share = (awake + tc.blocks - 1) / tc.blocks;
  const size_t old_lo = block * share < awake ? block * share : awake;
  unsigned n_old = static_cast<unsigned>((old_lo + share < awake ? old_lo + share : awake) - old_lo);
  // the bodies that fired in this iteration's body sweep: one per lane, a chunk of blockDim at a time
  const size_t count = static_cast<size_t>(tc.counters[1]);
  const double ninf = -__builtin_huge_val();
  size_t f0 = block * (size_t)blockDim.x;
  for (;;) {
    const bool have = f0 + threadIdx.x < count;
    const size_t b = have ? static_cast<size_t>(tc.fired[f0 + threadIdx.x]) : 0;
    const double D = have ? op.drift[b] : 0.0;
    int32_t k = have ? op.inc_ptr[b] : 0;
    const int32_t kend = have ? op.inc_ptr[b + 1] : 0;
    double left = __builtin_huge_val();
    do {
      if (threadIdx.x == 0) *queued = 0u;
      __syncthreads();
      for (int claims = 0; k < kend && claims < kTierRoundClaims; ++k) {
        const int32_t e = op.inc[k];
        const size_t c = static_cast<size_t>(e >> 1);
        if (c < tc.H || c >= tc.I) continue;
        double* thr = tc.wake + 2 * (c - tc.H);
        const double mine = thr[e & 1];
        if (!(thr[0] > ninf)) continue;  // awake already
        if (D < mine) {
          if (mine < left) left = mine;
          continue;
        }
        // wake it; the contact's other body may be doing the same right now: slot 0 decides who lists and evaluates it
        const unsigned long long old = atomicExch(reinterpret_cast<unsigned long long*>(thr),
                                                  static_cast<unsigned long long>(__double_as_longlong(ninf)));
        if (__longlong_as_double(static_cast<long long>(old)) > ninf) {
          thr[1] = ninf;
          tc.list[atomicAdd(&tc.counters[0], 1ull)] = static_cast<int32_t>(c);
          queue[atomicAdd(queued, 1u)] = static_cast<int32_t>(c);
          ++claims;
        }
      }
      __syncthreads();
      const unsigned total = n_old + *queued;
      for (unsigned i = threadIdx.x; i < total; i += blockDim.x) {
        const int32_t c = i < n_old ? tc.list[old_lo + i] : queue[i - n_old];
        tier_evaluate<KIN>(op, vel, xt, xn, q, sp, step, step_is_zero, resid_kind, static_cast<size_t>(c), rmax, num,
                           den);
      }
      n_old = 0;
    } while (__syncthreads_or(k < kend));  // (the barrier also keeps the queue reads ahead of the next reset)
    if (have) tc.fire_at[b] = left;
    f0 += (size_t)tc.blocks * blockDim.x;
    if (f0 >= count) break;
  }
  const double m = block_max(rmax, scratch);
  const DD s1 = block_sum(num, scratch);
  const DD s2 = block_sum(den, scratch);
  if (threadIdx.x == 0) store_partial(partials, op.part_stride, slot, m, s1, s2);
}

// (Kept as its own body rather than an instantiation of constraint_sweep: as one, the same sweep ran 10 % slower --
// 0.144 against 0.130 ms at 10^6 rods -- for reasons the ISA diff of the loop does not show.)
// (the waves-per-SIMD floor keeps the service path, which the compiler would otherwise schedule into 97 VGPRs, within
// the sweep's own register budget)
template <int MODE, int KIN, bool PACKED>
__global__ void __launch_bounds__(kBlock, (KIN == KIN_RIGID ? 6 : 7))
    k_constraint(OpView op, const SolverState* __restrict__ st, double* __restrict__ X0, double* __restrict__ X1,
                 double* __restrict__ G0, double* __restrict__ G1, const double* __restrict__ q, Space sp,
                 int resid_kind, double* __restrict__ partials, TierCheck check = TierCheck{}) {
  __shared__ double scratch[2 * kBlock / 64];
  // (tiered solves) the first check.blocks workgroups of the grid serve the cold tail; their partial records follow
  // the sweeping workgroups' (the launch sets part_stride)
  const unsigned nblk = gridDim.x - check.blocks;
  if (blockIdx.x < check.blocks) {
    if (MODE == X_SOLVE && PACKED) {
      __shared__ int32_t queue[kTierQueue];
      __shared__ unsigned queued;
      tier_service<KIN>(op, st, X0, X1, q, sp, resid_kind, partials, check, blockIdx.x,
                        op.part_offset + nblk + blockIdx.x, scratch, queue, &queued);
    }
    return;
  }
  const unsigned bid = blockIdx.x - check.blocks;  // this workgroup's place among the sweeping ones
  const double* xt = X0;
  const double* gt = G0;
  double* xn = X1;
  double* gn = G1;
  double step = 0.0;
  if (MODE == X_SOLVE) {
    if (st->done) return;
    if (st->flips & 1u) {
      xt = X1; gt = G1; xn = X0; gn = G0;
    }
    step = st->step;
  }
  if (MODE == X_INIT) gn = G0;  // g_tmp = A x_tmp + q
  if (MODE == X_INIT && PACKED) {  // X0 = packed buffer of the first iterate, G0 = the caller's plain x
    xt = G0;
    xn = X0;
  }
  const double* vel = op.vel;  // the rows the body sweep of this iteration wrote
  if (MODE == X_SOLVE && op.vel_alt && !(st->flips & 1u)) vel = op.vel_alt;
  const bool step_is_zero = fabs(-step) < kZeroTol;
  double rmax = kLowest;
  DD num{0.0, 0.0}, den{0.0, 0.0};
  const size_t ntiles = (op.c_end - op.c_first + kBlock - 1) / kBlock;
  for (size_t lin = bid; lin < ntiles; lin += nblk) {
    const size_t c = op.c_first + xcd_tile(lin, ntiles, op.xcd_aware) * kBlock + threadIdx.x;
    if (c >= op.c_end) continue;
    constexpr bool NTC = (MHIP_NT & 2) != 0 && MODE == X_SOLVE && PACKED;
    // every load that depends on nothing but c first (the ISA of the round-2 form fetched the second arclength and q
    // each behind a wait of its own: two more memory round trips in the life of every workgroup), then the gathers of
    // the two body rows, which depend on the pair
    const int2 ij = ld_s<NTC>(op.pairs + c);
    const double2 pold = iterate_load<MODE, PACKED, NTC>(c, xt, gt);
    const V3 n = ld_s3<NTC>(op.normal, c);
    double arc_i = 0.0, arc_j = 0.0;
    V3 arm_i{0.0, 0.0, 0.0}, arm_j{0.0, 0.0, 0.0};
    if (KIN == KIN_ROD) {
      arc_i = ld_s<NTC>(op.arc_s + c);
      arc_j = ld_s<NTC>(op.arc_t + c);
    }
    if (KIN == KIN_RIGID) {
      arm_i = ld_s3<NTC>(op.ra, c);
      arm_j = ld_s3<NTC>(op.rb, c);
    }
    const double qc = (MODE != X_APPLY) ? ld_s<NTC>(q + c) : 0.0;
    const double2* vi2 = reinterpret_cast<const double2*>(vel + 6 * (size_t)ij.x);
    const double2* vj2 = reinterpret_cast<const double2*>(vel + 6 * (size_t)ij.y);
    const double2 a0 = vi2[0], a1 = vi2[1], b0 = vj2[0], b1 = vj2[1];
    double2 a2 = make_double2(0.0, 0.0), b2 = make_double2(0.0, 0.0);
    if (KIN != KIN_TRANS) {
      a2 = vi2[2];
      b2 = vj2[2];
    }
    const double x_old = (MODE == X_SOLVE) ? pold.x : 0.0, g_old = (MODE == X_SOLVE) ? pold.y : 0.0;
    const double xc = iterate_value<MODE>(pold, step, step_is_zero, sp);
    V3 vi{a0.x, a0.y, a1.x}, vj{b0.x, b0.y, b1.x};
    if (KIN == KIN_RIGID) {
      vi = vi + cross(V3{a1.y, a2.x, a2.y}, arm_i);
      vj = vj + cross(V3{b1.y, b2.x, b2.y}, arm_j);
    }
    if (KIN == KIN_ROD) {
      const double ci = rod_arm_coef(arc_i), cj = rod_arm_coef(arc_j);
      vi = vi + ci * V3{a1.y, a2.x, a2.y};
      vj = vj + cj * V3{b1.y, b2.x, b2.y};
    }
    // sdot = -n . (v_src - v_tgt)  (NgpLcp.cpp:526-528)
    const double sdot = -n.x * (vi.x - vj.x) - n.y * (vi.y - vj.y) - n.z * (vi.z - vj.z);
    const double y = op.dt * sdot;
    if (MODE == X_APPLY) {
      gn[c] = y;
    } else {
      const double g = 1.0 * qc + 1.0 * y;  // axpby(1, q, 1, grad)  (convex.hpp:623, :651)
      if (PACKED) {
        reinterpret_cast<double2*>(xn)[c] = make_double2(xc, g);
        if (op.body_mask != nullptr && sp.kind == MHIP_SPACE_LOWER_BOUND && sp.lo == 0.0) {
          // the masks start all-ZERO and the init sweep sets the bits of the contacts that are active in the first
          // iterate -- a third of them (rounds 1-3 started from all-ones and cleared the other two thirds: twice the
          // atomics, 0.49 ms of every step's init sweep at 10^6 rods); from then on a contact flips its two bits when
          // its state changes
          const bool was = (MODE == X_INIT) ? false : !(x_old == 0.0 && g_old >= 0.0 && g_old <= 1.7976931348623157e308);
          const bool now = !(xc == 0.0 && g >= 0.0 && g <= 1.7976931348623157e308);
          if (was != now) {
            const unsigned pi = op.pos[2 * c], pj = op.pos[2 * c + 1];
            if (pi < 64u) atomicXor(&op.body_mask[ij.x], 1ull << pi);
            if (pj < 64u) atomicXor(&op.body_mask[ij.y], 1ull << pj);
          }
        }
      } else {
        gn[c] = g;
        if (MODE == X_SOLVE) xn[c] = xc;
      }
      if (op.counted == nullptr || op.counted[c]) {
        const double r = residual_term(resid_kind, xc, g, sp);
        if (r > rmax) rmax = r;
        if (MODE == X_SOLVE) {
          const double dx = xc - x_old;
          dd_add(num, dx * dx);            // diff_dot(x, x_old)              (convex.hpp:507)
          dd_add(den, dx * (g - g_old));   // diff_dot(x, x_old, g, g_old)    (convex.hpp:508)
        }
      }
    }
  }
  if (MODE != X_APPLY) {
    const double m = block_max(rmax, scratch);
    const DD s1 = block_sum(num, scratch);
    const DD s2 = block_sum(den, scratch);
    if (threadIdx.x == 0) {  // kRed planes of values: the final pass reads them coalesced
      const size_t stride = op.part_stride ? op.part_stride : nblk;
      store_partial(partials, stride, op.part_offset + bid, m, s1, s2);
    }
  }
}

constexpr int kFinalBlock = 1024;  // threads of the single-workgroup final passes
// ordered reduction of nparts (max, num, den) records of kRed doubles; element (i, k) sits at partials[i * si + k * sk]
// (block partials: si = 1, sk = plane distance; the all-gathered per-rank records: si = kRed, sk = 1)
__device__ inline void reduce_records(int nparts, const double* __restrict__ partials, size_t si, size_t sk,
                                      double* scratch, double& rmax, DD& num, DD& den) {
  rmax = kLowest;
  num = DD{0.0, 0.0};
  den = DD{0.0, 0.0};
  for (int i = threadIdx.x; i < nparts; i += blockDim.x) {
    const double* r = partials + i * si;
    if (r[0] > rmax) rmax = r[0];
    dd_add(num, DD{r[sk], r[2 * sk]});
    dd_add(den, DD{r[3 * sk], r[4 * sk]});
  }
  rmax = block_max(rmax, scratch);
  num = block_sum(num, scratch);
  den = block_sum(den, scratch);
}

// first level of the final reduction when there are tens of thousands of block partials: workgroup b folds the
// contiguous slice b of each plane into one record (fixed slices, fixed order: deterministic)
constexpr int kFoldGroups = 64;
__global__ void __launch_bounds__(kBlock) k_fold_partials(int nparts, size_t stride,
                                                         const double* __restrict__ partials,
                                                         const SolverState* __restrict__ st, int check_done,
                                                         double* __restrict__ folded) {
  __shared__ double scratch[2 * kBlock / 64];
  if (check_done && st->done) return;
  const int per = (nparts + gridDim.x - 1) / gridDim.x;
  const int lo = blockIdx.x * per, hi = (lo + per < nparts) ? lo + per : nparts;
  double rmax;
  DD num, den;
  reduce_records(hi > lo ? hi - lo : 0, partials + lo, 1, stride, scratch, rmax, num, den);
  if (threadIdx.x == 0) store_partial(folded, gridDim.x, blockIdx.x, rmax, num, den);
}

// tiered: the solve keeps contacts in a cold tier whose iterates are only known to be "x = 0, g > 0"; a step outside
// [0, finite] would need their exact gradients, so the solve is paused (done = 2) for the host to leave the tiers first.
template <int MODE>
__device__ inline void finalize_state(SolverState* __restrict__ st, double rmax, DD numdd, DD dendd, int resid_kind,
                                      double tol, unsigned max_iters, int tiered,
                                      unsigned long long* __restrict__ tier_counters);
template <int MODE>
__global__ void __launch_bounds__(kFinalBlock) k_finalize(int nparts, const double* __restrict__ partials, size_t si,
                                                         size_t sk, SolverState* __restrict__ st, int resid_kind,
                                                         double tol, unsigned max_iters, int tiered = 0,
                                                         unsigned long long* __restrict__ tier_counters = nullptr) {
  __shared__ double scratch[2 * kFinalBlock / 64];
  if (MODE == X_SOLVE && st->done) return;
  double rmax;
  DD numdd, dendd;
  reduce_records(nparts, partials, si, sk, scratch, rmax, numdd, dendd);
  if (threadIdx.x != 0) return;
  finalize_state<MODE>(st, rmax, numdd, dendd, resid_kind, tol, max_iters, tiered, tier_counters);
}
template <int MODE>
__device__ inline void finalize_state(SolverState* __restrict__ st, double rmax, DD numdd, DD dendd, int resid_kind,
                                      double tol, unsigned max_iters, int tiered,
                                      unsigned long long* __restrict__ tier_counters) {
  if (tier_counters) {
    tier_counters[1] = 0;                 // the fired bodies of this iteration have been dealt with
    tier_counters[2] = tier_counters[0];  // the contacts awake when the next iteration begins
  }
  const double num = dd_value(numdd);  // the BB dot products, each rounded once
  double den = dd_value(dendd);
  const double res = (resid_kind == MHIP_RESIDUAL_PROJECTED_DIFF) ? rmax / kSmallStep : rmax;
  st->residual = res;
  if (MODE == X_INIT) {
    st->step = 1.0 / res;  // Dai-Fletcher initial step (convex.hpp:626-627)
    st->iter = 0;
    st->flips = 0;
    st->converged = (res <= tol) ? 1 : 0;
    st->converged_at_init = st->converged;
    st->done = (st->converged || max_iters == 0) ? 1 : 0;
    st->num = 0.0;
    st->den = 0.0;
  } else {
    if (res <= tol) {
      st->converged = 1;
      st->done = 1;
      return;
    }
    den += kBBEps * (fabs(den) < kBBEps ? 1.0 : 0.0);  // convex.hpp:511-512
    st->num = num;
    st->den = den;
    st->step = num / den;
    st->iter += 1;
    st->flips += 1;
    if (st->iter >= max_iters) st->done = 1;
    else if (tiered == 2 || (tiered && !(st->step >= 0.0 && st->step <= 1.7976931348623157e308))) st->done = 2;
  }
}

// Fold and finalize in ONE launch (the fused solve): workgroup b folds slice b of the block partials as k_fold_partials
// does, publishes its record with write-through stores, takes a ticket, and the workgroup whose ticket is the last
// finalizes from all the records.  No fence that writes back or invalidates an L2 anywhere (MI355X_MICROARCH.md,
// hand-offs with sc1 stores / loads): the records are stored with agent-scope relaxed atomics (global_store ... sc1: they
// leave the XCD's L2 as they are written), the storing lane drains them (s_waitcnt vmcnt(0)) before its agent-scope
// ticket add, and the last arriver reads them with agent-scope relaxed atomic loads (sc1: served past its L1) only
// after its add has returned.  Round 1 had tried the ticket with a release / acquire fence pair (3.4 us, as long as the
// launch it saved); as two launches fold + finalize took 4.8 + 4.8 us of every iteration.
template <int MODE>
__global__ void __launch_bounds__(kBlock)
    k_fold_finalize(int nparts, size_t stride, const double* __restrict__ partials, double* __restrict__ folded,
                    unsigned* __restrict__ ticket, SolverState* __restrict__ st, int resid_kind, double tol,
                    unsigned max_iters, int tiered, unsigned long long* __restrict__ tier_counters,
                    MailboxArgs mb = MailboxArgs{}) {
  __shared__ double scratch[2 * kBlock / 64];
  __shared__ int is_last;
  const int groups = static_cast<int>(gridDim.x);
  const int per = (nparts + groups - 1) / groups;
  const int lo = blockIdx.x * per, hi = (lo + per < nparts) ? lo + per : nparts;
  double rmax;
  DD num, den;
  reduce_records(hi > lo ? hi - lo : 0, partials + lo, 1, stride, scratch, rmax, num, den);
  // (after the loads above: one round trip, not two.)  ONE lane's reading decides for the workgroup, through LDS: the
  // last arriver of THIS launch may set `done` while a late wave of another workgroup has yet to look at it, and a
  // wave that returned on its own reading would leave its workgroup's barrier short of a wave
  const unsigned done_seen = (MODE == X_SOLVE) ? st->done : 0u;
  if (threadIdx.x == 0) is_last = 0;
  if (threadIdx.x == 0 && !done_seen) {
    const double rec[kRed] = {rmax, num.hi, num.lo, den.hi, den.lo};
#pragma unroll
    for (int k = 0; k < kRed; ++k)
      __hip_atomic_store(folded + (size_t)k * groups + blockIdx.x, rec[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    is_last = (t == static_cast<unsigned>(groups) - 1u) ? 1 : 0;
  }
  __syncthreads();
  if (!is_last) return;
  if (threadIdx.x >= 64) return;
  // the last arriver's first wave: record i by lane i (groups <= 64), the wave's reduction, one lane finalizes
  double m = kLowest;
  DD a{0.0, 0.0}, b{0.0, 0.0};
  const int i = threadIdx.x;
  if (i < groups) {
    m = __hip_atomic_load(folded + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    a.hi = __hip_atomic_load(folded + (size_t)groups + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    a.lo = __hip_atomic_load(folded + 2 * (size_t)groups + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    b.hi = __hip_atomic_load(folded + 3 * (size_t)groups + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    b.lo = __hip_atomic_load(folded + 4 * (size_t)groups + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  m = wave_max(m);
  a = wave_sum(a);
  b = wave_sum(b);
  if (mb.peers != nullptr) {
    // ranks of one node (the staged solve): what has been added up so far is THIS RANK's record; it goes to
    // everybody's mailbox, everybody's records come back into shared memory, and their sum -- formed as k_finalize
    // forms it from the all-gathered records: record r by lane r, the wave's reduction -- is what is finalized
    __shared__ double record[kRed];
    __shared__ double world_records[kMailboxMaxWorld * kRed];
    if (threadIdx.x == 0) store_partial(record, 1, 0, m, a, b);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    mailbox_exchange_wave(mb, record, world_records);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    m = kLowest;
    a = DD{0.0, 0.0};
    b = DD{0.0, 0.0};
    if (i < mb.world) {
      const double* r = world_records + (size_t)i * kRed;
      if (r[0] > m) m = r[0];
      dd_add(a, DD{r[1], r[2]});
      dd_add(b, DD{r[3], r[4]});
    }
    m = wave_max(m);
    a = wave_sum(a);
    b = wave_sum(b);
  }
  if (threadIdx.x != 0) return;
  __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // for the next launch
  finalize_state<MODE>(st, m, a, b, resid_kind, tol, max_iters, tiered, tier_counters);
}

// block partials -> one (max, num, den) record of kRed doubles (this rank's contribution to the all-gather of SURVEY
// 8e step 3; the sums travel as double-double pairs so that the rank count does not reach the rounded result)
__global__ void __launch_bounds__(kFinalBlock) k_reduce_local(int nparts, const double* __restrict__ partials,
                                                             size_t stride, const SolverState* __restrict__ st,
                                                             int check_done, double* __restrict__ out,
                                                             MailboxArgs mb = MailboxArgs{}) {
  __shared__ double scratch[2 * kFinalBlock / 64];
  if (check_done && st->done) return;
  double rmax;
  DD num, den;
  reduce_records(nparts, partials, 1, stride, scratch, rmax, num, den);
  __shared__ double record[kRed];
  if (threadIdx.x == 0) {
    store_partial(out, 1, 0, rmax, num, den);
    store_partial(record, 1, 0, rmax, num, den);
  }
  // ranks of one node: the record goes to everybody's mailbox and everybody's records are collected, in this launch
  if (mb.peers != nullptr) {
    __syncthreads();
    mailbox_exchange_wave(mb, record);
  }
}

// ---- the scrap app's BBPGD variant: resolve_collisions, scrap/lcp_spheres/NgpLcp.cpp:558-759 ----------
//   residual  = max_i ( lam_i < 1e-12 ? |min(g_i, 0)| : |g_i| )      (ComputeMaxAbsProjectedSep, :376-405)
//   converged iff residual < max_allowable_overlap (strict, :627, :671); first step 1/residual (:633)
//   step      = BB1 (xx/xg) when ite_count is even, BB2 (xg/gg) when odd; |b| < 1e-12 -> b += 1e-12 (:716-731)
//   ite_count counts started iterations, the converging one included (:636)
//   quirk kept: the first projected step uses the gradient WITHOUT the A x_0 term (signed_sep_dot is still zero at
//   :639; signed_sep_dot_tmp holds it) -- identical to the consistent form when the initial guess is zero.
template <int KIN, bool INIT>
__global__ void __launch_bounds__(kBlock)
    k_scrap_constraint(OpView op, const SolverState* __restrict__ st, double* __restrict__ X0, double* __restrict__ X1,
                       double* __restrict__ G0, double* __restrict__ G1, double* __restrict__ D0,
                       double* __restrict__ D1, const double* __restrict__ q, double* __restrict__ partials) {
  __shared__ double scratch[2 * kBlock / 64];
  const double* xt = X0;
  const double* gt = G0;
  double* xn = X1;
  double* gn = INIT ? G0 : G1;
  // dt * sep_dot of the two iterates, kept beside g = sep + dt * sep_dot: the scrap code forms gkdiff from the sep_dot
  // values themselves (dt * (sep_dot - sep_dot_tmp), :700-712), which is not (g - g_tmp) bit for bit
  const double* dt_old = D0;
  double* dn = INIT ? D0 : D1;
  double step = 0.0;
  bool first = false;
  if (!INIT) {
    if (st->done) return;
    if (st->flips & 1u) {
      xt = X1; gt = G1; xn = X0; gn = G0; dt_old = D1; dn = D0;
    }
    step = st->step;
    first = (st->iter == 0);
  }
  const Space sp{MHIP_SPACE_LOWER_BOUND, 0.0, 0.0};
  double rmax = kLowest;
  DD xx{0.0, 0.0}, xg{0.0, 0.0}, gg{0.0, 0.0};
  for (size_t c = blockIdx.x * (size_t)blockDim.x + threadIdx.x; c < op.C; c += (size_t)gridDim.x * blockDim.x) {
    const int2 ij = op.pairs[c];
    // UpdateConGammas (:532-548): max(lam_tmp - alpha * (sep + dt * sep_dot), 0)
    const double gu = first ? q[c] : gt[c];
    const double xc = INIT ? xt[c] : sp.project(xt[c] - step * gu);
    const V3 n = load3(op.normal, c);
    const double2* vi2 = reinterpret_cast<const double2*>(op.vel + 6 * (size_t)ij.x);
    const double2* vj2 = reinterpret_cast<const double2*>(op.vel + 6 * (size_t)ij.y);
    const double2 a0 = vi2[0], a1 = vi2[1], b0 = vj2[0], b1 = vj2[1];
    V3 vi{a0.x, a0.y, a1.x}, vj{b0.x, b0.y, b1.x};
    if (KIN == KIN_RIGID) {
      const double2 a2 = vi2[2], b2 = vj2[2];
      vi = vi + cross(V3{a1.y, a2.x, a2.y}, load3(op.ra, c));
      vj = vj + cross(V3{b1.y, b2.x, b2.y}, load3(op.rb, c));
    }
    if (KIN == KIN_ROD) {
      const double2 a2 = vi2[2], b2 = vj2[2];
      const double ci = rod_arm_coef(op.arc_s[c]), cj = rod_arm_coef(op.arc_t[c]);
      vi = vi + ci * V3{a1.y, a2.x, a2.y};
      vj = vj + cj * V3{b1.y, b2.x, b2.y};
    }
    const double sdot = -n.x * (vi.x - vj.x) - n.y * (vi.y - vj.y) - n.z * (vi.z - vj.z);
    const double gdt = op.dt * sdot;
    const double g = q[c] + gdt;  // sep_new = sep_old + dt * sep_dot
    gn[c] = g;
    dn[c] = gdt;
    if (!INIT) xn[c] = xc;
    const double r = (xc < 1e-12) ? fabs((g < 0.0) ? g : 0.0) : fabs(g);
    if (r > rmax) rmax = r;
    if (!INIT) {
      const double dx = xc - xt[c];
      const double dg = gdt - dt_old[c];  // dt * (sep_dot - sep_dot_tmp)
      dd_add(xx, dx * dx);
      dd_add(xg, dx * dg);
      dd_add(gg, dg * dg);
    }
  }
  const double m = block_max(rmax, scratch);
  const DD s0 = block_sum(xx, scratch), s1 = block_sum(xg, scratch), s2 = block_sum(gg, scratch);
  if (threadIdx.x == 0) {  // record of 7: max, then the three sums as double-double pairs
    double* r = partials + 7 * blockIdx.x;
    r[0] = m;
    r[1] = s0.hi; r[2] = s0.lo; r[3] = s1.hi; r[4] = s1.lo; r[5] = s2.hi; r[6] = s2.lo;
  }
}

// the first scrap iteration's body sweep must use the same (quirky) gradient as k_scrap_constraint: the host passes q
// in place of g_tmp for that one launch (see mhip_scrap_bbpgd_solve_contact).
template <bool INIT>
__global__ void __launch_bounds__(kBlock) k_scrap_finalize(int nparts, const double* __restrict__ partials,
                                                          SolverState* __restrict__ st, double tol,
                                                          unsigned max_iters) {
  __shared__ double scratch[2 * kBlock / 64];
  if (!INIT && st->done) return;
  double rmax = kLowest;
  DD sxx{0.0, 0.0}, sxg{0.0, 0.0}, sgg{0.0, 0.0};
  for (int i = threadIdx.x; i < nparts; i += blockDim.x) {
    const double* r = partials + 7 * i;
    if (r[0] > rmax) rmax = r[0];
    dd_add(sxx, DD{r[1], r[2]});
    dd_add(sxg, DD{r[3], r[4]});
    dd_add(sgg, DD{r[5], r[6]});
  }
  rmax = block_max(rmax, scratch);
  sxx = block_sum(sxx, scratch);
  sxg = block_sum(sxg, scratch);
  sgg = block_sum(sgg, scratch);
  if (threadIdx.x != 0) return;
  const double xx = dd_value(sxx), xg = dd_value(sxg), gg = dd_value(sgg);
  st->residual = rmax;
  if (INIT) {
    st->iter = 0;
    st->flips = 0;
    st->converged = (rmax < tol) ? 1 : 0;
    st->converged_at_init = st->converged;
    st->done = (st->converged || max_iters == 0) ? 1 : 0;
    st->step = 1.0 / rmax;
    return;
  }
  st->iter += 1;  // ++ite_count at the top of the loop body (:636)
  if (rmax < tol) {
    st->converged = 1;
    st->done = 1;
    return;
  }
  double a, b;
  if (st->iter % 2 == 0) {
    a = xx; b = xg;   // Barzilai-Borwein choice 1
  } else {
    a = xg; b = gg;   // choice 2
  }
  if (fabs(b) < 1e-12) b += 1e-12;
  st->num = a;
  st->den = b;
  st->step = a / b;
  st->flips += 1;
  if (st->iter >= max_iters) st->done = 1;
}

// ComputeMaxVelocity (NgpLcp.cpp:743-755): max |U| over bodies
__global__ void __launch_bounds__(kBlock) k_max_speed(size_t n, const double* __restrict__ vel,
                                                     double* __restrict__ partials) {
  __shared__ double scratch[kBlock / 64];
  double m = 0.0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const double vx = vel[6 * i], vy = vel[6 * i + 1], vz = vel[6 * i + 2];
    const double v = sqrt(vx * vx + vy * vy + vz * vz);
    if (v > m) m = v;
  }
  const double r = block_max(m, scratch);
  if (threadIdx.x == 0) partials[blockIdx.x] = r;
}

// restores the reference's post-conditions (see file header)
__global__ void __launch_bounds__(kBlock) k_finish(size_t n, const SolverState* __restrict__ st,
                                                  double* __restrict__ X0, double* __restrict__ X1,
                                                  double* __restrict__ G0, double* __restrict__ G1) {
  const bool p = st->flips & 1u;
  const int mode = st->converged_at_init ? 0 : (st->converged ? (p ? 1 : 3) : 2);
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    if (mode == 0) {
      G1[i] = G0[i];  // grad <- grad_tmp (convex.hpp:632-635)
    } else if (mode == 1) {
      const double x0 = X0[i], x1 = X1[i], g0 = G0[i], g1 = G1[i];  // final iterate sits in the *_tmp arrays: swap
      X0[i] = x1; X1[i] = x0; G0[i] = g1; G1[i] = g0;
    } else if (mode == 2) {
      if (p) {  // x == x_tmp == latest iterate after the roll-forward copies (convex.hpp:662-663)
        X0[i] = X1[i]; G0[i] = G1[i];
      } else {
        X1[i] = X0[i]; G1[i] = G0[i];
      }
    }
  }
}

// the same post-conditions from the packed ping-pong pair P0 (the "tmp" role at even parity) / P1
__global__ void __launch_bounds__(kBlock) k_finish_packed(size_t n, const SolverState* __restrict__ st,
                                                         const double2* __restrict__ P0,
                                                         const double2* __restrict__ P1, double* __restrict__ x,
                                                         double* __restrict__ g, double* __restrict__ x_tmp,
                                                         double* __restrict__ g_tmp) {
  const bool p = st->flips & 1u;
  const double2 *cur, *old;  // latest iterate, previous iterate
  if (st->converged_at_init) { cur = P0; old = P0; }
  else if (st->converged) { cur = p ? P0 : P1; old = p ? P1 : P0; }  // the converging sweep wrote the "new" side
  else { cur = p ? P1 : P0; old = cur; }                            // rolled forward: x == x_tmp (convex.hpp:662-663)
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const double2 a = cur[i], b = old[i];
    x[i] = a.x; g[i] = a.y; x_tmp[i] = b.x; g_tmp[i] = b.y;
  }
}

// ---- incidence index build --------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) k_inc_count(size_t C, size_t N, const int2* __restrict__ pairs,
                                                     int32_t* __restrict__ deg, int* __restrict__ bad) {
  for (size_t c = blockIdx.x * (size_t)blockDim.x + threadIdx.x; c < C; c += (size_t)gridDim.x * blockDim.x) {
    const int2 ij = pairs[c];
    if (ij.x < 0 || ij.y < 0 || (size_t)ij.x >= N || (size_t)ij.y >= N || ij.x == ij.y) {
      *bad = 1;  // checked on the host before any gather runs: out-of-range indices never reach a kernel
      continue;
    }
    atomicAdd(&deg[ij.x], 1);
    atomicAdd(&deg[ij.y], 1);
  }
}
// entries use 31 bits; while the lists are being ordered bit 31 marks the second priority class (priority >= 0 or NaN),
// set here from one coalesced read of priority[c] instead of a gather per entry in the sort
__global__ void __launch_bounds__(kBlock) k_inc_fill(size_t C, const int2* __restrict__ pairs,
                                                    int32_t* __restrict__ cursor, int32_t* __restrict__ inc,
                                                    const double* __restrict__ priority) {
  for (size_t c = blockIdx.x * (size_t)blockDim.x + threadIdx.x; c < C; c += (size_t)gridDim.x * blockDim.x) {
    const int2 ij = pairs[c];
    const unsigned cls = (priority && !(priority[c] < 0.0)) ? 0x80000000u : 0u;
    const unsigned e = static_cast<unsigned>(c << 1) | cls;
    inc[atomicAdd(&cursor[ij.x], 1)] = static_cast<int32_t>(e);
    inc[atomicAdd(&cursor[ij.y], 1)] = static_cast<int32_t>(e | 1u);
  }
}
// ---- incidence build, fast path: the pair list is a neighbour list as the broad phase emits it -- rows sorted by the
// lower body, i < j in every row (GenNeighborLinks: unique pairs, sorted by (i, j)).  Then a body's SOURCE half (the
// contacts it is the lower body of) is its row of the list as it stands, and every contact it is the TARGET of lies in
// the row of a lower body, i.e. before its own row: only the transpose half needs counting with atomics, filling
// through atomic cursors and sorting (round 3 did all of that for both halves: k_inc_count, k_inc_fill and the segment
// sort over 2C entries were 1.6 of the 3.6 ms a rebuild step spent on list + operator; profiles/r03_kernel_stats.csv).
// flags: [0] an index out of range / a self pair, [1] the list is not of that form (-> the general path), [2] (k_inc_deg)
__global__ void __launch_bounds__(kBlock)
    k_inc_count_sorted(size_t C, size_t N, const int2* __restrict__ pairs, int32_t* __restrict__ tdeg,
                       int32_t* __restrict__ row_start, int* __restrict__ flags) {
  for (size_t c = blockIdx.x * (size_t)blockDim.x + threadIdx.x; c < C; c += (size_t)gridDim.x * blockDim.x) {
    const int2 ij = pairs[c];
    if (ij.x < 0 || ij.y < 0 || (size_t)ij.x >= N || (size_t)ij.y >= N || ij.x == ij.y) {
      flags[0] = 1;  // checked on the host before any gather runs: out-of-range indices never reach a kernel
      continue;
    }
    int prev = -1;
    if (c > 0) {
      prev = pairs[c - 1].x;
      if (prev < 0 || (size_t)prev >= N) prev = ij.x;  // (that row is reported by its own thread)
    }
    // (a run of more than kMaxEmptyRows bodies without a row of their own -- a dilute system, where the build costs
    // nothing either way -- is left to the general path rather than to one thread's loop)
    constexpr int kMaxEmptyRows = 4096;
    const bool tail_gap = (c + 1 == C) && N - (size_t)ij.x > (size_t)kMaxEmptyRows;
    if (ij.x >= ij.y || prev > ij.x || ij.x - prev > kMaxEmptyRows || tail_gap) {
      flags[1] = 1;
      continue;
    }
    atomicAdd(&tdeg[ij.y], 1);
    for (int b = prev + 1; b <= ij.x; ++b) row_start[b] = static_cast<int32_t>(c);  // first row of every body up to this one
    if (c + 1 == C)
      for (size_t b = (size_t)ij.x + 1; b <= N; ++b) row_start[b] = static_cast<int32_t>(C);
  }
}
// (flags[2] = the longest list: k_inc_arrange sorts a body's targets in one thread, so a list of thousands -- a large
//  body among small ones -- goes to the general path and its workgroup radix sort)
constexpr int kArrangeMaxList = 1024;
__global__ void __launch_bounds__(kBlock) k_inc_deg(size_t N, const int32_t* __restrict__ tdeg,
                                                   const int32_t* __restrict__ row_start, int32_t* __restrict__ deg,
                                                   int* __restrict__ flags) {
  for (size_t b = blockIdx.x * (size_t)blockDim.x + threadIdx.x; b < N; b += (size_t)gridDim.x * blockDim.x) {
    const int32_t d = tdeg[b] + (row_start[b + 1] - row_start[b]);
    deg[b] = d;
    if (d > kArrangeMaxList) flags[2] = 1;
  }
}
// staged lists: [targets, in the order their atomics arrived][sources, ascending]; the class bit as in k_inc_fill
__global__ void __launch_bounds__(kBlock)
    k_inc_fill_sorted(size_t C, const int2* __restrict__ pairs, const int32_t* __restrict__ inc_ptr,
                      const int32_t* __restrict__ tdeg, int32_t* __restrict__ tcursor,
                      const int32_t* __restrict__ row_start, unsigned* __restrict__ stage,
                      const double* __restrict__ priority) {
  for (size_t c = blockIdx.x * (size_t)blockDim.x + threadIdx.x; c < C; c += (size_t)gridDim.x * blockDim.x) {
    const int2 ij = pairs[c];
    const unsigned cls = (priority && !(priority[c] < 0.0)) ? 0x80000000u : 0u;
    const unsigned e = static_cast<unsigned>(c << 1) | cls;
    stage[inc_ptr[ij.x] + tdeg[ij.x] + (static_cast<int32_t>(c) - row_start[ij.x])] = e;
    stage[inc_ptr[ij.y] + atomicAdd(&tcursor[ij.y], 1)] = e | 1u;
  }
}
// One thread per body: its targets sorted (a handful), then the list in the order of the general path -- ascending
// (class, constraint, side) = [targets of class 0][sources of class 0][targets of class 1][sources of class 1] (every
// target constraint precedes every source constraint) -- with the class bit cleared, and the slot table beside it.
__global__ void __launch_bounds__(kBlock)
    k_inc_arrange(size_t N, const int32_t* __restrict__ inc_ptr, const int32_t* __restrict__ tdeg,
                  unsigned* __restrict__ stage, int32_t* __restrict__ inc, unsigned char* __restrict__ pos) {
  for (size_t b = blockIdx.x * (size_t)blockDim.x + threadIdx.x; b < N; b += (size_t)gridDim.x * blockDim.x) {
    const int32_t beg = inc_ptr[b], end = inc_ptr[b + 1], lt = tdeg[b];
    unsigned* T = stage + beg;
    for (int32_t a = 1; a < lt; ++a) {  // insertion sort by (class, constraint): the key as it stands
      const unsigned key = T[a];
      int32_t k = a - 1;
      while (k >= 0 && T[k] > key) {
        T[k + 1] = T[k];
        --k;
      }
      T[k + 1] = key;
    }
    int32_t out = beg;
    auto emit = [&](unsigned e) {
      const unsigned v = e & 0x7fffffffu;
      inc[out] = static_cast<int32_t>(v);
      const int32_t slot = out - beg;
      pos[2 * static_cast<size_t>(v >> 1) + (v & 1u)] = static_cast<unsigned char>(slot < 255 ? slot : 255);
      ++out;
    };
    for (unsigned cls = 0; cls < 2u; ++cls) {
      for (int32_t k = 0; k < lt; ++k)
        if ((T[k] >> 31) == cls) emit(T[k]);
      for (int32_t k = beg + lt; k < end; ++k)
        if ((stage[k] >> 31) == cls) emit(stage[k]);
    }
  }
}

// Each body's list in a fixed order whatever order the atomics arrived in: ascending (constraint, side) -- or, given a
// priority array, the contacts with priority < 0 first (ascending), then the others (ascending).  With priority = the
// signed separation the contacts that overlap at the start of the step, i.e. nearly all that will carry an impulse,
// sit together at the head of the list, so the entries and records the masked body sweep touches share sectors.
// The sort is sort_segments_u32 (sort.hip: one thread per list up to 32 entries, a workgroup radix sort for the long
// lists of a large body among small ones); the class bit is the key's top bit and is cleared afterwards.
__global__ void __launch_bounds__(kBlock) k_clear_class_bit(size_t nent, int32_t* __restrict__ inc) {
  unsigned* u = reinterpret_cast<unsigned*>(inc);
  for (size_t k = blockIdx.x * (size_t)blockDim.x + threadIdx.x; k < nent; k += (size_t)gridDim.x * blockDim.x)
    u[k] &= 0x7fffffffu;
}

// half-edge records in incidence order: the body sweep then streams them instead of gathering normals / arms
template <int KIN>
__global__ void __launch_bounds__(kBlock)
    k_half_build(size_t nent, const int32_t* __restrict__ inc, const double* __restrict__ normal,
                 const double* __restrict__ ra, const double* __restrict__ rb, const double* __restrict__ arc_s,
                 const double* __restrict__ arc_t, double* __restrict__ half) {
  constexpr int HW = (KIN == KIN_RIGID) ? 6 : (KIN == KIN_ROD ? 4 : 3);
  for (size_t k = blockIdx.x * (size_t)blockDim.x + threadIdx.x; k < nent; k += (size_t)gridDim.x * blockDim.x) {
    const int32_t e = inc[k];
    const size_t c = static_cast<size_t>(e >> 1);
    double* H = half + k * HW;
    const V3 n = load3(normal, c);
    H[0] = n.x; H[1] = n.y; H[2] = n.z;
    if (KIN == KIN_RIGID) {
      const V3 r = (e & 1) ? load3(rb, c) : load3(ra, c);
      H[3] = r.x; H[4] = r.y; H[5] = r.z;
    }
    if (KIN == KIN_ROD) H[3] = rod_arm_coef((e & 1) ? arc_t[c] : arc_s[c]);
  }
}

// slot of every (contact, side) in its body's incidence list, for the activity masks
__global__ void __launch_bounds__(kBlock) k_pos_build(size_t N, const int32_t* __restrict__ inc_ptr,
                                                     const int32_t* __restrict__ inc, unsigned char* __restrict__ pos) {
  for (size_t b = blockIdx.x * (size_t)blockDim.x + threadIdx.x; b < N; b += (size_t)gridDim.x * blockDim.x) {
    const int32_t beg = inc_ptr[b], end = inc_ptr[b + 1];
    for (int32_t k = beg; k < end; ++k) {
      const int32_t e = inc[k];
      const int32_t slot = k - beg;
      pos[2 * static_cast<size_t>(e >> 1) + (e & 1)] = static_cast<unsigned char>(slot < 255 ? slot : 255);
    }
  }
}
// ---- active lists: snapshot of the flagged entries (see OpView) -----------------------------------------------------------
__global__ void __launch_bounds__(kBlock)
    k_active_count(size_t first, size_t count, const int32_t* __restrict__ inc_ptr,
                   const unsigned long long* __restrict__ body_mask, int32_t* __restrict__ cnt) {
  for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < count; t += (size_t)gridDim.x * blockDim.x) {
    const size_t b = first + t;
    const int32_t deg = inc_ptr[b + 1] - inc_ptr[b];
    unsigned long long m = body_mask[b];
    if (deg < 64) m &= (1ull << deg) - 1ull;
    cnt[t] = __popcll(m);
  }
}
// The flagged entries and their records (HW doubles each), body by body in slot order, with the OUTPUT slots dealt to
// the lanes (one lane per body walked its flagged entries one after the
// other, 8 bytes at a time: 1.4 TB/s): a workgroup takes 256 bodies, keeps their masks, list starts and output offsets
// in LDS, and every lane then fills output slots o, o + 256, ... of the tile -- it finds the body by bisection of the
// offsets and the entry as the (o - offset)-th set bit of its mask.  Stores are contiguous, loads come from the tile's
// own stretch of the incidence lists.  Same arrays, same order.
template <int HW>
__global__ void __launch_bounds__(kBlock)
    k_active_fill_flat(size_t first, size_t count, const int32_t* __restrict__ inc_ptr, const int32_t* __restrict__ inc,
                       const double* __restrict__ half, const unsigned long long* __restrict__ body_mask,
                       const int32_t* __restrict__ aptr_local, int32_t* __restrict__ aptr, int32_t* __restrict__ aent,
                       double* __restrict__ arec, unsigned long long* __restrict__ snap_mask) {
  __shared__ unsigned long long tmask[kBlock];
  __shared__ int32_t tbeg[kBlock];
  __shared__ int32_t tptr[kBlock + 1];
  const size_t t0 = blockIdx.x * (size_t)kBlock;
  if (t0 >= count) return;
  const int nb = static_cast<int>(count - t0 < (size_t)kBlock ? count - t0 : (size_t)kBlock);
  const int tid = static_cast<int>(threadIdx.x);
  if (tid < nb) {
    const size_t t = t0 + tid, b = first + t;
    const int32_t beg = inc_ptr[b], deg = inc_ptr[b + 1] - beg;
    unsigned long long m = body_mask[b];
    if (deg < 64) m &= (1ull << deg) - 1ull;
    snap_mask[b] = m;
    const int32_t out = aptr_local[t];
    aptr[b] = out;
    if (t + 1 == count) aptr[b + 1] = aptr_local[count];
    tmask[tid] = m;
    tbeg[tid] = beg;
    tptr[tid] = out;
  }
  if (tid == 0) tptr[nb] = aptr_local[t0 + nb];
  __syncthreads();
  const int32_t E0 = tptr[0], E1 = tptr[nb];
  for (int32_t o = E0 + tid; o < E1; o += kBlock) {
    int lo = 0, hi = nb;  // the body j with tptr[j] <= o < tptr[j + 1]
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (tptr[mid] <= o) lo = mid; else hi = mid;
    }
    unsigned long long m = tmask[lo];
    for (int r = o - tptr[lo]; r > 0; --r) m &= m - 1ull;
    const size_t k = static_cast<size_t>(tbeg[lo] + (__ffsll(static_cast<long long>(m)) - 1));
    aent[o] = inc[k];
    if (HW % 2 == 0) {
      const double2* src = reinterpret_cast<const double2*>(half + k * HW);
      double2* dst = reinterpret_cast<double2*>(arec + static_cast<size_t>(o) * HW);
#pragma unroll
      for (int w = 0; w < HW / 2; ++w) dst[w] = src[w];
    } else {
#pragma unroll
      for (int w = 0; w < HW; ++w) arec[static_cast<size_t>(o) * HW + w] = half[k * HW + w];
    }
  }
}

// rod axes u = p1 - p0 from the 64-byte segment records
__global__ void __launch_bounds__(kBlock) k_rod_axes(size_t n, const double* __restrict__ seg, double* __restrict__ axis) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const double* r = seg + 8 * i;
    store3(axis, i, V3{r[3], r[4], r[5]} - V3{r[0], r[1], r[2]});
  }
}
// (U, Z) rows + omega -> (U, W) rows for the integrator
__global__ void __launch_bounds__(kBlock) k_assemble_velocity(size_t n, const double* __restrict__ vel,
                                                             const double* __restrict__ omega,
                                                             double* __restrict__ out) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    out[6 * i] = vel[6 * i]; out[6 * i + 1] = vel[6 * i + 1]; out[6 * i + 2] = vel[6 * i + 2];
    out[6 * i + 3] = omega[3 * i]; out[6 * i + 4] = omega[3 * i + 1]; out[6 * i + 5] = omega[3 * i + 2];
  }
}

// ------------------------------------------------------------------------------------------------------------------
// BUILD EXTENSION -- Coulomb friction as a cone complementarity problem (BASELINE configs[2] says "frictional LCP"; the
// reference has no frictional solver at all, SURVEY F2: parity unpinned, flagged wherever it is exposed).
// The same BBPGD iteration with a per-contact cone projection in place of the 1-D space projection:
//   unknown       p_c in R^3, the contact impulse in world coordinates (no tangent basis is ever built)
//   cone          K_c = { p : |p - (p.n) n| <= mu (p.n) }            (mu = 0: the ray lambda n, lambda >= 0)
//   forces        -p_c on the source body, +p_c on the target, acting at the contact points (lever arms ra, rb)
//   gradient      g_c = dt * [(U_j + W_j x rb) - (U_i + W_i x ra)] + sep_c n_c      (n . g is the frictionless g)
//   iteration     p <- Proj_K(p - step g), BB1 step and Linf projected-difference residual over the 3C components
// This is the convex (associative) relaxation of Coulomb friction used by APGD / BBPGD multibody solvers
// (Anitescu 2006; Mazhar, Heyn, Negrut, Tasora 2015): p in K, g in K* = { g : mu |g_t| <= g.n }, p . g = 0.
// The iterate is packed as (p, g), 48 bytes per contact, ping-pong by parity like the frictionless solver.
// ------------------------------------------------------------------------------------------------------------------
__device__ inline V3 project_cone(V3 v, V3 n, double mu) {
  const double a = dot(v, n);
  const V3 b = v - a * n;
  const double bn = norm(b);
  if (a >= 0.0 && bn <= mu * a) return v;        // inside the cone (a >= 0 matters only for mu = 0)
  if (mu * bn <= -a) return V3{0.0, 0.0, 0.0};   // inside the polar cone
  const double an = (mu * bn + a) / (mu * mu + 1.0);
  const V3 t = (bn > 0.0) ? ((mu * an) / bn) * b : V3{0.0, 0.0, 0.0};
  return an * n + t;
}
struct PG {
  V3 p, g;
};
__device__ inline PG load_pg(const double* P, size_t c) {
  const double2* q = reinterpret_cast<const double2*>(P + 6 * c);
  const double2 a = q[0], b = q[1], d = q[2];
  return {{a.x, a.y, b.x}, {b.y, d.x, d.y}};
}
__device__ inline void store_pg(double* P, size_t c, V3 p, V3 g) {
  double2* q = reinterpret_cast<double2*>(P + 6 * c);
  q[0] = make_double2(p.x, p.y);
  q[1] = make_double2(p.z, g.x);
  q[2] = make_double2(g.y, g.z);
}
// the impulse a contact carries in this sweep: the given p (INIT) or the projected BB step from the packed iterate
template <bool INIT>
__device__ inline V3 friction_iterate(size_t c, const double* __restrict__ Pt, const double* __restrict__ p0, V3 n,
                                      double mu, double step, bool step_is_zero, V3* p_old, V3* g_old) {
  if (INIT) return load3(p0, c);
  const PG s = load_pg(Pt, c);
  if (p_old) *p_old = s.p;
  if (g_old) *g_old = s.g;
  const V3 v = step_is_zero ? s.p : V3{s.p.x + (-step) * s.g.x, s.p.y + (-step) * s.g.y, s.p.z + (-step) * s.g.z};
  return project_cone(v, n, mu);
}

// body sweep (48-byte (n, r) half-edge records of the vector-arm operator)
template <bool INIT, int G, int U>
__global__ void __launch_bounds__(kBlock)
    k_body_friction(OpView op, const SolverState* __restrict__ st, const double* __restrict__ P0,
                    const double* __restrict__ P1, const double* __restrict__ p_init, double mu) {
  const double* Pt = P0;
  double step = 0.0;
  if (!INIT) {
    if (st->done) return;
    if (st->flips & 1u) Pt = P1;
    step = st->step;
  }
  const bool step_is_zero = fabs(-step) < kZeroTol;
  const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  const int sub = static_cast<int>(t % G);
  if (t / G >= op.body_count) return;
  const size_t b = op.body_first + t / G;
  DD3 Fdd{{0.0, 0.0}, {0.0, 0.0}, {0.0, 0.0}}, Tdd{{0.0, 0.0}, {0.0, 0.0}, {0.0, 0.0}};
  const int32_t beg = op.inc_ptr[b], end = op.inc_ptr[b + 1];
  for (int32_t k0 = beg + sub; k0 < end; k0 += G * U) {
    int32_t e[U];
    double2 h0[U], h1[U], h2[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int32_t k = k0 + u * G;
      e[u] = (k < end) ? op.inc[k] : -1;
      h0[u] = h1[u] = h2[u] = make_double2(0.0, 0.0);
      if (k < end) {
        const double2* H2 = reinterpret_cast<const double2*>(op.half + (size_t)k * 6);
        h0[u] = H2[0];
        h1[u] = H2[1];
        h2[u] = H2[2];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (e[u] < 0) continue;
      const V3 n{h0[u].x, h0[u].y, h1[u].x}, r{h1[u].y, h2[u].x, h2[u].y};
      const V3 p = friction_iterate<INIT>(static_cast<size_t>(e[u] >> 1), Pt, p_init, n, mu, step, step_is_zero,
                                          nullptr, nullptr);
      const V3 f = (e[u] & 1) ? p : V3{-p.x, -p.y, -p.z};  // source: -p, target: +p
      dd_add(Fdd, f);
      dd_add(Tdd, cross(r, f));
    }
  }
#pragma unroll
  for (int off = G / 2; off > 0; off >>= 1) {
    dd_add(Fdd.x, dd_shfl_xor(Fdd.x, off)); dd_add(Fdd.y, dd_shfl_xor(Fdd.y, off)); dd_add(Fdd.z, dd_shfl_xor(Fdd.z, off));
    dd_add(Tdd.x, dd_shfl_xor(Tdd.x, off)); dd_add(Tdd.y, dd_shfl_xor(Tdd.y, off)); dd_add(Tdd.z, dd_shfl_xor(Tdd.z, off));
  }
  if (sub != 0) return;
  const V3 F = dd_value(Fdd), T = dd_value(Tdd);
  const double mt = op.mt[b], mr = op.mr[b];
  double2* v = reinterpret_cast<double2*>(op.vel + 6 * b);
  v[0] = make_double2(mt * F.x, mt * F.y);
  v[1] = make_double2(mt * F.z, mr * T.x);
  v[2] = make_double2(mr * T.y, mr * T.z);
}

// constraint sweep: one 256-contact tile per workgroup (grid-stride beyond the cap), block partials as k_constraint
template <bool INIT>
__global__ void __launch_bounds__(kBlock)
    k_constraint_friction(OpView op, const SolverState* __restrict__ st, double* __restrict__ P0,
                          double* __restrict__ P1, const double* __restrict__ p_init, const double* __restrict__ sep,
                          double mu, double* __restrict__ partials) {
  __shared__ double scratch[2 * kBlock / 64];
  const double* Pt = P0;
  double* Pn = INIT ? P0 : P1;
  double step = 0.0;
  if (!INIT) {
    if (st->done) return;
    if (st->flips & 1u) {
      Pt = P1;
      Pn = P0;
    }
    step = st->step;
  }
  const bool step_is_zero = fabs(-step) < kZeroTol;
  double rmax = kLowest;
  DD num{0.0, 0.0}, den{0.0, 0.0};
  const size_t ntiles = (op.C + kBlock - 1) / kBlock;
  for (size_t lin = blockIdx.x; lin < ntiles; lin += gridDim.x) {
    const size_t c = lin * kBlock + threadIdx.x;
    if (c >= op.C) continue;
    const int2 ij = op.pairs[c];
    const V3 n = load3(op.normal, c);
    V3 p_old{0.0, 0.0, 0.0}, g_old{0.0, 0.0, 0.0};
    const V3 p = friction_iterate<INIT>(c, Pt, p_init, n, mu, step, step_is_zero, &p_old, &g_old);
    const double2* vi2 = reinterpret_cast<const double2*>(op.vel + 6 * (size_t)ij.x);
    const double2* vj2 = reinterpret_cast<const double2*>(op.vel + 6 * (size_t)ij.y);
    const double2 a0 = vi2[0], a1 = vi2[1], a2 = vi2[2], b0 = vj2[0], b1 = vj2[1], b2 = vj2[2];
    const V3 vi = V3{a0.x, a0.y, a1.x} + cross(V3{a1.y, a2.x, a2.y}, load3(op.ra, c));
    const V3 vj = V3{b0.x, b0.y, b1.x} + cross(V3{b1.y, b2.x, b2.y}, load3(op.rb, c));
    const double q = sep[c];
    const V3 g{op.dt * (vj.x - vi.x) + q * n.x, op.dt * (vj.y - vi.y) + q * n.y, op.dt * (vj.z - vi.z) + q * n.z};
    store_pg(Pn, c, p, g);
    if (op.counted == nullptr || op.counted[c]) {
      // Linf projected-difference residual over the three components (the reference's policy, convex.hpp:468-496,
      // with the cone projection)
      const V3 w = project_cone(V3{p.x - kSmallStep * g.x, p.y - kSmallStep * g.y, p.z - kSmallStep * g.z}, n, mu);
      const double r = fmax(fabs(p.x - w.x), fmax(fabs(p.y - w.y), fabs(p.z - w.z)));
      if (r > rmax) rmax = r;
      if (!INIT) {
        const V3 dp = p - p_old, dg = g - g_old;
        dd_add(num, dot(dp, dp));
        dd_add(den, dot(dp, dg));
      }
    }
  }
  const double m = block_max(rmax, scratch);
  const DD s1 = block_sum(num, scratch);
  const DD s2 = block_sum(den, scratch);
  if (threadIdx.x == 0) store_partial(partials, gridDim.x, blockIdx.x, m, s1, s2);
}

// latest iterate -> caller's p [C][3], g [C][3]
__global__ void __launch_bounds__(kBlock) k_finish_friction(size_t C, const SolverState* __restrict__ st,
                                                           const double* __restrict__ P0,
                                                           const double* __restrict__ P1, double* __restrict__ p,
                                                           double* __restrict__ g) {
  const bool odd = st->flips & 1u;
  // converged at init / never iterated: P0; converged at parity q: the sweep wrote the "new" side; otherwise rolled forward
  const double* cur = st->converged_at_init ? P0 : (st->converged ? (odd ? P0 : P1) : (odd ? P1 : P0));
  for (size_t c = blockIdx.x * (size_t)blockDim.x + threadIdx.x; c < C; c += (size_t)gridDim.x * blockDim.x) {
    const PG s = load_pg(cur, c);
    store3(p, c, s.p);
    store3(g, c, s.g);
  }
}


// ---- APGD for the same cone complementarity problem (BUILD EXTENSION, parity unpinned like the rest of this block) ----------
// Accelerated projected gradient descent with adaptive restart (Mazhar, Heyn, Negrut, Tasora 2015 -- the paper
// mundy_math/convex.hpp:476 cites for the residual), arranged so that an iteration is ONE operator application:
//   y   = p_k + beta (p_k - p_{k-1}),  g_y = g_k + beta (g_k - g_{k-1})      (g = N p + q is affine: no second sweep)
//   p+  = Proj_K(y - g_y / L),  g+ = N p+ + q                                (body sweep + constraint sweep, as BBPGD)
//   accept iff  (p+ - y) . (g+ - g_y) <= L |p+ - y|^2   -- the sufficient-decrease test f(p+) <= f(y) + g_y.(p+ - y) +
//               L/2 |p+ - y|^2 with f(p+) - f(y) written out (d . N d against L d . d: no cancellation of large f's);
//               otherwise L <- 2 L and the iteration is repeated from the same (p_k, p_{k-1}, beta)
//   accepted:   converged iff the Linf projected-difference residual of (p+, g+) <= tol;
//               theta+ = (-theta^2 + theta sqrt(theta^2 + 4)) / 2,  beta+ = theta (1 - theta) / (theta^2 + theta+);
//               restart (beta+ = 0, theta+ = 1) iff g_y . (p+ - p_k) > 0;  L <- 0.9 L
// Three packed (p, g) buffers rotate (current, previous, new): a rejected iterate is simply overwritten.  L starts at
// the initial residual (the reciprocal of BBPGD's first step).  Every sweep -- accepted or rejected -- counts as an iteration.
struct ApgdState {
  double L, theta, beta;
  int cur, prev, nxt;     // which of the three buffers holds p_k, p_{k-1}, and receives p+
  unsigned rejected;      // sweeps whose step was refused
};
constexpr int kApgdRed = 7;  // max residual term; (p+ - y).(g+ - g_y), |p+ - y|^2, g_y.(p+ - p_k) as double-double pairs
struct ApgdBufs {
  double* P[3];
};
// the iterate a contact carries in an APGD sweep (bitwise the same wherever it is evaluated)
__device__ inline V3 apgd_iterate(const ApgdBufs& B, const ApgdState& a, size_t c, V3 n, double mu, V3* y_out,
                                  V3* gy_out, V3* pk_out) {
  const PG k = load_pg(B.P[a.cur], c), m = load_pg(B.P[a.prev], c);
  const V3 y{k.p.x + a.beta * (k.p.x - m.p.x), k.p.y + a.beta * (k.p.y - m.p.y), k.p.z + a.beta * (k.p.z - m.p.z)};
  const V3 gy{k.g.x + a.beta * (k.g.x - m.g.x), k.g.y + a.beta * (k.g.y - m.g.y), k.g.z + a.beta * (k.g.z - m.g.z)};
  if (y_out) *y_out = y;
  if (gy_out) *gy_out = gy;
  if (pk_out) *pk_out = k.p;
  const double t = 1.0 / a.L;
  return project_cone(V3{y.x + (-t) * gy.x, y.y + (-t) * gy.y, y.z + (-t) * gy.z}, n, mu);
}
template <int G, int U>
__global__ void __launch_bounds__(kBlock)
    k_body_friction_apgd(OpView op, const SolverState* __restrict__ st, const ApgdState* __restrict__ as, ApgdBufs B,
                         double mu) {
  if (st->done) return;
  const ApgdState a = *as;
  const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  const int sub = static_cast<int>(t % G);
  if (t / G >= op.body_count) return;
  const size_t b = op.body_first + t / G;
  DD3 Fdd{{0.0, 0.0}, {0.0, 0.0}, {0.0, 0.0}}, Tdd{{0.0, 0.0}, {0.0, 0.0}, {0.0, 0.0}};
  const int32_t beg = op.inc_ptr[b], end = op.inc_ptr[b + 1];
  for (int32_t k0 = beg + sub; k0 < end; k0 += G * U) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int32_t k = k0 + u * G;
      if (k >= end) continue;
      const int32_t e = op.inc[k];
      const double2* H2 = reinterpret_cast<const double2*>(op.half + (size_t)k * 6);
      const double2 h0 = H2[0], h1 = H2[1], h2 = H2[2];
      const V3 n{h0.x, h0.y, h1.x}, r{h1.y, h2.x, h2.y};
      const V3 p = apgd_iterate(B, a, static_cast<size_t>(e >> 1), n, mu, nullptr, nullptr, nullptr);
      const V3 f = (e & 1) ? p : V3{-p.x, -p.y, -p.z};
      dd_add(Fdd, f);
      dd_add(Tdd, cross(r, f));
    }
  }
#pragma unroll
  for (int off = G / 2; off > 0; off >>= 1) {
    dd_add(Fdd.x, dd_shfl_xor(Fdd.x, off)); dd_add(Fdd.y, dd_shfl_xor(Fdd.y, off)); dd_add(Fdd.z, dd_shfl_xor(Fdd.z, off));
    dd_add(Tdd.x, dd_shfl_xor(Tdd.x, off)); dd_add(Tdd.y, dd_shfl_xor(Tdd.y, off)); dd_add(Tdd.z, dd_shfl_xor(Tdd.z, off));
  }
  if (sub != 0) return;
  const V3 F = dd_value(Fdd), T = dd_value(Tdd);
  const double mt = op.mt[b], mr = op.mr[b];
  double2* v = reinterpret_cast<double2*>(op.vel + 6 * b);
  v[0] = make_double2(mt * F.x, mt * F.y);
  v[1] = make_double2(mt * F.z, mr * T.x);
  v[2] = make_double2(mr * T.y, mr * T.z);
}
__global__ void __launch_bounds__(kBlock)
    k_constraint_friction_apgd(OpView op, const SolverState* __restrict__ st, const ApgdState* __restrict__ as,
                               ApgdBufs B, const double* __restrict__ sep, double mu, double* __restrict__ partials) {
  __shared__ double scratch[2 * kBlock / 64];
  if (st->done) return;
  const ApgdState a = *as;
  double* Pn = B.P[a.nxt];
  double rmax = kLowest;
  DD sa{0.0, 0.0}, sb{0.0, 0.0}, sr{0.0, 0.0};
  const size_t ntiles = (op.C + kBlock - 1) / kBlock;
  for (size_t lin = blockIdx.x; lin < ntiles; lin += gridDim.x) {
    const size_t c = lin * kBlock + threadIdx.x;
    if (c >= op.C) continue;
    const int2 ij = op.pairs[c];
    const V3 n = load3(op.normal, c);
    V3 y, gy, pk;
    const V3 p = apgd_iterate(B, a, c, n, mu, &y, &gy, &pk);
    const double2* vi2 = reinterpret_cast<const double2*>(op.vel + 6 * (size_t)ij.x);
    const double2* vj2 = reinterpret_cast<const double2*>(op.vel + 6 * (size_t)ij.y);
    const double2 a0 = vi2[0], a1 = vi2[1], a2 = vi2[2], b0 = vj2[0], b1 = vj2[1], b2 = vj2[2];
    const V3 vi = V3{a0.x, a0.y, a1.x} + cross(V3{a1.y, a2.x, a2.y}, load3(op.ra, c));
    const V3 vj = V3{b0.x, b0.y, b1.x} + cross(V3{b1.y, b2.x, b2.y}, load3(op.rb, c));
    const double q = sep[c];
    const V3 g{op.dt * (vj.x - vi.x) + q * n.x, op.dt * (vj.y - vi.y) + q * n.y, op.dt * (vj.z - vi.z) + q * n.z};
    store_pg(Pn, c, p, g);
    if (op.counted == nullptr || op.counted[c]) {
      const V3 w = project_cone(V3{p.x - kSmallStep * g.x, p.y - kSmallStep * g.y, p.z - kSmallStep * g.z}, n, mu);
      const double r = fmax(fabs(p.x - w.x), fmax(fabs(p.y - w.y), fabs(p.z - w.z)));
      if (r > rmax) rmax = r;
      const V3 d = p - y;
      dd_add(sa, dot(d, g - gy));
      dd_add(sb, dot(d, d));
      dd_add(sr, dot(gy, p - pk));
    }
  }
  const double m = block_max(rmax, scratch);
  const DD s1 = block_sum(sa, scratch);
  const DD s2 = block_sum(sb, scratch);
  const DD s3 = block_sum(sr, scratch);
  if (threadIdx.x == 0) {
    const size_t stride = gridDim.x, slot = blockIdx.x;
    partials[slot] = m;
    partials[stride + slot] = s1.hi;
    partials[2 * stride + slot] = s1.lo;
    partials[3 * stride + slot] = s2.hi;
    partials[4 * stride + slot] = s2.lo;
    partials[5 * stride + slot] = s3.hi;
    partials[6 * stride + slot] = s3.lo;
  }
}
// after the INIT sweeps (k_finalize<X_INIT> has set residual / converged): the APGD state of a new solve
__global__ void k_apgd_begin(const SolverState* __restrict__ st, ApgdState* __restrict__ as) {
  as->L = st->residual;   // 1 / L = BBPGD's first step (convex.hpp:626-627)
  if (!(as->L > 0.0)) as->L = 1.0;
  as->theta = 1.0;
  as->beta = 0.0;
  as->cur = 0;            // the INIT sweep wrote (p_0, g_0) into buffer 0; p_{-1} = p_0
  as->prev = 0;
  as->nxt = 1;
  as->rejected = 0;
}
__global__ void __launch_bounds__(kFinalBlock)
    k_apgd_finalize(int nparts, const double* __restrict__ partials, SolverState* __restrict__ st,
                    ApgdState* __restrict__ as, double tol, unsigned max_iters) {
  __shared__ double scratch[2 * kFinalBlock / 64];
  if (st->done) return;
  double rmax = kLowest;
  DD sa{0.0, 0.0}, sb{0.0, 0.0}, sr{0.0, 0.0};
  for (int i = threadIdx.x; i < nparts; i += blockDim.x) {
    if (partials[i] > rmax) rmax = partials[i];
    dd_add(sa, DD{partials[(size_t)nparts + i], partials[2 * (size_t)nparts + i]});
    dd_add(sb, DD{partials[3 * (size_t)nparts + i], partials[4 * (size_t)nparts + i]});
    dd_add(sr, DD{partials[5 * (size_t)nparts + i], partials[6 * (size_t)nparts + i]});
  }
  rmax = block_max(rmax, scratch);
  sa = block_sum(sa, scratch);
  sb = block_sum(sb, scratch);
  sr = block_sum(sr, scratch);
  if (threadIdx.x != 0) return;
  const double A = dd_value(sa), Bq = dd_value(sb), R = dd_value(sr);
  st->iter += 1;  // every sweep counts
  if (A > as->L * Bq) {  // not enough decrease for this L: twice the curvature, same (p_k, p_{k-1}, beta) again
    as->L *= 2.0;
    as->rejected += 1;
    if (st->iter >= max_iters) st->done = 1;
    return;
  }
  const double res = rmax / kSmallStep;
  st->residual = res;
  st->step = 1.0 / as->L;
  const int accepted = as->nxt;
  if (res <= tol) {
    st->converged = 1;
    st->done = 1;
    as->cur = accepted;  // (k_finish_friction_apgd reads the solution from here)
    return;
  }
  const double th = as->theta;
  double th1 = (-(th * th) + th * sqrt(th * th + 4.0)) / 2.0;
  double beta = th * (1.0 - th) / (th * th + th1);
  if (R > 0.0) {  // the momentum points uphill: restart (O'Donoghue & Candes' gradient scheme, as APGD uses it)
    beta = 0.0;
    th1 = 1.0;
  }
  as->theta = th1;
  as->beta = beta;
  as->L *= 0.9;
  const int old_cur = as->cur;
  as->prev = old_cur;
  as->cur = accepted;
  as->nxt = 3 - old_cur - accepted;  // the buffer that is neither the new current nor the new previous
  st->flips += 1;
  if (st->iter >= max_iters) st->done = 1;
}
__global__ void __launch_bounds__(kBlock)
    k_finish_friction_apgd(size_t C, const SolverState* __restrict__ st, const ApgdState* __restrict__ as, ApgdBufs B,
                           double* __restrict__ p, double* __restrict__ g) {
  const double* cur = st->converged_at_init ? B.P[0] : B.P[as->cur];
  for (size_t c = blockIdx.x * (size_t)blockDim.x + threadIdx.x; c < C; c += (size_t)gridDim.x * blockDim.x) {
    const PG s = load_pg(cur, c);
    store3(p, c, s.p);
    store3(g, c, s.g);
  }
}

}  // namespace mhip

using namespace mhip;

struct mhip_contact_op {
  OpView view{};
  bool rot = false;  // any rotational kinematics (KIN_RIGID or KIN_ROD)
  int kin = KIN_TRANS;
  hipStream_t last_stream = nullptr;
  DeviceBuffer inc_ptr, inc, cursor, vel, partials, state, scanws, half, axis, omega, vel_out;
  DeviceBuffer iterate;  // packed (x, g) ping-pong pair of the fused / staged solvers: 2 x C x 16 bytes
  DeviceBuffer body_mask, pos;  // activity masks of the packed LCP solves (see OpView)
  DeviceBuffer sort_tmp, sort_list;  // workspaces of the incidence-list sort
  DeviceBuffer aptr, aent, arec, snap_mask, acnt;  // active lists (see OpView)
  int device = -1;  // the device current at create: where every buffer of this operator lives
  int tiering = 1;         // the fused solve may use the cold tier (mhip_contact_op_set_tiering); 2: test hook
  int drift_source = 0;    // tiered solves: 0 = by size (op_drift_source), 1 = rows, 2 = registers (time only)
  int lanes_per_body = 2;  // k_body's G (2, 4, 8 or 16), each lane keeping 4 (2 and 16 lanes: 2) half-edge chains in flight
  SolverState* host_state = nullptr;  // pinned
  // staged (multi-rank) solve context, set by mhip_bbpgd_stage_begin
  struct Stage {
    const double* q = nullptr;
    double *x = nullptr, *g = nullptr, *x_tmp = nullptr, *g_tmp = nullptr;
    Space sp{0, 0, 0};
    mhip_pgd_config cfg{0, 0, 0};
    bool active = false;
    unsigned part_used = 0;  // partial slots written by this iteration's constraint sweeps
    // cold tier of the staged solve: the packed pairs / q in use, the interior range [0, interior) whose contacts may
    // sleep (both bodies owned: their drifts are known here), whether a bad BB step pauses the solve (the same on every
    // rank: it must not depend on anything local)
    double *P0 = nullptr, *P1 = nullptr;
    const double* q_cur = nullptr;
    size_t interior = 0;
    bool interior_known = false, pause_on_bad_step = false;
    unsigned polls = 0, iter_at_poll = 0, snap_at = 0;
  } stage;
  // cold tier of the fused solve (see "Cold tier")
  struct Tier {
    bool active = false;    // the view points at the renumbered copies and inc is remapped
    bool tracking = false;  // the body rows ping-pong and the drifts accumulate
    bool disabled = false;  // this solve no longer tiers
    unsigned polled_at = 0; // iterations run at the last poll
    double* saved_vel = nullptr;
    int set = 0;            // geometry / iterate set in use
    size_t H = 0;           // hot contacts [0, H), cold tail [H, I)
    size_t I = 0;           // contacts [I, C) never go cold (the staged solver's boundary contacts: a ghost body's drift
                            // is not known here); the fused solve has I = C
    bool pingpong = false;  // the body rows alternate between two buffers (fused solve)
    DeviceBuffer geo[2], iter[2], misc, vel2, drift, arm, xprev;
    OpView saved{};         // the operator's own view, restored when the tiers are left
    // statistics of the last solve
    size_t tiered_iterations = 0, retiers = 0, wakeups = 0;
    unsigned service_blocks = 32;  // workgroups serving the cold tail in front of the hot sweep (set at polls)
    double hot_sum = 0.0;   // sum over tiered iterations of H / C
  } tier;
  // optional per-kernel timing (mhip_contact_op_set_profiling)
  bool profile = false;
  std::vector<hipEvent_t> events;  // 3 per enqueued iteration: before body, between, after constraint
  double body_ms = 0.0, constraint_ms = 0.0;
  size_t timed_iterations = 0;
};

namespace {

// Source of the body sweep's drift in tiered solves: 1 = difference of the two rows, 2 = accumulated in registers (see
// k_body).  Auto (0): the row form while both row tables (96 B per body) stay well inside the 256 MiB Infinity Cache,
// where its extra read is free and its smaller LDS image and register count pay; the register form beyond, and never
// for vector arms or a non-default lane layout.
constexpr size_t kRowDriftMaxBodies = 1750000;
static int op_drift_source(const mhip_contact_op* op) {
  if (op->kin == KIN_RIGID || op->lanes_per_body != 2) return 1;
  if (op->drift_source == 1 || op->drift_source == 2) return op->drift_source;
  return op->view.N > kRowDriftMaxBodies ? 2 : 1;
}
int op_launch_body(mhip_contact_op* op, int mode, const double* X0, const double* X1, const double* G0,
                   const double* G1, Space sp, hipStream_t s, bool packed = false) {
  if (op->view.N == 0) return MHIP_SUCCESS;
  const int G = (op->lanes_per_body == 1 || op->lanes_per_body == 2 || op->lanes_per_body == 8 || op->lanes_per_body == 16) ? op->lanes_per_body : 4;
  if (op->view.body_count == 0) return MHIP_SUCCESS;
  const unsigned grid = (grid_exact(op->view.body_count * (size_t)G) + 7u) & ~7u;  // multiple of 8: XCD tiles
  op->last_stream = s;
  const SolverState* st = op->state.as<SolverState>();
  // Flat sweep over the compact lists: a workgroup's bodies should fit ONE chunk of FLATP x 256 entries (a second,
  // mostly empty chunk is a second round of dependent loads behind two more barriers), and no more LDS than that
  // needs (24 KB per workgroup at FLATP = 2, 36 KB at 3: five against four workgroups per CU).  The length of the
  // lists is known from the last snapshot but one (it reaches the host with the polls); before that the whole
  // incidence list stands in.  10^6 rods (profiles/r03_ab_nt.txt): raw packing 640 entries per workgroup, 0.0900 ms
  // at FLATP = 2, 0.0865 at 3; relaxed packing (a fifth of that) 22.25 against 22.9 ms per step.
  int flatp = MHIP_KBODY_FLAT;
  if (MHIP_KBODY_FLAT == 2 && packed && mode == X_SOLVE && op->view.aptr != nullptr) {
    const int32_t known = op->host_state ? *reinterpret_cast<const int32_t*>(op->host_state + 1) : 0;
    const double entries = known > 0 ? static_cast<double>(known) : 2.0 * static_cast<double>(op->view.C);
    const double per_group = entries / static_cast<double>(op->view.body_count) * (kBlock / (double)G);
    if (per_group > MHIP_KBODY_FLAT3_ABOVE * 2 * kBlock) flatp = 3;
  }
  // tracked sweeps: where the drift comes from (see k_body's TRACK) -- the register form exists for the default lane
  // layout of rods and spheres only (the vector-arm sweep needs the smaller image of the row form for its occupancy)
  const bool regs = op_drift_source(op) == 2;
#define BODY_TRACKED(M, R, GG, UU, FP, LDS)                                                                    \
  do {                                                                                                         \
    if constexpr (GG == 2 && R != KIN_RIGID) {                                                                 \
      if (regs) { k_body<M, R, GG, UU, true, 2, FP><<<grid, kBlock, LDS, s>>>(op->view, st, X0, X1, G0, G1, sp); break; } \
    }                                                                                                          \
    k_body<M, R, GG, UU, true, 1, FP><<<grid, kBlock, LDS, s>>>(op->view, st, X0, X1, G0, G1, sp);            \
  } while (0)
#define BODY4(M, R, GG, UU)                                                                        \
  do {                                                                                             \
    if (flatp == 3 && packed && M == X_SOLVE && op->view.aptr != nullptr && op->view.drift != nullptr) \
      BODY_TRACKED(M, R, GG, UU, 3, MHIP_KBODY_DYN_LDS);                                           \
    else if (flatp == 3 && packed && M == X_SOLVE && op->view.aptr != nullptr)                     \
      k_body<M, R, GG, UU, true, 0, 3><<<grid, kBlock, MHIP_KBODY_DYN_LDS, s>>>(op->view, st, X0, X1, G0, G1, sp); \
    else if (MHIP_KBODY_FLAT > 0 && packed && M == X_SOLVE && op->view.aptr != nullptr && op->view.drift != nullptr) \
      BODY_TRACKED(M, R, GG, UU, MHIP_KBODY_FLAT, 0);                                              \
    else if (MHIP_KBODY_FLAT > 0 && packed && M == X_SOLVE && op->view.aptr != nullptr)             \
      k_body<M, R, GG, UU, true, 0, MHIP_KBODY_FLAT><<<grid, kBlock, 0, s>>>(op->view, st, X0, X1, G0, G1, sp); \
    else if (packed && M == X_SOLVE && op->view.drift != nullptr)                                  \
      BODY_TRACKED(M, R, GG, UU, 0, 0);                                                            \
    else if (packed && M == X_SOLVE)                                                               \
      k_body<M, R, GG, UU, true><<<grid, kBlock, 0, s>>>(op->view, st, X0, X1, G0, G1, sp);        \
    else                                                                                           \
      k_body<M, R, GG, UU, false><<<grid, kBlock, 0, s>>>(op->view, st, X0, X1, G0, G1, sp);       \
  } while (0)
#define BODY(M, R)                                    \
  do {                                                \
    if (G == 2) BODY4(M, R, 2, 2);                    \
    else if (G == 1) BODY4(M, R, 1, 2);               \
    else if (G == 8) BODY4(M, R, 8, 4);               \
    else if (G == 16) BODY4(M, R, 16, 2);             \
    else BODY4(M, R, 4, 4);                           \
  } while (0)
#define BODYK(K) \
  do { if (mode == X_APPLY) BODY(X_APPLY, K); else if (mode == X_INIT) BODY(X_INIT, K); else BODY(X_SOLVE, K); } while (0)
  if (op->kin == KIN_ROD) BODYK(KIN_ROD);
  else if (op->kin == KIN_RIGID) BODYK(KIN_RIGID);
  else BODYK(KIN_TRANS);
#undef BODYK
#undef BODY4
#undef BODY_TRACKED
#undef BODY
  MHIP_LAUNCH_CHECK();
  return MHIP_SUCCESS;
}

int op_launch_constraint(mhip_contact_op* op, int mode, double* X0, double* X1, double* G0, double* G1,
                         const double* q, Space sp, int resid_kind, unsigned grid, hipStream_t s,
                         bool packed = false) {
  if (op->view.C == 0) return MHIP_SUCCESS;
  const SolverState* st = op->state.as<SolverState>();
  double* parts = op->partials.as<double>();
#define CON(M, R)                                                                                              \
  do {                                                                                                         \
    if (packed && M != X_APPLY)                                                                                \
      k_constraint<M, R, true><<<grid, kBlock, MHIP_KCON_DYN_LDS, s>>>(op->view, st, X0, X1, G0, G1, q, sp, resid_kind, parts); \
    else                                                                                                       \
      k_constraint<M, R, false><<<grid, kBlock, 0, s>>>(op->view, st, X0, X1, G0, G1, q, sp, resid_kind, parts); \
  } while (0)
#define CONK(K) \
  do { if (mode == X_APPLY) CON(X_APPLY, K); else if (mode == X_INIT) CON(X_INIT, K); else CON(X_SOLVE, K); } while (0)
  if (op->kin == KIN_ROD) CONK(KIN_ROD);
  else if (op->kin == KIN_RIGID) CONK(KIN_RIGID);
  else CONK(KIN_TRANS);
#undef CONK
#undef CON
  MHIP_LAUNCH_CHECK();
  return MHIP_SUCCESS;
}

// block partials of the constraint sweep -> at most kFoldGroups triples (planes) ready for the single-workgroup pass
// `stride` is the distance between the three planes on entry and on return
#ifndef MHIP_FOLD_ABOVE
#define MHIP_FOLD_ABOVE 4096
#endif
#ifndef MHIP_FOLD_FINALIZE
#define MHIP_FOLD_FINALIZE 1
#endif
#ifndef MHIP_FOLD_BLOCK
#define MHIP_FOLD_BLOCK 128   // threads of a folding workgroup (64 / 128 / 256: 136.9 / 136.4 / 137.0 ms per step)
#endif
inline void fold_partials(unsigned& nparts, size_t& stride, double*& parts, const SolverState* st, int check_done,
                          hipStream_t s) {
  if (nparts <= MHIP_FOLD_ABOVE) return;  // one workgroup reads a few thousand triples as fast as a second launch would
  double* folded = parts + kRed * stride;
  k_fold_partials<<<kFoldGroups, kBlock, 0, s>>>((int)nparts, stride, parts, st, check_done, folded);
  parts = folded;
  nparts = kFoldGroups;
  stride = kFoldGroups;
}
// threads of a final pass over n partials: one wave is enough for a handful, 1024 for tens of thousands
inline unsigned final_block(size_t n) { return n <= 256 ? 64u : (n <= 2048 ? 256u : (unsigned)kFinalBlock); }
// plane distance of the staged solver's partials: up to two range sweeps per iteration share one set of planes
constexpr unsigned kStageStride = 2 * 32768;
// workgroups of the constraint sweep: one 256-constraint tile each up to the cap (10^6 rods: 29 775), grid-stride
// beyond; measured 0.145 ms at 2048 workgroups, 0.139 at 8192, 0.131 at one tile per workgroup.  One partial triple
// per workgroup.
constexpr int kMaxConstraintGrid = 32768;
unsigned constraint_grid(size_t C) {
  const size_t g = (C + kBlock - 1) / kBlock;
  return static_cast<unsigned>(g == 0 ? 1 : (g > (size_t)kMaxConstraintGrid ? kMaxConstraintGrid : g));
}

// snapshot of the flagged entries into the compact active lists (OpView::aptr ...); called where the host has just
// polled the solver, i.e. between two iterations
int op_snapshot_active(mhip_contact_op* op, hipStream_t s) {
  OpView& v = op->view;
  if (v.body_mask == nullptr || v.C == 0 || v.body_count == 0) return MHIP_SUCCESS;
  const size_t N = v.N, C = v.C, cnt = v.body_count;
  const int hw = (op->kin == KIN_RIGID) ? 6 : (op->kin == KIN_ROD ? 4 : 3);
  if (int e = op->aptr.reserve((N + 2) * sizeof(int32_t))) return e;
  if (int e = op->acnt.reserve((N + 2) * sizeof(int32_t))) return e;
  if (int e = op->aent.reserve((2 * C + 2) * sizeof(int32_t))) return e;
  if (int e = op->arec.reserve((2 * C + 2) * hw * sizeof(double))) return e;
  if (int e = op->snap_mask.reserve((N + 2) * sizeof(unsigned long long))) return e;
  int32_t* local = op->cursor.as<int32_t>();  // free since the incidence build
  k_active_count<<<grid_for(cnt), kBlock, 0, s>>>(v.body_first, cnt, v.inc_ptr, v.body_mask, op->acnt.as<int32_t>());
  MHIP_LAUNCH_CHECK();
  if (int e = exclusive_scan_i32(op->acnt.as<int32_t>(), local, cnt, op->scanws.ptr, s)) return e;
#define FILL(H)                                                                                                   \
  k_active_fill_flat<H><<<grid_exact(cnt), kBlock, 0, s>>>(v.body_first, cnt, v.inc_ptr, v.inc, v.half, v.body_mask, local, \
                                                           op->aptr.as<int32_t>(), op->aent.as<int32_t>(),          \
                                                           op->arec.as<double>(), op->snap_mask.as<unsigned long long>())
  if (hw == 6) FILL(6); else if (hw == 4) FILL(4); else FILL(3);
#undef FILL
  MHIP_LAUNCH_CHECK();
  // the length of the compact lists travels to the host with the next poll (it picks the flat sweep's chunk; see
  // op_launch_body): pinned spare words behind host_state, no synchronisation here
  if (op->host_state != nullptr)
    MHIP_HIP(hipMemcpyAsync(reinterpret_cast<int32_t*>(op->host_state + 1), op->aptr.as<int32_t>() + v.body_first + cnt,
                            sizeof(int32_t), hipMemcpyDeviceToHost, s));
  v.aptr = op->aptr.as<int32_t>();
  v.aent = op->aent.as<int32_t>();
  v.arec = op->arec.as<double>();
  v.snap_mask = op->snap_mask.as<unsigned long long>();
  return MHIP_SUCCESS;
}
// from this many completed iterations on the masks have settled enough for a snapshot to pay
constexpr unsigned kSnapshotAfter = 8;
#ifndef MHIP_SNAPSHOT_AT_INIT
#define MHIP_SNAPSHOT_AT_INIT 1
#endif


// ---- Cold tier ---------------------------------------------------------------------------------------------------------
// Two thirds of the contacts of a packing are inactive (x = 0, g > 0) and most of them stay so for the whole solve, yet
// the constraint sweep reads all 88 bytes of every contact every iteration just to find that out.  A contact adds
// nothing to an iteration while it is inactive: its force is zero, its residual term is zero, dx = 0 kills its terms
// of both BB sums.  What must never be missed is the iteration in which its gradient turns non-positive -- and that
// can be bounded: the gradient is sep + dt * sdot, sdot is a contraction of the two bodies' contact-point velocities
// with the unit normal, so between two iterates it moves by at most the bodies' drifts (k_body accumulates
// drift[b] += dt (|dU|_1 + |dZ|_1 / 2) every sweep; with vector arms dt (|dU|_1 + |dW|_1 max|r|)).  A contact that goes cold at gradient g0 > 0 with the drifts at
// D0 has g > 0.1 g0 > 0 for as long as  drift[i] + drift[j] < D0 + 0.9 g0 -- in particular while each body stays below
// a threshold of its own, drift[i] < D0_i + 0.45 g0 and drift[j] < D0_j + 0.45 g0 (kTierShare).
//   At a convergence poll the contacts are RENUMBERED hot-first (stable partition): geometry, q, the slot table and
//   both packed iterates are copied into that order, the incidence entries are remapped, the compact active lists
//   rebuilt.  A contact goes to the cold tail when x == 0 in the last two iterates and a quarter of its gradient
//   exceeds what either of its bodies is expected to drift until the next poll (its drift over the last period, scaled
//   to the length of the next one).  (Sharing the slack in proportion to the two expectations instead tiers 64
//   iterations earlier on the raw packing but wakes 2.3 x as many contacts: same step time, relaxed steps 5 % slower.)  The ordinary sweep then runs over [0, H) only.  Nothing scans the tail: every body
//   knows the smallest threshold among its sleeping contacts (fire_at), the body sweep -- which has just updated the
//   body's drift -- lists the bodies that reached it, and a few "service" workgroups at the front of the constraint
//   sweep's grid walk those bodies' incidence lists and wake the contacts concerned (TierCheck).  The same workgroups
//   evaluate the awake contacts of the tail, from then on every iteration, exactly as the ordinary sweep would (a
//   separate launch for them cost 10 us an iteration, a twentieth of it; a sleeper's stale
//   pair says x = 0, g > 0, which is all an evaluation uses of an inactive contact: Proj(0 - step g) = 0 and dx = 0).
//   Every sum is a double-double pair rounded once, so the partition does not reach the iterates: same bits, same
//   iteration count as the untiered solve (tests).
//   Leaving the tiers (end of the solve, or a BB step outside [0, finite], which would need the sleepers' exact
//   gradients -- k_finalize pauses the solve for that): the sleepers' gradients are evaluated for the last two iterates
//   from the two body-row buffers, everything is scattered back to the caller's numbering, inc is restored.
constexpr double kTierMinGap = 1e-9;        // a contact goes cold only with g above this (far above rounding noise)
// smaller problems are launch-bound, the extra launch per iteration and the renumbering passes cost more than the
// shorter sweep saves (rods, fused solve, tier on / off: 0.22M contacts 9.2 / 7.5 ms, 0.94M 18.5 / 16.8 ms, 1.9M 32.6 /
// 34.2 ms, 3.8M 94 / 108 ms, 7.6M 150 / 190 ms; scripts/tier_crossover.py)
constexpr size_t kTierMinContacts = 1500000;
#ifndef MHIP_TIER_RETIER_PERCENT
#define MHIP_TIER_RETIER_PERCENT 4  // (round 2: 10 -> 149.1 ms per step at 10^6 rods, 6 -> 147.3; round 3, tiers from iteration 56: 10 -> 133.6, 4 -> 133.2 with a fourth renumbering)
#endif
constexpr unsigned kTierHorizon = 64;  // iterations a sleeper's slack is sized for, at least
// share of a sleeper's gradient g0 each of its two bodies may drift by: g stays above (1 - 2 share) g0 > 0.  Round 2
// kept half the gradient in hand (share 1/4); nothing needs that margin -- an inactive contact only has to keep g > 0
// -- and with 0.45 the raw 10^6-rod packing tiers from iteration 56 instead of 120 (715 of 770 iterations tiered,
// 14 235 wake-ups against 8 974): 135.6 -> 134.4 ms per step, relaxed packing 22.0 -> 21.7
#ifndef MHIP_TIER_SHARE
#define MHIP_TIER_SHARE 0.45
#endif
constexpr double kTierShare = MHIP_TIER_SHARE;
#ifndef MHIP_TIER_SERVICE_BLOCKS
#define MHIP_TIER_SERVICE_BLOCKS 32  // (8 / 16 / 32 / 64: 125.2 / 124.2 / 123.4 / 124.1 ms per step at 10^6 rods: in the periods after a renumbering the walk of the fired bodies' lists, not the hot sweep, is what the launch waits for)
#endif
constexpr unsigned kTierFireBlocks = MHIP_TIER_SERVICE_BLOCKS;  // service workgroups in front of the hot sweep (a multiple of 8: XCDs)

struct TierGeo {
  int2* pairs;
  double *normal, *arc_s, *arc_t, *q, *ra, *rb;
  int32_t* orig;
  unsigned char* pos;
};
inline size_t tier_align(size_t b) { return (b + 255) & ~static_cast<size_t>(255); }
inline size_t tier_geo_bytes(size_t C, bool arms) {
  return tier_align(C * 8) + tier_align(3 * C * 8) + 3 * tier_align(C * 8) + tier_align(C * 4) + tier_align(2 * C) +
         (arms ? 2 * tier_align(3 * C * 8) : 0) + 256;
}
inline TierGeo tier_geo_at(void* base, size_t C) {
  char* p = static_cast<char*>(base);
  TierGeo g;
  g.pairs = reinterpret_cast<int2*>(p); p += tier_align(C * 8);
  g.normal = reinterpret_cast<double*>(p); p += tier_align(3 * C * 8);
  g.arc_s = reinterpret_cast<double*>(p); p += tier_align(C * 8);
  g.arc_t = reinterpret_cast<double*>(p); p += tier_align(C * 8);
  g.q = reinterpret_cast<double*>(p); p += tier_align(C * 8);
  g.orig = reinterpret_cast<int32_t*>(p); p += tier_align(C * 4);
  g.pos = reinterpret_cast<unsigned char*>(p); p += tier_align(2 * C);
  g.ra = reinterpret_cast<double*>(p); p += tier_align(3 * C * 8);  // (only reserved for the vector-arm operator)
  g.rb = reinterpret_cast<double*>(p);
  return g;
}
struct TierMisc {
  int32_t *flags, *rank, *new_of, *list, *fired;
  double* wake[2];  // [C][2] thresholds of the cold tail (see TierCheck)
  double *budget, *drift_prev, *fire_at;
  unsigned long long* counters;  // [0]: awake contacts of the tail (= length of list); [1]: bodies fired this iteration
};
inline size_t tier_misc_bytes(size_t C, size_t N) {
  return 4 * tier_align((C + 2) * 4) + tier_align((N + 2) * 4) + 2 * tier_align((2 * C + 4) * 8) +
         3 * tier_align((N + 2) * 8) + 256;
}
inline TierMisc tier_misc_at(void* base, size_t C, size_t N) {
  char* p = static_cast<char*>(base);
  TierMisc m;
  m.flags = reinterpret_cast<int32_t*>(p); p += tier_align((C + 2) * 4);
  m.rank = reinterpret_cast<int32_t*>(p); p += tier_align((C + 2) * 4);
  m.new_of = reinterpret_cast<int32_t*>(p); p += tier_align((C + 2) * 4);
  m.list = reinterpret_cast<int32_t*>(p); p += tier_align((C + 2) * 4);
  m.fired = reinterpret_cast<int32_t*>(p); p += tier_align((N + 2) * 4);
  m.wake[0] = reinterpret_cast<double*>(p); p += tier_align((2 * C + 4) * 8);
  m.wake[1] = reinterpret_cast<double*>(p); p += tier_align((2 * C + 4) * 8);
  m.budget = reinterpret_cast<double*>(p); p += tier_align((N + 2) * 8);
  m.drift_prev = reinterpret_cast<double*>(p); p += tier_align((N + 2) * 8);
  m.fire_at = reinterpret_cast<double*>(p); p += tier_align((N + 2) * 8);
  m.counters = reinterpret_cast<unsigned long long*>(p);
  return m;
}

// budget[b] = what body b is expected to drift until the next poll; drift_prev <- drift
__global__ void __launch_bounds__(kBlock) k_tier_budget(size_t N, const double* __restrict__ drift,
                                                       double* __restrict__ drift_prev, double scale,
                                                       double* __restrict__ budget) {
  for (size_t b = blockIdx.x * (size_t)blockDim.x + threadIdx.x; b < N; b += (size_t)gridDim.x * blockDim.x) {
    const double d = drift[b];
    budget[b] = scale * (d - drift_prev[b]);
    drift_prev[b] = d;
  }
}
// vector-arm operator: arm_max[b] = the longest lever arm among body b's half-edge records (n, r)
__global__ void __launch_bounds__(kBlock) k_tier_arm_max(size_t N, const int32_t* __restrict__ inc_ptr,
                                                        const double* __restrict__ half, double* __restrict__ arm_max) {
  for (size_t b = blockIdx.x * (size_t)blockDim.x + threadIdx.x; b < N; b += (size_t)gridDim.x * blockDim.x) {
    double m = 0.0;
    for (int32_t k = inc_ptr[b]; k < inc_ptr[b + 1]; ++k) {
      const double* H = half + static_cast<size_t>(k) * 6;
      const double r = sqrt(H[3] * H[3] + H[4] * H[4] + H[5] * H[5]);
      if (r > m) m = r;
    }
    arm_max[b] = m * (1.0 + 1e-12);
  }
}
// flags[c] = 1: hot.  H_old = C when the solve is not tiered yet (nobody is asleep).
__global__ void __launch_bounds__(kBlock)
    k_tier_classify(size_t C, const int2* __restrict__ pairs, const double2* __restrict__ Pcur,
                    const double2* __restrict__ Pprev, const double* __restrict__ budget,
                    const double* __restrict__ drift, const double* __restrict__ wake_old, size_t H_old,
                    int32_t* __restrict__ flags) {
  for (size_t c = blockIdx.x * (size_t)blockDim.x + threadIdx.x; c < C; c += (size_t)gridDim.x * blockDim.x) {
    const int2 ij = pairs[c];
    const double need_i = budget[ij.x], need_j = budget[ij.y];
    bool cold;
    if (c >= H_old && wake_old[2 * (c - H_old)] > -1.7976931348623157e308) {
      // asleep (its pair is stale): stays cold while what is left of either threshold covers the coming period
      cold = wake_old[2 * (c - H_old)] - drift[ij.x] > need_i && wake_old[2 * (c - H_old) + 1] - drift[ij.y] > need_j;
    } else {
      const double2 a = Pcur[c], b = Pprev[c];
      cold = a.x == 0.0 && b.x == 0.0 && a.y > kTierMinGap && a.y <= 1.7976931348623157e308 && kTierShare * a.y > need_i &&
             kTierShare * a.y > need_j;
    }
    flags[c] = cold ? 0 : 1;
  }
}
// src numbering -> hot-first numbering (stable): geometry, q, slot table, both packed iterates, wake levels
template <int KIN>
__global__ void __launch_bounds__(kBlock)
    k_tier_permute(size_t C, size_t I, size_t H, int cur_is_p1, const int2* __restrict__ pairs, const double* __restrict__ normal,
                   const double* __restrict__ arc_s, const double* __restrict__ arc_t, const double* __restrict__ ra,
                   const double* __restrict__ rb, const double* __restrict__ q, const int32_t* __restrict__ orig,
                   const unsigned char* __restrict__ pos, TierGeo dst,
                   const double2* __restrict__ P0s, const double2* __restrict__ P1s, double2* __restrict__ P0d,
                   double2* __restrict__ P1d, const int32_t* __restrict__ flags, const int32_t* __restrict__ rank,
                   int32_t* __restrict__ new_of, const double* __restrict__ wake_old, size_t H_old,
                   double* __restrict__ wake_new, const double* __restrict__ drift, double* __restrict__ fire_at) {
  for (size_t c = blockIdx.x * (size_t)blockDim.x + threadIdx.x; c < C; c += (size_t)gridDim.x * blockDim.x) {
    const bool fixed = c >= I;  // beyond the range that is partitioned: stays where it is, never cold
    const bool hot = fixed || flags[c] != 0;
    const size_t r = fixed ? 0 : static_cast<size_t>(rank[c]);
    const size_t nc = fixed ? c : (hot ? r : H + (c - r));
    new_of[c] = static_cast<int32_t>(nc);
    const int2 ij = pairs[c];
    dst.pairs[nc] = ij;
    store3(dst.normal, nc, load3(normal, c));
    if (KIN == KIN_ROD) {
      dst.arc_s[nc] = arc_s[c];
      dst.arc_t[nc] = arc_t[c];
    }
    if (KIN == KIN_RIGID) {
      store3(dst.ra, nc, load3(ra, c));
      store3(dst.rb, nc, load3(rb, c));
    }
    dst.q[nc] = q[c];
    dst.orig[nc] = orig ? orig[c] : static_cast<int32_t>(c);
    dst.pos[2 * nc] = pos[2 * c];
    dst.pos[2 * nc + 1] = pos[2 * c + 1];
    double2 a0 = P0s[c], a1 = P1s[c];
    if (!hot) {
      const bool asleep = c >= H_old && wake_old[2 * (c - H_old)] > -1.7976931348623157e308;
      double ti, tj;
      if (asleep) {  // its thresholds stand; the pair keeps saying x = 0, g > 0
        ti = wake_old[2 * (c - H_old)];
        tj = wake_old[2 * (c - H_old) + 1];
      } else {       // half the gradient is slack, each body gets half of that
        const double g = cur_is_p1 ? a1.y : a0.y;
        ti = drift[ij.x] + kTierShare * g;
        tj = drift[ij.y] + kTierShare * g;
        a0 = a1 = make_double2(0.0, g);
      }
      wake_new[2 * (nc - H)] = ti;
      wake_new[2 * (nc - H) + 1] = tj;
      // (thresholds are positive doubles: their bit patterns order like the values)
      atomicMin(reinterpret_cast<unsigned long long*>(fire_at + ij.x), static_cast<unsigned long long>(__double_as_longlong(ti)));
      atomicMin(reinterpret_cast<unsigned long long*>(fire_at + ij.y), static_cast<unsigned long long>(__double_as_longlong(tj)));
    }
    P0d[nc] = a0;
    P1d[nc] = a1;
  }
}
__global__ void __launch_bounds__(kBlock) k_tier_remap_inc(size_t n, int32_t* __restrict__ inc,
                                                          const int32_t* __restrict__ map) {
  for (size_t k = blockIdx.x * (size_t)blockDim.x + threadIdx.x; k < n; k += (size_t)gridDim.x * blockDim.x) {
    const int32_t e = inc[k];
    inc[k] = (map[e >> 1] << 1) | (e & 1);
  }
}
// leaving the tiers: the gradient of every sleeping contact at the last two iterates, from the rows of those iterates
// (cur_is_1: which of the packed pairs holds the latest iterate)
template <int KIN>
__global__ void __launch_bounds__(kBlock)
    k_tier_refresh_sleepers(OpView op /*tier view*/, size_t H, size_t I, int cur_is_1, const double* __restrict__ vcur,
                            const double* __restrict__ vprev, const double* __restrict__ q,
                            const double* __restrict__ wake, double2* __restrict__ P0, double2* __restrict__ P1) {
  double2* Pcur = cur_is_1 ? P1 : P0;
  double2* Pprev = cur_is_1 ? P0 : P1;
  for (size_t c = H + blockIdx.x * (size_t)blockDim.x + threadIdx.x; c < I; c += (size_t)gridDim.x * blockDim.x) {
    if (!(wake[2 * (c - H)] > -1.7976931348623157e308)) continue;  // awake: its pairs are exact already
    const int2 ij = op.pairs[c];
    Pcur[c] = make_double2(0.0, 1.0 * q[c] + 1.0 * contact_dt_sdot<KIN>(op, vcur, c, ij));
    Pprev[c] = make_double2(0.0, 1.0 * q[c] + 1.0 * contact_dt_sdot<KIN>(op, vprev, c, ij));
  }
}
// x of the previous iterate as a plain vector (tier numbering): what the body sweep needs to rebuild that iterate's rows
__global__ void __launch_bounds__(kBlock) k_tier_prev_x(size_t n, const double2* __restrict__ Pprev,
                                                       double* __restrict__ x) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) x[i] = Pprev[i].x;
}
// tier numbering -> the caller's: the solver's four vectors (final) ...
__global__ void __launch_bounds__(kBlock)
    k_tier_finish(size_t n, const SolverState* __restrict__ st, const double2* __restrict__ P0,
                  const double2* __restrict__ P1, const int32_t* __restrict__ orig, double* __restrict__ x,
                  double* __restrict__ g, double* __restrict__ x_tmp, double* __restrict__ g_tmp) {
  const bool p = st->flips & 1u;
  const double2 *cur, *old;  // as k_finish_packed
  if (st->converged_at_init) { cur = P0; old = P0; }
  else if (st->converged) { cur = p ? P0 : P1; old = p ? P1 : P0; }
  else { cur = p ? P1 : P0; old = cur; }
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const double2 a = cur[i], b = old[i];
    const size_t o = static_cast<size_t>(orig[i]);
    x[o] = a.x; g[o] = a.y; x_tmp[o] = b.x; g_tmp[o] = b.y;
  }
}
// ... or the packed pairs (the solve goes on untiered)
__global__ void __launch_bounds__(kBlock)
    k_tier_scatter_pairs(size_t n, const double2* __restrict__ P0, const double2* __restrict__ P1,
                         const int32_t* __restrict__ orig, double2* __restrict__ Q0, double2* __restrict__ Q1) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const size_t o = static_cast<size_t>(orig[i]);
    Q0[o] = P0[i];
    Q1[o] = P1[i];
  }
}

// what does not depend on the size or the partition of the problem (the same on every rank of a distributed solve)
bool tier_problem_kind(const Space& sp, int resid_kind) {
  return sp.kind == MHIP_SPACE_LOWER_BOUND && sp.lo == 0.0 && resid_kind == MHIP_RESIDUAL_PROJECTED_DIFF;
}
bool tier_eligible(const mhip_contact_op* op, const Space& sp, int resid_kind) {
  const OpView& v = op->view;
  return (v.C >= kTierMinContacts || op->tiering >= 2) && v.body_mask != nullptr && v.counted == nullptr &&
         v.body_first == 0 && v.body_count == v.N && tier_problem_kind(sp, resid_kind);
}

// service workgroups of a tiered sweep for an awake list of the given length (known at polls; a multiple of 8: XCDs)
unsigned tier_service_blocks(size_t awake) {
  const size_t want = kTierFireBlocks + 8 * (awake / (8 * 2 * kBlock));  // about two rounds per workgroup
  return static_cast<unsigned>(want < 256 ? want : 256);
}

// the packed pair a tiered (or not yet tiered) solve is iterating on
struct TierPairs {
  double *P0, *P1;
  const double* q;
};

// At a poll (the stream is idle, host_state is current).  First call of a solve: the drift bookkeeping starts (drift
// accumulates; with pingpong the body rows alternate between two buffers).  Only contacts [0, I) may go cold.  Later calls: classify against the drift of the period just run, and renumber
// hot-first when that pays.  iters_done = iterations run so far, next_period = iterations until the next poll.
// `cur` = the pairs / q in use; on return the ones to use from now on.
int tier_update(mhip_contact_op* op, TierPairs& cur, unsigned iters_done, unsigned next_period, size_t I, bool pingpong,
                hipStream_t s) {
  mhip_contact_op::Tier& t = op->tier;
  OpView& v = op->view;
  const size_t C = v.C, N = v.N;
  if (I > C) I = C;
  if (int e = t.misc.reserve(tier_misc_bytes(C, N))) return e;
  if (int e = op->scanws.reserve(scan_workspace_bytes(C + 2) + 64)) return e;
  const TierMisc m = tier_misc_at(t.misc.ptr, C, N);
  const bool cur_is_p1 = op->host_state->flips & 1u;
  if (!t.tracking) {
    if (int e = t.vel2.reserve((6 * N + 8) * sizeof(double))) return e;
    if (int e = t.drift.reserve((N + 8) * sizeof(double))) return e;
    MHIP_HIP(hipMemsetAsync(t.drift.ptr, 0, N * sizeof(double), s));
    MHIP_HIP(hipMemsetAsync(m.drift_prev, 0, N * sizeof(double), s));
    // the body rows start to ping-pong: the current rows must sit in the buffer of the current parity
    if (pingpong && cur_is_p1)
      MHIP_HIP(hipMemcpyAsync(t.vel2.ptr, v.vel, 6 * N * sizeof(double), hipMemcpyDeviceToDevice, s));
    if (op->kin == KIN_RIGID) {  // the drift of a body with vector arms is scaled by its longest arm
      if (int e = t.arm.reserve((N + 8) * sizeof(double))) return e;
      k_tier_arm_max<<<grid_for(N), kBlock, 0, s>>>(N, v.inc_ptr, v.half, t.arm.as<double>());
      MHIP_LAUNCH_CHECK();
      v.arm_max = t.arm.as<double>();
    }
    t.saved_vel = v.vel;
    t.pingpong = pingpong;
    v.vel_alt = pingpong ? t.vel2.as<double>() : nullptr;
    v.drift = t.drift.as<double>();
    t.tracking = true;
    t.polled_at = iters_done;
    return MHIP_SUCCESS;
  }
  const unsigned last_period = iters_done - t.polled_at;
  // (a poll that comes too soon after the last one only lengthens the window the drift rate is measured over: half the
  // horizon, less at the start of a solve, where the drifts are largest and a short window errs on the safe side)
  const unsigned min_window = iters_done / 2 < kTierHorizon / 2 ? iters_done / 2 : kTierHorizon / 2;
  if (last_period == 0 || last_period < min_window) return MHIP_SUCCESS;
  t.polled_at = iters_done;
  // the drift a body is expected to collect while its contacts sleep: the rate of the period just run over the
  // iterations until the next poll -- but at least kTierHorizon of them: a sleeper is meant to last until the next
  // RENUMBERING, which comes only when it pays, and BB steps are bursty, so the budget of a short polling period (the
  // caller's choice in the staged driver) predicts too little.  (10^6 rods, staged, polled every 16 / 32 / 64
  // iterations: 1 130 000 / 136 000 / 3 700 contacts woken and 259 / 170 / 156 ms per step before this floor.)
  const unsigned horizon = next_period > kTierHorizon ? next_period : kTierHorizon;
  const double scale = static_cast<double>(horizon) / static_cast<double>(last_period);
  k_tier_budget<<<grid_for(N), kBlock, 0, s>>>(N, t.drift.as<double>(), m.drift_prev, scale, m.budget);
  MHIP_LAUNCH_CHECK();
  const double2* Pc = reinterpret_cast<const double2*>(cur_is_p1 ? cur.P1 : cur.P0);
  const double2* Pp = reinterpret_cast<const double2*>(cur_is_p1 ? cur.P0 : cur.P1);
  const double* wake_old = t.active ? m.wake[t.set] : m.wake[0];
  const size_t H_old = t.active ? t.H : C;
  if (I == 0) return MHIP_SUCCESS;
  k_tier_classify<<<grid_for(I), kBlock, 0, s>>>(I, v.pairs, Pc, Pp, m.budget, t.drift.as<double>(), wake_old, H_old,
                                                 m.flags);
  MHIP_LAUNCH_CHECK();
  if (int e = exclusive_scan_i32(m.flags, m.rank, I, op->scanws.ptr, s)) return e;
  int32_t H32 = 0;
  unsigned long long awake = 0;
  MHIP_HIP(hipMemcpyAsync(&H32, m.rank + I, sizeof(int32_t), hipMemcpyDeviceToHost, s));
  if (t.active) MHIP_HIP(hipMemcpyAsync(&awake, m.counters, sizeof(awake), hipMemcpyDeviceToHost, s));
  MHIP_HIP(hipStreamSynchronize(s));
  const size_t H = static_cast<size_t>(H32);
#ifdef MHIP_TIER_DEBUG
  fprintf(stderr, "compact active entries at the last snapshot but one: %d of %zu half edges\n",
          *reinterpret_cast<const int32_t*>(op->host_state + 1), 2 * C);
  fprintf(stderr, "tier_update: iter %u last_period %u next %u scale %.2f  H %zu of I %zu (%.3f)  active %d awake %llu\n",
          iters_done, last_period, next_period, scale, H, I, (double)H / (double)I, (int)t.active, awake);
#endif
  if (t.active) {
    // (a long awake list gets more service workgroups: each evaluates its share in rounds of a workgroup's width)
    t.service_blocks = tier_service_blocks(static_cast<size_t>(awake));
    // renumbering costs about four constraint sweeps: only when what is swept in full (the hot range and the awake
    // part of the tail, the latter through scattered accesses) can shrink by MHIP_TIER_RETIER_PERCENT percent
    if (100 * H >= (100 - MHIP_TIER_RETIER_PERCENT) * (t.H + static_cast<size_t>(awake))) return MHIP_SUCCESS;
  } else if (10 * H > 9 * I) {
    return MHIP_SUCCESS;  // (almost) everything is hot: nothing to gain yet
  }
  const int dst_set = t.active ? (t.set ^ 1) : 0;
  if (int e = t.geo[dst_set].reserve(tier_geo_bytes(C, op->kin == KIN_RIGID))) return e;
  if (int e = t.iter[dst_set].reserve(2 * (C + 1) * sizeof(double2))) return e;
  const TierGeo dst = tier_geo_at(t.geo[dst_set].ptr, C);
  double2* P0d = t.iter[dst_set].as<double2>();
  double2* P1d = P0d + C;
  const int32_t* orig_src = nullptr;
  if (t.active)
    orig_src = tier_geo_at(t.geo[t.set].ptr, C).orig;
  else
    t.saved = v;
  if (int e = mhip_fill(N, m.fire_at, __builtin_huge_val(), reinterpret_cast<mhip_stream_t>(s))) return e;
#define PERMUTE(K)                                                                                                    \
  k_tier_permute<K><<<grid_for(C), kBlock, 0, s>>>(C, I, H, cur_is_p1 ? 1 : 0, v.pairs, v.normal, v.arc_s, v.arc_t, v.ra, \
                                                   v.rb, cur.q,                                                      \
                                                   orig_src, v.pos, dst, reinterpret_cast<const double2*>(cur.P0),   \
                                                   reinterpret_cast<const double2*>(cur.P1), P0d, P1d, m.flags,      \
                                                   m.rank, m.new_of, wake_old, H_old, m.wake[dst_set],               \
                                                   t.drift.as<double>(), m.fire_at)
  if (op->kin == KIN_ROD) PERMUTE(KIN_ROD); else if (op->kin == KIN_RIGID) PERMUTE(KIN_RIGID); else PERMUTE(KIN_TRANS);
#undef PERMUTE
  MHIP_LAUNCH_CHECK();
  k_tier_remap_inc<<<grid_for(2 * C), kBlock, 0, s>>>(2 * C, op->inc.as<int32_t>(), m.new_of);
  MHIP_LAUNCH_CHECK();
  MHIP_HIP(hipMemsetAsync(m.counters, 0, 4 * sizeof(unsigned long long), s));  // nobody of the new tail is awake
  v.pairs = dst.pairs;
  v.normal = dst.normal;
  v.arc_s = dst.arc_s;
  v.arc_t = dst.arc_t;
  if (op->kin == KIN_RIGID) {
    v.ra = dst.ra;
    v.rb = dst.rb;
  }
  v.pos = dst.pos;
  v.fire_at = m.fire_at;
  v.fired = m.fired;
  v.tier_counters = m.counters;
  cur.P0 = reinterpret_cast<double*>(P0d);
  cur.P1 = reinterpret_cast<double*>(P1d);
  cur.q = dst.q;
  t.wakeups += static_cast<size_t>(awake);
  t.active = true;
  t.service_blocks = kTierFireBlocks;
  t.set = dst_set;
  t.H = H;
  t.I = I;
  t.retiers += 1;
  return MHIP_SUCCESS;
}

// the rows of the latest iterate go back into the operator's own buffer and the row ping-pong / drift bookkeeping ends
int tier_stop_tracking(mhip_contact_op* op, hipStream_t s) {
  mhip_contact_op::Tier& t = op->tier;
  if (!t.tracking) return MHIP_SUCCESS;
  const SolverState& hs = *op->host_state;
  const unsigned curv = (hs.converged && !hs.converged_at_init) ? ((hs.flips + 1u) & 1u) : (hs.flips & 1u);
  if (t.pingpong && curv == 1u)
    MHIP_HIP(hipMemcpyAsync(t.saved_vel, t.vel2.ptr, 6 * op->view.N * sizeof(double), hipMemcpyDeviceToDevice, s));
  op->view.vel_alt = nullptr;
  op->view.drift = nullptr;
  t.tracking = false;
  return MHIP_SUCCESS;
}

// Leaves the tiers.  final: the four solver vectors are written in the caller's numbering; otherwise the packed pairs
// go back into (P0, P1) and the solve continues untiered.  host_state must be current.
int tier_release(mhip_contact_op* op, TierPairs& cur, bool final, double* P0, double* P1, const double* q_caller,
                 double* x, double* g, double* x_tmp, double* g_tmp, hipStream_t s) {
  mhip_contact_op::Tier& t = op->tier;
  if (!t.active) return tier_stop_tracking(op, s);
  OpView& v = op->view;
  const size_t C = v.C, N = v.N;
  const TierMisc m = tier_misc_at(t.misc.ptr, C, N);
  const TierGeo geo = tier_geo_at(t.geo[t.set].ptr, C);
  const SolverState* st = op->state.as<SolverState>();
  double2* T0 = reinterpret_cast<double2*>(cur.P0);
  double2* T1 = reinterpret_cast<double2*>(cur.P1);
  if (t.H < t.I) {
    const SolverState& hs = *op->host_state;
    const bool conv = hs.converged && !hs.converged_at_init;
    const unsigned cur1 = conv ? ((hs.flips + 1u) & 1u) : (hs.flips & 1u);  // which pair holds the latest iterate
    const double *vcur, *vprev;
    if (t.pingpong) {
      vcur = cur1 ? v.vel_alt : v.vel;
      vprev = cur1 ? v.vel : v.vel_alt;
    } else {
      // one row buffer (the staged solver's, filled by the halo as well): the rows of the previous iterate are rebuilt
      // from its multipliers -- the same sums of the same terms, each a double-double pair rounded once: the same bits.
      // Sleepers only touch owned bodies, whose rows the body sweep computes.
      if (int e = t.vel2.reserve((10 * N + 16) * sizeof(double))) return e;
      if (int e = t.xprev.reserve((C + 2) * sizeof(double))) return e;
      k_tier_prev_x<<<grid_for(C), kBlock, 0, s>>>(C, cur1 ? T0 : T1, t.xprev.as<double>());
      MHIP_LAUNCH_CHECK();
      double* rows = t.vel2.as<double>();
      double* const keep_vel = v.vel;
      double* const keep_omega = v.omega;
      double* const keep_drift = v.drift;
      const int32_t* const keep_aptr = v.aptr;
      v.vel = rows;
      v.omega = rows + 6 * N;  // (the integrator's angular velocities belong to the latest iterate: untouched)
      v.drift = nullptr;
      v.aptr = nullptr;
      const int e = op_launch_body(op, X_APPLY, t.xprev.as<double>(), t.xprev.as<double>(), nullptr, nullptr,
                                   Space{MHIP_SPACE_UNCONSTRAINED, 0.0, 0.0}, s);
      v.vel = keep_vel;
      v.omega = keep_omega;
      v.drift = keep_drift;
      v.aptr = keep_aptr;
      if (e) return e;
      vcur = v.vel;
      vprev = rows;
    }
    const unsigned grid = grid_for(t.I - t.H);
#define REFRESH(K)                                                                                              \
  k_tier_refresh_sleepers<K><<<grid, kBlock, 0, s>>>(v, t.H, t.I, static_cast<int>(cur1), vcur, vprev, cur.q,   \
                                                     m.wake[t.set], T0, T1)
    if (op->kin == KIN_ROD) REFRESH(KIN_ROD); else if (op->kin == KIN_RIGID) REFRESH(KIN_RIGID); else REFRESH(KIN_TRANS);
#undef REFRESH
    MHIP_LAUNCH_CHECK();
  }
  if (final)
    k_tier_finish<<<grid_for(C), kBlock, 0, s>>>(C, st, T0, T1, geo.orig, x, g, x_tmp, g_tmp);
  else
    k_tier_scatter_pairs<<<grid_for(C), kBlock, 0, s>>>(C, T0, T1, geo.orig, reinterpret_cast<double2*>(P0),
                                                        reinterpret_cast<double2*>(P1));
  MHIP_LAUNCH_CHECK();
  k_tier_remap_inc<<<grid_for(2 * C), kBlock, 0, s>>>(2 * C, op->inc.as<int32_t>(), geo.orig);
  MHIP_LAUNCH_CHECK();
  unsigned long long awake = 0;
  MHIP_HIP(hipMemcpyAsync(&awake, m.counters, sizeof(awake), hipMemcpyDeviceToHost, s));
  MHIP_HIP(hipStreamSynchronize(s));
  t.wakeups += static_cast<size_t>(awake);
  // the active lists refer to the tier numbering: gone with it
  const OpView keep = v;
  v = t.saved;
  v.xcd_aware = keep.xcd_aware;
  v.vel_alt = keep.vel_alt;
  v.drift = keep.drift;
  v.fire_at = nullptr;
  v.fired = nullptr;
  v.tier_counters = nullptr;
  v.aptr = nullptr;
  t.active = false;
  cur.P0 = P0;
  cur.P1 = P1;
  cur.q = q_caller;
  return tier_stop_tracking(op, s);
}

// the constraint sweep of a tiered iteration: [0, H) as ever, with the service workgroups of the tail [H, I) in front
// (their partial records follow the sweeping workgroups').  *nparts = partial records written.
int op_launch_constraint_tiered(mhip_contact_op* op, const TierPairs& cur, Space sp, int resid_kind, unsigned* nparts,
                                hipStream_t s) {
  mhip_contact_op::Tier& t = op->tier;
  const size_t C = op->view.C, N = op->view.N;
  const SolverState* st = op->state.as<SolverState>();
  double* parts = op->partials.as<double>();
  const TierMisc m = tier_misc_at(t.misc.ptr, C, N);
  const unsigned ghot = t.H ? constraint_grid(t.H) : 0u;
  // the tail is served by the first workgroups of the hot launch
  const unsigned gcheck = t.H < t.I ? t.service_blocks : 0u;
  OpView hot = op->view;
  hot.c_first = 0; hot.c_end = t.H; hot.part_offset = 0; hot.part_stride = kStageStride;
  const TierCheck tc{t.H, t.I, m.wake[t.set], m.list, m.counters, m.fired, m.fire_at, gcheck};
#define TIERED(K)                                                                                                 \
  k_constraint<X_SOLVE, K, true><<<ghot + gcheck, kBlock, MHIP_KCON_DYN_LDS, s>>>(hot, st, cur.P0, cur.P1, nullptr, nullptr,     \
                                                                 cur.q, sp, resid_kind, parts, tc)
  if (ghot + gcheck) {
    if (op->kin == KIN_ROD) TIERED(KIN_ROD); else if (op->kin == KIN_RIGID) TIERED(KIN_RIGID); else TIERED(KIN_TRANS);
    MHIP_LAUNCH_CHECK();
  }
#undef TIERED
  *nparts = ghot + gcheck;
  return MHIP_SUCCESS;
}

int check_config(const mhip_pgd_config* cfg) {
  MHIP_REQUIRE(cfg != nullptr, MHIP_ERR_INVALID_ARGUMENT, "config must not be null");
  MHIP_REQUIRE(cfg->residual_kind == MHIP_RESIDUAL_PROJECTED_DIFF ||
                   cfg->residual_kind == MHIP_RESIDUAL_PROJECTED_GRADIENT,
               MHIP_ERR_INVALID_ARGUMENT, "unknown residual kind %d", cfg->residual_kind);
  return MHIP_SUCCESS;
}

// The reference's driver, kernel by kernel (PGDStrategy::initialize / iterate, convex.hpp:614-666): used for the
// dense operator and as the unfused cross-check of the contact operator.
template <class Apply>
int solve_generic(size_t n, const Apply& apply, const double* q, Space sp, const mhip_pgd_config* cfg, double* x,
                  double* g, double* x_tmp, double* g_tmp, mhip_solve_result* result, hipStream_t s) {
  double res = 0.0;
  if (int e = launch_copy(n, x_tmp, x, s)) return e;
  if (int e = apply(x_tmp, g_tmp)) return e;
  if (int e = launch_axpby(n, 1.0, q, 1.0, g_tmp, s)) return e;
  if (int e = reduce_to_host<2>(n, x_tmp, g_tmp, nullptr, nullptr, cfg->residual_kind, sp, &res, s)) return e;
  if (cfg->residual_kind == MHIP_RESIDUAL_PROJECTED_DIFF) res = res / kSmallStep;
  double step = 1.0 / res;
  unsigned iter = 0;
  bool converged = (res <= cfg->tol);
  if (converged)
    if (int e = launch_copy(n, g, g_tmp, s)) return e;
  while (!(converged || iter >= cfg->max_iters)) {
    if (int e = launch_wrapped_axpbyz(n, 1.0, x_tmp, -step, g_tmp, x, sp, s)) return e;
    if (int e = apply(x, g)) return e;
    if (int e = launch_axpby(n, 1.0, q, 1.0, g, s)) return e;
    if (int e = reduce_to_host<2>(n, x, g, nullptr, nullptr, cfg->residual_kind, sp, &res, s)) return e;
    if (cfg->residual_kind == MHIP_RESIDUAL_PROJECTED_DIFF) res = res / kSmallStep;
    if (res <= cfg->tol) {
      converged = true;
      break;
    }
    double num = 0.0, den = 0.0;
    if (int e = reduce_to_host<0>(n, x, x_tmp, nullptr, nullptr, 0, sp, &num, s)) return e;
    if (int e = reduce_to_host<1>(n, x, x_tmp, g, g_tmp, 0, sp, &den, s)) return e;
    den += kBBEps * (fabs(den) < kBBEps);
    step = num / den;
    if (int e = launch_copy(n, x_tmp, x, s)) return e;
    if (int e = launch_copy(n, g_tmp, g, s)) return e;
    ++iter;
  }
  MHIP_HIP(hipStreamSynchronize(s));
  result->num_iters = iter;
  result->residual = res;
  result->converged = converged ? 1 : 0;
  return MHIP_SUCCESS;
}

}  // namespace

#define REQ(p) MHIP_REQUIRE((p) != nullptr || n == 0, MHIP_ERR_INVALID_ARGUMENT, "%s: %s is null", __func__, #p)

namespace mhip {
void stage_state_words(mhip_contact_op_t op, const unsigned** flips, const int** done) {
  const SolverState* st = op->state.as<SolverState>();
  *flips = &st->flips;
  *done = &st->done;
}
int stage_reduce_exchange_finalize(mhip_contact_op_t op, int init, const MailboxArgs& mb, hipStream_t s) {
  MHIP_REQUIRE(op != nullptr && op->stage.active, MHIP_ERR_RUNTIME, "mhip_bbpgd_stage_begin has not been called");
  MHIP_REQUIRE(mb.peers != nullptr && mb.width == kRed && mb.world >= 1 && mb.world <= kMailboxMaxWorld,
               MHIP_ERR_INVALID_ARGUMENT, "mailbox of %d ranks, record width %d", mb.world, mb.width);
  const unsigned np = op->stage.part_used;
  const size_t ps = kStageStride;
  double* pp = op->partials.as<double>();
  const auto& cfg = op->stage.cfg;
  SolverState* st = op->state.as<SolverState>();
  // (folded records behind the kRed planes of partials, ticket word behind the solver state: as in the fused solve)
  unsigned* ticket = reinterpret_cast<unsigned*>(op->state.as<char>() + sizeof(SolverState) + 16);
  if (init)
    k_fold_finalize<X_INIT><<<kFoldGroups, MHIP_FOLD_BLOCK, 0, s>>>((int)np, ps, pp, pp + kRed * ps, ticket, st,
                                                                     cfg.residual_kind, cfg.tol, cfg.max_iters, 0,
                                                                     nullptr, mb);
  else
    k_fold_finalize<X_SOLVE><<<kFoldGroups, MHIP_FOLD_BLOCK, 0, s>>>(
        (int)np, ps, pp, pp + kRed * ps, ticket, st, cfg.residual_kind, cfg.tol, cfg.max_iters,
        op->stage.pause_on_bad_step ? ((op->tiering == 2 && op->stage.polls >= 10) ? 2 : 1) : 0,
        op->tier.active ? op->view.tier_counters : nullptr, mb);
  MHIP_LAUNCH_CHECK();
  op->stage.part_used = 0;
  return MHIP_SUCCESS;
}
int stage_reduce_exchange(mhip_contact_op_t op, int init, double* local, const MailboxArgs& mb, hipStream_t s) {
  MHIP_REQUIRE(op != nullptr && op->stage.active, MHIP_ERR_RUNTIME, "mhip_bbpgd_stage_begin has not been called");
  MHIP_REQUIRE(local != nullptr, MHIP_ERR_INVALID_ARGUMENT, "local record is null");
  unsigned np = op->stage.part_used;
  size_t ps = kStageStride;
  double* pp = op->partials.as<double>();
  fold_partials(np, ps, pp, op->state.as<SolverState>(), init ? 0 : 1, s);
  k_reduce_local<<<1, final_block(np), 0, s>>>((int)np, pp, ps, op->state.as<SolverState>(), init ? 0 : 1, local, mb);
  MHIP_LAUNCH_CHECK();
  op->stage.part_used = 0;
  return MHIP_SUCCESS;
}
}  // namespace mhip

extern "C" {

int mhip_deep_copy(size_t n, double* dst, const double* src, mhip_stream_t stream) {
  REQ(dst); REQ(src);
  return launch_copy(n, dst, src, as_stream(stream));
}
int mhip_fill(size_t n, double* dst, double value, mhip_stream_t stream) {
  REQ(dst);
  if (n == 0) return MHIP_SUCCESS;
  k_fill<<<grid_for(n), kBlock, 0, as_stream(stream)>>>(n, dst, value);
  MHIP_LAUNCH_CHECK();
  return MHIP_SUCCESS;
}
int mhip_axpby(size_t n, double alpha, const double* x, double beta, double* y, mhip_stream_t stream) {
  REQ(x); REQ(y);
  return launch_axpby(n, alpha, x, beta, y, as_stream(stream));
}
int mhip_wrapped_axpbyz(size_t n, double alpha, const double* x, double beta, const double* y, double* z,
                        const mhip_space* space, mhip_stream_t stream) {
  REQ(x); REQ(y); REQ(z);
  Space sp;
  if (int e = to_space(space, &sp)) return e;
  return launch_wrapped_axpbyz(n, alpha, x, beta, y, z, sp, as_stream(stream));
}
int mhip_diff_dot2(size_t n, const double* x, const double* y, double* result, mhip_stream_t stream) {
  REQ(x); REQ(y);
  MHIP_REQUIRE(result != nullptr, MHIP_ERR_INVALID_ARGUMENT, "result is null");
  return reduce_to_host<0>(n, x, y, nullptr, nullptr, 0, Space{0, 0, 0}, result, as_stream(stream));
}
int mhip_diff_dot4(size_t n, const double* x1, const double* x2, const double* y1, const double* y2, double* result,
                   mhip_stream_t stream) {
  REQ(x1); REQ(x2); REQ(y1); REQ(y2);
  MHIP_REQUIRE(result != nullptr, MHIP_ERR_INVALID_ARGUMENT, "result is null");
  return reduce_to_host<1>(n, x1, x2, y1, y2, 0, Space{0, 0, 0}, result, as_stream(stream));
}
int mhip_residual(size_t n, int residual_kind, const double* x, const double* grad, const mhip_space* space,
                  double* result, mhip_stream_t stream) {
  REQ(x); REQ(grad);
  MHIP_REQUIRE(result != nullptr, MHIP_ERR_INVALID_ARGUMENT, "result is null");
  MHIP_REQUIRE(residual_kind == 0 || residual_kind == 1, MHIP_ERR_INVALID_ARGUMENT, "unknown residual kind");
  Space sp;
  if (int e = to_space(space, &sp)) return e;
  double r;
  if (int e = reduce_to_host<2>(n, x, grad, nullptr, nullptr, residual_kind, sp, &r, as_stream(stream))) return e;
  *result = (residual_kind == MHIP_RESIDUAL_PROJECTED_DIFF) ? r / kSmallStep : r;
  return MHIP_SUCCESS;
}
int mhip_bb_step(size_t n, const double* x_old, const double* g_old, const double* x, const double* g, double* result,
                 mhip_stream_t stream) {
  REQ(x_old); REQ(g_old); REQ(x); REQ(g);
  MHIP_REQUIRE(result != nullptr, MHIP_ERR_INVALID_ARGUMENT, "result is null");
  double num, den;
  if (int e = reduce_to_host<0>(n, x, x_old, nullptr, nullptr, 0, Space{0, 0, 0}, &num, as_stream(stream))) return e;
  if (int e = reduce_to_host<1>(n, x, x_old, g, g_old, 0, Space{0, 0, 0}, &den, as_stream(stream))) return e;
  den += kBBEps * (fabs(den) < kBBEps);
  *result = num / den;
  return MHIP_SUCCESS;
}
int mhip_gemv(size_t n, const double* A, const double* x, double* y, mhip_stream_t stream) {
  REQ(A); REQ(x); REQ(y);
  if (n == 0) return MHIP_SUCCESS;
  k_gemv<<<grid_for(n * 64), kBlock, 0, as_stream(stream)>>>(n, A, x, y);
  MHIP_LAUNCH_CHECK();
  return MHIP_SUCCESS;
}

// An operator lives for one solve (its lists follow the contacts), but its workspaces -- a dozen device buffers, a
// pinned state block, timing events -- are the same size step after step.  A destroyed operator therefore leaves them
// in one process-wide spare set that the next create adopts: no hipMalloc / hipFree in the steady state (hipFree
// alone cost 1.6 ms per step at 10^6 rods).  mhip_release_cached_workspaces() frees the spare set.
constexpr int kOpBuffers = 30;
struct OpWorkspaces {
  DeviceBuffer buf[kOpBuffers];
  SolverState* host_state = nullptr;
  std::vector<hipEvent_t> events;
  bool held = false;
  int device = -1;  // the spare set is only handed to an operator on the device that allocated it
};
static std::mutex g_spare_mutex;
static OpWorkspaces g_spare;
static DeviceBuffer* op_buffers(mhip_contact_op* op, int k) {
  DeviceBuffer* all[kOpBuffers] = {&op->inc_ptr, &op->inc,     &op->cursor,   &op->vel,      &op->partials, &op->state,
                                   &op->scanws,  &op->half,    &op->axis,     &op->omega,    &op->vel_out,  &op->iterate,
                                   &op->body_mask, &op->pos,   &op->sort_tmp, &op->sort_list, &op->aptr,    &op->aent,
                                   &op->arec,    &op->snap_mask, &op->acnt,
                                   &op->tier.geo[0], &op->tier.geo[1], &op->tier.iter[0], &op->tier.iter[1],
                                   &op->tier.misc, &op->tier.vel2, &op->tier.drift, &op->tier.arm, &op->tier.xprev};
  return all[k];
}
static void free_workspaces(OpWorkspaces& w) {
  for (auto& b : w.buf) b.release();
  if (w.host_state) (void)hipHostFree(w.host_state);
  w.host_state = nullptr;
  for (auto& ev : w.events) (void)hipEventDestroy(ev);
  w.events.clear();
  w.held = false;
}
static void adopt_spare_workspaces(mhip_contact_op* op) {
  std::lock_guard<std::mutex> lock(g_spare_mutex);
  int dev = -1;
  if (!g_spare.held || hipGetDevice(&dev) != hipSuccess || dev != g_spare.device) return;
  for (int k = 0; k < kOpBuffers; ++k) {
    *op_buffers(op, k) = g_spare.buf[k];
    g_spare.buf[k] = DeviceBuffer{};
  }
  op->host_state = g_spare.host_state;
  g_spare.host_state = nullptr;
  op->events.swap(g_spare.events);
  g_spare.held = false;
}

static int create_contact_op(mhip_contact_op_t* handle, int kin, size_t num_constraints, size_t num_bodies,
                             const int32_t* pairs, const double* normal, const double* ra, const double* rb,
                             const double* arc_s, const double* arc_t, const double* seg, const double* mob_trans,
                             const double* mob_rot, double dt, const double* priority, mhip_stream_t stream) {
  TraceRange trace_range("ContactOperator::build");
  MHIP_REQUIRE(handle != nullptr, MHIP_ERR_INVALID_ARGUMENT, "handle is null");
  *handle = nullptr;
  const size_t C = num_constraints, N = num_bodies;
  MHIP_REQUIRE(C == 0 || (pairs && normal), MHIP_ERR_INVALID_ARGUMENT, "pairs / normal must not be null");
  MHIP_REQUIRE(N == 0 || mob_trans, MHIP_ERR_INVALID_ARGUMENT, "mob_trans must not be null");
  MHIP_REQUIRE(C < (1u << 30) && N < (1u << 31), MHIP_ERR_RUNTIME, "problem too large for 32-bit incidence entries");
  hipStream_t s = as_stream(stream);
  mhip_contact_op* op = new mhip_contact_op();
  op->last_stream = s;
  (void)hipGetDevice(&op->device);
  adopt_spare_workspaces(op);
  auto bail = [&](int e) {
    mhip_contact_op_destroy(op);
    return e;
  };
  op->kin = kin;
  op->rot = (kin != KIN_TRANS);
  if (int e = op->inc_ptr.reserve((N + 2) * sizeof(int32_t))) return bail(e);
  if (int e = op->cursor.reserve((N + 2) * sizeof(int32_t))) return bail(e);
  if (int e = op->inc.reserve((2 * C + 2) * sizeof(int32_t))) return bail(e);
  if (int e = op->vel.reserve((6 * N + 2) * sizeof(double))) return bail(e);
  if (int e = op->partials.reserve(((size_t)kRed * kStageStride + kRed * kFoldGroups + 64) * sizeof(double))) return bail(e);
  if (int e = op->state.reserve(sizeof(SolverState) + 64)) return bail(e);
  if (int e = op->scanws.reserve(scan_workspace_bytes(N + 1) + 64)) return bail(e);
  if (!op->host_state) {
    hipError_t he = hipHostMalloc(reinterpret_cast<void**>(&op->host_state), sizeof(SolverState) + 64);
    if (he != hipSuccess) return bail(fail(MHIP_ERR_HIP, "hipHostMalloc failed: %s", hipGetErrorString(he)));
  }
  *reinterpret_cast<int32_t*>(op->host_state + 1) = 0;  // length of the compact active lists: not known yet
  int32_t* deg = op->cursor.as<int32_t>();
  int* bad = reinterpret_cast<int*>(op->state.as<char>() + sizeof(SolverState));
  hipError_t he = hipMemsetAsync(deg, 0, (N + 1) * sizeof(int32_t), s);
  if (he == hipSuccess) he = hipMemsetAsync(op->state.ptr, 0, sizeof(SolverState) + 64, s);
  if (he == hipSuccess) he = hipMemsetAsync(op->vel.ptr, 0, (6 * N + 2) * sizeof(double), s);
  if (he != hipSuccess) return bail(fail(MHIP_ERR_HIP, "hipMemsetAsync failed: %s", hipGetErrorString(he)));
  const int2* p2 = reinterpret_cast<const int2*>(pairs);
  if (int e = op->body_mask.reserve((N + 2) * sizeof(unsigned long long))) return bail(e);
  if (int e = op->pos.reserve(2 * C + 16)) return bail(e);
  bool pos_built = false;
  bool sorted_rows = false;   // the list has the broad phase's form: rows sorted by the lower body, i < j
  if (C > 0) {
    // first the fast path's count (it validates the indices as well); a list of another form takes the general path
    if (int e = op->acnt.reserve((N + 2) * sizeof(int32_t))) return bail(e);
    if (int e = op->aptr.reserve((N + 2) * sizeof(int32_t))) return bail(e);
    int32_t* row_start = op->aptr.as<int32_t>();   // (both free until the first snapshot of the active lists)
    k_inc_count_sorted<<<grid_for(C), kBlock, 0, s>>>(C, N, p2, deg, row_start, bad);
    // (the degrees too before the one host read: deg -> tdeg, deg = tdeg + rows; meaningless if the list is of another
    //  form, and then not used)
    he = hipMemcpyAsync(op->acnt.ptr, deg, (N + 1) * sizeof(int32_t), hipMemcpyDeviceToDevice, s);
    if (he != hipSuccess) return bail(fail(MHIP_ERR_HIP, "hipMemcpyAsync failed: %s", hipGetErrorString(he)));
    k_inc_deg<<<grid_for(N), kBlock, 0, s>>>(N, op->acnt.as<int32_t>(), row_start, deg, bad);
    int hflags[3] = {0, 0, 0};
    he = hipMemcpyAsync(hflags, bad, 3 * sizeof(int), hipMemcpyDeviceToHost, s);
    if (he == hipSuccess) he = hipStreamSynchronize(s);
    if (he != hipSuccess) return bail(fail(MHIP_ERR_HIP, "pair validation failed: %s", hipGetErrorString(he)));
    if (hflags[0])
      return bail(fail(MHIP_ERR_INVALID_ARGUMENT, "pairs contain an index outside [0, %zu) or a self pair", N));
    sorted_rows = hflags[1] == 0 && hflags[2] == 0;
    if (!sorted_rows) {
      he = hipMemsetAsync(deg, 0, (N + 1) * sizeof(int32_t), s);
      if (he == hipSuccess) he = hipMemsetAsync(bad, 0, 3 * sizeof(int), s);
      if (he != hipSuccess) return bail(fail(MHIP_ERR_HIP, "hipMemsetAsync failed: %s", hipGetErrorString(he)));
      k_inc_count<<<grid_for(C), kBlock, 0, s>>>(C, N, p2, deg, bad);
    }
  }
  if (sorted_rows) {
    int32_t* tdeg = op->acnt.as<int32_t>();      // (the target counts; deg holds the lists' lengths: see above)
    int32_t* row_start = op->aptr.as<int32_t>();
    if (int e = exclusive_scan_i32(deg, op->inc_ptr.as<int32_t>(), N, op->scanws.ptr, s)) return bail(e);
    he = hipMemsetAsync(deg, 0, (N + 1) * sizeof(int32_t), s);   // now the targets' cursors
    if (he != hipSuccess) return bail(fail(MHIP_ERR_HIP, "hipMemsetAsync failed: %s", hipGetErrorString(he)));
    if (int e = op->sort_tmp.reserve((2 * C + 2) * sizeof(unsigned))) return bail(e);
    k_inc_fill_sorted<<<grid_for(C), kBlock, 0, s>>>(C, p2, op->inc_ptr.as<int32_t>(), tdeg, deg, row_start,
                                                    op->sort_tmp.as<unsigned>(), priority);
    k_inc_arrange<<<grid_for(N), kBlock, 0, s>>>(N, op->inc_ptr.as<int32_t>(), tdeg, op->sort_tmp.as<unsigned>(),
                                                op->inc.as<int32_t>(), op->pos.as<unsigned char>());
    pos_built = true;
  } else {
    if (int e = exclusive_scan_i32(deg, op->inc_ptr.as<int32_t>(), N, op->scanws.ptr, s)) return bail(e);
    he = hipMemcpyAsync(deg, op->inc_ptr.ptr, (N + 1) * sizeof(int32_t), hipMemcpyDeviceToDevice, s);
    if (he != hipSuccess) return bail(fail(MHIP_ERR_HIP, "hipMemcpyAsync failed: %s", hipGetErrorString(he)));
    if (C > 0) {
      k_inc_fill<<<grid_for(C), kBlock, 0, s>>>(C, p2, deg, op->inc.as<int32_t>(), priority);
      if (int e = op->sort_tmp.reserve((2 * C + 2) * sizeof(unsigned))) return bail(e);
      if (int e = op->sort_list.reserve((N + 32) * sizeof(int32_t))) return bail(e);
      if (int e = sort_segments_u32(N, op->inc_ptr.as<int32_t>(), op->inc.as<unsigned>(), op->sort_tmp.as<unsigned>(), 32,
                                    op->sort_list.as<int32_t>(), s))
        return bail(e);
      if (priority) k_clear_class_bit<<<grid_for(2 * C), kBlock, 0, s>>>(2 * C, op->inc.as<int32_t>());
    }
  }
  he = hipGetLastError();
  if (he != hipSuccess) return bail(fail(MHIP_ERR_HIP, "incidence build failed: %s", hipGetErrorString(he)));
  const int hw = (kin == KIN_RIGID) ? 6 : (kin == KIN_ROD ? 4 : 3);
  if (int e = op->half.reserve((2 * C + 2) * hw * sizeof(double))) return bail(e);
  if (kin == KIN_ROD) {
    if (int e = op->axis.reserve((3 * N + 2) * sizeof(double))) return bail(e);
    if (int e = op->omega.reserve((3 * N + 2) * sizeof(double))) return bail(e);
    he = hipMemsetAsync(op->omega.ptr, 0, (3 * N + 2) * sizeof(double), s);
    if (he != hipSuccess) return bail(fail(MHIP_ERR_HIP, "hipMemsetAsync failed: %s", hipGetErrorString(he)));
    if (N > 0) k_rod_axes<<<grid_for(N), kBlock, 0, s>>>(N, seg, op->axis.as<double>());
  }
  if (C > 0) {
    const size_t ne = 2 * C;
    int32_t* inc = op->inc.as<int32_t>();
    double* half = op->half.as<double>();
    if (kin == KIN_ROD) k_half_build<KIN_ROD><<<grid_for(ne), kBlock, 0, s>>>(ne, inc, normal, ra, rb, arc_s, arc_t, half);
    else if (kin == KIN_RIGID) k_half_build<KIN_RIGID><<<grid_for(ne), kBlock, 0, s>>>(ne, inc, normal, ra, rb, arc_s, arc_t, half);
    else k_half_build<KIN_TRANS><<<grid_for(ne), kBlock, 0, s>>>(ne, inc, normal, ra, rb, arc_s, arc_t, half);
  }
  he = hipGetLastError();
  if (he != hipSuccess) return bail(fail(MHIP_ERR_HIP, "half-edge build failed: %s", hipGetErrorString(he)));
  {
    // G lanes per body, each keeping U half-edge chains in flight.  Measured at 10^6 rods (mean degree 15.2, a third
    // of it active; MI355X), walking every entry: (16,1) 0.198 ms, (8,1) 0.164, (8,2) 0.154, (4,4) 0.157, (8,4) 0.150
    // per sweep; walking only the entries the activity masks flag: (8,4) 0.144, (4,4) 0.137, (2,8) 0.142, (1,8) 0.156.
    // With double-double sums (one rounding per sum): (4,4) 0.138 ms, (2,4) 0.151, (8,4) 0.182, (8,2) 0.156, (4,8) 0.149.
    // With the compact active lists (a sweep streams the active third of a list): (2,4) 0.103 ms, (4,4) 0.108, (8,4)
    // 0.162, (16,2) 0.255; whole step from the relaxed packing 26.6 ms with two lanes against 28.8 with four; with the
    // drift bookkeeping of the cold tier (2,2) 0.100, (2,3) 0.103, (2,4) 0.103, (2,6) 0.114.
    // The layout only moves time: the sums, hence the iterates, are the same for every G (tests).
    const double mean_deg = N ? 2.0 * (double)C / (double)N : 0.0;
    op->lanes_per_body = mean_deg <= 24.0 ? 2 : (mean_deg <= 48.0 ? 4 : (mean_deg <= 96.0 ? 8 : 16));
  }
  op->view = OpView{C, N, p2, normal, ra, rb, mob_trans, mob_rot, op->inc_ptr.as<int32_t>(), op->inc.as<int32_t>(),
                    op->half.as<double>(), op->vel.as<double>(), dt, 0, N, nullptr, arc_s, arc_t,
                    op->axis.as<double>(), op->omega.as<double>(), 0, 0, C, 0, 0, nullptr, nullptr,
                    nullptr, nullptr, nullptr, nullptr};
  if (N > 0 && C > 0 && !pos_built) {
    k_pos_build<<<grid_for(N), kBlock, 0, s>>>(N, op->inc_ptr.as<int32_t>(), op->inc.as<int32_t>(),
                                              op->pos.as<unsigned char>());
    he = hipGetLastError();
    if (he != hipSuccess) return bail(fail(MHIP_ERR_HIP, "slot table build failed: %s", hipGetErrorString(he)));
  }
  op->view.body_mask = op->body_mask.as<unsigned long long>();
  op->view.pos = op->pos.as<unsigned char>();
  // XCD-contiguous tiles (xcd_tile): 32 consecutive tiles per XCD inside windows of 256.  Measured at 10^6 rods:
  // FETCH_SIZE per launch 740 -> 637 MB (k_constraint) and 683 -> 621 MB (k_body), k_constraint 0.1305 -> 0.128 ms,
  // k_body unchanged.  mhip_contact_op_set_work_mapping overrides (0 = identity mapping).
  op->view.xcd_aware = 32;
  *handle = op;
  return MHIP_SUCCESS;
}

int mhip_contact_op_create(mhip_contact_op_t* handle, size_t num_constraints, size_t num_bodies, const int32_t* pairs,
                           const double* normal, const double* ra, const double* rb, const double* mob_trans,
                           const double* mob_rot, double dt, const double* priority, mhip_stream_t stream) {
  const int nrot = (ra != nullptr) + (rb != nullptr) + (mob_rot != nullptr);
  MHIP_REQUIRE(nrot == 0 || nrot == 3, MHIP_ERR_INVALID_ARGUMENT,
               "ra, rb and mob_rot must be given together (rigid bodies) or all be null (translation only)");
  return create_contact_op(handle, nrot == 3 ? KIN_RIGID : KIN_TRANS, num_constraints, num_bodies, pairs, normal, ra,
                           rb, nullptr, nullptr, nullptr, mob_trans, mob_rot, dt, priority, stream);
}

int mhip_contact_op_create_rods(mhip_contact_op_t* handle, size_t num_constraints, size_t num_bodies,
                                const int32_t* pairs, const double* normal, const double* arc_s, const double* arc_t,
                                const double* seg, const double* mob_trans, const double* mob_rot, double dt,
                                const double* priority, mhip_stream_t stream) {
  MHIP_REQUIRE(num_constraints == 0 || (arc_s && arc_t), MHIP_ERR_INVALID_ARGUMENT, "arclength arrays must not be null");
  MHIP_REQUIRE(num_bodies == 0 || (seg && mob_rot), MHIP_ERR_INVALID_ARGUMENT, "seg / mob_rot must not be null");
  return create_contact_op(handle, KIN_ROD, num_constraints, num_bodies, pairs, normal, nullptr, nullptr, arc_s, arc_t,
                           seg, mob_trans, mob_rot, dt, priority, stream);
}

// Same contact list, new geometry: the incidence index, the slot table and the list order stay (they depend on the pairs
// only; the priority classes keep the order they had when the operator was built, which affects nothing but locality
// and the summation order), the half-edge records, the rod axes and the views of the per-contact arrays are redone.
static int refresh_contact_op(mhip_contact_op_t op, const double* normal, const double* ra, const double* rb,
                              const double* arc_s, const double* arc_t, const double* seg, mhip_stream_t stream) {
  TraceRange trace_range("ContactOperator::refresh");
  MHIP_REQUIRE(op != nullptr, MHIP_ERR_INVALID_ARGUMENT, "operator handle is null");
  const size_t C = op->view.C, N = op->view.N;
  MHIP_REQUIRE(C == 0 || normal, MHIP_ERR_INVALID_ARGUMENT, "normal must not be null");
  if (op->kin == KIN_RIGID) MHIP_REQUIRE(C == 0 || (ra && rb), MHIP_ERR_INVALID_ARGUMENT, "ra / rb must not be null");
  if (op->kin == KIN_ROD)
    MHIP_REQUIRE((C == 0 || (arc_s && arc_t)) && (N == 0 || seg), MHIP_ERR_INVALID_ARGUMENT,
                 "arclength arrays / seg must not be null");
  MHIP_REQUIRE(!op->stage.active, MHIP_ERR_RUNTIME, "a staged solve is in progress");
  hipStream_t s = as_stream(stream);
  op->last_stream = s;
  if (op->kin == KIN_ROD && N > 0) {
    MHIP_HIP(hipMemsetAsync(op->omega.ptr, 0, (3 * N + 2) * sizeof(double), s));
    k_rod_axes<<<grid_for(N), kBlock, 0, s>>>(N, seg, op->axis.as<double>());
  }
  if (op->view.vel == op->vel.as<double>()) MHIP_HIP(hipMemsetAsync(op->view.vel, 0, 6 * N * sizeof(double), s));
  if (C > 0) {
    const size_t ne = 2 * C;
    const int32_t* inc = op->inc.as<int32_t>();
    double* half = op->half.as<double>();
    if (op->kin == KIN_ROD) k_half_build<KIN_ROD><<<grid_for(ne), kBlock, 0, s>>>(ne, inc, normal, ra, rb, arc_s, arc_t, half);
    else if (op->kin == KIN_RIGID) k_half_build<KIN_RIGID><<<grid_for(ne), kBlock, 0, s>>>(ne, inc, normal, ra, rb, arc_s, arc_t, half);
    else k_half_build<KIN_TRANS><<<grid_for(ne), kBlock, 0, s>>>(ne, inc, normal, ra, rb, arc_s, arc_t, half);
  }
  MHIP_LAUNCH_CHECK();
  op->view.aptr = nullptr;
  op->view.normal = normal;
  op->view.ra = ra;
  op->view.rb = rb;
  op->view.arc_s = arc_s;
  op->view.arc_t = arc_t;
  return MHIP_SUCCESS;
}

int mhip_contact_op_refresh(mhip_contact_op_t op, const double* normal, const double* ra, const double* rb,
                            mhip_stream_t stream) {
  MHIP_REQUIRE(op != nullptr, MHIP_ERR_INVALID_ARGUMENT, "operator handle is null");
  MHIP_REQUIRE(op->kin != KIN_ROD, MHIP_ERR_INVALID_ARGUMENT, "a rod operator is refreshed with mhip_contact_op_refresh_rods");
  return refresh_contact_op(op, normal, op->kin == KIN_RIGID ? ra : nullptr, op->kin == KIN_RIGID ? rb : nullptr, nullptr,
                            nullptr, nullptr, stream);
}

int mhip_contact_op_refresh_rods(mhip_contact_op_t op, const double* normal, const double* arc_s, const double* arc_t,
                                 const double* seg, mhip_stream_t stream) {
  MHIP_REQUIRE(op != nullptr, MHIP_ERR_INVALID_ARGUMENT, "operator handle is null");
  MHIP_REQUIRE(op->kin == KIN_ROD, MHIP_ERR_INVALID_ARGUMENT, "not a rod operator");
  return refresh_contact_op(op, normal, nullptr, nullptr, arc_s, arc_t, seg, stream);
}

int mhip_contact_op_destroy(mhip_contact_op_t op) {
  if (!op) return MHIP_SUCCESS;
  // whatever still runs on the operator's stream reads these buffers: wait before they can be handed on
  // (hipFree, which this replaces, synchronised the whole device)
  (void)hipStreamSynchronize(op->last_stream);
  OpWorkspaces w;
  for (int k = 0; k < kOpBuffers; ++k) w.buf[k] = *op_buffers(op, k);
  w.host_state = op->host_state;
  w.events.swap(op->events);
  const int home = op->device;  // the device the buffers were allocated on, whatever device is current now
  delete op;
  {
    std::lock_guard<std::mutex> lock(g_spare_mutex);
    if (!g_spare.held && home >= 0) {
      g_spare = std::move(w);
      g_spare.held = true;
      g_spare.device = home;
      return MHIP_SUCCESS;
    }
  }
  free_workspaces(w);
  return MHIP_SUCCESS;
}

int mhip_release_cached_workspaces(void) {
  std::lock_guard<std::mutex> lock(g_spare_mutex);
  free_workspaces(g_spare);
  return MHIP_SUCCESS;
}

int mhip_contact_op_apply(mhip_contact_op_t op, const double* x, double* y, mhip_stream_t stream) {
  MHIP_REQUIRE(op != nullptr, MHIP_ERR_INVALID_ARGUMENT, "operator handle is null");
  MHIP_REQUIRE(op->view.C == 0 || (x && y), MHIP_ERR_INVALID_ARGUMENT, "x / y must not be null");
  hipStream_t s = as_stream(stream);
  const Space none{MHIP_SPACE_UNCONSTRAINED, 0, 0};
  if (int e = op_launch_body(op, X_APPLY, x, x, nullptr, nullptr, none, s)) return e;
  // APPLY reads the iterate from X0 and writes y through G1
  return op_launch_constraint(op, X_APPLY, const_cast<double*>(x), nullptr, nullptr, y, nullptr, none, 0,
                              grid_for(op->view.C), s);
}

int mhip_contact_op_sizes(mhip_contact_op_t op, size_t* num_constraints, size_t* num_bodies) {
  MHIP_REQUIRE(op != nullptr, MHIP_ERR_INVALID_ARGUMENT, "operator handle is null");
  if (num_constraints) *num_constraints = op->view.C;
  if (num_bodies) *num_bodies = op->view.N;
  return MHIP_SUCCESS;
}

int mhip_contact_op_set_work_mapping(mhip_contact_op_t op, int xcd_tile, int lanes_per_body) {
  MHIP_REQUIRE(op != nullptr, MHIP_ERR_INVALID_ARGUMENT, "operator handle is null");
  MHIP_REQUIRE(xcd_tile >= -1 && xcd_tile <= 4096, MHIP_ERR_INVALID_ARGUMENT,
               "xcd_tile must be in [0, 4096] (or -1 to keep the current value), got %d", xcd_tile);
  MHIP_REQUIRE(lanes_per_body == -1 || lanes_per_body == 1 || lanes_per_body == 2 || lanes_per_body == 4 ||
                   lanes_per_body == 8 || lanes_per_body == 16,
               MHIP_ERR_INVALID_ARGUMENT, "lanes_per_body must be 1, 2, 4, 8 or 16 (or -1 to keep), got %d", lanes_per_body);
  MHIP_REQUIRE(!op->stage.active, MHIP_ERR_RUNTIME, "a staged solve is in progress");
  if (xcd_tile >= 0) op->view.xcd_aware = xcd_tile;
  if (lanes_per_body > 0) op->lanes_per_body = lanes_per_body;
  return MHIP_SUCCESS;
}

int mhip_contact_op_set_drift_source(mhip_contact_op_t op, int source) {
  MHIP_REQUIRE(op != nullptr, MHIP_ERR_INVALID_ARGUMENT, "operator handle is null");
  MHIP_REQUIRE(source >= 0 && source <= 2, MHIP_ERR_INVALID_ARGUMENT, "drift source must be 0 (by size), 1 (rows) or 2 (registers), got %d", source);
  op->drift_source = source;
  return MHIP_SUCCESS;
}

int mhip_contact_op_get_drift_source(mhip_contact_op_t op, int* source) {
  MHIP_REQUIRE(op != nullptr && source != nullptr, MHIP_ERR_INVALID_ARGUMENT, "null argument");
  *source = op_drift_source(op);
  return MHIP_SUCCESS;
}

int mhip_contact_op_set_profiling(mhip_contact_op_t op, int enable) {
  MHIP_REQUIRE(op != nullptr, MHIP_ERR_INVALID_ARGUMENT, "operator handle is null");
  op->profile = enable != 0;
  op->body_ms = op->constraint_ms = 0.0;
  op->timed_iterations = 0;
  return MHIP_SUCCESS;
}

int mhip_contact_op_get_profile(mhip_contact_op_t op, double* body_ms, double* constraint_ms, size_t* iterations) {
  MHIP_REQUIRE(op != nullptr, MHIP_ERR_INVALID_ARGUMENT, "operator handle is null");
  if (body_ms) *body_ms = op->body_ms;
  if (constraint_ms) *constraint_ms = op->constraint_ms;
  if (iterations) *iterations = op->timed_iterations;
  return MHIP_SUCCESS;
}

int mhip_contact_op_body_velocity(mhip_contact_op_t op, const double** velocity) {
  MHIP_REQUIRE(op != nullptr && velocity != nullptr, MHIP_ERR_INVALID_ARGUMENT, "null argument");
  if (op->kin == KIN_ROD) {
    // rod-compressed rows hold (U, W x u); assemble (U, W) in stream order after the last sweep
    const size_t N = op->view.N;
    if (int e = op->vel_out.reserve((6 * N + 2) * sizeof(double))) return e;
    if (N > 0) {
      k_assemble_velocity<<<grid_for(N), kBlock, 0, op->last_stream>>>(N, op->view.vel, op->omega.as<double>(),
                                                                      op->vel_out.as<double>());
      MHIP_LAUNCH_CHECK();
    }
    *velocity = op->vel_out.as<double>();
    return MHIP_SUCCESS;
  }
  *velocity = op->view.vel;
  return MHIP_SUCCESS;
}

int mhip_bbpgd_solve_contact(mhip_contact_op_t op, const double* q, const mhip_space* space,
                             const mhip_pgd_config* config, double* x, double* g, double* x_tmp, double* g_tmp,
                             mhip_solve_result* result, mhip_stream_t stream) {
  TraceRange trace_range("solve_cqpp (fused BBPGD)");
  MHIP_REQUIRE(op != nullptr && result != nullptr, MHIP_ERR_INVALID_ARGUMENT, "null handle / result");
  if (int e = check_config(config)) return e;
  Space sp;
  if (int e = to_space(space, &sp)) return e;
  const size_t C = op->view.C;
  hipStream_t s = as_stream(stream);
  if (C == 0) {  // empty reduce_max: the identity survives (Kokkos::Max), residual = lowest / 1e-6
    result->num_iters = 0;
    result->residual = config->residual_kind == MHIP_RESIDUAL_PROJECTED_DIFF ? kLowest / kSmallStep : kLowest;
    result->converged = 1;
    return MHIP_SUCCESS;
  }
  MHIP_REQUIRE(q && x && g && x_tmp && g_tmp, MHIP_ERR_INVALID_ARGUMENT, "solver vectors must not be null");
  MHIP_REQUIRE(x != x_tmp && g != g_tmp && x != g, MHIP_ERR_INVALID_ARGUMENT, "solver vectors must not alias");
  SolverState* st = op->state.as<SolverState>();
  double* parts = op->partials.as<double>();
  const unsigned cgrid = constraint_grid(C);
  const int rk = config->residual_kind;
  if (int e = op->iterate.reserve(2 * (C + 1) * sizeof(double2))) return e;
  double* P0 = op->iterate.as<double>();
  double* P1 = P0 + 2 * C;
  if (op->view.body_mask) MHIP_HIP(hipMemsetAsync(op->view.body_mask, 0x00, op->view.N * sizeof(unsigned long long), s));
  op->view.aptr = nullptr;  // no snapshot of the active entries yet (the init sweep writes the masks)
  // initialize: x_tmp = x ; g_tmp = A x_tmp + q ; residual ; step = 1/res   (the pair lands packed in P0)
  if (int e = op_launch_body(op, X_INIT, x, x, nullptr, nullptr, sp, s)) return e;
  if (int e = op_launch_constraint(op, X_INIT, P0, P1, x, nullptr, q, sp, rk, cgrid, s, true)) return e;
  {
    unsigned np = cgrid;
    size_t ps = cgrid;
    double* pp = parts;
    fold_partials(np, ps, pp, st, 0, s);
    k_finalize<X_INIT><<<1, final_block(np), 0, s>>>((int)np, pp, 1, ps, st, rk, config->tol, config->max_iters);
  }
  MHIP_LAUNCH_CHECK();
  unsigned enqueued = 0, chunk = 8, last_todo = 0, iter_before = 0;
  const bool prof = op->profile;
  if (prof && op->events.empty()) {
    op->events.resize(3 * 64);
    for (auto& ev : op->events) MHIP_HIP(hipEventCreate(&ev));
  }
  // cold tier (see "Cold tier"): from the first snapshot on, where the problem is of the kind it covers
  mhip_contact_op::Tier& tier = op->tier;
  tier.disabled = !tier_eligible(op, sp, rk) || op->tiering == 0;
  tier.tiered_iterations = tier.retiers = tier.wakeups = 0;
  tier.hot_sum = 0.0;
  TierPairs cur{P0, P1, q};
  // whatever ends the solve early, the operator must not be left in the tier numbering
  struct TierGuard {
    mhip_contact_op* op;
    ~TierGuard() {
      if (!op->tier.active) {
        op->view.vel_alt = nullptr;
        op->view.drift = nullptr;
        op->tier.tracking = false;
        return;
      }
      op->tier.tracking = false;
      const size_t C2 = op->view.C;
      const TierGeo geo = tier_geo_at(op->tier.geo[op->tier.set].ptr, C2);
      k_tier_remap_inc<<<grid_for(2 * C2), kBlock, 0, op->last_stream>>>(2 * C2, op->inc.as<int32_t>(), geo.orig);
      (void)hipStreamSynchronize(op->last_stream);
      const int xa = op->view.xcd_aware;
      op->view = op->tier.saved;
      op->view.xcd_aware = xa;
      op->view.aptr = nullptr;
      op->tier.active = false;
    }
  } tier_guard{op};
  PollPlan plan;
  for (;;) {
    MHIP_HIP(hipMemcpyAsync(op->host_state, st, sizeof(SolverState), hipMemcpyDeviceToHost, s));
    MHIP_HIP(hipStreamSynchronize(s));
    if (prof && last_todo) {
      // iterations of the last chunk that did real work: those that advanced iter, plus the converging one
      unsigned eff = op->host_state->iter - iter_before + ((op->host_state->converged && op->host_state->iter < config->max_iters) ? 1u : 0u);
      if (eff > last_todo) eff = last_todo;
      for (unsigned k = 0; k < eff; k += kProfileStride) {  // every kProfileStride-th iteration is bracketed
        float a = 0.f, b = 0.f;
        MHIP_HIP(hipEventElapsedTime(&a, op->events[3 * k], op->events[3 * k + 1]));
        MHIP_HIP(hipEventElapsedTime(&b, op->events[3 * k + 1], op->events[3 * k + 2]));
        op->body_ms += a;
        op->constraint_ms += b;
        op->timed_iterations += 1;
      }
    }
    if (tier.active && last_todo) {  // the iterations of the last chunk that ran (the launches after `done` were idle)
      unsigned ran = op->host_state->iter - iter_before + ((op->host_state->converged && op->host_state->iter < config->max_iters) ? 1u : 0u);
      if (ran > last_todo) ran = last_todo;
      tier.tiered_iterations += ran;
      tier.hot_sum += ran * (static_cast<double>(tier.H) / static_cast<double>(C));
    }
    if (op->host_state->done == 2) {
      // paused before a step outside [0, finite]: the sleepers' exact gradients are needed -- leave the tiers for good
      if (int e = tier_release(op, cur, false, P0, P1, q, nullptr, nullptr, nullptr, nullptr, s)) return e;
      tier.disabled = true;
      op->host_state->done = 0;
      MHIP_HIP(hipMemcpyAsync(&st->done, &op->host_state->done, sizeof(int), hipMemcpyHostToDevice, s));
      // the launches that were enqueued behind the pause did nothing: they do not count
      enqueued = op->host_state->iter;
    }
    if (op->host_state->done || enqueued >= config->max_iters) break;
#if MHIP_SNAPSHOT_AT_INIT
    // the masks of the INITIAL iterate exist (the init sweep wrote them): a first snapshot of the active lists at the
    // first poll -- unless the problem was solved already -- so that iterations 1 ... 8 stream compact lists too instead
    // of walking every body's mask (round 3: k_body at 130-196 us in those iterations against 110 with a snapshot); what
    // becomes active since takes the per-body path
    if (enqueued == 0)
      if (int e = op_snapshot_active(op, s)) return e;
#endif
    // cold tier: the drift bookkeeping starts with the first iteration, the first classification comes with the first
    // snapshot (short solves -- a relaxed packing needs ~100 iterations -- get their tiers early)
    const bool light = plan.light_poll();
    if (!light) {
      const size_t retiers_before = tier.retiers;
      if (!tier.disabled && (enqueued >= kSnapshotAfter || !tier.tracking)) {
        const unsigned left = config->max_iters - enqueued;
        if (int e = tier_update(op, cur, op->host_state->iter, left < chunk ? left : chunk, C, /*pingpong=*/true, s)) return e;
      }
      if (enqueued >= kSnapshotAfter)
        if (int e = op_snapshot_active(op, s)) return e;
      plan.full_poll_done(enqueued, *reinterpret_cast<const int32_t*>(op->host_state + 1),
                          tier.retiers != retiers_before, !tier.disabled && !tier.active);
    }
    iter_before = op->host_state->iter;
    const unsigned stretch = plan.stretch(chunk, op->host_state->iter, op->host_state->residual, config->tol, C);
    const unsigned todo = (config->max_iters - enqueued < stretch) ? config->max_iters - enqueued : stretch;
#ifdef MHIP_TIER_DEBUG
    fprintf(stderr, "poll: iter %u residual %.3e light %d quiet %u stretch %u (regular %u) snapshot entries %d\n",
            op->host_state->iter, op->host_state->residual, (int)light, plan.quiet, stretch, chunk,
            *reinterpret_cast<const int32_t*>(op->host_state + 1));
#endif
    for (unsigned k = 0; k < todo; ++k) {
      const bool pk = prof && (k % kProfileStride == 0);
      if (pk) MHIP_HIP(hipEventRecord(op->events[3 * k], s));
      if (int e = op_launch_body(op, X_SOLVE, cur.P0, cur.P1, nullptr, nullptr, sp, s, true)) return e;
      if (pk) MHIP_HIP(hipEventRecord(op->events[3 * k + 1], s));
      unsigned np = cgrid;
      size_t ps = cgrid;
      if (tier.active) {
        if (int e = op_launch_constraint_tiered(op, cur, sp, rk, &np, s)) return e;
        ps = kStageStride;
      } else {
        if (int e = op_launch_constraint(op, X_SOLVE, cur.P0, cur.P1, nullptr, nullptr, q, sp, rk, cgrid, s, true)) return e;
      }
      if (pk) MHIP_HIP(hipEventRecord(op->events[3 * k + 2], s));
      double* pp = parts;
      if (MHIP_FOLD_FINALIZE && np > MHIP_FOLD_ABOVE) {
        // (folded records behind the kRed planes of partials, as fold_partials places them; the ticket word lives in
        // the spare words behind the solver state and is left at zero by every launch that finalizes)
        k_fold_finalize<X_SOLVE><<<kFoldGroups, MHIP_FOLD_BLOCK, 0, s>>>(
            (int)np, ps, pp, pp + kRed * ps, reinterpret_cast<unsigned*>(op->state.as<char>() + sizeof(SolverState) + 16),
            st, rk, config->tol, config->max_iters, tier.active ? op->tiering : 0,
            tier.active ? op->view.tier_counters : nullptr);
      } else {
        fold_partials(np, ps, pp, st, 1, s);
        k_finalize<X_SOLVE><<<1, final_block(np), 0, s>>>((int)np, pp, 1, ps, st, rk, config->tol, config->max_iters,
                                                          tier.active ? op->tiering : 0,
                                                          tier.active ? op->view.tier_counters : nullptr);
      }
      MHIP_LAUNCH_CHECK();
    }
    enqueued += todo;
    last_todo = todo;
    if (chunk < 64) chunk *= 2;
  }
  if (tier.active) {
    if (int e = tier_release(op, cur, true, P0, P1, q, x, g, x_tmp, g_tmp, s)) return e;
  } else {
    if (int e = tier_stop_tracking(op, s)) return e;
    k_finish_packed<<<grid_for(C), kBlock, 0, s>>>(C, st, reinterpret_cast<const double2*>(P0),
                                                   reinterpret_cast<const double2*>(P1), x, g, x_tmp, g_tmp);
    MHIP_LAUNCH_CHECK();
  }
  // rods: the angular velocities of the final iterate (the iterations' sweeps keep only the (U, Z) rows)
  if (op->kin == KIN_ROD)
    if (int e = op_launch_body(op, X_APPLY, x, nullptr, nullptr, nullptr, sp, s)) return e;
  MHIP_HIP(hipStreamSynchronize(s));
  result->num_iters = op->host_state->iter;
  result->residual = op->host_state->residual;
  result->converged = op->host_state->converged;
  return MHIP_SUCCESS;
}

#ifdef MHIP_EXP_COUNT_MM
int mhip_debug_counters(unsigned long long* out4) {
  MHIP_HIP(hipMemcpyFromSymbol(out4, HIP_SYMBOL(g_dbg), 4 * sizeof(unsigned long long)));
  const unsigned long long z[4] = {0, 0, 0, 0};
  MHIP_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_dbg), z, sizeof(z)));
  return MHIP_SUCCESS;
}
#endif
/* cold-tier statistics of the last mhip_bbpgd_solve_contact on this operator (all zero when the solve did not tier) */
int mhip_contact_op_tier_stats(mhip_contact_op_t op, size_t* tiered_iterations, double* mean_hot_fraction,
                               size_t* renumberings, size_t* wakeups) {
  MHIP_REQUIRE(op != nullptr, MHIP_ERR_INVALID_ARGUMENT, "operator handle is null");
  const mhip_contact_op::Tier& t = op->tier;
  if (tiered_iterations) *tiered_iterations = t.tiered_iterations;
  if (mean_hot_fraction) *mean_hot_fraction = t.tiered_iterations ? t.hot_sum / static_cast<double>(t.tiered_iterations) : 0.0;
  if (renumberings) *renumberings = t.retiers;
  if (wakeups) *wakeups = t.wakeups;
  return MHIP_SUCCESS;
}

/* 1 (default): mhip_bbpgd_solve_contact may keep inactive contacts in a cold tier; 0: every contact is swept every
 * iteration; 2 (for tests): as 1, and the solve is paused after its first tiered iteration as it would be before a BB
 * step outside [0, finite], i.e. it leaves the tiers and goes on untiered; 3 (for tests): as 1 whatever the number of
 * contacts (so is 2).  Time only: the iterates are the same bits. */
int mhip_contact_op_set_tiering(mhip_contact_op_t op, int mode) {
  MHIP_REQUIRE(op != nullptr, MHIP_ERR_INVALID_ARGUMENT, "operator handle is null");
  MHIP_REQUIRE(mode >= 0 && mode <= 3, MHIP_ERR_INVALID_ARGUMENT, "tiering mode must be 0 ... 3, got %d", mode);
  op->tiering = mode;
  return MHIP_SUCCESS;
}

int mhip_bbpgd_solve_contact_friction(mhip_contact_op_t op, const double* sep, double mu,
                                      const mhip_pgd_config* config, double* p, double* g,
                                      mhip_solve_result* result, mhip_stream_t stream) {
  TraceRange trace_range("solve_friction_contact (extension)");
  MHIP_REQUIRE(op != nullptr && result != nullptr, MHIP_ERR_INVALID_ARGUMENT, "null handle / result");
  if (int e = check_config(config)) return e;
  MHIP_REQUIRE(config->residual_kind == MHIP_RESIDUAL_PROJECTED_DIFF, MHIP_ERR_INVALID_ARGUMENT,
               "the friction extension supports the projected-difference residual only");
  MHIP_REQUIRE(mu >= 0.0 && mu == mu, MHIP_ERR_INVALID_ARGUMENT, "friction coefficient must be >= 0, got %g", mu);
  MHIP_REQUIRE(op->kin == KIN_RIGID, MHIP_ERR_INVALID_ARGUMENT,
               "friction needs the vector-arm operator (mhip_contact_op_create with ra, rb, mob_rot)");
  const size_t C = op->view.C;
  hipStream_t s = as_stream(stream);
  if (C == 0) {
    result->num_iters = 0;
    result->residual = kLowest / kSmallStep;
    result->converged = 1;
    return MHIP_SUCCESS;
  }
  MHIP_REQUIRE(sep && p && g, MHIP_ERR_INVALID_ARGUMENT, "solver vectors must not be null");
  MHIP_REQUIRE(p != g, MHIP_ERR_INVALID_ARGUMENT, "solver vectors must not alias");
  MHIP_REQUIRE((reinterpret_cast<uintptr_t>(op->view.half) & 15) == 0, MHIP_ERR_RUNTIME, "misaligned records");
  if (int e = op->iterate.reserve(2 * (6 * C + 2) * sizeof(double))) return e;
  double* P0 = op->iterate.as<double>();
  double* P1 = P0 + 6 * C;
  SolverState* st = op->state.as<SolverState>();
  double* parts = op->partials.as<double>();
  const unsigned cgrid = constraint_grid(C);
  const int G = 8;  // lanes per body of the body sweep, two half-edge chains each
  const unsigned bgrid = grid_exact(op->view.body_count * (size_t)G);
  op->last_stream = s;
  auto finalize = [&](bool init) {
    unsigned np = cgrid;
    size_t ps = cgrid;
    double* pp = parts;
    fold_partials(np, ps, pp, st, init ? 0 : 1, s);
    if (init)
      k_finalize<X_INIT><<<1, final_block(np), 0, s>>>((int)np, pp, 1, ps, st, config->residual_kind, config->tol,
                                                       config->max_iters);
    else
      k_finalize<X_SOLVE><<<1, final_block(np), 0, s>>>((int)np, pp, 1, ps, st, config->residual_kind, config->tol,
                                                        config->max_iters);
  };
  if (op->view.body_count > 0) k_body_friction<true, 8, 2><<<bgrid, kBlock, 0, s>>>(op->view, st, P0, P1, p, mu);
  k_constraint_friction<true><<<cgrid, kBlock, 0, s>>>(op->view, st, P0, P1, p, sep, mu, parts);
  finalize(true);
  MHIP_LAUNCH_CHECK();
  unsigned enqueued = 0, chunk = 8;
  for (;;) {
    MHIP_HIP(hipMemcpyAsync(op->host_state, st, sizeof(SolverState), hipMemcpyDeviceToHost, s));
    MHIP_HIP(hipStreamSynchronize(s));
    if (op->host_state->done || enqueued >= config->max_iters) break;
    const unsigned todo = (config->max_iters - enqueued < chunk) ? config->max_iters - enqueued : chunk;
    for (unsigned k = 0; k < todo; ++k) {
      if (op->view.body_count > 0)
        k_body_friction<false, 8, 2><<<bgrid, kBlock, 0, s>>>(op->view, st, P0, P1, nullptr, mu);
      k_constraint_friction<false><<<cgrid, kBlock, 0, s>>>(op->view, st, P0, P1, nullptr, sep, mu, parts);
      finalize(false);
    }
    MHIP_LAUNCH_CHECK();
    enqueued += todo;
    if (chunk < 64) chunk *= 2;
  }
  k_finish_friction<<<grid_for(C), kBlock, 0, s>>>(C, st, P0, P1, p, g);
  MHIP_LAUNCH_CHECK();
  MHIP_HIP(hipStreamSynchronize(s));
  result->num_iters = op->host_state->iter;
  result->residual = op->host_state->residual;
  result->converged = op->host_state->converged;
  return MHIP_SUCCESS;
}

/* BUILD EXTENSION (parity unpinned): the same cone complementarity problem by APGD -- Nesterov-accelerated projected
 * gradient descent with adaptive restart and a backtracked curvature estimate (Mazhar et al. 2015), one operator
 * application per iteration (see k_constraint_friction_apgd).  Same arguments, result and stopping rule as
 * mhip_bbpgd_solve_contact_friction; num_iters counts every sweep, refused steps included. */
int mhip_apgd_solve_contact_friction(mhip_contact_op_t op, const double* sep, double mu, const mhip_pgd_config* config,
                                     double* p, double* g, mhip_solve_result* result, mhip_stream_t stream) {
  TraceRange trace_range("solve_friction_contact, APGD (extension)");
  MHIP_REQUIRE(op != nullptr && result != nullptr, MHIP_ERR_INVALID_ARGUMENT, "null handle / result");
  if (int e = check_config(config)) return e;
  MHIP_REQUIRE(config->residual_kind == MHIP_RESIDUAL_PROJECTED_DIFF, MHIP_ERR_INVALID_ARGUMENT,
               "the friction extension supports the projected-difference residual only");
  MHIP_REQUIRE(mu >= 0.0 && mu == mu, MHIP_ERR_INVALID_ARGUMENT, "friction coefficient must be >= 0, got %g", mu);
  MHIP_REQUIRE(op->kin == KIN_RIGID, MHIP_ERR_INVALID_ARGUMENT,
               "friction needs the vector-arm operator (mhip_contact_op_create with ra, rb, mob_rot)");
  const size_t C = op->view.C;
  hipStream_t s = as_stream(stream);
  if (C == 0) {
    result->num_iters = 0;
    result->residual = kLowest / kSmallStep;
    result->converged = 1;
    return MHIP_SUCCESS;
  }
  MHIP_REQUIRE(sep && p && g, MHIP_ERR_INVALID_ARGUMENT, "solver vectors must not be null");
  MHIP_REQUIRE(p != g, MHIP_ERR_INVALID_ARGUMENT, "solver vectors must not alias");
  MHIP_REQUIRE((reinterpret_cast<uintptr_t>(op->view.half) & 15) == 0, MHIP_ERR_RUNTIME, "misaligned records");
  if (int e = op->iterate.reserve((3 * (6 * C + 2) + 16) * sizeof(double))) return e;
  ApgdBufs B;
  B.P[0] = op->iterate.as<double>();
  B.P[1] = B.P[0] + 6 * C + 2;
  B.P[2] = B.P[1] + 6 * C + 2;
  ApgdState* as = reinterpret_cast<ApgdState*>(B.P[2] + 6 * C + 2);
  SolverState* st = op->state.as<SolverState>();
  double* parts = op->partials.as<double>();
  const unsigned cgrid = constraint_grid(C);
  static_assert(kApgdRed * kMaxConstraintGrid <= kRed * (int)kStageStride, "partials buffer holds the APGD records");
  const int G = 8;
  const unsigned bgrid = grid_exact(op->view.body_count * (size_t)G);
  op->last_stream = s;
  // p_0, g_0 = N p_0 + q and the initial residual: the INIT sweeps of the BBPGD extension (buffer 0 receives them)
  if (op->view.body_count > 0) k_body_friction<true, 8, 2><<<bgrid, kBlock, 0, s>>>(op->view, st, B.P[0], B.P[1], p, mu);
  k_constraint_friction<true><<<cgrid, kBlock, 0, s>>>(op->view, st, B.P[0], B.P[1], p, sep, mu, parts);
  {
    unsigned np = cgrid;
    size_t ps = cgrid;
    double* pp = parts;
    fold_partials(np, ps, pp, st, 0, s);
    k_finalize<X_INIT><<<1, final_block(np), 0, s>>>((int)np, pp, 1, ps, st, config->residual_kind, config->tol,
                                                     config->max_iters);
  }
  k_apgd_begin<<<1, 1, 0, s>>>(st, as);
  MHIP_LAUNCH_CHECK();
  unsigned enqueued = 0, chunk = 8;
  for (;;) {
    MHIP_HIP(hipMemcpyAsync(op->host_state, st, sizeof(SolverState), hipMemcpyDeviceToHost, s));
    MHIP_HIP(hipStreamSynchronize(s));
    if (op->host_state->done || enqueued >= config->max_iters) break;
    const unsigned todo = (config->max_iters - enqueued < chunk) ? config->max_iters - enqueued : chunk;
    for (unsigned k = 0; k < todo; ++k) {
      if (op->view.body_count > 0) k_body_friction_apgd<8, 2><<<bgrid, kBlock, 0, s>>>(op->view, st, as, B, mu);
      k_constraint_friction_apgd<<<cgrid, kBlock, 0, s>>>(op->view, st, as, B, sep, mu, parts);
      k_apgd_finalize<<<1, kFinalBlock, 0, s>>>((int)cgrid, parts, st, as, config->tol, config->max_iters);
    }
    MHIP_LAUNCH_CHECK();
    enqueued += todo;
    if (chunk < 64) chunk *= 2;
  }
  k_finish_friction_apgd<<<grid_for(C), kBlock, 0, s>>>(C, st, as, B, p, g);
  MHIP_LAUNCH_CHECK();
  MHIP_HIP(hipStreamSynchronize(s));
  result->num_iters = op->host_state->iter;
  result->residual = op->host_state->residual;
  result->converged = op->host_state->converged;
  return MHIP_SUCCESS;
}

int mhip_solve_small_cqpp_batch(size_t batch, int n, const double* A, const double* q, const mhip_space* space,
                                const mhip_pgd_config* config, double* x, double* grad, unsigned* num_iters,
                                double* residual, int* converged, mhip_stream_t stream) {
  if (int e = check_config(config)) return e;
  Space sp;
  if (int e = to_space(space, &sp)) return e;
  MHIP_REQUIRE(n >= 1 && n <= kSmallMax, MHIP_ERR_INVALID_ARGUMENT, "small-problem size must be in [1, %d], got %d",
               kSmallMax, n);
  if (batch == 0) return MHIP_SUCCESS;
  MHIP_REQUIRE(A && q && x && grad && num_iters && residual && converged, MHIP_ERR_INVALID_ARGUMENT,
               "null argument");
  k_small_cqpp<<<grid_exact(batch, 64), 64, 0, as_stream(stream)>>>(batch, n, A, q, sp, config->residual_kind,
                                                                   config->max_iters, config->tol, x, grad, num_iters,
                                                                   residual, converged);
  MHIP_LAUNCH_CHECK();
  return MHIP_SUCCESS;
}

int mhip_scrap_bbpgd_solve_contact(mhip_contact_op_t op, const double* sep, double max_allowable_overlap,
                                   unsigned max_iterations, double* lam, double* lam_tmp, double* g, double* g_tmp,
                                   mhip_solve_result* result, double* max_speed, mhip_stream_t stream) {
  TraceRange trace_range("resolve_collisions");
  MHIP_REQUIRE(op != nullptr && result != nullptr, MHIP_ERR_INVALID_ARGUMENT, "null handle / result");
  const size_t C = op->view.C;
  hipStream_t s = as_stream(stream);
  result->num_iters = 0;
  result->residual = -1.0;  // maximum_abs_projected_sep's initial value (:617)
  result->converged = 1;
  if (max_speed) *max_speed = 0.0;
  if (C == 0) return MHIP_SUCCESS;
  MHIP_REQUIRE(sep && lam && lam_tmp && g && g_tmp, MHIP_ERR_INVALID_ARGUMENT, "solver vectors must not be null");
  MHIP_REQUIRE(lam != lam_tmp && g != g_tmp && lam != g, MHIP_ERR_INVALID_ARGUMENT, "solver vectors must not alias");
  SolverState* st = op->state.as<SolverState>();
  double* parts = op->partials.as<double>();
  const unsigned cgrid = grid_for(C);
  const Space lcp{MHIP_SPACE_LOWER_BOUND, 0.0, 0.0};
  if (int e = op->iterate.reserve(2 * (C + 1) * sizeof(double2))) return e;
  double* D0 = op->iterate.as<double>();  // dt * sep_dot of the two iterates (see k_scrap_constraint)
  double* D1 = D0 + C;
  auto constraint = [&](bool init) {
#define SCON(R, I) k_scrap_constraint<R, I><<<cgrid, kBlock, 0, s>>>(op->view, st, lam_tmp, lam, g_tmp, g, D0, D1, sep, parts)
    if (op->kin == KIN_ROD) { if (init) SCON(KIN_ROD, true); else SCON(KIN_ROD, false); }
    else if (op->kin == KIN_RIGID) { if (init) SCON(KIN_RIGID, true); else SCON(KIN_RIGID, false); }
    else { if (init) SCON(KIN_TRANS, true); else SCON(KIN_TRANS, false); }
#undef SCON
  };
  // gkm1 = D^T M D xkm1 with xkm1 = the given multipliers (:576-611)
  if (int e = launch_copy(C, lam_tmp, lam, s)) return e;
  MHIP_HIP(hipMemsetAsync(st, 0, sizeof(SolverState), s));
  if (int e = op_launch_body(op, X_INIT, lam_tmp, lam, g_tmp, g, lcp, s)) return e;
  constraint(true);
  MHIP_LAUNCH_CHECK();
  k_scrap_finalize<true><<<1, kBlock, 0, s>>>((int)cgrid, parts, st, max_allowable_overlap, max_iterations);
  MHIP_LAUNCH_CHECK();
  unsigned enqueued = 0, chunk = 8;
  for (;;) {
    MHIP_HIP(hipMemcpyAsync(op->host_state, st, sizeof(SolverState), hipMemcpyDeviceToHost, s));
    MHIP_HIP(hipStreamSynchronize(s));
    if (op->host_state->done || enqueued >= max_iterations) break;
    const unsigned todo = (max_iterations - enqueued < chunk) ? max_iterations - enqueued : chunk;
    for (unsigned k = 0; k < todo; ++k) {
      // first iteration: the projected step sees sep only (signed_sep_dot == 0 at :639) -> q stands in for g_tmp
      const double* gbody = (enqueued + k == 0) ? sep : g_tmp;
      if (int e = op_launch_body(op, X_SOLVE, lam_tmp, lam, gbody, g, lcp, s)) return e;
      constraint(false);
      MHIP_LAUNCH_CHECK();
      k_scrap_finalize<false><<<1, kBlock, 0, s>>>((int)cgrid, parts, st, max_allowable_overlap, max_iterations);
      MHIP_LAUNCH_CHECK();
    }
    enqueued += todo;
    if (chunk < 64) chunk *= 2;
  }
  k_finish<<<grid_for(C), kBlock, 0, s>>>(C, st, lam_tmp, lam, g_tmp, g);
  MHIP_LAUNCH_CHECK();
  if (max_speed && op->view.N > 0) {
    ReduceScratch& rs = reduce_scratch();
    if (int e = rs.ensure()) return e;
    double* mp = rs.dev.as<double>();
    const unsigned g2 = grid_for(op->view.N);
    k_max_speed<<<g2, kBlock, 0, s>>>(op->view.N, op->view.vel, mp);
    MHIP_LAUNCH_CHECK();
    k_reduce_final<2><<<1, kBlock, 0, s>>>((int)g2, mp, mp + 2 * kMaxGrid);
    MHIP_LAUNCH_CHECK();
    MHIP_HIP(hipMemcpyAsync(rs.host, mp + 2 * kMaxGrid, sizeof(double), hipMemcpyDeviceToHost, s));
    MHIP_HIP(hipStreamSynchronize(s));
    *max_speed = rs.host[0] < 0.0 ? 0.0 : rs.host[0];
  }
  MHIP_HIP(hipStreamSynchronize(s));
  result->num_iters = op->host_state->iter;
  result->residual = op->host_state->residual;
  result->converged = op->host_state->converged;
  return MHIP_SUCCESS;
}

int mhip_contact_op_set_partition(mhip_contact_op_t op, size_t body_first, size_t body_count,
                                  const unsigned char* counted_contacts, double* velocity) {
  MHIP_REQUIRE(op != nullptr, MHIP_ERR_INVALID_ARGUMENT, "operator handle is null");
  MHIP_REQUIRE(body_first + body_count <= op->view.N, MHIP_ERR_INVALID_ARGUMENT,
               "owned body range [%zu, %zu) exceeds the %zu local bodies", body_first, body_first + body_count,
               op->view.N);
  op->view.aptr = nullptr;
  op->view.body_first = body_first;
  op->view.body_count = body_count;
  op->view.counted = counted_contacts;
  op->view.vel = velocity ? velocity : op->vel.as<double>();
  return MHIP_SUCCESS;
}

int mhip_bbpgd_stage_begin(mhip_contact_op_t op, const double* q, const mhip_space* space,
                           const mhip_pgd_config* config, double* x, double* g, double* x_tmp, double* g_tmp,
                           mhip_stream_t stream) {
  MHIP_REQUIRE(op != nullptr, MHIP_ERR_INVALID_ARGUMENT, "operator handle is null");
  if (int e = check_config(config)) return e;
  Space sp;
  if (int e = to_space(space, &sp)) return e;
  const size_t C = op->view.C;
  MHIP_REQUIRE(C == 0 || (q && x && g && x_tmp && g_tmp), MHIP_ERR_INVALID_ARGUMENT,
               "solver vectors must not be null");
  MHIP_REQUIRE(C == 0 || (x != x_tmp && g != g_tmp && x != g), MHIP_ERR_INVALID_ARGUMENT,
               "solver vectors must not alias");
  op->stage.q = q; op->stage.x = x; op->stage.g = g; op->stage.x_tmp = x_tmp; op->stage.g_tmp = g_tmp;
  op->stage.sp = sp;
  op->stage.cfg = *config;
  op->stage.active = true;
  op->stage.part_used = 0;
  op->view.aptr = nullptr;
  if (op->view.body_mask)
    MHIP_HIP(hipMemsetAsync(op->view.body_mask, 0x00, op->view.N * sizeof(unsigned long long), as_stream(stream)));
  if (int e = op->iterate.reserve(2 * (C + 1) * sizeof(double2))) return e;
  MHIP_HIP(hipMemsetAsync(op->state.ptr, 0, sizeof(SolverState), as_stream(stream)));
  // cold tier (see "Cold tier"): interior contacts only, one row buffer
  op->stage.P0 = op->iterate.as<double>();
  op->stage.P1 = op->stage.P0 + 2 * C;
  op->stage.q_cur = q;
  op->stage.interior = 0;
  op->stage.interior_known = false;
  op->stage.polls = 0;
  op->stage.iter_at_poll = 0;
  op->stage.snap_at = 0;
  op->stage.pause_on_bad_step = op->tiering != 0 && tier_problem_kind(sp, config->residual_kind);
  mhip_contact_op::Tier& tier = op->tier;
  if (tier.active) {  // a staged solve that ended in an error left the operator in the tier numbering
    const TierGeo geo = tier_geo_at(tier.geo[tier.set].ptr, C);
    k_tier_remap_inc<<<grid_for(2 * C), kBlock, 0, as_stream(stream)>>>(2 * C, op->inc.as<int32_t>(), geo.orig);
    MHIP_LAUNCH_CHECK();
    op->view.pairs = tier.saved.pairs;   // (the caller's arrays; everything else of the view is the caller's to set)
    op->view.normal = tier.saved.normal;
    op->view.ra = tier.saved.ra;
    op->view.rb = tier.saved.rb;
    op->view.arc_s = tier.saved.arc_s;
    op->view.arc_t = tier.saved.arc_t;
    op->view.pos = tier.saved.pos;
    op->view.aptr = nullptr;
  }
  op->view.vel_alt = nullptr;
  op->view.drift = nullptr;
  op->view.fire_at = nullptr;
  op->view.fired = nullptr;
  op->view.tier_counters = nullptr;
  tier.active = tier.tracking = false;
  tier.disabled = !op->stage.pause_on_bad_step || op->view.body_mask == nullptr ||
                  !(C >= kTierMinContacts || op->tiering >= 2);
  tier.tiered_iterations = tier.retiers = tier.wakeups = 0;
  tier.hot_sum = 0.0;
  return MHIP_SUCCESS;
}

int mhip_bbpgd_stage_body(mhip_contact_op_t op, int init, mhip_stream_t stream) {
  MHIP_REQUIRE(op != nullptr && op->stage.active, MHIP_ERR_RUNTIME, "mhip_bbpgd_stage_begin has not been called");
  auto& st = op->stage;
  if (init) return op_launch_body(op, X_INIT, st.x, st.x, nullptr, nullptr, st.sp, as_stream(stream));
  return op_launch_body(op, X_SOLVE, st.P0, st.P1, nullptr, nullptr, st.sp, as_stream(stream), true);
}

int mhip_bbpgd_stage_constraint_range(mhip_contact_op_t op, int init, size_t c_first, size_t c_count,
                                      mhip_stream_t stream) {
  MHIP_REQUIRE(op != nullptr && op->stage.active, MHIP_ERR_RUNTIME, "mhip_bbpgd_stage_begin has not been called");
  MHIP_REQUIRE(c_first + c_count <= op->view.C, MHIP_ERR_INVALID_ARGUMENT,
               "constraint range [%zu, %zu) exceeds the %zu constraints", c_first, c_first + c_count, op->view.C);
  auto& st = op->stage;
  // the first range of an iteration that starts at 0 is the interior one (both bodies of its contacts are owned)
  if (c_first == 0 && !st.interior_known) {
    st.interior = c_count;
    st.interior_known = true;
  }
  mhip_contact_op::Tier& tier = op->tier;
  if (tier.active) {
    // tiered: the ranges the tiers were built for are [0, interior) -- of which only the hot part is swept, with the
    // workgroups that wake the contacts of fired bodies in front -- and [interior, C)
    MHIP_REQUIRE(!init && ((c_first == 0 && c_count == tier.I) || (c_first == tier.I && c_first + c_count == op->view.C)),
                 MHIP_ERR_RUNTIME, "constraint range [%zu, %zu) does not match the ranges of the tiered solve", c_first,
                 c_first + c_count);
    hipStream_t s = as_stream(stream);
    const SolverState* sst = op->state.as<SolverState>();
    double* parts = op->partials.as<double>();
    const TierMisc m = tier_misc_at(tier.misc.ptr, op->view.C, op->view.N);
    OpView vw = op->view;
    vw.part_offset = st.part_used;
    vw.part_stride = kStageStride;
    unsigned grid = 0, extra = 0;
    TierCheck tc{};
    if (c_first == 0) {
      vw.c_first = 0;
      vw.c_end = tier.H;
      grid = tier.H ? constraint_grid(tier.H) : 0u;
      if (tier.H < tier.I) {
        extra = tier.service_blocks;
        tc = TierCheck{tier.H, tier.I, m.wake[tier.set], m.list, m.counters, m.fired, m.fire_at, extra};
      }
    } else {
      if (c_count == 0) return MHIP_SUCCESS;
      vw.c_first = c_first;
      vw.c_end = c_first + c_count;
      grid = constraint_grid(c_count);
    }
    if (grid + extra == 0) return MHIP_SUCCESS;
    MHIP_REQUIRE(st.part_used + grid + extra <= kStageStride, MHIP_ERR_RUNTIME, "too many constraint sweeps in one iteration");
#define STAGED(K)                                                                                                    \
  k_constraint<X_SOLVE, K, true><<<grid + extra, kBlock, 0, s>>>(vw, sst, st.P0, st.P1, nullptr, nullptr, st.q_cur, \
                                                                st.sp, st.cfg.residual_kind, parts, tc)
    if (op->kin == KIN_ROD) STAGED(KIN_ROD); else if (op->kin == KIN_RIGID) STAGED(KIN_RIGID); else STAGED(KIN_TRANS);
#undef STAGED
    MHIP_LAUNCH_CHECK();
    st.part_used += grid + extra;  // (the service workgroups' records follow the sweeping ones')
    return MHIP_SUCCESS;
  }
  if (c_count == 0) return MHIP_SUCCESS;
  const unsigned grid = constraint_grid(c_count);
  MHIP_REQUIRE(st.part_used + grid <= kStageStride, MHIP_ERR_RUNTIME, "too many constraint sweeps in one iteration");
  op->view.c_first = c_first;
  op->view.c_end = c_first + c_count;
  op->view.part_offset = st.part_used;
  op->view.part_stride = kStageStride;
  const int e = op_launch_constraint(op, init ? X_INIT : X_SOLVE, st.P0, st.P1, init ? st.x : nullptr, nullptr, st.q_cur,
                                     st.sp, st.cfg.residual_kind, grid, as_stream(stream), true);
  op->view.c_first = 0;
  op->view.c_end = op->view.C;
  op->view.part_offset = 0;
  op->view.part_stride = 0;
  if (e) return e;
  st.part_used += grid;
  return MHIP_SUCCESS;
}

int mhip_bbpgd_stage_reduce(mhip_contact_op_t op, int init, double* local, mhip_stream_t stream) {
  return mhip::stage_reduce_exchange(op, init, local, mhip::MailboxArgs{}, as_stream(stream));
}

int mhip_bbpgd_stage_constraint(mhip_contact_op_t op, int init, double* local, mhip_stream_t stream) {
  MHIP_REQUIRE(op != nullptr && op->stage.active, MHIP_ERR_RUNTIME, "mhip_bbpgd_stage_begin has not been called");
  op->stage.part_used = 0;
  if (int e = mhip_bbpgd_stage_constraint_range(op, init, 0, op->view.C, stream)) return e;
  return mhip_bbpgd_stage_reduce(op, init, local, stream);
}

int mhip_bbpgd_stage_finalize(mhip_contact_op_t op, int init, const double* gathered, int nparts,
                              mhip_stream_t stream) {
  MHIP_REQUIRE(op != nullptr && op->stage.active, MHIP_ERR_RUNTIME, "mhip_bbpgd_stage_begin has not been called");
  MHIP_REQUIRE(gathered != nullptr && nparts >= 1, MHIP_ERR_INVALID_ARGUMENT, "gathered records missing");
  const auto& cfg = op->stage.cfg;
  SolverState* st = op->state.as<SolverState>();
  if (init)
    k_finalize<X_INIT><<<1, final_block(nparts), 0, as_stream(stream)>>>(nparts, gathered, kRed, 1, st, cfg.residual_kind,
                                                                         cfg.tol, cfg.max_iters);
  else
    k_finalize<X_SOLVE><<<1, final_block(nparts), 0, as_stream(stream)>>>(
        nparts, gathered, kRed, 1, st, cfg.residual_kind, cfg.tol, cfg.max_iters,
        // (test hook, mode 2: every rank pauses at the first iteration after its tenth snapshot poll -- a condition that
        // is the same on all ranks, as a bad step would be)
        op->stage.pause_on_bad_step ? ((op->tiering == 2 && op->stage.polls >= 10) ? 2 : 1) : 0,
        op->tier.active ? op->view.tier_counters : nullptr);
  MHIP_LAUNCH_CHECK();
  return MHIP_SUCCESS;
}

int mhip_bbpgd_stage_poll(mhip_contact_op_t op, mhip_solve_result* result, int* done, mhip_stream_t stream) {
  MHIP_REQUIRE(op != nullptr && result != nullptr && done != nullptr, MHIP_ERR_INVALID_ARGUMENT, "null argument");
  hipStream_t s = as_stream(stream);
  MHIP_HIP(hipMemcpyAsync(op->host_state, op->state.ptr, sizeof(SolverState), hipMemcpyDeviceToHost, s));
  MHIP_HIP(hipStreamSynchronize(s));
  if (op->stage.active) {
    if (op->tier.active && op->host_state->iter > op->stage.iter_at_poll) {  // (statistics: iterations that ran tiered)
      const unsigned ran = op->host_state->iter - op->stage.iter_at_poll;
      op->tier.tiered_iterations += ran;
      op->tier.hot_sum += ran * (static_cast<double>(op->tier.H + (op->view.C - op->tier.I)) / static_cast<double>(op->view.C));
    }
    op->stage.iter_at_poll = op->host_state->iter;
  }
  if (op->host_state->done == 2 && op->stage.active) {
    // paused before a BB step outside [0, finite] (every rank pauses at the same iteration: the step is global): the
    // sleepers' exact gradients are needed, so the tiers are left for good and the solve goes on
    mhip_contact_op::Tier& tier = op->tier;
    if (tier.active) {
      TierPairs cur{op->stage.P0, op->stage.P1, op->stage.q_cur};
      double* P0 = op->iterate.as<double>();
      if (int e = tier_release(op, cur, false, P0, P0 + 2 * op->view.C, op->stage.q, nullptr, nullptr, nullptr, nullptr, s))
        return e;
      op->stage.P0 = cur.P0;
      op->stage.P1 = cur.P1;
      op->stage.q_cur = cur.q;
    } else if (int e = tier_stop_tracking(op, s)) {
      return e;
    }
    tier.disabled = true;
    op->stage.pause_on_bad_step = false;
    op->host_state->done = 0;
    MHIP_HIP(hipMemcpyAsync(&op->state.as<SolverState>()->done, &op->host_state->done, sizeof(int), hipMemcpyHostToDevice, s));
    MHIP_HIP(hipStreamSynchronize(s));
  }
  result->num_iters = op->host_state->iter;
  result->residual = op->host_state->residual;
  result->converged = op->host_state->converged;
  *done = op->host_state->done;
  return MHIP_SUCCESS;
}

int mhip_bbpgd_stage_snapshot_active(mhip_contact_op_t op, mhip_stream_t stream) {
  MHIP_REQUIRE(op != nullptr && op->stage.active, MHIP_ERR_RUNTIME, "mhip_bbpgd_stage_begin has not been called");
  hipStream_t s = as_stream(stream);
  auto& st = op->stage;
  st.polls += 1;
  mhip_contact_op::Tier& tier = op->tier;
  const size_t retiers_before = tier.retiers;
  // cold tier: needs the interior range (learnt from the sweeps of the iterations run so far) and host_state (the
  // caller has just polled).  Only contacts between two owned bodies may sleep: a ghost's drift is not known here.
  if (!tier.disabled && st.interior_known && !op->host_state->done) {
    TierPairs cur{st.P0, st.P1, st.q_cur};
    const unsigned period = op->host_state->iter > tier.polled_at ? op->host_state->iter - tier.polled_at : 1u;
    if (int e = tier_update(op, cur, op->host_state->iter, period, st.interior, /*pingpong=*/false, s)) return e;
    st.P0 = cur.P0;
    st.P1 = cur.P1;
    st.q_cur = cur.q;
  }
  // the snapshot itself (a pass over the incidence lists, 0.2 ms at 10^6 rods) pays once per few dozen iterations: a
  // caller that polls more often gets it at every other or every fourth poll
  const unsigned it = op->host_state->iter;
  const bool renumbered = tier.retiers != retiers_before;  // (the lists follow the numbering)
  if (!renumbered && op->view.aptr != nullptr && st.snap_at != 0 && it - st.snap_at < (it / 2 < 32u ? it / 2 : 32u))
    return MHIP_SUCCESS;
  st.snap_at = it;
  return op_snapshot_active(op, s);
}

int mhip_bbpgd_stage_end(mhip_contact_op_t op, mhip_solve_result* result, mhip_stream_t stream) {
  MHIP_REQUIRE(op != nullptr && op->stage.active, MHIP_ERR_RUNTIME, "mhip_bbpgd_stage_begin has not been called");
  auto& st = op->stage;
  hipStream_t s = as_stream(stream);
  int done = 0;
  mhip_solve_result r{};
  if (int e = mhip_bbpgd_stage_poll(op, result ? result : &r, &done, stream)) return e;  // host_state for the tiers
  if (op->tier.active) {
    TierPairs cur{st.P0, st.P1, st.q_cur};
    double* P0 = op->iterate.as<double>();
    if (int e = tier_release(op, cur, true, P0, P0 + 2 * op->view.C, st.q, st.x, st.g, st.x_tmp, st.g_tmp, s)) return e;
  } else {
    if (int e = tier_stop_tracking(op, s)) return e;
    if (op->view.C > 0) {
      const double2* P0 = op->iterate.as<double2>();
      k_finish_packed<<<grid_for(op->view.C), kBlock, 0, s>>>(op->view.C, op->state.as<SolverState>(), P0,
                                                              P0 + op->view.C, st.x, st.g, st.x_tmp, st.g_tmp);
      MHIP_LAUNCH_CHECK();
    }
  }
  // rods: the angular velocities of the final iterate (the iterations' sweeps keep only the (U, Z) rows)
  if (op->kin == KIN_ROD && op->view.C > 0)
    if (int e = op_launch_body(op, X_APPLY, st.x, nullptr, nullptr, nullptr, st.sp, s)) return e;
  MHIP_HIP(hipStreamSynchronize(s));
  st.active = false;
  return MHIP_SUCCESS;
}

int mhip_bbpgd_solve_contact_unfused(mhip_contact_op_t op, const double* q, const mhip_space* space,
                                     const mhip_pgd_config* config, double* x, double* g, double* x_tmp,
                                     double* g_tmp, mhip_solve_result* result, mhip_stream_t stream) {
  MHIP_REQUIRE(op != nullptr && result != nullptr, MHIP_ERR_INVALID_ARGUMENT, "null handle / result");
  if (int e = check_config(config)) return e;
  Space sp;
  if (int e = to_space(space, &sp)) return e;
  const size_t n = op->view.C;
  REQ(q); REQ(x); REQ(g); REQ(x_tmp); REQ(g_tmp);
  auto apply = [&](const double* in, double* out) { return mhip_contact_op_apply(op, in, out, stream); };
  return solve_generic(n, apply, q, sp, config, x, g, x_tmp, g_tmp, result, as_stream(stream));
}

int mhip_bbpgd_solve_dense(size_t n, const double* A, const double* q, const mhip_space* space,
                           const mhip_pgd_config* config, double* x, double* g, double* x_tmp, double* g_tmp,
                           mhip_solve_result* result, mhip_stream_t stream) {
  MHIP_REQUIRE(result != nullptr, MHIP_ERR_INVALID_ARGUMENT, "result is null");
  if (int e = check_config(config)) return e;
  Space sp;
  if (int e = to_space(space, &sp)) return e;
  REQ(A); REQ(q); REQ(x); REQ(g); REQ(x_tmp); REQ(g_tmp);
  auto apply = [&](const double* in, double* out) { return mhip_gemv(n, A, in, out, stream); };
  return solve_generic(n, apply, q, sp, config, x, g, x_tmp, g_tmp, result, as_stream(stream));
}

}  // extern "C"
