// mhip_internal.hpp -- shared host/device helpers of libmundy_hip.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>

#include "../../include/mundy_hip.h"

// The hand-offs between workgroups (k_fold_finalize, lockstep::post_start, the mailbox / inbox protocols) publish with
// relaxed agent-scope atomic stores drained by s_waitcnt vmcnt(0) and read with relaxed agent-scope loads -- no release /
// acquire pair.  That is sound on the CDNA3/4 memory system these kernels are written for (sc1 stores write through the
// XCD's L2, stores are counted in vmcnt); on any other target it would silently read stale records.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "libmundy_hip's inter-workgroup hand-offs assume the gfx950 (gfx942) memory model: build with --offload-arch=gfx950"
#endif

namespace mhip {

// ------------------------------------------------------------------------------------------------------------------
// error plumbing: status code + thread-local message (mundy_core/throw_assert.hpp conventions mapped onto a C ABI)
// ------------------------------------------------------------------------------------------------------------------
std::string& last_error_storage();
int fail(int code, const char* fmt, ...);

#define MHIP_REQUIRE(cond, code, ...)                      \
  do {                                                     \
    if (!(cond)) return ::mhip::fail((code), __VA_ARGS__); \
  } while (0)

#define MHIP_HIP(call)                                                                                  \
  do {                                                                                                  \
    hipError_t e_ = (call);                                                                             \
    if (e_ != hipSuccess)                                                                               \
      return ::mhip::fail(MHIP_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, \
                          __LINE__);                                                                    \
  } while (0)

#define MHIP_LAUNCH_CHECK() MHIP_HIP(hipGetLastError())

inline hipStream_t as_stream(mhip_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

constexpr int kBlock = 256;     // 4 waves per workgroup
constexpr int kMaxGrid = 2048;  // grid-stride cap: 256 CUs x 8 workgroups (guide, Guideline 11)

inline unsigned grid_for(size_t n, int block = kBlock) {
  const size_t g = (n + block - 1) / block;
  return static_cast<unsigned>(g == 0 ? 1 : (g > static_cast<size_t>(kMaxGrid) ? kMaxGrid : g));
}
inline unsigned grid_exact(size_t n, int block = kBlock) {
  const size_t g = (n + block - 1) / block;
  return static_cast<unsigned>(g == 0 ? 1 : g);
}

// A device buffer that only ever grows (workspace owned by a handle; never allocated inside a timed loop once warm).
struct DeviceBuffer {
  void* ptr = nullptr;
  size_t bytes = 0;
  int reserve(size_t need) {
    if (need <= bytes) return MHIP_SUCCESS;
    if (ptr) MHIP_HIP(hipFree(ptr));
    ptr = nullptr;
    bytes = 0;
    const size_t want = need + need / 4 + 256;
    MHIP_HIP(hipMalloc(&ptr, want));
    bytes = want;
    return MHIP_SUCCESS;
  }
  void release() {
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
    bytes = 0;
  }
  template <class T>
  T* as() const {
    return static_cast<T*>(ptr);
  }
};

// ------------------------------------------------------------------------------------------------------------------
// device math: the mundy::math subset the path needs, same operation order as the reference
// ------------------------------------------------------------------------------------------------------------------
struct V3 {
  double x, y, z;
};
struct Quat {
  double w, x, y, z;
};

__host__ __device__ inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__host__ __device__ inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__host__ __device__ inline V3 operator*(double s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
__host__ __device__ inline V3 operator*(V3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
// right fold: a0*b0 + (a1*b1 + a2*b2)   (mundy_math/impl/VectorImpl.hpp:339-344)
__host__ __device__ inline double dot(V3 a, V3 b) { return a.x * b.x + (a.y * b.y + a.z * b.z); }
__host__ __device__ inline double norm(V3 a) { return sqrt(dot(a, a)); }
__host__ __device__ inline V3 cross(V3 a, V3 b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__host__ __device__ inline double comp(V3 a, int k) { return k == 0 ? a.x : (k == 1 ? a.y : a.z); }

__host__ __device__ inline Quat qmul(Quat q, Quat o) {
  Quat r;
  r.w = q.w * o.w - q.x * o.x - q.y * o.y - q.z * o.z;
  r.x = q.w * o.x + q.x * o.w + q.y * o.z - q.z * o.y;
  r.y = q.w * o.y - q.x * o.z + q.y * o.w + q.z * o.x;
  r.z = q.w * o.z + q.x * o.y - q.y * o.x + q.z * o.w;
  return r;
}
// q * v = (q (0,v)) inverse(q), inverse = conjugate * (1/|q|^2)  (impl/QuaternionImpl.hpp:184-202)
__host__ __device__ inline V3 qrot(Quat q, V3 v) {
  const double inv_n2 = 1.0 / (q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
  const Quat qi{q.w * inv_n2, -q.x * inv_n2, -q.y * inv_n2, -q.z * inv_n2};
  const Quat vq{0.0, v.x, v.y, v.z};
  const Quat r = qmul(qmul(q, vq), qi);
  return {r.x, r.y, r.z};
}

constexpr double kZeroTol = 1e-15;  // get_zero_tolerance<double>() (mundy_math/Tolerance.hpp:38-50)

// Arclength of a rod's CONTACT POINT from the parameter the distance routines hand back.  distance(Point,
// LineSegment) clamps the closest point to an endpoint but leaves its parameter t unclamped
// (mundy_geom/distance/PointLineSegment.hpp:156-166), and the colinear branch of segment-segment returns that t
// (LineSegmentLineSegment.hpp:236-265); the contact point of the assembly is the clamped closest point
// (scrap/.../SpherocylinderSpherocylinderLinker.cpp:246-247), whose arclength is t clamped to [0, 1].
__host__ __device__ inline double contact_arclength(double t) { return t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t); }
// coefficient of the rod-compressed lever arm, (arclength - 1/2) (p1 - p0): |coef| <= 1/2 whatever the caller passed
__host__ __device__ inline double rod_arm_coef(double t) { return contact_arclength(t) - 0.5; }


__device__ inline V3 load3(const double* p, size_t i) { return {p[3 * i], p[3 * i + 1], p[3 * i + 2]}; }
__device__ inline void store3(double* p, size_t i, V3 v) {
  p[3 * i] = v.x;
  p[3 * i + 1] = v.y;
  p[3 * i + 2] = v.z;
}
__device__ inline Quat load4q(const double* p, size_t i) {
  const double2 a = *reinterpret_cast<const double2*>(p + 4 * i);
  const double2 b = *reinterpret_cast<const double2*>(p + 4 * i + 2);
  return {a.x, a.y, b.x, b.y};
}

// ------------------------------------------------------------------------------------------------------------------
// wave / block reductions (64-lane wavefronts)
// ------------------------------------------------------------------------------------------------------------------
__device__ inline double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}
__device__ inline double wave_max(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, 64));
  return v;
}
__device__ inline int wave_or(int v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v |= __shfl_down(v, off, 64);
  return v;
}
// Block reductions for kBlock threads; result valid in thread 0.  `scratch` holds kBlock/64 doubles.
__device__ inline double block_sum(double v, double* scratch) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) scratch[w] = v;
  __syncthreads();
  double r = 0.0;
  if (threadIdx.x == 0) {
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) r += scratch[i];
  }
  __syncthreads();
  return r;
}
__device__ inline double block_max(double v, double* scratch) {
  v = wave_max(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) scratch[w] = v;
  __syncthreads();
  double r = -1.7976931348623157e308;
  if (threadIdx.x == 0) {
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) r = fmax(r, scratch[i]);
  }
  __syncthreads();
  return r;
}

// ------------------------------------------------------------------------------------------------------------------
// Double-double accumulation (cascaded summation with error-free TwoSum: Ogita, Rump, Oishi 2005).  Every sum whose
// rounding feeds the Barzilai-Borwein step -- the two BB dot products and the per-body force / torque sums -- is kept as
// an unevaluated pair hi + lo and rounded to one double once, at the end.  The pair carries ~106 bits, so the rounded
// result is the correctly rounded exact sum unless that sum lies within ~n 2^-106 (relative) of a rounding boundary:
// the summation order (lanes per body, tile -> workgroup mapping, rank count) no longer reaches the iterates, and a
// serial CPU evaluation with the same pair arithmetic lands on the same bits.
// ------------------------------------------------------------------------------------------------------------------
struct DD {
  double hi, lo;
};
__host__ __device__ inline void dd_add(DD& a, double b) {
  const double s = a.hi + b;
  const double bb = s - a.hi;
  const double e = (a.hi - (s - bb)) + (b - bb);  // a.hi + b == s + e exactly
  a.hi = s;
  a.lo += e;
}
__host__ __device__ inline void dd_add(DD& a, const DD& b) {
  dd_add(a, b.hi);
  a.lo += b.lo;
}
// one rounding of hi + lo; an overflowed / NaN sum is reported as the plain sum would (lo is NaN there)
__host__ __device__ inline double dd_value(const DD& a) { return (a.hi - a.hi == 0.0) ? a.hi + a.lo : a.hi; }
struct DD3 {
  DD x, y, z;
};
__host__ __device__ inline void dd_add(DD3& a, V3 b) {
  dd_add(a.x, b.x);
  dd_add(a.y, b.y);
  dd_add(a.z, b.z);
}
__host__ __device__ inline V3 dd_value(const DD3& a) { return {dd_value(a.x), dd_value(a.y), dd_value(a.z)}; }
__device__ inline DD dd_shfl_xor(DD v, int off) { return {__shfl_xor(v.hi, off, 64), __shfl_xor(v.lo, off, 64)}; }
__device__ inline DD dd_shfl_down(DD v, int off) { return {__shfl_down(v.hi, off, 64), __shfl_down(v.lo, off, 64)}; }
__device__ inline DD wave_sum(DD v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) dd_add(v, dd_shfl_down(v, off));
  return v;
}
// `scratch` holds 2 * kBlock/64 doubles; result valid in thread 0
__device__ inline DD block_sum(DD v, double* scratch) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) {
    scratch[2 * w] = v.hi;
    scratch[2 * w + 1] = v.lo;
  }
  __syncthreads();
  DD r{0.0, 0.0};
  if (threadIdx.x == 0) {
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) dd_add(r, DD{scratch[2 * i], scratch[2 * i + 1]});
  }
  __syncthreads();
  return r;
}

// projection onto the separable 1-D convex spaces of convex.hpp:46-115
struct Space {
  int kind;
  double lo, hi;
  __host__ __device__ inline double project(double x) const {
    // max(a,b) = a<b ? b : a ; min(a,b) = b<a ? b : a  (explicit, so -0.0 / ties behave as on the host)
    switch (kind) {
      case MHIP_SPACE_LOWER_BOUND: return (x < lo) ? lo : x;
      case MHIP_SPACE_UPPER_BOUND: return (hi < x) ? hi : x;
      case MHIP_SPACE_BOUNDED: {
        const double m = (x < lo) ? lo : x;
        return (hi < m) ? hi : m;
      }
      default: return x;
    }
  }
};
inline int to_space(const mhip_space* s, Space* out) {
  MHIP_REQUIRE(s != nullptr, MHIP_ERR_INVALID_ARGUMENT, "space must not be null");
  MHIP_REQUIRE(s->kind >= MHIP_SPACE_UNCONSTRAINED && s->kind <= MHIP_SPACE_BOUNDED, MHIP_ERR_INVALID_ARGUMENT,
               "unknown convex space kind %d", s->kind);
  *out = Space{s->kind, s->lower_bound, s->upper_bound};
  return MHIP_SUCCESS;
}

// exclusive scan of n int32 counts into out[0..n] (out[n] = total); workspace >= scan_workspace_bytes(n)
size_t scan_workspace_bytes(size_t n);
// Host-side trace ranges around the stages of the path (the reference labels its kernels for Kokkos Tools, e.g.
// "PrefixSum" / "ScatterValid" GenNeighborLinkers.hpp:148,164, "axpby" / "diff_dot" / "reduce_max" convex.hpp:208-283):
// roctx ranges with the reference's names, visible to rocprofv3 --marker-trace.  Off unless MHIP_TRACE=1 or
// mhip_set_tracing(1); libroctx64 is looked up at run time, so there is no link dependency.
struct TraceRange {
  explicit TraceRange(const char* name);
  ~TraceRange();
  TraceRange(const TraceRange&) = delete;
  TraceRange& operator=(const TraceRange&) = delete;
  bool pushed;
};

int exclusive_scan_i32(const int32_t* in, int32_t* out, size_t n, void* workspace, hipStream_t stream);

// ---- mailbox (dist.hip: mhip_comm_mailbox_open): the ranks of one node exchange a small record per solver iteration
// through slots in each other's device memory (IPC-mapped, fine-grained).  PUSH: a rank writes its record into its
// slot of EVERY rank's box (posted writes over xGMI), then polls its OWN box (local reads).  A record of `width`
// doubles travels as 2 * width 8-byte words, each (32 bits of data, 32-bit flag = the exchange number): every word is
// one atomic store and validates itself, so no ordering between stores is needed (writes to a peer may arrive in any
// order) -- the low-latency protocol of the collective libraries.  Box layout: [parity of the exchange][rank][kSlotWords].
constexpr int kSlotWords = 16;  // words per slot: records of up to 8 doubles
struct MailboxArgs {
  unsigned long long* const* peers = nullptr;  // [world] device pointers: every rank's box (null: no mailbox)
  int world = 1, rank = 0, width = 0;
  double* gathered = nullptr;            // [world][width] out
  // [0] != 0: an exchange timed out (sticky); [1] = exchanges REALLY made so far (device-resident: see below)
  unsigned long long* status = nullptr;
  unsigned long long timeout = 0;        // ticks of the 100 MHz wall clock
};
// convex.hip: mhip_bbpgd_stage_reduce with the exchange inside the launch that forms the record
int stage_reduce_exchange(mhip_contact_op_t op, int init, double* local, const MailboxArgs& mb, hipStream_t s);
// convex.hip: the same and mhip_bbpgd_stage_finalize in ONE launch (fold, record, exchange, finalize)
int stage_reduce_exchange_finalize(mhip_contact_op_t op, int init, const MailboxArgs& mb, hipStream_t s);
constexpr int kMailboxMaxWorld = 64;  // ranks whose records one wave collects and adds up
// convex.hip: where the staged solve keeps its `flips` (completed non-terminal iterations) and `done` words on the
// device -- the per-iteration halo (dist.hip) numbers its exchanges with the first and skips them on the second
void stage_state_words(mhip_contact_op_t op, const unsigned** flips, const int** done);
// Called by the first wave of a workgroup (threads 0 .. 63); `mine` = this rank's record, readable by every lane of the
// wave (shared or global memory).  Two slot sets alternate: a rank can post exchange k + 2 only after it has read
// everybody's k + 1, which the others posted after reading everybody's k -- nobody still reads the slots of k when they
// are written again.  That argument needs every exchange number to be exchanged, so the number lives ON THE DEVICE
// (status[1], advanced by the wave that exchanges): launches enqueued behind a converged solve return before the
// exchange (`done` is the same on every rank), and a host-side count of enqueued launches would let an odd number of
// skipped ones hand the next real exchange the slot set of the last one while a slow peer still reads it.
// All accesses are relaxed system-scope atomics on fine-grained memory (uncached: nothing to write
// back or invalidate -- a release / acquire pair at system scope would flush the whole L2 every iteration).  The wait
// is bounded: a rank that never posts ends in an error on the host (status[0]), not in waves that never finish.
// `into` (optional, shared memory): the collected records go there instead of m.gathered -- for a caller that goes on
// to use them in the same launch.
__device__ inline void mailbox_exchange_wave(const MailboxArgs& m, const double* mine, double* into = nullptr) {
  if (threadIdx.x >= 64) return;
  const int nw = 2 * m.width;
  // exchanges on one rank are launches on one stream, one wave each: nothing else touches status[1] meanwhile
  const unsigned long long seq = __hip_atomic_load(&m.status[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1ull;
  const unsigned flag = static_cast<unsigned>(seq);
  const size_t set = (seq & 1ull) * (size_t)m.world * kSlotWords;
  // post: word w of my record into my slot of rank r's box, for every (r, w)
  for (int t = threadIdx.x; t < m.world * nw; t += 64) {
    const int r = t / nw, w = t % nw;
    const unsigned long long bits = static_cast<unsigned long long>(__double_as_longlong(mine[w >> 1]));
    const unsigned half = static_cast<unsigned>((w & 1) ? (bits >> 32) : (bits & 0xffffffffull));
    unsigned long long* dst = m.peers[r] + set + (size_t)m.rank * kSlotWords + w;
    __hip_atomic_store(dst, (static_cast<unsigned long long>(flag) << 32) | half, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
  }
  // collect: word w of rank r's record from my own box
  unsigned long long* box = m.peers[m.rank] + set;
  for (int t = threadIdx.x; t < m.world * m.width; t += 64) {
    const int r = t / m.width, k = t % m.width;
    unsigned long long lo = 0, hi = 0;
    const unsigned long long t0 = wall_clock64();
    bool ok = true;
    for (;;) {
      lo = __hip_atomic_load(&box[(size_t)r * kSlotWords + 2 * k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      hi = __hip_atomic_load(&box[(size_t)r * kSlotWords + 2 * k + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if (static_cast<unsigned>(lo >> 32) == flag && static_cast<unsigned>(hi >> 32) == flag) break;
      if (__hip_atomic_load(&m.status[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull ||
          wall_clock64() - t0 > m.timeout) {
        ok = false;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    if (!ok) __hip_atomic_store(&m.status[0], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long bits = (hi << 32) | (lo & 0xffffffffull);
    (into ? into : m.gathered)[(size_t)r * m.width + k] =
        ok ? __longlong_as_double(static_cast<long long>(bits)) : __builtin_nan("");
  }
  // (every lane has read status[1] above: the wave runs in lockstep and the barrier keeps the store below the loops)
  __builtin_amdgcn_wave_barrier();
  if (threadIdx.x == 0) __hip_atomic_store(&m.status[1], seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// What a convergence poll of the fused / staged drivers does besides reading the solver state, and how many iterations
// are enqueued until the next one.
//   * A full poll reclassifies the tiers and rebuilds the compact active lists: ~0.35 ms at 10^6 rods (0.25 ms of
//     kernels, the rest bubbles around the host's reads), 14 of them in a 770-iteration solve.  Both are optimisations
//     -- the wake-up mechanism keeps a tiered sweep exact whatever the age of the classification, and an entry that
//     became active after the snapshot takes the per-body path -- so once two full polls in a row have found nothing to
//     do (no renumbering, the compact lists within 0.1 % of their length), the next kQuietPolls polls only read the
//     state.
//   * Near the tolerance the stretches are short: the launches enqueued behind the converging iteration return at
//     once, but three launches of 10^4 workgroups that return at once still cost time (54 of them behind iteration 770
//     of the 10^6-rod solve).  The residual of a BB iteration is far too bursty to extrapolate (at the polls of that
//     solve: 4.3, 5.3, 2.7e-2, 8.0e-2, 2.9e-3, 8.2e-3, 5.0e-2, 2.0e-3, 4.2e-4, 4.9e-4, 5.7e-5, converged ten
//     iterations later), so the rule is only: within a factor kNearTol of the tolerance, kShortStretch iterations at
//     a time, polled lightly.
struct PollPlan {
  static constexpr unsigned kQuietPolls = 3, kShortStretch = 16;
  static constexpr double kNearTol = 32.0;
  static constexpr size_t kShortStretchMinContacts = 2000000;
  unsigned light_left = 0, quiet = 0;
  int32_t snap_prev = -1;
  bool shortened = false;  // the stretch that has just run was a short one
  bool light_poll() {
    if (shortened) return true;  // (the early poll of a shortened stretch never does the full work)
    if (light_left == 0) return false;
    --light_left;
    return true;
  }
  void full_poll_done(unsigned enqueued, int32_t snapshot_entries, bool renumbered, bool tier_pending) {
    const bool still = enqueued >= 64 && !renumbered && !tier_pending && snap_prev > 0 &&
                       (snapshot_entries > snap_prev ? snapshot_entries - snap_prev : snap_prev - snapshot_entries) <=
                           snap_prev / 1024;
    snap_prev = snapshot_entries;
    quiet = still ? quiet + 1 : 0;
    if (quiet >= 2) light_left = kQuietPolls;
  }
  // (contacts: the size of the sweeps' grids -- behind a small system's converging iteration the idle launches cost less
  //  than the bubble of an extra poll: 125 000 rods, 17.0 ms per step with regular stretches, 17.3 with short ones)
  unsigned stretch(unsigned regular, unsigned iter, double residual, double tol, size_t contacts) {
    shortened = contacts >= kShortStretchMinContacts && iter >= 64 && regular > kShortStretch && residual > tol &&
                residual < kNearTol * tol;
    return shortened ? kShortStretch : regular;
  }
};

// sort.hip: stable LSD radix sort of (u64 key, u32 value) records over the key bits [0, 8 * passes), passes even (the
// result is back in keys / vals); segment sorts (ascending, unsigned) -- see sort.hip
constexpr int kShortSegment = 32;  // segments up to this length are sorted by one thread
size_t radix_sort_workspace_bytes(size_t n);
int radix_sort_u64(size_t n, unsigned long long* keys, unsigned* vals, unsigned long long* keys_tmp, unsigned* vals_tmp,
                   int passes, void* workspace, hipStream_t stream);
int sort_listed_segments_u32(const int32_t* seg_ptr, unsigned* data, unsigned* tmp, int key_bits,
                             const int32_t* long_count, const int32_t* long_list, hipStream_t stream);
int sort_segments_u32(size_t nseg, const int32_t* seg_ptr, unsigned* data, unsigned* tmp, int key_bits,
                      int32_t* scratch, hipStream_t stream);

}  // namespace mhip
