// halo.hip -- stream compaction helpers of the domain-decomposed path: pair filtering by ownership,
// ghost-candidate selection by box overlap, rank bounding boxes.  The compactions do the job of filter_view
// (mundy_mesh/GenNeighborLinkers.hpp:141-183: PrefixSum + ScatterValid) with wavefront ballots.  Integer work, HBM bound.
#include "geom_device.hpp"

namespace mhip {

struct HaloScratch {
  DeviceBuffer flags, pos, scanws, partials;
  size_t* host = nullptr;  // pinned
  int ensure(size_t n) {
    if (int e = flags.reserve((n + 2) * sizeof(int32_t))) return e;
    if (int e = pos.reserve((n + 2) * sizeof(int32_t))) return e;
    if (int e = scanws.reserve(scan_workspace_bytes(n + 2) + 64)) return e;
    if (int e = partials.reserve((6 * kMaxGrid + 8) * sizeof(double))) return e;
    if (!host) MHIP_HIP(hipHostMalloc(reinterpret_cast<void**>(&host), 64));
    return MHIP_SUCCESS;
  }
};
HaloScratch& halo_scratch() {
  thread_local HaloScratch s;
  return s;
}

// ---- order-preserving stream compaction with wavefront ballots -------------------------------------------------------
// A workgroup owns a tile of kCompactTile consecutive elements, visited in kCompactRounds rounds of 256 (element =
// tile base + round * 256 + thread), so every wavefront always looks at 64 CONSECUTIVE elements: the ballot of the
// predicate is that segment's keep-mask, its popcount the segment's size, and a kept element's slot inside the segment
// is the popcount of the mask below its lane.  Pass 1 stores one count per tile; a scan of the (n / 1024) tile counts
// gives tile bases; pass 2 re-evaluates the (cheap) predicate and writes the kept elements in order.  No flag or
// position arrays of length n (the filter_view structure, GenNeighborLinkers.hpp:141-183, needs both).
constexpr int kCompactRounds = 4;
constexpr int kCompactTile = kBlock * kCompactRounds;
constexpr int kCompactSegs = kCompactRounds * (kBlock / 64);  // 64-element segments per tile

struct KeepOwnedPair {  // a pair survives when at least one body is owned (ghost-ghost pairs are dropped)
  const int2* pairs;
  int first, last;
  int2* out;
  unsigned char* counted;
  __device__ bool keep(size_t k) const {
    const int2 ij = pairs[k];
    return (ij.x >= first && ij.x < last) || (ij.y >= first && ij.y < last);
  }
  __device__ void emit(size_t k, size_t slot) const {
    const int2 ij = pairs[k];
    out[slot] = ij;
    if (counted) {
      const int lo = ij.x < ij.y ? ij.x : ij.y;
      counted[slot] = (lo >= first && lo < last) ? 1 : 0;
    }
  }
};
// the two classes of the staged solver's constraint sweep: both bodies owned (interior: its body rows are final as
// soon as this rank's body sweep ends) / exactly one owned (boundary: needs the ghost's row from the velocity halo)
template <bool INTERIOR>
struct KeepPairClass {
  const int2* pairs;
  int first, last;
  int2* out;
  unsigned char* counted;
  __device__ bool keep(size_t k) const {
    const int2 ij = pairs[k];
    const bool oi = ij.x >= first && ij.x < last, oj = ij.y >= first && ij.y < last;
    return INTERIOR ? (oi && oj) : (oi != oj);
  }
  __device__ void emit(size_t k, size_t slot) const {
    const int2 ij = pairs[k];
    out[slot] = ij;
    if (counted) {
      const int lo = ij.x < ij.y ? ij.x : ij.y;
      counted[slot] = (lo >= first && lo < last) ? 1 : 0;
    }
  }
};
struct KeepBoxOverlap {  // closed interval test of geom::intersects (AABB.hpp:420-431) on the grown box
  const double* aabb;
  double buffer;
  Box box;
  int32_t* out;
  __device__ bool keep(size_t i) const {
    const double* b = aabb + 6 * i;
    const bool disjoint = (b[3] + buffer) < box.lo.x || (b[4] + buffer) < box.lo.y || (b[5] + buffer) < box.lo.z ||
                          box.hi.x < (b[0] - buffer) || box.hi.y < (b[1] - buffer) || box.hi.z < (b[2] - buffer);
    return !disjoint;
  }
  __device__ void emit(size_t i, size_t slot) const { out[slot] = static_cast<int32_t>(i); }
};

// The same test against a rank described by several boxes: boxes [nboxes + 1][6] on the device, the last one their
// union (a quick reject).  A rank's curve-ordered bodies are cut into nboxes consecutive chunks, each a compact blob, so
// the union of the chunk boxes hugs the rank's true region where one bounding box of a curve range does not (a range
// that ends a little past an octant of the Hilbert curve has a box a whole slab larger).
struct KeepAnyBoxOverlap {
  const double* aabb;
  double buffer;
  const double* boxes;
  int nboxes;
  int32_t* out;
  __device__ static bool meets(const double* b, double buffer, const double* q) {
    const bool disjoint = (b[3] + buffer) < q[0] || (b[4] + buffer) < q[1] || (b[5] + buffer) < q[2] ||
                          q[3] < (b[0] - buffer) || q[4] < (b[1] - buffer) || q[5] < (b[2] - buffer);
    return !disjoint;
  }
  __device__ bool keep(size_t i) const {
    const double* b = aabb + 6 * i;
    if (!meets(b, buffer, boxes + 6 * (size_t)nboxes)) return false;
    for (int k = 0; k < nboxes; ++k)
      if (meets(b, buffer, boxes + 6 * (size_t)k)) return true;
    return false;
  }
  __device__ void emit(size_t i, size_t slot) const { out[slot] = static_cast<int32_t>(i); }
};

// workgroup k: bounds of the grown boxes of chunk k = bodies [k * per, (k + 1) * per); an empty chunk gives the inverted
// box, which meets nothing.  Workgroup nchunks: the union, from the chunk boxes (second launch).
__global__ void __launch_bounds__(kBlock) k_chunk_bounds(size_t n, size_t per, const double* __restrict__ aabb,
                                                        double buffer, double* __restrict__ out) {
  __shared__ double scratch[kBlock / 64];
  const size_t first = blockIdx.x * per, last = (first + per < n) ? first + per : n;
  double lo[3] = {1.7976931348623157e308, 1.7976931348623157e308, 1.7976931348623157e308};
  double hi[3] = {-1.7976931348623157e308, -1.7976931348623157e308, -1.7976931348623157e308};
  for (size_t i = first + threadIdx.x; i < last; i += blockDim.x) {
    const double* b = aabb + 6 * i;
    for (int k = 0; k < 3; ++k) {
      lo[k] = dmin(lo[k], b[k] - buffer);
      hi[k] = dmax(hi[k], b[3 + k] + buffer);
    }
  }
  for (int k = 0; k < 3; ++k) {
    const double a = -block_max(-lo[k], scratch);
    const double b = block_max(hi[k], scratch);
    if (threadIdx.x == 0) {
      out[6 * blockIdx.x + k] = a;
      out[6 * blockIdx.x + 3 + k] = b;
    }
  }
}
__global__ void k_union_box(int nboxes, double* __restrict__ boxes) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double lo[3] = {1.7976931348623157e308, 1.7976931348623157e308, 1.7976931348623157e308};
  double hi[3] = {-1.7976931348623157e308, -1.7976931348623157e308, -1.7976931348623157e308};
  for (int i = 0; i < nboxes; ++i)
    for (int k = 0; k < 3; ++k) {
      lo[k] = dmin(lo[k], boxes[6 * i + k]);
      hi[k] = dmax(hi[k], boxes[6 * i + 3 + k]);
    }
  for (int k = 0; k < 3; ++k) {
    boxes[6 * (size_t)nboxes + k] = lo[k];
    boxes[6 * (size_t)nboxes + 3 + k] = hi[k];
  }
}

template <class Op>
__global__ void __launch_bounds__(kBlock) k_compact_count(size_t n, Op op, int32_t* __restrict__ tile_count) {
  __shared__ int seg[kCompactSegs];
  const size_t base = (size_t)blockIdx.x * kCompactTile;
  const int wave = threadIdx.x >> 6;
#pragma unroll
  for (int r = 0; r < kCompactRounds; ++r) {
    const size_t k = base + (size_t)r * kBlock + threadIdx.x;
    const bool keep = (k < n) && op.keep(k);
    const unsigned long long mask = __ballot(keep);
    if ((threadIdx.x & 63) == 0) seg[r * (kBlock / 64) + wave] = __popcll(mask);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int total = 0;
    for (int i = 0; i < kCompactSegs; ++i) total += seg[i];
    tile_count[blockIdx.x] = total;
  }
}
template <class Op>
__global__ void __launch_bounds__(kBlock) k_compact_emit(size_t n, Op op, const int32_t* __restrict__ tile_base) {
  __shared__ int seg[kCompactSegs + 1];
  const size_t base = (size_t)blockIdx.x * kCompactTile;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  bool keep[kCompactRounds];
  unsigned long long mask[kCompactRounds];
#pragma unroll
  for (int r = 0; r < kCompactRounds; ++r) {
    const size_t k = base + (size_t)r * kBlock + threadIdx.x;
    keep[r] = (k < n) && op.keep(k);
    mask[r] = __ballot(keep[r]);
    if (lane == 0) seg[r * (kBlock / 64) + wave] = __popcll(mask[r]);
  }
  __syncthreads();
  if (threadIdx.x == 0) {  // exclusive prefix over the tile's 16 segments
    int run = 0;
    for (int i = 0; i < kCompactSegs; ++i) {
      const int c = seg[i];
      seg[i] = run;
      run += c;
    }
  }
  __syncthreads();
  const size_t out0 = static_cast<size_t>(tile_base[blockIdx.x]);
#pragma unroll
  for (int r = 0; r < kCompactRounds; ++r) {
    if (!keep[r]) continue;
    const int below = __popcll(mask[r] & ((1ull << lane) - 1ull));
    op.emit(base + (size_t)r * kBlock + threadIdx.x, out0 + seg[r * (kBlock / 64) + wave] + below);
  }
}

// count, scan of the tile counts, emit; the total lands in host memory after one synchronisation
template <class Op>
int compact(size_t n, const Op& op, size_t* count_out, hipStream_t s) {
  TraceRange trace_range("filter_view (PrefixSum + ScatterValid)");
  HaloScratch& hs = halo_scratch();
  const size_t ntiles = (n + kCompactTile - 1) / kCompactTile;
  if (int e = hs.ensure(ntiles)) return e;
  int32_t* counts = hs.flags.as<int32_t>();
  int32_t* bases = hs.pos.as<int32_t>();
  k_compact_count<<<static_cast<unsigned>(ntiles), kBlock, 0, s>>>(n, op, counts);
  MHIP_LAUNCH_CHECK();
  if (int e = exclusive_scan_i32(counts, bases, ntiles, hs.scanws.ptr, s)) return e;
  k_compact_emit<<<static_cast<unsigned>(ntiles), kBlock, 0, s>>>(n, op, bases);
  MHIP_LAUNCH_CHECK();
  MHIP_HIP(hipMemcpyAsync(hs.host, bases + ntiles, sizeof(int32_t), hipMemcpyDeviceToHost, s));
  MHIP_HIP(hipStreamSynchronize(s));
  int32_t total = 0;
  memcpy(&total, hs.host, sizeof(int32_t));
  *count_out = static_cast<size_t>(total);
  return MHIP_SUCCESS;
}

__global__ void __launch_bounds__(kBlock) k_box_bounds(size_t n, const double* __restrict__ aabb, double buffer,
                                                      double* __restrict__ partials) {
  __shared__ double scratch[kBlock / 64];
  double lo[3] = {1.7976931348623157e308, 1.7976931348623157e308, 1.7976931348623157e308};
  double hi[3] = {-1.7976931348623157e308, -1.7976931348623157e308, -1.7976931348623157e308};
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const double* b = aabb + 6 * i;
    for (int k = 0; k < 3; ++k) {
      lo[k] = dmin(lo[k], b[k] - buffer);
      hi[k] = dmax(hi[k], b[3 + k] + buffer);
    }
  }
  for (int k = 0; k < 3; ++k) {
    const double a = -block_max(-lo[k], scratch);
    const double b = block_max(hi[k], scratch);
    if (threadIdx.x == 0) {
      partials[6 * blockIdx.x + k] = a;
      partials[6 * blockIdx.x + 3 + k] = b;
    }
  }
}
__global__ void k_box_bounds_final(int nparts, const double* __restrict__ partials, double* __restrict__ out6) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double lo[3] = {1.7976931348623157e308, 1.7976931348623157e308, 1.7976931348623157e308};
  double hi[3] = {-1.7976931348623157e308, -1.7976931348623157e308, -1.7976931348623157e308};
  for (int i = 0; i < nparts; ++i)
    for (int k = 0; k < 3; ++k) {
      lo[k] = dmin(lo[k], partials[6 * i + k]);
      hi[k] = dmax(hi[k], partials[6 * i + 3 + k]);
    }
  for (int k = 0; k < 3; ++k) {
    out6[k] = lo[k];
    out6[3 + k] = hi[k];
  }
}

// contact compaction: the neighbour list holds every pair whose grown volumes meet; the contacts of THIS step are those
// within `cutoff` of touching.  Keeps NaN separations (coincident centres) so that they are not silently lost.
struct KeepCloseContact {
  const double* sep;
  double cutoff;
  int32_t* kept;
  __device__ bool keep(size_t k) const { return !(sep[k] > cutoff); }
  __device__ void emit(size_t k, size_t out) const { kept[out] = static_cast<int32_t>(k); }
};

}  // namespace mhip

using namespace mhip;

extern "C" {

int mhip_select_contacts(size_t c, const double* sep, double cutoff, int32_t* kept_index, size_t* count_out,
                         mhip_stream_t stream) {
  MHIP_REQUIRE(count_out != nullptr, MHIP_ERR_INVALID_ARGUMENT, "count_out is null");
  *count_out = 0;
  if (c == 0) return MHIP_SUCCESS;
  MHIP_REQUIRE(sep && kept_index, MHIP_ERR_INVALID_ARGUMENT, "sep / kept_index is null");
  MHIP_REQUIRE(c < (1u << 31), MHIP_ERR_RUNTIME, "too many contacts");
  return compact(c, KeepCloseContact{sep, cutoff, kept_index}, count_out, as_stream(stream));
}

int mhip_filter_pairs_owned(size_t c, const int32_t* pairs_in, size_t first, size_t count, int32_t* pairs_out,
                            unsigned char* counted_out, size_t* count_out, mhip_stream_t stream) {
  MHIP_REQUIRE(count_out != nullptr, MHIP_ERR_INVALID_ARGUMENT, "count_out is null");
  *count_out = 0;
  if (c == 0) return MHIP_SUCCESS;
  MHIP_REQUIRE(pairs_in && pairs_out, MHIP_ERR_INVALID_ARGUMENT, "pairs_in / pairs_out is null");
  MHIP_REQUIRE(pairs_in != pairs_out, MHIP_ERR_INVALID_ARGUMENT, "in-place filtering is not supported");
  MHIP_REQUIRE(c < (1u << 31) && first + count < (1u << 31), MHIP_ERR_RUNTIME, "too many pairs / bodies");
  const KeepOwnedPair op{reinterpret_cast<const int2*>(pairs_in), static_cast<int>(first),
                         static_cast<int>(first + count), reinterpret_cast<int2*>(pairs_out), counted_out};
  return compact(c, op, count_out, as_stream(stream));
}

int mhip_partition_pairs_owned(size_t c, const int32_t* pairs_in, size_t first, size_t count, int32_t* pairs_out,
                               unsigned char* counted_out, size_t* interior_out, size_t* count_out,
                               mhip_stream_t stream) {
  MHIP_REQUIRE(count_out != nullptr && interior_out != nullptr, MHIP_ERR_INVALID_ARGUMENT, "count outputs are null");
  *count_out = 0;
  *interior_out = 0;
  if (c == 0) return MHIP_SUCCESS;
  MHIP_REQUIRE(pairs_in && pairs_out, MHIP_ERR_INVALID_ARGUMENT, "pairs_in / pairs_out is null");
  MHIP_REQUIRE(pairs_in != pairs_out, MHIP_ERR_INVALID_ARGUMENT, "in-place filtering is not supported");
  MHIP_REQUIRE(c < (1u << 31) && first + count < (1u << 31), MHIP_ERR_RUNTIME, "too many pairs / bodies");
  const int2* in = reinterpret_cast<const int2*>(pairs_in);
  const int f = static_cast<int>(first), l = static_cast<int>(first + count);
  size_t n_int = 0, n_bnd = 0;
  const KeepPairClass<true> interior{in, f, l, reinterpret_cast<int2*>(pairs_out), counted_out};
  if (int e = compact(c, interior, &n_int, as_stream(stream))) return e;
  const KeepPairClass<false> boundary{in, f, l, reinterpret_cast<int2*>(pairs_out) + n_int,
                                      counted_out ? counted_out + n_int : nullptr};
  if (int e = compact(c, boundary, &n_bnd, as_stream(stream))) return e;
  *interior_out = n_int;
  *count_out = n_int + n_bnd;
  return MHIP_SUCCESS;
}

int mhip_select_aabb_overlap(size_t n, const double* aabb, double buffer, const double* box6, int32_t* idx_out,
                             size_t* count_out, mhip_stream_t stream) {
  MHIP_REQUIRE(count_out != nullptr && box6 != nullptr, MHIP_ERR_INVALID_ARGUMENT, "count_out / box6 is null");
  *count_out = 0;
  if (n == 0) return MHIP_SUCCESS;
  MHIP_REQUIRE(aabb && idx_out, MHIP_ERR_INVALID_ARGUMENT, "aabb / idx_out is null");
  MHIP_REQUIRE(n < (1u << 31), MHIP_ERR_RUNTIME, "too many bodies");
  const KeepBoxOverlap op{aabb, buffer, Box{{box6[0], box6[1], box6[2]}, {box6[3], box6[4], box6[5]}}, idx_out};
  return compact(n, op, count_out, as_stream(stream));
}

int mhip_aabb_chunk_bounds(size_t n, const double* aabb, double buffer, int nchunks, double* boxes,
                           mhip_stream_t stream) {
  MHIP_REQUIRE(nchunks >= 1 && nchunks <= 4096, MHIP_ERR_INVALID_ARGUMENT, "nchunks must be in [1, 4096]");
  MHIP_REQUIRE(boxes != nullptr && (n == 0 || aabb != nullptr), MHIP_ERR_INVALID_ARGUMENT, "null argument");
  hipStream_t s = as_stream(stream);
  const size_t per = (n + (size_t)nchunks - 1) / (size_t)nchunks;
  k_chunk_bounds<<<(unsigned)nchunks, kBlock, 0, s>>>(n, per ? per : 1, aabb, buffer, boxes);
  MHIP_LAUNCH_CHECK();
  k_union_box<<<1, 64, 0, s>>>(nchunks, boxes);
  MHIP_LAUNCH_CHECK();
  return MHIP_SUCCESS;
}

int mhip_select_aabb_overlap_any(size_t n, const double* aabb, double buffer, int nboxes, const double* boxes,
                                 int32_t* idx_out, size_t* count_out, mhip_stream_t stream) {
  MHIP_REQUIRE(count_out != nullptr, MHIP_ERR_INVALID_ARGUMENT, "count_out is null");
  *count_out = 0;
  MHIP_REQUIRE(nboxes >= 1 && boxes != nullptr, MHIP_ERR_INVALID_ARGUMENT, "boxes missing");
  if (n == 0) return MHIP_SUCCESS;
  MHIP_REQUIRE(aabb && idx_out, MHIP_ERR_INVALID_ARGUMENT, "aabb / idx_out is null");
  MHIP_REQUIRE(n < (1u << 31), MHIP_ERR_RUNTIME, "too many bodies");
  const KeepAnyBoxOverlap op{aabb, buffer, boxes, nboxes, idx_out};
  return compact(n, op, count_out, as_stream(stream));
}

int mhip_aabb_bounds(size_t n, const double* aabb, double buffer, double* out6, mhip_stream_t stream) {
  MHIP_REQUIRE(out6 != nullptr, MHIP_ERR_INVALID_ARGUMENT, "out6 is null");
  for (int k = 0; k < 3; ++k) {
    out6[k] = 1.7976931348623157e308;
    out6[3 + k] = -1.7976931348623157e308;
  }
  if (n == 0) return MHIP_SUCCESS;
  MHIP_REQUIRE(aabb != nullptr, MHIP_ERR_INVALID_ARGUMENT, "aabb is null");
  hipStream_t s = as_stream(stream);
  HaloScratch& hs = halo_scratch();
  if (int e = hs.ensure(16)) return e;
  const unsigned g = grid_for(n);
  double* parts = hs.partials.as<double>();
  double* dout = parts + 6 * kMaxGrid;
  k_box_bounds<<<g, kBlock, 0, s>>>(n, aabb, buffer, parts);
  MHIP_LAUNCH_CHECK();
  k_box_bounds_final<<<1, 64, 0, s>>>((int)g, parts, dout);
  MHIP_LAUNCH_CHECK();
  MHIP_HIP(hipMemcpyAsync(out6, dout, 6 * sizeof(double), hipMemcpyDeviceToHost, s));
  MHIP_HIP(hipStreamSynchronize(s));
  return MHIP_SUCCESS;
}

}  // extern "C"
