// halo.hip -- stream compaction helpers of the domain-decomposed path: pair filtering by ownership,
// ghost-candidate selection by box overlap, rank bounding boxes.  Flag -> exclusive scan -> stable scatter, the
// structure of filter_view (mundy_mesh/GenNeighborLinkers.hpp:141-183).  Integer work, HBM bound.
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

__global__ void __launch_bounds__(kBlock) k_flag_pairs(size_t c, const int2* __restrict__ pairs, int first, int last,
                                                      int32_t* __restrict__ flags) {
  for (size_t k = blockIdx.x * (size_t)blockDim.x + threadIdx.x; k < c; k += (size_t)gridDim.x * blockDim.x) {
    const int2 ij = pairs[k];
    const bool oi = ij.x >= first && ij.x < last, oj = ij.y >= first && ij.y < last;
    flags[k] = (oi || oj) ? 1 : 0;
  }
}
__global__ void __launch_bounds__(kBlock)
    k_scatter_pairs(size_t c, const int2* __restrict__ pairs, const int32_t* __restrict__ flags,
                    const int32_t* __restrict__ pos, int first, int last, int2* __restrict__ out,
                    unsigned char* __restrict__ counted) {
  for (size_t k = blockIdx.x * (size_t)blockDim.x + threadIdx.x; k < c; k += (size_t)gridDim.x * blockDim.x) {
    if (!flags[k]) continue;
    const int2 ij = pairs[k];
    out[pos[k]] = ij;
    if (counted) {
      const int lo = ij.x < ij.y ? ij.x : ij.y;
      counted[pos[k]] = (lo >= first && lo < last) ? 1 : 0;
    }
  }
}
__global__ void __launch_bounds__(kBlock) k_flag_overlap(size_t n, const double* __restrict__ aabb, double buffer,
                                                        Box box, int32_t* __restrict__ flags) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const double* b = aabb + 6 * i;
    // closed interval test of geom::intersects (AABB.hpp:420-431) on the grown box
    const bool disjoint = (b[3] + buffer) < box.lo.x || (b[4] + buffer) < box.lo.y || (b[5] + buffer) < box.lo.z ||
                          box.hi.x < (b[0] - buffer) || box.hi.y < (b[1] - buffer) || box.hi.z < (b[2] - buffer);
    flags[i] = disjoint ? 0 : 1;
  }
}
__global__ void __launch_bounds__(kBlock) k_scatter_index(size_t n, const int32_t* __restrict__ flags,
                                                         const int32_t* __restrict__ pos, int32_t* __restrict__ out) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    if (flags[i]) out[pos[i]] = static_cast<int32_t>(i);
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

}  // namespace mhip

using namespace mhip;

extern "C" {

int mhip_filter_pairs_owned(size_t c, const int32_t* pairs_in, size_t first, size_t count, int32_t* pairs_out,
                            unsigned char* counted_out, size_t* count_out, mhip_stream_t stream) {
  MHIP_REQUIRE(count_out != nullptr, MHIP_ERR_INVALID_ARGUMENT, "count_out is null");
  *count_out = 0;
  if (c == 0) return MHIP_SUCCESS;
  MHIP_REQUIRE(pairs_in && pairs_out, MHIP_ERR_INVALID_ARGUMENT, "pairs_in / pairs_out is null");
  MHIP_REQUIRE(pairs_in != pairs_out, MHIP_ERR_INVALID_ARGUMENT, "in-place filtering is not supported");
  MHIP_REQUIRE(c < (1u << 31) && first + count < (1u << 31), MHIP_ERR_RUNTIME, "too many pairs / bodies");
  hipStream_t s = as_stream(stream);
  HaloScratch& hs = halo_scratch();
  if (int e = hs.ensure(c)) return e;
  const int f = static_cast<int>(first), l = static_cast<int>(first + count);
  const int2* in = reinterpret_cast<const int2*>(pairs_in);
  k_flag_pairs<<<grid_for(c), kBlock, 0, s>>>(c, in, f, l, hs.flags.as<int32_t>());
  MHIP_LAUNCH_CHECK();
  if (int e = exclusive_scan_i32(hs.flags.as<int32_t>(), hs.pos.as<int32_t>(), c, hs.scanws.ptr, s)) return e;
  k_scatter_pairs<<<grid_for(c), kBlock, 0, s>>>(c, in, hs.flags.as<int32_t>(), hs.pos.as<int32_t>(), f, l,
                                                reinterpret_cast<int2*>(pairs_out), counted_out);
  MHIP_LAUNCH_CHECK();
  int32_t total = 0;
  MHIP_HIP(hipMemcpyAsync(hs.host, hs.pos.as<int32_t>() + c, sizeof(int32_t), hipMemcpyDeviceToHost, s));
  MHIP_HIP(hipStreamSynchronize(s));
  memcpy(&total, hs.host, sizeof(int32_t));
  *count_out = static_cast<size_t>(total);
  return MHIP_SUCCESS;
}

int mhip_select_aabb_overlap(size_t n, const double* aabb, double buffer, const double* box6, int32_t* idx_out,
                             size_t* count_out, mhip_stream_t stream) {
  MHIP_REQUIRE(count_out != nullptr && box6 != nullptr, MHIP_ERR_INVALID_ARGUMENT, "count_out / box6 is null");
  *count_out = 0;
  if (n == 0) return MHIP_SUCCESS;
  MHIP_REQUIRE(aabb && idx_out, MHIP_ERR_INVALID_ARGUMENT, "aabb / idx_out is null");
  MHIP_REQUIRE(n < (1u << 31), MHIP_ERR_RUNTIME, "too many bodies");
  hipStream_t s = as_stream(stream);
  HaloScratch& hs = halo_scratch();
  if (int e = hs.ensure(n)) return e;
  const Box box{{box6[0], box6[1], box6[2]}, {box6[3], box6[4], box6[5]}};
  k_flag_overlap<<<grid_for(n), kBlock, 0, s>>>(n, aabb, buffer, box, hs.flags.as<int32_t>());
  MHIP_LAUNCH_CHECK();
  if (int e = exclusive_scan_i32(hs.flags.as<int32_t>(), hs.pos.as<int32_t>(), n, hs.scanws.ptr, s)) return e;
  k_scatter_index<<<grid_for(n), kBlock, 0, s>>>(n, hs.flags.as<int32_t>(), hs.pos.as<int32_t>(), idx_out);
  MHIP_LAUNCH_CHECK();
  int32_t total = 0;
  MHIP_HIP(hipMemcpyAsync(hs.host, hs.pos.as<int32_t>() + n, sizeof(int32_t), hipMemcpyDeviceToHost, s));
  MHIP_HIP(hipStreamSynchronize(s));
  memcpy(&total, hs.host, sizeof(int32_t));
  *count_out = static_cast<size_t>(total);
  return MHIP_SUCCESS;
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
