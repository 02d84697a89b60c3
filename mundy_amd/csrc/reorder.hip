// reorder.hip -- Z-order (Morton) body reordering, row gathers and the explicit Euler update either side of the solve.
//
// Bodies are binned on a lattice of edge `cell_size` anchored at `lo`; the Morton code interleaves the lattice
// coordinates with z most significant, which is the order zorder_knn::Less (mundy_math/zmort.hpp:195-220) gives to
// non-negative lattice points.  The sort is a counting sort over codes (histogram, scan, scatter) followed by a
// per-code tie-break by body index, so the permutation is deterministic.  Integer work: HBM/atomic bound.
#include "geom_device.hpp"

namespace mhip {

__device__ inline unsigned spread3(unsigned v) {  // 10 bits -> every third bit
  v &= 0x3ffu;
  v = (v | (v << 16)) & 0x030000ffu;
  v = (v | (v << 8)) & 0x0300f00fu;
  v = (v | (v << 4)) & 0x030c30c3u;
  v = (v | (v << 2)) & 0x09249249u;
  return v;
}

__device__ inline unsigned morton_code(const double* __restrict__ center, size_t i, V3 lo, double inv_cell, int bits) {
  const V3 c = load3(center, i);
  const int maxc = (1 << bits) - 1;
  int ix = static_cast<int>(floor((c.x - lo.x) * inv_cell));
  int iy = static_cast<int>(floor((c.y - lo.y) * inv_cell));
  int iz = static_cast<int>(floor((c.z - lo.z) * inv_cell));
  ix = ix < 0 ? 0 : (ix > maxc ? maxc : ix);
  iy = iy < 0 ? 0 : (iy > maxc ? maxc : iy);
  iz = iz < 0 ? 0 : (iz > maxc ? maxc : iz);
  return spread3((unsigned)ix) | (spread3((unsigned)iy) << 1) | (spread3((unsigned)iz) << 2);
}

__global__ void __launch_bounds__(kBlock)
    k_morton_count(size_t n, const double* __restrict__ center, V3 lo, double inv_cell, int bits,
                   unsigned* __restrict__ code, int32_t* __restrict__ hist) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const unsigned m = morton_code(center, i, lo, inv_cell, bits);
    code[i] = m;
    atomicAdd(&hist[m], 1);
  }
}
// curve order by table: code = key_table[ix][iy][iz] on a (2^bits)^3 lattice (the Hilbert visiting index of the cell,
// generated on the host by the recursion of mundy_math/Hilbert.hpp:48-83)
__global__ void __launch_bounds__(kBlock)
    k_table_count(size_t n, const double* __restrict__ center, V3 lo, V3 span, int bits,
                  const int32_t* __restrict__ key_table, unsigned* __restrict__ code, int32_t* __restrict__ hist) {
  const int ns = 1 << bits, maxc = ns - 1;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const V3 c = load3(center, i);
    const double nsd = static_cast<double>(ns);  // ((c - lo) / span) * ns, the host partitioner's expression
    double fx = floor((c.x - lo.x) / span.x * nsd), fy = floor((c.y - lo.y) / span.y * nsd),
           fz = floor((c.z - lo.z) / span.z * nsd);
    fx = fx < 0.0 ? 0.0 : (fx > maxc ? (double)maxc : fx);  // clamp before the integer conversion
    fy = fy < 0.0 ? 0.0 : (fy > maxc ? (double)maxc : fy);
    fz = fz < 0.0 ? 0.0 : (fz > maxc ? (double)maxc : fz);
    const int ix = static_cast<int>(fx), iy = static_cast<int>(fy), iz = static_cast<int>(fz);
    const unsigned m = static_cast<unsigned>(key_table[((size_t)ix * ns + iy) * ns + iz]);
    code[i] = m;
    atomicAdd(&hist[m], 1);
  }
}
__global__ void __launch_bounds__(kBlock) k_morton_scatter(size_t n, const unsigned* __restrict__ code,
                                                          int32_t* __restrict__ cursor, int32_t* __restrict__ perm) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    perm[atomicAdd(&cursor[code[i]], 1)] = static_cast<int32_t>(i);
}
__global__ void __launch_bounds__(kBlock) k_segment_sort(size_t ncodes, const int32_t* __restrict__ ptr,
                                                        int32_t* __restrict__ perm) {
  for (size_t m = blockIdx.x * (size_t)blockDim.x + threadIdx.x; m < ncodes; m += (size_t)gridDim.x * blockDim.x) {
    const int32_t beg = ptr[m], end = ptr[m + 1];
    for (int32_t a = beg + 1; a < end; ++a) {
      const int32_t v = perm[a];
      int32_t b = a - 1;
      while (b >= beg && perm[b] > v) {
        perm[b + 1] = perm[b];
        --b;
      }
      perm[b + 1] = v;
    }
  }
}

__global__ void __launch_bounds__(kBlock)
    k_gather_rows(size_t n, size_t width, const int32_t* __restrict__ perm, const double* __restrict__ src,
                  double* __restrict__ dst) {
  const size_t total = n * width;
  for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const size_t k = e / width, w = e - k * width;
    dst[e] = src[(size_t)perm[k] * width + w];
  }
}

// x += dt*U (NgpLcp.cpp:898); q <- rotate_quaternion(q, W, dt) (mundy_math/Quaternion.hpp:1366-1390)
__global__ void __launch_bounds__(kBlock)
    k_integrate(size_t n, double dt, const double* __restrict__ vel, double* __restrict__ center,
                double* __restrict__ quat) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const double* v = vel + 6 * i;
    center[3 * i] = dt * v[0] + 1.0 * center[3 * i];  // axpby(dt, U, 1, x)
    center[3 * i + 1] = dt * v[1] + 1.0 * center[3 * i + 1];
    center[3 * i + 2] = dt * v[2] + 1.0 * center[3 * i + 2];
    if (quat) {
      const V3 om{v[3], v[4], v[5]};
      const double w = norm(om);
      if (w < kZeroTol) continue;
      const double winv = 1.0 / w;
      double sw, cw;
      det_sincos(0.5 * w * dt, sw, cw);
      const Quat q = load4q(quat, i);
      const double s = q.w;
      const V3 p{q.x, q.y, q.z};
      const V3 cr = cross(om, p);
      // xyz = s*sw*omega*winv + cw*p + sw*winv*cross(omega, p), left to right
      const double a = s * sw, b = sw * winv;
      const V3 xyz{a * om.x * winv + cw * p.x + b * cr.x, a * om.y * winv + cw * p.y + b * cr.y,
                   a * om.z * winv + cw * p.z + b * cr.z};
      const double qw = s * cw - dot(om, p) * sw * winv;
      const double inv = 1.0 / sqrt(qw * qw + xyz.x * xyz.x + xyz.y * xyz.y + xyz.z * xyz.z);
      quat[4 * i] = qw * inv;
      quat[4 * i + 1] = xyz.x * inv;
      quat[4 * i + 2] = xyz.y * inv;
      quat[4 * i + 3] = xyz.z * inv;
    }
  }
}

// PeriodicScaledMetric::sep / wrap (mundy_geom/periodicity.hpp:812-823); wrap_rigid of a Sphere / Spherocylinder /
// Ellipsoid wraps its centre, orientation and size untouched (:1088-1113, :1156-1160)
template <class Metric>
__global__ void __launch_bounds__(kBlock) k_periodic_sep(size_t n, Metric pm, const double* __restrict__ p1,
                                                        const double* __restrict__ p2, double* __restrict__ out) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    store3(out, i, periodic_sep(pm, load3(p1, i), load3(p2, i)));
}
template <class Metric>
__global__ void __launch_bounds__(kBlock) k_wrap_rigid(size_t n, Metric pm, double* __restrict__ center) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    store3(center, i, periodic_wrap(pm, load3(center, i)));
}
// PeriodicMetric::shift_image: translate(point, h * num_images)  (periodicity.hpp:323-327)
__global__ void __launch_bounds__(kBlock) k_shift_image(size_t n, Triclinic pm, const double* __restrict__ p,
                                                       const int32_t* __restrict__ images, double* __restrict__ out) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const V3 k{static_cast<double>(images[3 * i]), static_cast<double>(images[3 * i + 1]),
               static_cast<double>(images[3 * i + 2])};
    store3(out, i, load3(p, i) + matvec3(pm.h, k));
  }
}

struct ReorderScratch {
  DeviceBuffer code, hist, ptr, scanws;
};
ReorderScratch& reorder_scratch() {
  thread_local ReorderScratch s;
  return s;
}

}  // namespace mhip

using namespace mhip;

extern "C" {

int mhip_morton_order(size_t n, const double* center, const double* lo, double cell_size, int32_t* perm,
                      mhip_stream_t stream) {
  TraceRange trace_range("zmorton reorder");
  MHIP_REQUIRE(n == 0 || (center && perm), MHIP_ERR_INVALID_ARGUMENT, "center / perm is null");
  MHIP_REQUIRE(lo != nullptr, MHIP_ERR_INVALID_ARGUMENT, "lo is null");
  MHIP_REQUIRE(cell_size > 0.0, MHIP_ERR_INVALID_ARGUMENT, "cell_size must be positive");
  MHIP_REQUIRE(n < (1u << 31), MHIP_ERR_RUNTIME, "too many bodies");
  if (n == 0) return MHIP_SUCCESS;
  hipStream_t s = as_stream(stream);
  // lattice resolution: the finest of 2^bits per axis whose code space stays within 8 codes per body (>= 2^12)
  int bits = 4;
  while (bits < 8 && (size_t(1) << (3 * (bits + 1))) <= 8 * n) ++bits;
  const size_t ncodes = size_t(1) << (3 * bits);
  ReorderScratch& rs = reorder_scratch();
  if (int e = rs.code.reserve(n * sizeof(unsigned))) return e;
  if (int e = rs.hist.reserve((ncodes + 2) * sizeof(int32_t))) return e;
  if (int e = rs.ptr.reserve((ncodes + 2) * sizeof(int32_t))) return e;
  if (int e = rs.scanws.reserve(scan_workspace_bytes(ncodes + 2) + 64)) return e;
  MHIP_HIP(hipMemsetAsync(rs.hist.ptr, 0, (ncodes + 1) * sizeof(int32_t), s));
  const V3 l{lo[0], lo[1], lo[2]};
  k_morton_count<<<grid_for(n), kBlock, 0, s>>>(n, center, l, 1.0 / cell_size, bits, rs.code.as<unsigned>(),
                                               rs.hist.as<int32_t>());
  MHIP_LAUNCH_CHECK();
  if (int e = exclusive_scan_i32(rs.hist.as<int32_t>(), rs.ptr.as<int32_t>(), ncodes, rs.scanws.ptr, s)) return e;
  MHIP_HIP(hipMemcpyAsync(rs.hist.ptr, rs.ptr.ptr, (ncodes + 1) * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
  k_morton_scatter<<<grid_for(n), kBlock, 0, s>>>(n, rs.code.as<unsigned>(), rs.hist.as<int32_t>(), perm);
  MHIP_LAUNCH_CHECK();
  k_segment_sort<<<grid_for(ncodes), kBlock, 0, s>>>(ncodes, rs.ptr.as<int32_t>(), perm);
  MHIP_LAUNCH_CHECK();
  return MHIP_SUCCESS;
}

int mhip_curve_order(size_t n, const double* center, const double* lo, const double* hi, int level,
                     const int32_t* key_table, int32_t* perm, mhip_stream_t stream) {
  TraceRange trace_range("hilbert reorder");
  MHIP_REQUIRE(n == 0 || (center && perm), MHIP_ERR_INVALID_ARGUMENT, "center / perm is null");
  MHIP_REQUIRE(lo != nullptr && hi != nullptr && key_table != nullptr, MHIP_ERR_INVALID_ARGUMENT,
               "lo / hi / key_table is null");
  MHIP_REQUIRE(level >= 1 && level <= 8, MHIP_ERR_INVALID_ARGUMENT, "level must be in [1, 8], got %d", level);
  MHIP_REQUIRE(hi[0] > lo[0] && hi[1] > lo[1] && hi[2] > lo[2], MHIP_ERR_INVALID_ARGUMENT, "empty domain");
  MHIP_REQUIRE(n < (1u << 31), MHIP_ERR_RUNTIME, "too many bodies");
  if (n == 0) return MHIP_SUCCESS;
  hipStream_t s = as_stream(stream);
  const size_t ncodes = size_t(1) << (3 * level);
  ReorderScratch& rs = reorder_scratch();
  if (int e = rs.code.reserve(n * sizeof(unsigned))) return e;
  if (int e = rs.hist.reserve((ncodes + 2) * sizeof(int32_t))) return e;
  if (int e = rs.ptr.reserve((ncodes + 2) * sizeof(int32_t))) return e;
  if (int e = rs.scanws.reserve(scan_workspace_bytes(ncodes + 2) + 64)) return e;
  MHIP_HIP(hipMemsetAsync(rs.hist.ptr, 0, (ncodes + 1) * sizeof(int32_t), s));
  const V3 l{lo[0], lo[1], lo[2]}, span{hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]};
  k_table_count<<<grid_for(n), kBlock, 0, s>>>(n, center, l, span, level, key_table, rs.code.as<unsigned>(),
                                              rs.hist.as<int32_t>());
  MHIP_LAUNCH_CHECK();
  if (int e = exclusive_scan_i32(rs.hist.as<int32_t>(), rs.ptr.as<int32_t>(), ncodes, rs.scanws.ptr, s)) return e;
  MHIP_HIP(hipMemcpyAsync(rs.hist.ptr, rs.ptr.ptr, (ncodes + 1) * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
  k_morton_scatter<<<grid_for(n), kBlock, 0, s>>>(n, rs.code.as<unsigned>(), rs.hist.as<int32_t>(), perm);
  MHIP_LAUNCH_CHECK();
  k_segment_sort<<<grid_for(ncodes), kBlock, 0, s>>>(ncodes, rs.ptr.as<int32_t>(), perm);
  MHIP_LAUNCH_CHECK();
  return MHIP_SUCCESS;
}

int mhip_curve_keys(size_t n, const double* center, const double* lo, const double* hi, int level,
                    const int32_t* key_table, uint32_t* keys, mhip_stream_t stream) {
  MHIP_REQUIRE(n == 0 || (center && keys), MHIP_ERR_INVALID_ARGUMENT, "center / keys is null");
  MHIP_REQUIRE(lo != nullptr && hi != nullptr && key_table != nullptr, MHIP_ERR_INVALID_ARGUMENT,
               "lo / hi / key_table is null");
  MHIP_REQUIRE(level >= 1 && level <= 8, MHIP_ERR_INVALID_ARGUMENT, "level must be in [1, 8], got %d", level);
  MHIP_REQUIRE(hi[0] > lo[0] && hi[1] > lo[1] && hi[2] > lo[2], MHIP_ERR_INVALID_ARGUMENT, "empty domain");
  if (n == 0) return MHIP_SUCCESS;
  hipStream_t s = as_stream(stream);
  const size_t ncodes = size_t(1) << (3 * level);
  ReorderScratch& rs = reorder_scratch();
  if (int e = rs.hist.reserve((ncodes + 2) * sizeof(int32_t))) return e;  // the counting kernel also fills a histogram
  MHIP_HIP(hipMemsetAsync(rs.hist.ptr, 0, (ncodes + 1) * sizeof(int32_t), s));
  const V3 l{lo[0], lo[1], lo[2]}, span{hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]};
  k_table_count<<<grid_for(n), kBlock, 0, s>>>(n, center, l, span, level, key_table, keys, rs.hist.as<int32_t>());
  MHIP_LAUNCH_CHECK();
  return MHIP_SUCCESS;
}

struct SortScratch {
  DeviceBuffer keys, keys_tmp, vals, vals_tmp, ws;
};
static SortScratch& sort_scratch() {
  thread_local SortScratch s;
  return s;
}
__global__ void __launch_bounds__(kBlock) k_iota_u32(size_t n, unsigned* __restrict__ out) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    out[i] = static_cast<unsigned>(i);
}

int mhip_sort_by_key_u64(size_t n, const uint64_t* keys, int32_t* perm, mhip_stream_t stream) {
  MHIP_REQUIRE(n == 0 || (keys && perm), MHIP_ERR_INVALID_ARGUMENT, "keys / perm is null");
  MHIP_REQUIRE(n < (1u << 31), MHIP_ERR_RUNTIME, "too many keys");
  if (n == 0) return MHIP_SUCCESS;
  hipStream_t s = as_stream(stream);
  SortScratch& ss = sort_scratch();
  if (int e = ss.keys.reserve(n * sizeof(unsigned long long))) return e;
  if (int e = ss.keys_tmp.reserve(n * sizeof(unsigned long long))) return e;
  if (int e = ss.vals_tmp.reserve(n * sizeof(unsigned))) return e;
  if (int e = ss.ws.reserve(radix_sort_workspace_bytes(n))) return e;
  MHIP_HIP(hipMemcpyAsync(ss.keys.ptr, keys, n * sizeof(unsigned long long), hipMemcpyDeviceToDevice, s));
  unsigned* vals = reinterpret_cast<unsigned*>(perm);
  k_iota_u32<<<grid_for(n), kBlock, 0, s>>>(n, vals);
  MHIP_LAUNCH_CHECK();
  return radix_sort_u64(n, ss.keys.as<unsigned long long>(), vals, ss.keys_tmp.as<unsigned long long>(),
                        ss.vals_tmp.as<unsigned>(), 8, ss.ws.ptr, s);
}

int mhip_gather_rows(size_t n, size_t width, const int32_t* perm, const double* src, double* dst,
                     mhip_stream_t stream) {
  MHIP_REQUIRE(n == 0 || (perm && src && dst), MHIP_ERR_INVALID_ARGUMENT, "null argument");
  MHIP_REQUIRE(width > 0, MHIP_ERR_INVALID_ARGUMENT, "width must be positive");
  MHIP_REQUIRE(src != dst, MHIP_ERR_INVALID_ARGUMENT, "gather cannot run in place");
  if (n == 0) return MHIP_SUCCESS;
  k_gather_rows<<<grid_for(n * width), kBlock, 0, as_stream(stream)>>>(n, width, perm, src, dst);
  MHIP_LAUNCH_CHECK();
  return MHIP_SUCCESS;
}

int mhip_copy_strided(size_t n, size_t width, const double* src, size_t src_stride, double* dst, size_t dst_stride,
                      mhip_stream_t stream) {
  MHIP_REQUIRE(width > 0 && src_stride >= width && dst_stride >= width, MHIP_ERR_INVALID_ARGUMENT,
               "strides (%zu, %zu) must be at least the width %zu", src_stride, dst_stride, width);
  if (n == 0) return MHIP_SUCCESS;
  MHIP_REQUIRE(src && dst, MHIP_ERR_INVALID_ARGUMENT, "null argument");
  MHIP_HIP(hipMemcpy2DAsync(dst, dst_stride * sizeof(double), src, src_stride * sizeof(double), width * sizeof(double), n,
                            hipMemcpyDeviceToDevice, as_stream(stream)));
  return MHIP_SUCCESS;
}

int mhip_periodic_sep(size_t n, const double* box, const double* p1, const double* p2, double* out,
                      mhip_stream_t stream) {
  MHIP_REQUIRE(box != nullptr && box[0] > 0 && box[1] > 0 && box[2] > 0, MHIP_ERR_INVALID_ARGUMENT,
               "periodic box must be positive");
  if (n == 0) return MHIP_SUCCESS;
  MHIP_REQUIRE(p1 && p2 && out, MHIP_ERR_INVALID_ARGUMENT, "null argument");
  k_periodic_sep<<<grid_for(n), kBlock, 0, as_stream(stream)>>>(n, make_periodic(box), p1, p2, out);
  MHIP_LAUNCH_CHECK();
  return MHIP_SUCCESS;
}

int mhip_wrap_rigid(size_t n, const double* box, double* center, mhip_stream_t stream) {
  MHIP_REQUIRE(box != nullptr && box[0] > 0 && box[1] > 0 && box[2] > 0, MHIP_ERR_INVALID_ARGUMENT,
               "periodic box must be positive");
  if (n == 0) return MHIP_SUCCESS;
  MHIP_REQUIRE(center != nullptr, MHIP_ERR_INVALID_ARGUMENT, "center is null");
  k_wrap_rigid<<<grid_for(n), kBlock, 0, as_stream(stream)>>>(n, make_periodic(box), center);
  MHIP_LAUNCH_CHECK();
  return MHIP_SUCCESS;
}

int mhip_unit_cell_inverse(const double* cell, double* cell_inv) {
  MHIP_REQUIRE(cell && cell_inv, MHIP_ERR_INVALID_ARGUMENT, "null argument");
  MHIP_REQUIRE(determinant3(cell) != 0.0, MHIP_ERR_INVALID_ARGUMENT, "unit cell matrix is singular");
  inverse3(cell, cell_inv);
  return MHIP_SUCCESS;
}

int mhip_periodic_sep_triclinic(size_t n, const double* cell, const double* p1, const double* p2, double* out,
                                mhip_stream_t stream) {
  MHIP_REQUIRE(cell != nullptr && determinant3(cell) != 0.0, MHIP_ERR_INVALID_ARGUMENT, "unit cell matrix is singular");
  if (n == 0) return MHIP_SUCCESS;
  MHIP_REQUIRE(p1 && p2 && out, MHIP_ERR_INVALID_ARGUMENT, "null argument");
  k_periodic_sep<<<grid_for(n), kBlock, 0, as_stream(stream)>>>(n, make_triclinic(cell), p1, p2, out);
  MHIP_LAUNCH_CHECK();
  return MHIP_SUCCESS;
}

int mhip_wrap_rigid_triclinic(size_t n, const double* cell, double* center, mhip_stream_t stream) {
  MHIP_REQUIRE(cell != nullptr && determinant3(cell) != 0.0, MHIP_ERR_INVALID_ARGUMENT, "unit cell matrix is singular");
  if (n == 0) return MHIP_SUCCESS;
  MHIP_REQUIRE(center != nullptr, MHIP_ERR_INVALID_ARGUMENT, "center is null");
  k_wrap_rigid<<<grid_for(n), kBlock, 0, as_stream(stream)>>>(n, make_triclinic(cell), center);
  MHIP_LAUNCH_CHECK();
  return MHIP_SUCCESS;
}

int mhip_shift_image_triclinic(size_t n, const double* cell, const double* p, const int32_t* images, double* out,
                               mhip_stream_t stream) {
  MHIP_REQUIRE(cell != nullptr, MHIP_ERR_INVALID_ARGUMENT, "unit cell matrix is null");
  if (n == 0) return MHIP_SUCCESS;
  MHIP_REQUIRE(p && images && out, MHIP_ERR_INVALID_ARGUMENT, "null argument");
  k_shift_image<<<grid_for(n), kBlock, 0, as_stream(stream)>>>(n, make_triclinic(cell), p, images, out);
  MHIP_LAUNCH_CHECK();
  return MHIP_SUCCESS;
}

int mhip_integrate_euler(size_t n, double dt, const double* velocity, double* center, double* quat,
                         mhip_stream_t stream) {
  TraceRange trace_range("integrate");
  MHIP_REQUIRE(n == 0 || (velocity && center), MHIP_ERR_INVALID_ARGUMENT, "velocity / center is null");
  if (n == 0) return MHIP_SUCCESS;
  k_integrate<<<grid_for(n), kBlock, 0, as_stream(stream)>>>(n, dt, velocity, center, quat);
  MHIP_LAUNCH_CHECK();
  return MHIP_SUCCESS;
}

}  // extern "C"

namespace mhip {
__global__ void __launch_bounds__(kBlock) k_compose_keys(size_t n, const uint32_t* __restrict__ major,
                                                        const double* __restrict__ minor, int shift,
                                                        unsigned long long* __restrict__ out) {
  for (size_t i = blockIdx.x * (size_t)kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock)
    out[i] = (static_cast<unsigned long long>(major[i]) << shift) | static_cast<unsigned long long>(minor[i]);
}
__global__ void __launch_bounds__(kBlock) k_sequence(size_t n, double first, double* __restrict__ out) {
  for (size_t i = blockIdx.x * (size_t)kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock)
    out[i] = first + static_cast<double>(i);
}
__global__ void __launch_bounds__(kBlock) k_pair_degree(size_t c, const int2* __restrict__ pairs, size_t first,
                                                       size_t count, double* __restrict__ weights) {
  for (size_t k = blockIdx.x * (size_t)kBlock + threadIdx.x; k < c; k += (size_t)gridDim.x * kBlock) {
    const int2 ij = pairs[k];
    const size_t i = static_cast<size_t>(ij.x) - first, j = static_cast<size_t>(ij.y) - first;  // wraps below `first`
    if (i < count) atomicAdd(&weights[i], 1.0);
    if (j < count) atomicAdd(&weights[j], 1.0);
  }
}
}  // namespace mhip

extern "C" {

int mhip_compose_keys_u64(size_t n, const uint32_t* major, const double* minor, int shift, uint64_t* out,
                          mhip_stream_t stream) {
  MHIP_REQUIRE(n == 0 || (major && minor && out), MHIP_ERR_INVALID_ARGUMENT, "null argument");
  MHIP_REQUIRE(shift >= 0 && shift <= 40, MHIP_ERR_INVALID_ARGUMENT, "shift %d is not in 0..40", shift);
  if (n == 0) return MHIP_SUCCESS;
  mhip::k_compose_keys<<<mhip::grid_for(n), mhip::kBlock, 0, mhip::as_stream(stream)>>>(
      n, major, minor, shift, reinterpret_cast<unsigned long long*>(out));
  MHIP_LAUNCH_CHECK();
  return MHIP_SUCCESS;
}

int mhip_fill_sequence(size_t n, double first, double* dst, mhip_stream_t stream) {
  MHIP_REQUIRE(n == 0 || dst, MHIP_ERR_INVALID_ARGUMENT, "dst is null");
  if (n == 0) return MHIP_SUCCESS;
  mhip::k_sequence<<<mhip::grid_for(n), mhip::kBlock, 0, mhip::as_stream(stream)>>>(n, first, dst);
  MHIP_LAUNCH_CHECK();
  return MHIP_SUCCESS;
}

int mhip_body_work_weights(size_t c, const int32_t* pairs, size_t first, size_t count, double* weights,
                           mhip_stream_t stream) {
  MHIP_REQUIRE(count == 0 || weights, MHIP_ERR_INVALID_ARGUMENT, "weights is null");
  MHIP_REQUIRE(c == 0 || pairs, MHIP_ERR_INVALID_ARGUMENT, "pairs is null");
  if (count == 0) return MHIP_SUCCESS;
  if (int e = mhip_fill(count, weights, 1.0, stream)) return e;
  if (c == 0) return MHIP_SUCCESS;
  mhip::k_pair_degree<<<mhip::grid_for(c), mhip::kBlock, 0, mhip::as_stream(stream)>>>(
      c, reinterpret_cast<const int2*>(pairs), first, count, weights);
  MHIP_LAUNCH_CHECK();
  return MHIP_SUCCESS;
}

}  // extern "C"
