// broadphase.hip -- neighbour-list construction (seam S3) on a uniform cell grid.
//
// Pipeline (all on the device; one host read of the pair total, where the reference's filter_view reads its scan
// total, GenNeighborLinkers.hpp:155-156):
//   k_bounds / k_grid_params   bin-point bounds, max reach -> grid with cell edge >= 2*reach (27-cell stencil suffices)
//   k_cell_count, scan, k_cell_scatter   counting sort of bodies into cells; 64-byte search records written in cell
//                              order so a wavefront walking a cell reads contiguous lines
//   k_pairs<COUNT>, scan, k_pairs<FILL>  per body: test the 27 neighbouring cells, count, then fill its CSR row,
//                              sort the row -> pairs sorted by (i, j) without a global sort
// The predicate is evaluated with the lower body index first, exactly as the CPU oracle does, so pair sets are
// bit-identical.  Integer/comparison work only: HBM/L2-bound, no MFMA.
#include <cstdlib>
#include <initializer_list>

#include "geom_device.hpp"

namespace mhip {

struct GridParams {
  double origin[3], ext[3];
  int nc[3];
  int ncell;
  double h;
};

struct SearchRec {  // 64 bytes
  double a[3];      // AABB: grown min corner        SPHERES: centre
  double b[3];      // AABB: grown max corner        SPHERES: (R + buffer, -, -)
  long long id;
  long long pad;
};

struct BpArgs {
  int kind, symmetric, periodic;
  double buffer;
  Periodic pm;
};

__device__ inline void body_volume(const BpArgs& A, size_t i, const double* __restrict__ aabb,
                                   const double* __restrict__ center, const double* __restrict__ brad, SearchRec& r,
                                   V3& binp, double& reach) {
  if (A.kind == MHIP_SEARCH_AABB) {
    const double* bx = aabb + 6 * i;
    reach = 0.0;
    double m[3];
    for (int k = 0; k < 3; ++k) {
      r.a[k] = bx[k] - A.buffer;
      r.b[k] = bx[3 + k] + A.buffer;
      m[k] = 0.5 * (r.a[k] + r.b[k]);
      reach = dmax(reach, 0.5 * (r.b[k] - r.a[k]));
    }
    binp = V3{m[0], m[1], m[2]};
  } else {
    const V3 c = load3(center, i);
    r.a[0] = c.x; r.a[1] = c.y; r.a[2] = c.z;
    r.b[0] = brad[i] + A.buffer;  // GenNeighborLinkers.hpp:582
    r.b[1] = 0.0; r.b[2] = 0.0;
    binp = c;
    reach = r.b[0];
  }
  if (A.periodic) binp = periodic_wrap(A.pm, binp);
  r.id = static_cast<long long>(i);
  r.pad = 0;
}

// partials[block][7] = min xyz, max xyz, reach
__global__ void __launch_bounds__(kBlock)
    k_bounds(size_t n, BpArgs A, const double* __restrict__ aabb, const double* __restrict__ center,
             const double* __restrict__ brad, double* __restrict__ partials) {
  __shared__ double scratch[kBlock / 64];
  double mn[3] = {1.7976931348623157e308, 1.7976931348623157e308, 1.7976931348623157e308};
  double mx[3] = {-1.7976931348623157e308, -1.7976931348623157e308, -1.7976931348623157e308};
  double reach = 0.0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    SearchRec r;
    V3 p;
    double rc;
    body_volume(A, i, aabb, center, brad, r, p, rc);
    mn[0] = dmin(mn[0], p.x); mn[1] = dmin(mn[1], p.y); mn[2] = dmin(mn[2], p.z);
    mx[0] = dmax(mx[0], p.x); mx[1] = dmax(mx[1], p.y); mx[2] = dmax(mx[2], p.z);
    reach = dmax(reach, rc);
  }
  for (int k = 0; k < 3; ++k) {
    const double a = -block_max(-mn[k], scratch);
    const double b = block_max(mx[k], scratch);
    if (threadIdx.x == 0) {
      partials[7 * blockIdx.x + k] = a;
      partials[7 * blockIdx.x + 3 + k] = b;
    }
  }
  const double rr = block_max(reach, scratch);
  if (threadIdx.x == 0) partials[7 * blockIdx.x + 6] = rr;
}

__global__ void __launch_bounds__(kBlock) k_grid_params(int nparts, const double* __restrict__ partials, BpArgs A, int cell_capacity,
                              GridParams* __restrict__ gp) {
  __shared__ double scratch[kBlock / 64];
  double mn[3] = {1.7976931348623157e308, 1.7976931348623157e308, 1.7976931348623157e308};
  double mx[3] = {-1.7976931348623157e308, -1.7976931348623157e308, -1.7976931348623157e308};
  double reach = 0.0;
  for (int i = threadIdx.x; i < nparts; i += blockDim.x) {
    for (int k = 0; k < 3; ++k) {
      mn[k] = dmin(mn[k], partials[7 * i + k]);
      mx[k] = dmax(mx[k], partials[7 * i + 3 + k]);
    }
    reach = dmax(reach, partials[7 * i + 6]);
  }
  for (int k = 0; k < 3; ++k) {
    mn[k] = -block_max(-mn[k], scratch);
    mx[k] = block_max(mx[k], scratch);
  }
  reach = block_max(reach, scratch);
  if (threadIdx.x != 0) return;
  double h = 2.0 * reach * (1.0 + 1e-12);
  if (!(h > 0.0)) h = 1.0;
  for (int k = 0; k < 3; ++k) {
    if (A.periodic) {
      gp->origin[k] = 0.0;
      gp->ext[k] = comp(A.pm.scale, k);
    } else {
      gp->origin[k] = mn[k];
      gp->ext[k] = mx[k] - mn[k];
    }
  }
  gp->nc[0] = gp->nc[1] = gp->nc[2] = 1;
  gp->ncell = 1;
  for (int it = 0; it < 200; ++it) {
    long long prod = 1;
    for (int k = 0; k < 3; ++k) {
      double q = floor(gp->ext[k] / h);
      if (!(q >= 1.0)) q = 1.0;
      if (q > 1024.0) q = 1024.0;
      gp->nc[k] = static_cast<int>(q);
      prod *= gp->nc[k];
    }
    if (prod <= cell_capacity) {
      gp->ncell = static_cast<int>(prod);
      break;
    }
    h *= 1.25;
  }
  gp->h = h;
}

__device__ inline int cell_coord(const GridParams& gp, double v, int k) {
  if (!(gp.ext[k] > 0.0)) return 0;
  int ci = static_cast<int>(floor((v - gp.origin[k]) / gp.ext[k] * gp.nc[k]));
  if (ci < 0) ci = 0;
  if (ci > gp.nc[k] - 1) ci = gp.nc[k] - 1;
  return ci;
}

__global__ void __launch_bounds__(kBlock)
    k_cell_count(size_t n, BpArgs A, const double* __restrict__ aabb, const double* __restrict__ center,
                 const double* __restrict__ brad, const GridParams* __restrict__ gpp, int32_t* __restrict__ cell_of,
                 int32_t* __restrict__ cell_count) {
  const GridParams gp = *gpp;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    SearchRec r;
    V3 p;
    double rc;
    body_volume(A, i, aabb, center, brad, r, p, rc);
    const int cid = (cell_coord(gp, p.z, 2) * gp.nc[1] + cell_coord(gp, p.y, 1)) * gp.nc[0] + cell_coord(gp, p.x, 0);
    cell_of[i] = cid;
    atomicAdd(&cell_count[cid], 1);
  }
}

__global__ void __launch_bounds__(kBlock)
    k_cell_scatter(size_t n, BpArgs A, const double* __restrict__ aabb, const double* __restrict__ center,
                   const double* __restrict__ brad, const int32_t* __restrict__ cell_of,
                   int32_t* __restrict__ cursor, SearchRec* __restrict__ recs, int32_t* __restrict__ slot_cell) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    SearchRec r;
    V3 p;
    double rc;
    body_volume(A, i, aabb, center, brad, r, p, rc);
    const int cid = cell_of[i];
    const int slot = atomicAdd(&cursor[cid], 1);
    recs[slot] = r;
    slot_cell[slot] = cid;
  }
}

// closed overlap test, a = lower index, b = higher index
__device__ inline bool volumes_overlap(const BpArgs& A, const SearchRec& a, const SearchRec& b) {
  if (A.kind == MHIP_SEARCH_SPHERES) {
    const V3 ca{a.a[0], a.a[1], a.a[2]}, cb{b.a[0], b.a[1], b.a[2]};
    const V3 s = A.periodic ? periodic_sep(A.pm, ca, cb) : (cb - ca);
    const double d2 = dot(s, s);
    const double rs = a.b[0] + b.b[0];
    return d2 <= rs * rs;
  }
  if (!A.periodic) {  // geom::intersects (AABB.hpp:420-431)
    if (a.b[0] < b.a[0] || a.b[1] < b.a[1] || a.b[2] < b.a[2]) return false;
    return !(b.b[0] < a.a[0] || b.b[1] < a.a[1] || b.b[2] < a.a[2]);
  }
  const V3 mi{0.5 * (a.a[0] + a.b[0]), 0.5 * (a.a[1] + a.b[1]), 0.5 * (a.a[2] + a.b[2])};
  const V3 mj{0.5 * (b.a[0] + b.b[0]), 0.5 * (b.a[1] + b.b[1]), 0.5 * (b.a[2] + b.b[2])};
  const V3 s = periodic_sep(A.pm, mi, mj);
  for (int k = 0; k < 3; ++k) {
    const double shift = (comp(mi, k) + comp(s, k)) - comp(mj, k);
    const double blo = b.a[k] + shift, bhi = b.b[k] + shift;
    if (a.b[k] < blo || bhi < a.a[k]) return false;
  }
  return true;
}

// distinct neighbour cell indices along one axis
__device__ inline int axis_neighbours(int c, int nc, bool periodic, int out[3]) {
  int m = 0;
  for (int d = -1; d <= 1; ++d) {
    int v = c + d;
    if (periodic) {
      v = (v + nc) % nc;
    } else if (v < 0 || v >= nc) {
      continue;
    }
    bool dup = false;
    for (int q = 0; q < m; ++q) dup |= (out[q] == v);
    if (!dup) out[m++] = v;
  }
  return m;
}

template <bool FILL>
__global__ void __launch_bounds__(kBlock)
    k_pairs(size_t n, BpArgs A, const GridParams* __restrict__ gpp, const SearchRec* __restrict__ recs,
            const int32_t* __restrict__ slot_cell, const int32_t* __restrict__ cell_ptr,
            int32_t* __restrict__ counts, const int32_t* __restrict__ row_ptr, int32_t* __restrict__ col,
            int2* __restrict__ pairs) {
  const GridParams gp = *gpp;
  const size_t s = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (s >= n) return;
  const SearchRec me = recs[s];
  const int i = static_cast<int>(me.id);
  const int cid = slot_cell[s];
  const int cx = cid % gp.nc[0], cy = (cid / gp.nc[0]) % gp.nc[1], cz = cid / (gp.nc[0] * gp.nc[1]);
  int xs[3], ys[3], zs[3];
  const int nx = axis_neighbours(cx, gp.nc[0], A.periodic, xs);
  const int ny = axis_neighbours(cy, gp.nc[1], A.periodic, ys);
  const int nz = axis_neighbours(cz, gp.nc[2], A.periodic, zs);
  int cnt = 0;
  const int32_t base = FILL ? row_ptr[i] : 0;
  for (int iz = 0; iz < nz; ++iz)
    for (int iy = 0; iy < ny; ++iy)
      for (int ix = 0; ix < nx; ++ix) {
        const int ncid = (zs[iz] * gp.nc[1] + ys[iy]) * gp.nc[0] + xs[ix];
        const int32_t beg = cell_ptr[ncid], end = cell_ptr[ncid + 1];
        for (int32_t t = beg; t < end; ++t) {
          const SearchRec o = recs[t];
          const int j = static_cast<int>(o.id);
          if (j == i || (!A.symmetric && j < i)) continue;
          const bool hit = (i < j) ? volumes_overlap(A, me, o) : volumes_overlap(A, o, me);
          if (hit) {
            if (FILL) col[base + cnt] = j;
            ++cnt;
          }
        }
      }
  if (!FILL) {
    counts[i] = cnt;
    return;
  }
  // sort the row ascending (rows are short), then emit (i, j)
  for (int a = 1; a < cnt; ++a) {
    const int32_t v = col[base + a];
    int b = a - 1;
    while (b >= 0 && col[base + b] > v) {
      col[base + b + 1] = col[base + b];
      --b;
    }
    col[base + b + 1] = v;
  }
  for (int a = 0; a < cnt; ++a) pairs[base + a] = make_int2(i, col[base + a]);
}

// The same search with the candidates staged through LDS (free boundaries; the periodic search keeps k_pairs, whose
// wrapped stencil is not a linear cell range).  A workgroup owns 256 consecutive slots of the cell-ordered records, i.e.
// the bodies of cells c_first..c_last.  Cells are numbered x-fastest, so for each of the 9 (dz, dy) rows of the stencil
// the candidates of ALL its bodies lie in the linear cell range [c_first + off - 1, c_last + off + 1], off = (dz ny +
// dy) nx -- one contiguous run of 64-byte records (at a row end the run also holds a few cells that are nobody's
// neighbours; they are staged but never tested).  The run is copied to LDS in tiles of kPairTile records with
// coalesced 16-byte loads, and every lane then walks only its own three x-cells inside the tile.  Global memory
// sees each candidate record once per workgroup instead of once per lane; the tests, their operand order and the
// row sort are those of k_pairs, so the lists are identical.
constexpr int kPairTile = 512;  // records per tile: 32 KB of LDS, 4 workgroups per CU
template <bool FILL>
__global__ void __launch_bounds__(kBlock)
    k_pairs_lds(size_t n, BpArgs A, const GridParams* __restrict__ gpp, const SearchRec* __restrict__ recs,
                const int32_t* __restrict__ slot_cell, const int32_t* __restrict__ cell_ptr,
                int32_t* __restrict__ counts, const int32_t* __restrict__ row_ptr, int32_t* __restrict__ col,
                int2* __restrict__ pairs) {
  __shared__ __attribute__((aligned(16))) SearchRec tile[kPairTile];
  const GridParams gp = *gpp;
  const size_t s0 = blockIdx.x * (size_t)kBlock;
  const size_t s = s0 + threadIdx.x;
  const bool live = s < n;
  SearchRec me{};
  int i = -1, cx = 0, cy = 0, cz = 0;
  if (live) {
    me = recs[s];
    i = static_cast<int>(me.id);
    const int cid = slot_cell[s];
    cx = cid % gp.nc[0];
    cy = (cid / gp.nc[0]) % gp.nc[1];
    cz = cid / (gp.nc[0] * gp.nc[1]);
  }
  const int c_first = slot_cell[s0];
  const int c_last = slot_cell[(s0 + kBlock <= n ? s0 + kBlock : n) - 1];
  int cnt = 0;
  const int32_t base = (FILL && live) ? row_ptr[i] : 0;
  for (int dz = -1; dz <= 1; ++dz)
    for (int dy = -1; dy <= 1; ++dy) {
      const long long off = (static_cast<long long>(dz) * gp.nc[1] + dy) * gp.nc[0];
      long long lo = c_first + off - 1, hi = c_last + off + 1;
      if (lo < 0) lo = 0;
      if (hi > gp.ncell - 1) hi = gp.ncell - 1;
      if (hi < lo) continue;  // the same for every thread of the workgroup
      const int32_t run_beg = cell_ptr[lo], run_end = cell_ptr[hi + 1];
      int32_t my_beg = 0, my_end = 0;  // this lane's candidates in the row: its cells x-1 .. x+1
      const int z = cz + dz, y = cy + dy;
      if (live && z >= 0 && z < gp.nc[2] && y >= 0 && y < gp.nc[1]) {
        const int x0 = cx > 0 ? cx - 1 : 0, x1 = cx + 1 < gp.nc[0] ? cx + 1 : gp.nc[0] - 1;
        const int row = (z * gp.nc[1] + y) * gp.nc[0];
        my_beg = cell_ptr[row + x0];
        my_end = cell_ptr[row + x1 + 1];
      }
      for (int32_t t0 = run_beg; t0 < run_end; t0 += kPairTile) {
        const int32_t t1 = t0 + kPairTile < run_end ? t0 + kPairTile : run_end;
        __syncthreads();  // everyone is done with the previous tile
        {
          const double2* __restrict__ src = reinterpret_cast<const double2*>(recs + t0);
          double2* dst = reinterpret_cast<double2*>(tile);
          const int nq = (t1 - t0) * 4;
          for (int q = threadIdx.x; q < nq; q += kBlock) dst[q] = src[q];
        }
        __syncthreads();
        const int32_t b = my_beg > t0 ? my_beg : t0, e = my_end < t1 ? my_end : t1;
        for (int32_t t = b; t < e; ++t) {
          const SearchRec& o = tile[t - t0];
          const int j = static_cast<int>(o.id);
          if (j == i || (!A.symmetric && j < i)) continue;
          const bool hit = (i < j) ? volumes_overlap(A, me, o) : volumes_overlap(A, o, me);
          if (hit) {
            if (FILL) col[base + cnt] = j;
            ++cnt;
          }
        }
      }
    }
  if (!live) return;
  if (!FILL) {
    counts[i] = cnt;
    return;
  }
  for (int a = 1; a < cnt; ++a) {  // sort the row ascending (rows are short), then emit (i, j)
    const int32_t v = col[base + a];
    int b = a - 1;
    while (b >= 0 && col[base + b] > v) {
      col[base + b + 1] = col[base + b];
      --b;
    }
    col[base + b + 1] = v;
  }
  for (int a = 0; a < cnt; ++a) pairs[base + a] = make_int2(i, col[base + a]);
}

// GenNeighborLinkers.hpp:603-615: moved iff sqrt(dx^2+dy^2+dz^2) > 0.5 * buffer (plain left-to-right sum there)
__global__ void __launch_bounds__(kBlock) k_moved(size_t n, const double* __restrict__ c_new,
                                                 const double* __restrict__ c_old, double buffer,
                                                 int* __restrict__ flag) {
  int moved = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const double dx = c_new[3 * i] - c_old[3 * i], dy = c_new[3 * i + 1] - c_old[3 * i + 1],
                 dz = c_new[3 * i + 2] - c_old[3 * i + 2];
    const double disp = sqrt(dx * dx + dy * dy + dz * dz);
    moved |= (disp > 0.5 * buffer) ? 1 : 0;
  }
  moved = wave_or(moved);
  if ((threadIdx.x & 63) == 0 && moved) atomicOr(flag, 1);
}

}  // namespace mhip

using namespace mhip;

struct mhip_broadphase {
  mhip_broadphase_config cfg{};
  bool built = false;
  size_t n = 0, num_pairs = 0;
  DeviceBuffer recs, cell_of, slot_cell, cell_cnt, cell_ptr, cursor, counts, row_ptr, col, pairs, old_center, params,
      partials, scanws, flag;
  int* host_scalar = nullptr;  // pinned
};

extern "C" {

int mhip_broadphase_create(mhip_broadphase_t* handle) {
  MHIP_REQUIRE(handle != nullptr, MHIP_ERR_INVALID_ARGUMENT, "handle is null");
  *handle = new mhip_broadphase();  // no HIP call here: device/pinned memory is acquired by the first build
  return MHIP_SUCCESS;
}

static int ensure_host_scalar(mhip_broadphase* h) {
  if (!h->host_scalar) MHIP_HIP(hipHostMalloc(reinterpret_cast<void**>(&h->host_scalar), 64));
  return MHIP_SUCCESS;
}

int mhip_broadphase_destroy(mhip_broadphase_t h) {
  if (!h) return MHIP_SUCCESS;
  for (DeviceBuffer* b : {&h->recs, &h->cell_of, &h->slot_cell, &h->cell_cnt, &h->cell_ptr, &h->cursor, &h->counts,
                          &h->row_ptr, &h->col, &h->pairs, &h->old_center, &h->params, &h->partials, &h->scanws,
                          &h->flag})
    b->release();
  if (h->host_scalar) (void)hipHostFree(h->host_scalar);
  delete h;
  return MHIP_SUCCESS;
}

int mhip_broadphase_build(mhip_broadphase_t h, const mhip_broadphase_config* config, size_t n, const double* aabb,
                          const double* center, const double* bounding_radius, size_t* num_pairs,
                          mhip_stream_t stream) {
  TraceRange trace_range("GenNeighborLinks::generate");
  MHIP_REQUIRE(h != nullptr, MHIP_ERR_INVALID_ARGUMENT, "broadphase handle is null");
  MHIP_REQUIRE(config != nullptr && num_pairs != nullptr, MHIP_ERR_INVALID_ARGUMENT, "config / num_pairs is null");
  MHIP_REQUIRE(config->search_kind == MHIP_SEARCH_SPHERES || config->search_kind == MHIP_SEARCH_AABB,
               MHIP_ERR_INVALID_ARGUMENT, "unknown search kind %d", config->search_kind);
  MHIP_REQUIRE(config->buffer >= 0.0, MHIP_ERR_INVALID_ARGUMENT, "search buffer must be >= 0");
  MHIP_REQUIRE(n < (1u << 31), MHIP_ERR_RUNTIME, "too many bodies for 32-bit indices");
  if (config->search_kind == MHIP_SEARCH_AABB)
    MHIP_REQUIRE(aabb != nullptr || n == 0, MHIP_ERR_INVALID_ARGUMENT, "aabb is required for MHIP_SEARCH_AABB");
  else
    MHIP_REQUIRE(bounding_radius != nullptr || n == 0, MHIP_ERR_INVALID_ARGUMENT,
                 "bounding_radius is required for MHIP_SEARCH_SPHERES");
  MHIP_REQUIRE(center != nullptr || n == 0, MHIP_ERR_INVALID_ARGUMENT, "center is null");
  if (config->periodic)
    MHIP_REQUIRE(config->box[0] > 0 && config->box[1] > 0 && config->box[2] > 0, MHIP_ERR_INVALID_ARGUMENT,
                 "periodic box must be positive");
  hipStream_t s = as_stream(stream);
  if (int e = ensure_host_scalar(h)) return e;
  h->cfg = *config;
  h->n = n;
  h->num_pairs = 0;
  h->built = true;
  *num_pairs = 0;
  if (int e = h->row_ptr.reserve((n + 2) * sizeof(int32_t))) return e;
  if (n == 0) {
    MHIP_HIP(hipMemsetAsync(h->row_ptr.ptr, 0, 2 * sizeof(int32_t), s));
    return MHIP_SUCCESS;
  }
  const int cell_capacity = static_cast<int>(n < 4096 ? 4096 : n);
  BpArgs A;
  A.kind = config->search_kind;
  A.symmetric = config->symmetric ? 1 : 0;
  A.periodic = config->periodic ? 1 : 0;
  A.buffer = config->buffer;
  const double one[3] = {1, 1, 1};
  A.pm = make_periodic(config->periodic ? config->box : one);

  if (int e = h->recs.reserve(n * sizeof(SearchRec))) return e;
  if (int e = h->cell_of.reserve(n * sizeof(int32_t))) return e;
  if (int e = h->slot_cell.reserve(n * sizeof(int32_t))) return e;
  if (int e = h->cell_cnt.reserve((cell_capacity + 2) * sizeof(int32_t))) return e;
  if (int e = h->cell_ptr.reserve((cell_capacity + 2) * sizeof(int32_t))) return e;
  if (int e = h->cursor.reserve((cell_capacity + 2) * sizeof(int32_t))) return e;
  if (int e = h->counts.reserve((n + 2) * sizeof(int32_t))) return e;
  if (int e = h->old_center.reserve(3 * n * sizeof(double))) return e;
  if (int e = h->params.reserve(sizeof(GridParams) + 64)) return e;
  if (int e = h->partials.reserve((7 * kMaxGrid + 8) * sizeof(double))) return e;
  {
    const size_t m = (size_t)cell_capacity > n ? (size_t)cell_capacity : n;
    if (int e = h->scanws.reserve(scan_workspace_bytes(m + 2) + 64)) return e;
  }
  if (int e = h->flag.reserve(64)) return e;

  GridParams* gp = h->params.as<GridParams>();
  const unsigned g = grid_for(n);
  k_bounds<<<g, kBlock, 0, s>>>(n, A, aabb, center, bounding_radius, h->partials.as<double>());
  MHIP_LAUNCH_CHECK();
  k_grid_params<<<1, kBlock, 0, s>>>((int)g, h->partials.as<double>(), A, cell_capacity, gp);
  MHIP_LAUNCH_CHECK();
  MHIP_HIP(hipMemsetAsync(h->cell_cnt.ptr, 0, (cell_capacity + 1) * sizeof(int32_t), s));
  k_cell_count<<<g, kBlock, 0, s>>>(n, A, aabb, center, bounding_radius, gp, h->cell_of.as<int32_t>(),
                                   h->cell_cnt.as<int32_t>());
  MHIP_LAUNCH_CHECK();
  // scanning all cell_capacity slots (unused cells hold 0) keeps the grid size off the host
  if (int e = exclusive_scan_i32(h->cell_cnt.as<int32_t>(), h->cell_ptr.as<int32_t>(), cell_capacity, h->scanws.ptr, s))
    return e;
  MHIP_HIP(hipMemcpyAsync(h->cursor.ptr, h->cell_ptr.ptr, (cell_capacity + 1) * sizeof(int32_t),
                          hipMemcpyDeviceToDevice, s));
  k_cell_scatter<<<g, kBlock, 0, s>>>(n, A, aabb, center, bounding_radius, h->cell_of.as<int32_t>(),
                                     h->cursor.as<int32_t>(), h->recs.as<SearchRec>(), h->slot_cell.as<int32_t>());
  MHIP_LAUNCH_CHECK();
  const unsigned gb = grid_exact(n);
  static const bool use_lds = [] {
    const char* e = getenv("MHIP_PAIRS_LDS");  // A/B switch of the LDS-staged search (default on)
    return !(e && atoi(e) == 0);
  }();
  const bool lds = use_lds && !A.periodic;
  if (lds)
    k_pairs_lds<false><<<gb, kBlock, 0, s>>>(n, A, gp, h->recs.as<SearchRec>(), h->slot_cell.as<int32_t>(),
                                            h->cell_ptr.as<int32_t>(), h->counts.as<int32_t>(), nullptr, nullptr,
                                            nullptr);
  else
    k_pairs<false><<<gb, kBlock, 0, s>>>(n, A, gp, h->recs.as<SearchRec>(), h->slot_cell.as<int32_t>(),
                                        h->cell_ptr.as<int32_t>(), h->counts.as<int32_t>(), nullptr, nullptr, nullptr);
  MHIP_LAUNCH_CHECK();
  if (int e = exclusive_scan_i32(h->counts.as<int32_t>(), h->row_ptr.as<int32_t>(), n, h->scanws.ptr, s)) return e;
  MHIP_HIP(hipMemcpyAsync(h->host_scalar, h->row_ptr.as<int32_t>() + n, sizeof(int32_t), hipMemcpyDeviceToHost, s));
  MHIP_HIP(hipStreamSynchronize(s));
  const int32_t total = h->host_scalar[0];
  MHIP_REQUIRE(total >= 0, MHIP_ERR_RUNTIME, "pair count overflowed 32 bits");
  h->num_pairs = static_cast<size_t>(total);
  if (int e = h->col.reserve((h->num_pairs + 2) * sizeof(int32_t))) return e;
  if (int e = h->pairs.reserve((h->num_pairs + 2) * sizeof(int2))) return e;
  if (lds)
    k_pairs_lds<true><<<gb, kBlock, 0, s>>>(n, A, gp, h->recs.as<SearchRec>(), h->slot_cell.as<int32_t>(),
                                           h->cell_ptr.as<int32_t>(), nullptr, h->row_ptr.as<int32_t>(),
                                           h->col.as<int32_t>(), h->pairs.as<int2>());
  else
    k_pairs<true><<<gb, kBlock, 0, s>>>(n, A, gp, h->recs.as<SearchRec>(), h->slot_cell.as<int32_t>(),
                                       h->cell_ptr.as<int32_t>(), nullptr, h->row_ptr.as<int32_t>(),
                                       h->col.as<int32_t>(), h->pairs.as<int2>());
  MHIP_LAUNCH_CHECK();
  MHIP_HIP(hipMemcpyAsync(h->old_center.ptr, center, 3 * n * sizeof(double), hipMemcpyDeviceToDevice, s));
  *num_pairs = h->num_pairs;
  return MHIP_SUCCESS;
}

int mhip_broadphase_get_pairs(mhip_broadphase_t h, int32_t* pairs, int32_t* row_ptr, int32_t* col,
                              mhip_stream_t stream) {
  MHIP_REQUIRE(h != nullptr, MHIP_ERR_INVALID_ARGUMENT, "broadphase handle is null");
  MHIP_REQUIRE(h->built, MHIP_ERR_RUNTIME, "mhip_broadphase_build must be called before get_pairs");
  hipStream_t s = as_stream(stream);
  if (pairs && h->num_pairs)
    MHIP_HIP(hipMemcpyAsync(pairs, h->pairs.ptr, h->num_pairs * sizeof(int2), hipMemcpyDeviceToDevice, s));
  if (row_ptr)
    MHIP_HIP(hipMemcpyAsync(row_ptr, h->row_ptr.ptr, (h->n + 1) * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
  if (col && h->num_pairs)
    MHIP_HIP(hipMemcpyAsync(col, h->col.ptr, h->num_pairs * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
  return MHIP_SUCCESS;
}

int mhip_broadphase_needs_rebuild(mhip_broadphase_t h, size_t n, const double* center, int* flag,
                                  mhip_stream_t stream) {
  MHIP_REQUIRE(h != nullptr && flag != nullptr, MHIP_ERR_INVALID_ARGUMENT, "null argument");
  if (!h->built || n != h->n) {  // first call / body set changed: rebuild (GenNeighborLinkers.hpp:513-533)
    *flag = 1;
    return MHIP_SUCCESS;
  }
  *flag = 0;
  if (n == 0) return MHIP_SUCCESS;
  MHIP_REQUIRE(center != nullptr, MHIP_ERR_INVALID_ARGUMENT, "center is null");
  hipStream_t s = as_stream(stream);
  if (int e = ensure_host_scalar(h)) return e;
  int* dflag = h->flag.as<int>();
  MHIP_HIP(hipMemsetAsync(dflag, 0, sizeof(int), s));
  k_moved<<<grid_for(n), kBlock, 0, s>>>(n, center, h->old_center.as<double>(), h->cfg.buffer, dflag);
  MHIP_LAUNCH_CHECK();
  MHIP_HIP(hipMemcpyAsync(h->host_scalar, dflag, sizeof(int), hipMemcpyDeviceToHost, s));
  MHIP_HIP(hipStreamSynchronize(s));
  *flag = h->host_scalar[0] ? 1 : 0;
  return MHIP_SUCCESS;
}

}  // extern "C"
