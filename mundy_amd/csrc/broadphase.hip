// broadphase.hip -- neighbour-list construction (seam S3): a uniform cell grid for bodies of similar size and a linear
// BVH over 63-bit Morton keys (the reference's search method, stk::search::MORTON_LBVH, GenNeighborLinkers.hpp:318,
// :658) for size-disperse systems, where one large body would set the cell edge for everybody.
//
// Grid pipeline (all on the device; one host read of the pair total, where the reference's filter_view reads its scan
// total, GenNeighborLinkers.hpp:155-156):
//   k_bounds / k_grid_params   bin-point bounds, max and mean reach -> grid with cell edge >= 2*reach (27-cell stencil)
//   k_cell_count, scan, k_cell_scatter   counting sort of bodies into cells; 64-byte search records written in cell
//                              order so a wavefront walking a cell reads contiguous lines
//   k_pairs<COUNT>, scan, k_pairs<FILL>  per body: test the 27 neighbouring cells, count, then fill its CSR row
// LBVH pipeline:
//   k_morton_keys, radix_sort_u64 (sort.hip)   63-bit keys of the bin points, records gathered in key order
//   k_lbvh_build    one thread per internal node: Karras' binary radix tree from the longest common prefixes
//   k_lbvh_refit_pass (x2), _top   bottom-up box union, second arriver at a node proceeds (workgroup-scope tickets)
//   k_lbvh_ropes    skip pointers -> stackless pre-order traversal
//   k_lbvh_pairs<COUNT>, scan, k_lbvh_pairs<FILL>   per body: traverse, exact predicate at the leaves (rows of up to 32
//                   partners are parked in a slab by the first pass: one traversal)
// Both: rows of <= 32 partners are sorted and emitted by their thread; longer rows (a large body among small ones) go
// through the workgroup radix sort of sort.hip -> pairs sorted by (i, j) without a global sort.
// The predicate is evaluated with the lower body index first, exactly as the CPU oracle does, so pair sets are
// bit-identical whichever structure found them.  Integer/comparison work only: HBM/L2-bound, no MFMA.
#include <algorithm>
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
  // Triclinic cell (PeriodicMetric, periodicity.hpp:233-332).  The STRUCTURES (cell grid, Morton tree) then work in
  // scaled fractional coordinates g_a = w_a (h^-1 x)_a, w_a = 1 / |row a of h^-1| = the width of the cell along its
  // a-th reciprocal axis: there the cell is the orthorhombic box [0, w) (pm above holds w) with lattice translations
  // along the axes, a ball of radius R spans exactly -/+ R on every axis, and a Cartesian box of half extents e spans
  // w_a sum_k |h^-1(a, k)| e_k.  The search records hold those g-space bounding boxes (they only bin and prune); the
  // PREDICATE is evaluated on the Cartesian volumes themselves, fetched by id, at the image PeriodicMetric::sep picks
  // (fractional minimum image) -- the same statement the orthorhombic search makes with PeriodicScaledMetric::sep.
  int triclinic;
  Triclinic tm;
  double gw[3];
  const double *aabb_c, *center_c, *brad_c;  // the Cartesian volumes, for the predicate
  // seam S3 result shaping (GenNeighborLinkers.hpp: acts_on(source, target) :486-507, search_filters :185-245):
  int include_self;                  // 0 = ExcludeSelfInteractions (the default)
  const unsigned char* is_source;    // [n] or null (every body is a source)
  const unsigned char* is_target;    // [n] or null
  const int32_t* ex_ptr;             // CSR of partners a source must not be paired with (ExcludeConnectedEntities;
  const int32_t* ex_idx;             //   already-linked neighbours when duplicate links are not allowed), or null
};

// does the ordered pair (source i, target j), i != j or include_self, pass the sets and the filters?
__device__ inline bool pair_allowed(const BpArgs& A, int i, int j) {
  if (i == j && !A.include_self) return false;
  if (!A.symmetric && j < i) return false;
  if (A.is_target && !A.is_target[j]) return false;
  if (A.ex_ptr) {
    for (int32_t k = A.ex_ptr[i], e = A.ex_ptr[i + 1]; k < e; ++k)
      if (A.ex_idx[k] == j) return false;
  }
  return true;
}
__device__ inline bool is_source(const BpArgs& A, int i) { return !A.is_source || A.is_source[i]; }

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
  if (A.triclinic) {
    // into the coordinates of the structures: midpoint / centre -> g, half extents -> their g-space bounds (rounded up
    // a little: these boxes only bin and prune)
    const V3 f = matvec3(A.tm.hi, binp);
    const V3 g{A.gw[0] * f.x, A.gw[1] * f.y, A.gw[2] * f.z};
    if (A.kind == MHIP_SEARCH_AABB) {
      const double e[3] = {0.5 * (r.b[0] - r.a[0]), 0.5 * (r.b[1] - r.a[1]), 0.5 * (r.b[2] - r.a[2])};
      reach = 0.0;
      for (int a = 0; a < 3; ++a) {
        double ge = A.gw[a] * (fabs(A.tm.hi[3 * a]) * e[0] + fabs(A.tm.hi[3 * a + 1]) * e[1] + fabs(A.tm.hi[3 * a + 2]) * e[2]);
        ge = ge * (1.0 + 1e-12) + 1e-300;
        r.a[a] = comp(g, a) - ge;
        r.b[a] = comp(g, a) + ge;
        reach = dmax(reach, ge);
      }
    } else {
      r.a[0] = g.x; r.a[1] = g.y; r.a[2] = g.z;  // r.b[0] = R + buffer: a ball spans -/+ R on every g axis
      reach = r.b[0] * (1.0 + 1e-12);
      r.b[0] = reach;
    }
    binp = g;
  }
  if (A.periodic) binp = periodic_wrap(A.pm, binp);
  r.id = static_cast<long long>(i);
  r.pad = 0;
}

// partials[block][8] = min xyz, max xyz, max reach, sum of reach
__global__ void __launch_bounds__(kBlock)
    k_bounds(size_t n, BpArgs A, const double* __restrict__ aabb, const double* __restrict__ center,
             const double* __restrict__ brad, double* __restrict__ partials) {
  __shared__ double scratch[kBlock / 64];
  double mn[3] = {1.7976931348623157e308, 1.7976931348623157e308, 1.7976931348623157e308};
  double mx[3] = {-1.7976931348623157e308, -1.7976931348623157e308, -1.7976931348623157e308};
  double reach = 0.0, rsum = 0.0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    SearchRec r;
    V3 p;
    double rc;
    body_volume(A, i, aabb, center, brad, r, p, rc);
    mn[0] = dmin(mn[0], p.x); mn[1] = dmin(mn[1], p.y); mn[2] = dmin(mn[2], p.z);
    mx[0] = dmax(mx[0], p.x); mx[1] = dmax(mx[1], p.y); mx[2] = dmax(mx[2], p.z);
    reach = dmax(reach, rc);
    rsum += rc;
  }
  for (int k = 0; k < 3; ++k) {
    const double a = -block_max(-mn[k], scratch);
    const double b = block_max(mx[k], scratch);
    if (threadIdx.x == 0) {
      partials[8 * blockIdx.x + k] = a;
      partials[8 * blockIdx.x + 3 + k] = b;
    }
  }
  const double rr = block_max(reach, scratch);
  const double rs = block_sum(rsum, scratch);  // only steers the choice of structure: its rounding is immaterial
  if (threadIdx.x == 0) {
    partials[8 * blockIdx.x + 6] = rr;
    partials[8 * blockIdx.x + 7] = rs;
  }
}

__global__ void __launch_bounds__(kBlock) k_grid_params(int nparts, const double* __restrict__ partials, BpArgs A, int cell_capacity,
                              GridParams* __restrict__ gp, double* __restrict__ summary /* 8 doubles */) {
  __shared__ double scratch[kBlock / 64];
  double mn[3] = {1.7976931348623157e308, 1.7976931348623157e308, 1.7976931348623157e308};
  double mx[3] = {-1.7976931348623157e308, -1.7976931348623157e308, -1.7976931348623157e308};
  double reach = 0.0, rsum = 0.0;
  for (int i = threadIdx.x; i < nparts; i += blockDim.x) {
    for (int k = 0; k < 3; ++k) {
      mn[k] = dmin(mn[k], partials[8 * i + k]);
      mx[k] = dmax(mx[k], partials[8 * i + 3 + k]);
    }
    reach = dmax(reach, partials[8 * i + 6]);
    rsum += partials[8 * i + 7];
  }
  for (int k = 0; k < 3; ++k) {
    mn[k] = -block_max(-mn[k], scratch);
    mx[k] = block_max(mx[k], scratch);
  }
  reach = block_max(reach, scratch);
  rsum = block_sum(rsum, scratch);
  if (threadIdx.x != 0) return;
  for (int k = 0; k < 3; ++k) {  // scene box of the bin points, max reach, sum of reach: read by the host and the LBVH
    summary[k] = mn[k];
    summary[3 + k] = mx[k];
  }
  summary[6] = reach;
  summary[7] = rsum;
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
  if (A.triclinic) {  // the records hold g-space boxes: the Cartesian volumes by id, PeriodicMetric::sep (:304-307)
    const size_t ia = static_cast<size_t>(a.id), ib = static_cast<size_t>(b.id);
    if (A.kind == MHIP_SEARCH_SPHERES) {
      const V3 s = periodic_sep(A.tm, load3(A.center_c, ia), load3(A.center_c, ib));
      const double d2 = dot(s, s);
      const double rs = (A.brad_c[ia] + A.buffer) + (A.brad_c[ib] + A.buffer);
      return d2 <= rs * rs;
    }
    double alo[3], ahi[3], blo[3], bhi[3];
    for (int k = 0; k < 3; ++k) {
      alo[k] = A.aabb_c[6 * ia + k] - A.buffer;
      ahi[k] = A.aabb_c[6 * ia + 3 + k] + A.buffer;
      blo[k] = A.aabb_c[6 * ib + k] - A.buffer;
      bhi[k] = A.aabb_c[6 * ib + 3 + k] + A.buffer;
    }
    const V3 mi{0.5 * (alo[0] + ahi[0]), 0.5 * (alo[1] + ahi[1]), 0.5 * (alo[2] + ahi[2])};
    const V3 mj{0.5 * (blo[0] + bhi[0]), 0.5 * (blo[1] + bhi[1]), 0.5 * (blo[2] + bhi[2])};
    const V3 s = periodic_sep(A.tm, mi, mj);
    for (int k = 0; k < 3; ++k) {
      const double shift = (comp(mi, k) + comp(s, k)) - comp(mj, k);
      const double lo = blo[k] + shift, hi = bhi[k] + shift;
      if (ahi[k] < lo || hi < alo[k]) return false;
    }
    return true;
  }
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

// Row of body i: cnt unsorted partners at col[base ..).  Rows of up to kShortSegment partners are sorted (insertion
// sort, rows are short) and emitted as (i, j) pairs by their own thread; longer ones are queued for the workgroup sort.
struct RowSink {
  int32_t* col;
  int2* pairs;
  int32_t* long_count;
  int32_t* long_list;
};
__device__ inline void finish_row(const RowSink& out, int i, int32_t base, int cnt) {
  if (cnt > kShortSegment) {
    out.long_list[atomicAdd(out.long_count, 1)] = i;
    return;
  }
  int32_t* col = out.col;
  for (int a = 1; a < cnt; ++a) {
    const int32_t v = col[base + a];
    int b = a - 1;
    while (b >= 0 && col[base + b] > v) {
      col[base + b + 1] = col[base + b];
      --b;
    }
    col[base + b + 1] = v;
  }
  for (int a = 0; a < cnt; ++a) out.pairs[base + a] = make_int2(i, col[base + a]);
}
// emits the pairs of the queued rows once sort_listed_segments_u32 has sorted them: one workgroup per row
__global__ void __launch_bounds__(kBlock) k_emit_long_rows(const int32_t* __restrict__ row_ptr,
                                                          const int32_t* __restrict__ col,
                                                          const int32_t* __restrict__ long_count,
                                                          const int32_t* __restrict__ long_list,
                                                          int2* __restrict__ pairs) {
  const int nlong = *long_count;
  for (int q = blockIdx.x; q < nlong; q += gridDim.x) {
    const int i = long_list[q];
    for (int32_t k = row_ptr[i] + threadIdx.x; k < row_ptr[i + 1]; k += blockDim.x) pairs[k] = make_int2(i, col[k]);
  }
}

// The counting pass of every search structure parks the partners it finds in a slab, up to kSlab per body (entry-major:
// slab[a * n + slot], so the lanes of a wave write neighbouring words); the filling pass then copies the rows that fit
// and searches again only for the bodies with more partners than that -- on a monodisperse system none.  (Round 2 gave
// this to the tree traversal; round 4 to the cell grid, whose second walk of the stencil was 0.65 of the 1.4 ms a
// rebuild of the 10^6-rod list took.)
constexpr int kSlab = kShortSegment;
__device__ inline void row_from_slab(const RowSink& out, const int32_t* __restrict__ slab, size_t n, size_t slot, int i,
                                     int32_t base, int have) {
  for (int a = 0; a < have; ++a) out.col[base + a] = slab[static_cast<size_t>(a) * n + slot];
  finish_row(out, i, base, have);
}

template <bool FILL>
__global__ void __launch_bounds__(kBlock)
    k_pairs(size_t n, BpArgs A, const GridParams* __restrict__ gpp, const SearchRec* __restrict__ recs,
            const int32_t* __restrict__ slot_cell, const int32_t* __restrict__ cell_ptr,
            int32_t* __restrict__ counts, const int32_t* __restrict__ row_ptr, RowSink out,
            int32_t* __restrict__ slab) {
  const GridParams gp = *gpp;
  const size_t s = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (s >= n) return;
  const SearchRec me = recs[s];
  const int i = static_cast<int>(me.id);
  if (!is_source(A, i)) {
    if (!FILL) counts[i] = 0;
    return;
  }
  int32_t* col = out.col;
  if (FILL) {
    const int32_t b0 = row_ptr[i];
    const int have = row_ptr[i + 1] - b0;
    if (have <= kSlab) {  // the counting pass kept the whole row
      row_from_slab(out, slab, n, s, i, b0, have);
      return;
    }
  }
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
          if (!pair_allowed(A, i, j)) continue;
          const bool hit = (i <= j) ? volumes_overlap(A, me, o) : volumes_overlap(A, o, me);
          if (hit) {
            if (FILL) col[base + cnt] = j;
            else if (cnt < kSlab) slab[static_cast<size_t>(cnt) * n + s] = j;
            ++cnt;
          }
        }
      }
  if (!FILL) {
    counts[i] = cnt;
    return;
  }
  finish_row(out, i, base, cnt);
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
                int32_t* __restrict__ counts, const int32_t* __restrict__ row_ptr, RowSink out,
                int32_t* __restrict__ slab) {
  __shared__ __attribute__((aligned(16))) SearchRec tile[kPairTile];
  int32_t* col = out.col;
  const GridParams gp = *gpp;
  const size_t s0 = blockIdx.x * (size_t)kBlock;
  const size_t s = s0 + threadIdx.x;
  const bool live = s < n;
  SearchRec me{};
  int i = -1, cx = 0, cy = 0, cz = 0;
  if (live) {
    me = recs[s];
    i = static_cast<int>(me.id);
  }
  bool searching = live && is_source(A, i);  // a body outside the source set stages tiles but owns no row
  const int32_t base = (FILL && live) ? row_ptr[i] : 0;
  if (FILL) {
    // rows the counting pass parked in the slab are copied; the stencil is walked again only by a workgroup that has a
    // longer row (its other bodies then only help staging the tiles)
    bool walk = false;
    if (searching) {
      const int have = row_ptr[i + 1] - base;
      if (have <= kSlab) {
        row_from_slab(out, slab, n, s, i, base, have);
        searching = false;
      } else {
        walk = true;
      }
    }
    if (!__syncthreads_or(walk ? 1 : 0)) return;
  }
  if (live) {
    const int cid = slot_cell[s];
    cx = cid % gp.nc[0];
    cy = (cid / gp.nc[0]) % gp.nc[1];
    cz = cid / (gp.nc[0] * gp.nc[1]);
  }
  const int c_first = slot_cell[s0];
  const int c_last = slot_cell[(s0 + kBlock <= n ? s0 + kBlock : n) - 1];
  int cnt = 0;
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
      if (searching && z >= 0 && z < gp.nc[2] && y >= 0 && y < gp.nc[1]) {
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
          if (!pair_allowed(A, i, j)) continue;
          const bool hit = (i <= j) ? volumes_overlap(A, me, o) : volumes_overlap(A, o, me);
          if (hit) {
            if (FILL) col[base + cnt] = j;
            else if (cnt < kSlab) slab[static_cast<size_t>(cnt) * n + s] = j;
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
  if (searching) finish_row(out, i, base, cnt);
}

// ------------------------------------------------------------------------------------------------------------------
// Linear BVH over 63-bit Morton keys (free boundaries).  Node numbering: internal nodes 0 .. n-2 (0 is the root), leaf k
// (k-th body in key order) is node n-1+k.  A body's search volume is boxed (AABB kind: the grown box itself; sphere kind:
// centre -/+ grown radius); an internal node's box is the union of its children's, so "query box meets node box" is
// necessary for any leaf below it to pass the exact predicate -- the structure only prunes, the predicate decides, and
// the lists are those of the grid search and of the brute-force oracle.
// ------------------------------------------------------------------------------------------------------------------
struct BvhNode {  // 64 bytes: one line per visited node
  double lo[3], hi[3];
  int32_t left;   // first child (node id); its sibling is the left child's rope
  int32_t rope;   // next node in pre-order once this subtree is done or skipped, -1 = end
  int32_t first, last;  // the leaves (positions in key order) below this node
};

__device__ inline unsigned long long spread21(unsigned long long v) {  // 21 bits -> every third bit
  v &= 0x1fffffull;
  v = (v | (v << 32)) & 0x1f00000000ffffull;
  v = (v | (v << 16)) & 0x1f0000ff0000ffull;
  v = (v | (v << 8)) & 0x100f00f00f00f00full;
  v = (v | (v << 4)) & 0x10c30c30c30c30c3ull;
  v = (v | (v << 2)) & 0x1249249249249249ull;
  return v;
}
// summary = (min xyz, max xyz of the bin points, ...) written by k_grid_params
__global__ void __launch_bounds__(kBlock)
    k_morton_keys(size_t n, BpArgs A, const double* __restrict__ aabb, const double* __restrict__ center,
                  const double* __restrict__ brad, const double* __restrict__ summary,
                  unsigned long long* __restrict__ keys, unsigned* __restrict__ order) {
  const double ox = summary[0], oy = summary[1], oz = summary[2];
  const double ex = summary[3] - ox, ey = summary[4] - oy, ez = summary[5] - oz;
  const double scale = 2097152.0;  // 2^21 lattice points per axis
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    SearchRec r;
    V3 p;
    double rc;
    body_volume(A, i, aabb, center, brad, r, p, rc);
    auto q = [&](double v, double o, double e) -> unsigned long long {
      if (!(e > 0.0)) return 0ull;
      double f = floor((v - o) / e * scale);
      if (!(f >= 0.0)) f = 0.0;
      if (f > scale - 1.0) f = scale - 1.0;
      return static_cast<unsigned long long>(f);
    };
    keys[i] = spread21(q(p.x, ox, ex)) | (spread21(q(p.y, oy, ey)) << 1) | (spread21(q(p.z, oz, ez)) << 2);
    order[i] = static_cast<unsigned>(i);
  }
}

// leaf records in key order; leaf boxes
__device__ inline void rec_box(const BpArgs& A, const SearchRec& r, double lo[3], double hi[3]) {
  if (A.kind == MHIP_SEARCH_AABB) {
    for (int k = 0; k < 3; ++k) {
      lo[k] = r.a[k];
      hi[k] = r.b[k];
    }
  } else {
    // the sphere predicate compares rounded squares; pad the box by a few ulps so that a pair the predicate accepts by
    // rounding is never pruned (the box only prunes)
    for (int k = 0; k < 3; ++k) {
      const double pad = 8.9e-16 * (fabs(r.a[k]) + r.b[0]);
      lo[k] = r.a[k] - r.b[0] - pad;
      hi[k] = r.a[k] + r.b[0] + pad;
    }
  }
  if (A.periodic) {
    // Periodic tree: every box translated by the lattice vector that takes its midpoint into the primary cell (the
    // bin point the Morton key is made of); a query then walks the tree once per image of its own box that can meet the
    // tree at all.  The translation rounds, so the box is padded by a few ulps: boxes only prune, the exact predicate
    // (minimum image of the untranslated volumes) decides at the leaves.
    const double mid[3] = {0.5 * (lo[0] + hi[0]), 0.5 * (lo[1] + hi[1]), 0.5 * (lo[2] + hi[2])};
    const V3 w = periodic_wrap(A.pm, V3{mid[0], mid[1], mid[2]});
    for (int k = 0; k < 3; ++k) {
      const double t = comp(w, k) - mid[k];
      const double pad = 8.9e-16 * (fabs(lo[k]) + fabs(hi[k]) + fabs(t) + comp(A.pm.scale, k));
      lo[k] = (lo[k] + t) - pad;
      hi[k] = (hi[k] + t) + pad;
    }
  }
}
__global__ void __launch_bounds__(kBlock)
    k_lbvh_leaves(size_t n, BpArgs A, const double* __restrict__ aabb, const double* __restrict__ center,
                  const double* __restrict__ brad, const unsigned* __restrict__ order, SearchRec* __restrict__ recs,
                  int32_t* __restrict__ slot_of) {
  for (size_t k = blockIdx.x * (size_t)blockDim.x + threadIdx.x; k < n; k += (size_t)gridDim.x * blockDim.x) {
    SearchRec r;
    V3 p;
    double rc;
    body_volume(A, order[k], aabb, center, brad, r, p, rc);
    recs[k] = r;
    slot_of[order[k]] = static_cast<int32_t>(k);
  }
}

// length of the common prefix of the (key, position) strings of leaves a and b; -1 outside [0, n)
__device__ inline int lcp(const unsigned long long* __restrict__ keys, int n, int a, int b) {
  if (b < 0 || b >= n) return -1;
  const unsigned long long x = keys[a] ^ keys[b];
  if (x) return __clzll(static_cast<long long>(x));
  return 64 + __clz(a ^ b);  // equal keys: the position breaks the tie (Karras 2012, section 4)
}

// Karras' binary radix tree: internal node i covers the leaf range it shares its longest prefix with
__global__ void __launch_bounds__(kBlock)
    k_lbvh_build(int n, const unsigned long long* __restrict__ keys, BvhNode* __restrict__ nodes,
                 int32_t* __restrict__ right, int32_t* __restrict__ parent) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n - 1) return;
  const int d = (lcp(keys, n, i, i + 1) - lcp(keys, n, i, i - 1)) >= 0 ? 1 : -1;
  const int dmin = lcp(keys, n, i, i - d);
  int lmax = 2;
  while (lcp(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
  int l = 0;
  for (int t = lmax / 2; t >= 1; t /= 2)
    if (lcp(keys, n, i, i + (l + t) * d) > dmin) l += t;
  const int j = i + l * d;
  const int dnode = lcp(keys, n, i, j);
  int sp = 0;
  int t = l;
  do {
    t = (t + 1) / 2;
    if (lcp(keys, n, i, i + (sp + t) * d) > dnode) sp += t;
  } while (t > 1);
  const int gamma = i + sp * d + (d < 0 ? d : 0);
  const int lo = i < j ? i : j, hi = i < j ? j : i;
  const int lc = (lo == gamma) ? (n - 1 + gamma) : gamma;
  const int rc = (hi == gamma + 1) ? (n - 1 + gamma + 1) : (gamma + 1);
  nodes[i].left = lc;
  nodes[i].first = lo;
  nodes[i].last = hi;
  right[i] = rc;
  parent[lc] = i;
  parent[rc] = i;
  if (i == 0) parent[0] = -1;
}

// Boxes bottom-up: every leaf climbs; at each internal node the first arriver stops, the second (which then knows both
// children are done) forms the union and climbs on.  min / max are exact, so the boxes do not depend on who arrives
// first.  The ticket is an acquire-release atomic, and its scope is what the time hangs on: at agent scope every arrival
// writes back and invalidates its XCD's L2 (the other child may have been written on another XCD) -- 2 * 10^6 of those
// took 3.5 ms at 10^6 bodies.  So no ticket is taken at agent scope:
// Passes over growing chunks of consecutive leaves, one workgroup per chunk (256 leaves, then 4096, then all): a node
// whose leaf range lies inside a chunk is only ever visited by that workgroup's threads, and workgroup scope orders
// those (a wait for the stores, no cache maintenance: the waves of a workgroup share their CU's L1).  A thread that
// reaches a node whose range crosses the chunk boundary leaves its arrival in its leaf's slot for the next pass; the
// kernel boundary makes the boxes visible to it.  10^6 bodies: 0.24 + ~0.2 + ~0.1 ms.
__device__ inline void lbvh_union_children(int n, const BpArgs& A, const SearchRec* __restrict__ recs, BvhNode* nodes,
                                           const int32_t* __restrict__ right, int cur) {
  double lo[3], hi[3];
  for (int side = 0; side < 2; ++side) {
    const int c = side == 0 ? nodes[cur].left : right[cur];
    double clo[3], chi[3];
    if (c >= n - 1) {
      rec_box(A, recs[c - (n - 1)], clo, chi);
    } else {
      const volatile BvhNode* cn = nodes + c;
      for (int a = 0; a < 3; ++a) {
        clo[a] = cn->lo[a];
        chi[a] = cn->hi[a];
      }
    }
    for (int a = 0; a < 3; ++a) {
      lo[a] = side == 0 ? clo[a] : dmin(lo[a], clo[a]);
      hi[a] = side == 0 ? chi[a] : dmax(hi[a], chi[a]);
    }
  }
  for (int a = 0; a < 3; ++a) {
    nodes[cur].lo[a] = lo[a];
    nodes[cur].hi[a] = hi[a];
  }
}
// climbs from `cur` while the nodes lie inside the leaf range [chunk_first, chunk_last] this workgroup owns; returns the
// node at which the range was left (an arrival for a later pass), or -1 when the climb ended
__device__ inline int lbvh_climb(int n, const BpArgs& A, const SearchRec* __restrict__ recs, BvhNode* nodes,
                                 const int32_t* __restrict__ right, const int32_t* __restrict__ parent,
                                 int32_t* __restrict__ ticket, int cur, int chunk_first, int chunk_last) {
  while (cur >= 0) {
    if (nodes[cur].first < chunk_first || nodes[cur].last > chunk_last) return cur;
    if (__hip_atomic_fetch_add(&ticket[cur], 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) return -1;
    lbvh_union_children(n, A, recs, nodes, right, cur);
    cur = parent[cur];
  }
  return -1;
}
// pass over chunks of `chunk` consecutive leaves, one workgroup each.  slots_in == nullptr: the first pass, every leaf
// starts at its parent; otherwise leaf slot k holds the node at which the previous pass left its chunk (or -1).
// Where THIS pass leaves its (larger) chunk goes to slots_out[k] (dense: the next pass reads its own chunk's slots) or
// is appended to list_out ([0] = count; few entries, read by the single workgroup of the last pass).
__global__ void __launch_bounds__(1024)
    k_lbvh_refit_pass(int n, int chunk, BpArgs A, const SearchRec* __restrict__ recs, BvhNode* nodes,
                      const int32_t* __restrict__ right, const int32_t* __restrict__ parent,
                      int32_t* __restrict__ ticket, const int32_t* __restrict__ slots_in,
                      int32_t* __restrict__ slots_out, int32_t* __restrict__ list_out) {
  const long long first = static_cast<long long>(blockIdx.x) * chunk;
  const int chunk_first = static_cast<int>(first);
  const int chunk_last = static_cast<int>(first + chunk - 1 < n - 1 ? first + chunk - 1 : n - 1);
  for (int k = chunk_first + threadIdx.x; k <= chunk_last; k += blockDim.x) {
    const int start = slots_in ? slots_in[k] : parent[n - 1 + k];
    const int left_at = lbvh_climb(n, A, recs, nodes, right, parent, ticket, start, chunk_first, chunk_last);
    if (slots_out) slots_out[k] = left_at;
    if (list_out && left_at >= 0) list_out[1 + atomicAdd(&list_out[0], 1)] = left_at;
  }
}
// the last pass: one workgroup owns every leaf and replays the listed arrivals
__global__ void __launch_bounds__(1024)
    k_lbvh_refit_top(int n, BpArgs A, const SearchRec* __restrict__ recs, BvhNode* nodes,
                     const int32_t* __restrict__ right, const int32_t* __restrict__ parent,
                     int32_t* __restrict__ ticket, const int32_t* __restrict__ list_in) {
  const int count = list_in[0];
  for (int e = threadIdx.x; e < count; e += blockDim.x)
    lbvh_climb(n, A, recs, nodes, right, parent, ticket, list_in[1 + e], 0, n - 1);
}

// rope(x) = right sibling if x is a left child, else the rope of its parent
__global__ void __launch_bounds__(kBlock)
    k_lbvh_ropes(int n, BvhNode* __restrict__ nodes, const int32_t* __restrict__ right,
                 const int32_t* __restrict__ parent, int32_t* __restrict__ leaf_rope) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= 2 * n - 1) return;
  int cur = x, rope = -1;
  for (;;) {
    const int p = parent[cur];
    if (p < 0) break;
    if (nodes[p].left == cur) {
      rope = right[p];
      break;
    }
    cur = p;
  }
  if (x < n - 1) nodes[x].rope = rope;
  else leaf_rope[x - (n - 1)] = rope;
}

// one thread per body (in key order: the lanes of a wave walk neighbouring paths): stackless traversal.
// The tree is walked ONCE for most bodies: the counting pass parks a body's first kSlab partners in a slab
// (slab[slot * n + q]: lanes of a wave write neighbouring words), the filling pass copies them into the row and walks
// the tree again only for the bodies with more partners than that.
template <bool FILL>
__global__ void __launch_bounds__(kBlock)
    k_lbvh_pairs(int n, BpArgs A, const SearchRec* __restrict__ recs, const BvhNode* __restrict__ nodes,
                 const int32_t* __restrict__ leaf_rope, int32_t* __restrict__ counts,
                 const int32_t* __restrict__ row_ptr, RowSink out, int32_t* __restrict__ slab) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
  const SearchRec me = recs[q];
  const int i = static_cast<int>(me.id);
  if (!is_source(A, i)) {
    if (!FILL) counts[i] = 0;
    return;
  }
  int cnt = 0;
  const int32_t base = FILL ? row_ptr[i] : 0;
  if (FILL) {
    const int have = row_ptr[i + 1] - base;
    if (have <= kSlab) {  // the counting pass kept the whole row
      for (int a = 0; a < have; ++a) out.col[base + a] = slab[static_cast<size_t>(a) * n + q];
      finish_row(out, i, base, have);
      return;
    }
  }
  double blo[3], bhi[3];
  rec_box(A, me, blo, bhi);
  // periodic: one walk per image of the query box that meets the tree's root box -- for a body away from the faces of
  // the cell that is the unshifted one only; a leaf counts in the walk of the image its midpoint is nearest to (the
  // predicate's own image but for ties at half a box edge, where no volume the builder admits can overlap)
  const int nimg = A.periodic ? 27 : 1;
  double rlo[3], rhi[3], bmid[3];
  if (A.periodic) {
    for (int a = 0; a < 3; ++a) bmid[a] = 0.5 * (blo[a] + bhi[a]);
    if (n > 1) {
      for (int a = 0; a < 3; ++a) {
        rlo[a] = nodes[0].lo[a];
        rhi[a] = nodes[0].hi[a];
      }
    } else {
      rec_box(A, recs[0], rlo, rhi);
    }
  }
  for (int img = 0; img < nimg; ++img) {
    double qlo[3], qhi[3];
    int sh[3] = {0, 0, 0};
    if (A.periodic) {
      sh[0] = img % 3 - 1;
      sh[1] = (img / 3) % 3 - 1;
      sh[2] = img / 9 - 1;
      bool meets = true;
      for (int a = 0; a < 3; ++a) {
        const double L = comp(A.pm.scale, a);
        const double pad = sh[a] ? 8.9e-16 * (fabs(blo[a]) + fabs(bhi[a]) + L) : 0.0;
        qlo[a] = (blo[a] + sh[a] * L) - pad;
        qhi[a] = (bhi[a] + sh[a] * L) + pad;
        meets = meets && !(qhi[a] < rlo[a] || rhi[a] < qlo[a]);
      }
      if (!meets) continue;
    } else {
      for (int a = 0; a < 3; ++a) {
        qlo[a] = blo[a];
        qhi[a] = bhi[a];
      }
    }
    int node = 0;  // n == 1: node 0 is the only leaf
    while (node >= 0) {
      if (node < n - 1) {
        const BvhNode nd = nodes[node];
        const bool meet = !(qhi[0] < nd.lo[0] || qhi[1] < nd.lo[1] || qhi[2] < nd.lo[2] || nd.hi[0] < qlo[0] ||
                            nd.hi[1] < qlo[1] || nd.hi[2] < qlo[2]);
        node = meet ? nd.left : nd.rope;
      } else {
        const int k = node - (n - 1);
        const SearchRec o = recs[k];
        const int j = static_cast<int>(o.id);
        bool mine = true;
        if (A.periodic) {  // is this the image the leaf belongs to?
          double olo[3], ohi[3];
          rec_box(A, o, olo, ohi);
          for (int a = 0; a < 3; ++a) {
            const double d = 0.5 * (olo[a] + ohi[a]) - bmid[a];
            mine = mine && static_cast<int>(round(d * comp(A.pm.scale_inv, a))) == sh[a];
          }
        }
        if (mine && pair_allowed(A, i, j)) {
          const bool hit = (i <= j) ? volumes_overlap(A, me, o) : volumes_overlap(A, o, me);
          if (hit) {
            if (FILL) out.col[base + cnt] = j;
            else if (cnt < kSlab) slab[static_cast<size_t>(cnt) * n + q] = j;
            ++cnt;
          }
        }
        node = leaf_rope[k];
      }
    }
  }
  if (!FILL) {
    counts[i] = cnt;
    return;
  }
  finish_row(out, i, base, cnt);
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

// ---- results in the reference's vocabulary -------------------------------------------------------------------------------
// IdentProcIntersection pairs (GenNeighborLinkers.hpp:118-121): (entity id, owner rank) of source and target
__global__ void __launch_bounds__(kBlock)
    k_ident_pairs(size_t np, const int2* __restrict__ pairs, const uint64_t* __restrict__ id,
                  const int32_t* __restrict__ owner, uint64_t* __restrict__ src_id, int32_t* __restrict__ src_proc,
                  uint64_t* __restrict__ tgt_id, int32_t* __restrict__ tgt_proc) {
  for (size_t c = blockIdx.x * (size_t)blockDim.x + threadIdx.x; c < np; c += (size_t)gridDim.x * blockDim.x) {
    const int2 ij = pairs[c];
    if (src_id) src_id[c] = id ? id[ij.x] : static_cast<uint64_t>(ij.x);
    if (tgt_id) tgt_id[c] = id ? id[ij.y] : static_cast<uint64_t>(ij.y);
    if (src_proc) src_proc[c] = owner ? owner[ij.x] : 0;
    if (tgt_proc) tgt_proc[c] = owner ? owner[ij.y] : 0;
  }
}
// LinkCOOData rows: link entity id, linked entity ids [2], linked entity ranks [2] (LinkMetaData.hpp:102-106)
__global__ void __launch_bounds__(kBlock)
    k_links_coo(size_t np, const int2* __restrict__ pairs, const uint64_t* __restrict__ id, uint64_t first_link_id,
                unsigned char source_rank, unsigned char target_rank, uint64_t* __restrict__ link_id,
                uint64_t* __restrict__ linked_ids, unsigned char* __restrict__ linked_ranks) {
  for (size_t c = blockIdx.x * (size_t)blockDim.x + threadIdx.x; c < np; c += (size_t)gridDim.x * blockDim.x) {
    const int2 ij = pairs[c];
    if (link_id) link_id[c] = first_link_id + c;
    if (linked_ids) {
      linked_ids[2 * c] = id ? id[ij.x] : static_cast<uint64_t>(ij.x);
      linked_ids[2 * c + 1] = id ? id[ij.y] : static_cast<uint64_t>(ij.y);
    }
    if (linked_ranks) {
      linked_ranks[2 * c] = source_rank;
      linked_ranks[2 * c + 1] = target_rank;
    }
  }
}
__global__ void __launch_bounds__(kBlock) k_crs_count(size_t np, const int2* __restrict__ pairs,
                                                     int32_t* __restrict__ deg) {
  for (size_t c = blockIdx.x * (size_t)blockDim.x + threadIdx.x; c < np; c += (size_t)gridDim.x * blockDim.x) {
    const int2 ij = pairs[c];
    atomicAdd(&deg[ij.x], 1);
    atomicAdd(&deg[ij.y], 1);
  }
}
__global__ void __launch_bounds__(kBlock) k_crs_fill(size_t np, const int2* __restrict__ pairs,
                                                    int32_t* __restrict__ cursor, unsigned* __restrict__ conn) {
  for (size_t c = blockIdx.x * (size_t)blockDim.x + threadIdx.x; c < np; c += (size_t)gridDim.x * blockDim.x) {
    const int2 ij = pairs[c];
    conn[atomicAdd(&cursor[ij.x], 1)] = static_cast<unsigned>(c);
    conn[atomicAdd(&cursor[ij.y], 1)] = static_cast<unsigned>(c);
  }
}
// LinkCRSBucketConn per entity bucket (LinkCRSBucketConn.hpp:183-191): num_connected_links, bucket-local offsets
// [capacity + 1] and the connected links; bucket b holds entities [b * capacity, min((b + 1) * capacity, n))
__global__ void __launch_bounds__(kBlock)
    k_crs_emit(size_t n, size_t nconn, unsigned capacity, const int32_t* __restrict__ ptr, const unsigned* __restrict__ conn,
               uint64_t first_link_id, unsigned* __restrict__ num_connected, unsigned* __restrict__ bucket_offsets,
               uint64_t* __restrict__ connectivity, uint64_t* __restrict__ bucket_begin) {
  const size_t nb = (n + capacity - 1) / capacity;
  const size_t tid = blockIdx.x * (size_t)blockDim.x + threadIdx.x, nth = (size_t)gridDim.x * blockDim.x;
  for (size_t e = tid; e < n; e += nth) num_connected[e] = static_cast<unsigned>(ptr[e + 1] - ptr[e]);
  for (size_t q = tid; q < nb * (capacity + 1); q += nth) {
    const size_t b = q / (capacity + 1), k = q % (capacity + 1);
    const size_t first = b * capacity, e = first + k < n ? first + k : n;
    bucket_offsets[q] = static_cast<unsigned>(ptr[e] - ptr[first]);
  }
  for (size_t b = tid; b <= nb; b += nth) bucket_begin[b] = static_cast<uint64_t>(ptr[b * capacity < n ? b * capacity : n]);
  for (size_t t = tid; t < nconn; t += nth) connectivity[t] = first_link_id + conn[t];
}

__global__ void __launch_bounds__(kBlock) k_iota_u64(size_t n, uint64_t* __restrict__ out) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = i;
}
// AUTO picks the LBVH when the largest reach exceeds this multiple of the mean reach
constexpr double kLbvhReachSpread = 2.0;

}  // namespace mhip

using namespace mhip;

struct mhip_broadphase {
  mhip_broadphase_config cfg{};
  bool built = false;
  size_t n = 0, num_pairs = 0;
  int method_used = MHIP_SEARCH_METHOD_GRID;
  bool min_image_complete = true;  // last build: no two volumes could have met through any image but the nearest
  DeviceBuffer recs, cell_of, slot_cell, cell_cnt, cell_ptr, cursor, counts, row_ptr, col, pairs, old_center, params,
      partials, scanws, flag, longrows, coltmp;
  // LBVH
  DeviceBuffer keys, keys_tmp, order, order_tmp, sortws, nodes, right, parent, ticket, pending, leaf_rope, slot_of, slab;
  // seam S3: source / target sets, identities, exclusion lists (copies owned by the handle)
  size_t sets_n = 0, ident_n = 0, excl_n = 0;
  bool has_source = false, has_target = false, has_ident = false, has_excl = false;
  DeviceBuffer is_source, is_target, entity_id, owner_rank, ex_ptr, ex_idx;
  double* host_summary = nullptr;  // pinned, 8 doubles
  int* host_scalar = nullptr;      // pinned
};

extern "C" {

int mhip_broadphase_create(mhip_broadphase_t* handle) {
  MHIP_REQUIRE(handle != nullptr, MHIP_ERR_INVALID_ARGUMENT, "handle is null");
  *handle = new mhip_broadphase();  // no HIP call here: device/pinned memory is acquired by the first build
  return MHIP_SUCCESS;
}

static int ensure_host_scalar(mhip_broadphase* h) {
  if (!h->host_scalar) MHIP_HIP(hipHostMalloc(reinterpret_cast<void**>(&h->host_scalar), 64));
  if (!h->host_summary) MHIP_HIP(hipHostMalloc(reinterpret_cast<void**>(&h->host_summary), 8 * sizeof(double)));
  return MHIP_SUCCESS;
}

int mhip_broadphase_destroy(mhip_broadphase_t h) {
  if (!h) return MHIP_SUCCESS;
  for (DeviceBuffer* b : {&h->recs, &h->cell_of, &h->slot_cell, &h->cell_cnt, &h->cell_ptr, &h->cursor, &h->counts,
                          &h->row_ptr, &h->col, &h->pairs, &h->old_center, &h->params, &h->partials, &h->scanws,
                          &h->flag, &h->longrows, &h->coltmp, &h->keys, &h->keys_tmp, &h->order, &h->order_tmp,
                          &h->sortws, &h->nodes, &h->right, &h->parent, &h->ticket, &h->pending, &h->leaf_rope, &h->slot_of, &h->slab,
                          &h->is_source, &h->is_target, &h->entity_id, &h->owner_rank, &h->ex_ptr, &h->ex_idx})
    b->release();
  if (h->host_scalar) (void)hipHostFree(h->host_scalar);
  if (h->host_summary) (void)hipHostFree(h->host_summary);
  delete h;
  return MHIP_SUCCESS;
}

static int copy_in(DeviceBuffer& dst, const void* src, size_t bytes, hipStream_t s) {
  if (int e = dst.reserve(bytes + 16)) return e;
  if (bytes) MHIP_HIP(hipMemcpyAsync(dst.ptr, src, bytes, hipMemcpyDeviceToDevice, s));
  return MHIP_SUCCESS;
}

int mhip_broadphase_set_sets(mhip_broadphase_t h, size_t n, const unsigned char* is_source,
                             const unsigned char* is_target, mhip_stream_t stream) {
  MHIP_REQUIRE(h != nullptr, MHIP_ERR_INVALID_ARGUMENT, "broadphase handle is null");
  hipStream_t s = as_stream(stream);
  h->sets_n = n;
  h->has_source = is_source != nullptr;
  h->has_target = is_target != nullptr;
  if (is_source)
    if (int e = copy_in(h->is_source, is_source, n, s)) return e;
  if (is_target)
    if (int e = copy_in(h->is_target, is_target, n, s)) return e;
  h->built = false;  // the list no longer matches the sets: the next generate rebuilds
  return MHIP_SUCCESS;
}

int mhip_broadphase_set_identities(mhip_broadphase_t h, size_t n, const uint64_t* entity_id, const int32_t* owner_rank,
                                   mhip_stream_t stream) {
  MHIP_REQUIRE(h != nullptr, MHIP_ERR_INVALID_ARGUMENT, "broadphase handle is null");
  hipStream_t s = as_stream(stream);
  h->ident_n = n;
  h->has_ident = entity_id != nullptr || owner_rank != nullptr;
  if (int e = h->entity_id.reserve(n * sizeof(uint64_t) + 16)) return e;
  if (int e = h->owner_rank.reserve(n * sizeof(int32_t) + 16)) return e;
  if (entity_id) MHIP_HIP(hipMemcpyAsync(h->entity_id.ptr, entity_id, n * sizeof(uint64_t), hipMemcpyDeviceToDevice, s));
  else if (n) k_iota_u64<<<grid_for(n), kBlock, 0, s>>>(n, h->entity_id.as<uint64_t>());
  if (owner_rank) MHIP_HIP(hipMemcpyAsync(h->owner_rank.ptr, owner_rank, n * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
  else if (n) MHIP_HIP(hipMemsetAsync(h->owner_rank.ptr, 0, n * sizeof(int32_t), s));
  MHIP_LAUNCH_CHECK();
  return MHIP_SUCCESS;
}

int mhip_broadphase_set_exclusions(mhip_broadphase_t h, size_t n, const int32_t* ex_ptr, const int32_t* ex_idx,
                                   size_t num_entries, mhip_stream_t stream) {
  MHIP_REQUIRE(h != nullptr, MHIP_ERR_INVALID_ARGUMENT, "broadphase handle is null");
  hipStream_t s = as_stream(stream);
  h->has_excl = ex_ptr != nullptr;
  h->excl_n = n;
  if (ex_ptr) {
    MHIP_REQUIRE(ex_idx != nullptr || num_entries == 0, MHIP_ERR_INVALID_ARGUMENT, "ex_idx is null");
    if (int e = copy_in(h->ex_ptr, ex_ptr, (n + 1) * sizeof(int32_t), s)) return e;
    if (int e = copy_in(h->ex_idx, ex_idx, num_entries * sizeof(int32_t), s)) return e;
  }
  h->built = false;
  return MHIP_SUCCESS;
}

int mhip_broadphase_minimum_image_complete(mhip_broadphase_t h, int* complete) {
  MHIP_REQUIRE(h != nullptr && complete != nullptr, MHIP_ERR_INVALID_ARGUMENT, "null argument");
  *complete = h->min_image_complete ? 1 : 0;
  return MHIP_SUCCESS;
}

int mhip_broadphase_method_used(mhip_broadphase_t h, int* method) {
  MHIP_REQUIRE(h != nullptr && method != nullptr, MHIP_ERR_INVALID_ARGUMENT, "null argument");
  *method = h->method_used;
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
  MHIP_REQUIRE(config->method == MHIP_SEARCH_METHOD_AUTO || config->method == MHIP_SEARCH_METHOD_GRID ||
                   config->method == MHIP_SEARCH_METHOD_MORTON_LBVH,
               MHIP_ERR_INVALID_ARGUMENT, "unknown search method %d", config->method);
  MHIP_REQUIRE(config->buffer >= 0.0, MHIP_ERR_INVALID_ARGUMENT, "search buffer must be >= 0");
  MHIP_REQUIRE(n < (1u << 30), MHIP_ERR_RUNTIME, "too many bodies for 32-bit indices");
  if (config->search_kind == MHIP_SEARCH_AABB)
    MHIP_REQUIRE(aabb != nullptr || n == 0, MHIP_ERR_INVALID_ARGUMENT, "aabb is required for MHIP_SEARCH_AABB");
  else
    MHIP_REQUIRE(bounding_radius != nullptr || n == 0, MHIP_ERR_INVALID_ARGUMENT,
                 "bounding_radius is required for MHIP_SEARCH_SPHERES");
  MHIP_REQUIRE(center != nullptr || n == 0, MHIP_ERR_INVALID_ARGUMENT, "center is null");
  MHIP_REQUIRE(config->periodic >= 0 && config->periodic <= 2, MHIP_ERR_INVALID_ARGUMENT,
               "periodic must be 0 (free), 1 (orthorhombic box) or 2 (triclinic cell), got %d", config->periodic);
  if (config->periodic == 1) {
    MHIP_REQUIRE(config->box[0] > 0 && config->box[1] > 0 && config->box[2] > 0, MHIP_ERR_INVALID_ARGUMENT,
                 "periodic box must be positive");
  }
  if (config->periodic == 2) {
    const double det = determinant3(config->cell);
    MHIP_REQUIRE(det == det && det != 0.0 && fabs(det) <= 1.7976931348623157e308, MHIP_ERR_INVALID_ARGUMENT,
                 "triclinic unit cell must be invertible (determinant %g)", det);
  }
  MHIP_REQUIRE((!h->has_source && !h->has_target) || h->sets_n == n, MHIP_ERR_INVALID_ARGUMENT,
               "source / target sets were given for %zu bodies, the build has %zu", h->sets_n, n);
  MHIP_REQUIRE(!h->has_excl || h->excl_n == n, MHIP_ERR_INVALID_ARGUMENT,
               "exclusion lists were given for %zu bodies, the build has %zu", h->excl_n, n);
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
  A.pm = make_periodic(config->periodic == 1 ? config->box : one);
  A.triclinic = 0;
  A.aabb_c = aabb;
  A.center_c = center;
  A.brad_c = bounding_radius;
  double structure_box[3] = {config->box[0], config->box[1], config->box[2]};  // edges the structures wrap around
  if (config->periodic == 2) {
    A.triclinic = 1;
    A.tm = make_triclinic(config->cell);
    for (int a = 0; a < 3; ++a) {
      const double* row = A.tm.hi + 3 * a;
      A.gw[a] = 1.0 / std::sqrt(row[0] * row[0] + row[1] * row[1] + row[2] * row[2]);
      structure_box[a] = A.gw[a];
    }
    A.pm = make_periodic(A.gw);
  }
  A.include_self = config->include_self ? 1 : 0;
  A.is_source = h->has_source ? h->is_source.as<unsigned char>() : nullptr;
  A.is_target = h->has_target ? h->is_target.as<unsigned char>() : nullptr;
  A.ex_ptr = h->has_excl ? h->ex_ptr.as<int32_t>() : nullptr;
  A.ex_idx = h->has_excl ? h->ex_idx.as<int32_t>() : nullptr;

  if (int e = h->recs.reserve(n * sizeof(SearchRec))) return e;
  if (int e = h->counts.reserve((n + 2) * sizeof(int32_t))) return e;
  if (int e = h->old_center.reserve(3 * n * sizeof(double))) return e;
  if (int e = h->params.reserve(sizeof(GridParams) + 64 + 8 * sizeof(double))) return e;
  if (int e = h->partials.reserve((8 * kMaxGrid + 8) * sizeof(double))) return e;
  if (int e = h->longrows.reserve((n + 32) * sizeof(int32_t))) return e;
  {
    const size_t m = (size_t)cell_capacity > n ? (size_t)cell_capacity : n;
    if (int e = h->scanws.reserve(scan_workspace_bytes(m + 2) + 64)) return e;
  }
  if (int e = h->flag.reserve(64)) return e;

  GridParams* gp = h->params.as<GridParams>();
  double* summary = reinterpret_cast<double*>(h->params.as<char>() + sizeof(GridParams) + 8);
  const unsigned g = grid_for(n);
  k_bounds<<<g, kBlock, 0, s>>>(n, A, aabb, center, bounding_radius, h->partials.as<double>());
  MHIP_LAUNCH_CHECK();
  k_grid_params<<<1, kBlock, 0, s>>>((int)g, h->partials.as<double>(), A, cell_capacity, gp, summary);
  MHIP_LAUNCH_CHECK();

  // Which structure: the grid's cell edge is twice the LARGEST reach, so its work grows with (max / mean reach)^3; the
  // BVH adapts to every body's own size.  Measured on MI355X (profiles/r02_broadphase_methods.txt).
  int method = config->method;
  if (n >= 2 && (method == MHIP_SEARCH_METHOD_AUTO || (method == MHIP_SEARCH_METHOD_MORTON_LBVH && config->periodic))) {
    MHIP_HIP(hipMemcpyAsync(h->host_summary, summary, 8 * sizeof(double), hipMemcpyDeviceToHost, s));
    MHIP_HIP(hipStreamSynchronize(s));
    const double max_reach = h->host_summary[6], mean_reach = h->host_summary[7] / static_cast<double>(n);
    if (method == MHIP_SEARCH_METHOD_AUTO)
      method = max_reach > kLbvhReachSpread * mean_reach ? MHIP_SEARCH_METHOD_MORTON_LBVH : MHIP_SEARCH_METHOD_GRID;
    // the periodic tree assigns a leaf to ONE image of the query (the nearest midpoint): that is the predicate's image
    // as long as no two volumes can overlap across half a box edge; a cell that small is the grid's case
    if (config->periodic && method == MHIP_SEARCH_METHOD_MORTON_LBVH) {
      const double edge = std::min(structure_box[0], std::min(structure_box[1], structure_box[2]));
      if (!(4.0 * max_reach < edge)) method = MHIP_SEARCH_METHOD_GRID;
    }
  } else if (method == MHIP_SEARCH_METHOD_AUTO) {
    method = MHIP_SEARCH_METHOD_GRID;
  }
  h->method_used = method;
  int32_t* long_count = h->longrows.as<int32_t>();
  int32_t* long_list = long_count + 16;
  MHIP_HIP(hipMemsetAsync(long_count, 0, sizeof(int32_t), s));
  const unsigned gb = grid_exact(n);
  RowSink none{nullptr, nullptr, nullptr, nullptr};
  const int nn = static_cast<int>(n);

  // ---- count pass -----------------------------------------------------------------------------------------------
  const bool lds = !A.periodic;
  if (method == MHIP_SEARCH_METHOD_GRID) {
    if (int e = h->cell_of.reserve(n * sizeof(int32_t))) return e;
    if (int e = h->slot_cell.reserve(n * sizeof(int32_t))) return e;
    if (int e = h->cell_cnt.reserve((cell_capacity + 2) * sizeof(int32_t))) return e;
    if (int e = h->cell_ptr.reserve((cell_capacity + 2) * sizeof(int32_t))) return e;
    if (int e = h->cursor.reserve((cell_capacity + 2) * sizeof(int32_t))) return e;
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
    if (int e = h->slab.reserve((static_cast<size_t>(kSlab) * n + 2) * sizeof(int32_t))) return e;
    if (lds)
      k_pairs_lds<false><<<gb, kBlock, 0, s>>>(n, A, gp, h->recs.as<SearchRec>(), h->slot_cell.as<int32_t>(),
                                              h->cell_ptr.as<int32_t>(), h->counts.as<int32_t>(), nullptr, none,
                                              h->slab.as<int32_t>());
    else
      k_pairs<false><<<gb, kBlock, 0, s>>>(n, A, gp, h->recs.as<SearchRec>(), h->slot_cell.as<int32_t>(),
                                          h->cell_ptr.as<int32_t>(), h->counts.as<int32_t>(), nullptr, none,
                                          h->slab.as<int32_t>());
    MHIP_LAUNCH_CHECK();
  } else {
    if (int e = h->keys.reserve(n * sizeof(unsigned long long))) return e;
    if (int e = h->keys_tmp.reserve(n * sizeof(unsigned long long))) return e;
    if (int e = h->order.reserve(n * sizeof(unsigned))) return e;
    if (int e = h->order_tmp.reserve(n * sizeof(unsigned))) return e;
    if (int e = h->sortws.reserve(radix_sort_workspace_bytes(n))) return e;
    if (int e = h->nodes.reserve(n * sizeof(BvhNode))) return e;
    if (int e = h->right.reserve(n * sizeof(int32_t))) return e;
    if (int e = h->parent.reserve(2 * n * sizeof(int32_t))) return e;
    if (int e = h->ticket.reserve(n * sizeof(int32_t))) return e;
    if (int e = h->pending.reserve((2 * n + 1) * sizeof(int32_t))) return e;
    if (int e = h->leaf_rope.reserve(n * sizeof(int32_t))) return e;
    if (int e = h->slot_of.reserve(n * sizeof(int32_t))) return e;
    k_morton_keys<<<g, kBlock, 0, s>>>(n, A, aabb, center, bounding_radius, summary, h->keys.as<unsigned long long>(),
                                      h->order.as<unsigned>());
    MHIP_LAUNCH_CHECK();
    if (int e = radix_sort_u64(n, h->keys.as<unsigned long long>(), h->order.as<unsigned>(),
                               h->keys_tmp.as<unsigned long long>(), h->order_tmp.as<unsigned>(), 8, h->sortws.ptr, s))
      return e;
    k_lbvh_leaves<<<g, kBlock, 0, s>>>(n, A, aabb, center, bounding_radius, h->order.as<unsigned>(),
                                      h->recs.as<SearchRec>(), h->slot_of.as<int32_t>());
    MHIP_LAUNCH_CHECK();
    MHIP_HIP(hipMemsetAsync(h->parent.ptr, 0xFF, 2 * n * sizeof(int32_t), s));
    MHIP_HIP(hipMemsetAsync(h->ticket.ptr, 0, n * sizeof(int32_t), s));
    BvhNode* nodes = h->nodes.as<BvhNode>();
    if (n >= 2) {
      k_lbvh_build<<<grid_exact(n - 1), kBlock, 0, s>>>(nn, h->keys.as<unsigned long long>(), nodes,
                                                       h->right.as<int32_t>(), h->parent.as<int32_t>());
      MHIP_LAUNCH_CHECK();
      // passes over chunks of 256 leaves (one thread per leaf), then 4096, then one workgroup for what is left
      int32_t* slots = h->pending.as<int32_t>();
      int32_t* list = slots + n;  // [0] = count; at most one entry per leaf
      MHIP_HIP(hipMemsetAsync(list, 0, sizeof(int32_t), s));
      const bool two = nn > 4096;
      k_lbvh_refit_pass<<<gb, kBlock, 0, s>>>(nn, static_cast<int>(kBlock), A, h->recs.as<SearchRec>(), nodes,
                                             h->right.as<int32_t>(), h->parent.as<int32_t>(), h->ticket.as<int32_t>(),
                                             nullptr, two ? slots : nullptr, two ? nullptr : list);
      MHIP_LAUNCH_CHECK();
      if (two) {
        k_lbvh_refit_pass<<<static_cast<unsigned>((n + 4095) / 4096), 1024, 0, s>>>(
            nn, 4096, A, h->recs.as<SearchRec>(), nodes, h->right.as<int32_t>(), h->parent.as<int32_t>(),
            h->ticket.as<int32_t>(), slots, nullptr, list);
        MHIP_LAUNCH_CHECK();
      }
      k_lbvh_refit_top<<<1, 1024, 0, s>>>(nn, A, h->recs.as<SearchRec>(), nodes, h->right.as<int32_t>(),
                                         h->parent.as<int32_t>(), h->ticket.as<int32_t>(), list);
      MHIP_LAUNCH_CHECK();
    }
    k_lbvh_ropes<<<grid_exact(2 * n - 1), kBlock, 0, s>>>(nn, nodes, h->right.as<int32_t>(), h->parent.as<int32_t>(),
                                                         h->leaf_rope.as<int32_t>());
    MHIP_LAUNCH_CHECK();
    if (int e = h->slab.reserve((static_cast<size_t>(kSlab) * n + 2) * sizeof(int32_t))) return e;
    k_lbvh_pairs<false><<<gb, kBlock, 0, s>>>(nn, A, h->recs.as<SearchRec>(), nodes, h->leaf_rope.as<int32_t>(),
                                             h->counts.as<int32_t>(), nullptr, none, h->slab.as<int32_t>());
    MHIP_LAUNCH_CHECK();
  }

  // ---- row offsets, the one host read of the pair total ---------------------------------------------------------
  if (int e = exclusive_scan_i32(h->counts.as<int32_t>(), h->row_ptr.as<int32_t>(), n, h->scanws.ptr, s)) return e;
  MHIP_HIP(hipMemcpyAsync(h->host_scalar, h->row_ptr.as<int32_t>() + n, sizeof(int32_t), hipMemcpyDeviceToHost, s));
  if (config->periodic && n > 0)
    MHIP_HIP(hipMemcpyAsync(h->host_summary, summary, 8 * sizeof(double), hipMemcpyDeviceToHost, s));
  MHIP_HIP(hipStreamSynchronize(s));
  // the periodic predicate looks at the NEAREST image of a pair only (PeriodicScaledMetric::sep, periodicity.hpp:
  // 812-816, like every periodic distance of the reference): two volumes whose midpoints are d apart along an edge E
  // meet again at E - d >= E/2, which their reaches (<= 2 x the largest) cannot span when 4 x largest reach < E
  h->min_image_complete = true;
  if (config->periodic && n > 0)
    h->min_image_complete = 4.0 * h->host_summary[6] < std::min(structure_box[0], std::min(structure_box[1], structure_box[2]));
  const int32_t total = h->host_scalar[0];
  MHIP_REQUIRE(total >= 0, MHIP_ERR_RUNTIME, "pair count overflowed 32 bits");
  h->num_pairs = static_cast<size_t>(total);
  if (int e = h->col.reserve((h->num_pairs + 2) * sizeof(int32_t))) return e;
  if (int e = h->coltmp.reserve((h->num_pairs + 2) * sizeof(int32_t))) return e;
  if (int e = h->pairs.reserve((h->num_pairs + 2) * sizeof(int2))) return e;
  RowSink sink{h->col.as<int32_t>(), h->pairs.as<int2>(), long_count, long_list};

  // ---- fill pass ------------------------------------------------------------------------------------------------
  if (method == MHIP_SEARCH_METHOD_GRID) {
    if (lds)
      k_pairs_lds<true><<<gb, kBlock, 0, s>>>(n, A, gp, h->recs.as<SearchRec>(), h->slot_cell.as<int32_t>(),
                                             h->cell_ptr.as<int32_t>(), nullptr, h->row_ptr.as<int32_t>(), sink,
                                             h->slab.as<int32_t>());
    else
      k_pairs<true><<<gb, kBlock, 0, s>>>(n, A, gp, h->recs.as<SearchRec>(), h->slot_cell.as<int32_t>(),
                                         h->cell_ptr.as<int32_t>(), nullptr, h->row_ptr.as<int32_t>(), sink,
                                         h->slab.as<int32_t>());
  } else {
    k_lbvh_pairs<true><<<gb, kBlock, 0, s>>>(nn, A, h->recs.as<SearchRec>(), h->nodes.as<BvhNode>(),
                                            h->leaf_rope.as<int32_t>(), nullptr, h->row_ptr.as<int32_t>(), sink,
                                            h->slab.as<int32_t>());
  }
  MHIP_LAUNCH_CHECK();
  // rows with more than kShortSegment partners: workgroup radix sort, then their pairs
  int key_bits = 1;
  while ((size_t(1) << key_bits) < n) ++key_bits;
  if (int e = sort_listed_segments_u32(h->row_ptr.as<int32_t>(), h->col.as<unsigned>(), h->coltmp.as<unsigned>(),
                                       key_bits, long_count, long_list, s))
    return e;
  k_emit_long_rows<<<1024, kBlock, 0, s>>>(h->row_ptr.as<int32_t>(), h->col.as<int32_t>(), long_count, long_list,
                                          h->pairs.as<int2>());
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

int mhip_broadphase_get_ident_pairs(mhip_broadphase_t h, uint64_t* source_id, int32_t* source_proc, uint64_t* target_id,
                                    int32_t* target_proc, mhip_stream_t stream) {
  MHIP_REQUIRE(h != nullptr, MHIP_ERR_INVALID_ARGUMENT, "broadphase handle is null");
  MHIP_REQUIRE(h->built, MHIP_ERR_RUNTIME, "mhip_broadphase_build must be called before get_ident_pairs");
  MHIP_REQUIRE(!h->has_ident || h->ident_n == h->n, MHIP_ERR_INVALID_ARGUMENT,
               "identities were given for %zu bodies, the list has %zu", h->ident_n, h->n);
  if (h->num_pairs == 0) return MHIP_SUCCESS;
  k_ident_pairs<<<grid_for(h->num_pairs), kBlock, 0, as_stream(stream)>>>(
      h->num_pairs, h->pairs.as<int2>(), h->has_ident ? h->entity_id.as<uint64_t>() : nullptr,
      h->has_ident ? h->owner_rank.as<int32_t>() : nullptr, source_id, source_proc, target_id, target_proc);
  MHIP_LAUNCH_CHECK();
  return MHIP_SUCCESS;
}

int mhip_links_export_coo(mhip_broadphase_t h, uint64_t first_link_id, int source_rank, int target_rank,
                          uint64_t* link_id, uint64_t* linked_entity_ids, unsigned char* linked_entity_ranks,
                          mhip_stream_t stream) {
  MHIP_REQUIRE(h != nullptr, MHIP_ERR_INVALID_ARGUMENT, "broadphase handle is null");
  MHIP_REQUIRE(h->built, MHIP_ERR_RUNTIME, "mhip_broadphase_build must be called before the link export");
  MHIP_REQUIRE(source_rank >= 0 && source_rank < 256 && target_rank >= 0 && target_rank < 256,
               MHIP_ERR_INVALID_ARGUMENT, "entity ranks must fit 8 bits");
  MHIP_REQUIRE(!h->has_ident || h->ident_n == h->n, MHIP_ERR_INVALID_ARGUMENT,
               "identities were given for %zu bodies, the list has %zu", h->ident_n, h->n);
  if (h->num_pairs == 0) return MHIP_SUCCESS;
  k_links_coo<<<grid_for(h->num_pairs), kBlock, 0, as_stream(stream)>>>(
      h->num_pairs, h->pairs.as<int2>(), h->has_ident ? h->entity_id.as<uint64_t>() : nullptr, first_link_id,
      static_cast<unsigned char>(source_rank), static_cast<unsigned char>(target_rank), link_id, linked_entity_ids,
      linked_entity_ranks);
  MHIP_LAUNCH_CHECK();
  return MHIP_SUCCESS;
}

int mhip_links_export_crs(mhip_broadphase_t h, uint64_t first_link_id, unsigned bucket_capacity,
                          unsigned* num_connected_links, unsigned* sparse_connectivity_offsets,
                          uint64_t* sparse_connectivity, uint64_t* bucket_begin, mhip_stream_t stream) {
  MHIP_REQUIRE(h != nullptr, MHIP_ERR_INVALID_ARGUMENT, "broadphase handle is null");
  MHIP_REQUIRE(h->built, MHIP_ERR_RUNTIME, "mhip_broadphase_build must be called before the link export");
  MHIP_REQUIRE(bucket_capacity >= 1, MHIP_ERR_INVALID_ARGUMENT, "bucket capacity must be positive");
  const size_t n = h->n, np = h->num_pairs;
  if (n == 0) return MHIP_SUCCESS;
  MHIP_REQUIRE(num_connected_links && sparse_connectivity_offsets && bucket_begin && (sparse_connectivity || np == 0),
               MHIP_ERR_INVALID_ARGUMENT, "null output array");
  hipStream_t s = as_stream(stream);
  // entity -> links that name it (either ordinal): degrees, offsets, fill, each list ascending by link
  DeviceBuffer &deg = h->cell_cnt, &ptr = h->cell_ptr, &cur = h->cursor, &conn = h->keys, &tmp = h->keys_tmp;
  if (int e = deg.reserve((n + 2) * sizeof(int32_t))) return e;
  if (int e = ptr.reserve((n + 2) * sizeof(int32_t))) return e;
  if (int e = cur.reserve((n + 2) * sizeof(int32_t))) return e;
  if (int e = conn.reserve((2 * np + 2) * sizeof(unsigned))) return e;
  if (int e = tmp.reserve((2 * np + 2) * sizeof(unsigned))) return e;
  if (int e = h->scanws.reserve(scan_workspace_bytes(n + 2) + 64)) return e;
  if (int e = h->longrows.reserve((n + 32) * sizeof(int32_t))) return e;
  MHIP_HIP(hipMemsetAsync(deg.ptr, 0, (n + 1) * sizeof(int32_t), s));
  if (np) k_crs_count<<<grid_for(np), kBlock, 0, s>>>(np, h->pairs.as<int2>(), deg.as<int32_t>());
  MHIP_LAUNCH_CHECK();
  if (int e = exclusive_scan_i32(deg.as<int32_t>(), ptr.as<int32_t>(), n, h->scanws.ptr, s)) return e;
  MHIP_HIP(hipMemcpyAsync(cur.ptr, ptr.ptr, (n + 1) * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
  if (np) k_crs_fill<<<grid_for(np), kBlock, 0, s>>>(np, h->pairs.as<int2>(), cur.as<int32_t>(), conn.as<unsigned>());
  MHIP_LAUNCH_CHECK();
  int key_bits = 1;
  while ((size_t(1) << key_bits) < np + 1) ++key_bits;
  if (int e = sort_segments_u32(n, ptr.as<int32_t>(), conn.as<unsigned>(), tmp.as<unsigned>(), key_bits,
                                h->longrows.as<int32_t>(), s))
    return e;
  k_crs_emit<<<grid_for(2 * np + n), kBlock, 0, s>>>(n, 2 * np, bucket_capacity, ptr.as<int32_t>(), conn.as<unsigned>(),
                                                    first_link_id, num_connected_links, sparse_connectivity_offsets,
                                                    sparse_connectivity, bucket_begin);
  MHIP_LAUNCH_CHECK();
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
