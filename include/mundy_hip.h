/* mundy_hip.h -- C ABI of the MI355X (gfx950) implementation of MuNDy's contact hot path.
 *
 * One shared library, libmundy_hip.so, hand-written HIP underneath.  Every entry point returns an int status
 * (MHIP_SUCCESS == 0) and records a message retrievable with mhip_last_error() (thread local).  Unless a parameter is
 * marked [host], every pointer is a DEVICE pointer (hipMalloc / torch.cuda memory); `stream` is a hipStream_t passed
 * as void* (NULL = the null stream).  Entry points that return host scalars synchronise `stream`, exactly where the
 * reference's Kokkos parallel_reduce / deep_copy-to-host calls block; everything else is asynchronous.
 * Handles are not thread safe; use one handle per host thread / stream.
 *
 * All floating point is fp64; the device code is built with -ffp-contract=off and follows the reference's operation
 * order (right-fold dot, q*(0,v)*inverse(q), tolerance constants, branch structure), so per-element results are
 * bit-identical to a scalar evaluation of the reference formulas.
 *
 * "Replaces" lines cite the reference interface (path:line under the MuNDy tree) that a maintainer would route
 * through each entry point; INTEGRATION.md shows the C++ binding.  Layouts: centre [n][3], quaternion [n][4] as
 * (w,x,y,z), AABB [n][6] as (min xyz, max xyz), pairs [c][2] int32 (source, target).
 */
#ifndef MUNDY_HIP_H_
#define MUNDY_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* mhip_stream_t;

/* Status codes.  The C++ adapter (include/mundy_hip/adapter.hpp) maps them back to the exception types the reference
 * throws through MUNDY_THROW_REQUIRE (mundy/core/src/mundy_core/throw_assert.hpp:135-203). */
enum {
  MHIP_SUCCESS = 0,
  MHIP_ERR_INVALID_ARGUMENT = 1, /* std::invalid_argument: size mismatch, null pointer, bad enum */
  MHIP_ERR_LOGIC = 2,            /* std::logic_error */
  MHIP_ERR_RUNTIME = 3,          /* std::runtime_error: builder misuse, capacity */
  MHIP_ERR_HIP = 4,              /* a HIP runtime call failed (message carries hipGetErrorString) */
  MHIP_ERR_NO_DEVICE = 5         /* no gfx950 device visible: the library never falls back to the CPU */
};

const char* mhip_last_error(void);
int mhip_version(void);
/* roctx ranges around the stages of the path, named after the reference's functions / Kokkos kernel labels (visible to
 * rocprofv3 --marker-trace).  Off by default; MHIP_TRACE=1 in the environment enables it too. */
int mhip_set_tracing(int enable);
/* Fails with MHIP_ERR_NO_DEVICE when no GPU is visible. [host] out pointers. */
int mhip_device_info(int* device_count, char* arch_name, size_t arch_name_len);

/* Plain device memory for hosts that do not bring their own allocator. */
int mhip_malloc(void** ptr, size_t bytes);
int mhip_free(void* ptr);
int mhip_memcpy_h2d(void* dst, const void* src_host, size_t bytes, mhip_stream_t stream);
int mhip_memcpy_d2h(void* dst_host, const void* src, size_t bytes, mhip_stream_t stream); /* synchronises */
int mhip_stream_synchronize(mhip_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Per-body geometry (seam S4).
 * Replaces: compute_aabb(Sphere/Spherocylinder/Ellipsoid/SpherocylinderSegment)
 *           mundy/geom/src/mundy_geom/compute_aabb.hpp:72-143
 *           compute_bounding_radius(...) mundy/geom/src/mundy_geom/compute_bounding_radius.hpp:61-93
 * ---------------------------------------------------------------------------------------------------------------- */
int mhip_compute_aabb_spheres(size_t n, const double* center, const double* radius, double* aabb, mhip_stream_t stream);
int mhip_compute_aabb_spherocylinders(size_t n, const double* center, const double* quat, const double* radius,
                                      const double* length, double* aabb, mhip_stream_t stream);
int mhip_compute_aabb_ellipsoids(size_t n, const double* center, const double* quat, const double* radii,
                                 double* aabb, mhip_stream_t stream);
/* Build extension, the flagged option of SURVEY row a7: the reference's ellipsoid box (min/max of centre -/+ q*radii)
 * is exact only for axis-aligned rotations and is NOT conservative in general; this one is the tight box of the
 * rotated ellipsoid (half extent_k = sqrt(sum_j (r_j (q*e_j)_k)^2)).  No reference implementation: parity unpinned. */
int mhip_compute_aabb_ellipsoids_conservative(size_t n, const double* center, const double* quat, const double* radii,
                                              double* aabb, mhip_stream_t stream);
/* segment records seg[n][8] = (p0 xyz, p1 xyz, radius, 0) */
int mhip_compute_aabb_segments(size_t n, const double* seg, double* aabb, mhip_stream_t stream);
int mhip_bounding_radius_spherocylinders(size_t n, const double* radius, const double* length, double* out,
                                         mhip_stream_t stream);
int mhip_bounding_radius_ellipsoids(size_t n, const double* radii, double* out, mhip_stream_t stream);
/* Spherocylinder -> SpherocylinderSegment records: endpoints c -/+ 0.5*L*(q*zhat) (compute_aabb.hpp:115-117),
 * one 64-byte record per body so the pair kernels gather a single line per body. */
int mhip_spherocylinder_segments(size_t n, const double* center, const double* quat, const double* radius,
                                 const double* length, double* seg, mhip_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Narrow phase (seam S4).
 * Replaces: distance(SharedNormalSigned, Sphere, Sphere[, sep])     mundy/geom/src/mundy_geom/distance/SphereSphere.hpp:54-76
 *           distance(Point, LineSegment, cp, t, sep)                .../distance/PointLineSegment.hpp:128-172
 *           distance(LineSegment, LineSegment, cp1, cp2, s, t, sep) .../distance/LineSegmentLineSegment.hpp:189-318
 *           the per-linker contact kernels of the (deprecated) operator
 *           ComputeSignedSeparationDistanceAndContactNormal: scrap/lcp_spheres/NgpLcp.cpp:332-374 (spheres) and
 *           scrap/parameter_interface/linkers/.../SpherocylinderSpherocylinderLinker.cpp:207-247 (rods).
 * Any output pointer may be NULL to skip that output.
 * ---------------------------------------------------------------------------------------------------------------- */
/* element-wise batches over n independent object pairs (test / adapter entry points) */
int mhip_distance_sphere_sphere(size_t n, const double* c1, const double* r1, const double* c2, const double* r2,
                                double* dist, double* sep, mhip_stream_t stream);
int mhip_distance_point_segment(size_t n, const double* p, const double* a0, const double* a1, double* dist,
                                double* cp, double* t, double* sep, mhip_stream_t stream);
/* Replaces: distance(SharedNormalSigned, Point, Sphere[, sep])       .../distance/PointSphere.hpp:46-80
 *           distance(SharedNormalSigned, LineSegment, Sphere[, cp, arch_length, sep])
 *                                                                    .../distance/LineSegmentSphere.hpp:47-100
 * (the centre's distance to the point / to the segment, minus the radius; sep rescaled to the sphere's surface:
 * point -> surface, and for the segment the separation PointLineSegment hands back, centre -> closest point). */
int mhip_distance_point_sphere(size_t n, const double* p, const double* c, const double* r, double* dist, double* sep,
                               mhip_stream_t stream);
int mhip_distance_segment_sphere(size_t n, const double* a0, const double* a1, const double* c, const double* r,
                                 double* dist, double* cp, double* t, double* sep, mhip_stream_t stream);
int mhip_distance_segment_segment(size_t n, const double* a0, const double* a1, const double* b0, const double* b1,
                                  double* dist, double* cp1, double* cp2, double* s, double* t, double* sep,
                                  mhip_stream_t stream);
/* Ellipsoids: shared-normal signed distance by 3x3 multistart L-BFGS(10) over the normal's (theta, phi).
 * Replaces: distance(SharedNormalSigned, Ellipsoid, Ellipsoid, cp1, cp2, n1, n2)
 *           mundy/geom/src/mundy_geom/distance/EllipsoidEllipsoid.hpp:106-151, distance(Point, Ellipsoid, cp, n)
 *           .../PointEllipsoid.hpp:94-135, minimiser mundy/math/src/mundy_math/impl/minimize_impl.hpp:151-605.
 * radii [n][3] = body-frame semi-axes.  Tolerance against the host: 1e-4 (the reference's own,
 * UnitTestEllipsoidEllipsoid.cpp:52-53): device sin/cos differ from libm in the last bits. */
int mhip_distance_ellipsoid_ellipsoid(size_t n, const double* c1, const double* q1, const double* r1,
                                      const double* c2, const double* q2, const double* r2, double* dist, double* cp1,
                                      double* cp2, double* n1, double* n2, mhip_stream_t stream);
int mhip_distance_point_ellipsoid(size_t n, const double* p, const double* c, const double* q, const double* r,
                                  double* dist, double* cp, double* normal, mhip_stream_t stream);
/* neighbour-list form: sep, normal = n1, foot points cp1/cp2, lever arms ra/rb about the body centres */
int mhip_contact_ellipsoids(size_t c, const int32_t* pairs, const double* center, const double* quat,
                            const double* radii, double* sep, double* normal, double* cp1, double* cp2, double* ra,
                            double* rb, mhip_stream_t stream);
/* contact generation over a neighbour list: sep = |c_j - c_i| - r_i - r_j, normal = (c_j - c_i)/|c_j - c_i|.
 * box [host] = NULL (free space) or 3 doubles: PeriodicScaledMetric::sep (periodicity.hpp:812-816) minimum image. */
int mhip_contact_spheres(size_t c, const int32_t* pairs, const double* center, const double* radius,
                         const double* box, double* sep, double* normal, mhip_stream_t stream);
/* rods: seg from mhip_spherocylinder_segments; sep = dist(seg_i, seg_j) - (r_i + r_j); normal = (cp2 - cp1)/dist;
 * cp1/cp2 centreline closest points; ra/rb = cp - body centre (lever arms); s/t arclength parameters. */
int mhip_contact_spherocylinders(size_t c, const int32_t* pairs, const double* seg, const double* center, double* sep,
                                 double* normal, double* cp1, double* cp2, double* ra, double* rb, double* s,
                                 double* t, mhip_stream_t stream);
/* The same in an orthorhombic periodic box [host: 3 edge lengths]: rod j is taken at the lattice image whose centre is
 * nearest to rod i's centre -- shift = (c_i + PeriodicScaledMetric::sep(c_i, c_j)) - c_j (periodicity.hpp:812-816), a
 * rigid translation as wrap_rigid applies to a spherocylinder (:1094-1113).  Contact points are in rod i's image; rb is
 * taken from the shifted centre of rod j.  center is required. */
int mhip_contact_spherocylinders_periodic(size_t c, const int32_t* pairs, const double* seg, const double* center,
                                          const double* box /*[host] 3*/, double* sep, double* normal, double* cp1,
                                          double* cp2, double* ra, double* rb, double* s, double* t,
                                          mhip_stream_t stream);

/* Mixed shapes (BASELINE configs[4]): kind[n] = 0 sphere, 1 spherocylinder, 2 ellipsoid; shape[n][3] = (r,-,-) /
 * (r,L,-) / (r1,r2,r3); quat is ignored for spheres.  compute_aabb dispatches on kind (compute_aabb.hpp:72-127) and
 * also returns the bounding radii (compute_bounding_radius.hpp:61-93).  contact_mixed bins the pairs by shape class and
 * runs one distance routine per class: S-S, S-R (scrap/.../SphereSpherocylinderLinker.cpp:210-239), R-R, E-E as above;
 * S-E = distance(Point, Ellipsoid) - r (exact by default, see mhip_contact_mixed_set_sphere_ellipsoid_route) and R-E = closest approach of the rod's centreline to the ellipsoid (exact
 * signed point - ellipsoid distance minimised along the centreline, closed form: csrc/segment_ellipsoid.hpp) - r are
 * build extensions (the reference's SphereEllipsoid.hpp / LineSegmentEllipsoid.hpp are empty stubs): parity unpinned.
 * class_counts [host, 6] (optional) = pairs per class in the order SS, SR, SE, RR, RE, EE (synchronises if given). */
int mhip_compute_aabb_mixed(size_t n, const int32_t* kind, const double* center, const double* quat,
                            const double* shape, double* aabb, double* bounding_radius, mhip_stream_t stream);
/* BUILD EXTENSION (SURVEY a7's flagged option): the same with the TIGHT CONSERVATIVE box for the ellipsoids (half extent
 * along lab axis k = sqrt(sum_j (r_j (q e_j)[k])^2)).  The reference's box (centre -/+ q*radii, compute_aabb.hpp:82-103)
 * is exact only for axis-aligned orientations: for a general rotation an extent can collapse towards zero, and a
 * neighbour search on those boxes silently misses overlapping ellipsoid pairs.  Spheres and rods are unchanged. */
int mhip_compute_aabb_mixed_conservative(size_t n, const int32_t* kind, const double* center, const double* quat,
                                         const double* shape, double* aabb, double* bounding_radius,
                                         mhip_stream_t stream);
int mhip_contact_mixed(size_t c, const int32_t* pairs, const int32_t* kind, const double* center, const double* quat,
                       const double* shape, double* sep, double* normal, double* cp1, double* cp2, double* ra,
                       double* rb, size_t* class_counts /*[host]*/, mhip_stream_t stream);
/* BUILD OPTION, labelled wherever it is exposed: on = 1 makes the following mhip_contact_mixed* calls (any thread) run
 * the S-E and E-E minimisation classes from a build with floating-point contraction ON (fused multiply-adds):
 * their results then agree with the default build -- which is bit-identical to the CPU oracle -- only to the
 * reference's own tolerance for ellipsoid distances, 1e-4 (UnitTestEllipsoidEllipsoid.cpp:53), on >= 99.5 % of pairs
 * (tests/test_gpu_mixed.py).  The closed-form classes (S-S, S-R, R-R, R-E) are not affected.  Default 0. */
int mhip_contact_mixed_set_contraction(int on);
/* S-E (a build extension: the reference's SphereEllipsoid.hpp is an empty stub) = signed distance of the sphere's centre
 * to the ellipsoid, minus the radius.  route 0 (default): the exact distance in closed form (csrc/segment_ellipsoid.hpp);
 * route 1: distance(Point, Ellipsoid) as the reference computes it (PointEllipsoid.hpp:94-135, nine-start L-BFGS over
 * the surface normal).  ROUTE 1 IS THE REFERENCE-ROUTINE ROUTE, the one SURVEY 8f.4 prescribes for S-E; route 0 solves
 * the same problem (closest surface point in the Euclidean sense, signed by inside / outside) exactly and matches
 * route 1 to that routine's own 1e-4 wherever its minimiser reaches the global minimum.  Where the two differ by more
 * than 1e-4 (the nine starts stall in a local minimum: needles, flakes, points near a long axis; < 0.5 % of the pairs
 * of the mixed packing) the tests certify PAIR BY PAIR that route 0's foot point is the closer one and satisfies the
 * optimality conditions (tests/test_gpu_mixed.py, tests/test_oracle_ellipsoid_kat.py:
 * certify_sphere_ellipsoid_disagreements).  Applies to the following mhip_contact_mixed* calls of any thread. */
int mhip_contact_mixed_set_sphere_ellipsoid_route(int route);
/* objective evaluations the L-BFGS classes (S-E, E-E; the middle word was R-E's, closed-form since round 3: 0)
 * needed in the last mhip_contact_mixed* call of this host thread, and in the last mhip_distance_ellipsoid_* / mhip_contact_ellipsoids call: these kernels are fp64-vector bound
 * (about 2.3 * 10^3 fp64 instructions per evaluation), so evaluations x that / time is their roofline figure */
int mhip_contact_mixed_last_evaluations(unsigned long long evaluations[3] /*[host]*/, mhip_stream_t stream);
int mhip_ellipsoid_last_evaluations(unsigned long long* evaluations /*[host]*/, mhip_stream_t stream);
/* periodic box [host: 3 edge lengths]: body j at the nearest lattice image of its centre, c_j' = c_i + sep(c_i, c_j) */
int mhip_contact_mixed_periodic(size_t c, const int32_t* pairs, const int32_t* kind, const double* center,
                                const double* quat, const double* shape, const double* box /*[host] 3*/, double* sep,
                                double* normal, double* cp1, double* cp2, double* ra, double* rb,
                                size_t* class_counts /*[host]*/, mhip_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Broad phase (seam S3).
 * Replaces: stk::search::coarse_search(domain, range, MORTON_LBVH, comm, results, exec, symmetry) +
 *           SearchFilter::apply / filter_view compaction     mundy/mesh/src/mundy_mesh/GenNeighborLinkers.hpp:650-664,
 *           :141-183; the rebuild rule :603-615; scrap: ArborX BVH query scrap/lcp_spheres/NgpLcp.cpp:297-330.
 * Predicate (defined here; the reference's is third party): MHIP_SEARCH_SPHERES = bounding spheres
 * (centre, bounding_radius + buffer), closed test d^2 <= (Ri + Rj)^2; MHIP_SEARCH_AABB = aabb grown by buffer on every
 * face, closed interval test of geom::intersects (AABB.hpp:420-431).  Results are canonical: sorted by (i, j);
 * symmetric = 0 gives unique i < j pairs, 1 gives both orientations; i == j only with include_self.
 * Search method (GenNeighborLinks::set_search_method, :443-447; the reference's default is stk::search::MORTON_LBVH):
 * the uniform cell grid (cell edge = twice the largest reach; fastest when bodies are of similar size), a linear BVH over
 * 63-bit Morton keys (adapts to each body's size: size-disperse systems), or AUTO = LBVH when the largest reach exceeds
 * twice the mean reach.  Periodic systems: the tree holds every volume translated into the primary cell and is walked
 * once per image of the query that can meet it; a cell with an edge below four times the largest reach goes to the grid
 * whatever was asked for (mhip_broadphase_method_used tells).  Same lists either way.
 * ---------------------------------------------------------------------------------------------------------------- */
enum { MHIP_SEARCH_SPHERES = 0, MHIP_SEARCH_AABB = 1 };
enum { MHIP_SEARCH_METHOD_AUTO = 0, MHIP_SEARCH_METHOD_GRID = 1, MHIP_SEARCH_METHOD_MORTON_LBVH = 2 };

typedef struct mhip_broadphase* mhip_broadphase_t;

typedef struct {
  int search_kind; /* MHIP_SEARCH_* */
  int symmetric;   /* 0: i<j ; 1: i!=j both orders */
  double buffer;   /* search buffer added to every volume */
  int periodic;    /* 0 free space, 1 orthorhombic periodic box [0,box), 2 triclinic unit cell `cell` */
  double box[3];
  int method;       /* MHIP_SEARCH_METHOD_* (0 = AUTO) */
  int include_self; /* 0 = ExcludeSelfInteractions (GenNeighborLinkers.hpp:185-200), 1 = (i, i) is a result */
  /* periodic == 2: the unit cell of PeriodicMetric (periodicity.hpp:233-332), row-major 3 x 3, lattice vectors as
   * COLUMNS.  A pair is tested at the image PeriodicMetric::sep picks for the centres / box midpoints (minimum image of
   * the fractional coordinates, :304-307).  Grid and tree work in scaled fractional coordinates (the cell becomes an
   * orthorhombic box of the three perpendicular widths of the cell); same lists as the brute-force statement. */
  double cell[9];
} mhip_broadphase_config;

int mhip_broadphase_create(mhip_broadphase_t* handle);
int mhip_broadphase_destroy(mhip_broadphase_t handle);
/* Builds the neighbour list of n bodies; *num_pairs [host] receives the pair count (synchronises, as filter_view's
 * deep_copy of the scan total does, GenNeighborLinkers.hpp:155-156).  Also snapshots the centres for
 * mhip_broadphase_needs_rebuild. */
int mhip_broadphase_build(mhip_broadphase_t handle, const mhip_broadphase_config* config /*[host]*/, size_t n,
                          const double* aabb, const double* center, const double* bounding_radius,
                          size_t* num_pairs /*[host]*/, mhip_stream_t stream);
/* Copies the pair list ([num_pairs][2] int32) and/or the CSR form (row_ptr[n+1], col[num_pairs]); NULL skips. */
int mhip_broadphase_get_pairs(mhip_broadphase_t handle, int32_t* pairs, int32_t* row_ptr, int32_t* col,
                              mhip_stream_t stream);
/* The rest of the seam: who may pair with whom, and results in the reference's vocabulary.
 *   set_sets        acts_on(source_selector, target_selector) (GenNeighborLinkers.hpp:486-507): is_source / is_target
 *                   [n] bytes (NULL = every body); a result (s, t) needs s in the sources and t in the targets -- with
 *                   symmetric = 0 additionally s < t.  The reference's domain and range views are two selections of one
 *                   entity population (specializations_, :626-629), which is what the two masks express.
 *   set_exclusions  search_filters::ExcludeConnectedEntities (:202-236) and the existing-link set kept when duplicate
 *                   links are not allowed (:91-113, :643-648): CSR (ex_ptr [n + 1], ex_idx) of partners a source must
 *                   not be paired with; the ordered pair (s, t) is dropped when t is in the list of s.
 *   set_identities  (stk::mesh::EntityId, owner rank) of every body (:575-584); NULL id = the local index, NULL owner = 0
 * All three copy their arrays into the handle; sets / exclusions invalidate the current list (the next needs_rebuild
 * says yes).  get_ident_pairs returns the list as IdentProcIntersection rows (domain id, domain proc, range id, range
 * proc), any output NULL to skip.  method_used: which structure the last build ran on (MHIP_SEARCH_METHOD_*). */
int mhip_broadphase_set_sets(mhip_broadphase_t handle, size_t n, const unsigned char* is_source,
                             const unsigned char* is_target, mhip_stream_t stream);
int mhip_broadphase_set_exclusions(mhip_broadphase_t handle, size_t n, const int32_t* ex_ptr, const int32_t* ex_idx,
                                   size_t num_entries, mhip_stream_t stream);
int mhip_broadphase_set_identities(mhip_broadphase_t handle, size_t n, const uint64_t* entity_id,
                                   const int32_t* owner_rank, mhip_stream_t stream);
int mhip_broadphase_get_ident_pairs(mhip_broadphase_t handle, uint64_t* source_id, int32_t* source_proc,
                                    uint64_t* target_id, int32_t* target_proc, mhip_stream_t stream);
int mhip_broadphase_method_used(mhip_broadphase_t handle, int* method /*[host]*/);
/* Periodic cells: the predicate tests a pair at the NEAREST image of its centres / box midpoints only -- the minimum
 * image of PeriodicScaledMetric::sep (periodicity.hpp:812-816), as every periodic distance of the reference does.  That
 * finds every overlap as long as the smallest cell edge exceeds four times the largest reach (half extent + buffer) of
 * the bodies; in a smaller cell two volumes can also meet through a second image, and such pairs are NOT reported.
 * *complete [host] = 1 when the last build's cell met the condition (always 1 in free space), 0 otherwise: a caller
 * that needs the multi-image list has to replicate the cell.  No synchronisation (the build has read the figure). */
int mhip_broadphase_minimum_image_complete(mhip_broadphase_t handle, int* complete /*[host]*/);
/* The list in MuNDy's link layout, so that it can be handed to LinkData without the host-serial
 * request_link loop of GenNeighborLinkers.hpp:714-738:
 *   export_coo  one row per link, as the fields LinkCOOData keeps per link entity (LinkMetaData.hpp:102-106): the link's
 *               own id first_link_id + k, MUNDY_LINKED_ENTITY_IDS [P][2] (source, target), MUNDY_LINKED_ENTITY_RANKS
 *               [P][2] (bytes); any output NULL to skip
 *   export_crs  entity -> connected links, as LinkCRSBucketConn keeps it per entity bucket (LinkCRSBucketConn.hpp:
 *               183-191): entities in index order cut into buckets of bucket_capacity (STK's default bucket: 512);
 *               num_connected_links [n]; sparse_connectivity_offsets [num_buckets][capacity + 1], bucket local;
 *               sparse_connectivity [2 P] link ids (a link is connected to both its entities), ascending per entity;
 *               bucket_begin [num_buckets + 1] = where each bucket's connectivity starts in sparse_connectivity */
int mhip_links_export_coo(mhip_broadphase_t handle, uint64_t first_link_id, int source_rank, int target_rank,
                          uint64_t* link_id, uint64_t* linked_entity_ids, unsigned char* linked_entity_ranks,
                          mhip_stream_t stream);
int mhip_links_export_crs(mhip_broadphase_t handle, uint64_t first_link_id, unsigned bucket_capacity,
                          unsigned* num_connected_links, unsigned* sparse_connectivity_offsets,
                          uint64_t* sparse_connectivity, uint64_t* bucket_begin, mhip_stream_t stream);
/* Rebuild rule: *flag [host] = 1 iff any centre moved more than 0.5*buffer since the last build (synchronises). */
int mhip_broadphase_needs_rebuild(mhip_broadphase_t handle, size_t n, const double* center, int* flag /*[host]*/,
                                  mhip_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Convex solver backend (seam S1) -- the static interface of convex::KokkosBackend as a C ABI.
 * Replaces: mundy/math/src/mundy_math/convex.hpp:141-285 (deep_copy, apply/gemv, axpby, wrapped_axpbyz, diff_dot x2,
 *           reduce_max), residual policies :434-496, BBStepStrategy :498-516.  Functors cannot cross a C ABI, so the
 *           four convex spaces (:46-115) and the two residual policies are enums + scalars.
 * ---------------------------------------------------------------------------------------------------------------- */
enum { MHIP_SPACE_UNCONSTRAINED = 0, MHIP_SPACE_LOWER_BOUND = 1, MHIP_SPACE_UPPER_BOUND = 2, MHIP_SPACE_BOUNDED = 3 };
enum { MHIP_RESIDUAL_PROJECTED_DIFF = 0, MHIP_RESIDUAL_PROJECTED_GRADIENT = 1 };

typedef struct {
  int kind; /* MHIP_SPACE_* */
  double lower_bound, upper_bound;
} mhip_space;

typedef struct {
  unsigned max_iters; /* PGDConfig::max_iters, default 1000 (convex.hpp:523) */
  double tol;         /* PGDConfig::tol, default 1e-8 (convex.hpp:524) */
  int residual_kind;  /* MHIP_RESIDUAL_*; default policy is PROJECTED_DIFF (convex.hpp:762-768) */
} mhip_pgd_config;

typedef struct {
  unsigned num_iters; /* completed non-terminal iterations (convex.hpp:664) */
  double residual;
  int converged; /* non-convergence is not an error (convex.hpp:673-675) */
} mhip_solve_result;

int mhip_deep_copy(size_t n, double* dst, const double* src, mhip_stream_t stream);
int mhip_fill(size_t n, double* dst, double value, mhip_stream_t stream);
int mhip_axpby(size_t n, double alpha, const double* x, double beta, double* y, mhip_stream_t stream);
int mhip_wrapped_axpbyz(size_t n, double alpha, const double* x, double beta, const double* y, double* z,
                        const mhip_space* space /*[host]*/, mhip_stream_t stream);
int mhip_diff_dot2(size_t n, const double* x, const double* y, double* result /*[host]*/, mhip_stream_t stream);
int mhip_diff_dot4(size_t n, const double* x1, const double* x2, const double* y1, const double* y2,
                   double* result /*[host]*/, mhip_stream_t stream);
int mhip_residual(size_t n, int residual_kind, const double* x, const double* grad, const mhip_space* space /*[host]*/,
                  double* result /*[host]*/, mhip_stream_t stream);
int mhip_bb_step(size_t n, const double* x_old, const double* g_old, const double* x, const double* g,
                 double* result /*[host]*/, mhip_stream_t stream);
/* y = A x, A row-major n x n (KokkosBlas::gemv "N", convex.hpp:168-174) */
int mhip_gemv(size_t n, const double* A, const double* x, double* y, mhip_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Matrix-free contact operator (seam S2):  y = dt * D^T M D x  over a neighbour list.
 * Replaces: sum_collision_force / compute_the_mobility_problem (dry) / compute_rate_of_change_of_sep,
 *           scrap/lcp_spheres/NgpLcp.cpp:442-530; rigid-body torque terms follow the dry local-drag mobility of
 *           scrap/parameter_interface/alens/tests/performance_tests/Bacteria.cpp:808-848 (U = F/(6 pi mu r),
 *           W = T/(8 pi mu r^3)), passed in as per-body scalars.
 * ra, rb, mob_rot may all be NULL (translation only, the sphere app).  The handle keeps views of the caller's
 * arrays (they must outlive it) and owns a body->constraint incidence index, so sums run in a fixed order (bitwise
 * reproducible, no atomics).
 * priority [C] or NULL: a locality hint for that index only (never the result beyond summation-order rounding) -- each
 * body lists the contacts with priority < 0 first.  Pass the signed separations: the contacts that overlap at the
 * start of the step are the ones that will carry impulses, and the body sweep of the LCP solver only walks those.
 * ---------------------------------------------------------------------------------------------------------------- */
typedef struct mhip_contact_op* mhip_contact_op_t;

int mhip_contact_op_create(mhip_contact_op_t* handle, size_t num_constraints, size_t num_bodies, const int32_t* pairs,
                           const double* normal, const double* ra, const double* rb, const double* mob_trans,
                           const double* mob_rot, double dt, const double* priority, mhip_stream_t stream);
/* Spherocylinders: the same operator with rod-compressed kinematics.  A rod's contact point lies on its centreline,
 * cp = c + (s - 1/2)(p1 - p0), so each lever arm is one scalar; the sweeps stream (s, t) [16 B per contact] instead of
 * (ra, rb) [48 B] and 32-byte instead of 48-byte half-edge records (-18 % bytes per BBPGD iteration).  arc_s / arc_t
 * [C] and seg [N][8] are the outputs of mhip_contact_spherocylinders / mhip_spherocylinder_segments.  Same operator up
 * to rounding (the arm (s - 1/2) u replaces (p0 + s u) - c). */
int mhip_contact_op_create_rods(mhip_contact_op_t* handle, size_t num_constraints, size_t num_bodies,
                                const int32_t* pairs, const double* normal, const double* arc_s, const double* arc_t,
                                const double* seg, const double* mob_trans, const double* mob_rot, double dt,
                                const double* priority, mhip_stream_t stream);
/* Same pairs, new geometry (a step that reuses the neighbour list): keeps the incidence index and redoes only the
 * half-edge records / rod axes from the new arrays (which replace the ones given at create; mobilities and dt stay).
 * A quarter of the cost of building the operator again.  The order inside a body's list is the one fixed at create. */
int mhip_contact_op_refresh(mhip_contact_op_t handle, const double* normal, const double* ra, const double* rb,
                            mhip_stream_t stream);
int mhip_contact_op_refresh_rods(mhip_contact_op_t handle, const double* normal, const double* arc_s,
                                 const double* arc_t, const double* seg, mhip_stream_t stream);
int mhip_contact_op_destroy(mhip_contact_op_t handle);
/* A destroyed operator leaves its device workspaces in one process-wide spare set that the next create adopts (no
 * allocation in the steady state of a time loop); this frees that set. */
int mhip_release_cached_workspaces(void);
int mhip_contact_op_apply(mhip_contact_op_t handle, const double* x, double* y, mhip_stream_t stream);
/* Per-kernel timing of the fused solver (measurement support, no effect on results): when enabled,
 * mhip_bbpgd_solve_contact brackets the k_body / k_constraint launches of every 8th iteration with HIP events on
 * `stream` and accumulates their device durations.  get_profile returns the totals in milliseconds and the number of timed iterations
 * (sampled launches of each kernel) since profiling was enabled.  All out pointers [host]. */
/* sizes the operator was created with (either pointer may be NULL) */
int mhip_contact_op_sizes(mhip_contact_op_t handle, size_t* num_constraints, size_t* num_bodies);
/* How the two sweeps are laid out on the chip -- time only, never results: every sum that reaches an iterate is a
 * double-double pair rounded once, so any mapping gives the same bits (tests/test_gpu_convex.py checks exactly that).
 *   xcd_tile        consecutive 256-item tiles handed to one XCD (8 XCDs with private L2s; 0 = identity mapping,
 *                   default 32); -1 keeps the current value
 *   lanes_per_body  lanes that share one body's contact list in the body sweep: 2, 4, 8 or 16 (default by mean
 *                   degree); -1 keeps the current value */
int mhip_contact_op_set_work_mapping(mhip_contact_op_t handle, int xcd_tile, int lanes_per_body);
/* Cold tier, TIME ONLY (the iterates are the same bits): where the body sweep of a tiered solve takes its drift bound
 * from -- 1: the difference of a body's new row and its row of the previous iterate (48 more bytes read per body: free
 * while both row tables sit in the Infinity Cache); 2: the change of the force accumulated in registers beside the
 * sums (no extra read; rods and spheres with the default lane layout only); 0 (default): by size -- rows up to
 * 1.75 * 10^6 bodies, registers beyond. */
int mhip_contact_op_set_drift_source(mhip_contact_op_t op, int source);
/* *source [host] = the form a tiered solve on this operator would take now (1 or 2) */
int mhip_contact_op_get_drift_source(mhip_contact_op_t op, int* source /*[host]*/);
/* Cold tier of mhip_bbpgd_solve_contact (time only, never results).  Two thirds of the contacts of a packing are inactive
 * (x = 0, g > 0) for most of a solve; such a contact adds exact zeros to every sum of an iteration, and how far its
 * gradient can have moved is bounded by its two bodies' accumulated velocity changes.  From the first convergence poll on
 * the solve therefore renumbers the contacts hot-first and leaves the cold tail alone (a body whose accumulated change
 * reaches the smallest bound of its sleeping contacts wakes those that have reached theirs; a woken contact is evaluated
 * like any other again).  Same
 * iterates, bit for bit, and the same iteration count as with tiering off (every sum is a double-double pair rounded
 * once, so the partition of the contacts does not reach the sums).  LCP solves (any of the three operator forms) with at
 * least 1.5 million contacts (below that the solve is launch-bound and the bookkeeping costs more than the shorter
 * sweep saves), in the fused solve and -- for the contacts between two bodies the rank owns -- in the staged /
 * distributed one; mode 0 = off, 1 = on (default), 2 = test hook: leave the tiers mid-solve, as the solve does before a BB
 * step outside [0, finite], 3 = test hook: no minimum size (2 has none either).
 * tier_stats: iterations that ran tiered, mean share of hot contacts over them, renumberings, contacts woken. */
int mhip_contact_op_set_tiering(mhip_contact_op_t handle, int mode);
int mhip_contact_op_tier_stats(mhip_contact_op_t handle, size_t* tiered_iterations, double* mean_hot_fraction,
                               size_t* renumberings, size_t* wakeups);
int mhip_contact_op_set_profiling(mhip_contact_op_t handle, int enable);
int mhip_contact_op_get_profile(mhip_contact_op_t handle, double* body_ms, double* constraint_ms, size_t* iterations);
/* body velocities [num_bodies][6] = (U xyz, W xyz) from the last apply / solve iterate (valid in stream order after
 * that call; for a partitioned operator only the owned rows are meaningful) */
int mhip_contact_op_body_velocity(mhip_contact_op_t handle, const double** velocity /*[host] out: device pointer*/);

/* ------------------------------------------------------------------------------------------------------------------
 * BBPGD drivers (seam S1): solve_cqpp / solve_lcp with PGDStrategy<BBStepStrategy, residual policy>.
 * Replaces: mundy/math/src/mundy_math/convex.hpp:592-681 (PGDStrategy::initialize/iterate/done/result), :789-845.
 * State vectors x, g, x_tmp, g_tmp are caller-owned (PGDState holds references, convex.hpp:551-585); on return they
 * hold what the reference leaves in them.  x is the initial guess.
 *   _dense   : A is a row-major n x n device matrix (the KokkosBlas::gemv path)
 *   _contact : A is a contact operator handle; the iteration is fused into 3 kernels per iteration with a
 *              device-resident step size and residual (no host round trip per reduction).
 * ---------------------------------------------------------------------------------------------------------------- */
int mhip_bbpgd_solve_dense(size_t n, const double* A, const double* q, const mhip_space* space /*[host]*/,
                           const mhip_pgd_config* config /*[host]*/, double* x, double* g, double* x_tmp,
                           double* g_tmp, mhip_solve_result* result /*[host]*/, mhip_stream_t stream);
int mhip_bbpgd_solve_contact(mhip_contact_op_t op, const double* q, const mhip_space* space /*[host]*/,
                             const mhip_pgd_config* config /*[host]*/, double* x, double* g, double* x_tmp,
                             double* g_tmp, mhip_solve_result* result /*[host]*/, mhip_stream_t stream);

/* BUILD EXTENSION, parity unpinned -- Coulomb friction (BASELINE configs[2] names a "frictional LCP"; the reference has
 * no frictional solver: SURVEY F2).  The same fused BBPGD iteration on the cone complementarity problem
 *   p_c in R^3 (world-frame contact impulse),  K_c = { |p - (p.n) n| <= mu (p.n) },
 *   g_c = dt [(U_j + W_j x rb) - (U_i + W_i x ra)] + sep_c n_c,   p in K, g in K*, p . g = 0
 * (the convex relaxation of Anitescu / Tasora used by APGD-type solvers; mu = 0 reduces to the frictionless LCP).
 * op must be the vector-arm operator (mhip_contact_op_create with ra, rb = centre -> CONTACT POINT ON THE SURFACE, and
 * mob_rot).  p [C][3] is the initial guess on entry (in K; zeros are fine) and the solution on return, g [C][3] its
 * gradient.  Residual: Linf projected difference over the 3C components (config->residual_kind must say so). */
int mhip_bbpgd_solve_contact_friction(mhip_contact_op_t op, const double* sep, double mu,
                                      const mhip_pgd_config* config /*[host]*/, double* p, double* g,
                                      mhip_solve_result* result /*[host]*/, mhip_stream_t stream);
/* The same problem, arguments, result and stopping rule by APGD -- Nesterov-accelerated projected gradient descent with
 * adaptive restart and a backtracked curvature estimate (Mazhar, Heyn, Negrut, Tasora 2015: the algorithm BASELINE's
 * north star names, and the paper mundy_math/convex.hpp:476 cites), arranged so that a sweep is ONE operator application
 * (the gradient at the extrapolated point is the same extrapolation of the two stored gradients).  num_iters counts
 * every sweep, refused steps included.  BUILD EXTENSION like the BBPGD form: the reference has no frictional solver. */
int mhip_apgd_solve_contact_friction(mhip_contact_op_t op, const double* sep, double mu,
                                      const mhip_pgd_config* config /*[host]*/, double* p, double* g,
                                      mhip_solve_result* result /*[host]*/, mhip_stream_t stream);
/* Staged form of the same fused iteration for domain-decomposed runs: the host interleaves the halo
 * exchange and the cross-rank reduction between the stages, all asynchronously on `stream`:
 *     begin;  body(init) -> [ghost velocity halo] -> constraint(init, local) -> [all-gather local] -> finalize(init)
 *     repeat: body -> [halo] -> constraint(local) -> [all-gather] -> finalize;   poll every k iterations;   end
 * mhip_contact_op_set_partition: bodies [first, first+count) of the local index space are owned (swept by the body
 * stage), the rest are ghosts whose rows of `velocity` ([num_bodies][6], caller owned so the halo can write it) are
 * filled by the exchange; counted_contacts[c] != 0 marks contacts this rank contributes to the reductions (NULL: all).
 * local / gathered are DEVICE arrays of reduction records of MHIP_BBPGD_REDUCTION_WIDTH doubles:
 *     (max residual term, sum dx^2 as a double-double pair hi, lo, sum dx dg as a double-double pair hi, lo).
 * The two Barzilai-Borwein sums travel unrounded (~106 bits) and finalize adds the `nparts` gathered records in order
 * and rounds once, so every rank derives bit-identical step sizes and the rank count does not reach the result (the
 * same holds inside a rank for the workgroup mapping and, in the body sweep, for the order of a body's contact list).
 * Replaces: stk::all_reduce_max / stk::all_reduce_sum x3 per iteration (scrap/lcp_spheres/NGPSpheresLCP.cpp:371,
 * :450-452) with one 5-double all-gather, and the ghost field refresh of :1057 (left "TODO" there). */
#define MHIP_BBPGD_REDUCTION_WIDTH 5
int mhip_contact_op_set_partition(mhip_contact_op_t handle, size_t body_first, size_t body_count,
                                  const unsigned char* counted_contacts, double* velocity);
int mhip_bbpgd_stage_begin(mhip_contact_op_t op, const double* q, const mhip_space* space /*[host]*/,
                           const mhip_pgd_config* config /*[host]*/, double* x, double* g, double* x_tmp,
                           double* g_tmp, mhip_stream_t stream);
int mhip_bbpgd_stage_body(mhip_contact_op_t op, int init, mhip_stream_t stream);
int mhip_bbpgd_stage_constraint(mhip_contact_op_t op, int init, double* local, mhip_stream_t stream);
/* the constraint stage in pieces: sweep a sub-range of the constraints (any number of disjoint ranges per iteration,
 * e.g. interior contacts before the halo has landed, boundary contacts after), then reduce every sweep's partials of
 * this iteration into local.  stage_constraint == constraint_range(0, C) + reduce. */
int mhip_bbpgd_stage_constraint_range(mhip_contact_op_t op, int init, size_t c_first, size_t c_count,
                                      mhip_stream_t stream);
int mhip_bbpgd_stage_reduce(mhip_contact_op_t op, int init, double* local, mhip_stream_t stream);
int mhip_bbpgd_stage_finalize(mhip_contact_op_t op, int init, const double* gathered, int nparts,
                              mhip_stream_t stream);
int mhip_bbpgd_stage_poll(mhip_contact_op_t op, mhip_solve_result* result /*[host]*/, int* done /*[host]*/,
                          mhip_stream_t stream); /* synchronises */
/* between two iterations (e.g. right after a poll): copies the entries the body sweep's activity masks flag, and their
 * records, into compact per-body lists which the next sweeps stream instead of picking them out of the full lists
 * (time only: the same terms are summed, and every sum is rounded once).  Optional; the fused and distributed drivers
 * call it at their polls. */
int mhip_bbpgd_stage_snapshot_active(mhip_contact_op_t op, mhip_stream_t stream);
int mhip_bbpgd_stage_end(mhip_contact_op_t op, mhip_solve_result* result /*[host]*/, mhip_stream_t stream);
/* In-kernel small problems: convex::MundyMathBackend<Scalar, N> (convex.hpp:288-350) runs the same solver
 * on fixed-size Vector/Matrix inside a kernel; here one thread solves one dense n x n problem (1 <= n <= 16) of a
 * batch: A [batch][n][n] row major, q [batch][n], x [batch][n] (initial guess in, solution out), grad [batch][n],
 * per-problem num_iters / residual / converged (all device arrays).  Right-fold dot products and no |alpha|,|beta|
 * fast paths, exactly as that backend evaluates them. */
int mhip_solve_small_cqpp_batch(size_t batch, int n, const double* A, const double* q,
                                const mhip_space* space /*[host]*/, const mhip_pgd_config* config /*[host]*/,
                                double* x, double* grad, unsigned* num_iters, double* residual, int* converged,
                                mhip_stream_t stream);
/* The scrap app's own solver variant: resolve_collisions, scrap/lcp_spheres/NgpLcp.cpp:558-759 --
 * Dai-Fletcher residual |min(g,0)| / |g| with the 1e-12 active-set test (:376-405), strict `< max_allowable_overlap`
 * convergence, first step 1/residual, BB1/BB2 alternating by the parity of ite_count with the `|b| < 1e-12` guard
 * (:716-731), ite_count counting started iterations.  lam is the initial guess (in) and the multipliers (out);
 * lam_tmp, g, g_tmp are work vectors [C] (g = sep + dt*sep_dot on return).  *max_speed [host] receives max|U| over
 * bodies (ComputeMaxVelocity :743-755; CollisionResult.max_displacement = max_speed * dt).  result->num_iters = ite_count,
 * result->residual = max_abs_projected_sep. */
int mhip_scrap_bbpgd_solve_contact(mhip_contact_op_t op, const double* sep, double max_allowable_overlap,
                                   unsigned max_iterations, double* lam, double* lam_tmp, double* g, double* g_tmp,
                                   mhip_solve_result* result /*[host]*/, double* max_speed /*[host]*/,
                                   mhip_stream_t stream);
/* Same algorithm driven kernel-by-kernel through the S1 vector entry points above (what the C++ adapter's
 * HipBackend does): the unfused reference structure, kept as an in-library cross-check of the fused path. */
int mhip_bbpgd_solve_contact_unfused(mhip_contact_op_t op, const double* q, const mhip_space* space /*[host]*/,
                                     const mhip_pgd_config* config /*[host]*/, double* x, double* g, double* x_tmp,
                                     double* g_tmp, mhip_solve_result* result /*[host]*/, mhip_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Stream compaction helpers (filter_view, mundy/mesh/src/mundy_mesh/GenNeighborLinkers.hpp:141-183: exclusive scan +
 * stable scatter; the total is read back on the host exactly as :155-156 does).
 *   filter_pairs_owned : keep pairs with at least one endpoint in [first, first+count) (drops ghost-ghost pairs);
 *                        counted_out[k] = 1 iff the kept pair's source (lower index) is owned.  pairs_out may alias
 *                        neither input.  *count_out [host].
 *   select_aabb_overlap: indices (ascending) of the boxes aabb[i] grown by `buffer` that intersect (closed test) the
 *                        box `box6` [host: min xyz, max xyz] -- the ghost candidates for a neighbouring rank.
 *   select_contacts    : BUILD OPTION (off by default: the reference's LCP app makes every neighbour pair a constraint,
 *                        scrap/lcp_spheres/NgpLcp.cpp:346-373).  kept_index [<= c] = ascending indices of the candidate
 *                        pairs whose signed separation is not above `cutoff` (NaN kept), by wavefront ballot + prefix
 *                        sum; *count_out [host].  The caller gathers pairs / normals / arms with mhip_gather_rows and
 *                        builds the operator on the kept contacts only: fewer constraints per sweep.  A dropped pair
 *                        is a constraint that must stay inactive: check g = sep + dt sdot >= 0 on the dropped pairs
 *                        afterwards (the steppers do, and fall back to the full list when it fails).
 * ---------------------------------------------------------------------------------------------------------------- */
int mhip_select_contacts(size_t c, const double* sep, double cutoff, int32_t* kept_index, size_t* count_out /*[host]*/,
                         mhip_stream_t stream);
int mhip_filter_pairs_owned(size_t c, const int32_t* pairs_in, size_t first, size_t count, int32_t* pairs_out,
                            unsigned char* counted_out, size_t* count_out /*[host]*/, mhip_stream_t stream);
/* Same selection, ordered for overlap: first the pairs with BOTH bodies owned (interior), then those with exactly one
 * (boundary), each block in input order.  interior_out = size of the first block, count_out = total.  The staged solver
 * sweeps [0, interior) while the ghost-velocity halo that the boundary block needs is still in flight. */
int mhip_partition_pairs_owned(size_t c, const int32_t* pairs_in, size_t first, size_t count, int32_t* pairs_out,
                               unsigned char* counted_out, size_t* interior_out /*[host]*/,
                               size_t* count_out /*[host]*/, mhip_stream_t stream);
int mhip_select_aabb_overlap(size_t n, const double* aabb, double buffer, const double* box6 /*[host]*/,
                             int32_t* idx_out, size_t* count_out /*[host]*/, mhip_stream_t stream);
/* A rank's region as several boxes: bodies [k * ceil(n / nchunks), ...) form chunk k (the owned bodies are in curve
 * order, so a chunk is a compact blob); boxes [device, (nchunks + 1)][6] receives the bounds of each chunk's boxes grown
 * by `buffer` and, last, their union.  select_aabb_overlap_any keeps the bodies whose grown box meets ANY of the first
 * nboxes boxes (boxes [device, (nboxes + 1)][6], the last one the union, used as a quick reject) -- the ghost
 * candidates for a rank described that way; ascending order.  No host round trip in chunk_bounds. */
int mhip_aabb_chunk_bounds(size_t n, const double* aabb, double buffer, int nchunks, double* boxes,
                           mhip_stream_t stream);
int mhip_select_aabb_overlap_any(size_t n, const double* aabb, double buffer, int nboxes, const double* boxes,
                                 int32_t* idx_out, size_t* count_out /*[host]*/, mhip_stream_t stream);
/* min / max corner over n boxes grown by `buffer`: out6 [host] (an empty set gives the inverted box) */
int mhip_aabb_bounds(size_t n, const double* aabb, double buffer, double* out6 /*[host]*/, mhip_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Time integration either side of the solve: x += dt * U (scrap/lcp_spheres/NgpLcp.cpp:898) and, for
 * rods, q <- rotate_quaternion(q, W, dt) (mundy/math/src/mundy_math/Quaternion.hpp:1366-1390).
 * velocity is [n][6] as produced by the contact operator.  quat may be NULL (spheres).
 * ---------------------------------------------------------------------------------------------------------------- */
int mhip_integrate_euler(size_t n, double dt, const double* velocity, double* center, double* quat,
                         mhip_stream_t stream);

/* Periodic box: PeriodicScaledMetric::sep (minimum-image separation p1 -> p2) and wrap_rigid of a
 * Sphere / Spherocylinder / Ellipsoid (the centre is wrapped into [0, box), orientation and size untouched).
 * Replaces: mundy/geom/src/mundy_geom/periodicity.hpp:812-823, :1088-1113, :1156-1160.  box [host] 3 doubles. */
int mhip_periodic_sep(size_t n, const double* box, const double* p1, const double* p2, double* out,
                      mhip_stream_t stream);
int mhip_wrap_rigid(size_t n, const double* box, double* center, mhip_stream_t stream);

/* Triclinic cell: PeriodicMetric (periodicity.hpp:233-332).  cell [host] = 3x3 unit-cell matrix, row
 * major, lattice vectors as COLUMNS; its inverse is math::inverse (Matrix.hpp:1596-1601: adjugate / determinant).
 *   mhip_unit_cell_inverse        host-only: that inverse, for callers that want fractional coordinates
 *   mhip_periodic_sep_triclinic   sep(p1, p2) = h * min_image(h_inv * (p2 - p1))          (:304-307)
 *   mhip_wrap_rigid_triclinic     centre <- h * safe_unit_mod1(h_inv * centre)             (:312-314, :1088-1113)
 *   mhip_shift_image_triclinic    p + h * images, images [n][3] int32                      (:323-327)
 *   mhip_contact_spheres_triclinic  mhip_contact_spheres with this metric's minimum image */
int mhip_unit_cell_inverse(const double* cell, double* cell_inv);
int mhip_periodic_sep_triclinic(size_t n, const double* cell, const double* p1, const double* p2, double* out,
                                mhip_stream_t stream);
int mhip_wrap_rigid_triclinic(size_t n, const double* cell, double* center, mhip_stream_t stream);
int mhip_shift_image_triclinic(size_t n, const double* cell, const double* p, const int32_t* images, double* out,
                               mhip_stream_t stream);
int mhip_contact_spheres_triclinic(size_t c, const int32_t* pairs, const double* center, const double* radius,
                                   const double* cell, double* sep, double* normal, mhip_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Body reordering: Z-order (Morton) permutation of bodies by centre, z most significant as
 * zorder_knn::Less (mundy/math/src/mundy_math/zmort.hpp:195-220) orders non-negative lattice coordinates.
 * perm[k] = index of the body that goes to position k.  Deterministic (ties broken by index).
 * The lattice has 2^b cells per axis, b = the largest of 4..8 with 8^b <= 8 n (the code space stays within 8 codes per
 * body: 128 cells per axis at 10^6 bodies); lattice coordinates floor((c - lo) / cell_size) are clamped to
 * [0, 2^b - 1], so bodies outside that cube share the boundary cells (a locality heuristic there, exact Z-order inside).
 * ---------------------------------------------------------------------------------------------------------------- */
int mhip_morton_order(size_t n, const double* center, const double* lo /*[host] 3*/, double cell_size, int32_t* perm,
                      mhip_stream_t stream);
/* The same reordering along any lattice curve given as a table: key_table [device, (2^level)^3 int32, indexed
 * [ix][iy][iz]] = visiting index of the cell.  With the table of mundy::math::hilbert_3d (mundy/math/src/mundy_math/
 * Hilbert.hpp:48-83; mundy_amd.distributed.hilbert_key_table generates it with that recursion) this is the Hilbert
 * order used for the domain decomposition.  Cells: floor((c - lo) / (hi - lo) * 2^level) clamped to the lattice;
 * ties by index.  level <= 8. */
int mhip_curve_order(size_t n, const double* center, const double* lo /*[host] 3*/, const double* hi /*[host] 3*/,
                     int level, const int32_t* key_table, int32_t* perm, mhip_stream_t stream);
/* keys[i] = key_table entry of body i's cell (the value mhip_curve_order sorts by): the Hilbert position of the cell --
 * what the work-weighted domain decomposition cuts and what decides a body's owner when it migrates */
int mhip_curve_keys(size_t n, const double* center, const double* lo /*[host] 3*/, const double* hi /*[host] 3*/,
                    int level, const int32_t* key_table, uint32_t* keys, mhip_stream_t stream);
/* perm = the stable ascending order of n 64-bit keys (radix sort, 8 bits per pass): perm[k] = index of the k-th key */
int mhip_sort_by_key_u64(size_t n, const uint64_t* keys, int32_t* perm, mhip_stream_t stream);
/* dst[k][0..width) = src[perm[k]][0..width)  for rows of `width` doubles */
int mhip_gather_rows(size_t n, size_t width, const int32_t* perm, const double* src, double* dst, mhip_stream_t stream);
/* dst[k * dst_stride + c] = src[k * src_stride + c], c < width (strides in doubles): packs per-body fields into
 * interleaved records for the ghost exchange and unpacks them again */
int mhip_copy_strided(size_t n, size_t width, const double* src, size_t src_stride, double* dst, size_t dst_stride,
                      mhip_stream_t stream);


/* ----------------------------------------------------------------------------------------------------------------
 * Multi-GPU transport and the domain-decomposed solve driven from C++.
 *
 * Replaces, on the reference side: the MPI communicator handed to stk::search::coarse_search and change_ghosting
 * (mundy/mesh/src/mundy_mesh/GenNeighborLinkers.hpp:658, :687-711), and per BBPGD iteration stk::all_reduce_max +
 * 3 x stk::all_reduce_sum (scrap/lcp_spheres/NGPSpheresLCP.cpp:371, :450-452) and the ghost field refresh left as a
 * TODO at :1057.
 *
 * A communicator is one rank's end of a group of `world` ranks, one rank per GPU.  Two transports:
 *   - RCCL (mhip_comm_create_rccl): grouped ncclSend / ncclRecv over xGMI, issued on a stream the communicator owns
 *     and ordered against the caller's stream with events, so an exchange started after one kernel overlaps the
 *     kernels launched before mhip_comm_exchange_finish; ncclAllGather goes straight onto the caller's stream (it sits
 *     on the critical path of every solver iteration).  A message addressed to oneself is delivered as a local copy
 *     (the host transport rejects it).  librccl is looked up at run time (dlopen of
 *     the already loaded librccl.so.1 or the one on the library path): no link dependency.  The 128-byte unique id
 *     is made by rank 0 (mhip_comm_unique_id) and handed to the other ranks by the launcher (MPI_Bcast in an MPI
 *     host, the torch.distributed store in bench.py).
 *   - host callbacks (mhip_comm_create_host): the same calls forwarded to two functions of the host program after a
 *     stream synchronisation (device pointers are passed through).  For hosts that bring their own message layer and
 *     for tests where several ranks share one GPU, which RCCL refuses.
 * All buffers are DEVICE pointers to doubles; counts are in doubles.  Handles are not thread-safe.
 * ---------------------------------------------------------------------------------------------------------------- */
typedef struct mhip_comm* mhip_comm_t;
#define MHIP_COMM_ID_BYTES 128
typedef int (*mhip_comm_exchange_fn)(void* user, int nsend, const int* send_peer, const double* const* send_buf,
                                     const size_t* send_count, int nrecv, const int* recv_peer, double* const* recv_buf,
                                     const size_t* recv_count); /* returns 0 when every message has landed */
typedef int (*mhip_comm_all_gather_fn)(void* user, const double* send, size_t count, double* recv /*[world][count]*/);

int mhip_comm_unique_id(unsigned char* id /*[host] MHIP_COMM_ID_BYTES*/);
int mhip_comm_create_rccl(mhip_comm_t* comm, const unsigned char* id /*[host]*/, int rank, int world);
int mhip_comm_create_host(mhip_comm_t* comm, int rank, int world, mhip_comm_exchange_fn exchange,
                          mhip_comm_all_gather_fn all_gather, void* user);
int mhip_comm_destroy(mhip_comm_t comm);
int mhip_comm_info(mhip_comm_t comm, int* rank, int* world, int* is_rccl);
/* Mailbox (ranks of ONE node): the 5-double reduction record every rank contributes to every BBPGD iteration
 * (NGPSpheresLCP.cpp:371, :450-452: 1 all_reduce_max + 3 all_reduce_sum there) travels through slots in the ranks'
 * device memory instead of a collective launch: every rank owns a box of fine-grained memory that all the others map
 * (hipIpc handles, exchanged through the communicator's own all-gather); a rank writes its record into its slot of
 * every box -- posted writes over xGMI -- and polls its own box, all inside the kernel that forms the record.  Every
 * 8-byte word carries 32 bits of data and the number of the exchange, so each store validates itself and the writes
 * may arrive in any order.  Collective call.  *opened = 1: every rank has mapped every box and two trial exchanges went
 * through everywhere (mhip_bbpgd_solve_contact_distributed then uses it); 0: somebody could not (another node, no peer
 * access) and everybody keeps the transport's all-gather.  A record that does not arrive within 20 s ends the solve in
 * MHIP_ERR_RUNTIME on the ranks that waited for it (bounded polling: no wave waits forever). */
int mhip_comm_mailbox_open(mhip_comm_t comm, int* opened /*[host]*/, mhip_stream_t stream);
int mhip_comm_mailbox_close(mhip_comm_t comm);
/* The per-iteration VELOCITY HALO of mhip_bbpgd_solve_contact_distributed without the collective library, for the
 * ranks of ONE node (same condition as the mailbox).  on = 1: from the next mhip_ghost_plan on (collective: every rank
 * must have made the same call) each rank keeps an inbox of fine-grained device memory, one slot per local body row,
 * IPC-mapped by all the others.  After its body sweep a rank writes the rows its peers hold as ghosts straight into
 * their inboxes (posted writes over xGMI; every 8-byte word carries 32 bits of data and the number of the exchange, so
 * each store validates itself); before its boundary sweep a rank collects its ghost rows from its own inbox into the
 * velocity table (bounded polling: a peer that never writes ends in MHIP_ERR_RUNTIME).  Replaces the gather kernel,
 * the grouped ncclSend / ncclRecv and two stream hand-overs per iteration by two launches on the solver's own stream.
 * Anything short of success on every rank (allocation, mapping, probing) leaves everybody on send / recv.  The solve
 * takes this path only for the halo of the communicator's current ghost plan; same rows, same bits.
 * Replaces: communicate_field_data (scrap/.../SpherocylinderSpherocylinderLinker.cpp:154-155) inside the iteration,
 * the ghost refresh the reference leaves as a TODO (NGPSpheresLCP.cpp:1057). */
int mhip_comm_halo_ipc_enable(mhip_comm_t comm, int on);
/* Bound of every wait on a peer's words (mailbox records, inbox rows) in seconds; default 20.  A wait that exceeds it
 * ends the solve with MHIP_ERR_RUNTIME on that rank -- no wave waits forever.  After ANY failed distributed solve the
 * ranks may disagree on the exchange numbers: the next mhip_ghost_plan (collective) re-agrees them (the maximum over the
 * ranks) and clears the sticky error flag, so the recovery is "rebuild, then solve again". */
int mhip_comm_set_exchange_timeout(mhip_comm_t comm, double seconds);
/* TEST HOOK: the next mhip_bbpgd_solve_contact_distributed on this communicator returns MHIP_ERR_RUNTIME at its
 * at_poll-th convergence poll (1 = the first), after the iterations enqueued until then; 0 = off.  One shot.  Exists so
 * that the tests can show a failed solve leaves the communicator usable (exchange numbers retired on every exit). */
int mhip_comm_inject_fault(mhip_comm_t comm, unsigned at_poll);
/* *active [host] = 1 when the current ghost plan's halo will travel through the inboxes */
int mhip_comm_halo_ipc_active(mhip_comm_t comm, int* active /*[host]*/);
/* recv[r][0..count) = rank r's send[0..count); later work on `stream` sees recv */
int mhip_comm_all_gather(mhip_comm_t comm, const double* send, size_t count, double* recv, mhip_stream_t stream);
/* One grouped point-to-point exchange: message k goes to send_peer[k] / comes from recv_peer[k]; empty messages are
 * skipped on both sides.  start: the transfers begin once the work already on `stream` is done; finish: later work on
 * `stream` waits for them.  The peer / pointer / count arrays are host arrays and are read during start only. */
int mhip_comm_exchange_start(mhip_comm_t comm, int nsend, const int* send_peer, const double* const* send_buf,
                             const size_t* send_count, int nrecv, const int* recv_peer, double* const* recv_buf,
                             const size_t* recv_count, mhip_stream_t stream);
int mhip_comm_exchange_finish(mhip_comm_t comm, mhip_stream_t stream);

/* Ghost-velocity halo of one rank, fixed between neighbour-list rebuilds.  velocity = the [num_local_bodies][6] array
 * given to mhip_contact_op_set_partition.  Rows send_index[...] (DEVICE, local body indices, peer after peer in the
 * order of send_peer, send_rows[k] of them for peer k) go out each iteration; rows [recv_first_row[k],
 * recv_first_row[k] + recv_rows[k]) are filled from recv_peer[k]. */
typedef struct mhip_velocity_halo {
  double* velocity;
  int num_send_peers;
  const int* send_peer;       /* [host] */
  const size_t* send_rows;    /* [host] */
  const int32_t* send_index;  /* [device] sum(send_rows) */
  int num_recv_peers;
  const int* recv_peer;            /* [host] */
  const size_t* recv_first_row;    /* [host] */
  const size_t* recv_rows;         /* [host] */
} mhip_velocity_halo;
/* Ghost bodies of one neighbour-list rebuild (replaces coarse_search(comm) + change_ghosting,
 * GenNeighborLinkers.hpp:658, :687-711).  Ranks own contiguous, increasing ranges of the global (curve-ordered) ids.
 *   plan:     all-gather of the rank regions (64 chunk boxes per rank, mhip_aabb_chunk_bounds), per peer the owned bodies
 *             whose grown box meets any of the peer's boxes (closed test, ascending order), all-gather of the count
 *             matrix.  The owned bodies should be in curve order (any order is correct, only looser).  Fills `layout`:
 *             local index order = ghosts of lower ranks (peer order), the n owned bodies, ghosts of higher ranks --
 *             also global-id order, so local pairs (i < j) keep their global orientation; and the velocity halo of
 *             this layout (lists owned by the communicator, valid until the next plan; halo.velocity is left NULL).
 *   exchange: moves rows of `width` doubles with that plan: local[num_ghost_lo + i] = records[i] for the owned rows,
 *             ghost rows filled from their owners.  Any number of record sets per plan. */
typedef struct mhip_ghost_layout {
  size_t num_ghost_lo, num_owned, num_ghost_hi;
  size_t num_sent; /* owned bodies that are ghosts somewhere, counted once per receiving rank */
  mhip_velocity_halo halo;
} mhip_ghost_layout;
/* The bookkeeping of the plan on its own (host arithmetic only, no device): counts[s * world + d] = bodies rank s sends
 * to rank d.  Peer lists are filled in ascending rank order, empty messages left out; the lists must hold `world`
 * entries.  recv_first_row = first local row of each peer's ghosts. */
int mhip_ghost_layout_from_counts(int world, int rank, size_t n_owned, const size_t* counts /*[host]*/,
                                  size_t* num_ghost_lo, size_t* num_ghost_hi, int* num_send, int* send_peer,
                                  size_t* send_rows, int* num_recv, int* recv_peer, size_t* recv_first_row,
                                  size_t* recv_rows);
int mhip_ghost_plan(mhip_comm_t comm, size_t n, const double* aabb, double buffer,
                    mhip_ghost_layout* layout /*[host] out*/, mhip_stream_t stream);
int mhip_ghost_exchange(mhip_comm_t comm, size_t width, const double* records /*[n][width]*/,
                        double* local /*[num_ghost_lo + n + num_ghost_hi][width]*/, mhip_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Ownership follows the bodies (replaces the RCB repartition of stk::balance::balanceStkMesh,
 * scrap/lcp_spheres/NGPSpheresLCP.cpp:956, which the bacteria app calls every load_balance_frequency steps,
 * scrap/parameter_interface/alens/tests/performance_tests/Bacteria.cpp:1076-1078).  Ranks own contiguous ranges of
 * the Hilbert curve over a fixed lattice: rank r owns the cells  splitters[r - 1] < key <= splitters[r].
 *   mhip_hilbert_key_table  [host] table[ix][iy][iz] = position of the cell along mundy::math::hilbert_3d
 *                           (mundy/math/src/mundy_math/Hilbert.hpp:48-83); (2^level)^3 entries, level <= 8; feed it to
 *                           mhip_curve_keys / mhip_curve_order
 *   mhip_curve_cut          cuts the curve at equal cumulative WORK: weight per body (null = 1; the steppers use
 *                           1 + contacts of the last step) histogrammed over the cells, all-gathered, summed in rank
 *                           order, cut where the running sum reaches r / world of the total.  Every rank computes the
 *                           same world - 1 splitters [host].  Collective.
 *   mhip_migrate_plan       destination rank of every owned body from its cell key; counts all-gathered.  n_new =
 *                           bodies this rank owns afterwards.  Collective.
 *   mhip_migrate_exchange   records [n][width] -> out [n_new][width]: the rows that stay (in their present order),
 *                           then the arrivals, peers in increasing rank; one grouped send / recv per peer.  Any number
 *                           of exchanges may follow one plan.  The caller then orders its rows as it likes (the
 *                           steppers: by cell key, ties by entity id -- the order a single rank would have).
 * Small helpers of that bookkeeping:
 *   mhip_body_work_weights  weights[k] = 1 + number of pairs that contain local body first + k, k < count
 *   mhip_compose_keys_u64   out[i] = major[i] << shift | (uint64) minor[i]  (cell key, then entity id: minor is an
 *                           integer-valued double below 2^shift)
 *   mhip_fill_sequence      dst[i] = first + i
 * ---------------------------------------------------------------------------------------------------------------- */
int mhip_hilbert_key_table(int level, int32_t* table /*[host] (2^level)^3*/);
int mhip_body_work_weights(size_t c, const int32_t* pairs, size_t first, size_t count, double* weights /*[count]*/,
                           mhip_stream_t stream);
int mhip_compose_keys_u64(size_t n, const uint32_t* major, const double* minor, int shift, uint64_t* out,
                          mhip_stream_t stream);
int mhip_fill_sequence(size_t n, double first, double* dst, mhip_stream_t stream);
int mhip_curve_cut(mhip_comm_t comm, size_t n, const uint32_t* keys, const double* weights /*[n] or null*/, size_t ncell,
                   int64_t* splitters /*[host] world - 1*/, mhip_stream_t stream);
int mhip_migrate_plan(mhip_comm_t comm, size_t n, const uint32_t* keys, const int64_t* splitters /*[host] world - 1*/,
                      size_t* n_new, size_t* num_sent /*or null*/, size_t* num_received /*or null*/,
                      mhip_stream_t stream);
int mhip_migrate_exchange(mhip_comm_t comm, size_t width, const double* records /*[n][width]*/,
                          double* out /*[n_new][width]*/, mhip_stream_t stream);
typedef struct mhip_dist_profile { /* HIP-event times of sampled iterations that did work, summed over them */
  double body_ms;        /* body sweep */
  double constraint_ms;  /* interior + boundary constraint sweeps (the wait for the halo excluded) */
  double halo_wait_ms;   /* between the interior and the boundary sweep: collect from the inbox / finish of send-recv */
  size_t timed_iterations;
  double halo_post_ms;   /* after the body sweep: the push into the peers' inboxes / gather + start of send-recv */
  double record_ms;      /* after the boundary sweep: this rank's record, its exchange (the wait for the slowest rank
                            included) and the finalize */
  int halo_path;         /* what this solve ACTUALLY used: 0 no halo (one rank / no ghosts), 1 IPC-mapped inboxes,
                            2 grouped send / recv of the transport */
  int record_path;       /* 1 mailbox inside the fold-finalize launch, 2 mailbox in the record's launch (world > the
                            fused limit), 3 all-gather of the transport */
} mhip_dist_profile;
/* The domain-decomposed BBPGD solve, whole loop on the host side of this library (no interpreter between the stages):
 *   begin;  per iteration: body sweep of the owned bodies -> pack + start the velocity halo -> constraint sweep of the
 *   interior contacts [0, interior_contacts) while the halo is in flight -> finish halo -> sweep of the boundary
 *   contacts -> local (max, sum dx^2, sum dx dg) record -> all-gather -> every rank reduces the records in rank order
 *   (bit-identical steps) ;  convergence polled after stretches of 8, 16, 32, ... iterations, at most poll_every (0: 64) ;  end.
 * Same iterates as the mhip_bbpgd_stage_* sequence driven by hand.  op must carry mhip_contact_op_set_partition; the
 * four solver vectors are caller owned as in mhip_bbpgd_solve_contact.  profile may be NULL. */
int mhip_bbpgd_solve_contact_distributed(mhip_contact_op_t op, mhip_comm_t comm, const mhip_velocity_halo* halo,
                                         size_t interior_contacts, const double* q, const mhip_space* space /*[host]*/,
                                         const mhip_pgd_config* config /*[host]*/, double* x, double* g, double* x_tmp,
                                         double* g_tmp, unsigned poll_every, mhip_solve_result* result /*[host]*/,
                                         mhip_dist_profile* profile /*[host]*/, mhip_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MUNDY_HIP_H_ */
