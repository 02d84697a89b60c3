"""Host-side mirror of the reference's hot-path interface over libmundy_hip.so, on torch CUDA(HIP) tensors.

Names follow the reference: compute_aabb / distance / GenNeighborLinks (mundy_mesh/GenNeighborLinkers.hpp) /
solve_cqpp, solve_lcp, PGDConfig, SolveResult (mundy_math/convex.hpp).  torch is plumbing (device memory + the
current stream); all arithmetic runs in the HIP library.  float64 everywhere; pairs are int32 [C, 2].
"""
import ctypes as C
from dataclasses import dataclass

import torch

from . import capi
from .capi import (RESIDUAL_PROJECTED_DIFF, RESIDUAL_PROJECTED_GRADIENT, SEARCH_AABB, SEARCH_METHOD_AUTO,  # noqa: F401
                   SEARCH_METHOD_GRID, SEARCH_METHOD_MORTON_LBVH, SEARCH_SPHERES, SPACE_BOUNDED, SPACE_LOWER_BOUND, SPACE_UNCONSTRAINED, SPACE_UPPER_BOUND)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t, dtype=torch.float64, cols=None, name="tensor", allow_none=False):
    if t is None:
        if allow_none:
            return None
        raise ValueError("%s must not be None" % name)
    if not t.is_cuda:
        raise ValueError("%s must live on the GPU (mundy_amd has no CPU path)" % name)
    if t.dtype != dtype:
        raise ValueError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous" % name)
    if cols is not None and (t.dim() != 2 or t.shape[1] != cols):
        raise ValueError("%s must have shape [n, %d], got %s" % (name, cols, tuple(t.shape)))
    return C.c_void_p(t.data_ptr())


def _new(ref, *shape, dtype=torch.float64):
    return torch.empty(shape, dtype=dtype, device=ref.device)


def device_info():
    n = C.c_int(0)
    name = C.create_string_buffer(256)
    capi.check(capi.load().mhip_device_info(C.byref(n), name, 256))
    return n.value, name.value.decode()


# ---- per-body geometry (compute_aabb.hpp / compute_bounding_radius.hpp) --------------------------------------------
def compute_aabb_spheres(center, radius):
    n = radius.shape[0]
    out = _new(center, n, 6)
    capi.check(capi.load().mhip_compute_aabb_spheres(n, _ptr(center, cols=3), _ptr(radius), _ptr(out), _stream()))
    return out


def compute_aabb_spherocylinders(center, quat, radius, length):
    n = radius.shape[0]
    out = _new(center, n, 6)
    capi.check(capi.load().mhip_compute_aabb_spherocylinders(n, _ptr(center, cols=3), _ptr(quat, cols=4),
                                                             _ptr(radius), _ptr(length), _ptr(out), _stream()))
    return out


def compute_aabb_ellipsoids(center, quat, radii):
    n = center.shape[0]
    out = _new(center, n, 6)
    capi.check(capi.load().mhip_compute_aabb_ellipsoids(n, _ptr(center, cols=3), _ptr(quat, cols=4),
                                                        _ptr(radii, cols=3), _ptr(out), _stream()))
    return out


def compute_aabb_ellipsoids_conservative(center, quat, radii):
    """build extension: tight box of the rotated ellipsoid (the reference's box is not conservative)"""
    n = center.shape[0]
    out = _new(center, n, 6)
    capi.check(capi.load().mhip_compute_aabb_ellipsoids_conservative(n, _ptr(center, cols=3), _ptr(quat, cols=4),
                                                                     _ptr(radii, cols=3), _ptr(out), _stream()))
    return out


def compute_aabb_segments(seg):
    n = seg.shape[0]
    out = _new(seg, n, 6)
    capi.check(capi.load().mhip_compute_aabb_segments(n, _ptr(seg, cols=8), _ptr(out), _stream()))
    return out


def bounding_radius_spherocylinders(radius, length):
    out = torch.empty_like(radius)
    capi.check(capi.load().mhip_bounding_radius_spherocylinders(radius.shape[0], _ptr(radius), _ptr(length),
                                                                _ptr(out), _stream()))
    return out


def bounding_radius_ellipsoids(radii):
    out = _new(radii, radii.shape[0])
    capi.check(capi.load().mhip_bounding_radius_ellipsoids(radii.shape[0], _ptr(radii, cols=3), _ptr(out), _stream()))
    return out


def spherocylinder_segments(center, quat, radius, length, out=None):
    n = radius.shape[0]
    seg = _new(center, n, 8) if out is None else out
    capi.check(capi.load().mhip_spherocylinder_segments(n, _ptr(center, cols=3), _ptr(quat, cols=4), _ptr(radius),
                                                        _ptr(length), _ptr(seg, cols=8), _stream()))
    return seg


# ---- distances (mundy_geom/distance/*.hpp) ----------------------------------------------------------------------------
def distance_sphere_sphere(c1, r1, c2, r2):
    n = r1.shape[0]
    dist, sep = _new(c1, n), _new(c1, n, 3)
    capi.check(capi.load().mhip_distance_sphere_sphere(n, _ptr(c1, cols=3), _ptr(r1), _ptr(c2, cols=3), _ptr(r2),
                                                       _ptr(dist), _ptr(sep), _stream()))
    return dist, sep


def distance_point_segment(p, a0, a1):
    n = p.shape[0]
    dist, cp, t, sep = _new(p, n), _new(p, n, 3), _new(p, n), _new(p, n, 3)
    capi.check(capi.load().mhip_distance_point_segment(n, _ptr(p, cols=3), _ptr(a0, cols=3), _ptr(a1, cols=3),
                                                       _ptr(dist), _ptr(cp), _ptr(t), _ptr(sep), _stream()))
    return dist, cp, t, sep


def distance_point_sphere(p, c, r):
    """distance(Point, Sphere, sep) (PointSphere.hpp:69-79)."""
    n = p.shape[0]
    dist, sep = _new(p, n), _new(p, n, 3)
    capi.check(capi.load().mhip_distance_point_sphere(n, _ptr(p, cols=3), _ptr(c, cols=3), _ptr(r), _ptr(dist),
                                                      _ptr(sep), _stream()))
    return dist, sep


def distance_segment_sphere(a0, a1, c, r):
    """distance(LineSegment, Sphere, closest_point, arch_length, sep) (LineSegmentSphere.hpp:88-100)."""
    n = a0.shape[0]
    dist, cp, t, sep = _new(a0, n), _new(a0, n, 3), _new(a0, n), _new(a0, n, 3)
    capi.check(capi.load().mhip_distance_segment_sphere(n, _ptr(a0, cols=3), _ptr(a1, cols=3), _ptr(c, cols=3), _ptr(r),
                                                        _ptr(dist), _ptr(cp), _ptr(t), _ptr(sep), _stream()))
    return dist, cp, t, sep


def distance_segment_segment(a0, a1, b0, b1):
    n = a0.shape[0]
    dist, cp1, cp2 = _new(a0, n), _new(a0, n, 3), _new(a0, n, 3)
    s, t, sep = _new(a0, n), _new(a0, n), _new(a0, n, 3)
    capi.check(capi.load().mhip_distance_segment_segment(n, _ptr(a0, cols=3), _ptr(a1, cols=3), _ptr(b0, cols=3),
                                                         _ptr(b1, cols=3), _ptr(dist), _ptr(cp1), _ptr(cp2), _ptr(s),
                                                         _ptr(t), _ptr(sep), _stream()))
    return dist, cp1, cp2, s, t, sep


def distance_ellipsoid_ellipsoid(c1, q1, r1, c2, q2, r2):
    n = c1.shape[0]
    out = dict(dist=_new(c1, n), cp1=_new(c1, n, 3), cp2=_new(c1, n, 3), n1=_new(c1, n, 3), n2=_new(c1, n, 3))
    capi.check(capi.load().mhip_distance_ellipsoid_ellipsoid(
        n, _ptr(c1, cols=3), _ptr(q1, cols=4), _ptr(r1, cols=3), _ptr(c2, cols=3), _ptr(q2, cols=4), _ptr(r2, cols=3),
        _ptr(out["dist"]), _ptr(out["cp1"]), _ptr(out["cp2"]), _ptr(out["n1"]), _ptr(out["n2"]), _stream()))
    return out


def distance_point_ellipsoid(p, c, q, r):
    n = p.shape[0]
    dist, cp, nrm = _new(p, n), _new(p, n, 3), _new(p, n, 3)
    capi.check(capi.load().mhip_distance_point_ellipsoid(n, _ptr(p, cols=3), _ptr(c, cols=3), _ptr(q, cols=4),
                                                         _ptr(r, cols=3), _ptr(dist), _ptr(cp), _ptr(nrm), _stream()))
    return dist, cp, nrm


def contact_ellipsoids(pairs, center, quat, radii):
    c = pairs.shape[0]
    out = dict(sep=_new(center, c), normal=_new(center, c, 3), cp1=_new(center, c, 3), cp2=_new(center, c, 3),
               ra=_new(center, c, 3), rb=_new(center, c, 3))
    capi.check(capi.load().mhip_contact_ellipsoids(
        c, _ptr(pairs, torch.int32, 2), _ptr(center, cols=3), _ptr(quat, cols=4), _ptr(radii, cols=3),
        _ptr(out["sep"]), _ptr(out["normal"]), _ptr(out["cp1"]), _ptr(out["cp2"]), _ptr(out["ra"]), _ptr(out["rb"]),
        _stream()))
    return out


KIND_SPHERE, KIND_ROD, KIND_ELLIPSOID = 0, 1, 2


def compute_aabb_mixed(kind, center, quat, shape, conservative_ellipsoids=False):
    """conservative_ellipsoids=True: BUILD EXTENSION, the tight conservative ellipsoid box instead of the reference's
    (compute_aabb.hpp:82-103, which is not conservative for general orientations)"""
    n = kind.shape[0]
    aabb, brad = _new(center, n, 6), _new(center, n)
    fn = capi.load().mhip_compute_aabb_mixed_conservative if conservative_ellipsoids else capi.load().mhip_compute_aabb_mixed
    capi.check(fn(n, _ptr(kind, torch.int32), _ptr(center, cols=3), _ptr(quat, cols=4), _ptr(shape, cols=3),
                  _ptr(aabb), _ptr(brad), _stream()))
    return aabb, brad


def contact_mixed(pairs, kind, center, quat, shape, want_counts=False, box=None):
    """box: 3 edge lengths of an orthorhombic periodic box (body j at the nearest image of its centre)"""
    c = pairs.shape[0]
    out = dict(sep=_new(center, c), normal=_new(center, c, 3), cp1=_new(center, c, 3), cp2=_new(center, c, 3),
               ra=_new(center, c, 3), rb=_new(center, c, 3))
    counts = (C.c_size_t * 6)() if want_counts else None
    if box is not None:
        capi.check(capi.load().mhip_contact_mixed_periodic(
            c, _ptr(pairs, torch.int32, 2), _ptr(kind, torch.int32), _ptr(center, cols=3), _ptr(quat, cols=4),
            _ptr(shape, cols=3), (C.c_double * 3)(*[float(b) for b in box]), _ptr(out["sep"]), _ptr(out["normal"]),
            _ptr(out["cp1"]), _ptr(out["cp2"]), _ptr(out["ra"]), _ptr(out["rb"]), counts, _stream()))
    else:
        capi.check(capi.load().mhip_contact_mixed(
            c, _ptr(pairs, torch.int32, 2), _ptr(kind, torch.int32), _ptr(center, cols=3), _ptr(quat, cols=4),
            _ptr(shape, cols=3), _ptr(out["sep"]), _ptr(out["normal"]), _ptr(out["cp1"]), _ptr(out["cp2"]),
            _ptr(out["ra"]), _ptr(out["rb"]), counts, _stream()))
    if want_counts:
        out["class_counts"] = dict(zip(("SS", "SR", "SE", "RR", "RE", "EE"), [int(v) for v in counts]))
    return out


def contact_mixed_set_sphere_ellipsoid_route(reference_minimiser):
    """S-E of contact_mixed: False (default) = the exact point - ellipsoid distance in closed form, True = the reference's
    own point - ellipsoid routine (nine-start L-BFGS, PointEllipsoid.hpp:94-135), which the closed form matches to 1e-4"""
    capi.check(capi.load().mhip_contact_mixed_set_sphere_ellipsoid_route(1 if reference_minimiser else 0))


def contact_mixed_set_contraction(on):
    """BUILD OPTION (labelled): the S-E / E-E minimisation classes of contact_mixed from the build with fused
    multiply-adds -- results at the reference's 1e-4 instead of bit parity with the oracle.  Default off."""
    capi.check(capi.load().mhip_contact_mixed_set_contraction(1 if on else 0))


def contact_mixed_last_evaluations():
    """objective evaluations of the (S-E, R-E, E-E) classes in the last contact_mixed call (R-E is closed-form: 0)"""
    ev = (C.c_ulonglong * 3)()
    capi.check(capi.load().mhip_contact_mixed_last_evaluations(ev, _stream()))
    return dict(SE=int(ev[0]), RE=int(ev[1]), EE=int(ev[2]))


def ellipsoid_last_evaluations():
    ev = C.c_ulonglong(0)
    capi.check(capi.load().mhip_ellipsoid_last_evaluations(C.byref(ev), _stream()))
    return int(ev.value)


def _cell(box):
    """periodic cell argument: 3 edge lengths (PeriodicScaledMetric) or a 3x3 unit-cell matrix with the lattice vectors
    as columns (PeriodicMetric).  Returns (is_triclinic, ctypes array)."""
    import numpy as _np
    a = _np.asarray(box.detach().cpu() if isinstance(box, torch.Tensor) else box, dtype=_np.float64)
    if a.size == 3:
        return False, (C.c_double * 3)(*a.reshape(3).tolist())
    if a.size == 9:
        return True, (C.c_double * 9)(*a.reshape(9).tolist())
    raise ValueError("periodic cell must be 3 edge lengths or a 3x3 unit-cell matrix, got shape %s" % (a.shape,))


def contact_spheres(pairs, center, radius, box=None, out=None):
    c = pairs.shape[0]
    sep, normal = (_new(center, c), _new(center, c, 3)) if out is None else out
    tri, boxp = (False, None) if box is None else _cell(box)
    fn = capi.load().mhip_contact_spheres_triclinic if tri else capi.load().mhip_contact_spheres
    capi.check(fn(c, _ptr(pairs, torch.int32, 2), _ptr(center, cols=3), _ptr(radius), boxp, _ptr(sep), _ptr(normal),
                  _stream()))
    return sep, normal


def contact_spherocylinders(pairs, seg, center, want_points=True, out=None, arms="vector", box=None):
    """arms="vector": lever arms ra / rb [C,3]; arms="arclength": only (s, t), for ContactOperator(rod=...).
    box: 3 edge lengths of an orthorhombic periodic box (rod j at the nearest image of its centre)."""
    c = pairs.shape[0]
    if out is None:
        out = dict(sep=_new(seg, c), normal=_new(seg, c, 3))
        if arms == "vector":
            out.update(ra=_new(seg, c, 3), rb=_new(seg, c, 3))
        if want_points or arms == "arclength":
            out.update(s=_new(seg, c), t=_new(seg, c))
        if want_points:
            out.update(cp1=_new(seg, c, 3), cp2=_new(seg, c, 3))
    g = lambda k: _ptr(out.get(k), allow_none=True, name=k)  # noqa: E731
    if box is not None:
        capi.check(capi.load().mhip_contact_spherocylinders_periodic(
            c, _ptr(pairs, torch.int32, 2), _ptr(seg, cols=8), _ptr(center, cols=3),
            (C.c_double * 3)(*[float(b) for b in box]), g("sep"), g("normal"), g("cp1"), g("cp2"), g("ra"), g("rb"),
            g("s"), g("t"), _stream()))
        return out
    capi.check(capi.load().mhip_contact_spherocylinders(c, _ptr(pairs, torch.int32, 2), _ptr(seg, cols=8),
                                                        _ptr(center, cols=3), g("sep"), g("normal"), g("cp1"),
                                                        g("cp2"), g("ra"), g("rb"), g("s"), g("t"), _stream()))
    return out


# ---- broad phase (GenNeighborLinks, mundy_mesh/GenNeighborLinkers.hpp:294-866) ---------------------------------------
class GenNeighborLinks:
    """Builder-style mirror of mundy::mesh::GenNeighborLinks: set_* -> concretize() -> generate().

    generate(aabb, center, bounding_radius) returns True when a search was performed (first call, or some centre moved
    more than half the search buffer, :510-543, :603-615); the links are then available as .pairs ([P, 2] int32,
    sorted by (source, target)), .row_ptr / .col (CSR).
    """

    def __init__(self):
        h = C.c_void_p()
        capi.check(capi.load().mhip_broadphase_create(C.byref(h)))
        self._h = h
        self._cfg = capi.BroadphaseConfig(SEARCH_SPHERES, 0, 0.0, 0, (C.c_double * 3)(0, 0, 0), 0, 0)
        self._concretized = False
        self._generated = False
        self.pairs = self.row_ptr = self.col = None
        self.num_pairs = 0

    def _setter_guard(self, what):
        if self._concretized:
            raise RuntimeError("Cannot set %s after concretization." % what)  # :402-462

    def set_search_buffer(self, search_buffer):
        self._setter_guard("search buffer")
        self._cfg.buffer = float(search_buffer)
        return self

    def set_search_kind(self, kind):
        self._setter_guard("search kind")
        self._cfg.search_kind = int(kind)
        return self

    def set_enforce_source_target_symmetry(self, value):
        self._setter_guard("enforce source-target symmetry")
        self._cfg.symmetric = 1 if value else 0
        return self

    def set_periodic_box(self, box):
        self._setter_guard("periodic box")
        if box is None:
            self._cfg.periodic = 0
        else:
            import numpy as np
            a = np.asarray(box, dtype=np.float64)
            if a.shape == (3,):
                self._cfg.periodic = 1
                self._cfg.box = (C.c_double * 3)(*[float(b) for b in a])
            elif a.shape == (3, 3):   # the unit cell of PeriodicMetric: lattice vectors as columns (periodicity.hpp:233-332)
                self._cfg.periodic = 2
                self._cfg.cell = (C.c_double * 9)(*[float(b) for b in a.reshape(9)])
            else:
                raise ValueError("periodic cell must be 3 edge lengths or a 3x3 unit-cell matrix, got shape %s" % (a.shape,))
        return self

    def set_search_method(self, method):
        """stk::search::SearchMethod of the reference (:443-447; its default is MORTON_LBVH): SEARCH_METHOD_AUTO,
        SEARCH_METHOD_GRID or SEARCH_METHOD_MORTON_LBVH -- same lists, different structure"""
        self._setter_guard("search method")
        self._cfg.method = int(method)
        return self

    def set_exclude_self_interactions(self, value=True):
        """search_filters::ExcludeSelfInteractions (:185-200); the default.  False lets (i, i) be a result."""
        self._setter_guard("search filter")
        self._cfg.include_self = 0 if value else 1
        return self

    def acts_on(self, source_mask=None, target_mask=None):
        """acts_on(source_selector, target_selector, ...) (:486-507): uint8 masks [n] over the bodies (None = all); a
        result (s, t) needs s among the sources and t among the targets"""
        self._setter_guard("source/targets")
        self._sets = (source_mask, target_mask)
        n = (source_mask if source_mask is not None else target_mask)
        capi.check(capi.load().mhip_broadphase_set_sets(
            self._h, 0 if n is None else n.shape[0], _ptr(source_mask, torch.uint8, allow_none=True, name="source_mask"),
            _ptr(target_mask, torch.uint8, allow_none=True, name="target_mask"), _stream()))
        return self

    def set_excluded_partners(self, ex_ptr, ex_idx):
        """search_filters::ExcludeConnectedEntities (:202-236) / the already-linked neighbours when duplicate links
        are not allowed (:91-113): CSR (int32 ex_ptr [n + 1], ex_idx) of partners each source must not be paired with.
        May be called again between generates (the connectivity of a mesh changes); it invalidates the list."""
        if ex_ptr is None:
            capi.check(capi.load().mhip_broadphase_set_exclusions(self._h, 0, None, None, 0, _stream()))
        else:
            capi.check(capi.load().mhip_broadphase_set_exclusions(
                self._h, ex_ptr.shape[0] - 1, _ptr(ex_ptr, torch.int32, name="ex_ptr"),
                _ptr(ex_idx, torch.int32, name="ex_idx"), ex_idx.shape[0], _stream()))
        self._generated = False
        return self

    def set_identities(self, entity_id=None, owner_rank=None, n=None):
        """(stk::mesh::EntityId, owner rank) of every body (:575-584): int64 ids (bit pattern of the u64), int32 ranks"""
        n = n if n is not None else (entity_id if entity_id is not None else owner_rank).shape[0]
        capi.check(capi.load().mhip_broadphase_set_identities(
            self._h, n, _ptr(entity_id, torch.int64, allow_none=True, name="entity_id"),
            _ptr(owner_rank, torch.int32, allow_none=True, name="owner_rank"), _stream()))
        return self

    def ident_pairs(self):
        """the links as stk::search IdentProcIntersection rows: (source id, source proc, target id, target proc)"""
        dev = self.pairs.device
        sid, tid = (torch.empty(self.num_pairs, dtype=torch.int64, device=dev) for _ in range(2))
        sp, tp = (torch.empty(self.num_pairs, dtype=torch.int32, device=dev) for _ in range(2))
        capi.check(capi.load().mhip_broadphase_get_ident_pairs(self._h, _ptr(sid, torch.int64), _ptr(sp, torch.int32),
                                                               _ptr(tid, torch.int64), _ptr(tp, torch.int32), _stream()))
        return sid, sp, tid, tp

    def method_used(self):
        m = C.c_int(0)
        capi.check(capi.load().mhip_broadphase_method_used(self._h, C.byref(m)))
        return m.value

    def minimum_image_complete(self):
        """False when the last build's periodic cell was so small (an edge <= 4 x the largest reach) that volumes can
        also meet through a second image -- pairs the minimum-image predicate does not report"""
        m = C.c_int(0)
        capi.check(capi.load().mhip_broadphase_minimum_image_complete(self._h, C.byref(m)))
        return bool(m.value)

    def export_coo(self, first_link_id=0, source_rank=3, target_rank=3):
        """MuNDy's LinkCOOData rows (LinkMetaData.hpp:102-106): (link ids [P], linked entity ids [P, 2], linked entity
        ranks [P, 2] uint8; 3 = stk::topology::ELEM_RANK)"""
        dev = self.pairs.device
        lid = torch.empty(self.num_pairs, dtype=torch.int64, device=dev)
        ids = torch.empty((self.num_pairs, 2), dtype=torch.int64, device=dev)
        ranks = torch.empty((self.num_pairs, 2), dtype=torch.uint8, device=dev)
        capi.check(capi.load().mhip_links_export_coo(self._h, int(first_link_id), int(source_rank), int(target_rank),
                                                     _ptr(lid, torch.int64), _ptr(ids, torch.int64),
                                                     _ptr(ranks, torch.uint8), _stream()))
        return lid, ids, ranks

    def export_crs(self, first_link_id=0, bucket_capacity=512):
        """entity -> connected links in LinkCRSBucketConn's layout (LinkCRSBucketConn.hpp:183-191), entities in index
        order cut into buckets of bucket_capacity: (num_connected_links [n], sparse_connectivity_offsets [nb, cap + 1],
        sparse_connectivity [2 P], bucket_begin [nb + 1])"""
        dev = self.pairs.device
        n = self.row_ptr.shape[0] - 1
        nb = (n + bucket_capacity - 1) // bucket_capacity
        num = torch.empty(n, dtype=torch.int32, device=dev)
        offs = torch.empty((nb, bucket_capacity + 1), dtype=torch.int32, device=dev)
        conn = torch.empty(2 * self.num_pairs, dtype=torch.int64, device=dev)
        begin = torch.empty(nb + 1, dtype=torch.int64, device=dev)
        capi.check(capi.load().mhip_links_export_crs(self._h, int(first_link_id), int(bucket_capacity),
                                                     _ptr(num, torch.int32), _ptr(offs, torch.int32),
                                                     _ptr(conn, torch.int64), _ptr(begin, torch.int64), _stream()))
        return num, offs, conn, begin

    def concretize(self):
        if self._concretized:
            raise RuntimeError("Cannot concretize more than once.")  # :494
        self._concretized = True
        return self

    def needs_rebuild(self, center):
        flag = C.c_int(0)
        capi.check(capi.load().mhip_broadphase_needs_rebuild(self._h, center.shape[0], _ptr(center, cols=3),
                                                             C.byref(flag), _stream()))
        return bool(flag.value)

    def invalidate(self):
        """the body numbering changed (reordering, migration): the next generate() rebuilds whatever the rebuild rule says"""
        self._generated = False

    def generate(self, aabb, center, bounding_radius, force=False):
        if not self._concretized:
            raise RuntimeError("Cannot generate links before concretization.")  # :511
        if self._generated and not force and not self.needs_rebuild(center):
            return False
        n = center.shape[0]
        cnt = C.c_size_t(0)
        capi.check(capi.load().mhip_broadphase_build(
            self._h, C.byref(self._cfg), n, _ptr(aabb, cols=6, allow_none=True, name="aabb"), _ptr(center, cols=3),
            _ptr(bounding_radius, allow_none=True, name="bounding_radius"), C.byref(cnt), _stream()))
        self.num_pairs = int(cnt.value)
        self.pairs = torch.empty((self.num_pairs, 2), dtype=torch.int32, device=center.device)
        self.row_ptr = torch.empty(n + 1, dtype=torch.int32, device=center.device)
        self.col = torch.empty(self.num_pairs, dtype=torch.int32, device=center.device)
        capi.check(capi.load().mhip_broadphase_get_pairs(self._h, _ptr(self.pairs, torch.int32),
                                                         _ptr(self.row_ptr, torch.int32), _ptr(self.col, torch.int32),
                                                         _stream()))
        self._generated = True
        return True

    def close(self):
        if self._h:
            capi.load().mhip_broadphase_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- convex (mundy_math/convex.hpp) ----------------------------------------------------------------------------------
@dataclass
class PGDConfig:  # convex.hpp:519-525
    max_iters: int = 1000
    tol: float = 1e-8
    residual_kind: int = RESIDUAL_PROJECTED_DIFF


@dataclass
class SolveResult:  # convex.hpp:527-541
    num_iters: int = 0
    residual: float = 0.0
    converged: bool = False


def _space(space):
    kind, lo, hi = space
    return capi.Space(int(kind), float(lo), float(hi))


def _cfg(cfg):
    return capi.PgdConfig(int(cfg.max_iters), float(cfg.tol), int(cfg.residual_kind))


LCP_SPACE = (SPACE_LOWER_BOUND, 0.0, 0.0)  # to_cqpp: LowerBound{0} (convex.hpp:424-428)


def axpby(alpha, x, beta, y):
    capi.check(capi.load().mhip_axpby(x.shape[0], alpha, _ptr(x), beta, _ptr(y), _stream()))


def wrapped_axpbyz(alpha, x, beta, y, z, space):
    sp = _space(space)
    capi.check(capi.load().mhip_wrapped_axpbyz(x.shape[0], alpha, _ptr(x), beta, _ptr(y), _ptr(z), C.byref(sp),
                                               _stream()))


def diff_dot(x, y, x2=None, y2=None):
    r = C.c_double()
    if x2 is None:
        capi.check(capi.load().mhip_diff_dot2(x.shape[0], _ptr(x), _ptr(y), C.byref(r), _stream()))
    else:  # diff_dot(x1, x2, y1, y2) = sum (x1-x2)(y1-y2)
        capi.check(capi.load().mhip_diff_dot4(x.shape[0], _ptr(x), _ptr(y), _ptr(x2), _ptr(y2), C.byref(r), _stream()))
    return r.value


def residual(kind, x, grad, space):
    r = C.c_double()
    sp = _space(space)
    capi.check(capi.load().mhip_residual(x.shape[0], kind, _ptr(x), _ptr(grad), C.byref(sp), C.byref(r), _stream()))
    return r.value


def bb_step(x_old, g_old, x, g):
    r = C.c_double()
    capi.check(capi.load().mhip_bb_step(x.shape[0], _ptr(x_old), _ptr(g_old), _ptr(x), _ptr(g), C.byref(r), _stream()))
    return r.value


def gemv(A, x):
    y = torch.empty_like(x)
    capi.check(capi.load().mhip_gemv(x.shape[0], _ptr(A), _ptr(x), _ptr(y), _stream()))
    return y


class ContactOperator:
    """Matrix-free A = dt D^T M D over a neighbour list (the LinearOp of seam S2; apply(x, y) as convex.hpp:133-136)."""

    def __init__(self, pairs, normal, mob_trans, dt, ra=None, rb=None, mob_rot=None, rod=None, priority=None):
        """rod = (arc_s, arc_t, seg): spherocylinders with rod-compressed lever arms (mhip_contact_op_create_rods).
        priority [C] (optional, e.g. the signed separations): locality hint -- each body lists the contacts with
        priority < 0 first; changes nothing but the summation order."""
        self.num_constraints = pairs.shape[0]
        self.num_bodies = mob_trans.shape[0]
        self._keep = (pairs, normal, ra, rb, mob_trans, mob_rot, rod, priority)  # the handle holds views of these
        prio = _ptr(priority, allow_none=True, name="priority")
        h = C.c_void_p()
        if rod is not None:
            arc_s, arc_t, seg = rod
            capi.check(capi.load().mhip_contact_op_create_rods(
                C.byref(h), self.num_constraints, self.num_bodies, _ptr(pairs, torch.int32, 2), _ptr(normal, cols=3),
                _ptr(arc_s), _ptr(arc_t), _ptr(seg, cols=8), _ptr(mob_trans), _ptr(mob_rot), float(dt), prio, _stream()))
        else:
            capi.check(capi.load().mhip_contact_op_create(
                C.byref(h), self.num_constraints, self.num_bodies, _ptr(pairs, torch.int32, 2), _ptr(normal, cols=3),
                _ptr(ra, allow_none=True, name="ra"), _ptr(rb, allow_none=True, name="rb"), _ptr(mob_trans),
                _ptr(mob_rot, allow_none=True, name="mob_rot"), float(dt), prio, _stream()))
        self._h = h
        self._device = normal.device

    def refresh(self, normal, ra=None, rb=None, rod=None):
        """same pairs, new geometry (a step that reuses the neighbour list): keeps the incidence index, redoes the
        half-edge records from the new arrays (mhip_contact_op_refresh[_rods])"""
        pairs, _, _, _, mob_trans, mob_rot, old_rod, priority = self._keep
        if (rod is None) != (old_rod is None):
            raise ValueError("refresh must keep the operator's kinematics")
        if rod is not None:
            arc_s, arc_t, seg = rod
            capi.check(capi.load().mhip_contact_op_refresh_rods(self._h, _ptr(normal, cols=3), _ptr(arc_s), _ptr(arc_t),
                                                                _ptr(seg, cols=8), _stream()))
        else:
            capi.check(capi.load().mhip_contact_op_refresh(self._h, _ptr(normal, cols=3),
                                                           _ptr(ra, allow_none=True, name="ra"),
                                                           _ptr(rb, allow_none=True, name="rb"), _stream()))
        self._keep = (pairs, normal, ra, rb, mob_trans, mob_rot, rod, priority)

    def apply(self, x, y=None):
        y = torch.empty_like(x) if y is None else y
        capi.check(capi.load().mhip_contact_op_apply(self._h, _ptr(x), _ptr(y), _stream()))
        return y

    def body_velocity(self):
        """[N, 6] (U, W) of the last evaluated iterate -- a view of handle-owned memory; clone to keep."""
        p = C.c_void_p()
        capi.check(capi.load().mhip_contact_op_body_velocity(self._h, C.byref(p)))
        n = self.num_bodies
        if n == 0:
            return torch.empty((0, 6), dtype=torch.float64, device=self._device)
        out = torch.empty((n, 6), dtype=torch.float64, device=self._device)
        capi.check(capi.load().mhip_deep_copy(6 * n, _ptr(out), p, _stream()))
        return out

    def set_work_mapping(self, xcd_tile=-1, lanes_per_body=-1):
        """layout of the sweeps on the chip (time only; results do not depend on it)"""
        capi.check(capi.load().mhip_contact_op_set_work_mapping(self._h, int(xcd_tile), int(lanes_per_body)))

    def set_tiering(self, mode=1):
        """cold tier of the fused solve: 0 off, 1 on (default), 2 test hook (time only; results do not depend on it)"""
        capi.check(capi.load().mhip_contact_op_set_tiering(self._h, int(mode)))

    def set_drift_source(self, source=0):
        """tiered solves, time only: 0 by size, 1 drift from the difference of the two body rows, 2 from the change of
        the force kept in registers (mhip_contact_op_set_drift_source)"""
        capi.check(capi.load().mhip_contact_op_set_drift_source(self._h, int(source)))

    def drift_source(self):
        """the form a tiered solve on this operator takes: 1 rows, 2 registers"""
        v = C.c_int(0)
        capi.check(capi.load().mhip_contact_op_get_drift_source(self._h, C.byref(v)))
        return int(v.value)

    def tier_stats(self):
        """(tiered iterations, mean hot share, renumberings, wake-ups) of the last solve_lcp on this operator"""
        it, rn, wk, hot = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0), C.c_double(0.0)
        capi.check(capi.load().mhip_contact_op_tier_stats(self._h, C.byref(it), C.byref(hot), C.byref(rn), C.byref(wk)))
        return dict(tiered_iterations=it.value, mean_hot_fraction=hot.value, renumberings=rn.value, wakeups=wk.value)

    def set_profiling(self, enable=True):
        capi.check(capi.load().mhip_contact_op_set_profiling(self._h, 1 if enable else 0))

    def get_profile(self):
        """(k_body ms, k_constraint ms, timed iterations) accumulated by the fused solver since set_profiling"""
        a, b, n = C.c_double(), C.c_double(), C.c_size_t()
        capi.check(capi.load().mhip_contact_op_get_profile(self._h, C.byref(a), C.byref(b), C.byref(n)))
        return a.value, b.value, int(n.value)

    def close(self):
        if getattr(self, "_h", None):
            capi.load().mhip_contact_op_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _state(x0, state):
    if state is not None:
        return state
    x = x0.clone()
    return x, torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)


def solve_cqpp(A, q, space, x0, cfg=None, state=None, fused=True):
    """solve_cqpp (convex.hpp:789-797).  A is a ContactOperator (matrix free) or a dense [n, n] tensor.
    Returns (x, grad, SolveResult); `state` = caller-owned (x, grad, x_tmp, grad_tmp) as PGDState holds them."""
    cfg = cfg or PGDConfig()
    x, g, x_tmp, g_tmp = _state(x0, state)
    res, sp, pc = capi.SolveResult(), _space(space), _cfg(cfg)
    lib = capi.load()
    if isinstance(A, ContactOperator):
        fn = lib.mhip_bbpgd_solve_contact if fused else lib.mhip_bbpgd_solve_contact_unfused
        capi.check(fn(A._h, _ptr(q), C.byref(sp), C.byref(pc), _ptr(x), _ptr(g), _ptr(x_tmp), _ptr(g_tmp),
                      C.byref(res), _stream()))
    else:
        n = q.shape[0]
        if A.dim() != 2 or A.shape[0] != n or A.shape[1] != n:
            raise ValueError("gemv: dimension mismatch A vs x")  # convex.hpp:171-172
        capi.check(lib.mhip_bbpgd_solve_dense(n, _ptr(A), _ptr(q), C.byref(sp), C.byref(pc), _ptr(x), _ptr(g),
                                              _ptr(x_tmp), _ptr(g_tmp), C.byref(res), _stream()))
    return x, g, SolveResult(int(res.num_iters), float(res.residual), bool(res.converged))


def solve_friction_contact(A, sep, mu, p0=None, cfg=None, method="bbpgd"):
    """BUILD EXTENSION, parity unpinned (the reference has no frictional solver): Coulomb friction as a
    cone complementarity problem on the vector-arm ContactOperator A (lever arms to the contact points ON THE
    SURFACES), solved by the fused BBPGD iteration with a per-contact cone projection.  Returns (p [C,3] world-frame
    impulses, g [C,3], SolveResult); mu = 0 reproduces the frictionless LCP (p = lambda n)."""
    cfg = cfg or PGDConfig()
    c = sep.shape[0]
    p = torch.zeros((c, 3), dtype=torch.float64, device=sep.device) if p0 is None else p0.clone()
    g = torch.empty_like(p)
    res, pc = capi.SolveResult(), _cfg(cfg)
    if method not in ("bbpgd", "apgd"):   # "apgd": Mazhar et al. 2015, one operator application per sweep
        raise ValueError("method must be 'bbpgd' or 'apgd'")
    fn = capi.load().mhip_bbpgd_solve_contact_friction if method == "bbpgd" else capi.load().mhip_apgd_solve_contact_friction
    capi.check(fn(A._h, _ptr(sep), float(mu), C.byref(pc), _ptr(p, cols=3), _ptr(g, cols=3), C.byref(res), _stream()))
    return p, g, SolveResult(int(res.num_iters), float(res.residual), bool(res.converged))


def surface_lever_arms(pairs, normal, ra, rb, radius):
    """lever arms to the contact points on the surfaces of two round-capped bodies (spheres, spherocylinders) from
    the centreline arms: ra + r_i n, rb - r_j n.  Friction acts there; for a frictionless contact the difference is
    parallel to the force and drops out of the torque."""
    ri = radius[pairs[:, 0].long()][:, None]
    rj = radius[pairs[:, 1].long()][:, None]
    return (ra + ri * normal).contiguous(), (rb - rj * normal).contiguous()


def solve_small_cqpp_batch(A, q, space, x0, cfg=None):
    """make_mundy_math_cqpp + solve_cqpp on a batch of small dense problems, one thread each (convex.hpp:288-350,
    :722-733).  A [b, n, n], q [b, n], x0 [b, n]; returns (x, grad, num_iters, residual, converged) device tensors."""
    cfg = cfg or PGDConfig()
    b, n = q.shape
    if A.shape != (b, n, n):
        raise ValueError("A must be [batch, n, n]")
    x = x0.clone()
    g = torch.empty_like(x)
    it = torch.empty(b, dtype=torch.int32, device=q.device)
    res = torch.empty(b, dtype=torch.float64, device=q.device)
    conv = torch.empty(b, dtype=torch.int32, device=q.device)
    sp, pc = _space(space), _cfg(cfg)
    capi.check(capi.load().mhip_solve_small_cqpp_batch(b, n, _ptr(A), _ptr(q), C.byref(sp), C.byref(pc), _ptr(x),
                                                       _ptr(g), _ptr(it, torch.int32), _ptr(res),
                                                       _ptr(conv, torch.int32), _stream()))
    return x, g, it, res, conv.bool()


@dataclass
class CollisionResult:  # scrap/lcp_spheres/NgpLcp.cpp:550-554
    max_abs_projected_sep: float = 0.0
    ite_count: int = 0
    max_displacement: float = 0.0


def resolve_collisions(op, sep, lam, dt, max_allowable_overlap=1e-5, max_col_iterations=10000):
    """The scrap app's own BBPGD (resolve_collisions, NgpLcp.cpp:558-759) on a ContactOperator.  `lam` is the initial
    guess and receives the multipliers.  Returns (lam, g = sep + dt*sep_dot, CollisionResult)."""
    lam_tmp, g, g_tmp = torch.empty_like(lam), torch.empty_like(lam), torch.empty_like(lam)
    res, spd = capi.SolveResult(), C.c_double(0.0)
    capi.check(capi.load().mhip_scrap_bbpgd_solve_contact(
        op._h, _ptr(sep), float(max_allowable_overlap), int(max_col_iterations), _ptr(lam), _ptr(lam_tmp), _ptr(g),
        _ptr(g_tmp), C.byref(res), C.byref(spd), _stream()))
    return lam, g, CollisionResult(float(res.residual), int(res.num_iters), float(spd.value) * float(dt))


def solve_lcp(A, q, x0, cfg=None, state=None, fused=True):
    """solve_lcp (convex.hpp:839-845): 0 <= A x + q  _|_  x >= 0."""
    return solve_cqpp(A, q, LCP_SPACE, x0, cfg, state, fused)


# ---- reordering / integration -----------------------------------------------------------------------------------------
def morton_order(center, lo, cell_size):
    n = center.shape[0]
    perm = torch.empty(n, dtype=torch.int32, device=center.device)
    lop = (C.c_double * 3)(*[float(v) for v in lo])
    capi.check(capi.load().mhip_morton_order(n, _ptr(center, cols=3), lop, float(cell_size), _ptr(perm, torch.int32),
                                             _stream()))
    return perm


def curve_order(center, lo, hi, level, key_table):
    """permutation along a lattice curve given by key_table [2^level]^3 (device int32), e.g. the Hilbert table of
    mundy_amd.distributed.hilbert_key_table; ties by index"""
    n = center.shape[0]
    perm = torch.empty(n, dtype=torch.int32, device=center.device)
    lop = (C.c_double * 3)(*[float(v) for v in lo])
    hip = (C.c_double * 3)(*[float(v) for v in hi])
    capi.check(capi.load().mhip_curve_order(n, _ptr(center, cols=3), lop, hip, int(level),
                                            _ptr(key_table.reshape(-1), torch.int32), _ptr(perm, torch.int32), _stream()))
    return perm


def curve_keys(center, lo, hi, level, key_table):
    """key_table entry of every body's cell (int32 tensor): what curve_order sorts by"""
    n = center.shape[0]
    keys = torch.empty(n, dtype=torch.int32, device=center.device)
    lop = (C.c_double * 3)(*[float(v) for v in lo])
    hip = (C.c_double * 3)(*[float(v) for v in hi])
    capi.check(capi.load().mhip_curve_keys(n, _ptr(center, cols=3), lop, hip, int(level),
                                           _ptr(key_table.reshape(-1), torch.int32), _ptr(keys, torch.int32), _stream()))
    return keys


def hilbert_key_table(level):
    """table[ix, iy, iz] = position of the lattice cell along mundy::math::hilbert_3d (Hilbert.hpp:48-83): the library's
    own generator (host code, needs no GPU); numpy int32 array of shape (2^level,) * 3"""
    import numpy as np
    n = 1 << int(level)
    table = np.empty((n, n, n), dtype=np.int32)
    capi.check(capi.load().mhip_hilbert_key_table(int(level), C.c_void_p(table.ctypes.data)))
    return table


def sort_by_key(keys):
    """stable ascending order of int64 keys (non-negative): int32 permutation (library radix sort)"""
    perm = torch.empty(keys.shape[0], dtype=torch.int32, device=keys.device)
    capi.check(capi.load().mhip_sort_by_key_u64(keys.shape[0], _ptr(keys, torch.int64), _ptr(perm, torch.int32), _stream()))
    return perm


def select_contacts(sep, cutoff):
    """BUILD OPTION: ascending indices (int32) of the candidate pairs whose signed separation is not above `cutoff`
    (NaN kept) -- wavefront ballot / prefix-sum compaction (mhip_select_contacts)"""
    kept = torch.empty(sep.shape[0], dtype=torch.int32, device=sep.device)
    cnt = C.c_size_t(0)
    capi.check(capi.load().mhip_select_contacts(sep.shape[0], _ptr(sep), float(cutoff), _ptr(kept, torch.int32),
                                                C.byref(cnt), _stream()))
    return kept[:int(cnt.value)]


def gather_rows(perm, src):
    src2 = src if src.dim() == 2 else src.unsqueeze(1)
    dst = torch.empty((perm.shape[0], src2.shape[1]), dtype=src2.dtype, device=src2.device)
    capi.check(capi.load().mhip_gather_rows(perm.shape[0], src2.shape[1], _ptr(perm, torch.int32), _ptr(src2),
                                            _ptr(dst), _stream()))
    return dst if src.dim() == 2 else dst.squeeze(1)


def periodic_sep(box, p1, p2):
    """metric.sep(p1, p2): PeriodicScaledMetric for 3 edge lengths, PeriodicMetric for a 3x3 unit cell"""
    out = torch.empty_like(p1)
    tri, b = _cell(box)
    fn = capi.load().mhip_periodic_sep_triclinic if tri else capi.load().mhip_periodic_sep
    capi.check(fn(p1.shape[0], b, _ptr(p1, cols=3), _ptr(p2, cols=3), _ptr(out), _stream()))
    return out


def wrap_rigid(box, center):
    """wrap_rigid_inplace of spheres / spherocylinders / ellipsoids: their centres are wrapped into the unit cell"""
    tri, b = _cell(box)
    fn = capi.load().mhip_wrap_rigid_triclinic if tri else capi.load().mhip_wrap_rigid
    capi.check(fn(center.shape[0], b, _ptr(center, cols=3), _stream()))
    return center


def shift_image(cell, p, images):
    """PeriodicMetric::shift_image: p + h * images (images [n, 3] int32)"""
    out = torch.empty_like(p)
    a = _cell(cell)[1] if _cell(cell)[0] else None
    if a is None:  # edge lengths -> diagonal unit cell (periodic_metric_from_unit_cell, periodicity.hpp:848-855)
        e = _cell(cell)[1]
        a = (C.c_double * 9)(e[0], 0, 0, 0, e[1], 0, 0, 0, e[2])
    capi.check(capi.load().mhip_shift_image_triclinic(p.shape[0], a, _ptr(p, cols=3), _ptr(images, torch.int32, 3),
                                                      _ptr(out), _stream()))
    return out


def unit_cell_inverse(cell):
    import numpy as _np
    a = (C.c_double * 9)(*_np.asarray(cell, dtype=_np.float64).reshape(9).tolist())
    out = (C.c_double * 9)()
    capi.check(capi.load().mhip_unit_cell_inverse(a, out))
    return _np.array(list(out)).reshape(3, 3)


def integrate_euler(dt, velocity, center, quat=None):
    capi.check(capi.load().mhip_integrate_euler(center.shape[0], float(dt), _ptr(velocity, cols=6),
                                                _ptr(center, cols=3), _ptr(quat, cols=4, allow_none=True), _stream()))
