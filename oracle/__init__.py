"""ctypes front-end of the CPU oracle (oracle/mundy_oracle.hpp).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by
mundy_amd/.  `build()` compiles the two shared objects with g++ (see oracle/Makefile); `lib()` is the bit-parity
build (-ffp-contract=off), `lib(fast=True)` the -O3 timing build used as the CPU baseline.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}

f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("oracle_capi.cpp", "mundy_oracle.hpp", "Makefile")]
    outs = [os.path.join(_HERE, f) for f in ("liboracle.so", "liboracle_fast.so")]
    stale = force or any(not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s) for s in srcs)
                         for o in outs)
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "-s", "all"])


def lib(fast=False):
    key = "fast" if fast else "exact"
    if key not in _LIBS:
        path = os.path.join(_HERE, "liboracle_fast.so" if fast else "liboracle.so")
        if not os.path.exists(path):
            build()
        _LIBS[key] = C.CDLL(path)
    return _LIBS[key]


def _p(a):
    """pointer (void*) to a contiguous numpy array or None"""
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"], "oracle arrays must be C-contiguous"
    return a.ctypes.data_as(C.c_void_p)


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def num_threads(fast=True):
    return int(lib(fast).o_num_threads())


def set_num_threads(n, fast=True):
    lib(fast).o_set_num_threads(C.c_int(int(n)))


# ---- per-body geometry ------------------------------------------------------------------------------------------
def compute_aabb_spheres(center, radius, fast=False):
    center, radius = _f(center), _f(radius)
    out = np.empty((len(radius), 6))
    lib(fast).o_compute_aabb_spheres(C.c_size_t(len(radius)), _p(center), _p(radius), _p(out))
    return out


def compute_aabb_spherocylinders(center, quat, radius, length, fast=False):
    center, quat, radius, length = _f(center), _f(quat), _f(radius), _f(length)
    out = np.empty((len(radius), 6))
    lib(fast).o_compute_aabb_spherocylinders(C.c_size_t(len(radius)), _p(center), _p(quat), _p(radius), _p(length),
                                             _p(out))
    return out


def compute_aabb_ellipsoids(center, quat, radii):
    center, quat, radii = _f(center), _f(quat), _f(radii)
    out = np.empty((len(center), 6))
    lib().o_compute_aabb_ellipsoids(C.c_size_t(len(center)), _p(center), _p(quat), _p(radii), _p(out))
    return out


def compute_aabb_ellipsoids_conservative(center, quat, radii):
    """build extension (SURVEY a7's flagged option): tight box of the rotated ellipsoid"""
    center, quat, radii = _f(center), _f(quat), _f(radii)
    out = np.empty((len(center), 6))
    lib().o_compute_aabb_ellipsoids_conservative(C.c_size_t(len(center)), _p(center), _p(quat), _p(radii), _p(out))
    return out


def compute_aabb_segments(p0, p1, radius):
    p0, p1, radius = _f(p0), _f(p1), _f(radius)
    out = np.empty((len(radius), 6))
    lib().o_compute_aabb_segments(C.c_size_t(len(radius)), _p(p0), _p(p1), _p(radius), _p(out))
    return out


def bounding_radius_spherocylinders(radius, length):
    radius, length = _f(radius), _f(length)
    out = np.empty(len(radius))
    lib().o_bounding_radius_spherocylinders(C.c_size_t(len(radius)), _p(radius), _p(length), _p(out))
    return out


def bounding_radius_ellipsoids(radii):
    radii = _f(radii)
    out = np.empty(len(radii))
    lib().o_bounding_radius_ellipsoids(C.c_size_t(len(radii)), _p(radii), _p(out))
    return out


def bounding_radius_segments(p0, p1, radius):
    p0, p1, radius = _f(p0), _f(p1), _f(radius)
    out = np.empty(len(radius))
    lib().o_bounding_radius_segments(C.c_size_t(len(radius)), _p(p0), _p(p1), _p(radius), _p(out))
    return out


def spherocylinder_segments(center, quat, radius, length, fast=False):
    center, quat, radius, length = _f(center), _f(quat), _f(radius), _f(length)
    seg = np.empty((len(radius), 8))
    lib(fast).o_spherocylinder_segments(C.c_size_t(len(radius)), _p(center), _p(quat), _p(radius), _p(length),
                                        _p(seg))
    return seg


def quat_rotate(quat, v):
    quat, v = _f(quat), _f(v)
    out = np.empty_like(v)
    lib().o_quat_rotate(C.c_size_t(len(v)), _p(quat), _p(v), _p(out))
    return out


def quat_from_parallel_transport(frm, to):
    frm, to = _f(frm), _f(to)
    out = np.empty((len(frm), 4))
    lib().o_quat_from_parallel_transport(C.c_size_t(len(frm)), _p(frm), _p(to), _p(out))
    return out


# ---- distances ----------------------------------------------------------------------------------------------------
def distance_point_segment(p, a0, a1):
    p, a0, a1 = _f(p), _f(a0), _f(a1)
    n = len(p)
    dist, cp, t, sep = np.empty(n), np.empty((n, 3)), np.empty(n), np.empty((n, 3))
    lib().o_distance_point_segment(C.c_size_t(n), _p(p), _p(a0), _p(a1), _p(dist), _p(cp), _p(t), _p(sep))
    return dist, cp, t, sep


def distance_segment_segment(a0, a1, b0, b1, fast=False):
    a0, a1, b0, b1 = _f(a0), _f(a1), _f(b0), _f(b1)
    n = len(a0)
    dist, cp1, cp2 = np.empty(n), np.empty((n, 3)), np.empty((n, 3))
    s, t, sep = np.empty(n), np.empty(n), np.empty((n, 3))
    lib(fast).o_distance_segment_segment(C.c_size_t(n), _p(a0), _p(a1), _p(b0), _p(b1), _p(dist), _p(cp1), _p(cp2),
                                         _p(s), _p(t), _p(sep))
    return dist, cp1, cp2, s, t, sep


def distance_sphere_sphere(c1, r1, c2, r2):
    c1, r1, c2, r2 = _f(c1), _f(r1), _f(c2), _f(r2)
    n = len(r1)
    dist, sep = np.empty(n), np.empty((n, 3))
    lib().o_distance_sphere_sphere(C.c_size_t(n), _p(c1), _p(r1), _p(c2), _p(r2), _p(dist), _p(sep))
    return dist, sep


def distance_point_sphere(p, c, r):
    p, c, r = _f(p), _f(c), _f(r)
    n = len(r)
    dist, sep = np.empty(n), np.empty((n, 3))
    lib().o_distance_point_sphere(C.c_size_t(n), _p(p), _p(c), _p(r), _p(dist), _p(sep))
    return dist, sep


def distance_segment_sphere(a0, a1, c, r):
    a0, a1, c, r = _f(a0), _f(a1), _f(c), _f(r)
    n = len(r)
    dist, cp, t, sep = np.empty(n), np.empty((n, 3)), np.empty(n), np.empty((n, 3))
    lib().o_distance_segment_sphere(C.c_size_t(n), _p(a0), _p(a1), _p(c), _p(r), _p(dist), _p(cp), _p(t), _p(sep))
    return dist, cp, t, sep


def contact_spheres(pairs, center, radius, box=None, fast=False):
    pairs = np.ascontiguousarray(pairs, dtype=np.int32)
    center, radius = _f(center), _f(radius)
    box = None if box is None else _f(box)
    c = len(pairs)
    sep, normal = np.empty(c), np.empty((c, 3))
    lib(fast).o_contact_spheres(C.c_size_t(c), _p(pairs), _p(center), _p(radius), _p(box), _p(sep), _p(normal))
    return sep, normal


def contact_spherocylinders(pairs, seg, center, fast=False, box=None):
    """box: orthorhombic periodic box (3 edge lengths): rod j at the nearest image of its centre"""
    pairs = np.ascontiguousarray(pairs, dtype=np.int32)
    seg, center = _f(seg), _f(center)
    box = None if box is None else _f(box)
    c = len(pairs)
    out = dict(sep=np.empty(c), normal=np.empty((c, 3)), cp1=np.empty((c, 3)), cp2=np.empty((c, 3)),
               ra=np.empty((c, 3)), rb=np.empty((c, 3)), s=np.empty(c), t=np.empty(c))
    lib(fast).o_contact_spherocylinders(C.c_size_t(c), _p(pairs), _p(seg), _p(center), _p(box), _p(out["sep"]),
                                        _p(out["normal"]), _p(out["cp1"]), _p(out["cp2"]), _p(out["ra"]),
                                        _p(out["rb"]), _p(out["s"]), _p(out["t"]))
    return out


def integrate_euler(dt, vel, center, quat=None):
    """x += dt U; q <- rotate_quaternion(q, W, dt); returns new (center, quat)"""
    vel, c = _f(vel), _f(center).copy()
    q = None if quat is None else _f(quat).copy()
    lib().o_integrate_euler(C.c_size_t(len(c)), C.c_double(dt), _p(vel), _p(c), _p(q))
    return c, q


# ---- periodicity --------------------------------------------------------------------------------------------------
def periodic_sep(box, p1, p2):
    box, p1, p2 = _f(box), _f(p1), _f(p2)
    out = np.empty_like(p1)
    lib().o_periodic_sep(C.c_size_t(len(p1)), _p(box), _p(p1), _p(p2), _p(out))
    return out


def periodic_wrap(box, p):
    box, p = _f(box), _f(p)
    out = np.empty_like(p)
    lib().o_periodic_wrap(C.c_size_t(len(p)), _p(box), _p(p), _p(out))
    return out


def unit_cell_inverse(h):
    """math::inverse of the 3x3 unit-cell matrix, the reference's cofactor formula (Matrix.hpp:1596-1601)"""
    h = _f(np.asarray(h, dtype=np.float64).reshape(3, 3))
    out = np.empty((3, 3))
    lib().o_unit_cell_inverse(_p(h), _p(out))
    return out


def periodic_sep_triclinic(h, p1, p2):
    """PeriodicMetric::sep (periodicity.hpp:304-307); h = 3x3 unit cell, lattice vectors as columns"""
    h, p1, p2 = _f(np.asarray(h, dtype=np.float64).reshape(3, 3)), _f(p1), _f(p2)
    out = np.empty_like(p1)
    lib().o_periodic_sep_triclinic(C.c_size_t(len(p1)), _p(h), _p(p1), _p(p2), _p(out))
    return out


def periodic_wrap_triclinic(h, p):
    h, p = _f(np.asarray(h, dtype=np.float64).reshape(3, 3)), _f(p)
    out = np.empty_like(p)
    lib().o_periodic_wrap_triclinic(C.c_size_t(len(p)), _p(h), _p(p), _p(out))
    return out


def shift_image_triclinic(h, p, images):
    h, p = _f(np.asarray(h, dtype=np.float64).reshape(3, 3)), _f(p)
    images = np.ascontiguousarray(images, dtype=np.int32)
    out = np.empty_like(p)
    lib().o_shift_image_triclinic(C.c_size_t(len(p)), _p(h), _p(p), _p(images), _p(out))
    return out


# ---- neighbour search ---------------------------------------------------------------------------------------------
SEARCH_SPHERES, SEARCH_AABB = 0, 1


def grow(aabb, bounding_radius, buffer):
    """search volumes: AABB grown by `buffer` on every face, bounding sphere radius + buffer
    (mundy/mesh/src/mundy_mesh/GenNeighborLinkers.hpp:579-583)"""
    aabb = _f(aabb)
    lo = np.ascontiguousarray(aabb[:, :3] - buffer)
    hi = np.ascontiguousarray(aabb[:, 3:] + buffer)
    return lo, hi, _f(bounding_radius) + buffer


def search(kind, lo, hi, center, R, box=None, symmetric=False, method="cell", fast=False):
    lo, hi, center, R = _f(lo), _f(hi), _f(center), _f(R)
    box = None if box is None else _f(box)
    L = lib(fast)
    if box is not None and box.shape == (3, 3):   # triclinic unit cell (lattice vectors as columns): brute force only
        L.o_search_triclinic.restype = C.c_size_t
        cnt = L.o_search_triclinic(C.c_int(kind), C.c_size_t(len(R)), _p(lo), _p(hi), _p(center), _p(R), _p(box),
                                   C.c_int(1 if symmetric else 0))
        pairs = np.empty((cnt, 2), dtype=np.int32)
        L.o_search_fetch(_p(pairs))
        return pairs
    L.o_search.restype = C.c_size_t
    cnt = L.o_search(C.c_int(kind), C.c_int(0 if method == "brute" else 1), C.c_size_t(len(R)), _p(lo), _p(hi),
                     _p(center), _p(R), _p(box), C.c_int(1 if symmetric else 0))
    pairs = np.empty((cnt, 2), dtype=np.int32)
    L.o_search_fetch(_p(pairs))
    return pairs


def moved_too_much(c_new, c_old, buffer):
    c_new, c_old = _f(c_new), _f(c_old)
    return bool(lib().o_moved_too_much(C.c_size_t(len(c_new)), _p(c_new), _p(c_old), C.c_double(buffer)))


# ---- convex -------------------------------------------------------------------------------------------------------
UNCONSTRAINED, LOWER_BOUND, UPPER_BOUND, BOUNDED = 0, 1, 2, 3
RESID_PROJECTED_DIFF, RESID_PROJECTED_GRADIENT = 0, 1


def axpby(alpha, x, beta, y):
    lib().o_axpby(C.c_size_t(len(x)), C.c_double(alpha), _p(x), C.c_double(beta), _p(y))


def wrapped_axpbyz(alpha, x, beta, y, z, space=(UNCONSTRAINED, 0.0, 0.0)):
    lib().o_wrapped_axpbyz(C.c_size_t(len(x)), C.c_double(alpha), _p(x), C.c_double(beta), _p(y), _p(z),
                           C.c_int(space[0]), C.c_double(space[1]), C.c_double(space[2]))


def diff_dot2(x, y):
    L = lib()
    L.o_diff_dot2.restype = C.c_double
    return L.o_diff_dot2(C.c_size_t(len(x)), _p(x), _p(y))


def diff_dot4(x1, x2, y1, y2):
    L = lib()
    L.o_diff_dot4.restype = C.c_double
    return L.o_diff_dot4(C.c_size_t(len(x1)), _p(x1), _p(x2), _p(y1), _p(y2))


def residual(kind, x, g, space):
    L = lib()
    L.o_residual.restype = C.c_double
    return L.o_residual(C.c_size_t(len(x)), C.c_int(kind), _p(x), _p(g), C.c_int(space[0]), C.c_double(space[1]),
                        C.c_double(space[2]))


def bb_step(x_old, g_old, x, g):
    L = lib()
    L.o_bb_step.restype = C.c_double
    return L.o_bb_step(C.c_size_t(len(x)), _p(x_old), _p(g_old), _p(x), _p(g))


def gemv(A, x):
    A, x = _f(A), _f(x)
    y = np.empty(len(x))
    lib().o_gemv(C.c_size_t(len(x)), _p(A), _p(x), _p(y))
    return y


def _result(it, res, conv):
    return dict(num_iters=int(it.value), residual=float(res.value), converged=bool(conv.value))


def solve_cqpp_dense(A, q, space, x0, resid_kind=RESID_PROJECTED_DIFF, max_iters=1000, tol=1e-8):
    A, q = _f(A), _f(q)
    n = len(q)
    x = _f(x0).copy()
    g, x_tmp, g_tmp = np.zeros(n), np.zeros(n), np.zeros(n)
    it, res, conv = C.c_uint(), C.c_double(), C.c_int()
    lib().o_solve_cqpp_dense(C.c_size_t(n), _p(A), _p(q), C.c_int(space[0]), C.c_double(space[1]),
                             C.c_double(space[2]), C.c_int(resid_kind), C.c_uint(max_iters), C.c_double(tol), _p(x),
                             _p(g), _p(x_tmp), _p(g_tmp), C.byref(it), C.byref(res), C.byref(conv))
    return x, g, _result(it, res, conv)


SUM_SERIAL, SUM_COMPENSATED = 0, 1


def set_sum_mode(mode, fast=False):
    """How the BB-step reductions and per-body sums are rounded (mundy_oracle.hpp, SumMode): SUM_SERIAL = plain serial
    sums (Kokkos-Serial), SUM_COMPENSATED = double-double pairs rounded once (what the device path does)."""
    lib(fast).o_set_sum_mode(C.c_int(int(mode)))


TRIG_LIBM, TRIG_SHARED = 0, 1


class shared_trig:
    """`with oracle.shared_trig():` -- sin / cos as the fixed operation sequence the device path evaluates
    (mundy_oracle.hpp, TrigMode) instead of libm, for the enclosed oracle calls (both builds)"""

    def __enter__(self):
        for fast in (False, True):
            lib(fast).o_set_trig_mode(C.c_int(TRIG_SHARED))
        return self

    def __exit__(self, *exc):
        for fast in (False, True):
            lib(fast).o_set_trig_mode(C.c_int(TRIG_LIBM))
        return False


class sphere_ellipsoid_minimiser_route:
    """`with oracle.sphere_ellipsoid_minimiser_route():` -- S-E of contact_mixed through distance(Point, Ellipsoid), the
    reference's own L-BFGS routine, instead of the exact closed form (the default), for the enclosed calls"""

    def __enter__(self):
        for fast in (False, True):
            lib(fast).o_set_sphere_ellipsoid_route(C.c_int(1))
        return self

    def __exit__(self, *exc):
        for fast in (False, True):
            lib(fast).o_set_sphere_ellipsoid_route(C.c_int(0))
        return False


def shared_sincos(x):
    x = _f(x)
    s, c = np.empty_like(x), np.empty_like(x)
    lib().o_shared_sincos(C.c_size_t(x.size), _p(x), _p(s), _p(c))
    return s, c


class compensated_sums:
    """`with oracle.compensated_sums():` -- order-independent sums for the enclosed oracle calls (both builds)."""

    def __enter__(self):
        for fast in (False, True):
            set_sum_mode(SUM_COMPENSATED, fast)
        return self

    def __exit__(self, *exc):
        for fast in (False, True):
            set_sum_mode(SUM_SERIAL, fast)
        return False


def _rod_args(rod, n_bodies):
    s, t, seg = rod
    seg = _f(seg)
    assert seg.shape == (n_bodies, 8), seg.shape
    return _f(s), _f(t), seg


def contact_op_apply(pairs, normal, ra, rb, mt, mr, dt, x, n_bodies, rod=None, body_velocity=False):
    """y = dt D^T M D x.  rod=(s, t, seg): the spherocylinder operator in rod-axis form (ContactOpRod) instead of the
    vector arms ra, rb; body_velocity=True also returns the (U, W) rows [N][6] (rod form only)."""
    pairs = np.ascontiguousarray(pairs, dtype=np.int32)
    y = np.empty(len(pairs))
    if rod is not None:
        s, t, seg = _rod_args(rod, n_bodies)
        vel = np.zeros((n_bodies, 6)) if body_velocity else None
        lib().o_contact_op_apply_rod(C.c_size_t(len(pairs)), C.c_size_t(n_bodies), _p(pairs), _p(_f(normal)), _p(s),
                                     _p(t), _p(seg), _p(_f(mt)), _p(_f(mr)), C.c_double(dt), _p(_f(x)), _p(y), _p(vel))
        return (y, vel) if body_velocity else y
    lib().o_contact_op_apply(C.c_size_t(len(pairs)), C.c_size_t(n_bodies), _p(pairs), _p(_f(normal)),
                             _p(None if ra is None else _f(ra)), _p(None if rb is None else _f(rb)), _p(_f(mt)),
                             _p(None if mr is None else _f(mr)), C.c_double(dt), _p(_f(x)), _p(y))
    return y


def solve_cqpp_contact(pairs, normal, ra, rb, mt, mr, dt, q, x0, space=(LOWER_BOUND, 0.0, 0.0),
                       resid_kind=RESID_PROJECTED_DIFF, max_iters=1000, tol=1e-8, threads=False, fast=False,
                       rod=None):
    """BBPGD (convex.hpp:614-666) on A = dt D^T M D (NgpLcp.cpp:442-548).  threads=True runs the OpenMP baseline.
    rod=(s, t, seg) solves with the rod-axis form of the spherocylinder operator (ContactOpRod, serial)."""
    pairs = np.ascontiguousarray(pairs, dtype=np.int32)
    if rod is not None:
        assert not threads
        mt, q = _f(mt), _f(q)
        s, t, seg = _rod_args(rod, len(mt))
        c = len(pairs)
        x = _f(x0).copy()
        g, x_tmp, g_tmp = np.zeros(c), np.zeros(c), np.zeros(c)
        it, res, conv = C.c_uint(), C.c_double(), C.c_int()
        lib(fast).o_solve_cqpp_contact_rod(
            C.c_size_t(c), C.c_size_t(len(mt)), _p(pairs), _p(_f(normal)), _p(s), _p(t), _p(seg), _p(mt), _p(_f(mr)),
            C.c_double(dt), _p(q), C.c_int(space[0]), C.c_double(space[1]), C.c_double(space[2]), C.c_int(resid_kind),
            C.c_uint(max_iters), C.c_double(tol), _p(x), _p(g), _p(x_tmp), _p(g_tmp), C.byref(it), C.byref(res),
            C.byref(conv))
        return x, g, _result(it, res, conv)
    normal, mt, q = _f(normal), _f(mt), _f(q)
    ra = None if ra is None else _f(ra)
    rb = None if rb is None else _f(rb)
    mr = None if mr is None else _f(mr)
    c = len(pairs)
    x = _f(x0).copy()
    g, x_tmp, g_tmp = np.zeros(c), np.zeros(c), np.zeros(c)
    it, res, conv = C.c_uint(), C.c_double(), C.c_int()
    L = lib(fast)
    fn = L.o_solve_cqpp_contact_mt if threads else L.o_solve_cqpp_contact
    fn(C.c_size_t(c), C.c_size_t(len(mt)), _p(pairs), _p(normal), _p(ra), _p(rb), _p(mt), _p(mr), C.c_double(dt),
       _p(q), C.c_int(space[0]), C.c_double(space[1]), C.c_double(space[2]), C.c_int(resid_kind),
       C.c_uint(max_iters), C.c_double(tol), _p(x), _p(g), _p(x_tmp), _p(g_tmp), C.byref(it), C.byref(res),
       C.byref(conv))
    return x, g, _result(it, res, conv)


def solve_friction_contact(pairs, normal, ra, rb, mt, mr, dt, sep, mu, p0=None, max_iters=10000, tol=1e-5,
                           method="bbpgd"):
    """BUILD EXTENSION, parity unpinned (the reference has no frictional solver): BBPGD (or, method="apgd", the
    accelerated projected gradient descent of Mazhar et al. 2015) on the cone complementarity problem with world-frame
    impulses p [C, 3]; returns (p, g, result)."""
    pairs = np.ascontiguousarray(pairs, dtype=np.int32)
    normal, ra, rb, mt, mr, sep = _f(normal), _f(ra), _f(rb), _f(mt), _f(mr), _f(sep)
    c = len(pairs)
    p = np.zeros((c, 3)) if p0 is None else _f(p0).copy()
    g = np.zeros((c, 3))
    it, res, conv = C.c_uint(), C.c_double(), C.c_int()
    if method not in ("bbpgd", "apgd"):
        raise ValueError("method must be 'bbpgd' or 'apgd'")
    fn = lib().o_solve_friction_contact if method == "bbpgd" else lib().o_solve_friction_contact_apgd
    fn(C.c_size_t(c), C.c_size_t(len(mt)), _p(pairs), _p(normal), _p(ra), _p(rb), _p(mt),
       _p(mr), C.c_double(dt), _p(sep), C.c_double(mu), C.c_uint(max_iters),
       C.c_double(tol), _p(p), _p(g), C.byref(it), C.byref(res), C.byref(conv))
    return p, g, _result(it, res, conv)


def project_cone(v, normal, mu):
    v, normal = _f(v), _f(normal)
    out = np.empty_like(v)
    lib().o_project_cone(C.c_size_t(len(v)), _p(v), _p(normal), C.c_double(mu), _p(out))
    return out


# ---- zmorton / hilbert ------------------------------------------------------------------------------------------
def float_xor_msb(p, q, single=False):
    L = lib()
    if single:
        return L.o_float_xor_msb_f32(C.c_float(p), C.c_float(q))
    return L.o_float_xor_msb_f64(C.c_double(p), C.c_double(q))


def float_exp(x, single=False):
    L = lib()
    return L.o_float_exp_f32(C.c_float(x)) if single else L.o_float_exp_f64(C.c_double(x))


def float_sig(x, single=False):
    L = lib()
    L.o_float_sig_f64.restype = C.c_uint64
    L.o_float_sig_f32.restype = C.c_uint32
    return L.o_float_sig_f32(C.c_float(x)) if single else L.o_float_sig_f64(C.c_double(x))


def uint_log_base2(x):
    return lib().o_uint_log_base2(C.c_uint64(x))


def zorder_less(p, q):
    p, q = _f(p), _f(q)
    return bool(lib().o_zorder_less_f64(_p(p), _p(q), C.c_int(len(p))))


def zmorton_less(p, q):
    p, q = _f(p), _f(q)
    return bool(lib().o_zmorton_less(_p(p), _p(q)))


def zorder_argsort(pts):
    pts = np.ascontiguousarray(pts)
    n, d = pts.shape
    order = np.empty(n, dtype=np.int64)
    if pts.dtype == np.float32:
        lib().o_zorder_argsort_f32(C.c_size_t(n), C.c_int(d), _p(pts), _p(order))
    else:
        pts = _f(pts)
        lib().o_zorder_argsort_f64(C.c_size_t(n), C.c_int(d), _p(pts), _p(order))
    return order


def hilbert_positions_and_directors(num_points, orientation=(1.0, 0.0, 0.0), side=1.0):
    L = lib()
    L.o_hilbert_num_positions.restype = C.c_size_t
    m = L.o_hilbert_num_positions(C.c_size_t(num_points))
    pos, dirs = np.empty((m, 3)), np.empty((m - 1, 3))
    L.o_hilbert_positions_and_directors(C.c_size_t(num_points), _p(_f(orientation)), C.c_double(side), _p(pos),
                                        _p(dirs))
    return pos, dirs


def hilbert_3d(s, cur=(0.0, 0.0, 0.0), dr1=(1.0, 0.0, 0.0), dr2=(0.0, 1.0, 0.0), dr3=(0.0, 0.0, 1.0)):
    pos = np.empty((s * s * s, 3))
    lib().o_hilbert_3d(C.c_size_t(s), _p(_f(cur)), _p(_f(dr1)), _p(_f(dr2)), _p(_f(dr3)), _p(pos))
    return pos


# ---- ellipsoids / minimize ------------------------------------------------------------------------------------------
def distance_ellipsoid_ellipsoid(c1, q1, r1, c2, q2, r2, fast=False):
    c1, q1, r1, c2, q2, r2 = _f(c1), _f(q1), _f(r1), _f(c2), _f(q2), _f(r2)
    n = len(c1)
    out = dict(dist=np.empty(n), cp1=np.empty((n, 3)), cp2=np.empty((n, 3)), n1=np.empty((n, 3)), n2=np.empty((n, 3)))
    lib(fast).o_distance_ellipsoid_ellipsoid(C.c_size_t(n), _p(c1), _p(q1), _p(r1), _p(c2), _p(q2), _p(r2),
                                             _p(out["dist"]), _p(out["cp1"]), _p(out["cp2"]), _p(out["n1"]),
                                             _p(out["n2"]))
    return out


def distance_point_ellipsoid(p, c, q, r):
    p, c, q, r = _f(p), _f(c), _f(q), _f(r)
    n = len(p)
    dist, cp, nrm = np.empty(n), np.empty((n, 3)), np.empty((n, 3))
    lib().o_distance_point_ellipsoid(C.c_size_t(n), _p(p), _p(c), _p(q), _p(r), _p(dist), _p(cp), _p(nrm))
    return dist, cp, nrm


def minimize_test(kind, x0):
    x = _f(x0).copy()
    L = lib()
    L.o_minimize_test.restype = C.c_double
    return L.o_minimize_test(C.c_int(kind), _p(x)), x


def scrap_resolve_collisions(pairs, normal, ra, rb, mt, mr, dt, sep, lam0, max_allowable_overlap=1e-5, max_iters=10000,
                             rod=None):
    """resolve_collisions of scrap/lcp_spheres/NgpLcp.cpp:558-759 (dry).  Returns (lam, g = sep + dt*sep_dot, result).
    rod=(s, t, seg): with the rod-axis form of the spherocylinder operator."""
    pairs = np.ascontiguousarray(pairs, dtype=np.int32)
    c = len(pairs)
    lam, g = _f(lam0).copy(), np.zeros(c)
    res, it, spd = C.c_double(), C.c_int(), C.c_double()
    mt = _f(mt)
    if rod is not None:
        s, t, seg = _rod_args(rod, len(mt))
        lib().o_scrap_resolve_collisions_rod(
            C.c_size_t(c), C.c_size_t(len(mt)), _p(pairs), _p(_f(normal)), _p(s), _p(t), _p(seg), _p(mt), _p(_f(mr)),
            C.c_double(dt), _p(_f(sep)), C.c_double(max_allowable_overlap), C.c_int(max_iters), _p(lam), _p(g),
            C.byref(res), C.byref(it), C.byref(spd))
        return lam, g, dict(max_abs_projected_sep=res.value, ite_count=it.value, max_speed=spd.value)
    lib().o_scrap_resolve_collisions(
        C.c_size_t(c), C.c_size_t(len(mt)), _p(pairs), _p(_f(normal)), _p(None if ra is None else _f(ra)),
        _p(None if rb is None else _f(rb)), _p(mt), _p(None if mr is None else _f(mr)), C.c_double(dt), _p(_f(sep)),
        C.c_double(max_allowable_overlap), C.c_int(max_iters), _p(lam), _p(g), C.byref(res), C.byref(it), C.byref(spd))
    return lam, g, dict(max_abs_projected_sep=res.value, ite_count=it.value, max_speed=spd.value)


def solve_small_cqpp_batch(A, q, space, x0, resid_kind=RESID_PROJECTED_DIFF, max_iters=1000, tol=1e-8):
    """MundyMathBackend (convex.hpp:288-350): a batch of independent n x n problems, one after the other."""
    A, q = _f(A), _f(q)
    b, n = q.shape
    x = _f(x0).copy()
    g = np.zeros((b, n))
    it, res, conv = np.zeros(b, np.uint32), np.zeros(b), np.zeros(b, np.int32)
    lib().o_solve_small_cqpp_batch(C.c_size_t(b), C.c_size_t(n), _p(A), _p(q), C.c_int(space[0]), C.c_double(space[1]),
                                   C.c_double(space[2]), C.c_int(resid_kind), C.c_uint(max_iters), C.c_double(tol),
                                   _p(x), _p(g), _p(it), _p(res), _p(conv))
    return x, g, it, res, conv.astype(bool)


# ---- mixed shapes -----------------------------------------------------------------------------------------------------
KIND_SPHERE, KIND_ROD, KIND_ELLIPSOID = 0, 1, 2


def aabb_mixed(kind, center, quat, shape, fast=False):
    kind = np.ascontiguousarray(kind, dtype=np.int32)
    center, quat, shape = _f(center), _f(quat), _f(shape)
    n = len(kind)
    aabb, brad = np.empty((n, 6)), np.empty(n)
    lib(fast).o_aabb_mixed(C.c_size_t(n), _p(kind), _p(center), _p(quat), _p(shape), _p(aabb), _p(brad))
    return aabb, brad


def contact_mixed(pairs, kind, center, quat, shape, fast=False, box=None):
    box = None if box is None else _f(box)
    pairs = np.ascontiguousarray(pairs, dtype=np.int32)
    kind = np.ascontiguousarray(kind, dtype=np.int32)
    center, quat, shape = _f(center), _f(quat), _f(shape)
    c = len(pairs)
    out = dict(sep=np.empty(c), normal=np.empty((c, 3)), cp1=np.empty((c, 3)), cp2=np.empty((c, 3)),
               ra=np.empty((c, 3)), rb=np.empty((c, 3)))
    lib(fast).o_contact_mixed(C.c_size_t(c), _p(pairs), _p(kind), _p(center), _p(quat), _p(shape), _p(box), _p(out["sep"]),
                              _p(out["normal"]), _p(out["cp1"]), _p(out["cp2"]), _p(out["ra"]), _p(out["rb"]))
    return out
