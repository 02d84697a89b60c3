"""Synthetic inputs for the contact hot path (numpy, host side).

The reference draws inputs from OpenRAND Philox keyed (seed, body index) (scrap/lcp_spheres/NgpLcp.cpp:810-833,
mundy_geom/randomize.hpp:56-300).  OpenRAND is not available, so this is an own counter-based generator with the same
keying: every value is a pure function of (seed, body index, stream), hence identical on every rank and for every
partition of the bodies.  Input streams are therefore not reproducible against OpenRAND (harmless: the same arrays
feed the CPU oracle and the GPU path).
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix(z):
    """splitmix64 finaliser on uint64 arrays"""
    with np.errstate(over="ignore"):
        z = (z + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def uniform01(seed, index, stream):
    """U[0,1) doubles, a pure function of (seed, index, stream); index is an integer array"""
    idx = np.asarray(index, dtype=np.uint64)
    with np.errstate(over="ignore"):
        key = _mix(np.uint64(seed) * np.uint64(0xD1342543DE82EF95) + np.uint64(stream))
        z = _mix(_mix(idx ^ key) + np.uint64(stream) * np.uint64(0x2545F4914F6CDD1D))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def box_edge(n, body_volume, volume_fraction):
    return float((n * body_volume / volume_fraction) ** (1.0 / 3.0))


def spheres(n, radius=1.0, volume_fraction=0.40, seed=1234, first=0):
    """n spheres, centres uniform in [0, L)^3 (generate_random_point, randomize.hpp:57-64); monodisperse r as
    NgpLcp.cpp:849.  Returns dict(center [n,3], radius [n], box L)."""
    idx = np.arange(first, first + n)
    L = box_edge(n, 4.0 / 3.0 * np.pi * radius ** 3, volume_fraction)
    c = np.stack([uniform01(seed, idx, s) * L for s in (0, 1, 2)], axis=1)
    return dict(center=np.ascontiguousarray(c), radius=np.full(n, float(radius)), box=L)


def spherocylinder_centers(indices, n_total, radius=0.5, length=2.0, volume_fraction=0.40, seed=1234):
    """centres only, for explicit global indices (used to order a large system before a rank materialises its slice)"""
    idx = np.asarray(indices)
    vol = np.pi * radius ** 2 * length + 4.0 / 3.0 * np.pi * radius ** 3
    L = box_edge(n_total, vol, volume_fraction)
    return np.ascontiguousarray(np.stack([uniform01(seed, idx, s) * L for s in (0, 1, 2)], axis=1)), L


def spherocylinders(n, radius=0.5, length=2.0, volume_fraction=0.40, seed=1234, first=0, n_total=None, indices=None):
    """n spherocylinders (centre, unit quaternion, radius, length).  Orientation: axis u uniform on the sphere,
    q = quat_from_parallel_transport(zhat, u) (randomize.hpp:79-84, Quaternion.hpp:1489-1505).  Body volume
    pi r^2 L + 4/3 pi r^3; the box edge comes from n_total (default n) so shards of one system share a box."""
    idx = np.arange(first, first + n) if indices is None else np.asarray(indices)
    n = len(idx)
    vol = np.pi * radius ** 2 * length + 4.0 / 3.0 * np.pi * radius ** 3
    L = box_edge(n if n_total is None else n_total, vol, volume_fraction)
    c = np.stack([uniform01(seed, idx, s) * L for s in (0, 1, 2)], axis=1)
    z = 2.0 * uniform01(seed, idx, 3) - 1.0
    phi = 2.0 * np.pi * uniform01(seed, idx, 4)
    z = np.clip(z, -1.0 + 1e-9, 1.0)  # u = -zhat makes the parallel-transport quaternion singular
    st = np.sqrt(1.0 - z * z)
    u = np.stack([st * np.cos(phi), st * np.sin(phi), z], axis=1)
    # quat_from_parallel_transport(zhat, u): w = sqrt((1 + z.u)/2), xyz = 0.5 (zhat x u) / w
    w = np.sqrt(0.5 * (1.0 + u[:, 2]))
    q = np.stack([w, 0.5 * (-u[:, 1]) / w, 0.5 * (u[:, 0]) / w, np.zeros(n)], axis=1)
    return dict(center=np.ascontiguousarray(c), quat=np.ascontiguousarray(q), radius=np.full(n, float(radius)),
                length=np.full(n, float(length)), box=L)


def aligned_spherocylinders(n, radius=0.5, length=2.0, volume_fraction=0.40, seed=1234, axis=(0.0, 0.0, 1.0)):
    """n PARALLEL spherocylinders (a nematic packing: every pair takes the colinear branch of the segment-segment
    distance, LineSegmentLineSegment.hpp:215-265 -- end-to-end pairs with an unclamped parameter, side-by-side pairs
    with overlapping projections).  Centres uniform in the box of the requested volume fraction; one common axis."""
    b = spherocylinders(n, radius, length, volume_fraction, seed)
    u = np.asarray(axis, dtype=np.float64)
    u = u / np.linalg.norm(u)
    # quat_from_parallel_transport(zhat, u)  (Quaternion.hpp:1489-1505)
    w = np.sqrt(0.5 * (1.0 + u[2]))
    q = np.array([w, 0.5 * (-u[1]) / w, 0.5 * u[0] / w, 0.0])
    b["quat"] = np.ascontiguousarray(np.tile(q, (n, 1)))
    return b


def mixed_bodies(n, volume_fraction=0.40, seed=1234, sphere_radius=0.5, rod=(0.5, 2.0), ellipsoid=(0.8, 0.5, 0.4)):
    """BASELINE configs[4]: n bodies, kind = index mod 3 (0 sphere, 1 spherocylinder, 2 ellipsoid), centres uniform
    in the box that gives the requested volume fraction, orientations uniform (unit quaternions from 4 normal
    deviates via Box-Muller on the counter-based stream).  shape rows: (r,0,0) / (r,L,0) / (r1,r2,r3)."""
    idx = np.arange(n)
    kind = (idx % 3).astype(np.int32)
    vols = np.array([4.0 / 3.0 * np.pi * sphere_radius ** 3,
                     np.pi * rod[0] ** 2 * rod[1] + 4.0 / 3.0 * np.pi * rod[0] ** 3,
                     4.0 / 3.0 * np.pi * ellipsoid[0] * ellipsoid[1] * ellipsoid[2]])
    L = float((vols[kind].sum() / volume_fraction) ** (1.0 / 3.0))
    c = np.stack([uniform01(seed, idx, s) * L for s in (0, 1, 2)], axis=1)
    u = [np.maximum(uniform01(seed, idx, 10 + s), 1e-300) for s in range(4)]
    g = np.stack([np.sqrt(-2 * np.log(u[0])) * np.cos(2 * np.pi * u[1]), np.sqrt(-2 * np.log(u[0])) * np.sin(2 * np.pi * u[1]),
                  np.sqrt(-2 * np.log(u[2])) * np.cos(2 * np.pi * u[3]), np.sqrt(-2 * np.log(u[2])) * np.sin(2 * np.pi * u[3])],
                 axis=1)
    q = g / np.linalg.norm(g, axis=1, keepdims=True)
    shape = np.zeros((n, 3))
    shape[kind == 0] = [sphere_radius, 0.0, 0.0]
    shape[kind == 1] = [rod[0], rod[1], 0.0]
    shape[kind == 2] = list(ellipsoid)
    return dict(kind=kind, center=np.ascontiguousarray(c), quat=np.ascontiguousarray(q), shape=shape, box=L)


def dry_mobility(radius, viscosity=1e-3, bounding_radius=None):
    """dry local drag (NgpLcp.cpp:484-486; Bacteria.cpp:810-848): U = F / (6 pi mu r), W = T / (8 pi mu r^3).
    For rods the effective radius is the bounding radius when given."""
    r = np.asarray(radius if bounding_radius is None else bounding_radius, dtype=np.float64)
    return 1.0 / (6.0 * np.pi * viscosity * r), 1.0 / (8.0 * np.pi * viscosity * r ** 3)
