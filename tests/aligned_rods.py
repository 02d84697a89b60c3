"""Aligned-rod inputs shared by the CPU and GPU tests of the rod-compressed contact operator.

Every pair of PARALLEL rods takes the colinear branch of the segment-segment distance
(mundy_geom/distance/LineSegmentLineSegment.hpp:215-265): the parameter it hands back is the UNCLAMPED one of
distance(Point, LineSegment) (PointLineSegment.hpp:156-166), while the contact point of the assembly is the clamped
closest point (scrap/.../SpherocylinderSpherocylinderLinker.cpp:246-247).  The operator form that keeps one scalar per
lever arm has to use the arclength of the contact point; the yardstick is the reference's own form with vector arms
ra = cp1 - c_i, rb = cp2 - c_j."""
import numpy as np


def problem(oracle, b, buffer=0.1):
    from mundy_amd import synth
    c = b["center"]
    aabb = oracle.compute_aabb_spherocylinders(c, b["quat"], b["radius"], b["length"])
    brad = oracle.bounding_radius_spherocylinders(b["radius"], b["length"])
    lo, hi, R = oracle.grow(aabb, brad, buffer)
    pairs = oracle.search(1, lo, hi, c, R)
    seg = oracle.spherocylinder_segments(c, b["quat"], b["radius"], b["length"])
    out = oracle.contact_spherocylinders(pairs, seg, c)
    mt, mr = synth.dry_mobility(b["radius"], bounding_radius=brad)
    return dict(N=len(c), pairs=pairs, seg=seg, mt=mt, mr=mr, center=c, **out)


def two_rods():
    """the round-2 review's case: r = 0.5, L = 2, both along z, centres (0,0,0) and (0.8,0,2.5): the second rod's raw
    parameter is -0.25, its contact point is the end p0 (arm (0,0,-1), not (0,0,-1.5))"""
    return dict(center=np.array([[0.0, 0.0, 0.0], [0.8, 0.0, 2.5]]), quat=np.tile([1.0, 0.0, 0.0, 0.0], (2, 1)),
                radius=np.full(2, 0.5), length=np.full(2, 2.0))


def nematic(n, seed, axis=(0.0, 0.0, 1.0)):
    from mundy_amd import synth
    return synth.aligned_spherocylinders(n, seed=seed, axis=axis)


def half_nematic(n, seed):
    """every second rod along z, the others uniformly oriented: both branches of the distance routine in one list"""
    from mundy_amd import synth
    b = synth.spherocylinders(n, seed=seed)
    b["quat"][::2] = [1.0, 0.0, 0.0, 0.0]
    return b


def body_velocity_vector_form(P, lam, dt=None):
    """(U, W) rows of the reference's force scatter with vector arms (NgpLcp.cpp:442-486 plus r x f torques), numpy"""
    N = P["N"]
    f = lam[:, None] * P["normal"]
    F = np.zeros((N, 3))
    T = np.zeros((N, 3))
    i, j = P["pairs"][:, 0], P["pairs"][:, 1]
    np.add.at(F, i, -f)
    np.add.at(F, j, f)
    np.add.at(T, i, np.cross(P["ra"], -f))
    np.add.at(T, j, np.cross(P["rb"], f))
    return np.concatenate([P["mt"][:, None] * F, P["mr"][:, None] * T], axis=1)
