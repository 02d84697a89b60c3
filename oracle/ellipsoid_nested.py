"""Test infrastructure (part of the oracle: restated reference code lives under oracle/ only): builds and wraps
oracle/ellipsoid_nested_ref.hip, the ellipsoid distances with the reference's minimiser written as plain nested loops
(one thread per pair; minimize_impl.hpp:151-605).  The production kernels run the same arithmetic as a
per-lane state machine; the GPU tests require the two to agree bit for bit.  Not imported by mundy_amd/."""
import ctypes as C
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "ellipsoid_nested_ref.hip")
LIB = os.path.join(HERE, "libellipsoid_nested_ref.so")
DEPS = [SRC] + [os.path.join(HERE, "..", "mundy_amd", "csrc", f)
                for f in ("ellipsoid_device.hpp", "geom_device.hpp", "mhip_internal.hpp")]
_lib = None


def build(force=False):
    stale = force or not os.path.exists(LIB) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in DEPS)
    if stale:
        hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
        subprocess.check_call([hipcc, "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-std=c++17", "-shared",
                               "-Wno-unused-function", SRC, "-o", LIB])
    return LIB


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
    return _lib


def _p(t):
    assert t.is_cuda and t.is_contiguous() and str(t.dtype) == "torch.float64"
    return C.c_void_p(t.data_ptr())


def distance_ellipsoid_ellipsoid(c1, q1, r1, c2, q2, r2):
    import torch
    n = c1.shape[0]
    out = dict(dist=torch.empty(n, dtype=torch.float64, device=c1.device))
    for k in ("cp1", "cp2", "n1", "n2"):
        out[k] = torch.empty((n, 3), dtype=torch.float64, device=c1.device)
    torch.cuda.synchronize()
    rc = lib().ref_distance_ellipsoid_ellipsoid(C.c_size_t(n), _p(c1), _p(q1), _p(r1), _p(c2), _p(q2), _p(r2),
                                                _p(out["dist"]), _p(out["cp1"]), _p(out["cp2"]), _p(out["n1"]), _p(out["n2"]))
    assert rc == 0, rc
    return out


def distance_point_ellipsoid(p, c, q, r):
    import torch
    n = c.shape[0]
    dist = torch.empty(n, dtype=torch.float64, device=c.device)
    cp, nrm = (torch.empty((n, 3), dtype=torch.float64, device=c.device) for _ in range(2))
    torch.cuda.synchronize()
    rc = lib().ref_distance_point_ellipsoid(C.c_size_t(n), _p(p), _p(c), _p(q), _p(r), _p(dist), _p(cp), _p(nrm))
    assert rc == 0, rc
    return dist, cp, nrm


def contact_rod_ellipsoid(rc_, rq, rshape, ec, eq, er):
    """rod (centre, quaternion, shape = (r, L, -)) against ellipsoid: (sep, normal = rod's outward normal, cp1, cp2)"""
    import torch
    n = rc_.shape[0]
    sep = torch.empty(n, dtype=torch.float64, device=rc_.device)
    normal, cp1, cp2 = (torch.empty((n, 3), dtype=torch.float64, device=rc_.device) for _ in range(3))
    torch.cuda.synchronize()
    rc = lib().ref_contact_rod_ellipsoid(C.c_size_t(n), _p(rc_), _p(rq), _p(rshape), _p(ec), _p(eq), _p(er), _p(sep),
                                         _p(normal), _p(cp1), _p(cp2))
    assert rc == 0, rc
    return sep, normal, cp1, cp2
