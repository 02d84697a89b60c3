import numpy as np
import torch


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def host(t):
    return t.detach().cpu().numpy()


def random_rods(rng, n, box, rmin=0.3, rmax=0.6, lmin=0.5, lmax=2.5):
    c = rng.uniform(0, box, (n, 3))
    q = rng.normal(size=(n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    return c, q, rng.uniform(rmin, rmax, n), rng.uniform(lmin, lmax, n)


def assert_bits_equal(a, b, what=""):
    """bit-for-bit equality of float64 arrays (NaN == NaN when the payloads match)"""
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    bad = a.view(np.uint64) != b.view(np.uint64)
    # +0.0 vs -0.0 would show up here too, deliberately
    assert not bad.any(), "%s: %d of %d elements differ; max |diff| = %g" % (
        what, int(bad.sum()), a.size, float(np.nanmax(np.abs(a - b)[bad])) if bad.any() else 0.0)
