"""Lockstep vs nested-loop ellipsoid kernels: bitwise comparison of every output and timing of both."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from mundy_amd import ops
rng = np.random.default_rng(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()


def ell(k):
    c = rng.uniform(0, 4, (k, 3)); q = rng.normal(size=(k, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    return dev(c), dev(q), dev(rng.uniform(0.4, 1.0, (k, 3)))


a, b = ell(n), ell(n)
res = {}
for mode in ("nested", "lockstep"):
    if mode == "nested":
        os.environ["MHIP_ELLIPSOID_NESTED"] = "1"
    else:
        os.environ.pop("MHIP_ELLIPSOID_NESTED", None)
    ops.distance_ellipsoid_ellipsoid(*a, *b); torch.cuda.synchronize()
    t = time.perf_counter(); out = ops.distance_ellipsoid_ellipsoid(*a, *b); torch.cuda.synchronize(); dt = time.perf_counter() - t
    res[mode] = out
    print("%-8s %d pairs: %.3f s -> %.3f us/pair, %.3g pairs/s" % (mode, n, dt, 1e6 * dt / n, n / dt), flush=True)
same = all(torch.equal(x, y) for x, y in zip(res["nested"].values(), res["lockstep"].values())) if isinstance(res["nested"], dict) \
    else all(torch.equal(x, y) for x, y in zip(res["nested"], res["lockstep"]))
print("bitwise identical outputs:", same)
sys.exit(0 if same else 1)
