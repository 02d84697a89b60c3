import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mundy_amd import ops
rng = np.random.default_rng(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
def ell():
    c = rng.uniform(0, 4, (n, 3)); q = rng.normal(size=(n, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    return dev(c), dev(q), dev(rng.uniform(0.4, 1.0, (n, 3)))
a, b = ell(), ell()
ops.distance_ellipsoid_ellipsoid(*a, *b); torch.cuda.synchronize()
ts = []
for _ in range(5):
    t = time.perf_counter(); ops.distance_ellipsoid_ellipsoid(*a, *b); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
dt = float(np.median(ts))
ev = ops.ellipsoid_last_evaluations()
print("ellipsoid pairs %d: %.4f s  -> %.3f us/pair, %.3g pairs/s; %.0f objective evaluations per pair" % (n, dt, 1e6 * dt / n, n / dt, ev / n))
