"""Neighbour-list build time on the bench input (10^6 rods, AABB + 0.1, unique pairs), on 10^6 spheres and on a
size-disperse sphere system, on both search structures (cell grid / Morton LBVH).  Bodies in Z order, as in the step."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mundy_amd import ops, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
def zorder(c, *arrs, cell=3.0):
    perm = ops.morton_order(c, [0.0, 0.0, 0.0], cell).long()
    return [c[perm].contiguous()] + [a[perm].contiguous() for a in arrs]


for name in ("rods", "spheres", "disperse spheres", "periodic spheres", "periodic disperse"):
    if name == "rods":
        b = synth.spherocylinders(n)
        c, q, r, L = zorder(dev(b["center"]), dev(b["quat"]), dev(b["radius"]), dev(b["length"]))
        aabb, brad = ops.compute_aabb_spherocylinders(c, q, r, L), ops.bounding_radius_spherocylinders(r, L)
        kind = ops.SEARCH_AABB
    elif name in ("spheres", "periodic spheres"):
        b = synth.spheres(n)
        pbox = [b["box"]] * 3
        c, r = zorder(dev(b["center"]), dev(b["radius"]))
        aabb, brad = ops.compute_aabb_spheres(c, r), r
        kind = ops.SEARCH_SPHERES
    else:
        rng = np.random.default_rng(3)
        m = n // 10
        rr = np.exp(rng.normal(0.0, 0.8, m)) * 0.3
        box = (4.0 / 3.0 * np.pi * (rr ** 3).sum() / 0.30) ** (1.0 / 3.0)
        c, r = zorder(dev(rng.uniform(0, box, (m, 3))), dev(rr), cell=box / 64)
        aabb, brad = ops.compute_aabb_spheres(c, r), r
        kind = ops.SEARCH_SPHERES
        pbox = [box] * 3
    for mname, method in (("grid", ops.SEARCH_METHOD_GRID), ("lbvh", ops.SEARCH_METHOD_MORTON_LBVH), ("auto", ops.SEARCH_METHOD_AUTO)):
        links = ops.GenNeighborLinks().set_search_buffer(0.1).set_search_kind(kind).set_search_method(method)
        if name.startswith("periodic"):
            links = links.set_periodic_box(pbox)
        links = links.concretize()
        links.generate(aabb, c, brad, force=True)
        torch.cuda.synchronize()
        ts = []
        for _ in range(10):
            t0 = time.perf_counter()
            links.generate(aabb, c, brad, force=True)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        ms = 1e3 * float(np.median(ts))
        print("%-16s %-5s (ran on %s): %8d bodies, %9d pairs, build %8.3f ms (median of 10) -> %.3g pairs/s"
              % (name, mname, {1: "grid", 2: "lbvh"}[links.method_used()], c.shape[0], links.num_pairs, ms,
                 links.num_pairs / ms * 1e3), flush=True)
        links.close()
