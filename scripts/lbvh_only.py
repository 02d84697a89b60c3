import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mundy_amd import ops, synth
n = 1_000_000
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
b = synth.spherocylinders(n)
c = dev(b["center"]); perm = ops.morton_order(c, [0.0, 0.0, 0.0], 3.0).long()
c, q, r, L = (t[perm].contiguous() for t in (c, dev(b["quat"]), dev(b["radius"]), dev(b["length"])))
aabb, brad = ops.compute_aabb_spherocylinders(c, q, r, L), ops.bounding_radius_spherocylinders(r, L)
links = ops.GenNeighborLinks().set_search_buffer(0.1).set_search_kind(ops.SEARCH_AABB).set_search_method(ops.SEARCH_METHOD_MORTON_LBVH).concretize()
for _ in range(6):
    links.generate(aabb, c, brad, force=True)
torch.cuda.synchronize()
print(links.num_pairs)
