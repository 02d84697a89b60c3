"""Achievable HBM rate on this box with this library's own streaming kernels (SURVEY 8d asks for a measured STREAM-like
ceiling next to the 8 TB/s spec): deep_copy (16 B/element), axpby y = a x + b y (24 B), wrapped_axpbyz z = P(a x + b y)
(24 B) on 2^27-element fp64 vectors (1 GiB each, far beyond the 256 MiB Infinity Cache), HIP-event timed."""
import sys
import torch
sys.path.insert(0, ".")
from mundy_amd import capi, ops

n = 1 << 27
x = torch.rand(n, dtype=torch.float64, device="cuda")
y = torch.rand(n, dtype=torch.float64, device="cuda")
z = torch.empty_like(x)
lib = capi.load()


def timed(fn, bytes_per_elem, label, reps=20):
    for _ in range(5):
        fn()
    best = None
    for _ in range(4):   # best of four batches: the clocks of a fresh box take a while to settle
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        t = a.elapsed_time(b) / reps
        best = t if best is None else min(best, t)
    ms = best
    print("%-28s %.3f ms  %.0f GB/s" % (label, ms, bytes_per_elem * n / ms / 1e6), flush=True)


timed(lambda: capi.check(lib.mhip_deep_copy(n, ops._ptr(z), ops._ptr(x), None)), 16, "deep_copy (8r + 8w)")
timed(lambda: ops.axpby(1.5, x, 0.5, y), 24, "axpby (16r + 8w)")
timed(lambda: ops.wrapped_axpbyz(1.0, x, -0.25, y, z, (ops.SPACE_LOWER_BOUND, 0.0, 0.0)), 24, "wrapped_axpbyz (16r + 8w)")
timed(lambda: z.copy_(x), 16, "torch copy_ (8r + 8w)")
timed(lambda: torch.add(x, y, out=z), 24, "torch add (16r + 8w)")
timed(lambda: ops.diff_dot(x, y), 16, "diff_dot(x, y) (16r, host sync)", reps=10)
timed(lambda: ops.diff_dot(x, y, z, x), 32, "diff_dot(x1,x2,y1,y2) (32r)", reps=10)
timed(lambda: ops.residual(0, x, y, (ops.SPACE_LOWER_BOUND, 0.0, 0.0)), 16, "residual (16r)", reps=10)
