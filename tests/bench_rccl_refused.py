"""bench.py with the library's RCCL communicator constructor replaced by one that refuses (test infrastructure: the
way into bench.py's host-transport refusal without a switch in the product)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from mundy_amd import distributed as D  # noqa: E402


def _refuse(*a, **k):
    raise RuntimeError("RCCL refused (injected by tests/bench_rccl_refused.py)")


D._rccl_create = _refuse
bench.main()
