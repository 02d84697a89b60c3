"""mundy_amd -- MI355X (gfx950) implementation of MuNDy's per-timestep contact hot path.

Layout: csrc/ (hand-written HIP kernels + the C ABI of include/mundy_hip.h), capi.py (ctypes binding), ops.py
(host-side mirror of the reference interface on torch device tensors), pipeline.py (the timestep composition),
synth.py (synthetic inputs), build.py (hipcc driver).  There is no CPU path: without the HIP library and a GPU every
compute call raises.
"""
from . import capi  # noqa: F401

__all__ = ["capi", "ops", "pipeline", "synth", "build"]
