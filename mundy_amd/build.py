"""Builds libmundy_hip.so (hand-written HIP for gfx950) in-tree with hipcc.  No CPU fallback is ever built."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib", "libmundy_hip.so")
SOURCES = ["runtime.hip", "sort.hip", "geometry.hip", "broadphase.hip", "convex.hip", "reorder.hip", "halo.hip", "ellipsoid.hip", "mixed.hip", "mixed_fma.hip", "dist.hip"]
# per-source flag substitutions: the contracted build of the ellipsoid minimisation classes (see mixed_fma.hip)
FLAG_OVERRIDES = {"mixed_fma.hip": {"-ffp-contract=off": "-ffp-contract=fast"}}
HEADERS = ["mhip_internal.hpp", "geom_device.hpp", "ellipsoid_device.hpp", "ellipsoid_lockstep.hpp", "segment_ellipsoid.hpp", os.path.join("..", "..", "include", "mundy_hip.h")]
# -ffp-contract=off: a*b+c stays two roundings so per-element results are bit-identical to the scalar reference order
FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"]


def _hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


STAMP = LIB + ".flags"


def _flag_string():
    return " ".join(FLAGS + os.environ.get("MHIP_EXTRA_HIPCC_FLAGS", "").split())


def sweep_kernels_stamp():
    """Identifies the code of the two BBPGD sweeps as built: hash of the sources they are compiled from + the flag
    string.  profiles/traffic*.json (PMC traffic per launch, measured by scripts/profile_bench.sh) carry it, and
    bench.py prints `traffic: null` when the running library's differs -- a figure measured on other kernels is not a
    measurement of these."""
    import hashlib
    h = hashlib.sha256()
    for f in ("convex.hip", "mhip_internal.hpp"):
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    h.update(_flag_string().encode())
    return h.hexdigest()[:16]


def is_stale():
    if not os.path.exists(LIB):
        return True
    # a library built with other flags (an A/B build with -D tuning macros) is stale whatever its age
    if not os.path.exists(STAMP) or open(STAMP).read() != _flag_string():
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not is_stale():
        return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    objdir = os.path.join(HERE, "lib", "obj")
    os.makedirs(objdir, exist_ok=True)
    procs = []
    objs = []
    for src in SOURCES:
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(obj)
        extra = os.environ.get("MHIP_EXTRA_HIPCC_FLAGS", "").split()  # A/B builds of tuning macros only
        flags = [FLAG_OVERRIDES.get(src, {}).get(f, f) for f in FLAGS]
        cmd = [_hipcc()] + flags + extra + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if out.strip():
            print(out, file=sys.stderr)
        if p.returncode != 0:
            failed = True
            print("hipcc failed on %s" % src, file=sys.stderr)
    if failed:
        raise RuntimeError("libmundy_hip.so build failed")
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    subprocess.check_call(cmd)
    with open(STAMP, "w") as f:
        f.write(_flag_string())
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
