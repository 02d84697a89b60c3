"""CPU-side checks of the drop-in boundary: libmundy_hip.so builds (hipcc cross-compiles gfx950 without a GPU), loads,
and exports every symbol include/mundy_hip.h declares; argument validation that happens before any HIP call maps to the
reference's exception types.  No compute entry point is called here."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from mundy_amd import build, capi
    build.build()
    return capi.load()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "mundy_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mhip_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound(lib):
    from mundy_amd import capi
    names = declared_symbols()
    assert len(names) >= 40
    bound = set(capi.SIGNATURES) | set(capi.OTHER_SYMBOLS)
    for name in names:
        assert hasattr(lib, name), "libmundy_hip.so does not export %s" % name
        assert name in bound, "%s is declared in mundy_hip.h but has no ctypes binding" % name
    assert bound <= set(names), "bindings without a header declaration: %s" % (bound - set(names))


def test_version_and_error_channel(lib):
    assert lib.mhip_version() >= 100
    assert isinstance(lib.mhip_last_error(), bytes)


def test_missing_library_is_loud(monkeypatch, tmp_path):
    from mundy_amd import capi
    monkeypatch.setattr(capi, "_lib", None)
    monkeypatch.setattr(capi, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(capi.MhipError, match="no CPU fallback"):
        capi.load()


def test_validation_before_any_hip_call(lib):
    from mundy_amd import capi
    # null handle -> std::invalid_argument
    with pytest.raises(ValueError):
        capi.check(lib.mhip_broadphase_create(None))
    h = C.c_void_p()
    capi.check(lib.mhip_broadphase_create(C.byref(h)))
    cnt = C.c_size_t()
    cfg = capi.BroadphaseConfig(7, 0, 0.0, 0, (C.c_double * 3)(0, 0, 0))
    with pytest.raises(ValueError, match="unknown search kind"):
        capi.check(lib.mhip_broadphase_build(h, C.byref(cfg), 0, None, None, None, C.byref(cnt), None))
    cfg = capi.BroadphaseConfig(capi.SEARCH_AABB, 0, -1.0, 0, (C.c_double * 3)(0, 0, 0))
    with pytest.raises(ValueError, match="buffer"):
        capi.check(lib.mhip_broadphase_build(h, C.byref(cfg), 0, None, None, None, C.byref(cnt), None))
    with pytest.raises(RuntimeError, match="before get_pairs"):
        capi.check(lib.mhip_broadphase_get_pairs(h, None, None, None, None))
    capi.check(lib.mhip_broadphase_destroy(h))
    # unknown convex space / residual kind
    sp = capi.Space(9, 0.0, 0.0)
    with pytest.raises(ValueError, match="convex space"):
        capi.check(lib.mhip_wrapped_axpbyz(0, 1.0, None, 1.0, None, None, C.byref(sp), None))
    # rigid-body operator needs ra, rb and mob_rot together
    op = C.c_void_p()
    with pytest.raises(ValueError, match="given together"):
        capi.check(lib.mhip_contact_op_create(C.byref(op), 0, 0, None, None, C.c_void_p(8), None, None, None, 1.0, None, None))


def test_communicator_validation_before_any_hip_call(lib):
    from mundy_amd import capi
    h = C.c_void_p()
    no_x, no_g = capi.EXCHANGE_FN(0), capi.ALL_GATHER_FN(0)
    with pytest.raises(ValueError, match="rank 2 of 2"):
        capi.check(lib.mhip_comm_create_host(C.byref(h), 2, 2, no_x, no_g, None))
    with pytest.raises(ValueError, match="needs both callbacks"):
        capi.check(lib.mhip_comm_create_host(C.byref(h), 0, 2, no_x, no_g, None))
    capi.check(lib.mhip_comm_create_host(C.byref(h), 0, 1, no_x, no_g, None))   # one rank needs no message layer
    rank, world, is_rccl = C.c_int(-1), C.c_int(-1), C.c_int(-1)
    capi.check(lib.mhip_comm_info(h, C.byref(rank), C.byref(world), C.byref(is_rccl)))
    assert (rank.value, world.value, is_rccl.value) == (0, 1, 0)
    with pytest.raises(RuntimeError, match="mhip_ghost_plan has not been called"):
        capi.check(lib.mhip_ghost_exchange(h, 12, None, None, None))
    with pytest.raises(RuntimeError, match="no exchange in flight"):
        capi.check(lib.mhip_comm_exchange_finish(h, None))
    with pytest.raises(ValueError, match="null argument"):
        capi.check(lib.mhip_bbpgd_solve_contact_distributed(None, h, None, 0, None, None, None, None, None, None, None,
                                                            32, None, None, None))
    capi.check(lib.mhip_comm_destroy(h))
    with pytest.raises(ValueError, match="strides"):
        capi.check(lib.mhip_copy_strided(4, 3, None, 2, None, 3, None))


def test_gen_neighbor_links_builder_misuse():
    # the builder's error behaviour (GenNeighborLinkers.hpp:401-511): std::runtime_error on misuse
    from mundy_amd import build, ops
    build.build()
    g = ops.GenNeighborLinks().set_search_buffer(0.5)
    with pytest.raises(RuntimeError, match="before concretization"):
        g.generate(None, None, None)
    g.concretize()
    with pytest.raises(RuntimeError, match="after concretization"):
        g.set_search_buffer(1.0)
    with pytest.raises(RuntimeError, match="more than once"):
        g.concretize()
    g.close()


def test_library_reads_no_tuning_switch_from_the_environment(lib):
    # the product has no run-time tuning switches -- the only environment variable the library looks at is
    # MHIP_TRACE (roctx ranges on / off); A/B macros exist at compile time only (MHIP_EXTRA_HIPCC_FLAGS of the build)
    src = os.path.join(ROOT, "mundy_amd", "csrc")
    calls = []
    for f in sorted(os.listdir(src)):
        text = open(os.path.join(src, f)).read()
        calls += [(f, m) for m in re.findall(r"getenv\(\s*\"([^\"]+)\"", text)]
    assert calls == [("runtime.hip", "MHIP_TRACE")], calls
