"""The C++ adapter (include/mundy_hip/adapter.hpp): compiles on the CPU box; its restatement of the reference's own
unit tests (tests/cpp/test_adapter.cpp) runs on the GPU box, linked against nothing but the C ABI."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "test_adapter")


def _build():
    from mundy_amd import build
    lib = build.build()
    libdir = os.path.dirname(lib)
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", os.path.join(ROOT, "tests", "cpp", "test_adapter.cpp"),
           "-I", os.path.join(ROOT, "include"), "-L", libdir, "-lmundy_hip", "-Wl,-rpath," + libdir,
           "-Wl,-rpath-link,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib", "-o", EXE]
    subprocess.check_call(cmd)
    return EXE


def _build_app():
    from mundy_amd import build
    libdir = os.path.dirname(build.build())
    exe = os.path.join(ROOT, "tests", "cpp", "ngp_lcp_app")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-Wextra", os.path.join(ROOT, "tests", "cpp", "ngp_lcp_app.cpp"),
                           "-I", os.path.join(ROOT, "include"), "-L", libdir, "-lmundy_hip", "-Wl,-rpath," + libdir,
                           "-Wl,-rpath-link,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib",
                           "-o", exe])
    return exe


def _build_rod_app():
    from mundy_amd import build
    libdir = os.path.dirname(build.build())
    exe = os.path.join(ROOT, "tests", "cpp", "rod_step_app")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-Wextra", os.path.join(ROOT, "tests", "cpp", "rod_step_app.cpp"),
                           "-I", os.path.join(ROOT, "include"), "-L", libdir, "-lmundy_hip", "-Wl,-rpath," + libdir,
                           "-Wl,-rpath-link,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib",
                           "-o", exe])
    return exe


def _build_dist_app():
    from mundy_amd import build
    libdir = os.path.dirname(build.build())
    exe = os.path.join(ROOT, "tests", "cpp", "rod_dist_app")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-Wextra", "-D__HIP_PLATFORM_AMD__",
                           os.path.join(ROOT, "tests", "cpp", "rod_dist_app.cpp"),
                           "-I", os.path.join(ROOT, "include"), "-I/opt/rocm/include", "-L", libdir, "-lmundy_hip",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath-link,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64",
                           "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    return exe


def test_adapter_header_compiles_and_links():
    assert os.path.exists(_build())
    assert os.path.exists(_build_app())
    assert os.path.exists(_build_rod_app())
    assert os.path.exists(_build_dist_app())


@pytest.mark.gpu
def test_cpp_distributed_stepper_over_rccl(tmp_path):
    # the domain-decomposed step from a C++ host program over the RCCL transport (one rank: the test box has one GPU
    # and RCCL refuses two ranks on a device): ghost plan, record exchange, partitioned operator, the distributed
    # BBPGD loop with its all-gather, integration -- three steps equal the single-GPU Python driver bit for bit
    import numpy as np
    import torch
    from mundy_amd import distributed as D, ops, pipeline, synth
    n = 20_000
    b = synth.spherocylinders(n, seed=11)
    order = D.hilbert_order(b["center"], 0.0, b["box"], level=5)
    c, q, r, ln = (np.ascontiguousarray(b[k][order]) for k in ("center", "quat", "radius", "length"))
    mt, mr = synth.dry_mobility(0.5 * ln + r)
    inp = tmp_path / "rods.bin"
    with open(inp, "wb") as f:
        f.write(np.uint64(n).tobytes())
        for a in (c, q, r, ln, mt, mr):
            f.write(np.ascontiguousarray(a, dtype=np.float64).tobytes())
    exe = _build_dist_app()
    p = subprocess.run([exe, str(inp), "3", "0", "1", str(tmp_path)], capture_output=True, text=True, timeout=600)
    print(p.stdout[-3000:], p.stderr[-2000:])
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    steps = [ln_.split() for ln_ in p.stdout.splitlines() if ln_.startswith("STEP")]
    assert len(steps) == 3
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    st = pipeline.ContactStepper("spherocylinder", dev(c), dev(r), dev(q), dev(ln), search_buffer=0.1,
                                 cfg=ops.PGDConfig(max_iters=10000, tol=1e-5), mob_trans=dev(mt), mob_rot=dev(mr))
    for k in range(3):
        s = st.step(force_rebuild=True)
        assert int(steps[k][5]) == s.num_contacts and int(steps[k][7]) == s.num_iters
        assert float(steps[k][9]) == s.residual and int(steps[k][11]) == int(s.converged)
        assert int(steps[k][13]) == 0 and int(steps[k][15]) == s.num_contacts      # one rank: no ghosts, all interior

    def checksum(a):
        h = 1469598103934665603
        for v in np.ascontiguousarray(a).view(np.uint64).ravel().tolist():
            h = ((h ^ v) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
        return "%016x" % h
    line = [ln_ for ln_ in p.stdout.splitlines() if ln_.startswith("CHECKSUM")][0].split()
    assert line[4] == checksum(st.center.cpu().numpy()) and line[6] == checksum(st.quat.cpu().numpy())

    # the same with the rebuild rule (across the one rank there is): list, partition and incidence index reused when
    # nobody moved more than half the buffer -- decisions, iteration counts and the final state equal the Python loop's
    p = subprocess.run([exe, str(inp), "10", "0", "1", str(tmp_path), "reuse"], capture_output=True, text=True, timeout=600)
    print(p.stdout[-3000:], p.stderr[-2000:])
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    steps = [ln_.split() for ln_ in p.stdout.splitlines() if ln_.startswith("STEP")]
    st = pipeline.ContactStepper("spherocylinder", dev(c), dev(r), dev(q), dev(ln), search_buffer=0.1,
                                 cfg=ops.PGDConfig(max_iters=10000, tol=1e-5), mob_trans=dev(mt), mob_rot=dev(mr))
    rebuilt = []
    for k in range(10):
        s = st.step()
        rebuilt.append(int(s.rebuilt))
        assert int(steps[k][5]) == s.num_contacts and int(steps[k][7]) == s.num_iters and int(steps[k][17]) == int(s.rebuilt)
        assert float(steps[k][9]) == s.residual
    assert 0 in rebuilt[1:] and rebuilt[0] == 1, rebuilt
    line = [ln_ for ln_ in p.stdout.splitlines() if ln_.startswith("CHECKSUM")][0].split()
    assert line[4] == checksum(st.center.cpu().numpy()) and line[6] == checksum(st.quat.cpu().numpy())

    # ownership follows the bodies (DistributedSpherocylinderStepper::rebalance: curve cut by work, migration plan and
    # exchange, re-sort by (cell, entity id)) -- on the one rank there is it re-orders the owned set at every rebuild;
    # the Python stepper drives the same library entry points, and its 3-rank trajectory is checked against a single
    # rank in test_gpu_distributed.py
    steps_n = 5
    p = subprocess.run([exe, str(inp), str(steps_n), "0", "1", str(tmp_path), "migrate", repr(float(b["box"]))],
                       capture_output=True, text=True, timeout=600)
    print(p.stdout[-3000:], p.stderr[-2000:])
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    steps = [ln_.split() for ln_ in p.stdout.splitlines() if ln_.startswith("STEP")]
    dst = D.DistributedContactStepper(dev(c), dev(q), dev(r), dev(ln), 0, cfg=ops.PGDConfig(max_iters=10000, tol=1e-5),
                                      domain=(0.0, float(b["box"])), curve_level=4, recut_every=3)
    for k in range(steps_n):
        s = dst.step(force_rebuild=True, migrate=True)
        assert int(steps[k][5]) == s["local_contacts"] and int(steps[k][7]) == s["num_iters"], (k, steps[k], s)
    line = [ln_ for ln_ in p.stdout.splitlines() if ln_.startswith("CHECKSUM")][0].split()
    assert int(line[10]) == n
    assert line[8] == checksum(dst.entity_id.cpu().numpy())       # the same bodies in the same (cell, id) order
    assert not np.array_equal(dst.entity_id.cpu().numpy(), np.arange(n, dtype=np.float64))   # ... and it is a new one
    assert line[4] == checksum(dst.center.cpu().numpy()) and line[6] == checksum(dst.quat.cpu().numpy())
    dst.op.close()


@pytest.mark.gpu
@pytest.mark.parametrize("periodic", [False, True])
def test_cpp_rod_stepper_reproduces_the_python_driver(tmp_path, periodic):
    # the headline hot path (spherocylinders, Z-order reorder, neighbour list, narrow phase, fused BBPGD, integration)
    # driven from a C++ host program (include/mundy_hip/stepper.hpp) with no Python in the process: same kernels in the
    # same order as mundy_amd/pipeline.py, so contacts, iteration counts and the final state agree bit for bit
    import numpy as np
    import torch
    from mundy_amd import ops, pipeline, synth
    n = 30_000
    b = synth.spherocylinders(n, seed=42)
    brad = 0.5 * b["length"] + b["radius"]
    mt, mr = synth.dry_mobility(brad)
    inp = tmp_path / "rods.bin"
    with open(inp, "wb") as f:
        f.write(np.uint64(n).tobytes())
        for a in (b["center"], b["quat"], b["radius"], b["length"], mt, mr):
            f.write(np.ascontiguousarray(a, dtype=np.float64).tobytes())
    exe = _build_rod_app()
    box = [float(b["box"])] * 3 if periodic else None   # periodic search, nearest-image contacts, wrap_rigid
    p = subprocess.run([exe, str(inp), "3", "3.0"] + (["%.17g" % b["box"]] if periodic else []), capture_output=True,
                       text=True, timeout=600)
    print(p.stdout[-3000:], p.stderr[-2000:])
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    steps = [ln.split() for ln in p.stdout.splitlines() if ln.startswith("STEP")]
    assert len(steps) == 3

    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    st = pipeline.ContactStepper("spherocylinder", dev(b["center"]), dev(b["radius"]), dev(b["quat"]), dev(b["length"]),
                                 search_buffer=0.1, cfg=ops.PGDConfig(max_iters=10000, tol=1e-5), mob_trans=dev(mt),
                                 mob_rot=dev(mr), periodic_box=box)
    st.reorder_bodies(cell_size=3.0, lo=[0.0, 0.0, 0.0])
    for k in range(3):
        s = st.step()
        assert int(steps[k][3]) == s.num_contacts and int(steps[k][5]) == s.num_iters
        assert float(steps[k][7]) == s.residual and int(steps[k][9]) == int(s.converged)

    def checksum(a):
        h = 1469598103934665603
        for v in np.ascontiguousarray(a).view(np.uint64).ravel().tolist():
            h = ((h ^ v) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
        return "%016x" % h
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("CHECKSUM")][0].split()
    assert line[2] == checksum(st.center.cpu().numpy()) and line[4] == checksum(st.quat.cpu().numpy())


@pytest.mark.gpu
def test_cpp_host_step_loop_like_the_reference_app():
    # the reference's scrap/lcp_spheres/NgpLcp.cpp main(), as a C++ host loop over the C ABI: 3 steps of 20k spheres at 8 % volume fraction
    exe = _build_app()
    p = subprocess.run([exe, "100", "20000", "3"], capture_output=True, text=True, timeout=300)
    print(p.stdout[-3000:], p.stderr[-2000:])
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]
    assert "Number of neighbor pairs" in p.stdout and "No overlap detected!" in p.stdout
    assert p.stdout.count("Max abs projected sep") == 3


@pytest.mark.gpu
def test_adapter_reference_tests_on_gpu():
    exe = _build()
    p = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(p.stdout[-4000:], p.stderr[-2000:])
    assert p.returncode == 0, p.stdout[-4000:]
    assert "ALL PASSED" in p.stdout
