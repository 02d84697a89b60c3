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


def test_adapter_header_compiles_and_links():
    assert os.path.exists(_build())
    assert os.path.exists(_build_app())


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
