"""The C-ABI boundary from a compiled caller's point of view (CPU only: host-side entry points)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib_dir():
    from librectify_amd import build

    build.build(verbose=False)
    return os.path.join(ROOT, "librectify_amd")


def test_reference_style_caller_compiles_links_and_runs(tmp_path, lib_dir):
    exe = str(tmp_path / "dropin_smoke")
    subprocess.check_call(["g++", "-std=c++14", os.path.join(ROOT, "tests", "cxx", "dropin_smoke.cpp"), "-I", os.path.join(ROOT, "include"),
                           "-L", lib_dir, "-l:librectify_amd.so", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    out = subprocess.check_output([exe], text=True).splitlines()
    assert out[0] == "vp1 5.0000 12.5000 1.0"
    assert out[1] == "vp2z 0.0"
    assert out[2] == "group 1"
    assert out[3] == "identity 0.000 0.000 640.000 480.000"
    assert out[4] == "sizes 28 12 80 20"


def test_headers_are_valid_c(tmp_path):
    subprocess.check_call(["gcc", "-std=c99", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cxx", "header_c.c")])
