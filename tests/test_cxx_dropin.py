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


def _build_recipe(tmp_path, lib_dir):
    exe = str(tmp_path / "rectify_recipe")
    subprocess.check_call(["g++", "-std=c++14", "-O2", "-Wall", "-Wextra", "-Werror", os.path.join(ROOT, "examples", "rectify_recipe.cpp"),
                           "-I", os.path.join(ROOT, "include"), "-L", lib_dir, "-l:librectify_amd.so",
                           "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    return exe


def _write_doc_pgm(path):
    import numpy as np

    a = np.load(os.path.join(ROOT, "tests", "golden", "doc_image_gray.npy"))
    with open(path, "wb") as f:
        f.write(b"P5\n# doc/image.jpg luma\n%d %d\n255\n" % (a.shape[1], a.shape[0]) + a.tobytes())
    return a


def test_caller_recipe_example_builds_and_writes_the_demo_csv_files(tmp_path, lib_dir):
    """examples/rectify_recipe.cpp without a GPU: the detector fails loudly, the program still writes
    the two CSV files in the reference demo's layout (no segments; identity corners and ideal points)."""
    exe = _build_recipe(tmp_path, lib_dir)
    pgm = str(tmp_path / "doc.pgm")
    _write_doc_pgm(pgm)
    env = dict(os.environ, HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="-1")
    r = subprocess.run([exe, pgm, str(tmp_path / "out"), "--max-size", "500"], text=True, capture_output=True, env=env)
    assert r.returncode == 0, r.stderr
    assert "no CPU fallback" in r.stderr
    assert "1000x563 -> 500x282 (scale 0.5), 0 segments" in r.stdout
    assert open(str(tmp_path / "out_lines.csv")).read() == ""
    assert open(str(tmp_path / "out_tform.csv")).read().split() == ["0,0", "1000,0", "0,563", "1000,563", "1,0,0", "0,1,0"]


@pytest.mark.gpu
def test_caller_recipe_example_reproduces_the_library_result_on_the_doc_image(tmp_path, lib_dir):
    """The recipe end to end on the doc fixture (1000x563 < 1200: no prescale): the CSV rows are the segments the
    Python binding returns for gray/256 with min_length 10, the transform is compute_rectification_transform with
    horizontal_vp_min_distance = 2, both at the 6 significant digits the demo prints."""
    import numpy as np

    import librectify_amd as L

    exe = _build_recipe(tmp_path, lib_dir)
    pgm = str(tmp_path / "doc.pgm")
    a = _write_doc_pgm(pgm)
    subprocess.check_call([exe, pgm, str(tmp_path / "out")])
    rows = np.loadtxt(str(tmp_path / "out_lines.csv"), delimiter=",", ndmin=2)
    ctx = L.Context(0)
    ctx.set_seed(0)
    ref = ctx.find_line_segment_groups((a.astype(np.float32) / np.float32(256.0)), 10.0)
    assert len(rows) == len(ref) > 500
    for j, name in enumerate(["x1", "y1", "x2", "y2", "weight", "err"]):
        np.testing.assert_allclose(rows[:, j], ref[name], rtol=6e-6, atol=0)
    np.testing.assert_array_equal(rows[:, 6].astype(np.int32), ref["group_id"])
    cfg = L.RectificationConfig()
    cfg.horizontal_vp_min_distance = 2
    T = L.compute_rectification_transform(ref, 1000, 563, cfg).as_array()
    got = [list(map(float, l.split(","))) for l in open(str(tmp_path / "out_tform.csv")).read().split()]
    for i in range(4):
        np.testing.assert_allclose(got[i], T[i][:2], rtol=6e-6)
    np.testing.assert_allclose(got[4], T[4], rtol=6e-6)
    np.testing.assert_allclose(got[5], T[5], rtol=6e-6)
    # prescaled run: half size, endpoints come back in full-image coordinates
    subprocess.check_call([exe, pgm, str(tmp_path / "half"), "--max-size", "500"])
    half = np.loadtxt(str(tmp_path / "half_lines.csv"), delimiter=",", ndmin=2)
    assert len(half) > 100
    assert half[:, [0, 2]].max() <= 1000.5 and half[:, [1, 3]].max() <= 563.5 and half[:, :4].max() > 600
