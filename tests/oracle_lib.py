"""ctypes loader for oracle/liboracle.so — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
The product package (librectify_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "liboracle.so")

# librectify.h:44-54 — 28-byte POD
LINE_DTYPE = np.dtype(
    [("x1", "<f4"), ("y1", "<f4"), ("x2", "<f4"), ("y2", "<f4"), ("weight", "<f4"), ("err", "<f4"), ("group_id", "<i4")]
)
assert LINE_DTYPE.itemsize == 28


class Point(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]


class ImageTransform(C.Structure):
    _fields_ = [
        ("width", C.c_int),
        ("height", C.c_int),
        ("top_left", Point),
        ("top_right", Point),
        ("bottom_left", Point),
        ("bottom_right", Point),
        ("horizontal_vp", Point),
        ("vertical_vp", Point),
    ]


class RectificationConfig(C.Structure):
    _fields_ = [
        ("vertical_vp_angular_tolerance", C.c_float),
        ("vertical_vp_min_distance", C.c_float),
        ("v_strategy", C.c_int),
        ("horizontal_vp_min_distance", C.c_float),
        ("h_strategy", C.c_int),
    ]


ROTATE_H, ROTATE_V, RECTIFY, KEEP = 0, 1, 2, 3


def default_config():
    return RectificationConfig(40.0, 1.5, RECTIFY, 1.5, RECTIFY)


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(
            os.path.join(ORACLE_DIR, "rectify_oracle.cpp")
        ):
            build()
        _lib = C.CDLL(LIB_PATH)
        _lib.orc_cos_threshold.restype = C.c_float
        _lib.orc_cos_threshold.argtypes = [C.c_float]
    return _lib


def _p(a, t=C.c_void_p):
    return a.ctypes.data_as(t) if a is not None else None


def as_lines(a):
    a = np.ascontiguousarray(a, dtype=LINE_DTYPE)
    return a


def lines_from_rows(rows):
    """rows: (n,7) float array x1,y1,x2,y2,weight,err,group_id"""
    rows = np.asarray(rows, dtype=np.float64)
    out = np.zeros(len(rows), dtype=LINE_DTYPE)
    for i, k in enumerate(["x1", "y1", "x2", "y2", "weight", "err"]):
        out[k] = rows[:, i].astype(np.float32)
    out["group_id"] = rows[:, 6].astype(np.int32)
    return out


def gauss_deriv_kernel(size=2, sigma=1.0, dir_x=True):
    n = 2 * size + 1
    out = np.zeros((n, n), np.float32)
    lib().orc_gauss_deriv_kernel(C.c_int(size), C.c_float(sigma), C.c_int(int(dir_x)), _p(out))
    return out


def bin_trig(n_bins=8):
    st = np.zeros(n_bins, np.float32)
    ct = np.zeros(n_bins, np.float32)
    lib().orc_bin_trig(C.c_int(n_bins), _p(st), _p(ct))
    return st, ct


def filter_stage(img, num_threads=-1, planes=False):
    img = np.ascontiguousarray(img, np.float32)
    h, w = img.shape
    dx = np.zeros((h, w), np.float32)
    dy = np.zeros((h, w), np.float32)
    mag = np.zeros((h, w), np.float32)
    bins = np.zeros((h, w), np.int32)
    dmask = np.zeros((h, w), np.uint8)
    pl = np.zeros((8, h, w), np.float32) if planes else None
    lib().orc_filter_stage(_p(img), C.c_int(w), C.c_int(h), C.c_int(num_threads), _p(dx), _p(dy), _p(mag), _p(bins), _p(dmask), _p(pl))
    return dict(dx=dx, dy=dy, mag=mag, bin=bins, dmask=dmask, planes=pl)


def conv_gradients_25tap(img):
    """the reference's 25-tap correlation (filter.cpp:65-98), second path of the oracle"""
    img = np.ascontiguousarray(img, np.float32)
    h, w = img.shape
    dx = np.zeros((h, w), np.float32)
    dy = np.zeros((h, w), np.float32)
    lib().orc_conv_gradients_25tap(_p(img), C.c_int(w), C.c_int(h), _p(dx), _p(dy))
    return dx, dy


def find_seeds(mag, bins, cap=None):
    h, w = mag.shape
    cap = cap or (h * w // 4 + 16)
    rows = np.zeros(cap, np.int32)
    cols = np.zeros(cap, np.int32)
    vals = np.zeros(cap, np.float32)
    sb = np.zeros(cap, np.int32)
    msv = C.c_float(0)
    n = lib().orc_find_seeds(_p(np.ascontiguousarray(mag)), _p(np.ascontiguousarray(bins)), C.c_int(w), C.c_int(h), _p(rows), _p(cols), _p(vals), _p(sb), C.c_int(cap), C.byref(msv))
    assert n <= cap
    return dict(rows=rows[:n], cols=cols[:n], vals=vals[:n], bins=sb[:n], min_seed_value=msv.value)


def find_line_segments(img, tolerance=0.25, num_threads=-1, want_label=True):
    img = np.ascontiguousarray(img, np.float32)
    h, w = img.shape
    cap = h * w // 6 + 16
    out = np.zeros(cap, LINE_DTYPE)
    label = np.zeros((h, w), np.int32) if want_label else None
    comp_seed = np.zeros(cap, np.int32)
    n_seeds = C.c_int(0)
    times = np.zeros(5, np.float64)
    n = lib().orc_find_line_segments(_p(img), C.c_int(w), C.c_int(h), C.c_float(tolerance), C.c_int(num_threads), _p(out), C.c_int(cap), _p(label), _p(comp_seed), C.byref(n_seeds), _p(times))
    assert n <= cap
    return dict(lines=out[:n].copy(), label=label, comp_seed=comp_seed[:n].copy(), n_seeds=n_seeds.value, times_ms=times)


def fit_line_parameters(xr, xc, w):
    xr = np.ascontiguousarray(xr, np.float32)
    xc = np.ascontiguousarray(xc, np.float32)
    w = np.ascontiguousarray(w, np.float32)
    out = np.zeros(1, LINE_DTYPE)
    lib().orc_fit_line_parameters(_p(xr), _p(xc), _p(w), C.c_int(len(w)), _p(out))
    return out[0]


def filter_lines(lines, min_length):
    lines = as_lines(lines)
    out = np.zeros(len(lines), LINE_DTYPE)
    n = lib().orc_filter_lines(_p(lines), C.c_int(len(lines)), C.c_float(min_length), _p(out))
    return out[:n].copy()


def refine_lines(lines, num_threads=-1):
    lines = as_lines(lines)
    out = np.zeros(len(lines), LINE_DTYPE)
    n = lib().orc_refine_lines(_p(lines), C.c_int(len(lines)), C.c_int(num_threads), _p(out))
    return out[:n].copy()


def refine_lines_params(lines, cos_gate=0.99, max_offset=0.02, lo=-0.5, hi=1.5, min_pair_length=0.0):
    """postprocess_lines_segments with other constants than today's (pin sweep, tests/test_oracle_pins.py)"""
    lines = as_lines(lines)
    out = np.zeros(len(lines), LINE_DTYPE)
    n = lib().orc_refine_lines_params(_p(lines), C.c_int(len(lines)), C.c_double(cos_gate), C.c_double(max_offset), C.c_double(lo), C.c_double(hi), C.c_double(min_pair_length), _p(out))
    return out[:n].copy()


def estimate_line_pencils(lines, max_models=4, inlier_deg=2.0, garbage_deg=4.0, n_iter=10000, seed=0, num_threads=-1):
    lines = as_lines(lines).copy()
    models = np.zeros((max_models, 3), np.float32)
    k = lib().orc_estimate_line_pencils(_p(lines), C.c_int(len(lines)), C.c_int(max_models), C.c_float(inlier_deg), C.c_float(garbage_deg), C.c_int(n_iter), C.c_uint64(seed), C.c_int(num_threads), _p(models))
    return lines, models[:k]


def normalize_lines(lines):
    lines = as_lines(lines)
    out = np.zeros(len(lines), LINE_DTYPE)
    c = np.zeros(2, np.float32)
    s = C.c_float(0)
    lib().orc_normalize_lines(_p(lines), C.c_int(len(lines)), _p(out), _p(c), C.byref(s))
    return out, c, s.value


def pencil_model(lines_norm):
    lines_norm = as_lines(lines_norm)
    n = len(lines_norm)
    h = np.zeros((n, 3), np.float32)
    a = np.zeros((n, 2), np.float32)
    d = np.zeros((n, 2), np.float32)
    ln = np.zeros(n, np.float32)
    lib().orc_pencil_model(_p(lines_norm), C.c_int(n), _p(h), _p(a), _p(d), _p(ln))
    return h, a, d, ln


def ransac_best(lines_norm, indices, tol, n_iter, seed, rnd=0, num_threads=-1):
    lines_norm = as_lines(lines_norm)
    indices = np.ascontiguousarray(indices, np.int32)
    bh = np.zeros(3, np.float32)
    rh = np.zeros(3, np.float32)
    bs = C.c_float(0)
    bi = C.c_int(0)
    lib().orc_ransac_best(_p(lines_norm), C.c_int(len(lines_norm)), _p(indices), C.c_int(len(indices)), C.c_float(tol), C.c_int(n_iter), C.c_uint64(seed), C.c_uint32(rnd), C.c_int(num_threads), _p(bh), C.byref(bs), C.byref(bi), _p(rh))
    return dict(best_h=bh, score=bs.value, iter=bi.value, refit_h=rh)


def sample_pair(seed, rnd, it, n):
    a = C.c_uint32(0)
    b = C.c_uint32(0)
    lib().orc_sample_pair(C.c_uint64(seed), C.c_uint32(rnd), C.c_uint32(it), C.c_uint32(n), C.byref(a), C.byref(b))
    return a.value, b.value


def choice_knuth_mt(mt_seed, N, n, n_draws):
    out = np.zeros((n_draws, n), np.int32)
    lib().orc_choice_knuth_mt(C.c_uint32(mt_seed), C.c_int(N), C.c_int(n), C.c_int(n_draws), _p(out))
    return out


def cos_threshold(deg):
    return lib().orc_cos_threshold(C.c_float(deg))


def get_weights(lines_norm, indices):
    lines_norm = as_lines(lines_norm)
    indices = np.ascontiguousarray(indices, np.int32)
    out = np.zeros(len(indices), np.float32)
    lib().orc_get_weights(_p(lines_norm), C.c_int(len(lines_norm)), _p(indices), C.c_int(len(indices)), _p(out))
    return out


def get_weights_fixed(lines_norm, indices):
    lines_norm = as_lines(lines_norm)
    indices = np.ascontiguousarray(indices, np.int32)
    out = np.zeros(len(indices), np.float32)
    lib().orc_get_weights_fixed(_p(lines_norm), C.c_int(len(lines_norm)), _p(indices), C.c_int(len(indices)), _p(out))
    return out


def prosac_solve(lines_norm, indices, tol, T_N=-1, seed=0, rnd=0):
    lines_norm = as_lines(lines_norm)
    indices = np.ascontiguousarray(indices, np.int32)
    h = np.zeros(3, np.float32)
    tr = np.zeros(4, np.int32)
    lib().orc_prosac_solve(_p(lines_norm), C.c_int(len(lines_norm)), _p(indices), C.c_int(len(indices)), C.c_float(tol), C.c_int(T_N), C.c_uint64(seed), C.c_uint32(rnd), _p(h), _p(tr))
    return dict(h=h, iterations=int(tr[0]), n_star=int(tr[1]), best_iter=int(tr[2]), I_N_best=int(tr[3]))


def estimate_line_pencils_prosac(lines, max_models=4, inlier_deg=2.0, garbage_deg=4.0, T_N=-1, seed=0):
    lines = as_lines(lines).copy()
    lib().orc_estimate_line_pencils_prosac(_p(lines), C.c_int(len(lines)), C.c_int(max_models), C.c_float(inlier_deg), C.c_float(garbage_deg), C.c_int(T_N), C.c_uint64(seed))
    return lines


def direct_solve(lines_norm, indices):
    lines_norm = as_lines(lines_norm)
    indices = np.ascontiguousarray(indices, np.int32)
    h = np.zeros(3, np.float32)
    lib().orc_direct_solve(_p(lines_norm), C.c_int(len(lines_norm)), _p(indices), C.c_int(len(indices)), _p(h))
    return h


def estimate_line_pencils_direct(lines, max_models=4, inlier_deg=2.0, garbage_deg=4.0):
    lines = as_lines(lines).copy()
    lib().orc_estimate_line_pencils_direct(_p(lines), C.c_int(len(lines)), C.c_int(max_models), C.c_float(inlier_deg), C.c_float(garbage_deg))
    return lines


def cht_vanishing_point(lines, d=128):
    lines = as_lines(lines)
    vp = np.zeros(3, np.float32)
    acc = np.zeros((d, d), np.uint64)
    lib().orc_cht_vanishing_point(_p(lines), C.c_int(len(lines)), C.c_int(d), _p(vp), _p(acc))
    return vp, acc


def estimate_line_pencils_cht(lines, max_models=4, inlier_deg=2.0, garbage_deg=4.0, d=128):
    lines = as_lines(lines).copy()
    models = np.zeros((max(max_models, 1), 3), np.float32)
    cells = np.zeros(max(max_models, 1), np.uint32)
    k = lib().orc_estimate_line_pencils_cht(_p(lines), C.c_int(len(lines)), C.c_int(max_models), C.c_float(inlier_deg), C.c_float(garbage_deg), C.c_int(d), _p(models), _p(cells))
    return lines, models[:k].copy(), cells[:k].copy()


def niter_ransac(p, eps, s, nmax=-1):
    f = lib().orc_niter_ransac
    f.restype = C.c_int
    return f(C.c_double(p), C.c_double(eps), C.c_int(s), C.c_int(nmax))


def fit_vanishing_points(lines):
    lines = as_lines(lines)
    ids = np.zeros(64, np.int32)
    vps = np.zeros((64, 3), np.float32)
    k = lib().orc_fit_vanishing_points(_p(lines), C.c_int(len(lines)), _p(ids), _p(vps), C.c_int(64))
    return ids[:k].copy(), vps[:k].copy()


def fit_vanishing_point(lines, group):
    lines = as_lines(lines)
    p = Point()
    lib().orc_fit_vanishing_point(_p(lines), C.c_int(len(lines)), C.c_int(group), C.byref(p))
    return np.array([p.x, p.y, p.z], np.float32)


def assign_to_group(lines, new_lines, tol_deg):
    lines = as_lines(lines)
    new_lines = as_lines(new_lines).copy()
    lib().orc_assign_to_group(_p(lines), C.c_int(len(lines)), _p(new_lines), C.c_int(len(new_lines)), C.c_float(tol_deg))
    return new_lines


def transform_to_array(T):
    """rows: TL, TR, BL, BR, hvp, vvp (the order autorectify.cpp:45-53 writes the tform csv in)"""
    pts = [T.top_left, T.top_right, T.bottom_left, T.bottom_right, T.horizontal_vp, T.vertical_vp]
    return np.array([[p.x, p.y, p.z] for p in pts], np.float32)


def compute_rectification_transform(lines, width, height, cfg=None):
    lines = as_lines(lines)
    cfg = cfg or default_config()
    T = ImageTransform()
    lib().orc_compute_rectification_transform(_p(lines), C.c_int(len(lines)), C.c_int(width), C.c_int(height), C.byref(cfg), C.byref(T))
    return T


def compute_rectification_transform_from_vp(width, height, vp_h, vp_v):
    T = ImageTransform()
    a = Point(*[float(v) for v in vp_h])
    b = Point(*[float(v) for v in vp_v])
    lib().orc_compute_rectification_transform_from_vp(C.c_int(width), C.c_int(height), C.byref(a), C.byref(b), C.byref(T))
    return T


def find_line_segment_groups(img, min_length, refine=False, num_threads=-1, seed=0, stride=None):
    img = np.asarray(img, np.float32)
    h, w = img.shape
    if stride is None:
        img = np.ascontiguousarray(img)
        stride = w
    cap = h * w // 6 + 16
    out = np.zeros(cap, LINE_DTYPE)
    times = np.zeros(7, np.float64)
    n = lib().orc_find_line_segment_groups(_p(img), C.c_int(w), C.c_int(h), C.c_int(stride), C.c_float(min_length), C.c_int(int(refine)), C.c_int(num_threads), C.c_uint64(seed), _p(out), C.c_int(cap), _p(times))
    return out[:n].copy(), times


def max_threads():
    return lib().orc_max_threads()
