"""librectify_amd — MI355X-native librectify hot path.

Python host-side mirror of the reference's C API (reference src/librectify.h) over the C-ABI
shared library librectify_amd.so (HIP kernels for gfx950).  This module is plumbing: ctypes
declarations, numpy views of the POD structs, and a thin Context class.  There is NO CPU
fallback: if the library or a GPU is missing, calls raise.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librectify_amd.so")

# reference src/librectify.h:44-54
LINE_DTYPE = np.dtype(
    [("x1", "<f4"), ("y1", "<f4"), ("x2", "<f4"), ("y2", "<f4"), ("weight", "<f4"), ("err", "<f4"), ("group_id", "<i4")]
)


class Point(C.Structure):  # reference src/librectify.h:60-63
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]


class ImageTransform(C.Structure):  # reference src/librectify.h:79-86
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

    def as_array(self):
        """rows TL, TR, BL, BR, hvp, vvp"""
        pts = [self.top_left, self.top_right, self.bottom_left, self.bottom_right, self.horizontal_vp, self.vertical_vp]
        return np.array([[p.x, p.y, p.z] for p in pts], np.float32)


ROTATE_H, ROTATE_V, RECTIFY, KEEP = 0, 1, 2, 3  # reference src/librectify.h:126-132


class RectificationConfig(C.Structure):  # reference src/librectify.h:137-150
    _fields_ = [
        ("vertical_vp_angular_tolerance", C.c_float),
        ("vertical_vp_min_distance", C.c_float),
        ("v_strategy", C.c_int),
        ("horizontal_vp_min_distance", C.c_float),
        ("h_strategy", C.c_int),
    ]

    def __init__(self, tol=40.0, vmin=1.5, v_strategy=RECTIFY, hmin=1.5, h_strategy=RECTIFY):
        super().__init__(tol, vmin, v_strategy, hmin, h_strategy)


BUF_DX, BUF_DY, BUF_DMASK, BUF_LABEL, BUF_SEED_IDX, BUF_SEED_BIN, BUF_SEED_THR, BUF_MAXMAG, BUF_SEED_SIZE = range(9)
T_UPLOAD, T_FILTER, T_SEEDS, T_FLOOD, T_FIT, T_RANSAC, T_TOTAL, T_FILTER_KERNEL, T_COUNT = range(9)

EXPORTS = [
    "find_line_segment_groups", "release_line_segments", "compute_rectification_transform",
    "compute_rectification_transform_from_vp", "fit_vanishing_point", "assign_to_group",
    "lr_context_create", "lr_context_destroy", "lr_last_error", "lr_synchronize", "lr_set_ransac_seed",
    "lr_set_ransac_iterations", "lr_set_flood_mode", "lr_device_count", "lr_find_line_segment_groups_device",
    "lr_find_line_segment_groups_host", "lr_find_line_segment_groups_batch_device", "lr_stage_filter",
    "lr_stage_filter_host", "lr_stage_seeds", "lr_stage_flood", "lr_stage_fit", "lr_download", "lr_stage_times",
    "lr_stage_counters", "lr_filter_kernel_ms", "lr_ransac_best", "lr_estimate_line_pencils",
    "lr_find_line_segment_groups_batch_host", "lr_find_line_segment_groups_batch_host_ptrs", "lr_host_alloc", "lr_host_free",
    "lr_set_seed_capacity", "lr_set_flood_blind_rounds", "lr_set_flood_staged", "lr_set_batch_streams", "lr_device_malloc", "lr_device_free", "lr_memcpy_h2d", "lr_cht_vanishing_point", "lr_refine_lines", "lr_set_estimator", "lr_ht_weights", "lr_prosac_solve", "lr_estimate_line_pencils_prosac", "lr_direct_solve", "lr_estimate_line_pencils_direct",
    "lr_estimate_line_pencils_cht", "lr_set_stage_timing", "lr_release_thread_context", "lr_set_flood_partial_commits", "lr_set_flood_multi_source", "lr_set_flood_logs", "lr_set_flood_just_in_time", "lr_set_flood_giant_step", "lr_context_trim", "lr_trim_thread_context",
    "lr_find_line_segment_groups_batch_host_multi",
]

_lib = None


class LibrectifyError(RuntimeError):
    pass


def lib():
    """Loads librectify_amd.so; raises if it has not been built (python -m librectify_amd.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise LibrectifyError(
                "librectify_amd.so is missing: build it with `python -m librectify_amd.build` (hipcc, gfx950); "
                "there is no CPU fallback"
            )
        L = C.CDLL(LIB_PATH)
        L.lr_last_error.restype = C.c_char_p
        L.find_line_segment_groups.restype = C.c_void_p
        L.find_line_segment_groups.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_bool, C.c_int, C.POINTER(C.c_int)]
        L.release_line_segments.argtypes = [C.POINTER(C.c_void_p)]
        L.release_line_segments.restype = None
        L.compute_rectification_transform.restype = ImageTransform
        L.compute_rectification_transform.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(RectificationConfig)]
        L.compute_rectification_transform_from_vp.restype = ImageTransform
        L.compute_rectification_transform_from_vp.argtypes = [C.c_int, C.c_int, C.POINTER(Point), C.POINTER(Point)]
        L.fit_vanishing_point.restype = Point
        L.fit_vanishing_point.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.assign_to_group.restype = None
        L.assign_to_group.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_float]
        L.lr_context_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        L.lr_context_destroy.argtypes = [C.c_void_p]
        L.lr_context_destroy.restype = None
        L.lr_set_ransac_seed.argtypes = [C.c_void_p, C.c_uint64]
        L.lr_set_ransac_seed.restype = None
        L.lr_set_ransac_iterations.argtypes = [C.c_void_p, C.c_int]
        L.lr_set_ransac_iterations.restype = None
        L.lr_set_flood_mode.argtypes = [C.c_void_p, C.c_int]
        L.lr_set_flood_mode.restype = None
        L.lr_synchronize.argtypes = [C.c_void_p]
        L.lr_find_line_segment_groups_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        L.lr_find_line_segment_groups_host.argtypes = L.lr_find_line_segment_groups_device.argtypes
        L.lr_find_line_segment_groups_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.lr_find_line_segment_groups_batch_host.argtypes = L.lr_find_line_segment_groups_batch_device.argtypes
        L.lr_find_line_segment_groups_batch_host_ptrs.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.lr_find_line_segment_groups_batch_host_multi.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.lr_host_alloc.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
        L.lr_host_free.argtypes = [C.c_void_p, C.c_void_p]
        L.lr_stage_filter.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.lr_stage_filter_host.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.lr_stage_seeds.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        L.lr_stage_flood.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        L.lr_stage_fit.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        L.lr_download.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
        L.lr_stage_times.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.lr_stage_counters.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.lr_filter_kernel_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        L.lr_ransac_best.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_uint64, C.c_uint32, C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int)]
        L.lr_estimate_line_pencils.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_uint64]
        L.lr_device_malloc.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
        L.lr_device_free.argtypes = [C.c_void_p, C.c_void_p]
        L.lr_memcpy_h2d.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        L.lr_set_batch_streams.argtypes = [C.c_void_p, C.c_int]
        L.lr_set_batch_streams.restype = None
        L.lr_set_seed_capacity.argtypes = [C.c_void_p, C.c_uint32]
        L.lr_set_seed_capacity.restype = None
        L.lr_set_flood_staged.argtypes = [C.c_void_p, C.c_int]
        L.lr_set_flood_staged.restype = None
        L.lr_set_flood_blind_rounds.argtypes = [C.c_void_p, C.c_int]
        L.lr_set_flood_blind_rounds.restype = None
        L.lr_cht_vanishing_point.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(Point), C.c_void_p]
        L.lr_refine_lines.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_int)]
        L.lr_set_flood_partial_commits.argtypes = [C.c_void_p, C.c_int]
        L.lr_set_flood_partial_commits.restype = None
        L.lr_set_flood_multi_source.argtypes = [C.c_void_p, C.c_int]
        L.lr_set_flood_multi_source.restype = None
        L.lr_set_flood_logs.argtypes = [C.c_void_p, C.c_int]
        L.lr_set_flood_logs.restype = None
        L.lr_set_flood_giant_step.argtypes = [C.c_void_p, C.c_int]
        L.lr_set_flood_giant_step.restype = None
        L.lr_set_flood_just_in_time.argtypes = [C.c_void_p, C.c_int]
        L.lr_set_flood_just_in_time.restype = None
        L.lr_release_thread_context.argtypes = []
        L.lr_release_thread_context.restype = None
        L.lr_set_stage_timing.argtypes = [C.c_void_p, C.c_int]
        L.lr_set_stage_timing.restype = None
        L.lr_set_estimator.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.lr_set_estimator.restype = None
        L.lr_ht_weights.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.lr_prosac_solve.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p]
        L.lr_estimate_line_pencils_prosac.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_uint64]
        L.lr_direct_solve.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.lr_estimate_line_pencils_direct.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float]
        L.lr_estimate_line_pencils_cht.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_void_p, C.POINTER(C.c_int), C.c_void_p, C.POINTER(C.c_uint64)]
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise LibrectifyError(lib().lr_last_error().decode() or "librectify_amd call failed")


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def device_count():
    return lib().lr_device_count()


class Context:
    """One device, one stream, one workspace (include/librectify_amd.h)."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        _check(lib().lr_context_create(device, C.byref(self._h)))
        self.device = device
        self.shape = None

    def close(self):
        if self._h:
            lib().lr_context_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_seed(self, seed):
        lib().lr_set_ransac_seed(self._h, C.c_uint64(seed))

    def set_iterations(self, n):
        lib().lr_set_ransac_iterations(self._h, int(n))

    def set_flood_mode(self, mode):
        lib().lr_set_flood_mode(self._h, int(mode))

    def synchronize(self):
        _check(lib().lr_synchronize(self._h))

    # ---- stage API ----
    def stage_filter_host(self, img):
        img = np.ascontiguousarray(img, np.float32)
        h, w = img.shape
        self.shape = (h, w)
        _check(lib().lr_stage_filter_host(self._h, _ptr(img), w, h, w))

    def stage_filter_device(self, dptr, w, h, stride=None):
        self.shape = (h, w)
        _check(lib().lr_stage_filter(self._h, C.c_void_p(dptr), w, h, stride or w))

    def stage_seeds(self):
        n = C.c_int(0)
        _check(lib().lr_stage_seeds(self._h, C.byref(n)))
        self.n_seeds = n.value
        return n.value

    def stage_flood(self):
        n = C.c_int(0)
        _check(lib().lr_stage_flood(self._h, C.byref(n)))

    def stage_fit(self):
        h, w = self.shape
        cap = h * w // 6 + 16
        out = np.zeros(cap, LINE_DTYPE)
        n = C.c_int(0)
        _check(lib().lr_stage_fit(self._h, _ptr(out), cap, C.byref(n)))
        return out[: n.value].copy()

    def download(self, buf):
        h, w = self.shape
        if buf in (BUF_DX, BUF_DY):
            a = np.zeros((h, w), np.float32)
        elif buf == BUF_DMASK:
            a = np.zeros((h, w), np.uint8)
        elif buf == BUF_LABEL:
            a = np.zeros((h, w), np.int32)
        elif buf in (BUF_SEED_IDX, BUF_SEED_BIN, BUF_SEED_SIZE):
            a = np.zeros(self.n_seeds, np.int32)
        elif buf == BUF_SEED_THR:
            a = np.zeros(self.n_seeds, np.float32)
        elif buf == BUF_MAXMAG:
            a = np.zeros(1, np.float32)
        else:
            raise ValueError(buf)
        if a.nbytes:
            _check(lib().lr_download(self._h, buf, _ptr(a), a.nbytes))
        return a

    def stage_times(self):
        t = np.zeros(T_COUNT, np.float32)
        _check(lib().lr_stage_times(self._h, _ptr(t), T_COUNT))
        return t

    def stage_times_partial(self):
        """ms of the last filter kernel alone (valid right after stage_filter_*)."""
        ms = C.c_float(0)
        _check(lib().lr_filter_kernel_ms(self._h, C.byref(ms)))
        return ms.value

    def stage_counters(self):
        c = np.zeros(16, np.int64)
        _check(lib().lr_stage_counters(self._h, _ptr(c), 16))
        return dict(seeds=int(c[0]), components=int(c[1]), flood_rounds=int(c[2]), labelled_px=int(c[3]),
                    second_tier_seeds=int(c[4]), slabs=int(c[5]), ordered_tail_seeds=int(c[6]), frame_laps=int(c[7]),
                    walked_px=int(c[8]), walk_steps=int(c[9]), multi_source_walks=int(c[10]), log_rewalks=int(c[11]), log_give_ups=int(c[12]), giants_held=int(c[13]), giant_steps=int(c[14]),
                    quiet_round_misses=int(c[15]))

    # ---- full path ----
    def find_line_segment_groups(self, img, min_length, refine=False, num_threads=-1, capacity=None):
        """img: 2-D float32 numpy array (host)."""
        img = np.asarray(img, np.float32)
        h, w = img.shape
        stride = img.strides[0] // 4
        assert img.strides[1] == 4
        cap = capacity or (h * w // 6 + 16)
        out = np.zeros(cap, LINE_DTYPE)
        n = C.c_int(0)
        self.shape = (h, w)
        _check(lib().lr_find_line_segment_groups_host(self._h, _ptr(img), w, h, stride, min_length, int(refine), num_threads, _ptr(out), cap, C.byref(n)))
        return out[: min(n.value, cap)].copy()

    def find_line_segment_groups_device(self, dptr, w, h, min_length, refine=False, stride=None, capacity=None, out=None):
        cap = capacity or (h * w // 6 + 16)
        if out is None:
            out = np.zeros(cap, LINE_DTYPE)
        n = C.c_int(0)
        self.shape = (h, w)
        _check(lib().lr_find_line_segment_groups_device(self._h, C.c_void_p(dptr), w, h, stride or w, min_length, int(refine), -1, _ptr(out), cap, C.byref(n)))
        return out[: min(n.value, cap)]

    def device_upload(self, array):
        """Copies a contiguous numpy array to a fresh device buffer; returns its address (free with device_free)."""
        a = np.ascontiguousarray(array)
        p = C.c_void_p()
        _check(lib().lr_device_malloc(self._h, a.nbytes, C.byref(p)))
        _check(lib().lr_memcpy_h2d(self._h, p, _ptr(a), a.nbytes))
        return p.value

    def device_free(self, ptr):
        _check(lib().lr_device_free(self._h, C.c_void_p(ptr)))

    def set_seed_capacity(self, cap):
        lib().lr_set_seed_capacity(self._h, int(cap))

    def set_flood_staged(self, on):
        lib().lr_set_flood_staged(self._h, int(bool(on)))

    def set_flood_blind_rounds(self, rounds):
        lib().lr_set_flood_blind_rounds(self._h, int(rounds))

    def set_batch_streams(self, n):
        lib().lr_set_batch_streams(self._h, int(n))

    def find_line_segment_groups_batch_device(self, dptr, image_stride, batch, w, h, min_length, refine=False, capacity=4096, cfg=None, out=None):
        if out is None:
            out = np.zeros((batch, capacity), LINE_DTYPE)
        n = np.zeros(batch, np.int32)
        tf = (ImageTransform * batch)()
        cfg = cfg or RectificationConfig()
        self.shape = (h, w)
        _check(lib().lr_find_line_segment_groups_batch_device(self._h, C.c_void_p(dptr), image_stride, batch, w, h, w, min_length, int(refine), -1, _ptr(out), capacity, _ptr(n), C.byref(cfg), C.byref(tf)))
        return out, n, tf

    def find_line_segment_groups_batch_host(self, frames, min_length, refine=False, num_threads=-1, capacity=4096, cfg=None, out=None, devices=None):
        """frames: float32 array [B, H, W] (rows contiguous; any row/frame strides) or a list of 2-D float32 arrays
        of one shape, in HOST memory (pageable, or page-locked as host_alloc returns it).  devices: a list of device
        indices to deal the frames over in contiguous blocks (lr_find_line_segment_groups_batch_host_multi)."""
        if isinstance(frames, np.ndarray) and frames.ndim == 3:
            assert frames.dtype == np.float32 and frames.strides[2] == 4
            batch, h, w = frames.shape
            stride = frames.strides[1] // 4
            ptrs = (C.c_void_p * batch)(*[frames.ctypes.data + b * frames.strides[0] for b in range(batch)])
        else:
            frames = [np.asarray(f, np.float32) for f in frames]
            batch = len(frames)
            h, w = frames[0].shape
            stride = frames[0].strides[0] // 4
            assert all(f.shape == (h, w) and f.strides == frames[0].strides and f.strides[1] == 4 for f in frames)
            ptrs = (C.c_void_p * batch)(*[f.ctypes.data for f in frames])
        if out is None:
            out = np.zeros((batch, capacity), LINE_DTYPE)
        n = np.zeros(batch, np.int32)
        tf = (ImageTransform * batch)()
        cfg = cfg or RectificationConfig()
        self.shape = (h, w)
        if devices is not None:
            devs = (C.c_int * len(devices))(*[int(d) for d in devices])
            _check(lib().lr_find_line_segment_groups_batch_host_multi(self._h, devs, len(devices), ptrs, batch, w, h, stride, min_length, int(refine), num_threads, _ptr(out), capacity, _ptr(n), C.byref(cfg), C.byref(tf)))
            return out, n, tf
        _check(lib().lr_find_line_segment_groups_batch_host_ptrs(self._h, ptrs, batch, w, h, stride, min_length, int(refine), num_threads, _ptr(out), capacity, _ptr(n), C.byref(cfg), C.byref(tf)))
        return out, n, tf

    def host_alloc(self, shape, dtype=np.float32):
        """Page-locked host array (free with host_free(array)): frames in it are uploaded without a staging copy."""
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = C.c_void_p()
        _check(lib().lr_host_alloc(self._h, nbytes, C.byref(p)))
        buf = (C.c_ubyte * nbytes).from_address(p.value)
        a = np.frombuffer(buf, dtype=dtype).reshape(shape)
        self._pinned = getattr(self, "_pinned", {})
        self._pinned[a.ctypes.data] = p.value
        return a

    def host_free(self, array):
        p = self._pinned.pop(array.ctypes.data)
        _check(lib().lr_host_free(self._h, C.c_void_p(p)))

    # ---- RANSAC ----
    def ransac_best(self, lines_norm, indices, tol, n_iter, seed, rnd=0):
        lines_norm = np.ascontiguousarray(lines_norm, LINE_DTYPE)
        indices = np.ascontiguousarray(indices, np.int32)
        bh = np.zeros(3, np.float32)
        bs = C.c_float(0)
        bi = C.c_int(0)
        _check(lib().lr_ransac_best(self._h, _ptr(lines_norm), len(lines_norm), _ptr(indices), len(indices), tol, n_iter, C.c_uint64(seed), C.c_uint32(rnd), _ptr(bh), C.byref(bs), C.byref(bi)))
        return dict(best_h=bh, score=bs.value, iter=bi.value)

    def cht_vanishing_point(self, lines, d=128):
        lines = np.ascontiguousarray(lines, LINE_DTYPE)
        vp = Point()
        acc = np.zeros((d, d), np.uint64)
        _check(lib().lr_cht_vanishing_point(self._h, _ptr(lines), len(lines), d, C.byref(vp), _ptr(acc)))
        return np.array([vp.x, vp.y, vp.z], np.float32), acc

    def refine_lines(self, lines):
        lines = np.ascontiguousarray(lines, LINE_DTYPE)
        out = np.zeros(len(lines), LINE_DTYPE)
        n = C.c_int(0)
        _check(lib().lr_refine_lines(self._h, _ptr(lines), len(lines), _ptr(out), C.byref(n)))
        return out[: n.value].copy()

    def set_flood_partial_commits(self, on=True):
        lib().lr_set_flood_partial_commits(self._h, int(bool(on)))

    def set_flood_multi_source(self, on=True):
        lib().lr_set_flood_multi_source(self._h, int(bool(on)))

    def set_flood_just_in_time(self, on=True):
        lib().lr_set_flood_just_in_time(self._h, int(bool(on)))

    def trim(self):
        """gives the memory that is sized by the largest frame seen back to the system (the next call allocates what it needs)"""
        _check(lib().lr_context_trim(self._h))

    def set_flood_giant_step(self, on=True):
        """the lowest active seed's flood by the whole device when it outgrows the LDS tiers (default); False = the slab walk"""
        lib().lr_set_flood_giant_step(self._h, int(bool(on)))

    def set_flood_logs(self, on=1):
        """0 = off, 1 = on (default for single calls), 2 = on with every log through the fall-back path (test hook)."""
        lib().lr_set_flood_logs(self._h, int(on))

    def set_stage_timing(self, on=True):
        """stage timers of the frame calls (off by default: each event record idles the GPU for a few microseconds)"""
        lib().lr_set_stage_timing(self._h, int(bool(on)))

    def set_estimator(self, kind, param=-1):
        """0 RANSAC (default), 1 PROSAC (param = T_N), 2 DirectEstimator, 3 diamond-space accumulator (param = its size d)."""
        lib().lr_set_estimator(self._h, int(kind), int(param))

    def estimate_line_pencils_cht(self, lines, max_models=4, inlier_deg=2.0, garbage_deg=4.0, d=128):
        """-> (lines with group_id, refit models (k, 3), winning cells (k,), cells voted for)"""
        lines = np.ascontiguousarray(lines, LINE_DTYPE).copy()
        models = np.zeros((max(max_models, 1), 3), np.float32)
        cells = np.zeros(max(max_models, 1), np.uint32)
        k = C.c_int(0)
        votes = C.c_uint64(0)
        _check(lib().lr_estimate_line_pencils_cht(self._h, _ptr(lines), len(lines), max_models, inlier_deg, garbage_deg, d, _ptr(models), C.byref(k), _ptr(cells), C.byref(votes)))
        return lines, models[: k.value].copy(), cells[: k.value].copy(), int(votes.value)

    def ht_weights(self, lines_norm, indices):
        lines_norm = np.ascontiguousarray(lines_norm, LINE_DTYPE)
        indices = np.ascontiguousarray(indices, np.int32)
        out = np.zeros(len(indices), np.float32)
        _check(lib().lr_ht_weights(self._h, _ptr(lines_norm), len(lines_norm), _ptr(indices), len(indices), _ptr(out)))
        return out

    def prosac_solve(self, lines_norm, indices, tol, T_N=-1, seed=0, rnd=0):
        lines_norm = np.ascontiguousarray(lines_norm, LINE_DTYPE)
        indices = np.ascontiguousarray(indices, np.int32)
        h = np.zeros(3, np.float32)
        tr = np.zeros(4, np.int32)
        _check(lib().lr_prosac_solve(self._h, _ptr(lines_norm), len(lines_norm), _ptr(indices), len(indices), tol, T_N, C.c_uint64(seed), C.c_uint32(rnd), _ptr(h), _ptr(tr)))
        return dict(h=h, iterations=int(tr[0]), n_star=int(tr[1]), best_iter=int(tr[2]), I_N_best=int(tr[3]))

    def estimate_line_pencils_prosac(self, lines, max_models=4, inlier_deg=2.0, garbage_deg=4.0, T_N=-1, seed=0):
        lines = np.ascontiguousarray(lines, LINE_DTYPE).copy()
        _check(lib().lr_estimate_line_pencils_prosac(self._h, _ptr(lines), len(lines), max_models, inlier_deg, garbage_deg, T_N, C.c_uint64(seed)))
        return lines

    def direct_solve(self, lines_norm, indices):
        lines_norm = np.ascontiguousarray(lines_norm, LINE_DTYPE)
        indices = np.ascontiguousarray(indices, np.int32)
        h = np.zeros(3, np.float32)
        _check(lib().lr_direct_solve(self._h, _ptr(lines_norm), len(lines_norm), _ptr(indices), len(indices), _ptr(h)))
        return h

    def estimate_line_pencils_direct(self, lines, max_models=4, inlier_deg=2.0, garbage_deg=4.0):
        lines = np.ascontiguousarray(lines, LINE_DTYPE).copy()
        _check(lib().lr_estimate_line_pencils_direct(self._h, _ptr(lines), len(lines), max_models, inlier_deg, garbage_deg))
        return lines

    def estimate_line_pencils(self, lines, max_models=4, inlier_deg=2.0, garbage_deg=4.0, n_iter=10000, seed=0):
        lines = np.ascontiguousarray(lines, LINE_DTYPE).copy()
        _check(lib().lr_estimate_line_pencils(self._h, _ptr(lines), len(lines), max_models, inlier_deg, garbage_deg, n_iter, C.c_uint64(seed)))
        return lines


# ---- the reference's six functions, by name --------------------------------------------------

def release_thread_context():
    """frees the calling thread's drop-in context (the next drop-in call makes a new one)"""
    lib().lr_release_thread_context()


def find_line_segment_groups(buffer, min_length, refine=False, num_threads=-1):
    """reference find_line_segment_groups (src/librectify.h:111-116) on a 2-D float32 array.
    Returns a LINE_DTYPE array (empty where the reference returns NULL)."""
    buffer = np.asarray(buffer, np.float32)
    h, w = buffer.shape
    assert buffer.strides[1] == 4
    n = C.c_int(0)
    p = lib().find_line_segment_groups(_ptr(buffer), w, h, buffer.strides[0] // 4, min_length, bool(refine), num_threads, C.byref(n))
    if not p:
        err = lib().lr_last_error().decode()
        if err:  # NULL because of a missing GPU / HIP failure, not because nothing was found
            raise LibrectifyError(err)
        return np.zeros(0, LINE_DTYPE)
    out = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_ubyte)), shape=(n.value * LINE_DTYPE.itemsize,)).view(LINE_DTYPE).copy()
    pp = C.c_void_p(p)
    lib().release_line_segments(C.byref(pp))
    assert pp.value is None
    return out


def compute_rectification_transform(lines, width, height, cfg=None):
    lines = np.ascontiguousarray(lines, LINE_DTYPE)
    cfg = cfg or RectificationConfig()
    return lib().compute_rectification_transform(_ptr(lines), len(lines), width, height, C.byref(cfg))


def compute_rectification_transform_from_vp(width, height, vp_h, vp_v):
    a = Point(*[float(v) for v in vp_h])
    b = Point(*[float(v) for v in vp_v])
    return lib().compute_rectification_transform_from_vp(width, height, C.byref(a), C.byref(b))


def fit_vanishing_point(lines, group):
    lines = np.ascontiguousarray(lines, LINE_DTYPE)
    p = lib().fit_vanishing_point(_ptr(lines), len(lines), group)
    return np.array([p.x, p.y, p.z], np.float32)


def assign_to_group(lines, new_lines, angular_tolerance):
    lines = np.ascontiguousarray(lines, LINE_DTYPE)
    new_lines = np.ascontiguousarray(new_lines, LINE_DTYPE).copy()
    lib().assign_to_group(_ptr(lines), len(lines), _ptr(new_lines), len(new_lines), angular_tolerance)
    return new_lines
