"""Named frames for the tools in this directory: `make(name)` returns (image, min_length)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from librectify_amd import synth


def _radial(W, H, noise=0.0, seed=1):
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    img = 1.0 - np.hypot(xx - W / 2, yy - H / 2) / np.hypot(W / 2, H / 2)
    if noise > 0:
        img = img + np.random.RandomState(seed).normal(0, noise, (H, W))
    return img.astype(np.float32)


def _stripes(W, H, period, slope):
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    return (0.5 + 0.4 * np.sin((xx + slope * yy) * 2 * np.pi / period)).astype(np.float32)


def _natural(W, H):
    """the reference's doc image (luma) upsampled with a cubic spline: a natural image"""
    import scipy.ndimage as ndi
    g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "doc_image_gray.npy")).astype(np.float32) / np.float32(256.0)
    return np.ascontiguousarray(ndi.zoom(g, (H / g.shape[0], W / g.shape[1]), order=3).astype(np.float32)[:H, :W])


KINDS = {
    "bars1080": lambda: synth.frame(1920, 1080, 5),
    "bench4k": lambda: synth.frame(3840, 2160, 1),
    "bench4k_2": lambda: synth.frame(3840, 2160, 2),
    "bench4k_3": lambda: synth.frame(3840, 2160, 3),
    "bench4k_4": lambda: synth.frame(3840, 2160, 4),
    "natural4k": lambda: _natural(3840, 2160),
    "natural1080": lambda: _natural(1920, 1080),
    "doc": lambda: _natural(1000, 563),
    "regions1080": lambda: synth.region_frame(1920, 1080, 500),
    "regions4k": lambda: synth.region_frame(3840, 2160, 500),
    "edgeless4k": lambda: synth.region_frame(3840, 2160, 4),
    "radial1080": lambda: _radial(1920, 1080),
    "radial720": lambda: _radial(1280, 720),
    "radial4k": lambda: _radial(3840, 2160),
    "ramp1080": lambda: synth.ramp_frame(1920, 1080, 3),
    "ramp4k": lambda: synth.ramp_frame(3840, 2160),
    "stripes6_1080": lambda: _stripes(1920, 1080, 6, 0.0),
    "diag40_1080": lambda: _stripes(1920, 1080, 40, 0.5),
    "longbars4k": lambda: synth.long_bar_frame(3840, 2160, 3),
}


def make(name):
    img = np.ascontiguousarray(KINDS[name]())
    return img, max(img.shape) / 100.0
