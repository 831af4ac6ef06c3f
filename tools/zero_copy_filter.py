"""Experiment: the filter kernel reading a 4K frame straight from page-locked HOST memory (no upload at all) against
DMA + filter from HBM.  Prints the filter kernel's duration (HIP events) per variant."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import librectify_amd as L
from librectify_amd import synth

W, H = 3840, 2160
img = synth.frame(W, H, 1)
ctx = L.Context(0)
pinned = ctx.host_alloc((3, H, W))
for i in range(3):
    pinned[i] = np.roll(img, 7 * i, axis=1)
d = ctx.device_upload(pinned)
for name, base in (("HBM", d), ("page-locked host memory (zero copy)", pinned.ctypes.data)):
    ms = []
    for rep in range(4):
        for i in range(3):
            ctx.stage_filter_device(base + i * W * H * 4, W, H)
            ctx.synchronize()
            if rep:
                ms.append(ctx.stage_times_partial())
    print("filter from %s: %.1f us mean, %.1f min (%.1f GB/s of image bytes)" % (name, np.mean(ms) * 1e3, np.min(ms) * 1e3, W * H * 4 / np.mean(ms) / 1e6), flush=True)
# check the results are the same
ctx.stage_filter_device(pinned.ctypes.data, W, H)
a = ctx.download(L.BUF_DX).copy()
ctx.stage_filter_device(d, W, H)
b = ctx.download(L.BUF_DX)
print("identical:", bool((a == b).all()))
