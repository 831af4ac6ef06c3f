"""Does the H2D traffic itself slow the frames' kernels down?  HBM-resident batches (no upload in the pipeline) with and
without a second context copying 33 MB page-locked buffers to the device back to back on its own stream."""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import librectify_amd as L
from librectify_amd import synth

W, H, B = 3840, 2160, 32
frames = np.stack([synth.frame(W, H, 1 + (i % 4)) for i in range(B)])
ctx = L.Context(0)
ctx.set_seed(0)
d = ctx.device_upload(frames)
ctx2 = L.Context(0)
pin = ctx2.host_alloc((H, W))
pin[:] = frames[0]
dst = ctx2.device_upload(frames[0])
stop = False
copied = [0]


def pump():
    import ctypes as C
    while not stop:
        L._check(L.lib().lr_memcpy_h2d(ctx2._h, C.c_void_p(dst), L._ptr(pin), pin.nbytes))
        copied[0] += 1


def rate(tag):
    ts = []
    for rep in range(4):
        t0 = time.perf_counter()
        ctx.find_line_segment_groups_batch_device(d, W * H, B, W, H, 38.4)
        ts.append(time.perf_counter() - t0)
    c0 = copied[0]
    print("%s: %.0f Mpix/s (best of %s ms)" % (tag, B * W * H / min(ts[1:]) / 1e6, np.round(np.array(ts) * 1e3, 1)), flush=True)


rate("HBM-resident, link idle")
th = threading.Thread(target=pump)
t0 = time.perf_counter()
th.start()
rate("HBM-resident, link busy")
stop = True
th.join()
dt = time.perf_counter() - t0
print("background copies: %.1f GB/s" % (copied[0] * pin.nbytes / dt / 1e9))
rate("HBM-resident, link idle again")
