"""BASELINE config 4 on one GPU: a batch of 1920x1080 frames through the batch entry point (device-resident)."""
import os, sys; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, time
import librectify_amd as L
from librectify_amd import synth
W, H = 1920, 1080
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
distinct = 32
t = time.time(); frames = np.stack([synth.frame(W, H, 1000 + i) for i in range(distinct)]); print("gen %.1fs" % (time.time() - t), flush=True)
ctx = L.Context(0)
ctx.set_seed(0)
ctx.set_batch_streams(16)
d = ctx.device_upload(np.concatenate([frames] * (B // distinct)))
for rep in range(3):
    t = time.time()
    out, n, tf = ctx.find_line_segment_groups_batch_device(d, W * H, B, W, H, max(W, H) / 100.0, capacity=2048)
    dt = time.time() - t
    print("%d frames %dx%d: %.1f ms, %.0f frames/s, %.0f Mpix/s, mean segments %.0f" % (B, W, H, dt * 1e3, B / dt, B * W * H / dt / 1e6, float(np.mean(n))), flush=True)
ctx.device_free(d)
