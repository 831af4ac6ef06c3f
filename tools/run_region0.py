"""Region frame 0 of tools/time_regions.py (960x540): twenty rounds, every seed in the second tier -- per-round statistics
with LIBRECTIFY_FLOOD_DEBUG=1."""
import os, sys; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, time
import librectify_amd as L
from librectify_amd import synth
def regions(W, H, seed):
    rng = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    img = np.full((H, W), 0.4, np.float64)
    for _ in range(rng.randint(6, 30)):
        cx, cy, r = rng.uniform(0, W), rng.uniform(0, H), rng.uniform(30, 0.2 * W)
        img += rng.uniform(0.05, 0.3) * np.exp(-(((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * r * r)))
    img += rng.uniform(0, 0.3) * xx / W + rng.uniform(0, 0.2) * yy / H
    img = synth._gauss_blur(np.clip(img, 0, 1), rng.uniform(1.0, 3.0)) + rng.normal(0, rng.uniform(0.001, 0.005), size=img.shape)
    return img.astype(np.float32)
W, H = 960, 540
img = regions(W, H, 500)
ctx = L.Context(0)
ctx.set_stage_timing(True)
for rep in range(2):
    sys.stderr.write("---- pass %d\n" % rep)
    t = time.time(); got = ctx.find_line_segment_groups(img, max(W, H) / 100.0); dt = time.time() - t
print("total %.1f ms" % (dt * 1e3), "lines", len(got), ctx.stage_counters(), ctx.stage_times().round(3), flush=True)
