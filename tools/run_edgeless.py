"""The edge-less 4K frame of bench.py's worst_case leg (smooth ramp + sinusoid, blurred noise): stage times and counters;
with LIBRECTIFY_FLOOD_DEBUG=1 the per-round statistics."""
import os, sys; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, time
import librectify_amd as L
from librectify_amd import synth
W, H = 3840, 2160
rng = np.random.RandomState(77)
yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
ramp = 0.3 + 0.3 * xx / W + 0.1 * np.sin(yy / 300.0) + rng.normal(0, 0.002, size=(H, W))
img = synth._gauss_blur(ramp, 2.0).astype(np.float32)
ctx = L.Context(0)
ctx.set_stage_timing(True)
for rep in range(3):
    t = time.time(); got = ctx.find_line_segment_groups(img, max(W, H) / 100.0); dt = time.time() - t
    print("total %.1f ms" % (dt * 1e3), "lines", len(got), ctx.stage_counters(), ctx.stage_times().round(3), flush=True)
