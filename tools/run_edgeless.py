"""The ramp frame of bench.py's worst_case leg (synth.ramp_frame: smooth ramp + slow wave, blurred noise): stage times and counters;
with LIBRECTIFY_FLOOD_DEBUG=1 the per-round statistics."""
import os, sys; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, time
import librectify_amd as L
from librectify_amd import synth
W, H = 3840, 2160
img = synth.ramp_frame(W, H)
ctx = L.Context(0)
ctx.set_stage_timing(True)
for rep in range(3):
    t = time.time(); got = ctx.find_line_segment_groups(img, max(W, H) / 100.0); dt = time.time() - t
    print("total %.1f ms" % (dt * 1e3), "lines", len(got), ctx.stage_counters(), ctx.stage_times().round(3), flush=True)
