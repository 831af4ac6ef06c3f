"""Single-frame stage times of several 4K synthetic frames (seeds given on the command line; default 1 2 3 4)."""
import os, sys; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, time
import librectify_amd as L
from librectify_amd import synth
ctx = L.Context(0)
ctx.set_stage_timing(True)
W, H = 3840, 2160
for seed in [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4]:
    img = synth.frame(W, H, seed)
    for rep in range(2):
        t = time.time(); got = ctx.find_line_segment_groups(img, max(W, H) / 100.0); dt = time.time() - t
    print("seed", seed, "total %.1f ms" % (dt * 1e3), "lines", len(got), ctx.stage_counters(), ctx.stage_times().round(3), flush=True)
