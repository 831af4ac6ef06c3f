"""synth.region_frame(1920, 1080, 500): a frame of soft blobs the flood is slow on (tools/latency_fuzz.py) -- counters and times;
LIBRECTIFY_FLOOD_DEBUG=1 for the rounds."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import librectify_amd as L
from librectify_amd import synth
W, H = 1920, 1080
img = synth.region_frame(W, H, int(sys.argv[1]) if len(sys.argv) > 1 else 500)
ctx = L.Context(0)
ctx.set_stage_timing(True)
for rep in range(3):
    t = time.perf_counter(); got = ctx.find_line_segment_groups(img, 19.2); dt = (time.perf_counter() - t) * 1e3
print("wall %.2f ms" % dt, ctx.stage_counters(), ctx.stage_times().round(2))
