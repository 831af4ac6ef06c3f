import os, sys; sys.path.insert(0, "/root/repo")
import numpy as np, ctypes as C
import librectify_amd as L
from librectify_amd import synth
ctx = L.Context(0)
ctx.set_stage_timing(True)
img = synth.frame(3840, 2160, 1)
for rep in range(3):
    got = ctx.find_line_segment_groups(img, 38.4)
print(ctx.stage_times().round(3))
