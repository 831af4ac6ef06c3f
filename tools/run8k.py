import os, sys; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, time
import librectify_amd as L
from librectify_amd import synth
W = H = 8192
t = time.time(); img = synth.frame(W, H, 7, bars=6000, tile=512); print("gen %.1fs" % (time.time() - t), flush=True)
ctx = L.Context(0)
ctx.set_stage_timing(True)
ctx.set_seed(0)
for est in (0, 1, 3):
    ctx.set_estimator(est, 100000 if est == 1 else 128)
    for rep in range(2):
        t = time.time(); got = ctx.find_line_segment_groups(img, 20.0, capacity=200000); dt = time.time() - t
        print("estimator %d: %dx%d total %.1f ms (host buffer incl. H2D), lines %d, groups %s" % (est, W, H, dt * 1e3, len(got), np.bincount(got["group_id"] + 1).tolist()), ctx.stage_counters(), ctx.stage_times().round(3), flush=True)
T = L.compute_rectification_transform(got, W, H).as_array()
print(T)
