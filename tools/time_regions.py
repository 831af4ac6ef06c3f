"""Timing of the region frames of tools/soak_regions.py through the frame call (second repetition), with the flood's counters."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
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



ctx = L.Context(0)
ctx.set_stage_timing(True)
sizes = [(960, 540), (1283, 717), (1920, 1080), (2051, 1153), (3840, 2160)]
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    W, H = sizes[i % len(sizes)]
    img = regions(W, H, 500 + i)
    for rep in range(2):
        t = time.time(); got = ctx.find_line_segment_groups(img, max(W, H) / 100.0); dt = time.time() - t
    c = ctx.stage_counters()
    print("frame %d %dx%d: %.2f ms, flood %.2f ms, %d lines, seeds %d, rounds %d, second tier %d, slabs %d, tail %d" % (i, W, H, dt * 1e3, ctx.stage_times()[3], len(got), c["seeds"], c["flood_rounds"], c["second_tier_seeds"], c["slabs"], c["ordered_tail_seeds"]), flush=True)
