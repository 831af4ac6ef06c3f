"""Looks for content the pipeline is slow on: frames of several synthetic kinds at 1920x1080 through the frame call (second
repetition), wall time, flood time, rounds and the flood's counters."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import librectify_amd as L
from librectify_amd import synth

W, H = 1920, 1080
yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
rng = np.random.RandomState(1)


def blur(a, s):
    return synth._gauss_blur(a, s)


kinds = {
    "bars": lambda: synth.frame(W, H, 5),
    "white noise 0.05": lambda: (0.5 + rng.normal(0, 0.05, (H, W))).astype(np.float32),
    "white noise 0.3": lambda: np.clip(0.5 + rng.normal(0, 0.3, (H, W)), 0, 1).astype(np.float32),
    "blurred noise s=3": lambda: blur(0.5 + rng.normal(0, 0.2, (H, W)), 3.0).astype(np.float32),
    "blurred noise s=8": lambda: blur(0.5 + rng.normal(0, 0.5, (H, W)), 8.0).astype(np.float32),
    "checkerboard 16": lambda: ((((xx // 16) + (yy // 16)) % 2) * 0.6 + 0.2).astype(np.float32),
    "checkerboard 64 blurred": lambda: blur((((xx // 64) + (yy // 64)) % 2) * 0.6 + 0.2, 1.5).astype(np.float32),
    "stripes period 6": lambda: (0.5 + 0.4 * np.sin(xx * 2 * np.pi / 6)).astype(np.float32),
    "stripes period 40 diagonal": lambda: (0.5 + 0.4 * np.sin((xx + 0.5 * yy) * 2 * np.pi / 40)).astype(np.float32),
    "concentric circles": lambda: (0.5 + 0.4 * np.sin(np.hypot(xx - W / 2, yy - H / 2) / 12)).astype(np.float32),
    "one vertical edge": lambda: blur(np.where(xx > W / 2, 0.8, 0.2), 1.0).astype(np.float32),
    "one edge + noise": lambda: (blur(np.where(xx > W / 2 + 0.1 * yy, 0.8, 0.2), 1.0) + rng.normal(0, 0.01, (H, W))).astype(np.float32),
    "constant": lambda: np.full((H, W), 0.5, np.float32),
    "ramp": lambda: synth.ramp_frame(W, H, 3),
    "regions": lambda: synth.region_frame(W, H, 500),
    "radial gradient": lambda: (1.0 - np.hypot(xx - W / 2, yy - H / 2) / np.hypot(W / 2, H / 2)).astype(np.float32),
    "radial gradient + noise": lambda: (1.0 - np.hypot(xx - W / 2, yy - H / 2) / np.hypot(W / 2, H / 2) + rng.normal(0, 0.003, (H, W))).astype(np.float32),
    "text-like strokes": lambda: blur((rng.rand(H // 8, W // 8) > 0.7).astype(np.float64).repeat(8, 0).repeat(8, 1) * 0.7 + 0.15, 0.8).astype(np.float32),
    "long bars": lambda: synth.long_bar_frame(W, H, 3, K=30),
    "grid of thin lines": lambda: blur(np.where(((xx % 48) < 2) | ((yy % 48) < 2), 0.9, 0.2), 0.7).astype(np.float32),
}
ctx = L.Context(0)
ctx.set_stage_timing(True)
for name, make in kinds.items():
    img = np.ascontiguousarray(make())
    dts = []
    for rep in range(3):
        t = time.perf_counter()
        got = ctx.find_line_segment_groups(img, max(W, H) / 100.0)
        dts.append((time.perf_counter() - t) * 1e3)
    c = ctx.stage_counters()
    st = ctx.stage_times()
    print("%-28s wall %7.2f ms (first %7.2f)  flood %6.2f  lines %5d  seeds %7d comps %6d rounds %3d tier2 %5d slabs %3d tail %5d held %4d laps %d" % (
        name, min(dts[1:]), dts[0], float(st[L.T_FLOOD]), len(got), c["seeds"], c["components"], c["flood_rounds"], c["second_tier_seeds"], c["slabs"],
        c["ordered_tail_seeds"], c["giants_held"], c["frame_laps"]), flush=True)
