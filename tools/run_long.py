"""Stress frame for the flood's overflow path: a 4K frame whose bars run across most of the frame (edges of 2000-3600 px,
i.e. walks of several hundred tiles that leave the per-wave LDS table and continue in the global slabs)."""
import os, sys; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, time
import librectify_amd as L
from librectify_amd import synth

def long_frame(W, H, seed, K=60):
    rng = np.random.RandomState(seed)
    img = np.full((H, W), 0.5, np.float64)
    for _ in range(K):
        c = np.array([rng.uniform(0.3, 0.7) * W, rng.uniform(0.1, 0.9) * H])
        ang = rng.uniform(-0.25, 0.25) + (np.pi / 2 if rng.rand() < 0.3 else 0.0)
        d = np.array([np.cos(ang), np.sin(ang)]); nrm = np.array([-d[1], d[0]])
        length = rng.uniform(0.5, 0.95) * (W if abs(d[0]) > 0.7 else H)
        half_w = rng.uniform(3.0, 12.0)
        contrast = rng.uniform(0.1, 0.4) * (1 if rng.rand() < 0.5 else -1)
        yy, xx = np.mgrid[0:H, 0:W]
        px, py = xx - c[0], yy - c[1]
        m = (np.abs(px * d[0] + py * d[1]) <= length / 2) & (np.abs(px * nrm[0] + py * nrm[1]) <= half_w)
        img[m] += contrast
    img = synth._gauss_blur(np.clip(img, 0, 1), 1.0) + rng.normal(0, 0.005, size=img.shape)
    return img.astype(np.float32)

W, H = 3840, 2160
img = long_frame(W, H, 3)
ctx = L.Context(0)
ctx.set_stage_timing(True)
for rep in range(3):
    t = time.time(); got = ctx.find_line_segment_groups(img, max(W, H) / 100.0); dt = time.time() - t
    print(W, H, "total %.1f ms" % (dt * 1e3), "lines", len(got), ctx.stage_counters(), ctx.stage_times().round(3), flush=True)
if len(sys.argv) > 1:
    from tests import oracle_lib as O
    ref, _ = O.find_line_segment_groups(img, max(W, H) / 100.0, seed=0)
    a = np.frombuffer(np.ascontiguousarray(got).tobytes(), np.uint32).reshape(len(got), 7)
    b = np.frombuffer(np.ascontiguousarray(ref).tobytes(), np.uint32).reshape(len(ref), 7)
    print("oracle lines", len(ref), "identical", len(got) == len(ref) and bool((a == b).all()))
    print("longest", max(np.hypot(got["x2"] - got["x1"], got["y2"] - got["y1"])))
