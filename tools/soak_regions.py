"""One-off soak check of the flood on frames of REGIONS (soft blobs and ramps: wide walks, the second tier's teams, the
early hand-over) in modes 1, 6 and 7, with and without partial commits: label image and segment records against the oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import oracle_lib as O
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
bad = 0
sizes = [(960, 540), (1283, 717), (1920, 1080), (2051, 1153)]
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    W, H = sizes[i % len(sizes)]
    img = regions(W, H, 500 + i)
    ref = O.find_line_segments(img, num_threads=8)
    for mode in (1, 6, 7):
        for part in (1, 0):
            ctx.set_flood_partial_commits(part)
            ctx.set_flood_mode(mode)
            ctx.stage_filter_host(img); ctx.stage_seeds(); ctx.stage_flood()
            ok = np.array_equal(ctx.download(L.BUF_LABEL), ref["label"])
            c = ctx.stage_counters()
            lines = ctx.stage_fit()
            ok = ok and lines.tobytes() == ref["lines"].tobytes()
            print("frame %d %dx%d mode %d partial %d: %s  seeds %d, second tier %d, slabs %d, tail %d, rounds %d" % (i, W, H, mode, part, "ok" if ok else "MISMATCH", c["seeds"], c["second_tier_seeds"], c["slabs"], c["ordered_tail_seeds"], c["flood_rounds"]), flush=True)
            bad += 0 if ok else 1
print("mismatches:", bad)
sys.exit(1 if bad else 0)
