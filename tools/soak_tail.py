"""One-off soak check of the exhausted-storage paths with partial commits: long-edge frames (walks of several hundred tiles)
in the flood's test modes 2, 3 and 5, where the rounds stall and the ordered tail finishes -- also for seeds that already
own pixels.  Label image against the oracle; prints how many seeds the tail took."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import oracle_lib as O
import librectify_amd as L
from librectify_amd import synth


def long_frame(W, H, seed, K):
    rng = np.random.RandomState(seed)
    img = np.full((H, W), 0.5, np.float64)
    yy, xx = np.mgrid[0:H, 0:W]
    for _ in range(K):
        c = np.array([rng.uniform(0.3, 0.7) * W, rng.uniform(0.1, 0.9) * H])
        ang = rng.uniform(-0.25, 0.25) + (np.pi / 2 if rng.rand() < 0.3 else 0.0)
        d = np.array([np.cos(ang), np.sin(ang)]); nrm = np.array([-d[1], d[0]])
        length = rng.uniform(0.5, 0.95) * (W if abs(d[0]) > 0.7 else H)
        half_w = rng.uniform(3.0, 12.0)
        contrast = rng.uniform(0.1, 0.4) * (1 if rng.rand() < 0.5 else -1)
        px, py = xx - c[0], yy - c[1]
        m = (np.abs(px * d[0] + py * d[1]) <= length / 2) & (np.abs(px * nrm[0] + py * nrm[1]) <= half_w)
        img[m] += contrast
    img = synth._gauss_blur(np.clip(img, 0, 1), 1.0) + rng.normal(0, 0.005, size=img.shape)
    return img.astype(np.float32)


ctx = L.Context(0)
bad = 0
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    W, H = (2600, 1500) if i % 2 == 0 else (2597, 1503)  # (every other frame ragged in both directions: walks along the right and bottom borders)
    img = long_frame(W, H, 100 + i, 14)
    ref = O.find_line_segments(img, num_threads=8)
    for mode in (1, 2, 3, 5, 6, 7):
        for part in (1, 0):
            ctx.set_flood_partial_commits(part)
            ctx.set_flood_mode(mode)
            ctx.stage_filter_host(img); ctx.stage_seeds(); ctx.stage_flood()
            ok = np.array_equal(ctx.download(L.BUF_LABEL), ref["label"])
            c = ctx.stage_counters()
            lines = ctx.stage_fit()
            ok = ok and lines.tobytes() == ref["lines"].tobytes()
            print("frame %d mode %d partial %d: %s  tail seeds %d, second tier %d, slabs %d, rounds %d" % (i, mode, part, "ok" if ok else "MISMATCH", c["ordered_tail_seeds"], c["second_tier_seeds"], c["slabs"], c["flood_rounds"]), flush=True)
            bad += 0 if ok else 1
print("mismatches:", bad)
sys.exit(1 if bad else 0)
